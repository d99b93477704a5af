// Stand-in for LibStevi's correlation/image_based_refinement.h -- NOT an implementation of it.
//
// The reference header (2 795 lines) holds the FEATURE-domain sub-pixel refinement: barycentric / symmetric-barycentric / subpart /
// arbitrary-interpolation refinement of 1-D and 2-D disparities, each solving a small Eigen least-squares (or L1, least-median,
// spherical) problem per pixel on interpolated feature vectors (correlation/image_based_refinement.h:31-2795, through
// MatchingFunctionTraits<f>::barycentricBestApproximation and optimization/*.h).  That is outside the accelerated path (SURVEY.md
// section 8: the path ends with cost_based_refinement.h; section 2 lists this header out of scope), and there is deliberately no CPU
// fallback in this tree.
//
// What this file is for: translation units that include the header for completeness and call none of it keep compiling when the shim
// directory comes first on the include path -- examples/stereo-match/main.cpp:30 is one (its only use, refineBarycentric2dDisp at
// :202-203, is commented out upstream).  Every function name of the reference header is declared here as a template whose body is a
// static_assert, so that a call site gets one readable message ("has no GPU path") instead of an undeclared-identifier cascade -- and
// can never silently run something else.  A project that needs these functions includes the reference's own header for them (they only
// take and return Multidim::Array, so they compose with the arrays this tree returns).
#ifndef STEREOVISION_IMAGE_BASED_REFINEMENT_H
#define STEREOVISION_IMAGE_BASED_REFINEMENT_H

#include "./cross_correlations.h"
#include "./matching_costs.h"

namespace StereoVision {
namespace Correlation {

namespace HipBridge {
template <matchingFunctions> inline constexpr bool featureDomainRefinementOnGpuPath = false;
}

// The reference's entry points come in two template shapes: <matchFunc, non-type policies...> on feature volumes, and
// <matchFunc, T_L, T_R, non-type policies...> on images.  A disparity map refined in 1-D is (H, W), in 2-D (H, W, 2).
#define SVH_NOT_ON_GPU_PATH(NAME, RANK)                                                                                              \
    template <matchingFunctions matchFunc, auto... policies, class... Args> Multidim::Array<float, RANK> NAME(Args const &...) {     \
        static_assert(HipBridge::featureDomainRefinementOnGpuPath<matchFunc>,                                                        \
                      "libstevi_hip: " #NAME " (correlation/image_based_refinement.h, feature-domain refinement) has no GPU path and "  \
                      "there is no CPU fallback; the cost-domain refinement of correlation/cost_based_refinement.h is the one on the path"); \
        return Multidim::Array<float, RANK>();                                                                                       \
    }
#define SVH_NOT_ON_GPU_PATH_IMG(NAME, RANK)                                                                                          \
    template <matchingFunctions matchFunc, class T_L, class T_R, auto... policies, class... Args>                                    \
    Multidim::Array<float, RANK> NAME(Args const &...) {                                                                             \
        static_assert(HipBridge::featureDomainRefinementOnGpuPath<matchFunc>,                                                        \
                      "libstevi_hip: " #NAME " (correlation/image_based_refinement.h, feature-domain refinement) has no GPU path and "  \
                      "there is no CPU fallback; the cost-domain refinement of correlation/cost_based_refinement.h is the one on the path"); \
        return Multidim::Array<float, RANK>();                                                                                       \
    }

// on feature volumes, 1-D disparities (image_based_refinement.h:31-673, :1591-1690)
SVH_NOT_ON_GPU_PATH(refineBarycentricSymmetricDisp, 2)
SVH_NOT_ON_GPU_PATH(refineSubpartBarycentricSymmetricDisp, 2)
SVH_NOT_ON_GPU_PATH(refineBarycentricDisp, 2)
SVH_NOT_ON_GPU_PATH(refineArbitraryInterpolationDisp, 2)
SVH_NOT_ON_GPU_PATH(refineSubpartBarycentricDisp, 2)
SVH_NOT_ON_GPU_PATH(refineCostSymmetricDisp, 2)
// on feature volumes, 2-D disparities (:675-1589)
SVH_NOT_ON_GPU_PATH(refineBarycentric2dDisp, 3)
SVH_NOT_ON_GPU_PATH(refineSubpartBarycentric2dDisp, 3)
SVH_NOT_ON_GPU_PATH(refineBarycentricSymmetric2dDisp, 3)
SVH_NOT_ON_GPU_PATH(refineSubpartBarycentricSymmetric2dDisp, 3)
SVH_NOT_ON_GPU_PATH(refineArbitraryInterpolation2dDisp, 3)
// matching + refinement from feature volumes (:1692-2424)
SVH_NOT_ON_GPU_PATH(refinedBarycentricSymmetricDispFeatureVol, 2)
SVH_NOT_ON_GPU_PATH(refinedSubpartBarycentricSymmetricDispFeatureVol, 2)
SVH_NOT_ON_GPU_PATH(refinedBarycentricDispFeatureVol, 2)
SVH_NOT_ON_GPU_PATH(refinedSubpartBarycentricDispFeatureVol, 2)
SVH_NOT_ON_GPU_PATH(refinedCostSymmetricDispFeatureVol, 2)
SVH_NOT_ON_GPU_PATH(refinedBarycentric2dDispFeatureVol, 3)
SVH_NOT_ON_GPU_PATH(refinedSubpartBarycentric2dDispFeatureVol, 3)
SVH_NOT_ON_GPU_PATH(refinedBarycentricSymmetric2dDispFeatureVol, 3)
SVH_NOT_ON_GPU_PATH(refinedSubpartBarycentricSymmetric2dDispFeatureVol, 3)
// matching + refinement from images (:2426-2795)
SVH_NOT_ON_GPU_PATH_IMG(refinedBarycentricSymmetricDisp, 2)
SVH_NOT_ON_GPU_PATH_IMG(refinedBarycentricDisp, 2)
SVH_NOT_ON_GPU_PATH_IMG(refinedCostSymmetricDisp, 2)
SVH_NOT_ON_GPU_PATH_IMG(refinedBarycentric2dDisp, 3)
SVH_NOT_ON_GPU_PATH_IMG(refinedBarycentricSymmetric2dDisp, 3)

#undef SVH_NOT_ON_GPU_PATH
#undef SVH_NOT_ON_GPU_PATH_IMG

} // namespace Correlation
} // namespace StereoVision

#endif // STEREOVISION_IMAGE_BASED_REFINEMENT_H

// Drop-in for LibStevi's correlation/cost_based_refinement.h: the 1-D refinement and the three 2-D variants (no Eigen needed:
// the 3x3 patch fit has a closed form on the device).
#ifndef STEREOVISION_COST_BASED_REFINEMENT_H
#define STEREOVISION_COST_BASED_REFINEMENT_H

#include <algorithm>
#include <cmath>

#include "./correlation_base.h"

namespace StereoVision {
namespace Correlation {

enum class InterpolationKernel { Equiangular, Parabola, Gaussian }; // cost_based_refinement.h:30-35
enum class IsotropyHypothesis { Isotropic, Anisotropic };

// refineCostTriplet<kernel>(c(-1), c(0), c(+1)), cost_based_refinement.h:43-69: the sub-sample offset of the extremum of the kernel's
// curve through three neighbouring costs, as an inline HOST function for callers that hold a triplet themselves (the map-sized
// refinement below runs the same three formulas on the GPU, svh_select_refine.hip; nothing in this tree routes a map through this one).
//   Equiangular: (c1 - cm1) / (2 a), a = the larger of |c0 - cm1|, |c1 - c0| carrying the sign of c0 - cm1
//   Parabola:    (cm1 - c1) / (2 (c1 - 2 c0 + cm1))
//   Gaussian:    the parabola through the logarithms
// No guard against a flat triplet: 0/0 gives NaN exactly as in the reference.
template <InterpolationKernel kernel> inline float refineCostTriplet(float cm1, float c0, float c1) {
    if constexpr (kernel == InterpolationKernel::Equiangular) {
        const float rise = c0 - cm1;
        const float alpha = std::copysign(1.f, rise) * std::max(std::fabs(rise), std::fabs(c1 - c0));
        return (c1 - cm1) / (2 * alpha);
    } else if constexpr (kernel == InterpolationKernel::Parabola) {
        return (cm1 - c1) / (2 * (c1 - 2 * c0 + cm1));
    } else {
        const float lm1 = std::log(cm1), l0 = std::log(c0), l1 = std::log(c1);
        return (lm1 - l1) / (2 * (l1 - 2 * l0 + lm1));
    }
}

// refineDispCostInterpolation<kernel>(truncatedCostVolume, rawDisparity), cost_based_refinement.h:128-163
template <InterpolationKernel kernel>
Multidim::Array<float, 2> refineDispCostInterpolation(Multidim::Array<float, 3> const &truncatedCostVolume, Multidim::Array<disp_t, 2> const &rawDisparity) {
    auto shape = rawDisparity.shape();
    int depth = truncatedCostVolume.shape()[2];
    int cv_radius = (depth - 1) / 2;
    if (cv_radius < 1 or 2 * cv_radius + 1 != depth) return Multidim::Array<float, 2>(); // :141-143
    auto refined = HipBridge::makeResult<Multidim::Array<float, 2>>(shape);
    if (refined.empty()) return refined;
    svh_array tcv = HipBridge::describe(truncatedCostVolume), raw = HipBridge::describe(rawDisparity), out = HipBridge::describe(refined);
    if (!HipBridge::check(svh_refine_disp_cost_interpolation(HipBridge::context(), static_cast<int>(kernel), &tcv, &raw, &out)))
        return Multidim::Array<float, 2>();
    return refined;
}

// the same on arrays in GPU memory (HipBridge::DeviceArray)
template <InterpolationKernel kernel>
DeviceArray<float, 2> refineDispCostInterpolation(DeviceArray<float, 3> const &truncatedCostVolume, DeviceArray<disp_t, 2> const &rawDisparity) {
    int depth = truncatedCostVolume.shape()[2];
    int cv_radius = (depth - 1) / 2;
    if (cv_radius < 1 or 2 * cv_radius + 1 != depth) return DeviceArray<float, 2>(); // :141-143
    DeviceArray<float, 2> refined(rawDisparity.shape());
    if (refined.empty()) return refined;
    svh_array tcv = HipBridge::describe(truncatedCostVolume), raw = HipBridge::describe(rawDisparity), out = HipBridge::describe(refined);
    if (!HipBridge::check(svh_refine_disp_cost_interpolation(HipBridge::context(), static_cast<int>(kernel), &tcv, &raw, &out))) return DeviceArray<float, 2>();
    return refined;
}

// refineDisp2dCostInterpolation<kernel, isotropHypothesis>(truncatedCostVolume, rawDisparity), cost_based_refinement.h:165-376
template <InterpolationKernel kernel, IsotropyHypothesis isotropHypothesis = IsotropyHypothesis::Isotropic>
Multidim::Array<float, 3> refineDisp2dCostInterpolation(Multidim::Array<float, 4> const &truncatedCostVolume, Multidim::Array<disp_t, 3> const &rawDisparity) {
    auto cv_shape = truncatedCostVolume.shape();
    int cv_radius0 = (cv_shape[2] - 1) / 2, cv_radius1 = (cv_shape[3] - 1) / 2;
    if (cv_radius0 < 1 or cv_radius1 < 1 or 2 * cv_radius0 + 1 != cv_shape[2] or 2 * cv_radius1 + 1 != cv_shape[3])
        return Multidim::Array<float, 3>(); // :180-182
    auto refined = HipBridge::makeResult<Multidim::Array<float, 3>>(rawDisparity.shape());
    if (refined.empty()) return refined;
    svh_array tcv = HipBridge::describe(truncatedCostVolume), raw = HipBridge::describe(rawDisparity), out = HipBridge::describe(refined);
    if (!HipBridge::check(svh_refine_disp_2d_cost_interpolation(HipBridge::context(), static_cast<int>(kernel), static_cast<int>(isotropHypothesis), &tcv,
                                                                &raw, &out)))
        return Multidim::Array<float, 3>();
    return refined;
}

// refineDisp2dCostPatchInterpolation<kernel>(truncatedCostVolume, rawDisparity), cost_based_refinement.h:378-436
template <InterpolationKernel kernel>
Multidim::Array<float, 3> refineDisp2dCostPatchInterpolation(Multidim::Array<float, 4> const &truncatedCostVolume, Multidim::Array<disp_t, 3> const &rawDisparity) {
    static_assert(kernel == InterpolationKernel::Parabola or kernel == InterpolationKernel::Gaussian,
                  "Unsupported kernel used for patch cost refinement"); // :83
    auto cv_shape = truncatedCostVolume.shape();
    int cv_radius0 = (cv_shape[2] - 1) / 2, cv_radius1 = (cv_shape[3] - 1) / 2;
    if (cv_radius0 < 1 or cv_radius1 < 1 or 2 * cv_radius0 + 1 != cv_shape[2] or 2 * cv_radius1 + 1 != cv_shape[3])
        return Multidim::Array<float, 3>(); // :393-395
    auto refined = HipBridge::makeResult<Multidim::Array<float, 3>>(rawDisparity.shape());
    if (refined.empty()) return refined;
    svh_array tcv = HipBridge::describe(truncatedCostVolume), raw = HipBridge::describe(rawDisparity), out = HipBridge::describe(refined);
    if (!HipBridge::check(svh_refine_disp_2d_cost_patch_interpolation(HipBridge::context(), static_cast<int>(kernel), &tcv, &raw, &out)))
        return Multidim::Array<float, 3>();
    return refined;
}

} // namespace Correlation
} // namespace StereoVision

#endif // STEREOVISION_COST_BASED_REFINEMENT_H

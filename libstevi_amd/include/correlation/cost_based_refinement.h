// Drop-in for LibStevi's correlation/cost_based_refinement.h (1-D part; the 2-D Eigen-based variants are out of scope).
#ifndef STEREOVISION_COST_BASED_REFINEMENT_H
#define STEREOVISION_COST_BASED_REFINEMENT_H

#include <cmath>

#include "./correlation_base.h"

namespace StereoVision {
namespace Correlation {

enum class InterpolationKernel { Equiangular, Parabola, Gaussian }; // cost_based_refinement.h:30-35
enum class IsotropyHypothesis { Isotropic, Anisotropic };

// refineDispCostInterpolation<kernel>(truncatedCostVolume, rawDisparity), cost_based_refinement.h:128-163
template <InterpolationKernel kernel>
Multidim::Array<float, 2> refineDispCostInterpolation(Multidim::Array<float, 3> const &truncatedCostVolume, Multidim::Array<disp_t, 2> const &rawDisparity) {
    auto shape = rawDisparity.shape();
    int depth = truncatedCostVolume.shape()[2];
    int cv_radius = (depth - 1) / 2;
    if (cv_radius < 1 or 2 * cv_radius + 1 != depth) return Multidim::Array<float, 2>(); // :141-143
    Multidim::Array<float, 2> refined(shape);
    if (refined.empty()) return refined;
    std::lock_guard<std::mutex> g(HipBridge::lock());
    svh_array tcv = HipBridge::describe(truncatedCostVolume), raw = HipBridge::describe(rawDisparity), out = HipBridge::describe(refined);
    if (!HipBridge::check(svh_refine_disp_cost_interpolation(HipBridge::context(), static_cast<int>(kernel), &tcv, &raw, &out)))
        return Multidim::Array<float, 2>();
    return refined;
}

} // namespace Correlation
} // namespace StereoVision

#endif // STEREOVISION_COST_BASED_REFINEMENT_H

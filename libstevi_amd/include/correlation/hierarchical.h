// Drop-in for LibStevi's correlation/hierarchical.h: OffsetedCostVolume, computeGuidedCV on feature volumes and both overloads of
// hiearchicalTruncatedCostVolume.  The pyramid never leaves the device (one C call per hiearchicalTruncatedCostVolume).
#ifndef STEREOVISION_HIEARCHICAL_H
#define STEREOVISION_HIEARCHICAL_H

#include <array>
#include <vector>

#include "../interpolation/downsampling.h"
#include "./cross_correlations.h"

namespace StereoVision {
namespace Correlation {

template <typename TCV> struct OffsetedCostVolume { // hierarchical.h:33-37
    Multidim::Array<TCV, 3> truncated_cost_volume;
    Multidim::Array<disp_t, 2> disp_estimate;
};

// computeGuidedCV<matchFunc, T_L, T_R, dDir, TCV>(feature_vol_l, feature_vol_r, disp_guide, upscale_disp_radius), hierarchical.h:74-229.
// The feature volumes are what getFeatureVolumeForMatchFunc returns: float, or uint32 census words.
template <matchingFunctions matchFunc, typename T_L, typename T_R, dispDirection dDir = dispDirection::RightToLeft, typename TCV = float>
OffsetedCostVolume<TCV> computeGuidedCV(Multidim::Array<T_L, 3> const &feature_vol_l, Multidim::Array<T_R, 3> const &feature_vol_r,
                                        Multidim::Array<disp_t, 2> disp_guide, disp_t upscale_disp_radius) {
    static_assert(HipBridge::onGpuPath<matchFunc>(), "libstevi_hip: this matching function has no GPU path");
    static_assert(std::is_same_v<TCV, float>, "libstevi_hip: cost volumes are float");
    auto l_shape = feature_vol_l.shape();
    auto r_shape = feature_vol_r.shape();
    if (l_shape[0] != r_shape[0]) return {Multidim::Array<TCV, 3>(), Multidim::Array<disp_t, 2>()}; // :90-92
    auto const &src = (dDir == dispDirection::RightToLeft) ? r_shape : l_shape;
    OffsetedCostVolume<TCV> ret = {HipBridge::makeResult<Multidim::Array<TCV, 3>>(src[0], src[1], 2 * upscale_disp_radius + 1), HipBridge::makeResult<Multidim::Array<disp_t, 2>>(src[0], src[1])};
    svh_array l = HipBridge::describe(feature_vol_l), r = HipBridge::describe(feature_vol_r), gd = HipBridge::describe(disp_guide);
    svh_array tcv = HipBridge::describe(ret.truncated_cost_volume), disp = HipBridge::describe(ret.disp_estimate);
    if (!HipBridge::check(svh_guided_cost_volume(HipBridge::context(), static_cast<int>(matchFunc), static_cast<int>(dDir), &l, &r, &gd, upscale_disp_radius,
                                                 &tcv, &disp)))
        return {Multidim::Array<TCV, 3>(), Multidim::Array<disp_t, 2>()};
    return ret;
}

// hiearchicalTruncatedCostVolume<matchFunc, depth, T_L, T_R, nImDim, dDir, TCV>(img_l, img_r, h_radiuses, v_radiuses, disp_width,
// upscale_disp_radius), hierarchical.h:232-294
template <matchingFunctions matchFunc, int depth, class T_L, class T_R, int nImDim = 2, dispDirection dDir = dispDirection::RightToLeft, class TCV = float>
OffsetedCostVolume<TCV> hiearchicalTruncatedCostVolume(Multidim::Array<T_L, nImDim> const &img_l, Multidim::Array<T_R, nImDim> const &img_r,
                                                       std::array<uint8_t, depth + 1> h_radiuses, std::array<uint8_t, depth + 1> v_radiuses,
                                                       disp_t disp_width, disp_t upscale_disp_radius = 2) {
    static_assert(depth > 0, "Minimum depth is 1"); // :243
    static_assert(HipBridge::onGpuPath<matchFunc>(), "libstevi_hip: this matching function has no GPU path");
    static_assert(std::is_same_v<TCV, float>, "libstevi_hip: cost volumes are float");
    auto l_shape = img_l.shape();
    auto r_shape = img_r.shape();
    if (l_shape[0] != r_shape[0]) return {Multidim::Array<TCV, 3>(), Multidim::Array<disp_t, 2>()};
    auto const &src = (dDir == dispDirection::RightToLeft) ? r_shape : l_shape;
    OffsetedCostVolume<TCV> ret = {HipBridge::makeResult<Multidim::Array<TCV, 3>>(src[0], src[1], 2 * upscale_disp_radius + 1), HipBridge::makeResult<Multidim::Array<disp_t, 2>>(src[0], src[1])};
    int32_t hr[depth + 1], vr[depth + 1];
    for (int i = 0; i < depth + 1; i++) {
        hr[i] = h_radiuses[i];
        vr[i] = v_radiuses[i];
    }
    svh_array l = HipBridge::describe(img_l), r = HipBridge::describe(img_r);
    svh_array tcv = HipBridge::describe(ret.truncated_cost_volume), disp = HipBridge::describe(ret.disp_estimate);
    if (!HipBridge::check(svh_hierarchical_truncated_cost_volume(HipBridge::context(), static_cast<int>(matchFunc), static_cast<int>(dDir), depth, &l, &r, hr,
                                                                 vr, disp_width, upscale_disp_radius, &tcv, &disp)))
        return {Multidim::Array<TCV, 3>(), Multidim::Array<disp_t, 2>()};
    return ret;
}

// single-radius overload, hierarchical.h:296-315
template <matchingFunctions matchFunc, int depth, class T_L, class T_R, int nImDim = 2, dispDirection dDir = dispDirection::RightToLeft, class TCV = float>
OffsetedCostVolume<TCV> hiearchicalTruncatedCostVolume(Multidim::Array<T_L, nImDim> const &img_l, Multidim::Array<T_R, nImDim> const &img_r, uint8_t h_radius,
                                                       uint8_t v_radius, disp_t disp_width, disp_t upscale_disp_radius = 2) {
    std::array<uint8_t, depth + 1> h_radiuses;
    std::array<uint8_t, depth + 1> v_radiuses;
    for (int i = 0; i < depth + 1; i++) {
        h_radiuses[i] = h_radius;
        v_radiuses[i] = v_radius;
    }
    return hiearchicalTruncatedCostVolume<matchFunc, depth, T_L, T_R, nImDim, dDir, TCV>(img_l, img_r, h_radiuses, v_radiuses, disp_width, upscale_disp_radius);
}

} // namespace Correlation
} // namespace StereoVision

#endif // STEREOVISION_HIEARCHICAL_H

// Drop-in for LibStevi's correlation/cross_correlations.h (1-D disparity path).
#ifndef STEREOVISION_CROSS_CORRELATIONS_H
#define STEREOVISION_CROSS_CORRELATIONS_H

#include "./census.h"
#include "./correlation_base.h"
#include "./matching_costs.h"
#include "./unfold.h"

namespace StereoVision {
namespace Correlation {

namespace HipBridge {
inline void rangeOf(disp_t width, int32_t &lower, int32_t &count) {
    lower = 0;
    count = width;
}
inline void rangeOf(searchOffset<1> const &r, int32_t &lower, int32_t &count) {
    lower = r.lowerOffset(0);
    count = r.dimRange(0);
}
} // namespace HipBridge

// featureVolume2CostVolume<matchFunc, T_L, T_R, SearchRangeType, dDir, TCV>, cross_correlations.h:724-738: disp_t / searchOffset<1>
// give a (row, col, disparity) volume (aggregateCost :194-308), searchOffset<2> a (row, col, dh, dw) one (:310-374)
template <matchingFunctions matchFunc, class T_L, class T_R, typename SearchRangeType, dispDirection dDir = dispDirection::RightToLeft, typename TCV = float>
inline Multidim::Array<TCV, searchRangeTypeInfos<SearchRangeType>::CostVolumeDims>
featureVolume2CostVolume(Multidim::Array<T_L, 3> const &feature_vol_l, Multidim::Array<T_R, 3> const &feature_vol_r, SearchRangeType searchRange) {
    static_assert(HipBridge::onGpuPath<matchFunc>(), "libstevi_hip: this matching function has no GPU path");
    static_assert(std::is_same_v<TCV, float>, "libstevi_hip: cost volumes are float");
    constexpr int nCvDims = searchRangeTypeInfos<SearchRangeType>::CostVolumeDims;
    auto const &src = (dDir == dispDirection::RightToLeft) ? feature_vol_r.shape() : feature_vol_l.shape();
    if constexpr (nCvDims == 3) {
        int32_t lower, count;
        HipBridge::rangeOf(searchRange, lower, count);
        if (feature_vol_l.shape()[0] != feature_vol_r.shape()[0] || count <= 0) return Multidim::Array<TCV, 3>(0, 0, 0); // :209-211
        // aggregateCost's own layout (cross_correlations.h:220) is {w*D, 1, w}; the GPU's native one is dense (row, col, disparity)
        Multidim::Array<TCV, 3> cv(src[0], src[1], count);
        std::lock_guard<std::mutex> g(HipBridge::lock());
        svh_array l = HipBridge::describe(feature_vol_l), r = HipBridge::describe(feature_vol_r), out = HipBridge::describe(cv);
        if (!HipBridge::check(svh_feature_cost_volume(HipBridge::context(), static_cast<int>(matchFunc), static_cast<int>(dDir), &l, &r, lower, count, &out)))
            return Multidim::Array<TCV, 3>(0, 0, 0);
        return cv;
    } else {
        static_assert(nCvDims == 4, "libstevi_hip: disp_t, searchOffset<1> or searchOffset<2>");
        const int Dh = searchRange.dimRange(0), Dw = searchRange.dimRange(1);
        if (feature_vol_l.shape()[0] != feature_vol_r.shape()[0] || Dh <= 0 || Dw <= 0) return Multidim::Array<TCV, 4>(); // :324-326, :338-340
        Multidim::Array<TCV, 4> cv(src[0], src[1], Dh, Dw);
        std::lock_guard<std::mutex> g(HipBridge::lock());
        svh_array l = HipBridge::describe(feature_vol_l), r = HipBridge::describe(feature_vol_r), out = HipBridge::describe(cv);
        if (!HipBridge::check(svh_feature_cost_volume_2d(HipBridge::context(), static_cast<int>(matchFunc), static_cast<int>(dDir), &l, &r,
                                                         searchRange.template lowerOffset<0>(), searchRange.template upperOffset<0>(),
                                                         searchRange.template lowerOffset<1>(), searchRange.template upperOffset<1>(), &out)))
            return Multidim::Array<TCV, 4>();
        return cv;
    }
}

// unfoldBasedCostVolume<matchFunc, T_L, T_R, nImDim, dDir, TCV>(img_l, img_r, h_radius, v_radius, disp_width), cross_correlations.h:740-765
template <matchingFunctions matchFunc, class T_L, class T_R, int nImDim = 2, dispDirection dDir = dispDirection::RightToLeft, typename TCV = float>
Multidim::Array<TCV, 3> unfoldBasedCostVolume(Multidim::Array<T_L, nImDim> const &img_l, Multidim::Array<T_R, nImDim> const &img_r, uint8_t h_radius,
                                              uint8_t v_radius, disp_t disp_width) {
    static_assert(HipBridge::onGpuPath<matchFunc>(), "libstevi_hip: this matching function has no GPU path");
    static_assert(std::is_same_v<TCV, float>, "libstevi_hip: cost volumes are float");
    auto l_shape = img_l.shape();
    auto r_shape = img_r.shape();
    if (l_shape[0] != r_shape[0] || disp_width <= 0) return Multidim::Array<TCV, 3>(0, 0, 0); // :751-753
    if (nImDim == 3 && l_shape[nImDim - 1] != r_shape[nImDim - 1]) return Multidim::Array<TCV, 3>(0, 0, 0); // :755-759
    auto const &src = (dDir == dispDirection::RightToLeft) ? r_shape : l_shape;
    Multidim::Array<TCV, 3> cv(src[0], src[1], disp_width);
    std::lock_guard<std::mutex> g(HipBridge::lock());
    svh_array l = HipBridge::describe(img_l), r = HipBridge::describe(img_r), out = HipBridge::describe(cv);
    if (!HipBridge::check(svh_unfold_cost_volume(HipBridge::context(), static_cast<int>(matchFunc), static_cast<int>(dDir), &l, &r, h_radius, v_radius, 0,
                                                 disp_width, &out)))
        return Multidim::Array<TCV, 3>(0, 0, 0);
    return cv;
}

// unfoldBased2dDisparityCostVolume<matchFunc, T_L, T_R, nImDim, dDir, TCV>(img_l, img_r, h_radius, v_radius, searchOffset<2>),
// cross_correlations.h:794-822
template <matchingFunctions matchFunc, class T_L, class T_R, int nImDim = 2, dispDirection dDir = dispDirection::RightToLeft, typename TCV = float>
Multidim::Array<TCV, 4> unfoldBased2dDisparityCostVolume(Multidim::Array<T_L, nImDim> const &img_l, Multidim::Array<T_R, nImDim> const &img_r,
                                                         uint8_t h_radius, uint8_t v_radius, searchOffset<2> const &searchWindows) {
    static_assert(HipBridge::onGpuPath<matchFunc>(), "libstevi_hip: this matching function has no GPU path");
    static_assert(std::is_same_v<TCV, float>, "libstevi_hip: cost volumes are float");
    auto l_shape = img_l.shape();
    auto r_shape = img_r.shape();
    if (l_shape[0] != r_shape[0] || l_shape[1] != r_shape[1]) return Multidim::Array<TCV, 4>(); // :804-810
    if (nImDim == 3 && l_shape[nImDim - 1] != r_shape[nImDim - 1]) return Multidim::Array<TCV, 4>();
    const int Dh = searchWindows.dimRange(0), Dw = searchWindows.dimRange(1);
    if (Dh <= 0 || Dw <= 0) return Multidim::Array<TCV, 4>(); // :338-340
    Multidim::Array<TCV, 4> cv(l_shape[0], l_shape[1], Dh, Dw);
    std::lock_guard<std::mutex> g(HipBridge::lock());
    svh_array l = HipBridge::describe(img_l), r = HipBridge::describe(img_r), out = HipBridge::describe(cv);
    if (!HipBridge::check(svh_unfold_cost_volume_2d(HipBridge::context(), static_cast<int>(matchFunc), static_cast<int>(dDir), &l, &r, h_radius, v_radius,
                                                    searchWindows.lowerOffset(0), searchWindows.upperOffset(0), searchWindows.lowerOffset(1),
                                                    searchWindows.upperOffset(1), &out)))
        return Multidim::Array<TCV, 4>();
    return cv;
}

} // namespace Correlation
} // namespace StereoVision

#endif // STEREOVISION_CROSS_CORRELATIONS_H

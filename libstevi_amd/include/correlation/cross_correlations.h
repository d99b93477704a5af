// Drop-in for LibStevi's correlation/cross_correlations.h (1-D disparity path).
#ifndef STEREOVISION_CROSS_CORRELATIONS_H
#define STEREOVISION_CROSS_CORRELATIONS_H

#include "./census.h"
#include "./correlation_base.h"
#include "./matching_costs.h"
#include "./unfold.h"

#include "../utils/contiguity.h" // as the reference does (cross_correlations.h:29)

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

// ---- per-pixel statistics and feature-volume transforms (rows A7 / A8), float volumes ------------------------------------
namespace HipBridge {
template <class Fn, class... Maps> inline Multidim::Array<float, 2> mapOf(Fn fn, svh_array in, int h, int w, Maps... maps) {
    auto out = HipBridge::makeResult<Multidim::Array<float, 2>>(h, w);
    if (out.empty()) return out;
    svh_array o = describe(out);
    check(fn(context(), &in, maps..., &o));
    return out;
}
template <class Fn, class... Maps> inline Multidim::Array<float, 3> volumeOf(Fn fn, svh_array in, int h, int w, int f, Maps... maps) {
    auto out = HipBridge::makeResult<Multidim::Array<float, 3>>(h, w, f);
    if (out.empty()) return out;
    svh_array o = describe(out);
    check(fn(context(), &in, maps..., &o));
    return out;
}
} // namespace HipBridge

// channelsZeroMeanNorm(in_data, mean) and channelsZeroMeanNorm(in_data), cross_correlations.h:61-122
template <class T_I, class T_M, class T_O = float, Multidim::ArrayDataAccessConstness Cd, Multidim::ArrayDataAccessConstness Cm>
Multidim::Array<T_O, 2> channelsZeroMeanNorm(Multidim::Array<T_I, 3, Cd> const &in_data, Multidim::Array<T_M, 2, Cm> const &mean) {
    static_assert(std::is_same_v<T_O, float>, "libstevi_hip: float statistics");
    svh_array m = HipBridge::describe(mean);
    return HipBridge::mapOf(svh_channels_zero_mean_norm, HipBridge::describe(in_data), in_data.shape()[0], in_data.shape()[1], (const svh_array *)&m);
}
template <class T_I, class T_M = float, class T_O = float, Multidim::ArrayDataAccessConstness C>
Multidim::Array<T_O, 2> channelsZeroMeanNorm(Multidim::Array<T_I, 3, C> const &in_data) {
    static_assert(std::is_same_v<T_O, float>, "libstevi_hip: float statistics");
    return HipBridge::mapOf(svh_channels_zero_mean_norm, HipBridge::describe(in_data), in_data.shape()[0], in_data.shape()[1], (const svh_array *)nullptr);
}
// channelsNorm(in_data), cross_correlations.h:149-191
template <class T_I, class T_O = float, Multidim::ArrayDataAccessConstness C> Multidim::Array<T_O, 2> channelsNorm(Multidim::Array<T_I, 3, C> const &in_data) {
    static_assert(std::is_same_v<T_O, float>, "libstevi_hip: float statistics");
    return HipBridge::mapOf(svh_channels_norm, HipBridge::describe(in_data), in_data.shape()[0], in_data.shape()[1]);
}
// zeromeanNormalizedFeatureVolume(feature_vol, mean, norm), cross_correlations.h:416-462
template <class T_I, class T_M, class T_N, class T_O = float, Multidim::ArrayDataAccessConstness Cf, Multidim::ArrayDataAccessConstness Cm,
          Multidim::ArrayDataAccessConstness Cn>
inline Multidim::Array<T_O, 3> zeromeanNormalizedFeatureVolume(Multidim::Array<T_I, 3, Cf> const &feature_vol, Multidim::Array<T_M, 2, Cm> const &mean,
                                                               Multidim::Array<T_N, 2, Cn> const &norm) {
    static_assert(std::is_same_v<T_O, float>, "libstevi_hip: float feature volumes");
    svh_array m = HipBridge::describe(mean), n = HipBridge::describe(norm);
    auto shp = feature_vol.shape();
    return HipBridge::volumeOf(svh_zeromean_normalized_feature_volume, HipBridge::describe(feature_vol), shp[0], shp[1], shp[2], (const svh_array *)&m,
                               (const svh_array *)&n);
}
// normalizedFeatureVolume(feature_vol, norm), cross_correlations.h:504-550
template <class T_I, class T_N, class T_O = float, Multidim::ArrayDataAccessConstness Cf, Multidim::ArrayDataAccessConstness Cn>
inline Multidim::Array<T_O, 3> normalizedFeatureVolume(Multidim::Array<T_I, 3, Cf> const &feature_vol, Multidim::Array<T_N, 2, Cn> const &norm) {
    static_assert(std::is_same_v<T_O, float>, "libstevi_hip: float feature volumes");
    svh_array n = HipBridge::describe(norm);
    auto shp = feature_vol.shape();
    return HipBridge::volumeOf(svh_normalized_feature_volume, HipBridge::describe(feature_vol), shp[0], shp[1], shp[2], (const svh_array *)&n);
}
// zeromeanFeatureVolume(feature_vol, mean), cross_correlations.h:570-594
template <class T_I, class T_M, class T_O = float, Multidim::ArrayDataAccessConstness Cf, Multidim::ArrayDataAccessConstness Cm>
inline Multidim::Array<T_O, 3> zeromeanFeatureVolume(Multidim::Array<T_I, 3, Cf> const &feature_vol, Multidim::Array<T_M, 2, Cm> const &mean) {
    static_assert(std::is_same_v<T_O, float>, "libstevi_hip: float feature volumes");
    svh_array m = HipBridge::describe(mean);
    auto shp = feature_vol.shape();
    return HipBridge::volumeOf(svh_zeromean_feature_volume, HipBridge::describe(feature_vol), shp[0], shp[1], shp[2], (const svh_array *)&m);
}
// getFeatureVolumeForMatchFunc<matchFunc>(feature_vol), cross_correlations.h:645-722: float features, uint32 census words for CENSUS / HAMMING
template <matchingFunctions matchFunc, class T_I, Multidim::ArrayDataAccessConstness C,
          class FType = std::conditional_t<MatchingFunctionTraits<matchFunc>::isCensusBased, census_data_t, float>>
Multidim::Array<FType, 3> getFeatureVolumeForMatchFunc(Multidim::Array<T_I, 3, C> const &feature_vol) {
    static_assert(HipBridge::onGpuPath<matchFunc>(), "libstevi_hip: this matching function has no GPU path");
    auto shp = feature_vol.shape();
    constexpr bool census = MatchingFunctionTraits<matchFunc>::isCensusBased;
    if (census && shp[2] <= 1) return Multidim::Array<FType, 3>(); // census.h:76-78
    auto out = HipBridge::makeResult<Multidim::Array<FType, 3>>(shp[0], shp[1], census ? (shp[2] - 1) / 32 + 1 : shp[2]);
    if (out.empty()) return out;
    svh_array in = HipBridge::describe(feature_vol), o = HipBridge::describe(out);
    if (!HipBridge::check(svh_feature_volume_for_match_func(HipBridge::context(), static_cast<int>(matchFunc), &in, &o))) return Multidim::Array<FType, 3>();
    return out;
}

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
        auto cv = HipBridge::makeResult<Multidim::Array<TCV, 3>>(src[0], src[1], count);
        svh_array l = HipBridge::describe(feature_vol_l), r = HipBridge::describe(feature_vol_r), out = HipBridge::describe(cv);
        if (!HipBridge::check(svh_feature_cost_volume(HipBridge::context(), static_cast<int>(matchFunc), static_cast<int>(dDir), &l, &r, lower, count, &out)))
            return Multidim::Array<TCV, 3>(0, 0, 0);
        return cv;
    } else {
        static_assert(nCvDims == 4, "libstevi_hip: disp_t, searchOffset<1> or searchOffset<2>");
        const int Dh = searchRange.dimRange(0), Dw = searchRange.dimRange(1);
        if (feature_vol_l.shape()[0] != feature_vol_r.shape()[0] || Dh <= 0 || Dw <= 0) return Multidim::Array<TCV, 4>(); // :324-326, :338-340
        auto cv = HipBridge::makeResult<Multidim::Array<TCV, 4>>(src[0], src[1], Dh, Dw);
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
    static_assert(HipBridge::imageTypeOnGpuPath<T_L, MatchingFunctionTraits<matchFunc>::ZeroMean, MatchingFunctionTraits<matchFunc>::Normalized>() &&
                      HipBridge::imageTypeOnGpuPath<T_R, MatchingFunctionTraits<matchFunc>::ZeroMean, MatchingFunctionTraits<matchFunc>::Normalized>(),
                  "libstevi_hip: images are float, or uint8 with CENSUS / HAMMING / CC / SSD / SAD");
    auto l_shape = img_l.shape();
    auto r_shape = img_r.shape();
    if (l_shape[0] != r_shape[0] || disp_width <= 0) return Multidim::Array<TCV, 3>(0, 0, 0); // :751-753
    if (nImDim == 3 && l_shape[nImDim - 1] != r_shape[nImDim - 1]) return Multidim::Array<TCV, 3>(0, 0, 0); // :755-759
    auto const &src = (dDir == dispDirection::RightToLeft) ? r_shape : l_shape;
    auto cv = HipBridge::makeResult<Multidim::Array<TCV, 3>>(src[0], src[1], disp_width);
    svh_array l = HipBridge::describe(img_l), r = HipBridge::describe(img_r), out = HipBridge::describe(cv);
    if (!HipBridge::check(svh_unfold_cost_volume(HipBridge::context(), static_cast<int>(matchFunc), static_cast<int>(dDir), &l, &r, h_radius, v_radius, 0,
                                                 disp_width, &out)))
        return Multidim::Array<TCV, 3>(0, 0, 0);
    return cv;
}

// unfoldBasedCostVolume with the volume left in GPU memory: the head of a call chain on HipBridge::DeviceArray (the images are
// host arrays and cross PCIe once; sgmCostVolume / extractSelectedIndex / ... then take the DeviceArray overloads)
template <matchingFunctions matchFunc, class T_L, class T_R, int nImDim = 2, dispDirection dDir = dispDirection::RightToLeft, typename TCV = float>
DeviceArray<TCV, 3> unfoldBasedCostVolumeOnDevice(Multidim::Array<T_L, nImDim> const &img_l, Multidim::Array<T_R, nImDim> const &img_r, uint8_t h_radius,
                                                  uint8_t v_radius, disp_t disp_width) {
    static_assert(HipBridge::onGpuPath<matchFunc>(), "libstevi_hip: this matching function has no GPU path");
    static_assert(std::is_same_v<TCV, float>, "libstevi_hip: cost volumes are float");
    static_assert(HipBridge::imageTypeOnGpuPath<T_L, MatchingFunctionTraits<matchFunc>::ZeroMean, MatchingFunctionTraits<matchFunc>::Normalized>() &&
                      HipBridge::imageTypeOnGpuPath<T_R, MatchingFunctionTraits<matchFunc>::ZeroMean, MatchingFunctionTraits<matchFunc>::Normalized>(),
                  "libstevi_hip: images are float, or uint8 with CENSUS / HAMMING / CC / SSD / SAD");
    auto l_shape = img_l.shape();
    auto r_shape = img_r.shape();
    if (l_shape[0] != r_shape[0] || disp_width <= 0) return DeviceArray<TCV, 3>(); // :751-753
    if (nImDim == 3 && l_shape[nImDim - 1] != r_shape[nImDim - 1]) return DeviceArray<TCV, 3>(); // :755-759
    auto const &src = (dDir == dispDirection::RightToLeft) ? r_shape : l_shape;
    DeviceArray<TCV, 3> cv({src[0], src[1], static_cast<int>(disp_width)});
    if (cv.empty()) return cv;
    svh_array l = HipBridge::describe(img_l), r = HipBridge::describe(img_r), out = HipBridge::describe(cv);
    if constexpr (MatchingFunctionTraits<matchFunc>::extractionStrategy == dispExtractionStartegy::Cost) {
        // a Cost-strategy function: also keep the per-pixel regional minima a later sgmCostVolume<.., Cost> would otherwise read the volume
        // for.  The library says which kind it could make: 1 integer costs (census / Hamming), 2 float costs inside the regime, 0 none.
        DeviceArray<float, 3> minima({src[0], src[1], 2});
        svh_array mn = HipBridge::describe(minima);
        int written = 0;
        if (!HipBridge::check(svh_unfold_cost_volume_minima(HipBridge::context(), static_cast<int>(matchFunc), static_cast<int>(dDir), &l, &r, h_radius, v_radius,
                                                            0, disp_width, &out, &mn, &written)))
            return DeviceArray<TCV, 3>();
        if (written) {
            auto st = std::make_shared<typename DeviceArray<TCV, 3>::Statement>();
            st->minima = minima.share();
            st->minima_kind = written;
            const int channels = nImDim == 3 ? l_shape[nImDim - 1] : 1;
            st->max_abs = written == 1 ? static_cast<float>(32 * (((2 * h_radius + 1) * (2 * v_radius + 1) * channels - 1) / 32)) : 1e30f;
            cv.attach(st);
        }
        return cv;
    } else {
        // a Score-strategy function (no regional minima to keep: the Score branch of sgmCostVolume has no use for them): keep the index a
        // later extractSelectedIndex<Score> on the untouched volume would scan it for -- the reference benchmark's own sequence
        // (benchmarkCrossCorrelationAlgorithms.cpp:92-96).  The library says whether the kernel that ran could pick it.
        DeviceArray<int32_t, 2> winner({src[0], src[1]});
        svh_array wi = HipBridge::describe(winner);
        int written = 0;
        if (!HipBridge::check(svh_unfold_cost_volume_winner(HipBridge::context(), static_cast<int>(matchFunc), static_cast<int>(dDir), &l, &r, h_radius, v_radius,
                                                            0, disp_width, &out, &wi, &written)))
            return DeviceArray<TCV, 3>();
        if (written) {
            auto st = std::make_shared<typename DeviceArray<TCV, 3>::Statement>();
            st->winner = winner.share();
            st->winner_strategy = static_cast<int>(MatchingFunctionTraits<matchFunc>::extractionStrategy);
            cv.attach(st);
        }
        return cv;
    }
}

// unfoldBasedCostVolume(img_l, img_r, compressor, disp_width), cross_correlations.h:767-791
template <matchingFunctions matchFunc, class T_L, class T_R, int nImDim = 2, dispDirection dDir = dispDirection::RightToLeft, typename TCV = float>
Multidim::Array<TCV, 3> unfoldBasedCostVolume(Multidim::Array<T_L, nImDim> const &img_l, Multidim::Array<T_R, nImDim> const &img_r,
                                              UnFoldCompressor const &compressor, disp_t disp_width) {
    auto l_shape = img_l.shape();
    auto r_shape = img_r.shape();
    if (l_shape[0] != r_shape[0]) return Multidim::Array<TCV, 3>(0, 0, 0);
    if (nImDim == 3 && l_shape[nImDim - 1] != r_shape[nImDim - 1]) return Multidim::Array<TCV, 3>(0, 0, 0);
    Multidim::Array<float, 3> left_feature_volume = unfold(compressor, img_l);
    Multidim::Array<float, 3> right_feature_volume = unfold(compressor, img_r);
    return featureVolume2CostVolume<matchFunc, float, float, disp_t, dDir, TCV>(left_feature_volume, right_feature_volume, disp_width);
}

// unfoldBased2dDisparityCostVolume(img_l, img_r, compressor, searchOffset<2>), cross_correlations.h:824-851
template <matchingFunctions matchFunc, class T_L, class T_R, int nImDim = 2, dispDirection dDir = dispDirection::RightToLeft, typename TCV = float>
Multidim::Array<TCV, 4> unfoldBased2dDisparityCostVolume(Multidim::Array<T_L, nImDim> const &img_l, Multidim::Array<T_R, nImDim> const &img_r,
                                                         UnFoldCompressor const &compressor, searchOffset<2> const &searchWindows) {
    auto l_shape = img_l.shape();
    auto r_shape = img_r.shape();
    if (l_shape[0] != r_shape[0] || l_shape[1] != r_shape[1]) return Multidim::Array<TCV, 4>();
    if (nImDim == 3 && l_shape[nImDim - 1] != r_shape[nImDim - 1]) return Multidim::Array<TCV, 4>();
    Multidim::Array<float, 3> left_feature_volume = unfold(compressor, img_l);
    Multidim::Array<float, 3> right_feature_volume = unfold(compressor, img_r);
    return featureVolume2CostVolume<matchFunc, float, float, searchOffset<2>, dDir, TCV>(left_feature_volume, right_feature_volume, searchWindows);
}

// unfoldBased2dDisparityCostVolume<matchFunc, T_L, T_R, nImDim, dDir, TCV>(img_l, img_r, h_radius, v_radius, searchOffset<2>),
// cross_correlations.h:794-822
template <matchingFunctions matchFunc, class T_L, class T_R, int nImDim = 2, dispDirection dDir = dispDirection::RightToLeft, typename TCV = float>
Multidim::Array<TCV, 4> unfoldBased2dDisparityCostVolume(Multidim::Array<T_L, nImDim> const &img_l, Multidim::Array<T_R, nImDim> const &img_r,
                                                         uint8_t h_radius, uint8_t v_radius, searchOffset<2> const &searchWindows) {
    static_assert(HipBridge::onGpuPath<matchFunc>(), "libstevi_hip: this matching function has no GPU path");
    static_assert(std::is_same_v<TCV, float>, "libstevi_hip: cost volumes are float");
    static_assert(HipBridge::imageTypeOnGpuPath<T_L, MatchingFunctionTraits<matchFunc>::ZeroMean, MatchingFunctionTraits<matchFunc>::Normalized>() &&
                      HipBridge::imageTypeOnGpuPath<T_R, MatchingFunctionTraits<matchFunc>::ZeroMean, MatchingFunctionTraits<matchFunc>::Normalized>(),
                  "libstevi_hip: images are float, or uint8 with CENSUS / HAMMING / CC / SSD / SAD");
    auto l_shape = img_l.shape();
    auto r_shape = img_r.shape();
    if (l_shape[0] != r_shape[0] || l_shape[1] != r_shape[1]) return Multidim::Array<TCV, 4>(); // :804-810
    if (nImDim == 3 && l_shape[nImDim - 1] != r_shape[nImDim - 1]) return Multidim::Array<TCV, 4>();
    const int Dh = searchWindows.dimRange(0), Dw = searchWindows.dimRange(1);
    if (Dh <= 0 || Dw <= 0) return Multidim::Array<TCV, 4>(); // :338-340
    auto cv = HipBridge::makeResult<Multidim::Array<TCV, 4>>(l_shape[0], l_shape[1], Dh, Dw);
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

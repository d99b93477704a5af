// Drop-in for LibStevi's correlation/correlation_base.h (hot-path subset): same names and signatures, bodies
// marshal to libstevi_hip.so.  Reference lines are cited per function.
#ifndef STEREOVISION_CORRELATION_BASE_H
#define STEREOVISION_CORRELATION_BASE_H

#include <array>
#include <cstdint>
#include <type_traits>

#include "./stevi_hip_bridge.h"

namespace StereoVision {
namespace Correlation {

enum class dispExtractionStartegy { Cost = 0, Score = 1 };                       // correlation_base.h:31-34
enum class dispDirection { LeftToRight = 0, RightToLeft = 1 };                   // :36-39
enum class truncatedCostVolumeDirection { Same = 0, Reversed = 1, Both = 2 };    // :41-45
typedef int32_t disp_t;                                                          // :47

// searchOffset<nDim>, correlation_base.h:288-409 (the accessors the 1-D path uses)
template <int nDim> class searchOffset {
  public:
    searchOffset() : _isValid(false) {
        _upperOffsets.fill(0);
        _lowerOffsets.fill(0);
    }
    template <typename... Ds> searchOffset(disp_t lowerOffset0, disp_t upperOffset0, Ds... nextOffsets) : _isValid(true) {
        static_assert(sizeof...(nextOffsets) == 2 * (nDim - 1), "The number of offsets provided to the constructor should be twice the number of dimensions !");
        std::array<disp_t, 2 * (nDim - 1) + 1> n{static_cast<disp_t>(nextOffsets)..., 0};
        _lowerOffsets[0] = lowerOffset0;
        _upperOffsets[0] = upperOffset0;
        for (int i = 1; i < nDim; i++) {
            _lowerOffsets[i] = n[2 * (i - 1)];
            _upperOffsets[i] = n[2 * (i - 1) + 1];
        }
    }
    bool isValid() const { return _isValid; }
    template <int dim> int const &upperOffset() const { return _upperOffsets[dim]; }
    template <int dim> int const &lowerOffset() const { return _lowerOffsets[dim]; }
    int const &upperOffset(int dim) const { return _upperOffsets[dim]; }
    int const &lowerOffset(int dim) const { return _lowerOffsets[dim]; }
    int dimRange(int dim) const { return _upperOffsets[dim] - _lowerOffsets[dim] + 1; }
    int idx2disp(int dim, int idx) const { return _lowerOffsets[dim] + idx; }
    int disp2idx(int dim, int disp) const { return disp - _lowerOffsets[dim]; }

  private:
    std::array<disp_t, nDim> _upperOffsets, _lowerOffsets;
    bool _isValid;
};

template <typename SearchRangeType> struct searchRangeTypeInfos {};
template <> struct searchRangeTypeInfos<disp_t> { static const int CostVolumeDims = 3; };
template <int nDim> struct searchRangeTypeInfos<searchOffset<nDim>> { static const int CostVolumeDims = 2 + nDim; };

// condImgRef<T_L, T_R, dDir, nImDims>, correlation_base.h:824-878: which of the (left, right) pair is the SOURCE (the image whose pixels
// get a disparity) and which the TARGET (the image searched), with their element types T_S / T_T: the right image is the source when
// disparities run RightToLeft, the left one otherwise.  Holds references; a helper of the reference's own bodies that callers of
// aggregateCost-like templates (hierarchical.h:82, cross_correlations.h:199) name.
template <class T_L, class T_R, dispDirection dDir, int nImDims = 2> class condImgRef {
    static constexpr bool rightIsSource = dDir == dispDirection::RightToLeft;

  public:
    typedef std::conditional_t<rightIsSource, T_R, T_L> T_S;
    typedef std::conditional_t<rightIsSource, T_L, T_R> T_T;

    explicit condImgRef(Multidim::Array<T_L, nImDims> const &im_l, Multidim::Array<T_R, nImDims> const &im_r) : img_l(im_l), img_r(im_r) {}

    Multidim::Array<T_S, nImDims> const &source() const {
        if constexpr (rightIsSource) return img_r;
        else return img_l;
    }
    Multidim::Array<T_T, nImDims> const &target() const {
        if constexpr (rightIsSource) return img_l;
        else return img_r;
    }

  private:
    Multidim::Array<T_L, nImDims> const &img_l;
    Multidim::Array<T_R, nImDims> const &img_r;
};

// extractSelectedIndex<strategy>, correlation_base.h:427-464
template <dispExtractionStartegy strategy, class T_CV> Multidim::Array<disp_t, 2> extractSelectedIndex(Multidim::Array<T_CV, 3> const &costVolume) {
    auto s = costVolume.shape();
    auto disp = HipBridge::makeResult<Multidim::Array<disp_t, 2>>(s[0], s[1]);
    if (disp.empty()) return disp;
    svh_array cv = HipBridge::describe(costVolume), out = HipBridge::describe(disp);
    HipBridge::check(svh_extract_selected_index(HipBridge::context(), static_cast<int>(strategy), &cv, &out));
    return disp;
}

// selectedIndexToDisp<DT, dDir>, correlation_base.h:511-532
template <typename DT, dispDirection dDir = dispDirection::RightToLeft>
Multidim::Array<DT, 2> selectedIndexToDisp(Multidim::Array<DT, 2> const &selectedIndex, disp_t disp_offset = 0) {
    auto s = selectedIndex.shape();
    auto disp = HipBridge::makeResult<Multidim::Array<DT, 2>>(s[0], s[1]);
    if (disp.empty()) return disp;
    svh_array in = HipBridge::describe(selectedIndex), out = HipBridge::describe(disp);
    HipBridge::check(svh_selected_index_to_disp(HipBridge::context(), static_cast<int>(dDir), &in, disp_offset, &out));
    return disp;
}

// selectedCost, correlation_base.h:557-577
template <class T_CV> Multidim::Array<T_CV, 2> selectedCost(Multidim::Array<T_CV, 3> const &costVolume, Multidim::Array<disp_t, 2> const &selectedIndex) {
    auto s = costVolume.shape();
    auto tcv = HipBridge::makeResult<Multidim::Array<T_CV, 2>>(s[0], s[1]);
    if (tcv.empty()) return tcv;
    svh_array cv = HipBridge::describe(costVolume), idx = HipBridge::describe(selectedIndex), out = HipBridge::describe(tcv);
    HipBridge::check(svh_selected_cost(HipBridge::context(), &cv, &idx, &out));
    return tcv;
}

// truncatedCostVolume<T_CV, dir, sdir>, correlation_base.h:579-674
template <class T_CV, dispDirection dir = dispDirection::RightToLeft, truncatedCostVolumeDirection sdir = truncatedCostVolumeDirection::Same>
Multidim::Array<T_CV, 3> truncatedCostVolume(Multidim::Array<T_CV, 3> const &costVolume, Multidim::Array<disp_t, 2> const &selectedIndex,
                                             uint8_t h_radius, uint8_t v_radius, uint8_t cost_vol_radius) {
    auto s = costVolume.shape();
    auto tcv = HipBridge::makeResult<Multidim::Array<T_CV, 3>>(s[0], s[1], (sdir == truncatedCostVolumeDirection::Both) ? cost_vol_radius * 4 + 1 : cost_vol_radius * 2 + 1);
    if (tcv.empty()) return tcv;
    svh_array cv = HipBridge::describe(costVolume), idx = HipBridge::describe(selectedIndex), out = HipBridge::describe(tcv);
    HipBridge::check(svh_truncated_cost_volume(HipBridge::context(), static_cast<int>(sdir), static_cast<int>(dir), &cv, &idx, h_radius, v_radius,
                                               cost_vol_radius, &out));
    return tcv;
}

// extractSelected2dIndex<strategy>, correlation_base.h:466-509
template <dispExtractionStartegy strategy, class T_CV> Multidim::Array<disp_t, 3> extractSelected2dIndex(Multidim::Array<T_CV, 4> const &costVolume) {
    auto s = costVolume.shape();
    auto disp = HipBridge::makeResult<Multidim::Array<disp_t, 3>>(s[0], s[1], 2);
    if (costVolume.empty()) return disp;
    svh_array cv = HipBridge::describe(costVolume), out = HipBridge::describe(disp);
    HipBridge::check(svh_extract_selected_2d_index(HipBridge::context(), static_cast<int>(strategy), &cv, &out));
    return disp;
}

// selected2dIndexToDisp(selectedIndex, searchOffset<2>), correlation_base.h:534-555
template <typename DT> Multidim::Array<DT, 3> selected2dIndexToDisp(Multidim::Array<DT, 3> const &selectedIndex, searchOffset<2> const &offset) {
    auto s = selectedIndex.shape();
    auto disp = HipBridge::makeResult<Multidim::Array<DT, 3>>(s[0], s[1], 2);
    if (selectedIndex.empty()) return disp;
    svh_array in = HipBridge::describe(selectedIndex), out = HipBridge::describe(disp);
    HipBridge::check(svh_selected_2d_index_to_disp(HipBridge::context(), &in, offset.lowerOffset(0), offset.lowerOffset(1), &out));
    return disp;
}

// truncatedBidirectionaCostVolume<T_CV, dir>(costVolume, selectedIndex, radius0, radius1), correlation_base.h:677-725
// (the reference's default radii are uint8_t(-1); pass them explicitly)
template <class T_CV, dispDirection dir = dispDirection::RightToLeft>
Multidim::Array<T_CV, 4> truncatedBidirectionaCostVolume(Multidim::Array<T_CV, 4> const &costVolume, Multidim::Array<disp_t, 3> const &selectedIndex,
                                                         uint8_t cost_vol_radius0, uint8_t cost_vol_radius1) {
    uint8_t r0 = cost_vol_radius0 < 1 ? 1 : cost_vol_radius0, r1 = cost_vol_radius1 < 1 ? r0 : cost_vol_radius1; // :690-696
    auto s = costVolume.shape();
    auto tcv = HipBridge::makeResult<Multidim::Array<T_CV, 4>>(s[0], s[1], r0 * 2 + 1, r1 * 2 + 1);
    if (costVolume.empty()) return tcv;
    svh_array cv = HipBridge::describe(costVolume), idx = HipBridge::describe(selectedIndex), out = HipBridge::describe(tcv);
    HipBridge::check(svh_truncated_bidirectional_cost_volume(HipBridge::context(), &cv, &idx, r0, r1, &out));
    return tcv;
}

// ---- the same functions on arrays that stay in GPU memory (HipBridge::DeviceArray, stevi_hip_bridge.h): same names, same template
// parameters, selected by the argument type; results are device arrays too, download() brings one to the host
using HipBridge::DeviceArray;

template <dispExtractionStartegy strategy, class T_CV> DeviceArray<disp_t, 2> extractSelectedIndex(DeviceArray<T_CV, 3> const &costVolume) {
    DeviceArray<disp_t, 2> disp({costVolume.shape()[0], costVolume.shape()[1]});
    if (disp.empty()) return disp;
    if constexpr (std::is_same_v<std::remove_const_t<T_CV>, float> && std::is_same_v<disp_t, int32_t>) {
        // an aggregated volume sgmCostVolume wrote and nobody has touched since carries its winner (stevi_hip_bridge.h: Statement): a copy of
        // that map (the caller may write to what it gets) instead of a scan of the volume
        auto known = costVolume.statement();
        if (known && known->winner && known->winner_strategy == static_cast<int>(strategy)) {
            HipBridge::check(svh_device_copy(HipBridge::context(), disp.data(), known->winner.get(), disp.flatLenght() * sizeof(disp_t)));
            return disp;
        }
    }
    svh_array cv = HipBridge::describe(costVolume), out = HipBridge::describe(disp);
    HipBridge::check(svh_extract_selected_index(HipBridge::context(), static_cast<int>(strategy), &cv, &out));
    return disp;
}

template <typename DT, dispDirection dDir = dispDirection::RightToLeft>
DeviceArray<DT, 2> selectedIndexToDisp(DeviceArray<DT, 2> const &selectedIndex, disp_t disp_offset = 0) {
    DeviceArray<DT, 2> disp(selectedIndex.shape());
    if (disp.empty()) return disp;
    svh_array in = HipBridge::describe(selectedIndex), out = HipBridge::describe(disp);
    HipBridge::check(svh_selected_index_to_disp(HipBridge::context(), static_cast<int>(dDir), &in, disp_offset, &out));
    return disp;
}

template <class T_CV> DeviceArray<T_CV, 2> selectedCost(DeviceArray<T_CV, 3> const &costVolume, DeviceArray<disp_t, 2> const &selectedIndex) {
    DeviceArray<T_CV, 2> tcv({costVolume.shape()[0], costVolume.shape()[1]});
    if (tcv.empty()) return tcv;
    svh_array cv = HipBridge::describe(costVolume), idx = HipBridge::describe(selectedIndex), out = HipBridge::describe(tcv);
    HipBridge::check(svh_selected_cost(HipBridge::context(), &cv, &idx, &out));
    return tcv;
}

template <class T_CV, dispDirection dir = dispDirection::RightToLeft, truncatedCostVolumeDirection sdir = truncatedCostVolumeDirection::Same>
DeviceArray<T_CV, 3> truncatedCostVolume(DeviceArray<T_CV, 3> const &costVolume, DeviceArray<disp_t, 2> const &selectedIndex, uint8_t h_radius,
                                         uint8_t v_radius, uint8_t cost_vol_radius) {
    DeviceArray<T_CV, 3> tcv({costVolume.shape()[0], costVolume.shape()[1],
                              (sdir == truncatedCostVolumeDirection::Both) ? cost_vol_radius * 4 + 1 : cost_vol_radius * 2 + 1});
    if (tcv.empty()) return tcv;
    svh_array cv = HipBridge::describe(costVolume), idx = HipBridge::describe(selectedIndex), out = HipBridge::describe(tcv);
    HipBridge::check(svh_truncated_cost_volume(HipBridge::context(), static_cast<int>(sdir), static_cast<int>(dir), &cv, &idx, h_radius, v_radius,
                                               cost_vol_radius, &out));
    return tcv;
}

// channelsMean<T_I, T_O>(in_data), correlation_base.h:1100-1136
template <class T_I, class T_O = float, Multidim::ArrayDataAccessConstness C>
inline Multidim::Array<T_O, 2> channelsMean(Multidim::Array<T_I, 3, C> const &in_data) {
    static_assert(std::is_same_v<std::remove_const_t<T_I>, float> && std::is_same_v<T_O, float>, "libstevi_hip: float feature volumes only");
    auto mean = HipBridge::makeResult<Multidim::Array<T_O, 2>>(in_data.shape()[0], in_data.shape()[1]);
    if (mean.empty()) return mean;
    svh_array in = HipBridge::describe(in_data), out = HipBridge::describe(mean);
    HipBridge::check(svh_channels_mean(HipBridge::context(), &in, &out));
    return mean;
}

} // namespace Correlation
} // namespace StereoVision

#endif // STEREOVISION_CORRELATION_BASE_H

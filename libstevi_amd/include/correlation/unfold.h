// Drop-in for LibStevi's correlation/unfold.h: plain unfold (all four patch orientations) and the UnFoldCompressor variant with its two generators.
#ifndef STEREOVISION_CORRELATION_UNFOLD_H
#define STEREOVISION_CORRELATION_UNFOLD_H

#include <algorithm>
#include <map>
#include <vector>

#include "../utils/margins.h"
#include "./correlation_base.h"

namespace StereoVision {
namespace Correlation {

// UnFoldCompressor(mask), unfold.h:36-137: the accessors callers use; the GPU rebuilds the same index list from the mask
class UnFoldCompressor {
  public:
    struct pixelIndex {
        int verticalShift;
        int horizontalShift;
        int featureIndex;
        float weight;
    };

    template <Multidim::ArrayDataAccessConstness viewConstness> explicit UnFoldCompressor(Multidim::Array<int, 2, viewConstness> const &mask) {
        _mask_h = mask.shape()[0];
        _mask_w = mask.shape()[1];
        _mask.resize(static_cast<size_t>(_mask_h) * _mask_w);
        std::map<int, int> pixels;
        int minH = 0, maxH = 0, minW = 0, maxW = 0; // the box always contains the centre
        for (int i = 0; i < _mask_h; i++)
            for (int j = 0; j < _mask_w; j++) {
                const int label = mask.valueUnchecked(i, j);
                _mask[static_cast<size_t>(i) * _mask_w + j] = label;
                if (label <= 0) continue;
                minH = std::min(minH, i - _mask_h / 2);
                maxH = std::max(maxH, i - _mask_h / 2);
                minW = std::min(minW, j - _mask_w / 2);
                maxW = std::max(maxW, j - _mask_w / 2);
                pixels[label]++;
            }
        _height = maxH - minH + 1;
        _width = maxW - minW + 1;
        _margins = PaddingMargins(-minW, -minH, maxW, maxH);
        _nFeatures = static_cast<int>(pixels.size());
        int f = 0;
        for (auto const &kv : pixels) { // increasing label order
            for (int i = 0; i < _mask_h; i++)
                for (int j = 0; j < _mask_w; j++)
                    if (_mask[static_cast<size_t>(i) * _mask_w + j] == kv.first)
                        _indices.push_back({i - _mask_h / 2, j - _mask_w / 2, f, static_cast<float>(1. / kv.second)});
            f++;
        }
    }

    inline int nFeatures() const { return _nFeatures; }
    inline int width() const { return _width; }
    inline int height() const { return _height; }
    inline PaddingMargins margins() const { return _margins; }
    inline std::vector<pixelIndex> indices() const { return _indices; }
    // what the C ABI takes
    inline const int32_t *maskData() const { return _mask.data(); }
    inline int maskHeight() const { return _mask_h; }
    inline int maskWidth() const { return _mask_w; }

  protected:
    int _nFeatures, _width, _height, _mask_h, _mask_w;
    PaddingMargins _margins;
    std::vector<pixelIndex> _indices;
    std::vector<int32_t> _mask;
};

enum UnfoldPatchOrientation { Rotate0, Rotate90, Rotate180, Rotate270 }; // unfold.h:139-144

// unfold<T_I, T_O>(h_radius, v_radius, in_data, padding, orientation), unfold.h:247-344
template <class T_I, class T_O = float, int nImDim, Multidim::ArrayDataAccessConstness constness>
Multidim::Array<T_O, 3> unfold(uint8_t h_radius, uint8_t v_radius, Multidim::Array<T_I, nImDim, constness> const &in_data,
                               PaddingMargins const &padding = PaddingMargins(), UnfoldPatchOrientation orientation = Rotate0) {
    static_assert(nImDim == 2 || nImDim == 3, "unfold takes grey (2-D) or multi-channel (3-D) images");
    static_assert(std::is_same_v<std::remove_const_t<T_I>, float> || std::is_same_v<std::remove_const_t<T_I>, uint8_t>,
                  "libstevi_hip: unfold takes float or uint8 images");
    static_assert(std::is_arithmetic_v<T_O>, "libstevi_hip: unfold writes arithmetic element types");
    const int32_t pad[4] = {padding.left(), padding.top(), padding.right(), padding.bottom()};
    const int32_t *pp = padding.isAuto() ? nullptr : pad;
    svh_array in = HipBridge::describe(in_data);
    int64_t shp[3];
    if (svh_unfold_shape(&in, h_radius, v_radius, pp, shp) != SVH_OK || shp[0] <= 0 || shp[1] <= 0) return Multidim::Array<T_O, 3>();
    auto out = HipBridge::makeResult<Multidim::Array<float, 3>>(static_cast<int>(shp[0]), static_cast<int>(shp[1]), static_cast<int>(shp[2])); // the device writes float samples
    {
        svh_array o = HipBridge::describe(out);
        if (!HipBridge::check(svh_unfold_oriented(HipBridge::context(), &in, h_radius, v_radius, pp, static_cast<int>(orientation), &o)))
            return Multidim::Array<T_O, 3>();
    }
    if constexpr (std::is_same_v<T_O, float>) return out;
    else return out.template cast<T_O>(); // static_cast per element, as unfold.h:283 stores them
}

// unfold(compressor, in_data, padding), unfold.h:346-471 (2-D and 3-D images)
template <class T_I, class T_O = float, int nImDim, Multidim::ArrayDataAccessConstness constness>
Multidim::Array<T_O, 3> unfold(UnFoldCompressor const &compressor, Multidim::Array<T_I, nImDim, constness> const &in_data,
                               PaddingMargins const &padding = PaddingMargins()) {
    static_assert(nImDim == 2 || nImDim == 3, "unfold takes grey (2-D) or multi-channel (3-D) images");
    static_assert((std::is_same_v<std::remove_const_t<T_I>, float> || std::is_same_v<std::remove_const_t<T_I>, uint8_t>) && std::is_same_v<T_O, float>,
                  "libstevi_hip: compressed unfold takes float or uint8 images and writes float features");
    const int32_t pad[4] = {padding.left(), padding.top(), padding.right(), padding.bottom()};
    const int32_t *pp = padding.isAuto() ? nullptr : pad;
    svh_array in = HipBridge::describe(in_data);
    int64_t shp[3];
    if (svh_unfold_compressed_shape(&in, compressor.maskData(), compressor.maskHeight(), compressor.maskWidth(), pp, shp) != SVH_OK || shp[0] <= 0 ||
        shp[1] <= 0 || shp[2] <= 0)
        return Multidim::Array<T_O, 3>();
    auto out = HipBridge::makeResult<Multidim::Array<T_O, 3>>(static_cast<int>(shp[0]), static_cast<int>(shp[1]), static_cast<int>(shp[2]));
    svh_array o = HipBridge::describe(out);
    if (!HipBridge::check(svh_unfold_compressed(HipBridge::context(), &in, compressor.maskData(), compressor.maskHeight(), compressor.maskWidth(), pp, &o)))
        return Multidim::Array<T_O, 3>();
    return out;
}

namespace CompressorGenerators { // unfold.h:475-693: the two 17-superpixel window masks

namespace detail {
template <int N> inline Multidim::Array<int, 2> maskFromRows(const int (&rows)[N][N]) {
    auto out = HipBridge::makeResult<Multidim::Array<int, 2>>(N, N);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) out.atUnchecked(i, j) = rows[i][j];
    return out;
}
} // namespace detail

inline Multidim::Array<int, 2> GrPix17R3Filter() {
    static const int rows[7][7] = {{14, 14, 10, 10, 10, 16, 16}, {14, 14, 6, 4, 7, 16, 16}, {11, 6, 6, 4, 7, 7, 13}, {11, 2, 2, 1, 3, 3, 13},
                                   {11, 8, 8, 5, 9, 9, 13},      {15, 15, 8, 5, 9, 17, 17}, {15, 15, 12, 12, 12, 17, 17}};
    return detail::maskFromRows(rows);
}

inline Multidim::Array<int, 2> GrPix17R4Filter() {
    static const int rows[9][9] = {{14, 14, 14, 10, 10, 10, 16, 16, 16}, {14, 14, 14, 10, 10, 10, 16, 16, 16}, {14, 14, 6, 6, 4, 7, 7, 16, 16},
                                   {11, 11, 6, 6, 4, 7, 7, 13, 13},      {11, 11, 2, 2, 1, 3, 3, 13, 13},      {11, 11, 8, 8, 5, 9, 9, 13, 13},
                                   {15, 15, 8, 8, 5, 9, 9, 17, 17},      {15, 15, 15, 12, 12, 12, 17, 17, 17}, {15, 15, 15, 12, 12, 12, 17, 17, 17}};
    return detail::maskFromRows(rows);
}

} // namespace CompressorGenerators

} // namespace Correlation
} // namespace StereoVision

#endif // STEREOVISION_CORRELATION_UNFOLD_H

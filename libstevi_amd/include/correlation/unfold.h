// Drop-in for LibStevi's correlation/unfold.h (plain unfold, Rotate0; the UnFoldCompressor variant is out of scope).
#ifndef STEREOVISION_CORRELATION_UNFOLD_H
#define STEREOVISION_CORRELATION_UNFOLD_H

#include "../utils/margins.h"
#include "./correlation_base.h"

namespace StereoVision {
namespace Correlation {

enum UnfoldPatchOrientation { Rotate0, Rotate90, Rotate180, Rotate270 }; // unfold.h:139-144

// unfold<T_I, T_O>(h_radius, v_radius, in_data, padding, orientation), unfold.h:247-344
template <class T_I, class T_O = float, int nImDim, Multidim::ArrayDataAccessConstness constness>
Multidim::Array<T_O, 3> unfold(uint8_t h_radius, uint8_t v_radius, Multidim::Array<T_I, nImDim, constness> const &in_data,
                               PaddingMargins const &padding = PaddingMargins(), UnfoldPatchOrientation orientation = Rotate0) {
    static_assert(nImDim == 2 || nImDim == 3, "unfold takes grey (2-D) or multi-channel (3-D) images");
    static_assert(std::is_same_v<T_I, float> && std::is_same_v<T_O, float>, "libstevi_hip: unfold is implemented for float images");
    if (orientation != Rotate0) throw std::runtime_error("libstevi_hip: only Rotate0 unfold is implemented");
    const int32_t pad[4] = {padding.left(), padding.top(), padding.right(), padding.bottom()};
    const int32_t *pp = padding.isAuto() ? nullptr : pad;
    svh_array in = HipBridge::describe(in_data);
    int64_t shp[3];
    if (svh_unfold_shape(&in, h_radius, v_radius, pp, shp) != SVH_OK || shp[0] <= 0 || shp[1] <= 0) return Multidim::Array<T_O, 3>();
    Multidim::Array<T_O, 3> out(static_cast<int>(shp[0]), static_cast<int>(shp[1]), static_cast<int>(shp[2]));
    std::lock_guard<std::mutex> g(HipBridge::lock());
    svh_array o = HipBridge::describe(out);
    if (!HipBridge::check(svh_unfold(HipBridge::context(), &in, h_radius, v_radius, pp, &o))) return Multidim::Array<T_O, 3>();
    return out;
}

} // namespace Correlation
} // namespace StereoVision

#endif // STEREOVISION_CORRELATION_UNFOLD_H

// Drop-in for LibStevi's correlation/census.h.
#ifndef STEREOVISION_CORRELATION_CENSUS_H
#define STEREOVISION_CORRELATION_CENSUS_H

#include "./correlation_base.h"
#include "./matching_costs.h"
#include "./unfold.h"

namespace StereoVision {
namespace Correlation {

// censusFeatures(baseFeatures), census.h:69-115
template <typename T_I, Multidim::ArrayDataAccessConstness C> Multidim::Array<census_data_t, 3> censusFeatures(Multidim::Array<T_I, 3, C> const &baseFeatures) {
    auto s = baseFeatures.shape();
    if (s[2] <= 1) return Multidim::Array<census_data_t, 3>(); // census.h:76-78
    auto census = HipBridge::makeResult<Multidim::Array<census_data_t, 3>>(s[0], s[1], (s[2] - 1) / 32 + 1);
    svh_array in = HipBridge::describe(baseFeatures), out = HipBridge::describe(census);
    if (!HipBridge::check(svh_census_features(HipBridge::context(), &in, &out))) return Multidim::Array<census_data_t, 3>();
    return census;
}

// censusTransform2D(input, h_radius, v_radius, padding), census.h:117-131
template <typename T_I, int nDim>
Multidim::Array<census_data_t, 3> censusTransform2D(Multidim::Array<T_I, nDim> const &input, int8_t h_radius, int8_t v_radius,
                                                    PaddingMargins const &padding = PaddingMargins()) {
    static_assert(nDim == 2 or nDim == 3, "Can process only 2D (grascale images) or 3D (colored images) arrays.");
    const int32_t pad[4] = {padding.left(), padding.top(), padding.right(), padding.bottom()};
    const int32_t *pp = padding.isAuto() ? nullptr : pad;
    svh_array in = HipBridge::describe(input);
    int64_t shp[3];
    if (svh_unfold_shape(&in, h_radius, v_radius, pp, shp) != SVH_OK || shp[0] <= 0 || shp[1] <= 0 || shp[2] <= 1)
        return Multidim::Array<census_data_t, 3>();
    auto census = HipBridge::makeResult<Multidim::Array<census_data_t, 3>>(static_cast<int>(shp[0]), static_cast<int>(shp[1]), static_cast<int>((shp[2] - 1) / 32 + 1));
    svh_array out = HipBridge::describe(census);
    if (!HipBridge::check(svh_census_transform(HipBridge::context(), &in, h_radius, v_radius, pp, &out))) return Multidim::Array<census_data_t, 3>();
    return census;
}

} // namespace Correlation
} // namespace StereoVision

#endif // STEREOVISION_CORRELATION_CENSUS_H

// Drop-in for LibStevi's correlation/patchmatch.h, the entry point examples/stereo-match calls: cachelessPatchMatch on on-demand
// feature volumes.  One C call runs initialisation, propagation sweeps and random search on the GPU.
#ifndef STEREOVISION_PATCHMATCH_H
#define STEREOVISION_PATCHMATCH_H

#include <functional>
#include <optional>
#include <random>

#include "./on_demand_cost_volume.h"

namespace StereoVision {
namespace Correlation {

namespace HipBridge {
// The reference seeds its engines from std::random_device on every call (patchmatch.h:76-77, :243-244).  So does this shim,
// unless a seed is fixed here for reproducible runs.
inline std::optional<uint64_t> &patchMatchSeed() {
    static std::optional<uint64_t> seed;
    return seed;
}
} // namespace HipBridge

// cachelessPatchMatch<matchFunc, searchSpaceDim>(f_s, f_t, searchOffset, nIter, nRandomSearch, initializer, randcache), patchmatch.h:560-621
template <matchingFunctions matchFunc, int searchSpaceDim, class T_FV_S, class T_FV_T, class InitializerT = std::nullopt_t, class RandCacheT = std::nullopt_t>
Multidim::Array<disp_t, 3> cachelessPatchMatch(T_FV_S const &f_s_p, T_FV_T const &f_t_p, searchOffset<searchSpaceDim> searchOffset, int nIter = 5,
                                               int nRandomSearch = 4, InitializerT = std::nullopt, RandCacheT = std::nullopt) {
    static_assert(searchSpaceDim == 1 or searchSpaceDim == 2, "patchMatch function can only be used to search in 1 or two dimension !");
    static_assert(HipBridge::OnDemandSupport<matchFunc, T_FV_S>::value && HipBridge::OnDemandSupport<matchFunc, T_FV_T>::value,
                  "libstevi_hip: PatchMatch runs on float images decorated with ZNFeaturesVolumeDecorator<ZeroMean, Normalized> of the matching function");
    static_assert(std::is_same_v<InitializerT, std::nullopt_t> && std::is_same_v<RandCacheT, std::nullopt_t>,
                  "libstevi_hip: custom initializers and random-number caches are host callbacks; pass std::nullopt");
    Multidim::Array<disp_t, 3> disp;
    if (f_s_p.shape()[2] != f_t_p.shape()[2]) return disp;                          // :583-585
    if (searchSpaceDim == 1 && f_s_p.shape()[0] != f_t_p.shape()[0]) return disp;   // :587-591
    svh_on_demand_params p{};
    p.match_func = static_cast<int>(matchFunc);
    p.search_dims = searchSpaceDim;
    HipBridge::windowRadii(f_s_p, f_t_p, p.v_radius, p.h_radius);
    if (searchSpaceDim == 2) {
        p.lower0 = searchOffset.lowerOffset(0);
        p.upper0 = searchOffset.upperOffset(0);
        p.lower1 = searchOffset.lowerOffset(1);
        p.upper1 = searchOffset.upperOffset(1);
    } else {
        p.lower1 = searchOffset.lowerOffset(0);
        p.upper1 = searchOffset.upperOffset(0);
    }
    disp = Multidim::Array<disp_t, 3>(f_s_p.shape()[0], f_s_p.shape()[1], searchSpaceDim);
    if (disp.empty()) return disp;
    uint64_t seed;
    if (HipBridge::patchMatchSeed().has_value()) seed = *HipBridge::patchMatchSeed();
    else {
        std::random_device rd;
        seed = (static_cast<uint64_t>(rd()) << 32) | rd();
    }
    svh_array s = HipBridge::describe(f_s_p.array()), t = HipBridge::describe(f_t_p.array()), o = HipBridge::describe(disp);
    if (!HipBridge::check(svh_cacheless_patch_match(HipBridge::context(), &p, &s, &t, nIter, nRandomSearch, seed, &o, nullptr))) return Multidim::Array<disp_t, 3>();
    return disp;
}

} // namespace Correlation
} // namespace StereoVision

#endif // STEREOVISION_PATCHMATCH_H

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

namespace HipBridge {
// The two trailing arguments of the reference's entry points.  `initializer` (std::optional<std::function<Array<disp_t, 3>(fs, ft)>>,
// patchmatch.h:44-48, or std::nullopt): when it holds a callback, the callback runs on the host, as in the reference (:538-545, :598-605),
// and its map goes to the library as the initial solution.  `randcache` (std::optional<Random::NumbersCache<int>> or std::nullopt) is
// accepted and NOT used: the reference draws from it -- or from engines seeded by std::random_device -- in an order that depends on its
// OpenMP schedule, so its stream is not reproducible even by itself; here every draw is a function of (seed, iteration, pixel, draw).
template <class T> struct is_nullopt : std::is_same<std::decay_t<T>, std::nullopt_t> {};
template <class InitializerT, class FS, class FT> inline Multidim::Array<disp_t, 3> initialSolution(InitializerT const &init, FS const &fs, FT const &ft) {
    if constexpr (is_nullopt<InitializerT>::value) {
        (void)init; (void)fs; (void)ft;
        return Multidim::Array<disp_t, 3>();
    } else {
        if (init.has_value()) return init.value()(fs, ft);
        return Multidim::Array<disp_t, 3>();
    }
}
inline uint64_t nextPatchMatchSeed() {
    if (patchMatchSeed().has_value()) return *patchMatchSeed();
    std::random_device rd;
    return (static_cast<uint64_t>(rd()) << 32) | rd();
}
template <int searchSpaceDim> inline void offsetsToParams(searchOffset<searchSpaceDim> const &so, svh_on_demand_params &p) {
    if constexpr (searchSpaceDim == 2) {
        p.lower0 = so.lowerOffset(0);
        p.upper0 = so.upperOffset(0);
        p.lower1 = so.lowerOffset(1);
        p.upper1 = so.upperOffset(1);
    } else {
        p.lower1 = so.lowerOffset(0);
        p.upper1 = so.upperOffset(0);
    }
}
} // namespace HipBridge

// cachelessPatchMatch<matchFunc, searchSpaceDim>(f_s, f_t, searchOffset, nIter, nRandomSearch, initializer, randcache), patchmatch.h:560-621
template <matchingFunctions matchFunc, int searchSpaceDim, class T_FV_S, class T_FV_T, class InitializerT = std::nullopt_t, class RandCacheT = std::nullopt_t>
Multidim::Array<disp_t, 3> cachelessPatchMatch(T_FV_S const &f_s_p, T_FV_T const &f_t_p, searchOffset<searchSpaceDim> searchOffset, int nIter = 5,
                                               int nRandomSearch = 4, InitializerT initializer = std::nullopt, RandCacheT = std::nullopt) {
    static_assert(searchSpaceDim == 1 or searchSpaceDim == 2, "patchMatch function can only be used to search in 1 or two dimension !");
    static_assert(HipBridge::OnDemandSupport<matchFunc, T_FV_S>::value && HipBridge::OnDemandSupport<matchFunc, T_FV_T>::value,
                  "libstevi_hip: PatchMatch runs on float images decorated with ZNFeaturesVolumeDecorator<ZeroMean, Normalized> of the matching function");
    Multidim::Array<disp_t, 3> disp;
    if (f_s_p.shape()[2] != f_t_p.shape()[2]) return disp;                          // :583-585
    if (searchSpaceDim == 1 && f_s_p.shape()[0] != f_t_p.shape()[0]) return disp;   // :587-591
    svh_on_demand_params p{};
    p.match_func = static_cast<int>(matchFunc);
    p.search_dims = searchSpaceDim;
    HipBridge::windowRadii(f_s_p, f_t_p, p.v_radius, p.h_radius);
    HipBridge::offsetsToParams(searchOffset, p);
    Multidim::Array<disp_t, 3> initial = HipBridge::initialSolution(initializer, f_s_p, f_t_p);
    disp = Multidim::Array<disp_t, 3>(f_s_p.shape()[0], f_s_p.shape()[1], searchSpaceDim);
    if (disp.empty()) return disp;
    svh_array s = HipBridge::describe(f_s_p.array()), t = HipBridge::describe(f_t_p.array()), o = HipBridge::describe(disp), ini = HipBridge::describe(initial);
    if (!HipBridge::check(svh_cacheless_patch_match_init(HipBridge::context(), &p, &s, &t, nIter, nRandomSearch, HipBridge::nextPatchMatchSeed(),
                                                         initial.empty() ? nullptr : &ini, &o, nullptr)))
        return Multidim::Array<disp_t, 3>();
    return disp;
}

// patchMatch<matchFunc, searchSpaceDim>(feature_vol_s, feature_vol_t, searchOffset, nIter, nRandomSearch, initializer, randcache), patchmatch.h:496-558:
// PatchMatch on feature volumes the caller built (benchmarkStereoMatchingModels.cpp:187-199 passes unfolded images); the reference puts
// its cached cost volume behind it, whose values are those of the cacheless one.
template <matchingFunctions matchFunc, int searchSpaceDim, class T_S, class T_T, Multidim::ArrayDataAccessConstness C_S, Multidim::ArrayDataAccessConstness C_T,
          class InitializerT = std::nullopt_t, class RandCacheT = std::nullopt_t>
Multidim::Array<disp_t, 3> patchMatch(Multidim::Array<T_S, 3, C_S> const &feature_vol_s_p, Multidim::Array<T_T, 3, C_T> const &feature_vol_t_p,
                                      searchOffset<searchSpaceDim> searchOffset, int nIter = 5, int nRandomSearch = 4, InitializerT initializer = std::nullopt,
                                      RandCacheT = std::nullopt) {
    static_assert(searchSpaceDim == 1 or searchSpaceDim == 2, "patchMatch function can only be used to search in 1 or two dimension !");
    static_assert(std::is_same_v<std::remove_const_t<T_S>, float> && std::is_same_v<std::remove_const_t<T_T>, float>,
                  "libstevi_hip: PatchMatch takes float feature volumes");
    static_assert(!MatchingFunctionTraits<matchFunc>::isCensusBased && HipBridge::onGpuPath<matchFunc>(), "libstevi_hip: PatchMatch takes the float matching functions (CC ... ZSAD)");
    Multidim::Array<disp_t, 3> disp;
    if (feature_vol_s_p.shape()[2] != feature_vol_t_p.shape()[2]) return disp;                          // :529-531
    if (searchSpaceDim == 1 && feature_vol_s_p.shape()[0] != feature_vol_t_p.shape()[0]) return disp;   // :533-537
    svh_on_demand_params p{};
    p.match_func = static_cast<int>(matchFunc);
    p.search_dims = searchSpaceDim;
    HipBridge::offsetsToParams(searchOffset, p);
    Multidim::Array<disp_t, 3> initial = HipBridge::initialSolution(initializer, feature_vol_s_p, feature_vol_t_p);
    disp = Multidim::Array<disp_t, 3>(feature_vol_s_p.shape()[0], feature_vol_s_p.shape()[1], searchSpaceDim);
    if (disp.empty()) return disp;
    svh_array s = HipBridge::describe(feature_vol_s_p), t = HipBridge::describe(feature_vol_t_p), o = HipBridge::describe(disp), ini = HipBridge::describe(initial);
    if (!HipBridge::check(svh_patch_match(HipBridge::context(), &p, &s, &t, nIter, nRandomSearch, HipBridge::nextPatchMatchSeed(), initial.empty() ? nullptr : &ini, &o, nullptr)))
        return Multidim::Array<disp_t, 3>();
    return disp;
}

} // namespace Correlation
} // namespace StereoVision

#endif // STEREOVISION_PATCHMATCH_H

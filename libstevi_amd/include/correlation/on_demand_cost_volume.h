// Drop-in for LibStevi's correlation/on_demand_cost_volume.h, the part examples/stereo-match uses: SearchSpaceBase / FixedSearchSpace
// and the cacheless on-demand cost volume with its truncatedCostVolume (evaluated on the GPU, as written in the reference).
#ifndef STEREOVISION_ON_DEMAND_COST_VOLUME_H
#define STEREOVISION_ON_DEMAND_COST_VOLUME_H

#include <array>

#include "./on_demand_features_volume.h"

namespace StereoVision {
namespace Correlation {

// SearchSpaceBase / FixedSearchSpace<Ds...>, on_demand_cost_volume.h (search-space part)
class SearchSpaceBase {
  public:
    enum DimType { Ignored, Search, Feature };
    struct SearchDim {
        SearchDim(int pMin, int pMax) : min(pMin), max(pMax) {}
        int min, max;
    };
    struct IgnoredDim {};
    struct FeatureDim {};
};

template <typename... Ds> class FixedSearchSpace : public SearchSpaceBase {
  public:
    static constexpr int nDim = sizeof...(Ds);
    explicit FixedSearchSpace() : _isValid(false) {}
    explicit FixedSearchSpace(Ds... dims) : _isValid(true) {
        int k = 0;
        (set(k++, dims), ...);
    }
    static constexpr int nDimsOfType(DimType type) { return ((typeOf<Ds>() == type ? 1 : 0) + ...); }
    static constexpr int featuresDim() {
        int idx = -1, k = 0;
        ((typeOf<Ds>() == Feature ? (idx = k, k++) : k++), ...);
        return idx;
    }
    DimType getDimType(int dim) const { return _dimType[dim]; }
    int getDimMinSearchRange(int dim) const { return _min[dim]; }
    int getDimMaxSearchRange(int dim) const { return _max[dim]; }
    int dimRange(int dim) const { return _dimType[dim] == Search ? _max[dim] - _min[dim] + 1 : 0; }
    int disp2idx(int dim, int disp) const { return disp - _min[dim]; }
    int idx2disp(int dim, int idx) const { return _min[dim] + idx; }
    bool isValid() const { return _isValid; }

  private:
    template <class D> static constexpr DimType typeOf() {
        return std::is_same_v<D, SearchDim> ? Search : (std::is_same_v<D, FeatureDim> ? Feature : Ignored);
    }
    void set(int k, SearchDim const &d) { _dimType[k] = Search; _min[k] = d.min; _max[k] = d.max; }
    void set(int k, IgnoredDim const &) { _dimType[k] = Ignored; _min[k] = 0; _max[k] = 0; }
    void set(int k, FeatureDim const &) { _dimType[k] = Feature; _min[k] = 0; _max[k] = 0; }
    bool _isValid;
    std::array<DimType, nDim> _dimType{};
    std::array<int, nDim> _min{}, _max{};
};

namespace HipBridge {
// on-demand volumes the GPU can evaluate: float (row, column, channel) images, channel axis as feature dimension, decorated as
// the matching function asks
template <matchingFunctions matchFunc, class FV> struct OnDemandSupport { static constexpr bool value = false; };
template <matchingFunctions matchFunc, bool ZM, bool N, Multidim::ArrayDataAccessConstness c>
struct OnDemandSupport<matchFunc, OnDemandDecoratedFeaturesVolume<ZNFeaturesVolumeDecorator<ZM, N>, float, 3, c, 2>> {
    static constexpr bool value = ZM == MatchingFunctionTraits<matchFunc>::ZeroMean && N == MatchingFunctionTraits<matchFunc>::Normalized &&
                                  !MatchingFunctionTraits<matchFunc>::isCensusBased;
};
template <class FV_S, class FV_T> inline void windowRadii(FV_S const &s, FV_T const &t, int &v_r, int &h_r) {
    int vt = 0, ht = 0;
    if (!s.rectangularWindow(v_r, h_r) || !t.rectangularWindow(vt, ht) || vt != v_r || ht != h_r)
        throw std::runtime_error("libstevi_hip: on-demand volumes take the same full rectangular window (rows, columns, every channel) on both images");
}
} // namespace HipBridge

// CachelessOnDemandCostVolume<matchFunc, T_CV, F_V_S_T, F_V_T_T, Ds...>, on_demand_cost_volume.h:345-612
template <matchingFunctions matchFunc, class T_CV, class F_V_S_T, class F_V_T_T, typename... Ds> class CachelessOnDemandCostVolume {
  public:
    using SearchSpaceType = FixedSearchSpace<Ds...>;
    static constexpr int nDim = SearchSpaceType::nDim;
    static constexpr int nSearchDim = SearchSpaceType::nDimsOfType(SearchSpaceBase::Search);
    static constexpr int nCostVolDim = nDim + nSearchDim - 1;
    static_assert(nDim == 3 && (nSearchDim == 1 || nSearchDim == 2), "libstevi_hip: stereo (IgnoredDim, SearchDim, FeatureDim) or flow (SearchDim, SearchDim, FeatureDim)");
    static_assert(HipBridge::OnDemandSupport<matchFunc, F_V_S_T>::value && HipBridge::OnDemandSupport<matchFunc, F_V_T_T>::value,
                  "libstevi_hip: on-demand volumes are evaluated for float images decorated with ZNFeaturesVolumeDecorator<ZeroMean, Normalized> of the matching function");
    static_assert(std::is_same_v<T_CV, float>, "libstevi_hip: cost volumes are float");

    explicit CachelessOnDemandCostVolume() : _source(nullptr), _target(nullptr), _search_space() {}
    explicit CachelessOnDemandCostVolume(F_V_S_T const &source, F_V_T_T const &target, SearchSpaceType const &searchSpace)
        : _source(&source), _target(&target), _search_space(searchSpace) {}

    inline std::array<int, nCostVolDim> shape() const {
        std::array<int, nCostVolDim> s{};
        s[0] = _source->shape()[0];
        s[1] = _source->shape()[1];
        for (int i = 0, k = 0; i < nDim; i++)
            if (_search_space.getDimType(i) == SearchSpaceBase::Search) s[2 + k++] = _search_space.dimRange(i);
        return s;
    }
    inline SearchSpaceType const &searchSpace() const { return _search_space; }

    svh_on_demand_params params() const {
        svh_on_demand_params p{};
        p.match_func = static_cast<int>(matchFunc);
        p.search_dims = nSearchDim;
        HipBridge::windowRadii(*_source, *_target, p.v_radius, p.h_radius);
        if (nSearchDim == 2) {
            p.lower0 = _search_space.getDimMinSearchRange(0);
            p.upper0 = _search_space.getDimMaxSearchRange(0);
        }
        p.lower1 = _search_space.getDimMinSearchRange(1);
        p.upper1 = _search_space.getDimMaxSearchRange(1);
        return p;
    }

    // truncatedCostVolume(disp, radius), :474-596 (as written there: see svh_on_demand_truncated_cost_volume)
    template <Multidim::ArrayDataAccessConstness viewConstness>
    Multidim::Array<T_CV, nCostVolDim> truncatedCostVolume(Multidim::Array<disp_t, nDim, viewConstness> const &disp, int radius = 1) const {
        if (disp.shape()[nDim - 1] != nSearchDim) return Multidim::Array<T_CV, nCostVolDim>(); // :478-480
        std::array<int, nCostVolDim> tshape{};
        tshape[0] = _source->shape()[0];
        tshape[1] = _source->shape()[1];
        for (int i = 0; i < nSearchDim; i++) tshape[2 + i] = 2 * radius + 1;
        auto tcv = HipBridge::makeResult<Multidim::Array<T_CV, nCostVolDim>>(tshape);
        if (tcv.empty()) return tcv;
        const svh_on_demand_params p = params();
        svh_array s = HipBridge::describe(_source->array()), t = HipBridge::describe(_target->array()), d = HipBridge::describe(disp), o = HipBridge::describe(tcv);
        if (!HipBridge::check(svh_on_demand_truncated_cost_volume(HipBridge::context(), &p, &s, &t, &d, radius, &o))) return Multidim::Array<T_CV, nCostVolDim>();
        return tcv;
    }

  protected:
    F_V_S_T const *_source;
    F_V_T_T const *_target;
    SearchSpaceType _search_space;
};

template <matchingFunctions matchFunc, class T_CV, class F_V_S_T, class F_V_T_T>
using CachelessOnDemandStereoCostVolume =
    CachelessOnDemandCostVolume<matchFunc, T_CV, F_V_S_T, F_V_T_T, SearchSpaceBase::IgnoredDim, SearchSpaceBase::SearchDim, SearchSpaceBase::FeatureDim>;
template <matchingFunctions matchFunc, class T_CV, class F_V_S_T, class F_V_T_T>
using CachelessOnDemandImageFlowVolume =
    CachelessOnDemandCostVolume<matchFunc, T_CV, F_V_S_T, F_V_T_T, SearchSpaceBase::SearchDim, SearchSpaceBase::SearchDim, SearchSpaceBase::FeatureDim>;

} // namespace Correlation
} // namespace StereoVision

#endif // STEREOVISION_ON_DEMAND_COST_VOLUME_H

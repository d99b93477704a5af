// Multidim::Array -- compatibility subset for the LibStevi correlation/ drop-in headers.
//
// LibStevi takes its array type from the un-vendored dependency MultidimArrays (CMakeLists.txt:62-68 of the
// reference).  When that library is available, put ITS include directory in front of this one and this file is
// never seen.  This header is an independent implementation of the part of the interface the hot path and its
// callers use, written against the call sites in the reference (cited at each member), so that the shim headers in
// ../correlation compile stand-alone:
//
//   shape / strides / element access            cross_correlations.h:220, unfold.h:272, correlation_base.h:434-458
//   views over foreign memory, takePointer()    benchmarkStereoMatchingModels.cpp:202-204, io/image_io.cpp:92
//   subView(DimIndex / DimSlice ...)            cross_correlations.h:227-228, hierarchical.h:160, testCorrelationFilters.cpp:303-307
//   sliceView(dim, index), indexDimView(...)    image_based_refinement.h:399, on_demand_cost_volume.h:164-172
//   buildReshapedView<N>(shape, strides)        cross_correlations.h:659, :677
//   cast<U>()                                   cross_correlations.h:713
//   IndexBlock::setZero / moveToNextIndex       io/image_io.h:56-67
//   IndexConverter<N>                           on_demand_cost_volume.h:208-213, :497-502
//   Multidim::ConstView / NonConstView          matching_costs.h:59, cross_correlations.h:654-658
//
// The marshalling code of the shims (stevi_hip_bridge.h) only relies on what those call sites show: shape(), strides(), the address
// of the first element via atUnchecked(0, ...) (io/image_io.h:96), flatLenght(), empty().
#pragma once

#include <array>
#include <cstddef>
#include <cstdlib>
#include <memory>
#include <set>
#include <stdexcept>
#include <type_traits>

#define STEVI_HIP_MULTIDIM_COMPAT 1 // (the shims' bridge asks: is this the compatibility header, with its result-memory hook?)

namespace Multidim {

namespace detail {
// Where the arrays the drop-in shims RETURN get their memory.  Every function of the reference returns a fresh owning array that it has
// written completely; the shims construct those inside a ResultAllocationScope, and while one is open on the calling thread an
// allocation of `from_bytes` and more comes from `alloc` (libstevi_hip's page-locked blocks, installed by stevi_hip_bridge.h) and is not
// cleared first.  Arrays constructed anywhere else are value-initialised new[] memory, as before.  Unset (the default): new[] always.
struct ResultMemory {
    void *(*alloc)(std::size_t bytes) = nullptr;
    void (*release)(void *) = nullptr;
    std::size_t from_bytes = std::size_t(1) << 20;
};
inline ResultMemory &result_memory() {
    static ResultMemory m;
    return m;
}
inline int &result_scope_depth() {
    static thread_local int depth = 0;
    return depth;
}
struct ResultAllocationScope {
    ResultAllocationScope() { ++result_scope_depth(); }
    ~ResultAllocationScope() { --result_scope_depth(); }
    ResultAllocationScope(ResultAllocationScope const &) = delete;
    ResultAllocationScope &operator=(ResultAllocationScope const &) = delete;
};
// what the copies and views of one allocation share besides the memory
struct Control {
    bool released = false;            // takePointer() handed the memory to the caller
    void (*release)(void *) = nullptr; // not new[] memory: given back through this
};
} // namespace detail

enum class AccessCheck { Check, Nocheck };
enum ArrayDataAccessConstness { NonConstView, ConstView };
using array_size_t = int;

// an index / shape / stride tuple: a std::array with the two iteration helpers the reference's IO code uses (io/image_io.h:56-67)
template <int nDim> struct IndexTuple : public std::array<array_size_t, nDim> {
    using Base = std::array<array_size_t, nDim>;
    IndexTuple() : Base() {}
    IndexTuple(Base const &b) : Base(b) {}
    template <typename... Ds, typename = std::enable_if_t<sizeof...(Ds) == nDim && (std::is_convertible_v<Ds, array_size_t> && ...)>>
    IndexTuple(Ds... v) : Base{{static_cast<array_size_t>(v)...}} {}
    void setZero() { this->fill(0); }
    // next index in an order that visits every index of `shape` exactly once starting from zero (the first index varies fastest);
    // wraps to zero after the last one
    void moveToNextIndex(Base const &shape) {
        for (int k = 0; k < nDim; k++) {
            if (++(*this)[k] < shape[k]) return;
            (*this)[k] = 0;
        }
    }
};

struct DimIndex { // one index of a dimension: the dimension disappears from the view
    array_size_t index;
    explicit DimIndex(array_size_t i) : index(i) {}
};
struct DimSlice { // [start, end) of a dimension (the whole of it by default): the dimension stays (testCorrelationFilters.cpp:303: DimSlice(0, 2 v_radius + 1))
    array_size_t start, end, step;
    bool whole;
    DimSlice() : start(0), end(0), step(1), whole(true) {}
    DimSlice(array_size_t s, array_size_t e, array_size_t st = 1) : start(s), end(e), step(st), whole(false) {}
};

namespace detail {
template <class... A> struct count_slices;
template <> struct count_slices<> { static constexpr int value = 0; };
template <class A0, class... A> struct count_slices<A0, A...> {
    static constexpr int value = (std::is_same_v<std::decay_t<A0>, DimSlice> ? 1 : 0) + count_slices<A...>::value;
};
} // namespace detail

template <class T, int nDim, ArrayDataAccessConstness viewConstness = NonConstView> class Array {
  public:
    using ShapeBlock = IndexTuple<nDim>;
    using IndexBlock = IndexTuple<nDim>;
    using ScalarT = T;
    template <class U, int M, ArrayDataAccessConstness C> friend class Array;

    Array() : _data(nullptr) {
        _shape.fill(0);
        _strides.fill(0);
    }

    template <typename... Ds, typename = std::enable_if_t<sizeof...(Ds) == nDim && (std::is_integral_v<Ds> && ...)>>
    explicit Array(Ds... dims) : Array(ShapeBlock(static_cast<array_size_t>(dims)...)) {}

    // dense, last index fastest
    explicit Array(ShapeBlock const &shape) : _shape(shape) {
        array_size_t s = 1;
        for (int k = nDim - 1; k >= 0; k--) {
            _strides[k] = s;
            s *= _shape[k];
        }
        allocate();
    }
    explicit Array(std::array<array_size_t, nDim> const &shape) : Array(ShapeBlock(shape)) {}

    Array(ShapeBlock const &shape, ShapeBlock const &strides) : _shape(shape), _strides(strides) { allocate(); }

    // view over (or adoption of) foreign memory
    Array(T *data, ShapeBlock const &shape, ShapeBlock const &strides, bool manage = false) : _shape(shape), _strides(strides), _data(data) {
        if (manage) _owner.reset(data, [](T *p) { delete[] p; });
    }

    // a view that may write can be read through a view that may not (cross_correlations.h:659: NonConstView result stored as ConstView)
    template <ArrayDataAccessConstness C2, typename = std::enable_if_t<C2 == NonConstView && viewConstness == ConstView>>
    Array(Array<T, nDim, C2> const &o) : _shape(o._shape), _strides(o._strides), _data(o._data), _owner(o._owner) {}

    ShapeBlock const &shape() const { return _shape; }
    ShapeBlock const &strides() const { return _strides; }
    bool empty() const { return _data == nullptr || flatLenght() == 0; }
    std::size_t flatLenght() const { // (sic) spelled as in MultidimArrays
        std::size_t n = 1;
        for (int k = 0; k < nDim; k++) n *= static_cast<std::size_t>(_shape[k] > 0 ? _shape[k] : 0);
        return n;
    }
    bool isDense() const {
        array_size_t s = 1;
        for (int k = nDim - 1; k >= 0; k--) {
            if (_shape[k] != 1 && _strides[k] != s) return false;
            s *= _shape[k];
        }
        return true;
    }

    // gives the memory up: the array becomes empty, the caller owns what the pointer points to (new[])
    // (benchmarkStereoMatchingModels.cpp:202-204: the pointer goes into Array(data, shape, strides, true))
    T *takePointer() {
        T *p = _data;
        if (_owner) {
            if (_owner.use_count() != 1) throw std::logic_error("Multidim::Array::takePointer on shared memory");
            if (_released->release) { // result memory of the shims: the caller is promised new[] memory, so it gets a copy
                const std::size_t span = memorySpan();
                p = new T[span];
                for (std::size_t e = 0; e < span; e++) p[e] = _data[e];
            } else {
                _released->released = true;
            }
        }
        _owner.reset();
        _released.reset();
        _data = nullptr;
        return p;
    }

    // trailing indices may be left out (they are zero): io/image_io.h:96 takes &atUnchecked(0) of an nDim array
    template <AccessCheck c = AccessCheck::Check, typename... Is, typename = std::enable_if_t<(std::is_integral_v<Is> && ...) && sizeof...(Is) <= nDim>>
    T &at(Is... idx) {
        static_assert(viewConstness == NonConstView, "at() on a const view");
        return _data[offset<c>(padded(idx...))];
    }
    template <AccessCheck c = AccessCheck::Check> T &at(std::array<array_size_t, nDim> const &idx) {
        static_assert(viewConstness == NonConstView, "at() on a const view");
        return _data[offset<c>(idx)];
    }
    template <typename... Is> T &atUnchecked(Is... idx) { return at<AccessCheck::Nocheck>(idx...); }

    template <AccessCheck c = AccessCheck::Check, typename... Is, typename = std::enable_if_t<(std::is_integral_v<Is> && ...) && sizeof...(Is) <= nDim>>
    T value(Is... idx) const {
        return _data[offset<c>(padded(idx...))];
    }
    template <AccessCheck c = AccessCheck::Check> T value(std::array<array_size_t, nDim> const &idx) const { return _data[offset<c>(idx)]; }
    template <typename... Is> T valueUnchecked(Is... idx) const { return value<AccessCheck::Nocheck>(idx...); }

    T valueOrAlt(std::array<array_size_t, nDim> const &idx, T alt) const {
        for (int k = 0; k < nDim; k++)
            if (idx[k] < 0 || idx[k] >= _shape[k]) return alt;
        return _data[offset<AccessCheck::Nocheck>(idx)];
    }

    // ---- views (share the memory and its lifetime) ----------------------------------------------------------------------
    // one argument per dimension: DimIndex removes the dimension, DimSlice keeps [start, end) of it
    template <typename... Args, typename = std::enable_if_t<sizeof...(Args) == nDim>>
    Array<T, detail::count_slices<Args...>::value, viewConstness> subView(Args... args) const {
        constexpr int M = detail::count_slices<Args...>::value;
        static_assert(M >= 1, "subView keeps at least one dimension");
        Array<T, M, viewConstness> out;
        std::size_t off = 0;
        int k = 0, m = 0;
        (sub_arg(args, k, m, off, out), ...);
        out._data = _data ? _data + off : nullptr;
        out._owner = _owner;
        out._released = _released;
        return out;
    }
    // the array without dimension `dim`, at index `idx` of it
    Array<T, (nDim > 1 ? nDim - 1 : 1), viewConstness> sliceView(int dim, array_size_t idx) const {
        static_assert(nDim > 1, "sliceView of a 1-D array");
        if (dim < 0 || dim >= nDim || idx < 0 || idx >= _shape[dim]) throw std::out_of_range("Multidim::Array::sliceView");
        Array<T, nDim - 1, viewConstness> out;
        for (int k = 0, m = 0; k < nDim; k++) {
            if (k == dim) continue;
            out._shape[m] = _shape[k];
            out._strides[m++] = _strides[k];
        }
        out._data = _data ? _data + static_cast<std::size_t>(idx) * static_cast<std::size_t>(_strides[dim]) : nullptr;
        out._owner = _owner;
        out._released = _released;
        return out;
    }
    // the 1-D line along dimension `dim` through the position `pos` (the indices of the other dimensions, in order)
    Array<T, 1, ConstView> indexDimView(int dim, std::array<array_size_t, (nDim > 1 ? nDim - 1 : 1)> const &pos) const {
        static_assert(nDim > 1, "indexDimView of a 1-D array");
        if (dim < 0 || dim >= nDim) throw std::out_of_range("Multidim::Array::indexDimView");
        std::size_t off = 0;
        for (int k = 0, m = 0; k < nDim; k++) {
            if (k == dim) continue;
            off += static_cast<std::size_t>(pos[m++]) * static_cast<std::size_t>(_strides[k]);
        }
        Array<T, 1, ConstView> out;
        out._shape[0] = _shape[dim];
        out._strides[0] = _strides[dim];
        out._data = _data ? _data + off : nullptr;
        out._owner = _owner;
        out._released = _released;
        return out;
    }
    // the same memory under another shape and strides (the caller answers for their consistency)
    template <int M> Array<T, M, viewConstness> buildReshapedView(std::array<array_size_t, M> const &shape, std::array<array_size_t, M> const &strides) const {
        Array<T, M, viewConstness> out;
        out._shape = IndexTuple<M>(shape);
        out._strides = IndexTuple<M>(strides);
        out._data = _data;
        out._owner = _owner;
        out._released = _released;
        return out;
    }

    template <class U> Array<U, nDim> cast() const {
        Array<U, nDim> out(_shape);
        std::array<array_size_t, nDim> idx;
        idx.fill(0);
        const std::size_t n = flatLenght();
        for (std::size_t e = 0; e < n; e++) {
            out.template at<AccessCheck::Nocheck>(idx) = static_cast<U>(value<AccessCheck::Nocheck>(idx));
            for (int k = nDim - 1; k >= 0; k--) {
                if (++idx[k] < _shape[k]) break;
                idx[k] = 0;
            }
        }
        return out;
    }

  private:
    void allocate() {
        const std::size_t n = flatLenght();
        if (n == 0) {
            _data = nullptr;
            return;
        }
        const std::size_t span = memorySpan();
        // the memory is released with the last array or view of it, unless takePointer() handed it to the caller first
        auto released = std::make_shared<detail::Control>();
        _released = released;
        if constexpr (std::is_trivially_copyable_v<T> && std::is_trivially_destructible_v<T>) {
            detail::ResultMemory const &rm = detail::result_memory();
            if (detail::result_scope_depth() > 0 && rm.alloc && rm.release && span * sizeof(T) >= rm.from_bytes) {
                if (void *block = rm.alloc(span * sizeof(T))) {
                    released->release = rm.release;
                    _owner.reset(static_cast<T *>(block), [released](T *p) { released->release(p); });
                    _data = _owner.get();
                    return;
                }
            }
        }
        _owner.reset(new T[span](), [released](T *p) {
            if (!released->released) delete[] p;
        });
        _data = _owner.get();
    }

    // the extent in memory may exceed the element count for non-dense strides
    std::size_t memorySpan() const {
        std::size_t span = 1;
        for (int k = 0; k < nDim; k++) span += static_cast<std::size_t>(_shape[k] - 1) * static_cast<std::size_t>(_strides[k]);
        return span;
    }

    template <typename... Is> std::array<array_size_t, nDim> padded(Is... idx) const {
        std::array<array_size_t, nDim> full;
        full.fill(0);
        array_size_t given[] = {static_cast<array_size_t>(idx)..., 0};
        for (std::size_t k = 0; k < sizeof...(Is); k++) full[k] = given[k];
        return full;
    }

    template <int M> void sub_arg(DimIndex const &a, int &k, int &, std::size_t &off, Array<T, M, viewConstness> &) const {
        if (a.index < 0 || a.index >= _shape[k]) throw std::out_of_range("Multidim::Array::subView index");
        off += static_cast<std::size_t>(a.index) * static_cast<std::size_t>(_strides[k]);
        k++;
    }
    template <int M> void sub_arg(DimSlice const &a, int &k, int &m, std::size_t &off, Array<T, M, viewConstness> &out) const {
        const array_size_t start = a.whole ? 0 : a.start, end = a.whole ? _shape[k] : a.end, step = a.step > 0 ? a.step : 1;
        if (start < 0 || end > _shape[k] || start > end) throw std::out_of_range("Multidim::Array::subView slice");
        off += static_cast<std::size_t>(start) * static_cast<std::size_t>(_strides[k]);
        out._shape[m] = (end - start + step - 1) / step;
        out._strides[m] = _strides[k] * step;
        k++;
        m++;
    }

    template <AccessCheck c> std::size_t offset(std::array<array_size_t, nDim> const &idx) const {
        std::size_t o = 0;
        for (int k = 0; k < nDim; k++) {
            if (c == AccessCheck::Check && (idx[k] < 0 || idx[k] >= _shape[k])) throw std::out_of_range("Multidim::Array index");
            o += static_cast<std::size_t>(idx[k]) * static_cast<std::size_t>(_strides[k]);
        }
        return o;
    }

    ShapeBlock _shape, _strides;
    T *_data;
    std::shared_ptr<T> _owner;
    std::shared_ptr<detail::Control> _released;
};

// Enumerates the indices of a shape by a flat number (on_demand_cost_volume.h:208-213: one loop, parallel over i, instead of nDim
// nested ones); dimensions in `excluded` keep index 0 and do not count.
template <int nDim> class IndexConverter {
  public:
    using ShapeBlock = std::array<array_size_t, nDim>;
    IndexConverter(ShapeBlock const &shape, std::set<int> const &excluded = {}) : _shape(shape) {
        _n = 1;
        for (int k = 0; k < nDim; k++) {
            _counted[k] = excluded.count(k) == 0;
            if (_counted[k]) _n *= static_cast<std::size_t>(_shape[k] > 0 ? _shape[k] : 0);
        }
    }
    int numberOfPossibleIndices() const { return static_cast<int>(_n); }
    ShapeBlock getIndexFromPseudoFlatId(int flat) const { // last counted dimension fastest
        ShapeBlock idx;
        idx.fill(0);
        std::size_t r = static_cast<std::size_t>(flat);
        for (int k = nDim - 1; k >= 0; k--) {
            if (!_counted[k] || _shape[k] <= 0) continue;
            idx[k] = static_cast<array_size_t>(r % static_cast<std::size_t>(_shape[k]));
            r /= static_cast<std::size_t>(_shape[k]);
        }
        return idx;
    }

  private:
    ShapeBlock _shape;
    std::array<bool, nDim> _counted;
    std::size_t _n;
};
template <std::size_t N> IndexConverter(std::array<array_size_t, N> const &) -> IndexConverter<static_cast<int>(N)>;
template <int N> IndexConverter(IndexTuple<N> const &) -> IndexConverter<N>;

} // namespace Multidim

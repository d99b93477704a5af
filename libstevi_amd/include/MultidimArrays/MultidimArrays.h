// Multidim::Array -- compatibility subset for the LibStevi correlation/ drop-in headers.
//
// LibStevi takes its array type from the un-vendored dependency MultidimArrays (CMakeLists.txt:62-68 of the
// reference).  When that library is available, put ITS include directory in front of this one and this file is
// never seen.  This header is an independent implementation of the part of the interface the hot path and its
// callers use (shape / strides / element access / views over foreign memory), written against the call sites in
// the reference (e.g. cross_correlations.h:220, unfold.h:272, correlation_base.h:434-458), so that the shim
// headers in ../correlation compile stand-alone.
#pragma once

#include <array>
#include <cstddef>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <type_traits>

namespace Multidim {

enum class AccessCheck { Check, Nocheck };
enum ArrayDataAccessConstness { NonConstView, ConstView };
using array_size_t = int;

template <class T, int nDim, ArrayDataAccessConstness viewConstness = NonConstView> class Array {
  public:
    using ShapeBlock = std::array<array_size_t, nDim>;
    using IndexBlock = std::array<array_size_t, nDim>;
    using ScalarT = T;

    Array() : _data(nullptr) {
        _shape.fill(0);
        _strides.fill(0);
    }

    template <typename... Ds, typename = std::enable_if_t<sizeof...(Ds) == nDim && (std::is_integral_v<Ds> && ...)>>
    explicit Array(Ds... dims) : Array(ShapeBlock{static_cast<array_size_t>(dims)...}) {}

    // dense, last index fastest
    explicit Array(ShapeBlock const &shape) : _shape(shape) {
        array_size_t s = 1;
        for (int k = nDim - 1; k >= 0; k--) {
            _strides[k] = s;
            s *= _shape[k];
        }
        allocate();
    }

    Array(ShapeBlock const &shape, ShapeBlock const &strides) : _shape(shape), _strides(strides) { allocate(); }

    // view over (or adoption of) foreign memory
    Array(T *data, ShapeBlock const &shape, ShapeBlock const &strides, bool manage = false) : _shape(shape), _strides(strides), _data(data) {
        if (manage) _owner.reset(data, [](T *p) { delete[] p; });
    }

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

    T *data() { return _data; }
    T const *data() const { return _data; }

    template <AccessCheck c = AccessCheck::Check, typename... Is> T &at(Is... idx) {
        static_assert(viewConstness == NonConstView, "at() on a const view");
        return _data[offset<c>(IndexBlock{static_cast<array_size_t>(idx)...})];
    }
    template <AccessCheck c = AccessCheck::Check> T &at(IndexBlock const &idx) { return _data[offset<c>(idx)]; }
    template <typename... Is> T &atUnchecked(Is... idx) { return at<AccessCheck::Nocheck>(idx...); }

    template <AccessCheck c = AccessCheck::Check, typename... Is> T value(Is... idx) const {
        return _data[offset<c>(IndexBlock{static_cast<array_size_t>(idx)...})];
    }
    template <AccessCheck c = AccessCheck::Check> T value(IndexBlock const &idx) const { return _data[offset<c>(idx)]; }
    template <typename... Is> T valueUnchecked(Is... idx) const { return value<AccessCheck::Nocheck>(idx...); }

    T valueOrAlt(IndexBlock const &idx, T alt) const {
        for (int k = 0; k < nDim; k++)
            if (idx[k] < 0 || idx[k] >= _shape[k]) return alt;
        return _data[offset<AccessCheck::Nocheck>(idx)];
    }

    template <class U> Array<U, nDim> cast() const {
        Array<U, nDim> out(_shape);
        IndexBlock idx;
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
        // the extent in memory may exceed the element count for non-dense strides
        std::size_t span = 1;
        for (int k = 0; k < nDim; k++) span += static_cast<std::size_t>(_shape[k] - 1) * static_cast<std::size_t>(_strides[k]);
        _owner.reset(new T[span](), [](T *p) { delete[] p; });
        _data = _owner.get();
    }

    template <AccessCheck c> std::size_t offset(IndexBlock const &idx) const {
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
};

} // namespace Multidim

// Marshalling between Multidim::Array and the C ABI of libstevi_hip.so (include/stevi_hip.h).
// Not a reference header: it is the glue the drop-in headers in this directory share.
#pragma once

#include <MultidimArrays/MultidimArrays.h>

#include <array>
#include <cstdint>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "../../../include/stevi_hip.h"

namespace StereoVision {
namespace Correlation {
namespace HipBridge {

// The reference's functions are stateless and re-entrant.  So are these: every calling thread works on its own context (device
// workspace cache + stream), created on the thread's first call on the then-current HIP device and destroyed with the thread;
// a caller that manages contexts itself (its own stream, several devices) installs one for a scope with ContextScope.  There is no
// process-wide state and no lock shared between threads.
struct ThreadState {
    svh_context *own = nullptr;     // created lazily, owned by the thread
    svh_context *current = nullptr; // what context() hands out: a ContextScope's context, else `own`
    std::mutex unshared;            // lock() below: kept so that the shims read the same as before; never contended
    ~ThreadState() {
        if (own) svh_context_destroy(own);
    }
};
inline ThreadState &thread_state() {
    static thread_local ThreadState ts;
    return ts;
}

inline svh_context *context() {
    ThreadState &ts = thread_state();
    if (ts.current) return ts.current;
    if (!ts.own) {
        int st = svh_context_create(&ts.own, -1, nullptr);
        if (st != SVH_OK) throw std::runtime_error(std::string("libstevi_hip: ") + svh_status_string(st) + " (there is no CPU fallback)");
    }
    return ts.own;
}

// run the calls of this thread on a caller-supplied context for the lifetime of the object
class ContextScope {
  public:
    explicit ContextScope(svh_context *ctx) : prev(thread_state().current) { thread_state().current = ctx; }
    ~ContextScope() { thread_state().current = prev; }
    ContextScope(ContextScope const &) = delete;
    ContextScope &operator=(ContextScope const &) = delete;

  private:
    svh_context *prev;
};

inline std::mutex &lock() { return thread_state().unshared; }

template <class T> struct DType {
    static_assert(sizeof(T) == 0, "libstevi_hip: this element type has no GPU path (float or uint8 images, float volumes, int32 index "
                                  "maps and uint32 census words are supported)");
};
template <> struct DType<uint8_t> { static constexpr int value = SVH_U8; }; // images only, see imageTypeOnGpuPath
template <> struct DType<float> { static constexpr int value = SVH_F32; };
template <> struct DType<int32_t> { static constexpr int value = SVH_I32; };
template <> struct DType<uint32_t> { static constexpr int value = SVH_U32; };

template <class T, int N, Multidim::ArrayDataAccessConstness C> inline svh_array describe(Multidim::Array<T, N, C> const &a) {
    static_assert(N <= SVH_MAX_DIMS, "too many dimensions");
    svh_array d{};
    d.data = const_cast<T *>(a.data());
    d.ndim = N;
    d.dtype = DType<std::remove_const_t<T>>::value;
    d.memspace = SVH_HOST;
    for (int k = 0; k < N; k++) {
        d.shape[k] = a.shape()[k];
        d.strides[k] = a.strides()[k];
    }
    return d;
}

// SVH_OK -> true, SVH_EMPTY_RESULT -> false (caller returns the empty array the reference returns), else throw
inline bool check(int status) {
    if (status == SVH_OK) return true;
    if (status == SVH_EMPTY_RESULT) return false;
    std::string msg = svh_last_error(context());
    throw std::runtime_error("libstevi_hip: " + (msg.empty() ? std::string(svh_status_string(status)) : msg));
}

// A dense (last index fastest) array in GPU memory, for call chains that should not cross PCIe between the reference's functions:
// the overloads of sgmCostVolume, extractSelectedIndex, selectedIndexToDisp, truncatedCostVolume, refineDispCostInterpolation and
// selectedCost that take DeviceArray arguments return DeviceArray results (same names, same template parameters), and
// unfoldBasedCostVolumeOnDevice / featureVolume2CostVolumeOnDevice start such a chain from host images.  download() is the one
// transfer back.  The memory belongs to the context it was allocated on and is freed with the last copy of the handle.
template <class T, int N> class DeviceArray {
  public:
    using ShapeBlock = std::array<int, N>;
    DeviceArray() { _shape.fill(0); }
    explicit DeviceArray(ShapeBlock const &shape) : _shape(shape) {
        if (flatLenght() == 0) return;
        svh_context *ctx = context();
        void *p = nullptr;
        if (!check(svh_device_alloc(ctx, flatLenght() * sizeof(T), &p))) return;
        _mem.reset(static_cast<T *>(p), [ctx](T *q) { svh_device_free(ctx, q); });
    }
    static DeviceArray upload(Multidim::Array<T, N> const &host) {
        if (!host.isDense()) throw std::runtime_error("libstevi_hip: DeviceArray::upload needs a dense array");
        ShapeBlock s;
        for (int k = 0; k < N; k++) s[k] = host.shape()[k];
        DeviceArray d(s);
        if (!d.empty()) check(svh_device_upload(context(), d._mem.get(), host.data(), d.flatLenght() * sizeof(T)));
        return d;
    }
    Multidim::Array<T, N> download() const {
        typename Multidim::Array<T, N>::ShapeBlock s;
        for (int k = 0; k < N; k++) s[k] = _shape[k];
        Multidim::Array<T, N> host(s);
        if (!empty()) check(svh_device_download(context(), host.data(), _mem.get(), flatLenght() * sizeof(T)));
        return host;
    }
    ShapeBlock const &shape() const { return _shape; }
    bool empty() const { return !_mem || flatLenght() == 0; }
    std::size_t flatLenght() const { // (sic) as Multidim::Array
        std::size_t n = 1;
        for (int k = 0; k < N; k++) n *= static_cast<std::size_t>(_shape[k] > 0 ? _shape[k] : 0);
        return n;
    }
    T *data() const { return _mem.get(); }

  private:
    ShapeBlock _shape;
    std::shared_ptr<T> _mem;
};

template <class T, int N> inline svh_array describe(DeviceArray<T, N> const &a) {
    static_assert(N <= SVH_MAX_DIMS, "too many dimensions");
    svh_array d{};
    d.data = a.data();
    d.ndim = N;
    d.dtype = DType<std::remove_const_t<T>>::value;
    d.memspace = SVH_DEVICE;
    int64_t stride = 1;
    for (int k = N - 1; k >= 0; k--) {
        d.shape[k] = a.shape()[k];
        d.strides[k] = stride;
        stride *= a.shape()[k];
    }
    return d;
}

// Image element types per matching function: float always; uint8 where the reference keeps the plain samples as features
// and casts to float inside the comparison (CENSUS, HAMMING, CC, SSD, SAD: matching_costs.h:749-783), so that widening on the
// device is exact.  uint8 with a normalised or zero-mean function takes the reference's int16 path, which its own comparison
// test skips (testCorrelationFilters.cpp:1249).
template <class T_I, bool ZeroMean, bool Normalized> constexpr bool imageTypeOnGpuPath() {
    using T = std::remove_const_t<T_I>;
    return std::is_same_v<T, float> || (std::is_same_v<T, uint8_t> && !ZeroMean && !Normalized);
}

} // namespace HipBridge
} // namespace Correlation
} // namespace StereoVision

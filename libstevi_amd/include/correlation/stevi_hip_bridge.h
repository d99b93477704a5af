// Marshalling between Multidim::Array and the C ABI of libstevi_hip.so (include/stevi_hip.h).
// Not a reference header: it is the glue the drop-in headers in this directory share.
#pragma once

#include <MultidimArrays/MultidimArrays.h>

#include <array>
#include <atomic>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

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
    ~ThreadState() {
        if (own) svh_context_destroy(own);
    }
};
inline ThreadState &thread_state() {
    static thread_local ThreadState ts;
    return ts;
}

// Result arrays of a megabyte and more live in page-locked host memory (svh_host_alloc): the DMA engines write them, and read them when
// the caller passes them to the next function of a chain, at the link's rate.  With this tree's MultidimArrays compatibility header the
// shims construct their results inside a ResultScope; with the real MultidimArrays (whose arrays allocate as they see fit) ResultScope is
// empty and the library stages the transfers of pageable memory itself (svh_transfer.hip).
#ifdef STEVI_HIP_MULTIDIM_COMPAT
inline void install_result_memory() {
    static const bool once = [] {
        Multidim::detail::ResultMemory &rm = Multidim::detail::result_memory();
        rm.alloc = [](std::size_t bytes) -> void * {
            void *p = nullptr;
            return svh_host_alloc(bytes, &p) == SVH_OK ? p : nullptr;
        };
        rm.release = [](void *p) { (void)svh_host_free(p); };
        return true;
    }();
    (void)once;
}
struct ResultScope {
    ResultScope() { install_result_memory(); }
    Multidim::detail::ResultAllocationScope scope;
};
#else
struct ResultScope {};
#endif

// an array a shim is about to have the library fill and then return
template <class A, class... Args> inline A makeResult(Args &&...args) {
    ResultScope scope;
    (void)scope;
    return A(std::forward<Args>(args)...);
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

template <class T> struct DType {
    static_assert(sizeof(T) == 0, "libstevi_hip: this element type has no GPU path (float or uint8 images, float volumes, int32 index "
                                  "maps and uint32 census words are supported)");
};
template <> struct DType<uint8_t> { static constexpr int value = SVH_U8; }; // images only, see imageTypeOnGpuPath
template <> struct DType<float> { static constexpr int value = SVH_F32; };
template <> struct DType<int32_t> { static constexpr int value = SVH_I32; };
template <> struct DType<uint32_t> { static constexpr int value = SVH_U32; };
template <> struct DType<int16_t> { static constexpr int value = SVH_I16; };  // (cost volumes handed to sgmCostVolume: T_CV)
template <> struct DType<uint16_t> { static constexpr int value = SVH_U16; };
// sgmCostVolume<.., T_CV>: the reference casts every cost it reads to float (sgm.h:234, :273, :299), so the integer types go through;
// double would make it subtract in double (sgm.h:299) and is not taken
template <class T> inline constexpr bool sgmVolumeTypeOnGpuPath =
    std::is_same_v<T, float> || std::is_same_v<T, uint8_t> || std::is_same_v<T, int16_t> || std::is_same_v<T, uint16_t> || std::is_same_v<T, int32_t> ||
    std::is_same_v<T, uint32_t>;

// ---- raw access to a Multidim::Array -------------------------------------------------------------------------------------------
// The reference never asks an array for a data() pointer.  Where its own code needs the address of the elements it takes
// &array.atUnchecked(0, ...) of a non-const NonConstView array (io/image_io.h:96, :160; on_demand_cost_volume.h:98), so that is the
// only route used here.  A ConstView array only hands out values (value<Nc>(idx), valueUnchecked(...)): if those are references
// the first one's address serves, otherwise the array is copied element by element -- the loop of io/image_io.h:56-67 -- into a
// dense private array that the calling thread keeps until eight more such copies have been made (a shim passes far fewer arrays to
// the C calls it makes, so the copy outlives them; the price of this fallback is memory held a little longer).
namespace detail {
template <class A, std::size_t... I> auto first_value(A const &a, std::index_sequence<I...>) -> decltype(a.valueUnchecked((static_cast<int>(I) * 0)...)) {
    return a.valueUnchecked((static_cast<int>(I) * 0)...);
}
template <class A, std::size_t... I> auto &first_element(A &a, std::index_sequence<I...>) { return a.atUnchecked((static_cast<int>(I) * 0)...); }
inline std::vector<std::shared_ptr<void>> &keep_alive() {
    static thread_local std::vector<std::shared_ptr<void>> v;
    return v;
}
} // namespace detail

// address of element (0, ..., 0) of an array that may be written through (nullptr when empty)
template <class T, int N> inline T *firstElement(Multidim::Array<T, N, Multidim::NonConstView> const &a) {
    if (a.empty()) return nullptr;
    return &detail::first_element(const_cast<Multidim::Array<T, N, Multidim::NonConstView> &>(a), std::make_index_sequence<N>{});
}

// dense (last index fastest) copy through the element accessors alone
template <class T, int N, Multidim::ArrayDataAccessConstness C> inline std::shared_ptr<std::vector<std::remove_const_t<T>>> denseCopy(Multidim::Array<T, N, C> const &a) {
    constexpr Multidim::AccessCheck Nc = Multidim::AccessCheck::Nocheck;
    auto out = std::make_shared<std::vector<std::remove_const_t<T>>>(a.flatLenght());
    std::array<int, N> idx;
    idx.fill(0);
    for (std::size_t e = 0; e < out->size(); e++) {
        (*out)[e] = a.template value<Nc>(idx);
        for (int k = N - 1; k >= 0; k--) {
            if (++idx[k] < a.shape()[k]) break;
            idx[k] = 0;
        }
    }
    return out;
}

template <class T, int N, Multidim::ArrayDataAccessConstness C> inline svh_array describe(Multidim::Array<T, N, C> const &a) {
    static_assert(N <= SVH_MAX_DIMS, "too many dimensions");
    svh_array d{};
    d.ndim = N;
    d.dtype = DType<std::remove_const_t<T>>::value;
    d.memspace = SVH_HOST;
    for (int k = 0; k < N; k++) {
        d.shape[k] = a.shape()[k];
        d.strides[k] = a.strides()[k];
    }
    if (a.empty()) return d;
    if constexpr (C == Multidim::NonConstView) {
        d.data = firstElement(a);
    } else if constexpr (std::is_lvalue_reference_v<decltype(detail::first_value(a, std::make_index_sequence<N>{}))>) {
        d.data = const_cast<std::remove_const_t<T> *>(&detail::first_value(a, std::make_index_sequence<N>{}));
    } else {
        auto copy = denseCopy(a);
        auto &kept = detail::keep_alive();
        if (kept.size() >= 8) kept.erase(kept.begin());
        kept.push_back(copy);
        d.data = copy->data();
        int64_t stride = 1;
        for (int k = N - 1; k >= 0; k--) {
            d.strides[k] = stride;
            stride *= a.shape()[k];
        }
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
// transfer back.  The memory is freed with the last copy of the handle, on its device, whatever has become of the context and the thread that allocated it.
template <class T, int N> class DeviceArray;
template <class T, int N> inline svh_array describe(DeviceArray<T, N> const &a);

template <class T, int N> class DeviceArray {
  public:
    using ShapeBlock = std::array<int, N>;
    DeviceArray() { _shape.fill(0); }
    explicit DeviceArray(ShapeBlock const &shape) : _shape(shape) {
        if (flatLenght() == 0) return;
        svh_context *ctx = context();
        void *p = nullptr;
        if (!check(svh_device_alloc(ctx, flatLenght() * sizeof(T), &p))) return;
        // freed by DEVICE, not through the allocating context: the array may outlive the thread (and with it the thread's context) that
        // made it -- a worker returns its result and exits
        const int device = svh_context_get_device(ctx);
        _st = std::make_shared<Storage>();
        _st->mem.reset(static_cast<T *>(p), [device](T *q) { svh_device_free_detached(device, q); });
    }
    template <Multidim::ArrayDataAccessConstness C> static DeviceArray upload(Multidim::Array<T, N, C> const &host) {
        ShapeBlock s;
        for (int k = 0; k < N; k++) s[k] = host.shape()[k];
        DeviceArray d(s);
        if (d.empty()) return d;
        // in place when the elements lie densely, last index fastest (strides() says so); else through a dense copy
        bool dense = true;
        int64_t stride = 1;
        for (int k = N - 1; k >= 0; k--) {
            if (host.shape()[k] != 1 && host.strides()[k] != stride) dense = false;
            stride *= host.shape()[k];
        }
        const T *src = nullptr;
        std::shared_ptr<std::vector<std::remove_const_t<T>>> copy;
        if constexpr (C == Multidim::NonConstView) {
            if (dense) src = firstElement(host);
        }
        if (!src) {
            copy = denseCopy(host);
            src = copy->data();
        }
        check(svh_device_upload(context(), d._st->mem.get(), src, d.flatLenght() * sizeof(T)));
        return d;
    }
    Multidim::Array<T, N> download() const {
        typename Multidim::Array<T, N>::ShapeBlock s;
        for (int k = 0; k < N; k++) s[k] = _shape[k];
        auto host = makeResult<Multidim::Array<T, N>>(s);
        if (empty()) return host;
        // Array(shape) lays its elements out as it sees fit: straight into it when that is dense last-index-fastest, else scattered
        bool dense = true;
        int64_t stride = 1;
        for (int k = N - 1; k >= 0; k--) {
            if (host.shape()[k] != 1 && host.strides()[k] != stride) dense = false;
            stride *= host.shape()[k];
        }
        if (dense) {
            check(svh_device_download(context(), firstElement(host), _st->mem.get(), flatLenght() * sizeof(T)));
        } else {
            constexpr Multidim::AccessCheck Nc = Multidim::AccessCheck::Nocheck;
            std::vector<T> tmp(flatLenght());
            check(svh_device_download(context(), tmp.data(), _st->mem.get(), flatLenght() * sizeof(T)));
            std::array<int, N> idx;
            idx.fill(0);
            for (std::size_t e = 0; e < tmp.size(); e++) {
                host.template at<Nc>(idx) = tmp[e];
                for (int k = N - 1; k >= 0; k--) {
                    if (++idx[k] < _shape[k]) break;
                    idx[k] = 0;
                }
            }
        }
        return host;
    }
    ShapeBlock const &shape() const { return _shape; }
    bool empty() const { return !_st || flatLenght() == 0; }
    // What the library knows about a volume it wrote itself and that has not been written to since (svh_unfold_cost_volume_minima /
    // svh_sgm_cost_volume_minima in include/stevi_hip.h): the regional minima of a census / Hamming cost volume and the bound on its
    // integer costs.  Attached by unfoldBasedCostVolumeOnDevice, read by sgmCostVolume<.., Cost>.  The statement is about the MEMORY, so
    // it lives with the memory (the control block every copy of the handle shares), not with one handle: the moment anybody asks any
    // handle for a pointer the volume could be written through (data() on a non-const array) it is gone for all of them.  const access
    // cannot invalidate it: data() const and share() hand out pointers to const.  A caller who casts that const away, or who kept a
    // mutable pointer from before the statement was attached, is outside what the type can see -- dropStatement() is for them.
    struct Statement {
        std::shared_ptr<const float> minima; // device, (H, W, 2); null when the statement says nothing about the minima
        int minima_kind = 1;                 // 1: integer costs with |c| <= max_abs; 2: float costs, finite magnitudes <= max_abs <= 1e30
        float max_abs = 0;
        // An aggregated volume sgmCostVolume wrote (svh_sgm_cost_volume_winner): the index extractSelectedIndex<winner_strategy> returns
        // for it, computed by the kernel that wrote the volume; read by extractSelectedIndex instead of the volume.  device, (H, W)
        std::shared_ptr<const int32_t> winner;
        int winner_strategy = -1;
    };
    std::shared_ptr<const Statement> statement() const { return _st ? std::atomic_load(&_st->statement) : std::shared_ptr<const Statement>(); }
    void attach(std::shared_ptr<const Statement> st) {
        if (_st) std::atomic_store(&_st->statement, std::move(st));
    }
    void dropStatement() const {
        if (_st) std::atomic_store(&_st->statement, std::shared_ptr<const Statement>());
    }
    // the memory itself, read-only, kept alive by whoever holds the pointer
    std::shared_ptr<const T> share() const { return _st ? std::shared_ptr<const T>(_st, _st->mem.get()) : std::shared_ptr<const T>(); }
    std::size_t flatLenght() const { // (sic) as Multidim::Array
        std::size_t n = 1;
        for (int k = 0; k < N; k++) n *= static_cast<std::size_t>(_shape[k] > 0 ? _shape[k] : 0);
        return n;
    }
    T const *data() const { return _st ? _st->mem.get() : nullptr; }
    T *data() {
        dropStatement(); // the caller may write: whatever was known about the contents is no longer, through any handle
        return _st ? _st->mem.get() : nullptr;
    }

  private:
    template <class U, int M> friend svh_array describe(DeviceArray<U, M> const &a);
    struct Storage {
        std::shared_ptr<T> mem;
        std::shared_ptr<const Statement> statement;
    };
    ShapeBlock _shape;
    std::shared_ptr<Storage> _st;
};

template <class T, int N> inline svh_array describe(DeviceArray<T, N> const &a) {
    static_assert(N <= SVH_MAX_DIMS, "too many dimensions");
    svh_array d{};
    d.data = a._st ? a._st->mem.get() : nullptr; // (the shims write results through descriptors of arrays they have just created: nothing is attached yet)
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

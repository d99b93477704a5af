// Marshalling between Multidim::Array and the C ABI of libstevi_hip.so (include/stevi_hip.h).
// Not a reference header: it is the glue the drop-in headers in this directory share.
#pragma once

#include <MultidimArrays/MultidimArrays.h>

#include <cstdint>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "../../../include/stevi_hip.h"

namespace StereoVision {
namespace Correlation {
namespace HipBridge {

// One context per process, created on first use on the current HIP device (null stream).  The reference's
// functions are stateless; the context only caches device workspace.  Calls are serialised by a mutex.
inline svh_context *context() {
    static svh_context *ctx = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        int st = svh_context_create(&ctx, -1, nullptr);
        if (st != SVH_OK) throw std::runtime_error(std::string("libstevi_hip: ") + svh_status_string(st) + " (there is no CPU fallback)");
    });
    return ctx;
}

inline std::mutex &lock() {
    static std::mutex m;
    return m;
}

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

// Image element types per matching function: float always; uint8 where the reference keeps the plain samples as features
// and casts to float inside the comparison (CENSUS, HAMMING, CC, SSD, SAD: matching_costs.h:749-783), so that widening on the
// device is exact.  uint8 with a normalised or zero-mean function takes the reference's int16 path, which its own comparison
// test skips (testCorrelationFilters.cpp:1249).
template <class T_I, bool ZeroMean, bool Normalized> constexpr bool imageTypeOnGpuPath() {
    using T = std::remove_const_t<T_I>;
    return std::is_same_v<T, float> || (std::is_same_v<T, uint8_t> && !ZeroMean && !Normalized);
}

// SVH_OK -> true, SVH_EMPTY_RESULT -> false (caller returns the empty array the reference returns), else throw
inline bool check(int status) {
    if (status == SVH_OK) return true;
    if (status == SVH_EMPTY_RESULT) return false;
    std::string msg = svh_last_error(context());
    throw std::runtime_error("libstevi_hip: " + (msg.empty() ? std::string(svh_status_string(status)) : msg));
}

} // namespace HipBridge
} // namespace Correlation
} // namespace StereoVision

// Context, workspace pool, profiling and array marshalling for libstevi_hip.so.
#include <cstdarg>
#include <cstdlib>
#include <map>
#include <mutex>
#include <set>

#include "svh_internal.h"
#include "../../include/stevi_hip_test.h"

namespace svh {

int fail(svh_context *ctx, int status, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->last_error = buf;
    return status;
}

// ---- pool ------------------------------------------------------------------------------------------
Scratch::~Scratch() {
    for (size_t k : taken) ctx->pool[k].in_use = false;
}

static void drop_free_pool_blocks(svh_context *ctx) {
    for (auto &b : ctx->pool)
        if (!b.in_use && b.ptr) {
            (void)hipFree(b.ptr);
            b.ptr = nullptr;
            b.bytes = 0;
        }
}

void *Scratch::get(size_t bytes) {
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~size_t(255);
    // best fit among free blocks
    size_t best = SIZE_MAX;
    for (size_t k = 0; k < ctx->pool.size(); k++) {
        const PoolBlock &b = ctx->pool[k];
        if (!b.in_use && b.bytes >= bytes && (best == SIZE_MAX || b.bytes < ctx->pool[best].bytes)) best = k;
    }
    if (best != SIZE_MAX && ctx->pool[best].bytes <= 2 * bytes + (1 << 20)) {
        ctx->pool[best].in_use = true;
        taken.push_back(best);
        return ctx->pool[best].ptr;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        // drop every cached free block -- this context's workspace and the device's cache of released svh_device_alloc blocks -- and retry once
        (void)hipGetLastError();
        drop_free_pool_blocks(ctx);
        device_cache_release_all(ctx->device);
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            fail(ctx, SVH_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
            return nullptr;
        }
    }
    ctx->pool.push_back({p, bytes, true});
    taken.push_back(ctx->pool.size() - 1);
    return p;
}

// ---- profiling -------------------------------------------------------------------------------------
static hipEvent_t take_event(svh_context *ctx) {
    if (!ctx->prof_free_events.empty()) {
        hipEvent_t e = ctx->prof_free_events.back();
        ctx->prof_free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(svh_context *c, const char *n) : ctx(c), name(n) {
    if (!ctx->profiling || (!ctx->prof_filter.empty() && ctx->prof_filter != n)) return;
    if (ctx->prof_every > 1 && (ctx->prof_seen++ % ctx->prof_every) != 0) return;
    start = take_event(ctx);
    stop = take_event(ctx);
    if (!start || !stop) return;
    active = true;
    (void)hipEventRecord(start, ctx->stream);
}

ProfScope::~ProfScope() {
    if (!active) return;
    (void)hipEventRecord(stop, ctx->stream);
    ctx->prof_pending.push_back({name, start, stop});
}

// ---- arrays ----------------------------------------------------------------------------------------
size_t dtype_size(int dtype) {
    switch (dtype) {
    case SVH_F32: case SVH_I32: case SVH_U32: return 4;
    case SVH_U8: return 1;
    case SVH_I16: case SVH_U16: return 2;
    case SVH_U64: return 8;
    default: return 0;
    }
}

int64_t num_elements(const svh_array &a) {
    int64_t n = 1;
    for (int k = 0; k < a.ndim; k++) n *= a.shape[k];
    return n;
}

bool is_dense(const svh_array &a) {
    int64_t expect = 1;
    for (int k = a.ndim - 1; k >= 0; k--) {
        if (a.shape[k] != 1 && a.strides[k] != expect) return false;
        expect *= a.shape[k];
    }
    return true;
}

int validate(svh_context *ctx, const svh_array *a, const char *what, int dtype, int ndim_min, int ndim_max) {
    if (!a) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "%s: null array descriptor", what);
    if (a->ndim < ndim_min || a->ndim > ndim_max || a->ndim > SVH_MAX_DIMS)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "%s: ndim %d not in [%d,%d]", what, a->ndim, ndim_min, ndim_max);
    if (a->dtype != dtype) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "%s: dtype %d, expected %d", what, a->dtype, dtype);
    if (a->memspace != SVH_HOST && a->memspace != SVH_DEVICE)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "%s: bad memspace %d", what, a->memspace);
    for (int k = 0; k < a->ndim; k++) {
        if (a->shape[k] < 0) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "%s: negative extent", what);
        if (a->shape[k] > 1 && a->strides[k] <= 0)
            return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "%s: strides must be positive", what);
    }
    if (num_elements(*a) > 0 && !a->data) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "%s: null data pointer", what);
    return SVH_OK;
}

int validate_volume(svh_context *ctx, const svh_array *cv, const char *what) {
    if (cv && (cv->dtype == SVH_U8 || cv->dtype == SVH_I16 || cv->dtype == SVH_U16 || cv->dtype == SVH_I32 || cv->dtype == SVH_U32))
        return validate(ctx, cv, what, cv->dtype, 3, 3);
    return validate(ctx, cv, what, SVH_F32, 3, 3);
}

bool any_host(std::initializer_list<const svh_array *> arrays) {
    for (const svh_array *a : arrays)
        if (a && a->memspace == SVH_HOST) return true;
    return false;
}

struct Layout4 {
    int64_t shape[4];
    int64_t sstr[4];
    int64_t dstr[4];
};

// element-size-generic strided copy; one thread per element, dense index order on the destination side
template <class T>
__global__ void relayout_kernel(const T *__restrict__ src, T *__restrict__ dst, Layout4 L, int64_t n) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = e;
        int64_t i3 = r % L.shape[3];
        r /= L.shape[3];
        int64_t i2 = r % L.shape[2];
        r /= L.shape[2];
        int64_t i1 = r % L.shape[1];
        int64_t i0 = r / L.shape[1];
        dst[i0 * L.dstr[0] + i1 * L.dstr[1] + i2 * L.dstr[2] + i3 * L.dstr[3]] =
            src[i0 * L.sstr[0] + i1 * L.sstr[1] + i2 * L.sstr[2] + i3 * L.sstr[3]];
    }
}

static void dense_strides(const svh_array &a, int64_t out[4]) {
    int64_t s = 1;
    for (int k = 3; k >= 0; k--) {
        int kk = k - (4 - a.ndim);
        out[k] = s;
        if (kk >= 0) s *= a.shape[kk];
    }
}

static Layout4 make_layout(const svh_array &a, bool user_is_src) {
    Layout4 L;
    int64_t dense[4];
    dense_strides(a, dense);
    for (int k = 0; k < 4; k++) {
        int kk = k - (4 - a.ndim);
        L.shape[k] = kk >= 0 ? a.shape[kk] : 1;
        int64_t us = kk >= 0 ? a.strides[kk] : 0;
        L.sstr[k] = user_is_src ? us : dense[k];
        L.dstr[k] = user_is_src ? dense[k] : us;
    }
    return L;
}

static int launch_relayout(svh_context *ctx, const void *src, void *dst, const Layout4 &L, int64_t n, size_t esize) {
    if (n == 0) return SVH_OK;
    int grid = grid_for(n, 256, 8192);
    if (esize == 4)
        SVH_LAUNCH(ctx, "relayout", relayout_kernel<uint32_t>, grid, 256, 0, (const uint32_t *)src, (uint32_t *)dst, L, n);
    else if (esize == 8)
        SVH_LAUNCH(ctx, "relayout", relayout_kernel<uint64_t>, grid, 256, 0, (const uint64_t *)src, (uint64_t *)dst, L, n);
    else
        SVH_LAUNCH(ctx, "relayout", relayout_kernel<uint8_t>, grid, 256, 0, (const uint8_t *)src, (uint8_t *)dst, L, n);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

template <class T> static void host_strided_copy(const T *src, T *dst, const Layout4 &L) {
    for (int64_t i0 = 0; i0 < L.shape[0]; i0++)
        for (int64_t i1 = 0; i1 < L.shape[1]; i1++)
            for (int64_t i2 = 0; i2 < L.shape[2]; i2++)
                for (int64_t i3 = 0; i3 < L.shape[3]; i3++)
                    dst[i0 * L.dstr[0] + i1 * L.dstr[1] + i2 * L.dstr[2] + i3 * L.dstr[3]] =
                        src[i0 * L.sstr[0] + i1 * L.sstr[1] + i2 * L.sstr[2] + i3 * L.sstr[3]];
}

static void host_copy(const void *src, void *dst, const Layout4 &L, size_t esize) {
    if (esize == 4) host_strided_copy((const uint32_t *)src, (uint32_t *)dst, L);
    else if (esize == 8) host_strided_copy((const uint64_t *)src, (uint64_t *)dst, L);
    else host_strided_copy((const uint8_t *)src, (uint8_t *)dst, L);
}

int stage_in(svh_context *ctx, Scratch &scr, const svh_array &a, void **dptr) {
    int64_t n = num_elements(a);
    size_t es = dtype_size(a.dtype);
    bool dense = is_dense(a);
    if (a.memspace == SVH_DEVICE && dense) {
        *dptr = a.data;
        return SVH_OK;
    }
    void *buf = scr.get((size_t)n * es);
    if (!buf) return SVH_ERR_OUT_OF_MEMORY;
    *dptr = buf;
    if (n == 0) return SVH_OK;
    if (a.memspace == SVH_DEVICE) return launch_relayout(ctx, a.data, buf, make_layout(a, true), n, es);
    if (dense) return copy_h2d(ctx, buf, a.data, (size_t)n * es); // (returns when the caller may reuse the source)
    std::vector<uint8_t> packed((size_t)n * es);
    host_copy(a.data, packed.data(), make_layout(a, true), es);
    return copy_h2d(ctx, buf, packed.data(), (size_t)n * es);
}

int validate_image(svh_context *ctx, const svh_array *img, const char *what, int match_func) {
    if (img && img->dtype == SVH_U8) {
        SVH_TRY(validate(ctx, img, what, SVH_U8, 2, 3));
        if (match_func >= 0 && (func_zero_mean(match_func) || func_normalized(match_func)))
            return fail(ctx, SVH_ERR_UNSUPPORTED,
                        "%s: uint8 images with a normalised or zero-mean matching function follow the reference's int16 path, which its own "
                        "tests skip; convert the images to float32",
                        what);
        return SVH_OK;
    }
    return validate(ctx, img, what, SVH_F32, 2, 3);
}

__global__ void widen_u8_kernel(const uint8_t *__restrict__ in, float *__restrict__ out, int64_t n) {
    // four samples per lane and step: one dword in, one 16-byte store out
    const int64_t n4 = n >> 2;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t w = reinterpret_cast<const uint32_t *>(in)[q];
        reinterpret_cast<float4 *>(out)[q] = make_float4((float)(w & 0xFF), (float)((w >> 8) & 0xFF), (float)((w >> 16) & 0xFF), (float)(w >> 24));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[(n4 << 2) + threadIdx.x] = (float)in[(n4 << 2) + threadIdx.x];
}

int stage_image(svh_context *ctx, Scratch &scr, const svh_array &img, void **dptr) {
    if (img.dtype != SVH_U8) return stage_in(ctx, scr, img, dptr);
    void *bytes;
    SVH_TRY(stage_in(ctx, scr, img, &bytes));
    const int64_t n = num_elements(img);
    float *wide = scr.get_n<float>((size_t)n);
    if (!wide) return SVH_ERR_OUT_OF_MEMORY;
    *dptr = wide;
    if (n == 0) return SVH_OK;
    if ((reinterpret_cast<uintptr_t>(bytes) & 3) != 0) { // a device view that starts inside a dword: repack to an aligned buffer
        void *aligned = scr.get((size_t)n);
        if (!aligned) return SVH_ERR_OUT_OF_MEMORY;
        SVH_HIP_CHECK(ctx, hipMemcpyAsync(aligned, bytes, (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
        bytes = aligned;
    }
    SVH_LAUNCH(ctx, "widen_u8", widen_u8_kernel, grid_for((n + 3) / 4, 256, 4096), 256, 0, (const uint8_t *)bytes, wide, n);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

// integer cost volumes (sgmCostVolume's T_CV): one conversion to float on the device, the reference's own static_cast<float>
template <class T> __global__ void to_float_kernel(const T *__restrict__ in, float *__restrict__ out, int64_t n) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) out[q] = static_cast<float>(in[q]);
}

int stage_volume_as_float(svh_context *ctx, Scratch &scr, const svh_array &a, void **dptr) {
    if (a.dtype == SVH_F32) return stage_in(ctx, scr, a, dptr);
    if (a.dtype == SVH_U8) return stage_image(ctx, scr, a, dptr);
    void *raw;
    SVH_TRY(stage_in(ctx, scr, a, &raw));
    const int64_t n = num_elements(a);
    float *wide = scr.get_n<float>((size_t)n);
    if (!wide) return SVH_ERR_OUT_OF_MEMORY;
    *dptr = wide;
    if (n == 0) return SVH_OK;
    const int grid = grid_for(n, 256, 8192);
    switch (a.dtype) {
    case SVH_I16: SVH_LAUNCH(ctx, "to_float", to_float_kernel<int16_t>, grid, 256, 0, (const int16_t *)raw, wide, n); break;
    case SVH_U16: SVH_LAUNCH(ctx, "to_float", to_float_kernel<uint16_t>, grid, 256, 0, (const uint16_t *)raw, wide, n); break;
    case SVH_I32: SVH_LAUNCH(ctx, "to_float", to_float_kernel<int32_t>, grid, 256, 0, (const int32_t *)raw, wide, n); break;
    case SVH_U32: SVH_LAUNCH(ctx, "to_float", to_float_kernel<uint32_t>, grid, 256, 0, (const uint32_t *)raw, wide, n); break;
    default: return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "cost volume: element type %d", a.dtype);
    }
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

int stage_out(svh_context *ctx, Scratch &scr, const svh_array &a, OutStage *st) {
    st->dst = &a;
    if (a.memspace == SVH_DEVICE && is_dense(a)) {
        st->dptr = a.data;
        st->direct = true;
        return SVH_OK;
    }
    st->direct = false;
    st->dptr = scr.get((size_t)num_elements(a) * dtype_size(a.dtype));
    if (!st->dptr) return SVH_ERR_OUT_OF_MEMORY;
    (void)ctx;
    return SVH_OK;
}

int finish_out(svh_context *ctx, const OutStage &st) {
    if (st.direct || !st.dst) return SVH_OK;
    const svh_array &a = *st.dst;
    int64_t n = num_elements(a);
    size_t es = dtype_size(a.dtype);
    if (n == 0) return SVH_OK;
    if (a.memspace == SVH_DEVICE) return launch_relayout(ctx, st.dptr, a.data, make_layout(a, false), n, es);
    if (is_dense(a)) return copy_d2h(ctx, a.data, st.dptr, (size_t)n * es);
    std::vector<uint8_t> packed((size_t)n * es);
    SVH_TRY(copy_d2h(ctx, packed.data(), st.dptr, (size_t)n * es));
    host_copy(packed.data(), a.data, make_layout(a, false), es);
    return SVH_OK;
}

} // namespace svh

using namespace svh;

extern "C" {

int svh_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n > 0 ? 1 : 0;
}

int svh_context_create(svh_context **out, int device, void *stream) {
    if (!out) return SVH_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SVH_ERR_NO_DEVICE;
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) return SVH_ERR_HIP;
    }
    if (device >= n) return SVH_ERR_INVALID_ARGUMENT;
    svh_context *ctx = new svh_context(); // (nothing is allocated here: the caller's current device is left alone)
    ctx->device = device;
    ctx->stream = (hipStream_t)stream; // NULL = the device's default (null) stream
    *out = ctx;
    return SVH_OK;
}

int svh_context_destroy(svh_context *ctx) {
    if (!ctx) return SVH_OK;
    DeviceGuard guard(ctx->device); // (holds the device number, not the context)
    (void)hipStreamSynchronize(ctx->stream);
    staging_destroy(ctx);
    for (auto &b : ctx->pool)
        if (b.ptr) (void)hipFree(b.ptr);
    for (auto &p : ctx->prof_pending) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    for (auto e : ctx->prof_free_events) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SVH_OK;
}

int svh_context_set_stream(svh_context *ctx, void *stream) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->stream = (hipStream_t)stream;
    return SVH_OK;
}

// the five options of the product surface (include/stevi_hip.h); everything else is a test switch (include/stevi_hip_test.h)
static int set_public_option(svh_context *ctx, const char *name, int value, bool *known) {
    *known = true;
    if (strcmp(name, "census_float_overflow") == 0) {
        if (value != 0 && value != 1) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "census_float_overflow: 0 (a target word that rounds to 2^32 becomes 0xFFFFFFFF) or 1 (becomes 0)");
        ctx->census_float_overflow = value;
        return SVH_OK;
    }
    if (strcmp(name, "census_winner_shortcut") == 0) {
        ctx->census_winner_shortcut = value != 0;
        return SVH_OK;
    }
    if (strcmp(name, "census_sweep") == 0) {
        if (value != 0 && value != 1 && value != 3) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "census_sweep: 0 (auto), 1 (vector ALU) or 3 (matrix cores, FP4)");
        ctx->census_sweep_mode = (int)value;
        return SVH_OK;
    }
    if (strcmp(name, "sgm_score_fused") == 0) {
        if (value < 0 || value > 2) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "sgm_score_fused: 0 (a launch per pass), 1 (automatic) or 2 (bands of rows)");
        ctx->sgm_score_fused = value;
        return SVH_OK;
    }
    if (strcmp(name, "literal_cost_volumes") == 0) {
        ctx->literal_cost_volumes = value != 0;
        return SVH_OK;
    }
    *known = false;
    return SVH_OK;
}

int svh_context_set_option(svh_context *ctx, const char *name, int value) {
    if (!ctx || !name) return SVH_ERR_INVALID_ARGUMENT;
    bool known = false;
    const int st = set_public_option(ctx, name, value, &known);
    if (known) return st;
    return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "unknown option %s", name);
}

// include/stevi_hip_test.h: both sides of the library's A/Bs, for the parity tests
int svh_test_set_option(svh_context *ctx, const char *name, int value) {
    if (!ctx || !name) return SVH_ERR_INVALID_ARGUMENT;
    if (strcmp(name, "sgm_score_fused") == 0 && value == 3) { // the bands with 16-column strips forced
        ctx->sgm_score_fused = 3;
        return SVH_OK;
    }
    bool known = false;
    const int st = set_public_option(ctx, name, value, &known);
    if (known) return st;
    struct Switch {
        const char *name;
        bool svh_context::*flag;
    };
    static const Switch switches[] = {
        {"census_fast_path", &svh_context::census_fast_path},
        {"census_tiles", &svh_context::census_tiles},
        {"cost_volume_colsum", &svh_context::cost_volume_colsum},
        {"patchmatch_pred_costs", &svh_context::patchmatch_pred_costs},
        {"patchmatch_run_batches", &svh_context::patchmatch_run_batches},
        {"patchmatch_lookback", &svh_context::patchmatch_lookback},
        {"patchmatch_scan_chunks", &svh_context::patchmatch_scan_chunks},
        {"feature_volume_tiled", &svh_context::feature_volume_tiled},
        {"feature_volume_records", &svh_context::feature_volume_records},
        {"extract_index_wide", &svh_context::extract_index_wide},
        {"sgm_score_pad", &svh_context::sgm_score_pad},
        {"fold_2d_offsets", &svh_context::fold_2d_offsets},
        {"cost_reduce_fused", &svh_context::cost_reduce_fused},
        {"sgm_cost_two_minima", &svh_context::sgm_cost_two_minima},
        {"sgm_score_finish_fused", &svh_context::sgm_score_finish_fused},
    };
    for (const Switch &sw : switches)
        if (strcmp(name, sw.name) == 0) {
            ctx->*(sw.flag) = value != 0;
            return SVH_OK;
        }
    if (strcmp(name, "census_sweep_rl") == 0) {
        ctx->census_sweep_rl = value;
        return SVH_OK;
    }
    if (strcmp(name, "guided_shared") == 0) {
        if (value < 0 || value > 3) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "guided_shared: 0 .. 3");
        ctx->guided_shared = value;
        return SVH_OK;
    }
    if (strcmp(name, "patchmatch_search_form") == 0) {
        if (value < 0 || value > 3) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "patchmatch_search_form: 0 .. 3");
        ctx->patchmatch_search_form = value;
        return SVH_OK;
    }
    return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "unknown option %s", name);
}

int svh_context_synchronize(svh_context *ctx) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    SVH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return SVH_OK;
}

int svh_context_trim(svh_context *ctx) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    SVH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<PoolBlock> keep;
    for (auto &b : ctx->pool) {
        if (b.in_use) keep.push_back(b);
        else if (b.ptr) (void)hipFree(b.ptr);
    }
    ctx->pool.swap(keep);
    return svh_device_cache_trim(ctx->device);
}

// ---- memory behind HipBridge::DeviceArray: a per-device cache of released blocks -------------------------------------------------------
// A chain written with the reference's names allocates and releases a volume per call (2.1 GB each at 1080p x 256), and hipMalloc /
// hipFree of that size cost tens of milliseconds each: the C3 chain through the drop-in headers took 128 ms for 1.4 ms of kernels.
// Released blocks are kept (best fit, at most twice the request) and handed out again; hipFree only when the cache passes its cap
// (SVH_DEVICE_CACHE_MB, default a quarter of the device's memory) or on svh_context_trim / svh_device_cache_trim.  A block is only cached after the wait its release
// always did (the device idle, or the context's stream drained), so whoever gets it next may use it on any stream.
namespace {
struct DeviceCache {
    std::mutex m;
    std::multimap<size_t, void *> free_blocks; // size -> block
    std::map<void *, size_t> sizes;            // every live block of svh_device_alloc
    std::set<void *> is_cached;                // the blocks of `sizes` that sit in free_blocks (a second release of one is an error, not a second entry)
    size_t cached = 0;
    size_t cap = SIZE_MAX;                     // SIZE_MAX: not looked up yet
};
// (never destroyed: an array released by a static object's destructor after this library's own statics are gone must still find it)
DeviceCache *const g_device_cache = new DeviceCache[64];
// (lock held, the cache's device current) SVH_DEVICE_CACHE_MB when set, else a quarter of the device's memory: idle blocks must not be what
// makes somebody else's allocation (torch, RCCL, a workspace) fail
size_t device_cache_cap(DeviceCache &c) {
    if (c.cap != SIZE_MAX) return c.cap;
    const char *e = getenv("SVH_DEVICE_CACHE_MB");
    if (e) {
        const long long mb = atoll(e);
        c.cap = (size_t)(mb < 0 ? 0 : mb) << 20;
    } else {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
            (void)hipGetLastError();
            total_b = size_t(64) << 30;
        }
        c.cap = total_b / 4;
    }
    return c.cap;
}
// (lock held) hipFree cached blocks, largest first, until at most `keep` bytes stay cached
void device_cache_shrink(DeviceCache &c, size_t keep) {
    while (c.cached > keep && !c.free_blocks.empty()) {
        auto it = std::prev(c.free_blocks.end());
        (void)hipFree(it->second);
        c.sizes.erase(it->second);
        c.is_cached.erase(it->second);
        c.cached -= it->first;
        c.free_blocks.erase(it);
    }
}
int device_release(int device, void *ptr) { // after the caller's wait
    DeviceCache &c = g_device_cache[device & 63];
    std::lock_guard<std::mutex> lock(c.m);
    auto it = c.sizes.find(ptr);
    if (it == c.sizes.end()) return hipFree(ptr) == hipSuccess ? SVH_OK : SVH_ERR_HIP; // (not from svh_device_alloc)
    if (!c.is_cached.insert(ptr).second) return SVH_ERR_INVALID_ARGUMENT;             // released twice: it must not be handed out twice
    c.free_blocks.emplace(it->second, ptr);
    c.cached += it->second;
    device_cache_shrink(c, device_cache_cap(c));
    return SVH_OK;
}
} // namespace

extern "C++" {
namespace svh {
void device_cache_release_all(int device) { // (the device is current: called from an allocator's out-of-memory path)
    DeviceCache &c = g_device_cache[device & 63];
    std::lock_guard<std::mutex> lock(c.m);
    device_cache_shrink(c, 0);
}
} // namespace svh
} // extern "C++"

int svh_device_alloc(svh_context *ctx, size_t bytes, void **ptr) {
    if (!ctx || !ptr) return SVH_ERR_INVALID_ARGUMENT;
    *ptr = nullptr;
    DeviceGuard guard(ctx->device);
    const size_t want = ((bytes ? bytes : 16) + 255) & ~(size_t)255;
    DeviceCache &c = g_device_cache[ctx->device & 63];
    std::lock_guard<std::mutex> lock(c.m);
    auto it = c.free_blocks.lower_bound(want);
    if (it != c.free_blocks.end() && it->first <= 2 * want + (1u << 20)) {
        *ptr = it->second;
        c.cached -= it->first;
        c.is_cached.erase(it->second);
        c.free_blocks.erase(it);
        return SVH_OK;
    }
    if (hipMalloc(ptr, want) != hipSuccess) { // out of memory: give the cached blocks and this context's idle workspace back and try once more
        (void)hipGetLastError();
        device_cache_shrink(c, 0);
        drop_free_pool_blocks(ctx);
        const hipError_t e = hipMalloc(ptr, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            *ptr = nullptr;
            return fail(ctx, SVH_ERR_OUT_OF_MEMORY, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
    }
    c.sizes[*ptr] = want;
    return SVH_OK;
}

int svh_context_get_device(const svh_context *ctx) { return ctx ? ctx->device : -1; }

int svh_device_free_detached(int device, void *ptr) {
    if (!ptr) return SVH_OK;
    if (device < 0) return SVH_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(device);
    if (hipDeviceSynchronize() != hipSuccess) return SVH_ERR_HIP; // any stream of any context may still be using it
    return device_release(device, ptr);
}

int svh_device_free(svh_context *ctx, void *ptr) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    if (!ptr) return SVH_OK;
    DeviceGuard guard(ctx->device);
    SVH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); // nothing enqueued on this context still uses it
    // (another context's stream may: the block is handed out again to any stream, so wait for the device like the detached release)
    SVH_HIP_CHECK(ctx, hipDeviceSynchronize());
    return device_release(ctx->device, ptr);
}

int svh_device_cache_trim(int device) {
    if (device < 0) return SVH_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(device);
    DeviceCache &c = g_device_cache[device & 63];
    std::lock_guard<std::mutex> lock(c.m);
    device_cache_shrink(c, 0);
    return SVH_OK;
}

int svh_device_upload(svh_context *ctx, void *device_dst, const void *host_src, size_t bytes) {
    if (!ctx || (bytes && (!device_dst || !host_src))) return SVH_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    return copy_h2d(ctx, device_dst, host_src, bytes);
}

int svh_device_download(svh_context *ctx, void *host_dst, const void *device_src, size_t bytes) {
    if (!ctx || (bytes && (!host_dst || !device_src))) return SVH_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    return copy_d2h(ctx, host_dst, device_src, bytes);
}

int svh_device_copy(svh_context *ctx, void *device_dst, const void *device_src, size_t bytes) {
    if (!ctx || (bytes && (!device_dst || !device_src))) return SVH_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    SVH_HIP_CHECK(ctx, hipMemcpyAsync(device_dst, device_src, bytes, hipMemcpyDeviceToDevice, ctx->stream)); // (stream-ordered like the kernels: no wait)
    return SVH_OK;
}

const char *svh_status_string(int status) {
    switch (status) {
    case SVH_OK: return "ok";
    case SVH_EMPTY_RESULT: return "empty result (the reference returns an empty array for these arguments)";
    case SVH_ERR_INVALID_ARGUMENT: return "invalid argument";
    case SVH_ERR_UNSUPPORTED: return "unsupported on the GPU path";
    case SVH_ERR_NO_DEVICE: return "no HIP device available";
    case SVH_ERR_HIP: return "HIP runtime error";
    case SVH_ERR_OUT_OF_MEMORY: return "out of device memory";
    default: return "unknown status";
    }
}

const char *svh_last_error(const svh_context *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int svh_profile_enable(svh_context *ctx, int enable) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    ctx->profiling = enable != 0;
    return SVH_OK;
}

int svh_profile_filter(svh_context *ctx, const char *kernel_name) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    ctx->prof_filter = kernel_name ? kernel_name : "";
    return SVH_OK;
}

int svh_profile_sampling(svh_context *ctx, int every) {
    if (!ctx || every < 1) return SVH_ERR_INVALID_ARGUMENT;
    ctx->prof_every = every;
    ctx->prof_seen = 0;
    return SVH_OK;
}

int svh_profile_collect(svh_context *ctx) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    SVH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &p : ctx->prof_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            auto it = ctx->prof_stats.find(p.name);
            if (it == ctx->prof_stats.end()) {
                ctx->prof_order.push_back(p.name);
                it = ctx->prof_stats.emplace(p.name, ProfStat()).first;
            }
            it->second.total_ms += ms;
            it->second.launches += 1;
        }
        ctx->prof_free_events.push_back(p.start);
        ctx->prof_free_events.push_back(p.stop);
    }
    ctx->prof_pending.clear();
    return SVH_OK;
}

int svh_profile_reset(svh_context *ctx) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    int s = svh_profile_collect(ctx);
    ctx->prof_stats.clear();
    ctx->prof_order.clear();
    return s;
}

int svh_profile_count(const svh_context *ctx) { return ctx ? (int)ctx->prof_order.size() : 0; }

int svh_profile_get(const svh_context *ctx, int k, char *name, size_t name_len, double *total_ms, int64_t *launches) {
    if (!ctx || k < 0 || k >= (int)ctx->prof_order.size()) return SVH_ERR_INVALID_ARGUMENT;
    const std::string &n = ctx->prof_order[k];
    const ProfStat &s = ctx->prof_stats.at(n);
    if (name && name_len) {
        strncpy(name, n.c_str(), name_len - 1);
        name[name_len - 1] = 0;
    }
    if (total_ms) *total_ms = s.total_ms;
    if (launches) *launches = s.launches;
    return SVH_OK;
}

} // extern "C"

// Host <-> device transfers of libstevi_hip.so, and the pinned host memory the drop-in headers and the Python mirror put their result
// arrays in.
//
// The reference's callers hand every function host arrays and get host arrays back (benchmarkCrossCorrelationAlgorithms.cpp:92-96,
// :288-294: unfoldBasedCostVolume -> sgmCostVolume -> extractSelectedIndex), so a chain written with its names moves each 2.1 GB volume
// (1080p x 256, float32) over PCIe down and up again: 8.5 GB per chain against 1.4 ms of kernels.  hipMemcpy from pageable memory ran that
// at 7.5 GB/s (1.1 - 1.2 s per chain, rounds 2 - 4).  Two things here:
//   * svh_host_alloc / svh_host_free: page-locked blocks from a process-wide cache.  Result arrays allocated in them (the shims allocate
//     every result: "returns a fresh owning Multidim::Array by value") are written by the DMA engines directly, and read directly when the
//     caller passes them to the next function: one hipMemcpyAsync at the link's rate, nothing staged, nothing faulted in.
//   * pageable memory the caller owns: chunks copied by a few host threads through a page-locked ring while the DMA engines move the
//     previous chunks on two copy streams -- the host's memcpy and the link run concurrently instead of in turns.
// No content is ever cached: every call moves what the caller's array holds at that moment.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <set>
#include <system_error>
#include <thread>

#include "svh_internal.h"

namespace svh {

// ---- page-locked host blocks -------------------------------------------------------------------------------------------------------------
namespace {
struct HostCache {
    std::mutex m;
    std::multimap<size_t, void *> free_blocks; // size -> block (released, kept pinned)
    std::map<const void *, size_t> live;       // every block hipHostMalloc gave us (handed out or cached): base -> bytes
    std::set<const void *> is_cached;
    size_t cached = 0;
};
// (never destroyed: arrays released by static destructors after this library's statics are gone must still find it)
HostCache *const g_host = new HostCache;

size_t host_cache_cap() {
    static const size_t cap = [] {
        const char *e = getenv("SVH_HOST_CACHE_MB");
        const long long mb = e ? atoll(e) : 8192; // two 1080p x 256 float volumes and change
        return (size_t)(mb < 0 ? 0 : mb) << 20;
    }();
    return cap;
}

void host_cache_shrink(HostCache &c, size_t keep) { // (lock held) largest first
    while (c.cached > keep && !c.free_blocks.empty()) {
        auto it = std::prev(c.free_blocks.end());
        (void)hipHostFree(it->second);
        c.live.erase(it->second);
        c.is_cached.erase(it->second);
        c.cached -= it->first;
        c.free_blocks.erase(it);
    }
}
} // namespace

// does [p, p + bytes) lie in page-locked memory the DMA engines can address directly?
static bool host_range_is_pinned(const void *p, size_t bytes) {
    {
        HostCache &c = *g_host;
        std::lock_guard<std::mutex> lock(c.m);
        auto it = c.live.upper_bound(p);
        if (it != c.live.begin()) {
            --it;
            const char *base = static_cast<const char *>(it->first);
            if (static_cast<const char *>(p) >= base && static_cast<const char *>(p) + bytes <= base + it->second) return true;
        }
    }
    // somebody else's page-locked memory (hipHostMalloc / hipHostRegister / a framework's pinned tensor)
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError(); // plain malloc'd memory: not an error of ours
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// ---- the staging ring of a context ---------------------------------------------------------------------------------------------------------
struct Staging {
    static constexpr int kMaxThreads = 8;
    static constexpr size_t kSlotBytes = size_t(4) << 20;
    int threads = 4;
    char *ring = nullptr;                    // 2 * threads slots
    hipEvent_t ev[2 * kMaxThreads] = {};     // "the DMA that reads / writes this slot is done"
    bool pending[2 * kMaxThreads] = {};      // an H2D out of the slot may still be in flight (its event says when it is not)
    hipStream_t copy[2] = {nullptr, nullptr};
    hipEvent_t fence = nullptr, done[2] = {nullptr, nullptr};
};

static int staging_get(svh_context *ctx, Staging **out) {
    if (ctx->staging) {
        *out = ctx->staging;
        return SVH_OK;
    }
    Staging *s = new Staging();
    const char *e = getenv("SVH_COPY_THREADS");
    s->threads = std::max(1, std::min(Staging::kMaxThreads, e ? atoi(e) : 4));
    ctx->staging = s; // (from here on svh_context_destroy releases whatever got created)
    SVH_HIP_CHECK(ctx, hipHostMalloc((void **)&s->ring, 2 * s->threads * Staging::kSlotBytes, hipHostMallocDefault));
    for (int k = 0; k < 2 * s->threads; k++) SVH_HIP_CHECK(ctx, hipEventCreateWithFlags(&s->ev[k], hipEventDisableTiming));
    for (int k = 0; k < 2; k++) {
        SVH_HIP_CHECK(ctx, hipStreamCreateWithFlags(&s->copy[k], hipStreamNonBlocking));
        SVH_HIP_CHECK(ctx, hipEventCreateWithFlags(&s->done[k], hipEventDisableTiming));
    }
    SVH_HIP_CHECK(ctx, hipEventCreateWithFlags(&s->fence, hipEventDisableTiming));
    *out = s;
    return SVH_OK;
}

void staging_destroy(svh_context *ctx) {
    Staging *s = ctx->staging;
    if (!s) return;
    for (int k = 0; k < 2; k++)
        if (s->copy[k]) {
            (void)hipStreamSynchronize(s->copy[k]);
            (void)hipStreamDestroy(s->copy[k]);
        }
    for (auto &e : s->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : s->done)
        if (e) (void)hipEventDestroy(e);
    if (s->fence) (void)hipEventDestroy(s->fence);
    if (s->ring) (void)hipHostFree(s->ring);
    delete s;
    ctx->staging = nullptr;
}

static constexpr size_t kPipelineFrom = size_t(8) << 20; // smaller pageable copies: one hipMemcpyAsync (the runtime's own staging)

// one copy on the context's stream, waited for (page-locked memory: the DMA engines directly; small pageable arrays: the runtime's own staging)
static int plain_copy(svh_context *ctx, void *device, void *host, size_t bytes, bool h2d) {
    SVH_HIP_CHECK(ctx, h2d ? hipMemcpyAsync(device, host, bytes, hipMemcpyHostToDevice, ctx->stream) : hipMemcpyAsync(host, device, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SVH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); // (H2D: the source may be reused or freed by the caller right after we return)
    return SVH_OK;
}

// pageable host memory <-> device through the ring; h2d: `host` is the source
static int pipelined_copy(svh_context *ctx, void *device, void *host, size_t bytes, bool h2d) {
    Staging *s = nullptr;
    if (staging_get(ctx, &s) != SVH_OK) { // no page-locked memory for the ring: the runtime's own staging then
        (void)hipGetLastError();
        staging_destroy(ctx);
        return plain_copy(ctx, device, host, bytes, h2d);
    }
    const int T = s->threads;
    const size_t nchunks = (bytes + Staging::kSlotBytes - 1) / Staging::kSlotBytes;
    // the copy streams start after everything the context's stream holds: the kernels that wrote `device` (D2H), or the last readers of the
    // workspace block it is (H2D)
    SVH_HIP_CHECK(ctx, hipEventRecord(s->fence, ctx->stream));
    for (int k = 0; k < 2; k++) SVH_HIP_CHECK(ctx, hipStreamWaitEvent(s->copy[k], s->fence, 0));
    std::atomic<int> err{(int)hipSuccess};
    const int device_index = ctx->device;
    auto worker = [&](int t) {
        auto ok = [&](hipError_t e) {
            if (e != hipSuccess) {
                int expected = (int)hipSuccess;
                err.compare_exchange_strong(expected, (int)e);
            }
            return e == hipSuccess;
        };
        if (!ok(hipSetDevice(device_index))) return;
        hipStream_t st = s->copy[t & 1];
        auto slot_of = [&](size_t c) { return t + T * (int)((c / T) & 1); };
        auto span = [&](size_t c, size_t *off, size_t *len) {
            *off = c * Staging::kSlotBytes;
            *len = std::min(Staging::kSlotBytes, bytes - *off);
        };
        auto slot_free = [&](int slot) {
            if (!s->pending[slot]) return true;
            s->pending[slot] = false;
            return ok(hipEventSynchronize(s->ev[slot]));
        };
        if (h2d) {
            for (size_t c = t; c < nchunks && err.load() == (int)hipSuccess; c += T) {
                const int slot = slot_of(c);
                size_t off, len;
                span(c, &off, &len);
                if (!slot_free(slot)) return;
                char *sp = s->ring + slot * Staging::kSlotBytes;
                memcpy(sp, static_cast<const char *>(host) + off, len);
                if (!ok(hipMemcpyAsync(static_cast<char *>(device) + off, sp, len, hipMemcpyHostToDevice, st))) return;
                if (!ok(hipEventRecord(s->ev[slot], st))) return;
                s->pending[slot] = true;
            }
        } else {
            auto issue = [&](size_t c) {
                const int slot = slot_of(c);
                size_t off, len;
                span(c, &off, &len);
                if (!slot_free(slot)) return false;
                return ok(hipMemcpyAsync(s->ring + slot * Staging::kSlotBytes, static_cast<const char *>(device) + off, len, hipMemcpyDeviceToHost, st)) &&
                       ok(hipEventRecord(s->ev[slot], st));
            };
            if ((size_t)t < nchunks && !issue(t)) return;
            for (size_t c = t; c < nchunks; c += T) {
                if (c + T < nchunks && !issue(c + T)) return; // the next chunk crosses the link while this one is copied out of its slot
                const int slot = slot_of(c);
                size_t off, len;
                span(c, &off, &len);
                if (!ok(hipEventSynchronize(s->ev[slot]))) return;
                memcpy(static_cast<char *>(host) + off, s->ring + slot * Staging::kSlotBytes, len);
            }
        }
    };
    std::thread pool[Staging::kMaxThreads];
    const int used = (int)std::min<size_t>(T, nchunks);
    int started = 1; // (worker 0 is this thread)
    for (int t = 1; t < used; t++) {
        try {
            pool[t] = std::thread(worker, t);
            started = t + 1;
        } catch (const std::system_error &) { // no more threads to be had: the remaining chunk sets run on this thread, after its own
            break;
        }
    }
    worker(0);
    for (int t = started; t < used; t++) worker(t);
    for (int t = 1; t < started; t++) pool[t].join();
    if (err.load() != (int)hipSuccess) {
        for (int k = 0; k < 2; k++) (void)hipStreamSynchronize(s->copy[k]);
        for (auto &p : s->pending) p = false;
        return fail(ctx, err.load() == (int)hipErrorOutOfMemory ? SVH_ERR_OUT_OF_MEMORY : SVH_ERR_HIP, "staged %s copy of %zu bytes failed: %s", h2d ? "host-to-device" : "device-to-host",
                    bytes, hipGetErrorString((hipError_t)err.load()));
    }
    if (h2d) { // later work on the context's stream sees the data; the caller's memory has been read already
        for (int k = 0; k < 2; k++) {
            SVH_HIP_CHECK(ctx, hipEventRecord(s->done[k], s->copy[k]));
            SVH_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, s->done[k], 0));
        }
    }
    return SVH_OK;
}

// Returns when the caller may reuse `src`; the data is in `dst` for everything enqueued on the context's stream afterwards.
int copy_h2d(svh_context *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return SVH_OK;
    if (bytes >= kPipelineFrom && !host_range_is_pinned(src, bytes)) return pipelined_copy(ctx, dst, const_cast<void *>(src), bytes, true);
    return plain_copy(ctx, dst, const_cast<void *>(src), bytes, true);
}

// Returns when `dst` holds the data (everything enqueued on the context's stream before the call has run).
int copy_d2h(svh_context *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return SVH_OK;
    if (bytes >= kPipelineFrom && !host_range_is_pinned(dst, bytes)) return pipelined_copy(ctx, const_cast<void *>(src), dst, bytes, false);
    return plain_copy(ctx, const_cast<void *>(src), dst, bytes, false);
}

} // namespace svh

using namespace svh;

extern "C" {

int svh_host_alloc(size_t bytes, void **ptr) {
    if (!ptr) return SVH_ERR_INVALID_ARGUMENT;
    *ptr = nullptr;
    const size_t want = ((bytes ? bytes : 16) + 4095) & ~(size_t)4095;
    HostCache &c = *g_host;
    std::lock_guard<std::mutex> lock(c.m);
    auto it = c.free_blocks.lower_bound(want);
    if (it != c.free_blocks.end() && it->first <= 2 * want + (1u << 20)) {
        *ptr = it->second;
        c.cached -= it->first;
        c.is_cached.erase(it->second);
        c.free_blocks.erase(it);
        return SVH_OK;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return SVH_ERR_NO_DEVICE;
    }
    if (hipHostMalloc(ptr, want, hipHostMallocDefault) != hipSuccess) { // give the cached blocks back and try once more
        (void)hipGetLastError();
        host_cache_shrink(c, 0);
        if (hipHostMalloc(ptr, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            *ptr = nullptr;
            return SVH_ERR_OUT_OF_MEMORY;
        }
    }
    c.live[*ptr] = want;
    return SVH_OK;
}

int svh_host_free(void *ptr) {
    if (!ptr) return SVH_OK;
    HostCache &c = *g_host;
    std::lock_guard<std::mutex> lock(c.m);
    auto it = c.live.find(ptr);
    if (it == c.live.end() || c.is_cached.count(ptr)) return SVH_ERR_INVALID_ARGUMENT; // not from svh_host_alloc, or released twice
    // (a transfer into or out of the block has always completed when the call that made it returned: nothing to wait for)
    c.free_blocks.emplace(it->second, ptr);
    c.is_cached.insert(ptr);
    c.cached += it->second;
    host_cache_shrink(c, host_cache_cap());
    return SVH_OK;
}

int svh_host_is_pinned(const void *ptr, size_t bytes) { return ptr && host_range_is_pinned(ptr, bytes ? bytes : 1) ? 1 : 0; }

int svh_host_cache_trim(void) {
    HostCache &c = *g_host;
    std::lock_guard<std::mutex> lock(c.m);
    host_cache_shrink(c, 0);
    return SVH_OK;
}

} // extern "C"

// The one exchange of the disparity-sharded census + SGM protocol (include/stevi_hip.h, "disparity-sharded census"), for callers
// that are not Python: an int32 MIN all-reduce of the regional winner keys over RCCL, enqueued on the context's stream.
//
// The library does not link librccl: single-GPU users never load it, and a process must run ONE instance of RCCL -- the one its
// communicators come from.  ncclAllReduce is therefore resolved at the first call from the image that is already loaded (a C++ host
// that linked -lrccl to create its communicators; PyTorch's bundled librccl.so.1), falling back to opening librccl.so.1 by soname.
#include <dlfcn.h>

#include <atomic>

#include <rccl/rccl.h>

#include "svh_internal.h"

namespace svh {

namespace {

using AllReduceFn = ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
using ErrorStringFn = const char *(*)(ncclResult_t);

struct Rccl {
    AllReduceFn all_reduce = nullptr;
    ErrorStringFn error_string = nullptr;
};

const Rccl *rccl() {
    static std::atomic<const Rccl *> cached{nullptr};
    const Rccl *r = cached.load(std::memory_order_acquire);
    if (r) return r;
    auto *fresh = new Rccl;
    void *sym = dlsym(RTLD_DEFAULT, "ncclAllReduce"); // the instance the caller's communicator belongs to, when it is in the global scope
    void *lib = nullptr;
    if (!sym) {
        lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD); // loaded, but with local visibility (Python extension modules)
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW);
        if (lib) sym = dlsym(lib, "ncclAllReduce");
    }
    fresh->all_reduce = reinterpret_cast<AllReduceFn>(sym);
    void *es = lib ? dlsym(lib, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString");
    fresh->error_string = reinterpret_cast<ErrorStringFn>(es);
    const Rccl *expected = nullptr;
    if (!cached.compare_exchange_strong(expected, fresh, std::memory_order_acq_rel)) {
        delete fresh; // another thread was first
        return expected;
    }
    return fresh;
}

__global__ void gather_plane0_kernel(const int2 *__restrict__ keys, int64_t n, int *__restrict__ plane) {
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p < n) plane[p] = keys[p].x;
}
__global__ void scatter_plane0_kernel(const int *__restrict__ plane, int64_t n, int2 *__restrict__ keys) {
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p < n) keys[p].x = plane[p];
}

} // namespace

} // namespace svh

using namespace svh;

extern "C" int svh_census_exchange_keys(svh_context *ctx, void *nccl_comm, svh_array *keys, int plane0_only) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    if (!nccl_comm || !keys) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "svh_census_exchange_keys: communicator and keys are required");
    SVH_TRY(validate(ctx, keys, "keys", SVH_I32, 3, 3));
    if (keys->shape[2] != 2) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "keys must have shape (H,W,2)");
    if (keys->memspace != SVH_DEVICE || !is_dense(*keys))
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "svh_census_exchange_keys: the keys must be a dense array in device memory (the collective runs on them in place)");
    const int64_t npx = keys->shape[0] * keys->shape[1];
    if (npx == 0) return SVH_OK;
    const Rccl *r = rccl();
    if (!r->all_reduce) return fail(ctx, SVH_ERR_UNSUPPORTED, "svh_census_exchange_keys: no RCCL in this process (librccl.so.1 could not be opened)");
    Scratch scr(ctx);
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    ncclResult_t st;
    if (plane0_only) { // the second key plane is global as written: 4 bytes per pixel travel instead of 8
        int *plane = scr.get_n<int>((size_t)npx);
        if (!plane) return SVH_ERR_OUT_OF_MEMORY;
        int2 *k2 = static_cast<int2 *>(keys->data);
        SVH_LAUNCH(ctx, "keys_gather_plane0", gather_plane0_kernel, grid_for(npx, 256), 256, 0, k2, npx, plane);
        SVH_CHECK_LAUNCH(ctx);
        {
            ProfScope prof(ctx, "rccl_all_reduce_min");
            st = r->all_reduce(plane, plane, (size_t)npx, ncclInt32, ncclMin, comm, ctx->stream);
        }
        if (st == ncclSuccess) {
            SVH_LAUNCH(ctx, "keys_scatter_plane0", scatter_plane0_kernel, grid_for(npx, 256), 256, 0, plane, npx, k2);
            SVH_CHECK_LAUNCH(ctx);
            // the scratch plane returns to the pool when this call ends: later calls on this context are ordered behind the scatter by the
            // stream, and another context never sees the pool
        }
    } else {
        ProfScope prof(ctx, "rccl_all_reduce_min");
        st = r->all_reduce(keys->data, keys->data, (size_t)npx * 2, ncclInt32, ncclMin, comm, ctx->stream);
    }
    if (st != ncclSuccess) return fail(ctx, SVH_ERR_HIP, "ncclAllReduce failed: %s", r->error_string ? r->error_string(st) : "error");
    return SVH_OK;
}

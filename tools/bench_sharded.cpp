// BASELINE configuration 5 from a C++ host: census 9x9 + SGM-8 at 8192 x 4320, D = 512, the disparity range split over the GPUs of
// one node, ONE process per GPU, the regional winner keys exchanged by an int32 MIN all-reduce over RCCL (xGMI) -- the north star's
// protocol without Python: the drop-in header correlation/sharded.h on top of svh_census_shard_keys / svh_census_exchange_keys /
// svh_census_shard_finish.  This program creates the communicator (it links librccl); the library takes ncclAllReduce from it.
//
//   g++ -std=c++17 -O2 -I libstevi_amd/include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/bench_sharded.cpp -o tools/bench_sharded
//       -L libstevi_amd -lstevi_hip -Wl,-rpath,$PWD/libstevi_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lrccl -lamdhip64 -lpthread
//   for r in 0 1 2 3 4 5 6 7; do tools/bench_sharded --rank $r --world 8 --id-file /tmp/svh_rccl_id & done; wait
//
// Rank 0 prints one JSON line: ms per frame (barrier + device synchronize on both sides, max over ranks by a MAX all-reduce),
// Mdisparities/s of the whole job, and the number of pixels in which the sharded map differs from the map rank 0 computes alone on the
// whole range (must be 0).  With fewer visible GPUs than ranks it prints {"skipped": ...} and exits 0: it never oversubscribes a GPU.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <thread>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <correlation/sharded.h>

namespace SC = StereoVision::Correlation;

#define RCCL_OK(x)                                                                        \
    do {                                                                                  \
        ncclResult_t r_ = (x);                                                            \
        if (r_ != ncclSuccess) {                                                          \
            std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_));                 \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

int main(int argc, char **argv) {
    int rank = 0, world = 1, W = 8192, H = 4320, D = 512, steps = 10;
    bool always = false; // --exchange-always 1: the all-reduce also with one rank (one-GPU rehearsal)
    std::string id_file = "/tmp/svh_rccl_id";
    for (int a = 1; a + 1 < argc; a += 2) {
        const std::string k = argv[a];
        if (k == "--rank") rank = std::atoi(argv[a + 1]);
        else if (k == "--world") world = std::atoi(argv[a + 1]);
        else if (k == "--id-file") id_file = argv[a + 1];
        else if (k == "--width") W = std::atoi(argv[a + 1]);
        else if (k == "--height") H = std::atoi(argv[a + 1]);
        else if (k == "--disparities") D = std::atoi(argv[a + 1]);
        else if (k == "--steps") steps = std::atoi(argv[a + 1]);
        else if (k == "--exchange-always") always = std::atoi(argv[a + 1]) != 0;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < world) {
        if (rank == 0) std::printf("{\"skipped\": \"%d ranks need %d GPUs, %d visible\"}\n", world, world, n_dev);
        return 0;
    }
    if (hipSetDevice(rank) != hipSuccess) return 2;

    // rendezvous: rank 0 writes the unique id to a file, the others wait for it (one node: a shared /tmp)
    ncclUniqueId id;
    if (rank == 0) {
        RCCL_OK(ncclGetUniqueId(&id));
        std::ofstream(id_file + ".tmp", std::ios::binary).write(id.internal, sizeof id.internal);
        std::rename((id_file + ".tmp").c_str(), id_file.c_str());
    } else {
        for (int tries = 0;; tries++) {
            std::ifstream in(id_file, std::ios::binary);
            if (in.read(id.internal, sizeof id.internal)) break;
            if (tries > 600) return 3;
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }
    ncclComm_t nccl = nullptr;
    RCCL_OK(ncclCommInitRank(&nccl, world, id, rank));
    if (rank == 0) std::remove(id_file.c_str());
    SC::ShardCommunicator comm{nccl, rank, world, always};

    // the same synthetic pair on every rank (seeded), resident on the rank's GPU
    Multidim::Array<float, 2> l(H, W), r(H, W);
    {
        std::mt19937 rng(5);
        std::uniform_real_distribution<float> u(-1.f, 1.f);
        float *pl = SC::HipBridge::firstElement(l), *pr = SC::HipBridge::firstElement(r);
        for (size_t e = 0; e < l.flatLenght(); e++) {
            pr[e] = u(rng);
            pl[e] = u(rng);
        }
    }
    auto dl = SC::HipBridge::DeviceArray<float, 2>::upload(l), dr = SC::HipBridge::DeviceArray<float, 2>::upload(r);
    auto frame = [&] { return SC::censusSgmDisparityShardedOnDevice(dl, dr, 4, 4, D, comm); };
    auto barrier = [&]() -> int {
        int *flag = nullptr;
        if (hipMalloc((void **)&flag, sizeof(int)) != hipSuccess) return 2;
        (void)hipMemset(flag, 0, sizeof(int));
        RCCL_OK(ncclAllReduce(flag, flag, 1, ncclInt32, ncclSum, nccl, nullptr));
        (void)hipDeviceSynchronize();
        (void)hipFree(flag);
        return 0;
    };
    auto map = frame(); // warm-up (workspace, RCCL channels)
    map = frame();
    if (barrier()) return 2;
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < steps; k++) map = frame();
    (void)hipDeviceSynchronize();
    float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
    {   // max over ranks
        float *d_ms = nullptr;
        if (hipMalloc((void **)&d_ms, sizeof(float)) != hipSuccess) return 2;
        (void)hipMemcpy(d_ms, &ms, sizeof(float), hipMemcpyHostToDevice);
        RCCL_OK(ncclAllReduce(d_ms, d_ms, 1, ncclFloat, ncclMax, nccl, nullptr));
        (void)hipMemcpy(&ms, d_ms, sizeof(float), hipMemcpyDeviceToHost);
        (void)hipFree(d_ms);
    }
    int rc = 0;
    if (rank == 0) {
        const Multidim::Array<SC::disp_t, 2> got = map.download();
        const Multidim::Array<SC::disp_t, 2> alone = SC::censusSgmDisparitySharded(dl, dr, 4, 4, D, SC::ShardCommunicator{nullptr, 0, 1, false});
        long diff = 0, sum = 0;
        const SC::disp_t *a = SC::HipBridge::firstElement(got), *b = SC::HipBridge::firstElement(alone);
        for (size_t e = 0; e < got.flatLenght(); e++) {
            diff += a[e] != b[e];
            sum += a[e];
        }
        std::printf("{\"workload\": \"%dx%d census 9x9 + SGM-8, D=%d, disparity range split over %d GPUs (one process each), int32 MIN all-reduce over RCCL "
                    "through svh_census_exchange_keys\", \"n_gpus\": %d, \"steps\": %d, \"ms_per_frame\": %.4f, \"Mdisparities_per_s\": %.1f, "
                    "\"pixels_differing_from_one_gpu\": %ld, \"disp_checksum\": %ld}\n",
                    W, H, D, world, world, steps, ms, (double)W * H * D / ms / 1e3, diff, sum);
        rc = diff ? 3 : 0;
    }
    map = SC::HipBridge::DeviceArray<SC::disp_t, 2>();
    if (barrier()) return 2;
    ncclCommDestroy(nccl);
    return rc;
}

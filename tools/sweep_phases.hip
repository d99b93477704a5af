// In-kernel phase timing of the matrix-core census sweep (svh_census_sweep_pm.hip built with SVH_SWEEP_STAMPS): every wave stamps
// s_memtime at kernel entry, when its first item is ready, and around the phases of its SECOND item (steady state): the tiles of
// its three column tiles, the wait for the next item's DMA, the expansion of the next window, the barrier; and at exit.
// A diagnostic build of the SAME kernel source; the product library never contains a stamp.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I libstevi_amd/csrc -I include \
//         tools/sweep_phases.hip -o tools/sweep_phases && tools/sweep_phases [W H D]
//
// Prints one JSON object: per-phase mean / median / p90 in shader cycles and microseconds, the spread of block start times
// (dispatch rounds), the in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz) and the launch's wall time.
#define SVH_SWEEP_STAMPS 1
#include "../libstevi_amd/csrc/svh_census_sweep_pm.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                     \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

__global__ void clock_probe(unsigned long long *out) {
    if (threadIdx.x == 0) {
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long r1 = r0;
        while (r1 - r0 < 2000) r1 = __builtin_amdgcn_s_memrealtime(); // 20 us at 100 MHz
        out[0] = __builtin_amdgcn_s_memtime() - c0;
        out[1] = r1 - r0;
    }
}

static double pct(std::vector<double> &v, double p) {
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[std::min(v.size() - 1, (size_t)(p * v.size()))];
}

int main(int argc, char **argv) {
    using namespace svh;
    const int W = argc > 1 ? atoi(argv[1]) : 1920, H = argc > 2 ? atoi(argv[2]) : 1080, D = argc > 3 ? atoi(argv[3]) : 256;
    constexpr int NW = 2, WAVES = 4, CT = 3, PXB = WAVES * CT * 32;
    std::mt19937 rng(1);
    std::vector<uint32_t> h_sw((size_t)H * W * NW);
    for (auto &x : h_sw) x = rng();
    std::vector<uint32_t> h_tw((size_t)H * W * NW);
    for (auto &x : h_tw) x = rng();
    uint32_t *sw, *tw;
    uint2 *keys;
    float *gmap;
    unsigned long long *stamps, *clk;
    size_t shmem;
    int grid;
    const SweepPlan plan = make_plan(H, W, PXB, NW, WAVES, CT, D, 256, &shmem, &grid);
    CK(hipMalloc(&sw, h_sw.size() * 4 + 256));
    CK(hipMalloc(&tw, h_tw.size() * 4 + 256));
    CK(hipMalloc(&keys, (size_t)H * W * 8));
    CK(hipMalloc(&gmap, (size_t)H * W * 4));
    CK(hipMalloc(&stamps, (size_t)grid * WAVES * 8 * 8));
    CK(hipMalloc(&clk, 16));
    CK(hipMemcpy(sw, h_sw.data(), h_sw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(tw, h_tw.data(), h_tw.size() * 4, hipMemcpyHostToDevice));
    CensusGeom g{sw, tw, NW, H, W, W, D, 1, 0, 0, -1};
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&census_sweep_pm_kernel<NW, WAVES, CT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int it = 0; it < 30; it++) { // warm: clocks, caches; the last launch is the one read back
        CK(hipMemsetAsync(stamps, 0, (size_t)grid * WAVES * 8 * 8, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((census_sweep_pm_kernel<NW, WAVES, CT>), dim3(grid), dim3(64 * WAVES), shmem, 0, g, 100.0f, keys, gmap, plan, SweepWinner(), stamps);
        CK(hipEventRecord(e1, 0));
    }
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, 0, clk);
    CK(hipDeviceSynchronize());
    unsigned long long h_clk[2];
    CK(hipMemcpy(h_clk, clk, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)h_clk[0] / (double)h_clk[1] * 0.1; // cycles per 10 ns
    std::vector<unsigned long long> st((size_t)grid * WAVES * 8);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t_min = ~0ull, t_max = 0;
    for (size_t w = 0; w < (size_t)grid * WAVES; w++) {
        if (!st[w * 8 + 7]) continue;
        t_min = std::min(t_min, st[w * 8]);
        t_max = std::max(t_max, st[w * 8 + 7]);
    }
    const char *names[] = {"entry_to_first_item_ready", "first_item_whole_round", "second_item_three_column_tiles", "second_item_wait_vmcnt0",
                           "second_item_expand_issue", "second_item_expand_lgkmcnt0", "second_item_dma_issue"};
    std::vector<double> ph[7], start;
    for (size_t w = 0; w < (size_t)grid * WAVES; w++) {
        const unsigned long long *s = &st[w * 8];
        if (!s[7] || !s[6]) continue; // (waves with fewer than three items carry no steady-state stamps)
        for (int k = 0; k < 7; k++) ph[k].push_back((double)(s[k + 1] - s[k]));
        start.push_back((double)(s[0] - t_min));
    }
    printf("{\"kernel\": \"census_sweep_pm_kernel<2,4,3>\", \"items_per_block\": %.2f, \"W\": %d, \"H\": %d, \"D\": %d, \"blocks\": %d, \"waves\": %zu, \"launch_ms_hipevents\": %.4f, "
           "\"in_kernel_clock_ghz\": %.3f, \"first_stamp_to_last_stamp_cycles\": %llu, \"first_to_last_us\": %.2f, \"phases\": {",
           (double)plan.items / plan.step, W, H, D, grid, ph[6].size(), ms, ghz, t_max - t_min, (double)(t_max - t_min) / ghz / 1e3);
    for (int k = 0; k < 7; k++) {
        double sum = 0;
        for (double x : ph[k]) sum += x;
        const double mean = sum / std::max<size_t>(1, ph[k].size());
        printf("%s\"%s\": {\"mean_cyc\": %.0f, \"median_cyc\": %.0f, \"p90_cyc\": %.0f, \"mean_us\": %.3f}", k ? ", " : "", names[k], mean, pct(ph[k], 0.5), pct(ph[k], 0.9),
               mean / ghz / 1e3);
    }
    printf("}, \"block_start_after_first_cycles\": {\"p10\": %.0f, \"p50\": %.0f, \"p90\": %.0f, \"max\": %.0f}}\n", pct(start, 0.1), pct(start, 0.5), pct(start, 0.9),
           pct(start, 0.999));
    return 0;
}

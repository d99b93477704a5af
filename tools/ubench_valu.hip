// Micro-benchmark: issue rate of the integer VALU instructions the census sweeps are made of (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o gpurun_out/ubench_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP> __global__ void k(uint32_t *out, uint32_t seed, int iters) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    for (int i = 0; i < iters; i++) {
#define STEP(x)                                                                                  \
    if (OP == 0) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x) : "v"(seed));                  \
    if (OP == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(seed));                       \
    if (OP == 2) asm volatile("v_min3_i32 %0, %0, %1, %1" : "+v"(x) : "v"(seed));                  \
    if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(seed));               \
    if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(seed));                       \
    if (OP == 5) asm volatile("v_lshl_or_b32 %0, %0, 10, %1" : "+v"(x) : "v"(seed));               \
    if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(seed));              \
    if (OP == 7) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(seed));
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int OP> void run(const char *name, uint32_t *d) {
    const int blocks = 256 * 8, threads = 256, iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(d, 1, 16);
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(d, 1, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * (threads / 64) * iters * 16;
    printf("%-16s %8.3f ms  %.3f T wave-instr/s  (%.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", name, ms, winstr / ms / 1e9,
           1024.0 * 2.4e9 / (winstr / (ms * 1e-3)));
}

int main() {
    uint32_t *d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_bcnt_u32_b32", d);
    run<1>("v_xor_b32", d);
    run<2>("v_min3_i32", d);
    run<3>("v_mad_u32_u24", d);
    run<4>("v_add_u32", d);
    run<5>("v_lshl_or_b32", d);
    run<6>("v_cndmask_b32", d);
    run<7>("v_min_u32", d);
    return 0;
}

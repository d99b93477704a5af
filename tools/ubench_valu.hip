// Micro-benchmark: issue rate of the integer VALU instructions the census sweeps are made of (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o gpurun_out/ubench_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP> __global__ void k(uint32_t *out, uint32_t seed, int iters) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pa0 = {1.0f * a0, 2.0f}, pa1 = pa0 * 3.0f, pa2 = pa0 * 5.0f, pa3 = pa0 * 7.0f, pa4 = pa0 * 9.0f, pa5 = pa0 * 11.0f, pa6 = pa0 * 13.0f, pa7 = pa0 * 15.0f;
    const f32x2 pseed = {1.0f * seed, 0.5f}, pref = {0.25f * seed, 0.25f * seed};
    for (int i = 0; i < iters; i++) {
#define STEP(x)                                                                                  \
    if (OP == 0) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(x) : "v"(seed));                  \
    if (OP == 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(seed));                       \
    if (OP == 2) asm volatile("v_min3_i32 %0, %0, %1, %1" : "+v"(x) : "v"(seed));                  \
    if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(seed));               \
    if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(seed));                       \
    if (OP == 5) asm volatile("v_lshl_or_b32 %0, %0, 10, %1" : "+v"(x) : "v"(seed));               \
    if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(seed));              \
    if (OP == 7) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(seed));                        \
    if (OP == 8) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(x), "v"(seed) : "vcc");           \
    if (OP == 9) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x) : : "vcc");           \
    if (OP == 10) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(x) : "v"(seed));             \
    if (OP == 11) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(seed));                      \
    if (OP == 12) asm volatile("v_cmp_gt_f32 vcc, %1, %0\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x) : "v"(seed) : "vcc"); \
    if (OP == 13) { uint32_t t; asm volatile("v_sub_f32 %1, %2, %0\n\tv_alignbit_b32 %0, %0, %1, 31" : "+v"(x), "=&v"(t) : "v"(seed)); } \
    if (OP == 14) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(seed) : "vcc"); \
    if (OP == 15) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##x) : "v"(pseed));               \
    if (OP == 16) { asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(p##x) : "v"(pseed), "v"(pref)); \
                    asm volatile("v_alignbit_b32 %0, %0, %1, 31\n\tv_alignbit_b32 %0, %0, %2, 31" : "+v"(x) : "v"(p##x[0]), "v"(p##x[1])); } \
    if (OP == 17) asm volatile("v_readlane_b32 s20, %0, 3\n\ts_nop 0" : : "v"(x) : "s20");    \
    if (OP == 18) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc"); \
    if (OP == 19) asm volatile("v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x)); \
    if (OP == 20) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(seed)); \
    if (OP == 21) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x) : "v"(seed));    \
    if (OP == 22) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(seed));                      \
    if (OP == 23) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(seed));                 \
    if (OP == 24) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(x), "v"(seed) : "vcc");       \
    if (OP == 25) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(seed));                          \
    if (OP == 26) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x));                               \
    if (OP == 27) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(seed));                      \
    if (OP == 28) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(x) : "v"(seed));                     \
    if (OP == 29) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(seed));                      \
    if (OP == 30) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(seed));                  \
    if (OP == 31) asm volatile("v_cmp_eq_f32_e64 s[20:21], %0, %1" : : "v"(x), "v"(seed) : "s20", "s21"); \
    if (OP == 32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(seed));                      \
    if (OP == 33) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(seed));         \
    if (OP == 34) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(x));                                \
    if (OP == 35) asm volatile("v_mov_b64 %0, %1" : "=v"(p##x) : "v"(pseed));
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(pa0[0] + pa1[1] + pa2[0] + pa3[1] + pa4[0] + pa5[1] + pa6[0] + pa7[1]);
}

// the instruction mix of one census_sweep voxel pair, on registers only (no LDS): 4 v_xor, 4 v_bcnt (two with a running sum),
// 2 v_lshl_or, 1 v_min3 -- what the sweep could reach if nothing but VALU issue limited it
__global__ void sweep_mix(uint32_t *out, uint32_t seed, int iters) {
    uint32_t s0 = seed + threadIdx.x, s1 = s0 * 3, t0 = s0 * 5, t1 = s0 * 7, t2 = s0 * 11, t3 = s0 * 13, k = 0x7FFFFFFFu, c0, c1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            asm volatile("v_xor_b32 %0, %2, %4\n\tv_xor_b32 %1, %3, %5\n\tv_bcnt_u32_b32 %0, %0, 0\n\tv_bcnt_u32_b32 %0, %1, %0\n\tv_lshl_or_b32 %0, %0, 12, %6"
                         : "=&v"(c0), "=&v"(c1) : "v"(s0), "v"(s1), "v"(t0), "v"(t1), "s"(i + u));
            asm volatile("v_xor_b32 %0, %2, %4\n\tv_xor_b32 %1, %3, %5\n\tv_bcnt_u32_b32 %0, %0, 0\n\tv_bcnt_u32_b32 %0, %1, %0\n\tv_lshl_or_b32 %0, %0, 12, %6"
                         : "=&v"(c1), "=&v"(t2) : "v"(s0), "v"(s1), "v"(t1), "v"(t3), "s"(i + u + 1));
            asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(k) : "v"(c0), "v"(c1));
            t0 += k & 1; // keeps the chain alive without a memory access
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = k ^ t0 ^ t2;
}

void run_mix(uint32_t *d) {
    const int blocks = 256 * 8, threads = 256, iters = 2048;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    sweep_mix<<<blocks, threads>>>(d, 1, 16);
    hipEventRecord(e0);
    sweep_mix<<<blocks, threads>>>(d, 1, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double voxels = (double)blocks * threads * iters * 8 * 2;
    printf("sweep mix        %8.3f ms  %.1f Gvoxels/s  (1080p x 256 = 530.8 Mvoxels in %.1f us at this rate)\n", ms, voxels / ms / 1e6, 530.8e6 / (voxels / ms) * 1e3);
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
    run<8>("v_cmp_gt_f32", d);
    run<9>("v_addc_co_u32", d);
    run<10>("v_alignbit_b32", d);
    run<11>("v_sub_f32", d);
    run<12>("cmp+addc (x2)", d);
    run<13>("sub+alignbit(x2)", d);
    run<14>("v_sub_co_u32", d);
    run<15>("v_pk_add_f32", d);
    run<16>("pk_sub+2alignbit", d);
    run<17>("v_readlane+nop", d);
    run<18>("s_add_u32", d);
    run<19>("v_max_f32_dpp", d);
    run<20>("v_mov_b32_dpp", d);
    run<21>("v_cndmask_e64 sgpr", d);
    run<22>("v_max_f32", d);
    run<23>("v_max3_f32", d);
    run<24>("v_cmp_class_f32", d);
    run<25>("v_mov_b32", d);
    run<26>("v_lshlrev_b32", d);
    run<27>("v_and_b32", d);
    run<28>("v_fmac_f32", d);
    run<29>("v_mul_f32", d);
    run<30>("v_fma_f32", d);
    run<31>("v_cmp_eq_f32_e64", d);
    run<32>("v_add_f32", d);
    run<33>("v_cndmask_e32 vcc", d);
    run<34>("v_bfe_u32", d);
    run<35>("v_mov_b64", d);
    run_mix(d);
    return 0;
}

// Micro-benchmark: issue rates of what a float division and its replacement by a shared double reciprocal (svh_guided_wave_impl.h,
// div_by_shared) are made of (gfx950).  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench_div.hip -o gpurun_out/ubench_div
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP> __global__ void k(float *out, float seed, int iters) {
    float a[8];
    double d[8];
    for (int u = 0; u < 8; u++) {
        a[u] = seed + threadIdx.x * (u + 1);
        d[u] = a[u];
    }
    const double rd = 1.0 / (double)(seed + 3.0f);
    const float y = seed + 3.0f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (OP == 0) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[u]) : "v"(a[u]));
                if (OP == 1) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[u]) : "v"(d[u]));
                if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[u]) : "v"(rd));
                if (OP == 3) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[u]) : "v"(rd));
                if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[u]));
                if (OP == 5) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[u]) : "v"(y) : "vcc");
                if (OP == 6) asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(a[u]) : "v"(y) : "vcc");
                if (OP == 7) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(a[u]) : "v"(y));
                if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[u]) : "v"(y));
                if (OP == 9) a[u] = a[u] / y;                                           // the float division as compiled
                if (OP == 10) a[u] = (float)((double)a[u] * rd);                        // div_by_shared without its flag
                if (OP == 11) { a[u] = (float)((double)a[u] * rd); asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(a[u]), "v"(0x90) : "vcc"); }
                if (OP == 12) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[u]));
            }
        }
    }
    float s = 0;
    for (int u = 0; u < 8; u++) s += a[u] + (float)d[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP> void run(const char *name, float *d) {
    const int blocks = 256 * 8, threads = 256, iters = 2048;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(d, 1.5f, 16);
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(d, 1.5f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * (threads / 64) * iters * 16;
    printf("%-28s %8.3f ms  %.2f cycles per wave and item per SIMD at 2.4 GHz\n", name, ms, 1024.0 * 2.4e9 / (n / (ms * 1e-3)));
}

int main() {
    float *d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_cvt_f64_f32", d);
    run<1>("v_cvt_f32_f64", d);
    run<2>("v_mul_f64", d);
    run<3>("v_fma_f64", d);
    run<4>("v_rcp_f32", d);
    run<5>("v_div_scale_f32", d);
    run<6>("v_div_fmas_f32", d);
    run<7>("v_div_fixup_f32", d);
    run<8>("v_fma_f32", d);
    run<9>("x / y (float)", d);
    run<10>("(float)((double)x * rd)", d);
    run<11>("  + v_cmp_class_f32", d);
    run<12>("v_rcp_f64", d);
    return 0;
}

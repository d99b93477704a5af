// Micro-benchmark: HBM write rate against the size of the contiguous piece a store instruction gives each pixel (gfx950).
// A volume of NPX pixels x 1 KiB (256 floats); a block of eight waves owns 128 consecutive pixels (128 KiB) and writes every byte of
// them once, in pieces of PIECE bytes per pixel and instruction: PIECE / 16 neighbouring lanes share a pixel, a wave's instruction
// covers 1024 / PIECE pixels 1 KiB apart, wave w of the block takes the pieces w, w + 8, ... of a pixel's run (what the column-sum cost
// kernel does with PIECE = 64: cost_volume_colsum_kernel, svh_cost_volume_tiled.hip).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_store.hip -o _ab/ubench_store ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int PIECE> __global__ void __launch_bounds__(512) k(float4 *out, float v) {
    constexpr int LPP = PIECE / 16, PPI = 64 / LPP, PIECES = 1024 / PIECE; // lanes per pixel, pixels per instruction, pieces per pixel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a = lane / LPP, c = lane % LPP;
    float4 *base = out + (size_t)blockIdx.x * 128 * 64; // 128 pixels x 64 float4
    const float4 val = make_float4(v, v + lane, v, v);
    for (int piece = wave; piece < PIECES; piece += 8)
        for (int p0 = 0; p0 < 128; p0 += PPI) base[(size_t)(p0 + a) * 64 + piece * LPP + c] = val;
}

template <int PIECE> void run(float4 *d, int npx) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<PIECE><<<npx / 128, 512>>>(d, 1.0f);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++) k<PIECE><<<npx / 128, 512>>>(d, 1.0f + r);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("pieces of %4d bytes: %.3f ms for %.2f GB  %.2f TB/s\n", PIECE, ms, npx * 1024.0 / 1e9, npx * 1024.0 / ms / 1e9);
}

int main() {
    const int npx = 1920 * 1080 - (1920 * 1080) % 128; // 1080p x 256 floats
    float4 *d;
    (void)hipMalloc(&d, (size_t)npx * 1024);
    run<64>(d, npx);
    run<128>(d, npx);
    run<256>(d, npx);
    run<512>(d, npx);
    run<1024>(d, npx);
    run<64>(d, npx);
    return 0;
}

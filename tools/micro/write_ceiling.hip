// Write-only ceiling on this GPU for a volume of the C2 size (1920 x 1080 x 128 floats): what a cost-volume kernel that only stores
// could reach.  hipcc --offload-arch=gfx950 -O3 tools/micro/write_ceiling.hip -o gpurun_out/write_ceiling && gpurun_out/write_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// a wave stores 256 B per instruction, rows of `D` floats per pixel, 4 waves interleaved over pixels (the shape of hamming_volume_tiled_kernel)
template <int NT> __global__ void __launch_bounds__(256) rows_b32(float *out, int D, int px_per_block) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * px_per_block;
    for (int u = wave; u < px_per_block; u += 4) {
        float *o = out + (p0 + u) * D;
        for (int d = lane; d < D; d += 64) {
            if (NT) __builtin_nontemporal_store((float)(d + u), o + d); else o[d] = (float)(d + u);
        }
    }
}
// the same with NWV waves per block (blockDim = 64 NWV)
template <int NWV> __global__ void __launch_bounds__(64 * NWV) rows_b32_waves(float *out, int D, int px_per_block) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t p0 = (int64_t)blockIdx.x * px_per_block;
    for (int u = wave; u < px_per_block; u += NWV) {
        float *o = out + (p0 + u) * D;
        for (int d = lane; d < D; d += 64) o[d] = (float)(d + u);
    }
}
// 16 B per lane: 1 KiB per wave instruction, a wave owns consecutive pixels
template <int NT> __global__ void __launch_bounds__(256) rows_b128(float *out, int D, int px_per_block) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t p0 = (int64_t)blockIdx.x * px_per_block;
    const int per_wave = px_per_block / 4;
    v4f *o = reinterpret_cast<v4f *>(out + (p0 + (int64_t)wave * per_wave) * D);
    const int n4 = per_wave * D / 4;
    for (int e = lane; e < n4; e += 64) {
        const v4f v = {(float)e, 1.f, 2.f, 3.f};
        if (NT) __builtin_nontemporal_store(v, o + e); else o[e] = v;
    }
}
template <int NT> __global__ void __launch_bounds__(256) flat_b128(v4f *out, int64_t n4) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        const v4f v = {(float)e, 1.f, 2.f, 3.f};
        if (NT) __builtin_nontemporal_store(v, out + e); else out[e] = v;
    }
}

template <class F> static void timeit(const char *name, double bytes, F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) f();
    CK(hipEventRecord(a));
    const int n = 20;
    for (int i = 0; i < n; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-28s %.4f ms  %.2f TB/s\n", name, ms / n, bytes / (ms / n * 1e-3) / 1e12);
}

int main() {
    const int W = 1920, H = 1080, D = 128;
    const int64_t n = (int64_t)W * H * D;
    float *buf; CK(hipMalloc(&buf, n * 4));
    const double bytes = n * 4.0;
    const int ppb = 64, blocks = W * H / ppb;
    timeit("rows_b32", bytes, [&] { rows_b32<0><<<blocks, 256>>>(buf, D, ppb); });
    timeit("rows_b32 nt", bytes, [&] { rows_b32<1><<<blocks, 256>>>(buf, D, ppb); });
    timeit("rows_b128", bytes, [&] { rows_b128<0><<<blocks, 256>>>(buf, D, ppb); });
    timeit("rows_b128 nt", bytes, [&] { rows_b128<1><<<blocks, 256>>>(buf, D, ppb); });
    for (int pb : {8, 16, 32, 128, 256}) {
        char nm[64]; snprintf(nm, 64, "rows_b128 %d px/block", pb);
        timeit(nm, bytes, [&] { rows_b128<0><<<W * H / pb, 256>>>(buf, D, pb); });
        snprintf(nm, 64, "rows_b32 %d px/block", pb);
        timeit(nm, bytes, [&] { rows_b32<0><<<W * H / pb, 256>>>(buf, D, pb); });
    }
    for (int pb : {32, 64, 128}) {
        char nm[64]; snprintf(nm, 64, "rows_b32 8 waves %d px", pb);
        timeit(nm, bytes, [&] { rows_b32_waves<8><<<W * H / pb, 512>>>(buf, D, pb); });
        snprintf(nm, 64, "rows_b32 16 waves %d px", pb);
        timeit(nm, bytes, [&] { rows_b32_waves<16><<<W * H / pb, 1024>>>(buf, D, pb); });
    }
    for (int g : {65536, 262144}) {
        char nm[64]; snprintf(nm, 64, "flat_b128 grid %d", g);
        timeit(nm, bytes, [&] { flat_b128<0><<<g, 256>>>((v4f *)buf, n / 4); });
        snprintf(nm, 64, "flat_b128 nt grid %d", g);
        timeit(nm, bytes, [&] { flat_b128<1><<<g, 256>>>((v4f *)buf, n / 4); });
    }
    timeit("hipMemsetAsync", bytes, [&] { CK(hipMemsetAsync(buf, 0, n * 4)); });
    CK(hipDeviceSynchronize());
    return 0;
}

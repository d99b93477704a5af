// Winner extraction (A10), index -> disparity, selected cost, truncated volume (A11) and cost-based
// sub-pixel refinement (A12).
#include "svh_internal.h"
#include "svh_sgm_lines.h"

namespace svh {

__device__ __forceinline__ uint32_t order_key(float v) {
    if (v == 0.0f) v = 0.0f; // fold -0 onto +0, they compare equal
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// extractSelectedIndex, correlation_base.h:427-464.  One wavefront per pixel; lane l scans disparities
// l, l+64, ... in increasing order (256-byte coalesced reads of the pixel's contiguous costs), then the wave
// combines with "extremum wins, ties to the larger index".  A NaN never replaces the incumbent, and a NaN at
// index 0 is never replaced (every comparison with it is false).
// EXTRACT_PB pixels per wave iteration: their first chunks are loaded before any of them is reduced, so a wave keeps
// EXTRACT_PB loads in flight instead of one (the kernel is latency-bound otherwise: one 1 KiB load per wave and round trip)
constexpr int EXTRACT_PB = 4;

// four consecutive costs with only the 4-byte alignment every cost has: global_load_dwordx4 does not need more on gfx950,
// so rows of any length (2-D volumes flatten to odd lengths) are read 16 bytes per lane
struct __attribute__((packed, aligned(4))) Costs4 {
    float x, y, z, w;
};

template <bool COST>
__global__ void __launch_bounds__(256) extract_index_kernel(const float *__restrict__ cv, int64_t npx, int D, int32_t *__restrict__ idx,
                                                           unsigned long long *__restrict__ keys, int key_offset, int key_total) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int D4 = D & ~3; // the last D - D4 costs are read one per lane
    for (int64_t p0 = wave * EXTRACT_PB; p0 < npx; p0 += nwaves * EXTRACT_PB) {
        Costs4 first[EXTRACT_PB];
        float head[EXTRACT_PB], tail[EXTRACT_PB];
#pragma unroll
        for (int q = 0; q < EXTRACT_PB; q++) {
            const int64_t p = min(p0 + q, npx - 1);
            const float *row = cv + p * D;
            first[q] = Costs4{0.f, 0.f, 0.f, 0.f};
            if (4 * lane < D4) first[q] = *reinterpret_cast<const Costs4 *>(row + 4 * lane);
            tail[q] = D4 + lane < D ? row[D4 + lane] : 0.0f;
            head[q] = row[0];
        }
#pragma unroll
        for (int q = 0; q < EXTRACT_PB; q++) {
            const int64_t p = p0 + q;
            if (p >= npx) break;
            const float *row = cv + p * D;
            float bv = 0.0f;
            int bd = -1;
            auto consider = [&](float v, int d) {
                if (isnan(v)) return;
                const bool take = bd < 0 || (COST ? v <= bv : v >= bv); // later index wins ties
                if (take) {
                    bv = v;
                    bd = d;
                }
            };
            if (4 * lane < D4) {
                consider(first[q].x, 4 * lane);
                consider(first[q].y, 4 * lane + 1);
                consider(first[q].z, 4 * lane + 2);
                consider(first[q].w, 4 * lane + 3);
            }
            for (int d = 4 * lane + 256; d < D4; d += 256) {
                const Costs4 v = *reinterpret_cast<const Costs4 *>(row + d);
                consider(v.x, d);
                consider(v.y, d + 1);
                consider(v.z, d + 2);
                consider(v.w, d + 3);
            }
            if (D4 + lane < D) consider(tail[q], D4 + lane);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                float ov = __shfl_xor(bv, off);
                int od = __shfl_xor(bd, off);
                bool take = od >= 0 && (bd < 0 || (COST ? (ov < bv || (ov == bv && od > bd)) : (ov > bv || (ov == bv && od > bd))));
                if (take) {
                    bv = ov;
                    bd = od;
                }
            }
            const bool first_nan = isnan(head[q]);
            if (lane == 0) {
                if (idx) idx[p] = (first_nan || bd < 0) ? 0 : bd;
                if (keys) {
                    unsigned long long key;
                    if (key_offset == 0 && first_nan) key = COST ? (unsigned long long)(uint32_t)(key_total - 1) : (0xFFFFFFFFull << 32);
                    else if (bd < 0) key = COST ? ~0ull : 0ull;
                    else {
                        uint32_t gd = (uint32_t)(key_offset + bd);
                        key = ((unsigned long long)order_key(bv) << 32) | (COST ? (uint32_t)(key_total - 1) - gd : gd);
                    }
                    keys[p] = key;
                }
            }
        }
    }
}

// The same for rows of at most 256 costs, a multiple of four: a pixel takes LPP = 16 / 32 / 64 lanes (four costs per lane, one 16-byte load), a
// wave works on four, two or one pixels at a time, and the combine is not six rounds of (value, index) exchanges but two all-reduces inside
// the pixel's lanes (DPP row rotations, v_permlane16/32_swap): the extremum, then the largest index among the lanes that hold a value EQUAL
// to it.  "The value at index 0 is NaN" (index 0 wins: every comparison with it is false) enters the index reduction as a key above every
// index.  The wave-per-pixel kernel took 0.44 ms at 1080p for 64, 128 and 256 costs alike -- the time of its instruction stream, not of its
// bytes.
template <bool COST, int LPP>
__global__ void __launch_bounds__(256) extract_index_packed_kernel(const float *__restrict__ cv, int64_t npx, int D, int32_t *__restrict__ idx,
                                                                  unsigned long long *__restrict__ keys, int key_offset, int key_total) {
    constexpr int PPW = 64 / LPP, G = 4, PB = G * PPW;
    const int lane = threadIdx.x & 63, sub = lane / LPP, dl = lane % LPP, d0 = dl * 4;
    const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const bool lane_on = d0 < D;
    const bool fold_first_nan = !keys || key_offset == 0; // (a later shard's key ignores the rule, its index map does not: reduced separately then)
    constexpr int FIRST_NAN = 1 << 20;
    for (int64_t p0 = wave * PB; p0 < npx; p0 += nwaves * PB) {
        Costs4 v[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int64_t p = min(p0 + g * PPW + sub, npx - 1);
            v[g] = *reinterpret_cast<const Costs4 *>(cv + p * D + (lane_on ? d0 : 0)); // (a lane past D: the row's first costs, unused)
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (p0 + g * PPW >= npx) break; // (wave-uniform)
            const int64_t p = p0 + g * PPW + sub;
            const float s[4] = {v[g].x, v[g].y, v[g].z, v[g].w};
            float A = COST ? INFINITY : -INFINITY;
#pragma unroll
            for (int k = 0; k < 4; k++) A = COST ? fminf(A, s[k]) : fmaxf(A, s[k]); // (a NaN never enters)
            const float M = pixel_allreduce_f32<LPP, COST>(lane_on ? A : (COST ? INFINITY : -INFINITY));
            int kb = -1;
#pragma unroll
            for (int k = 0; k < 4; k++) kb = (s[k] == M) ? k : kb; // ties: the larger index
            int key = (lane_on && kb >= 0) ? d0 + kb + 1 : 0;
            const bool first_nan_here = dl == 0 && isnan(s[0]);
            if (fold_first_nan && first_nan_here) key = FIRST_NAN;
            const int best = pixel_allreduce_max<LPP>(key);
            int first_nan = best == FIRST_NAN;
            if (!fold_first_nan) first_nan = pixel_allreduce_max<LPP>(first_nan_here ? 1 : 0); // (wave-uniform branch)
            const bool writer = lane_on && p < npx && (best == 0 || best == FIRST_NAN ? dl == 0 : key == best);
            if (writer) {
                const int bd = (best == 0 || best == FIRST_NAN) ? -1 : best - 1;
                if (idx) idx[p] = (first_nan || bd < 0) ? 0 : bd;
                if (keys) {
                    unsigned long long kk;
                    if (key_offset == 0 && first_nan) kk = COST ? (unsigned long long)(uint32_t)(key_total - 1) : (0xFFFFFFFFull << 32);
                    else if (bd < 0) kk = COST ? ~0ull : 0ull;
                    else {
                        const uint32_t gd = (uint32_t)(key_offset + bd);
                        kk = ((unsigned long long)order_key(M) << 32) | (COST ? (uint32_t)(key_total - 1) - gd : gd);
                    }
                    keys[p] = kk;
                }
            }
        }
    }
}

// Rows of any length up to 1 024 costs (2-D volumes flatten to 17 x 17 = 289, 9 x 33 = 297; the reference's 1080p benchmark rows search 320
// disparities) with the packed kernel's combine: a wave per pixel, NQ pieces of four costs per lane (cost d of the pixel in lane (d % 256) / 4,
// piece d / 256), the extremum by one all-reduce, the largest index among the costs EQUAL to it by a second.  The wave-per-pixel kernel above
// spends six rounds of (value, index) exchanges per pixel and fetches what lies beyond the first 256 costs in a dependent second round:
// 0.66 ms for 1080p x 320 (4.0 TB/s).  A piece that would reach past its row is read cost by cost.
template <bool COST, int NQ>
__global__ void __launch_bounds__(256) extract_index_wide_kernel(const float *__restrict__ cv, int64_t npx, int D, int32_t *__restrict__ idx,
                                                                unsigned long long *__restrict__ keys, int key_offset, int key_total) {
    constexpr int G = 4;
    const int lane = threadIdx.x & 63;
    const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const bool fold_first_nan = !keys || key_offset == 0; // (a later shard's key ignores the rule, its index map does not: reduced separately then)
    constexpr int FIRST_NAN = 1 << 20;
    for (int64_t p0 = wave * G; p0 < npx; p0 += nwaves * G) {
        Costs4 v[G][NQ];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const float *row = cv + min(p0 + g, npx - 1) * D;
#pragma unroll
            for (int n = 0; n < NQ; n++) {
                const int d = 4 * lane + 256 * n;
                if (d + 3 < D) {
                    v[g][n] = *reinterpret_cast<const Costs4 *>(row + d);
                } else { // (the row's last piece, or past the row: never read beyond the row's end)
                    v[g][n].x = d < D ? row[d] : 0.0f;
                    v[g][n].y = d + 1 < D ? row[d + 1] : 0.0f;
                    v[g][n].z = d + 2 < D ? row[d + 2] : 0.0f;
                    v[g][n].w = 0.0f;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (p0 + g >= npx) break; // (wave-uniform)
            const int64_t p = p0 + g;
            float A = COST ? INFINITY : -INFINITY;
#pragma unroll
            for (int n = 0; n < NQ; n++) {
                const float s[4] = {v[g][n].x, v[g][n].y, v[g][n].z, v[g][n].w};
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (4 * lane + 256 * n + k < D) A = COST ? fminf(A, s[k]) : fmaxf(A, s[k]); // (a NaN never enters)
            }
            const float M = pixel_allreduce_f32<64, COST>(A);
            int key = 0; // 1 + the largest index among this lane's costs equal to the extremum
#pragma unroll
            for (int n = 0; n < NQ; n++) {
                const float s[4] = {v[g][n].x, v[g][n].y, v[g][n].z, v[g][n].w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int d = 4 * lane + 256 * n + k;
                    if (d < D && s[k] == M) key = d + 1;
                }
            }
            const bool first_nan_here = lane == 0 && isnan(v[g][0].x);
            if (fold_first_nan && first_nan_here) key = FIRST_NAN;
            const int best = pixel_allreduce_max<64>(key);
            int first_nan = best == FIRST_NAN;
            if (!fold_first_nan) first_nan = pixel_allreduce_max<64>(first_nan_here ? 1 : 0); // (wave-uniform branch)
            const bool writer = (best == 0 || best == FIRST_NAN) ? lane == 0 : key == best;
            if (writer) {
                const int bd = (best == 0 || best == FIRST_NAN) ? -1 : best - 1;
                if (idx) idx[p] = (first_nan || bd < 0) ? 0 : bd;
                if (keys) {
                    unsigned long long kk;
                    if (key_offset == 0 && first_nan) kk = COST ? (unsigned long long)(uint32_t)(key_total - 1) : (0xFFFFFFFFull << 32);
                    else if (bd < 0) kk = COST ? ~0ull : 0ull;
                    else {
                        const uint32_t gd = (uint32_t)(key_offset + bd);
                        kk = ((unsigned long long)order_key(M) << 32) | (COST ? (uint32_t)(key_total - 1) - gd : gd);
                    }
                    keys[p] = kk;
                }
            }
        }
    }
}

__global__ void keys_to_index_kernel(const unsigned long long *__restrict__ keys, int64_t n, int total, bool cost,
                                     int32_t *__restrict__ idx) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        uint32_t low = (uint32_t)(keys[p] & 0xFFFFFFFFull);
        int d = cost ? total - 1 - (int)low : (int)low;
        if (d < 0 || d >= total) d = 0; // "nothing comparable anywhere": the reference stays on index 0
        idx[p] = d;
    }
}

// selectedIndexToDisp, correlation_base.h:511-532
__global__ void index_to_disp_kernel(const int32_t *__restrict__ idx, int64_t n, int sign, int32_t offset, int32_t *__restrict__ disp) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x)
        disp[p] = sign * idx[p] + offset;
}

// selectedCost, correlation_base.h:557-577
__global__ void selected_cost_kernel(const float *__restrict__ cv, const int32_t *__restrict__ idx, int64_t n, int D,
                                     float *__restrict__ out) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x)
        out[p] = cv[p * D + (uint32_t)idx[p]];
}

// truncatedCostVolume, correlation_base.h:579-674.  One thread per output tap.
__global__ void truncated_cv_kernel(int sdir, int ddir, const float *__restrict__ cv, const int32_t *__restrict__ idx, int H, int W,
                                    int D, int h_r, int v_r, int r, float *__restrict__ tcv) {
    const int T = sdir == SVH_TCV_BOTH ? 4 * r + 1 : 2 * r + 1;
    const int64_t n = (int64_t)H * W * T;
    const float nan = __uint_as_float(0x7FC00000u);
    const int sgn = (ddir == SVH_RIGHT_TO_LEFT) ? -1 : 1; // :618, :635
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int slot = (int)(e % T);
        const int64_t px = e / T;
        const int j = (int)(px % W), i = (int)(px / W);
        // which tap d in [0, 2r] and which flavour (same-pixel or shifted-pixel) does this slot hold?
        int d;
        bool shifted;
        if (sdir == SVH_TCV_SAME) {
            d = slot;
            shifted = false;
        } else if (sdir == SVH_TCV_REVERSED) {
            d = slot;
            shifted = true;
        } else { // Both: d<r -> (2d, 2d+1); d==r -> 2r; d>r -> (2d-1, 2d)   (:640-650)
            if (slot < 2 * r) {
                d = slot / 2;
                shifted = slot & 1;
            } else if (slot == 2 * r) {
                d = r;
                shifted = false;
            } else {
                d = (slot + 1) / 2;
                shifted = !(slot & 1);
            }
        }
        const int p = idx[px] + d - r;
        const bool rows_bad = i < v_r || i + v_r >= H;
        float val;
        if (!shifted) {
            bool bad = p < 0 || p >= D || j < h_r || j + p + h_r >= W || rows_bad; // :606-608
            val = bad ? nan : cv[px * D + p];
        } else {
            int jp = j + sgn * (d - r);
            int mn = min(jp, j), mx = max(jp, j);
            bool bad = p < 0 || p >= D || mn < h_r || mx + h_r >= W || rows_bad; // :623-625
            val = bad ? nan : cv[((int64_t)i * W + jp) * D + p];
        }
        tcv[e] = val;
    }
}

// refineCostTriplet, cost_based_refinement.h:43-69
__device__ __forceinline__ float refine_triplet(int kernel, float cm1, float c0, float c1) {
    if (kernel == SVH_EQUIANGULAR) {
        float alpha = copysignf(1.f, c0 - cm1);
        alpha *= fmaxf(fabsf(c0 - cm1), fabsf(c1 - c0));
        return (c1 - cm1) / (2 * alpha);
    }
    if (kernel == SVH_PARABOLA) return (cm1 - c1) / (2 * (c1 - 2 * c0 + cm1));
    return (logf(cm1) - logf(c1)) / (2 * (logf(c1) - 2 * logf(c0) + logf(cm1)));
}

// refineDispCostInterpolation, cost_based_refinement.h:128-163
__global__ void refine_kernel(int kernel, const float *__restrict__ tcv, const int32_t *__restrict__ raw, int64_t n, int T,
                              float *__restrict__ refined) {
    const int r = (T - 1) / 2;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const float *t = tcv + p * T;
        refined[p] = (float)raw[p] + refine_triplet(kernel, t[r - 1], t[r], t[r + 1]);
    }
}

// winner records of the Score branch's fused finish (wave_emit_record, svh_sgm_lines.h) -> selectedIndexToDisp / refineDispCostInterpolation
__global__ void finish_records_kernel(const float4 *__restrict__ rec, int64_t n, int kernel, int sign, int offset, int32_t *__restrict__ idx,
                                      int32_t *__restrict__ disp, float *__restrict__ refined) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const float4 r = rec[p];
        const int sel = __float_as_int(r.w);
        if (idx) idx[p] = sel;
        if (disp) disp[p] = sign * sel + offset;
        if (refined) refined[p] = (float)sel + refine_triplet(kernel, r.x, r.y, r.z);
    }
}

// flat index over (D1, D2) -> (d1, d2): the row-major '<=' scan of extractSelected2dIndex (correlation_base.h:466-509) is the
// 1-D scan over the flattened axis
__global__ void split_index_kernel(const int32_t *__restrict__ flat, int64_t n, int D2, int32_t *__restrict__ out2) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int f = flat[p];
        out2[2 * p] = f / D2;
        out2[2 * p + 1] = f % D2;
    }
}

// selected2dIndexToDisp, correlation_base.h:534-555
__global__ void index_2d_to_disp_kernel(const int32_t *__restrict__ idx, int64_t n, int lower0, int lower1, int32_t *__restrict__ disp) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        disp[2 * p] = idx[2 * p] + lower0;
        disp[2 * p + 1] = idx[2 * p + 1] + lower1;
    }
}

// truncatedBidirectionaCostVolume, correlation_base.h:677-725: valueOrAlt(..., NaN) around the selected 2-D index
__global__ void truncated_bidirectional_kernel(const float *__restrict__ cv, const int32_t *__restrict__ idx, int64_t npx, int D1, int D2, int r0,
                                               int r1, float *__restrict__ tcv) {
    const int T0 = 2 * r0 + 1, T1 = 2 * r1 + 1;
    const int64_t n = npx * T0 * T1;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int d1 = (int)(e % T1), d0 = (int)((e / T1) % T0);
        const int64_t px = e / ((int64_t)T0 * T1);
        const int p0 = idx[2 * px] + d0 - r0, p1 = idx[2 * px + 1] + d1 - r1;
        tcv[e] = (p0 >= 0 && p0 < D1 && p1 >= 0 && p1 < D2) ? cv[(px * D1 + p0) * D2 + p1] : __uint_as_float(0x7FC00000u);
    }
}

// ---- 2-D cost-based refinement (what examples/stereo-match --refine calls) -------------
// refineDisp2dCostInterpolation<kernel, isotropy>, cost_based_refinement.h:165-376.  One lane per pixel; tcv is
// (pixel, T0, T1), raw / refined (pixel, 2).  The score / cost nature of the volume is probed at the centre pixel like the
// reference does (:184-203; every comparison with a NaN is false).
template <bool ANISO>
__global__ void refine_2d_kernel(int kernel, const float *__restrict__ tcv, const int32_t *__restrict__ raw, int64_t npx, int64_t centre_px, int T0,
                                 int T1, float *__restrict__ refined) {
    const int r0 = (T0 - 1) / 2, r1 = (T1 - 1) / 2;
    bool is_score = false;
    if (ANISO) {
        const float *c = tcv + centre_px * T0 * T1;
        const float v0 = c[r0 * T1 + r1];
        is_score = v0 > c[(r0 + 1) * T1 + r1] || v0 > c[(r0 - 1) * T1 + r1] || v0 > c[r0 * T1 + r1 + 1] || v0 > c[r0 * T1 + r1 - 1];
    }
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npx; p += (int64_t)gridDim.x * blockDim.x) {
        const float *t = tcv + p * T0 * T1;
        auto at = [&](int a, int b) { return t[a * T1 + b]; };
        float delta0, delta1;
        const float d0c = refine_triplet(kernel, at(r0 - 1, r1), at(r0, r1), at(r0 + 1, r1));
        const float d1c = refine_triplet(kernel, at(r0, r1 - 1), at(r0, r1), at(r0, r1 + 1));
        if (!ANISO) { // :257-268
            delta0 = d0c;
            delta1 = d1c;
        } else {
            // extremum along axis 0 inside columns r1 -+ 1 (argminForRow, :205-225) and along axis 1 inside rows r0 -+ 1
            // (argminForCol, :227-247): '<=' / '>=' scans from +-inf, last extremum wins, NaN never selected, default 0
            int am[4];
#pragma unroll
            for (int which = 0; which < 4; which++) {
                const int fixed = which == 0 ? r1 - 1 : which == 1 ? r1 + 1 : which == 2 ? r0 - 1 : r0 + 1;
                const int n = which < 2 ? T0 : T1;
                float hat = is_score ? -INFINITY : INFINITY;
                int arg = 0;
                for (int a = 0; a < n; a++) {
                    const float v = which < 2 ? at(a, fixed) : at(fixed, a);
                    if (is_score ? (v >= hat) : (v <= hat)) {
                        hat = v;
                        arg = a;
                    }
                }
                am[which] = arg;
            }
            float d0_0 = d0c, d0_2 = d0c, d1_0 = d1c, d1_2 = d1c;
            if (am[0] > 0 && am[0] < T0 - 1) d0_0 = am[0] - r0 + refine_triplet(kernel, at(am[0] - 1, r1 - 1), at(am[0], r1 - 1), at(am[0] + 1, r1 - 1));
            if (am[1] > 0 && am[1] < T0 - 1) d0_2 = am[1] - r0 + refine_triplet(kernel, at(am[1] - 1, r1 + 1), at(am[1], r1 + 1), at(am[1] + 1, r1 + 1));
            if (am[2] > 0 && am[2] < T1 - 1) d1_0 = am[2] - r1 + refine_triplet(kernel, at(r0 - 1, am[2] - 1), at(r0 - 1, am[2]), at(r0 - 1, am[2] + 1));
            if (am[3] > 0 && am[3] < T1 - 1) d1_2 = am[3] - r1 + refine_triplet(kernel, at(r0 + 1, am[3] - 1), at(r0 + 1, am[3]), at(r0 + 1, am[3] + 1));
            // the two fitted lines delta0 = a0 delta1 + b0 and delta1 = a1 delta0 + b1, and their intersection (:310-358)
            const float a0 = (d0_2 - d0_0) / 2, b0 = (d0_0 + d0c + d0_2) / 3;
            const float a1 = (d1_2 - d1_0) / 2, b1 = (d1_0 + d1c + d1_2) / 3;
            delta0 = (a0 * b1 + b0) / (1 - a0 * a1);
            delta1 = (a1 * b0 + b1) / (1 - a0 * a1);
        }
        if (fabsf(delta0) > 1 || fabsf(delta1) > 1 || isnan(delta0) || isnan(delta1)) { // :362-366
            delta0 = 0;
            delta1 = 0;
        }
        refined[2 * p] = (float)raw[2 * p] + delta0;
        refined[2 * p + 1] = (float)raw[2 * p + 1] + delta1;
    }
}

// refineDisp2dCostPatchInterpolation<Parabola|Gaussian>, cost_based_refinement.h:378-436 with refineCostPatch (:71-126).
// The reference solves the 9x6 least-squares system numerically for every pixel; the design matrix is constant, so the
// normal equations have the closed form below (v, h in {-1,0,1}: S = sum L, Sv = sum v^2 L, Sh = sum h^2 L):
//     f_vv = Sv/2 - S/3   f_vh = (sum v h L)/4   f_hh = Sh/2 - S/3   f_v = (sum v L)/6   f_h = (sum h L)/6
// evaluated on L - L(0,0) (the five parameters do not depend on a constant offset; this keeps the cancellation small).
// Stationary point: [2 f_vv, f_vh; f_vh, 2 f_hh]^-1 [-f_v, -f_h] as adjugate / determinant (:109-116).
__global__ void refine_2d_patch_kernel(int kernel, const float *__restrict__ tcv, const int32_t *__restrict__ raw, int64_t npx, int T0, int T1,
                                       float *__restrict__ refined) {
    const int r0 = (T0 - 1) / 2, r1 = (T1 - 1) / 2;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npx; p += (int64_t)gridDim.x * blockDim.x) {
        const float *t = tcv + p * T0 * T1;
        float L[3][3];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) {
                const float c = t[(r0 - 1 + a) * T1 + r1 - 1 + b];
                L[a][b] = kernel == SVH_GAUSSIAN ? logf(c) : c; // :119-121
            }
        const float c00 = L[1][1];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) L[a][b] -= c00;
        const float top = L[0][0] + L[0][1] + L[0][2], mid = L[1][0] + L[1][1] + L[1][2], bot = L[2][0] + L[2][1] + L[2][2];
        const float lft = L[0][0] + L[1][0] + L[2][0], rgt = L[0][2] + L[1][2] + L[2][2];
        const float S = top + mid + bot;
        const float fvv = 0.5f * (top + bot) - S / 3.0f, fhh = 0.5f * (lft + rgt) - S / 3.0f;
        const float fvh = 0.25f * ((L[0][0] + L[2][2]) - (L[0][2] + L[2][0]));
        const float fv = (bot - top) / 6.0f, fh = (rgt - lft) / 6.0f;
        const float m00 = 2 * fvv, m01 = fvh, m11 = 2 * fhh;
        const float invdet = 1.0f / (m00 * m11 - m01 * m01);
        float delta0 = (m11 * invdet) * (-fv) + (-m01 * invdet) * (-fh);
        float delta1 = (-m01 * invdet) * (-fv) + (m00 * invdet) * (-fh);
        if (fabsf(delta0) > 1 || fabsf(delta1) > 1 || isnan(delta0) || isnan(delta1)) { // :424-428
            delta0 = 0;
            delta1 = 0;
        }
        refined[2 * p] = (float)raw[2 * p] + delta0;
        refined[2 * p + 1] = (float)raw[2 * p + 1] + delta1;
    }
}

int dev_extract_index(svh_context *ctx, int strategy, const float *cv, int64_t n_pixels, int D, int32_t *idx,
                      unsigned long long *keys, int key_index_offset, int key_total_D) {
    if (n_pixels == 0) return SVH_OK;
    if (D % 4 == 0 && D >= 4 && D <= 256) { // several pixels per wave (rows need no more than the 4-byte alignment every cost has)
        const int lpp = D <= 64 ? 16 : D <= 128 ? 32 : 64;
        const int gridp = grid_for(ceil_div(n_pixels, 4 * (64 / lpp)), 4, 256 * 8 * 4);
#define SVH_EXTRACT_P(C, L) SVH_LAUNCH(ctx, "extract_index", (extract_index_packed_kernel<C, L>), gridp, 256, 0, cv, n_pixels, D, idx, keys, key_index_offset, key_total_D)
        if (strategy == SVH_COST) {
            if (lpp == 16) SVH_EXTRACT_P(true, 16);
            else if (lpp == 32) SVH_EXTRACT_P(true, 32);
            else SVH_EXTRACT_P(true, 64);
        } else {
            if (lpp == 16) SVH_EXTRACT_P(false, 16);
            else if (lpp == 32) SVH_EXTRACT_P(false, 32);
            else SVH_EXTRACT_P(false, 64);
        }
#undef SVH_EXTRACT_P
        SVH_CHECK_LAUNCH(ctx);
        return SVH_OK;
    }
    if (D >= 1 && D <= 1024 && ctx->extract_index_wide) { // a wave per pixel, up to four pieces of four costs per lane
        const int nq = D <= 256 ? 1 : D <= 512 ? 2 : 4;
        const int gridw = grid_for(ceil_div(n_pixels, 4), 4, 256 * 8 * 4);
#define SVH_EXTRACT_W(C, N) SVH_LAUNCH(ctx, "extract_index", (extract_index_wide_kernel<C, N>), gridw, 256, 0, cv, n_pixels, D, idx, keys, key_index_offset, key_total_D)
        if (strategy == SVH_COST) {
            if (nq == 1) SVH_EXTRACT_W(true, 1);
            else if (nq == 2) SVH_EXTRACT_W(true, 2);
            else SVH_EXTRACT_W(true, 4);
        } else {
            if (nq == 1) SVH_EXTRACT_W(false, 1);
            else if (nq == 2) SVH_EXTRACT_W(false, 2);
            else SVH_EXTRACT_W(false, 4);
        }
#undef SVH_EXTRACT_W
        SVH_CHECK_LAUNCH(ctx);
        return SVH_OK;
    }
    int grid = grid_for(ceil_div(n_pixels, EXTRACT_PB), 4, 256 * 8 * 4);
#define SVH_EXTRACT(C) SVH_LAUNCH(ctx, "extract_index", (extract_index_kernel<C>), grid, 256, 0, cv, n_pixels, D, idx, keys, key_index_offset, key_total_D)
    if (strategy == SVH_COST) SVH_EXTRACT(true);
    else SVH_EXTRACT(false);
#undef SVH_EXTRACT
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

int dev_keys_to_index(svh_context *ctx, int strategy, const unsigned long long *keys, int64_t n, int total_D, int32_t *idx) {
    if (n == 0) return SVH_OK;
    SVH_LAUNCH(ctx, "keys_to_index", keys_to_index_kernel, grid_for(n, 256, 8192), 256, 0, keys, n, total_D, strategy == SVH_COST, idx);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

int dev_index_to_disp(svh_context *ctx, int ddir, const int32_t *idx, int64_t n, int32_t offset, int32_t *disp) {
    if (n == 0) return SVH_OK;
    SVH_LAUNCH(ctx, "index_to_disp", index_to_disp_kernel, grid_for(n, 256, 8192), 256, 0, idx, n, ddir == SVH_RIGHT_TO_LEFT ? 1 : -1,
               offset, disp);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

int dev_selected_cost(svh_context *ctx, const float *cv, const int32_t *idx, int64_t n, int D, float *out) {
    if (n == 0) return SVH_OK;
    SVH_LAUNCH(ctx, "selected_cost", selected_cost_kernel, grid_for(n, 256, 8192), 256, 0, cv, idx, n, D, out);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

int dev_truncated_cv(svh_context *ctx, int sdir, int ddir, const float *cv, const int32_t *idx, int H, int W, int D, int h_r,
                     int v_r, int r, float *tcv) {
    int T = sdir == SVH_TCV_BOTH ? 4 * r + 1 : 2 * r + 1;
    int64_t n = (int64_t)H * W * T;
    if (n == 0) return SVH_OK;
    SVH_LAUNCH(ctx, "truncated_cost_volume", truncated_cv_kernel, grid_for(n, 256, 16384), 256, 0, sdir, ddir, cv, idx, H, W, D, h_r, v_r,
               r, tcv);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

int dev_finish_records(svh_context *ctx, const float *records, int64_t npx, int refine_kernel, int disp_sign, int disp_offset, int32_t *idx,
                       int32_t *disp, float *refined) {
    if (npx == 0) return SVH_OK;
    SVH_LAUNCH(ctx, "finish_records", finish_records_kernel, grid_for(npx, 256, 8192), 256, 0, reinterpret_cast<const float4 *>(records), npx, refine_kernel,
               disp_sign, disp_offset, idx, disp, refine_kernel >= 0 ? refined : nullptr);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

int dev_refine(svh_context *ctx, int kernel, const float *tcv, const int32_t *raw, int64_t n, int T, float *refined) {
    if (n == 0) return SVH_OK;
    SVH_LAUNCH(ctx, "refine_disp", refine_kernel, grid_for(n, 256, 8192), 256, 0, kernel, tcv, raw, n, T, refined);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

} // namespace svh

using namespace svh;

static int same_map_shape(svh_context *ctx, const svh_array *vol, const svh_array *map, const char *what) {
    if (map->shape[0] != vol->shape[0] || map->shape[1] != vol->shape[1])
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "%s must have shape (%lld,%lld)", what, (long long)vol->shape[0],
                    (long long)vol->shape[1]);
    return SVH_OK;
}

extern "C" {

int svh_extract_selected_index(svh_context *ctx, int strategy, const svh_array *cv, svh_array *idx) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, cv, "cv", SVH_F32, 3, 3));
    SVH_TRY(validate(ctx, idx, "idx", SVH_I32, 2, 2));
    if (strategy != SVH_COST && strategy != SVH_SCORE) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad strategy");
    SVH_TRY(same_map_shape(ctx, cv, idx, "idx"));
    if (cv->shape[2] < 1) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "cost volume has no disparity");
    Scratch scr(ctx);
    void *dcv;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *cv, &dcv));
    SVH_TRY(stage_out(ctx, scr, *idx, &os));
    SVH_TRY(dev_extract_index(ctx, strategy, (const float *)dcv, cv->shape[0] * cv->shape[1], (int)cv->shape[2], (int32_t *)os.dptr,
                              nullptr, 0, 0));
    return finish_out(ctx, os);
}

int svh_selected_index_to_disp(svh_context *ctx, int disp_direction, const svh_array *idx, int32_t disp_offset, svh_array *disp) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, idx, "idx", SVH_I32, 2, 2));
    SVH_TRY(validate(ctx, disp, "disp", SVH_I32, 2, 2));
    SVH_TRY(same_map_shape(ctx, idx, disp, "disp"));
    Scratch scr(ctx);
    void *di;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *idx, &di));
    SVH_TRY(stage_out(ctx, scr, *disp, &os));
    SVH_TRY(dev_index_to_disp(ctx, disp_direction, (const int32_t *)di, num_elements(*idx), disp_offset, (int32_t *)os.dptr));
    return finish_out(ctx, os);
}

int svh_selected_cost(svh_context *ctx, const svh_array *cv, const svh_array *idx, svh_array *cost) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, cv, "cv", SVH_F32, 3, 3));
    SVH_TRY(validate(ctx, idx, "idx", SVH_I32, 2, 2));
    SVH_TRY(validate(ctx, cost, "cost", SVH_F32, 2, 2));
    SVH_TRY(same_map_shape(ctx, cv, idx, "idx"));
    SVH_TRY(same_map_shape(ctx, cv, cost, "cost"));
    Scratch scr(ctx);
    void *dcv, *di;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *cv, &dcv));
    SVH_TRY(stage_in(ctx, scr, *idx, &di));
    SVH_TRY(stage_out(ctx, scr, *cost, &os));
    SVH_TRY(dev_selected_cost(ctx, (const float *)dcv, (const int32_t *)di, num_elements(*idx), (int)cv->shape[2], (float *)os.dptr));
    return finish_out(ctx, os);
}

int svh_truncated_cost_volume(svh_context *ctx, int tcv_direction, int disp_direction, const svh_array *cv, const svh_array *idx,
                              int h_radius, int v_radius, int cost_vol_radius, svh_array *tcv) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, cv, "cv", SVH_F32, 3, 3));
    SVH_TRY(validate(ctx, idx, "idx", SVH_I32, 2, 2));
    SVH_TRY(validate(ctx, tcv, "tcv", SVH_F32, 3, 3));
    if (tcv_direction < SVH_TCV_SAME || tcv_direction > SVH_TCV_BOTH) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad truncation direction");
    if (h_radius < 0 || v_radius < 0 || cost_vol_radius < 0 || h_radius > 255 || v_radius > 255 || cost_vol_radius > 255)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "radii must be in [0,255] (uint8_t in the reference)");
    SVH_TRY(same_map_shape(ctx, cv, idx, "idx"));
    int T = tcv_direction == SVH_TCV_BOTH ? 4 * cost_vol_radius + 1 : 2 * cost_vol_radius + 1;
    if (tcv->shape[0] != cv->shape[0] || tcv->shape[1] != cv->shape[1] || tcv->shape[2] != T)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "tcv must have shape (%lld,%lld,%d)", (long long)cv->shape[0], (long long)cv->shape[1], T);
    Scratch scr(ctx);
    void *dcv, *di;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *cv, &dcv));
    SVH_TRY(stage_in(ctx, scr, *idx, &di));
    SVH_TRY(stage_out(ctx, scr, *tcv, &os));
    SVH_TRY(dev_truncated_cv(ctx, tcv_direction, disp_direction, (const float *)dcv, (const int32_t *)di, (int)cv->shape[0],
                             (int)cv->shape[1], (int)cv->shape[2], h_radius, v_radius, cost_vol_radius, (float *)os.dptr));
    return finish_out(ctx, os);
}

int svh_refine_disp_cost_interpolation(svh_context *ctx, int interp_kernel, const svh_array *tcv, const svh_array *raw,
                                       svh_array *refined) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, tcv, "tcv", SVH_F32, 3, 3));
    SVH_TRY(validate(ctx, raw, "raw", SVH_I32, 2, 2));
    SVH_TRY(validate(ctx, refined, "refined", SVH_F32, 2, 2));
    if (interp_kernel < SVH_EQUIANGULAR || interp_kernel > SVH_GAUSSIAN) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad interpolation kernel");
    int T = (int)tcv->shape[2];
    int r = (T - 1) / 2;
    if (r < 1 || 2 * r + 1 != T) return fail(ctx, SVH_EMPTY_RESULT, "truncated volume depth must be 2r+1, r >= 1"); // :141-143
    SVH_TRY(same_map_shape(ctx, raw, refined, "refined"));
    if (tcv->shape[0] != raw->shape[0] || tcv->shape[1] != raw->shape[1])
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "tcv and raw disagree on the image size");
    Scratch scr(ctx);
    void *dt, *dr;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *tcv, &dt));
    SVH_TRY(stage_in(ctx, scr, *raw, &dr));
    SVH_TRY(stage_out(ctx, scr, *refined, &os));
    SVH_TRY(dev_refine(ctx, interp_kernel, (const float *)dt, (const int32_t *)dr, num_elements(*raw), T, (float *)os.dptr));
    return finish_out(ctx, os);
}

int svh_keys_to_index(svh_context *ctx, int strategy, const svh_array *keys, int32_t disp_count, svh_array *idx) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, keys, "keys", SVH_U64, 2, 2));
    SVH_TRY(validate(ctx, idx, "idx", SVH_I32, 2, 2));
    SVH_TRY(same_map_shape(ctx, keys, idx, "idx"));
    if (strategy != SVH_COST && strategy != SVH_SCORE) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad strategy");
    Scratch scr(ctx);
    void *dk;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *keys, &dk));
    SVH_TRY(stage_out(ctx, scr, *idx, &os));
    SVH_TRY(dev_keys_to_index(ctx, strategy, (const unsigned long long *)dk, num_elements(*keys), disp_count, (int32_t *)os.dptr));
    return finish_out(ctx, os);
}

} // extern "C"


// ---- 2-D disparity volumes ----------------------------------------------------------
extern "C" int svh_extract_selected_2d_index(svh_context *ctx, int strategy, const svh_array *cv, svh_array *idx) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, cv, "cv", SVH_F32, 4, 4));
    SVH_TRY(validate(ctx, idx, "idx", SVH_I32, 3, 3));
    if (strategy != SVH_COST && strategy != SVH_SCORE) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad strategy");
    if (idx->shape[0] != cv->shape[0] || idx->shape[1] != cv->shape[1] || idx->shape[2] != 2)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "idx must have shape (H, W, 2)");
    const int64_t npx = cv->shape[0] * cv->shape[1];
    const int64_t D = cv->shape[2] * cv->shape[3];
    if (D < 1 || D > (1 << 30)) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad search range");
    Scratch scr(ctx);
    void *dcv;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *cv, &dcv));
    SVH_TRY(stage_out(ctx, scr, *idx, &os));
    int32_t *flat = scr.get_n<int32_t>((size_t)npx);
    if (!flat) return SVH_ERR_OUT_OF_MEMORY;
    SVH_TRY(dev_extract_index(ctx, strategy, (const float *)dcv, npx, (int)D, flat, nullptr, 0, 0));
    if (npx) {
        SVH_LAUNCH(ctx, "split_index", split_index_kernel, grid_for(npx, 256, 8192), 256, 0, flat, npx, (int)cv->shape[3], (int32_t *)os.dptr);
        SVH_CHECK_LAUNCH(ctx);
    }
    return finish_out(ctx, os);
}

extern "C" int svh_selected_2d_index_to_disp(svh_context *ctx, const svh_array *idx, int32_t lower0, int32_t lower1, svh_array *disp) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, idx, "idx", SVH_I32, 3, 3));
    SVH_TRY(validate(ctx, disp, "disp", SVH_I32, 3, 3));
    if (idx->shape[2] != 2 || disp->shape[0] != idx->shape[0] || disp->shape[1] != idx->shape[1] || disp->shape[2] != 2)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "idx and disp must have shape (H, W, 2)");
    const int64_t npx = idx->shape[0] * idx->shape[1];
    Scratch scr(ctx);
    void *di;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *idx, &di));
    SVH_TRY(stage_out(ctx, scr, *disp, &os));
    if (npx) {
        SVH_LAUNCH(ctx, "index_2d_to_disp", index_2d_to_disp_kernel, grid_for(npx, 256, 8192), 256, 0, (const int32_t *)di, npx, lower0, lower1,
                   (int32_t *)os.dptr);
        SVH_CHECK_LAUNCH(ctx);
    }
    return finish_out(ctx, os);
}

extern "C" int svh_truncated_bidirectional_cost_volume(svh_context *ctx, const svh_array *cv, const svh_array *idx, int radius0, int radius1,
                                                       svh_array *tcv) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, cv, "cv", SVH_F32, 4, 4));
    SVH_TRY(validate(ctx, idx, "idx", SVH_I32, 3, 3));
    SVH_TRY(validate(ctx, tcv, "tcv", SVH_F32, 4, 4));
    if (radius0 < 1 || radius1 < 1 || radius0 > 255 || radius1 > 255) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "radii must be in [1,255]");
    if (idx->shape[0] != cv->shape[0] || idx->shape[1] != cv->shape[1] || idx->shape[2] != 2)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "idx must have shape (H, W, 2)");
    if (tcv->shape[0] != cv->shape[0] || tcv->shape[1] != cv->shape[1] || tcv->shape[2] != 2 * radius0 + 1 || tcv->shape[3] != 2 * radius1 + 1)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "tcv must have shape (H, W, %d, %d)", 2 * radius0 + 1, 2 * radius1 + 1);
    const int64_t npx = cv->shape[0] * cv->shape[1];
    Scratch scr(ctx);
    void *dcv, *di;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *cv, &dcv));
    SVH_TRY(stage_in(ctx, scr, *idx, &di));
    SVH_TRY(stage_out(ctx, scr, *tcv, &os));
    const int64_t n = npx * (2 * radius0 + 1) * (2 * radius1 + 1);
    if (n) {
        SVH_LAUNCH(ctx, "truncated_bidirectional", truncated_bidirectional_kernel, grid_for(n, 256, 16384), 256, 0, (const float *)dcv,
                   (const int32_t *)di, npx, (int)cv->shape[2], (int)cv->shape[3], radius0, radius1, (float *)os.dptr);
        SVH_CHECK_LAUNCH(ctx);
    }
    return finish_out(ctx, os);
}

// ---- 2-D cost-based refinement -------------------------------------------------------
static int refine_2d_common(svh_context *ctx, const svh_array *tcv, const svh_array *raw, svh_array *refined, int *T0, int *T1) {
    SVH_TRY(validate(ctx, tcv, "tcv", SVH_F32, 4, 4));
    SVH_TRY(validate(ctx, raw, "raw", SVH_I32, 3, 3));
    SVH_TRY(validate(ctx, refined, "refined", SVH_F32, 3, 3));
    *T0 = (int)tcv->shape[2];
    *T1 = (int)tcv->shape[3];
    const int r0 = (*T0 - 1) / 2, r1 = (*T1 - 1) / 2;
    if (r0 < 1 || r1 < 1 || 2 * r0 + 1 != *T0 || 2 * r1 + 1 != *T1) // cost_based_refinement.h:180-182, :393-395
        return fail(ctx, SVH_EMPTY_RESULT, "truncated volume must be (H, W, 2r0+1, 2r1+1) with r0, r1 >= 1");
    if (raw->shape[2] != 2 || raw->shape[0] != tcv->shape[0] || raw->shape[1] != tcv->shape[1])
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "raw must have shape (H, W, 2)");
    if (refined->shape[0] != raw->shape[0] || refined->shape[1] != raw->shape[1] || refined->shape[2] != 2)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "refined must have shape (H, W, 2)");
    return SVH_OK;
}

extern "C" int svh_refine_disp_2d_cost_interpolation(svh_context *ctx, int interp_kernel, int isotropy, const svh_array *tcv, const svh_array *raw,
                                                     svh_array *refined) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    int T0, T1;
    SVH_TRY(refine_2d_common(ctx, tcv, raw, refined, &T0, &T1));
    if (interp_kernel < SVH_EQUIANGULAR || interp_kernel > SVH_GAUSSIAN) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad interpolation kernel");
    if (isotropy != SVH_ISOTROPIC && isotropy != SVH_ANISOTROPIC) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad isotropy hypothesis");
    const int64_t H = tcv->shape[0], W = tcv->shape[1], npx = H * W;
    Scratch scr(ctx);
    void *dt, *dr;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *tcv, &dt));
    SVH_TRY(stage_in(ctx, scr, *raw, &dr));
    SVH_TRY(stage_out(ctx, scr, *refined, &os));
    if (npx) {
        const int64_t centre = (H / 2) * W + W / 2; // cv_shape[0]/2, cv_shape[1]/2 (:186-190)
        const int grid = grid_for(npx, 256, 8192);
        if (isotropy == SVH_ANISOTROPIC)
            SVH_LAUNCH(ctx, "refine_disp_2d", refine_2d_kernel<true>, grid, 256, 0, interp_kernel, (const float *)dt, (const int32_t *)dr, npx, centre, T0,
                       T1, (float *)os.dptr);
        else
            SVH_LAUNCH(ctx, "refine_disp_2d", refine_2d_kernel<false>, grid, 256, 0, interp_kernel, (const float *)dt, (const int32_t *)dr, npx, centre, T0,
                       T1, (float *)os.dptr);
        SVH_CHECK_LAUNCH(ctx);
    }
    return finish_out(ctx, os);
}

extern "C" int svh_refine_disp_2d_cost_patch_interpolation(svh_context *ctx, int interp_kernel, const svh_array *tcv, const svh_array *raw,
                                                           svh_array *refined) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    int T0, T1;
    SVH_TRY(refine_2d_common(ctx, tcv, raw, refined, &T0, &T1));
    if (interp_kernel != SVH_PARABOLA && interp_kernel != SVH_GAUSSIAN) // static_assert, cost_based_refinement.h:83
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "patch refinement supports the Parabola and Gaussian kernels only");
    const int64_t npx = tcv->shape[0] * tcv->shape[1];
    Scratch scr(ctx);
    void *dt, *dr;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *tcv, &dt));
    SVH_TRY(stage_in(ctx, scr, *raw, &dr));
    SVH_TRY(stage_out(ctx, scr, *refined, &os));
    if (npx) {
        SVH_LAUNCH(ctx, "refine_disp_2d_patch", refine_2d_patch_kernel, grid_for(npx, 256, 8192), 256, 0, interp_kernel, (const float *)dt,
                   (const int32_t *)dr, npx, T0, T1, (float *)os.dptr);
        SVH_CHECK_LAUNCH(ctx);
    }
    return finish_out(ctx, os);
}

// Cost-volume construction (A4-A8): aggregateCost over processed feature vectors.
//
// CV(i,j,d) = cmp(src(i,j,:), tgt(i, j + sign*(disp_lower+d), :)), with an all-zero target vector where the
// target column falls outside the image (correlation/cross_correlations.h:235, :294).  The processed feature
// (zero-mean / normalised, cross_correlations.h:416-594) is evaluated on the fly from per-pixel mean and
// norm maps, with the same float operations and the same channel order as the reference, so no H x W x F
// volume is ever written.  Volumes are stored (row, col, disparity) with the disparity fastest: one pixel's D
// costs are contiguous, which is what the per-line SGM kernels and the winner scan want.
#include "svh_internal.h"
#include <climits>

#include "svh_compare.h"
#include "svh_sgm_lines.h"

namespace svh {

// one thread per voxel, disparity fastest (coalesced stores; the source vector is wave-uniform for D >= 64)
template <class A, int CMP, bool ZM, bool NORM>
__global__ void cost_volume_kernel(A src, A tgt, const float *__restrict__ mean_s, const float *__restrict__ norm_s,
                                   const float *__restrict__ mean_t, const float *__restrict__ norm_t, int H, int Ws, int Wt,
                                   int D, int sign, int disp_lower, int row_off, int64_t px_stride, int64_t out_off,
                                   float *__restrict__ cv) {
    const int64_t n = (int64_t)H * Ws * D;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        int d = (int)(e % D);
        int64_t p = e / D;
        int j = (int)(p % Ws), i = (int)(p / Ws);
        int jt = j + sign * (disp_lower + d);
        const int it = i + row_off;
        cv[p * px_stride + out_off + d] = compare_features<A, CMP, ZM, NORM>(src, tgt, mean_s, norm_s, mean_t, norm_t, H, Ws, Wt, i, j, it, jt);
    }
}

// Hamming volume from compact census words (only the fully written words are kept, the never-written trailing
// word is 0 on both sides and contributes nothing).  hammingDistance, matching_costs.h:236-263: the sum is a
// uint16_t returned as float.  Target words were rounded through float when they were produced (rule E2); a
// target column outside the image is the zero vector, so the cost is popcount(source).
__global__ void hamming_volume_kernel(const uint32_t *__restrict__ sw, const uint32_t *__restrict__ tw, int nWw, int H, int Ws,
                                      int Wt, int D, int sign, int disp_lower, int row_off, int64_t px_stride, int64_t out_off,
                                      float *__restrict__ cv) {
    const int64_t n = (int64_t)H * Ws * D;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        int d = (int)(e % D);
        int64_t p = e / D;
        int j = (int)(p % Ws), i = (int)(p / Ws);
        int jt = j + sign * (disp_lower + d);
        const int it = i + row_off;
        bool t_in = jt >= 0 && jt < Wt && it >= 0 && it < H;
        const uint32_t *s = sw + p * nWw;
        const uint32_t *t = tw + ((int64_t)(t_in ? it : 0) * Wt + (t_in ? jt : 0)) * nWw;
        uint32_t score = 0;
        for (int w = 0; w < nWw; w++) score += __popc(s[w] ^ (t_in ? t[w] : 0u));
        cv[p * px_stride + out_off + d] = (float)(uint16_t)score;
    }
}

// LDS-tiled form: a block owns HV_TP consecutive pixels of a row and stages their HV_TP + D - 1 target records once
// (mirrored for LeftToRight so that the record index grows with d); a wave then writes one pixel's D costs at a time,
// lane l taking d = l, l + 64, ...: 64 consecutive LDS records in, one coalesced 256-byte store out.  HBM-write bound
// (4 B/voxel, the API's float volume).
constexpr int HV_TP = 64;

template <int NW>
__global__ void __launch_bounds__(256) hamming_volume_tiled_kernel(const uint32_t *__restrict__ sw, const uint32_t *__restrict__ tw, int H,
                                                                   int Ws, int Wt, int D, int sign, int disp_lower, int row_off, int64_t px_stride,
                                                                   int64_t out_off, float *__restrict__ cv, float2 *__restrict__ minima,
                                                                   int32_t *__restrict__ win_idx, int store) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int i = blockIdx.y, j0 = blockIdx.x * HV_TP;
    const int n_rec = HV_TP + D - 1;
    const int it = i + row_off;
    const bool row_in = it >= 0 && it < H; // a target row outside the image is the zero vector
    const uint32_t *trow = tw + (int64_t)(row_in ? it : 0) * Wt * NW;
    uint32_t *lsrc = lds + n_rec * NW; // the tile's source records: no global load inside the store loop
    for (int e = threadIdx.x; e < n_rec * NW; e += blockDim.x) {
        const int y = e / NW, w = e - y * NW;
        const int jt = sign > 0 ? j0 + disp_lower + y : j0 + (HV_TP - 1) - disp_lower - y;
        lds[e] = (row_in && jt >= 0 && jt < Wt) ? trow[(int64_t)jt * NW + w] : 0u;
    }
    const int n_src = min(HV_TP, Ws - j0) * NW;
    for (int e = threadIdx.x; e < n_src; e += blockDim.x) lsrc[e] = sw[((int64_t)i * Ws + j0) * NW + e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int u = wave; u < HV_TP && j0 + u < Ws; u += 4) {
        const int64_t p = (int64_t)i * Ws + j0 + u;
        uint32_t s[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) s[w] = lsrc[u * NW + w];
        const uint32_t *base = lds + (sign > 0 ? u : HV_TP - 1 - u) * NW;
        float *out = cv + p * px_stride + out_off;
        if (win_idx) { // (uniform) CostReduce mode 1: extractSelectedIndex of the pixel's costs while a wave holds them -- the smallest distance, ties to
            // the larger index (a distance is never NaN) -- and, when nobody wants the volume itself (store == 0), nothing else is written
            int best = INT_MAX, bd = 0;
            for (int d = lane; d < D; d += 64) {
                uint32_t score = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) score += __popc(s[w] ^ base[d * NW + w]);
                if (store) out[d] = (float)score;
                if ((int)score <= best) { // (d grows: a later equal distance takes over)
                    best = (int)score;
                    bd = d;
                }
            }
            int m = best;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) m = min(m, __shfl_xor(m, off));
            int key = best == m ? bd + 1 : 0; // (a lane without a cost holds INT_MAX: D >= 1 gives lane 0 one)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) key = max(key, __shfl_xor(key, off));
            if (lane == 0) win_idx[p] = key - 1;
            continue;
        }
        if (!minima) { // (uniform)
            for (int d = lane; d < D; d += 64) {
                uint32_t score = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) score += __popc(s[w] ^ base[d * NW + w]);
                out[d] = (float)score;
            }
            continue;
        }
        // also the hint of svh_unfold_cost_volume_minima: the smallest cost among the disparities that do not / do pay Pout in a later
        // sgmCostVolume (sgm.h:287-289: j + d >= W), what that call would otherwise read the whole volume for.  The boundary is one
        // disparity per pixel: whole rounds of 64 disparities fall on one side of it (wave-uniform branches), at most one straddles.
        float m0 = INFINITY, m1 = INFINITY;
        const int first_paying = Ws - (j0 + u);
        for (int d0 = 0; d0 < D; d0 += 64) {
            const int d = d0 + lane;
            uint32_t score = 0;
            if (d < D) {
#pragma unroll
                for (int w = 0; w < NW; w++) score += __popc(s[w] ^ base[d * NW + w]);
                out[d] = (float)score;
            }
            const float c = d < D ? (float)score : INFINITY;
            if (d0 + 64 <= first_paying) m0 = fminf(m0, c);
            else if (d0 >= first_paying) m1 = fminf(m1, c);
            else {
                m0 = fminf(m0, d < first_paying ? c : INFINITY);
                m1 = fminf(m1, d < first_paying ? INFINITY : c);
            }
        }
        m0 = wave_min(m0);
        m1 = wave_min(m1);
        if (lane == 0) minima[p] = make_float2(m0, m1);
    }
}

template <int NW>
static void launch_hamming_tiled(svh_context *ctx, const CostVolumeArgs &a, const uint32_t *sw, const uint32_t *tw, int sign, float *cv) {
    dim3 grid(ceil_div(a.Ws, HV_TP), a.H);
    const size_t shmem = (size_t)NW * (2 * HV_TP + a.D - 1) * sizeof(uint32_t);
    // CostReduce mode 1 (the winner): whole 1-D volumes only (one launch holds a pixel's whole range)
    const bool winner = a.reduce && a.reduce->mode == 1 && !a.reduce->score && a.reduce->idx && !a.reduce->disp && a.n_dh == 1 && a.out_off == 0 &&
                        (a.reduce->store ? cv != nullptr : true);
    SVH_LAUNCH(ctx, "hamming_volume", hamming_volume_tiled_kernel<NW>, grid, 256, shmem, sw, tw, a.H, a.Ws, a.Wt, a.D, sign, a.disp_lower, a.tgt_row_off,
               a.px_stride(), a.out_off, cv, winner ? nullptr : reinterpret_cast<float2 *>(a.minima), winner ? a.reduce->idx : nullptr,
               winner && !a.reduce->store ? 0 : 1);
    if (winner) a.reduce->done = true;
}

// 2-D search ranges (aggregateCost(searchOffset<2>), cross_correlations.h:310-374): every vertical offset's target records are
// staged together and the lanes run over the flattened (dh, dw) axis, so a pixel's Dh x Dw costs leave as one contiguous run
// with every lane busy (one launch per vertical offset writes 4 Dw-byte pieces into 4 Dh Dw-byte pixels and idles the lanes
// beyond Dw).
template <int NW>
__global__ void __launch_bounds__(256) hamming_volume_2d_kernel(const uint32_t *__restrict__ sw, const uint32_t *__restrict__ tw, int H, int Ws, int Wt,
                                                                int Dh, int Dw, int lower0, int lower1, float *__restrict__ cv) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int i = blockIdx.y, j0 = blockIdx.x * HV_TP;
    const int n_rec = HV_TP + Dw - 1, DD = Dh * Dw;
    uint32_t *lsrc = lds + (size_t)Dh * n_rec * NW;
    uint16_t *lut = reinterpret_cast<uint16_t *>(lsrc + HV_TP * NW); // q = dh Dw + dw -> record offset dh n_rec + dw
    for (int e = threadIdx.x; e < Dh * n_rec * NW; e += blockDim.x) {
        const int w = e % NW, y = (e / NW) % n_rec, dh = e / (NW * n_rec);
        const int it = i + lower0 + dh, jt = j0 + lower1 + y;
        lds[e] = (it >= 0 && it < H && jt >= 0 && jt < Wt) ? tw[((int64_t)it * Wt + jt) * NW + w] : 0u; // zero vector outside the image (:359)
    }
    const int n_src = min(HV_TP, Ws - j0) * NW;
    for (int e = threadIdx.x; e < n_src; e += blockDim.x) lsrc[e] = sw[((int64_t)i * Ws + j0) * NW + e];
    for (int q = threadIdx.x; q < DD; q += blockDim.x) lut[q] = (uint16_t)((q / Dw) * n_rec + q % Dw);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int u = wave; u < HV_TP && j0 + u < Ws; u += 4) {
        uint32_t s[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) s[w] = lsrc[u * NW + w];
        float *out = cv + ((int64_t)i * Ws + j0 + u) * DD;
        for (int q = lane; q < DD; q += 64) {
            const uint32_t *rec = lds + ((int)lut[q] + u) * NW;
            uint32_t score = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) score += __popc(s[w] ^ rec[w]);
            out[q] = (float)score;
        }
    }
}

// true when the 2-D kernel took the job (compact words of 1..4 words, tile within the LDS budget, offsets addressable by the LUT)
bool dev_hamming_volume_2d(svh_context *ctx, const uint32_t *sw, const uint32_t *tw, int nWw, int H, int Ws, int Wt, int Dh, int Dw, int lower0, int lower1,
                           float *cv) {
    const size_t shmem = ((size_t)Dh * (HV_TP + Dw - 1) + HV_TP) * nWw * sizeof(uint32_t) + (size_t)Dh * Dw * sizeof(uint16_t);
    if (nWw < 1 || nWw > 4 || shmem > 60 * 1024 || (size_t)Dh * (HV_TP + Dw - 1) > 65535) return false;
    if ((int64_t)H * Ws == 0) return true;
    dim3 grid(ceil_div(Ws, HV_TP), H);
    switch (nWw) {
    case 1: SVH_LAUNCH(ctx, "hamming_volume", hamming_volume_2d_kernel<1>, grid, 256, shmem, sw, tw, H, Ws, Wt, Dh, Dw, lower0, lower1, cv); break;
    case 2: SVH_LAUNCH(ctx, "hamming_volume", hamming_volume_2d_kernel<2>, grid, 256, shmem, sw, tw, H, Ws, Wt, Dh, Dw, lower0, lower1, cv); break;
    case 3: SVH_LAUNCH(ctx, "hamming_volume", hamming_volume_2d_kernel<3>, grid, 256, shmem, sw, tw, H, Ws, Wt, Dh, Dw, lower0, lower1, cv); break;
    default: SVH_LAUNCH(ctx, "hamming_volume", hamming_volume_2d_kernel<4>, grid, 256, shmem, sw, tw, H, Ws, Wt, Dh, Dw, lower0, lower1, cv); break;
    }
    return true;
}

template <class A, int CMP>
static int launch_cv(svh_context *ctx, bool zm, bool nrm, A src, A tgt, const float *ms, const float *ns, const float *mt,
                     const float *nt, const CostVolumeArgs &a, float *cv) {
    int64_t n = (int64_t)a.H * a.Ws * a.D;
    int grid = grid_for(n, 256, 65536);
    int sign = a.sign();
#define SVH_CV_LAUNCH(ZM, NORM)                                                                                          \
    SVH_LAUNCH(ctx, "cost_volume", (cost_volume_kernel<A, CMP, ZM, NORM>), grid, 256, 0, src, tgt, ms, ns, mt, nt, a.H, a.Ws, \
               a.Wt, a.D, sign, a.disp_lower, a.tgt_row_off, a.px_stride(), a.out_off, cv)
    if (zm && nrm) SVH_CV_LAUNCH(true, true);
    else if (zm) SVH_CV_LAUNCH(true, false);
    else if (nrm) SVH_CV_LAUNCH(false, true);
    else SVH_CV_LAUNCH(false, false);
#undef SVH_CV_LAUNCH
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

// ---- feature volumes (featureVolume2CostVolume on unfolded / compressed features): processed once, compared from LDS ------------------
// The per-voxel kernel applies getFeatureVolumeForMatchFunc on the fly: for a normalised function every voxel subtracts and DIVIDES all F
// features of both vectors again (the true division of cross_correlations.h:540: ten instructions) -- the reference's compressor benchmark
// row (17 superpixel features, ZNCC, 1080p x 320) took 23.4 ms.  Here the two volumes are processed once (the same subtraction and
// division per feature: the values the reference's normalised volumes hold) and a block compares HV_TP source pixels of a row with their
// HV_TP + D - 1 target records from LDS: lanes over the disparities, the features in the reference's order, the same multiply and add per
// term as compare_features: same bits.  Records are padded to an odd stride (lanes read records one apart: no bank conflict).
// statistics and processing in one pass over a feature volume: a block stages the F features of 256 pixels in LDS (one flat, coalesced
// copy; pixel stride F | 1), a thread walks its pixel's features as stats_kernel does -- channelsMean (sequential sum, then * float(1. / F)),
// channelsNorm / channelsZeroMeanNorm (sqrtf of the sequential sum of squares): same operations, same order -- replaces them by the
// processed values, and the block stores them flat again.  (A thread per pixel reading its 17 floats straight from global memory, then a
// second kernel: 0.34 + 0.11 ms for two 1080p x 17 volumes.)
template <bool ZM, bool NORM>
__global__ void __launch_bounds__(256) process_features_kernel(const float *__restrict__ f, int64_t npx, int F, float *__restrict__ out, int ppb) {
    extern __shared__ float pf_lds[];
    const int FS = F | 1;
    const int64_t p0 = (int64_t)blockIdx.x * ppb; // ppb <= 256 pixels per block: what the LDS budget holds of vectors this long
    const int n_px = (int)(npx - p0 < ppb ? npx - p0 : ppb), n = n_px * F;
    const float *src = f + p0 * F;
    float *dst = out + p0 * F;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int pl = e / F;
        pf_lds[pl * FS + (e - pl * F)] = src[e];
    }
    __syncthreads();
    if ((int)threadIdx.x < n_px) {
        float *v = pf_lds + threadIdx.x * FS;
        const float scale = (float)(1. / (double)(float)F);
        float m = 0.0f, nr = 1.0f;
        if (ZM) {
            for (int c = 0; c < F; c++) m += v[c];
            m *= scale;
        }
        if (NORM) {
            float acc = 0.0f;
            for (int c = 0; c < F; c++) {
                const float tmp = v[c] - m; // m == 0 without zero-mean: x - 0 is exact
                acc += tmp * tmp;
            }
            nr = sqrtf(acc);
        }
        for (int c = 0; c < F; c++) {
            float x = v[c];
            if (ZM) x = x - m;
            if (NORM) x = x / nr;
            v[c] = x;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += 256) {
        const int pl = e / F;
        dst[e] = pf_lds[pl * FS + (e - pl * F)];
    }
}

// the HV_TP + D - 1 target records a block looks at (record y: target column j0 + disp_lower + y, or j0 + HV_TP - 1 - disp_lower - y when the
// disparities run leftwards) into LDS with stride FS, zero vectors outside the image.  Eight loads in flight per thread and no division in
// the loop (e = y F + c is advanced by 256 at a time): one load, one wait, one store per element with an integer division in between made
// this copy -- 25 elements per thread at 1080p x 320 x 17 -- as long as the comparisons behind it.
__device__ __forceinline__ void stage_target_records(float *flds, const float *__restrict__ trow, bool row_in, int n_rec, int F, int FS, int sign, int jt0, int Wt) {
    const int n = n_rec * F, dq = 256 / F, dr = 256 - dq * F;
    int y = (int)threadIdx.x / F, c = (int)threadIdx.x - y * F;
    for (int e0 = threadIdx.x; e0 < n; e0 += 8 * 256) {
        float got[8];
        int at[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int jt = jt0 + sign * y;
            const bool ok = e0 + 256 * k < n && row_in && jt >= 0 && jt < Wt;
            const float x = trow[(int64_t)min(max(jt, 0), Wt - 1) * F + c];
            got[k] = ok ? x : 0.0f;
            at[k] = e0 + 256 * k < n ? y * FS + c : -1;
            y += dq;
            c += dr;
            if (c >= F) {
                c -= F;
                y++;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (at[k] >= 0) flds[at[k]] = got[k];
    }
}

// FV_Q disparities per lane (d, d + 64, ...): a source feature is read once for all of them
constexpr int FV_Q = 4;
template <int CMP>
__global__ void __launch_bounds__(256) feature_volume_tiled_kernel(const float *__restrict__ ps, const float *__restrict__ pt, int H, int Ws, int Wt, int F,
                                                                   int D, int sign, int disp_lower, int row_off, int64_t px_stride, int64_t out_off,
                                                                   float *__restrict__ cv) {
    extern __shared__ __attribute__((aligned(16))) float flds[];
    const int i = blockIdx.y, j0 = blockIdx.x * HV_TP;
    const int n_rec = HV_TP + D - 1, FS = F | 1; // record stride
    const int it = i + row_off;
    const bool row_in = it >= 0 && it < H; // a target outside the image is the zero vector (cross_correlations.h:235)
    const float *trow = pt + (int64_t)(row_in ? it : 0) * Wt * F;
    float *lsrc = flds + n_rec * FS;
    stage_target_records(flds, trow, row_in, n_rec, F, FS, sign > 0 ? 1 : -1, sign > 0 ? j0 + disp_lower : j0 + (HV_TP - 1) - disp_lower, Wt);
    const int n_src = min(HV_TP, Ws - j0) * F;
    for (int e = threadIdx.x; e < n_src; e += blockDim.x) lsrc[e] = ps[((int64_t)i * Ws + j0) * F + e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int u = wave; u < HV_TP && j0 + u < Ws; u += 4) {
        const float *s = lsrc + u * F;
        const float *base = flds + (sign > 0 ? u : HV_TP - 1 - u) * FS;
        float *out = cv + ((int64_t)i * Ws + j0 + u) * px_stride + out_off;
        for (int d0 = 0; d0 < D; d0 += 64 * FV_Q) {
            const float *t[FV_Q];
            float score[FV_Q];
#pragma unroll
            for (int q = 0; q < FV_Q; q++) {
                t[q] = base + min(d0 + 64 * q + lane, D - 1) * FS; // (past the range: the last record again, not stored)
                score[q] = 0.0f;
            }
#pragma unroll 2
            for (int c = 0; c < F; c++) {
                const float sv = s[c];
#pragma unroll
                for (int q = 0; q < FV_Q; q++) {
                    const float tv = t[q][c];
                    if (CMP == CMP_DOT) {
                        score[q] += sv * tv;
                    } else if (CMP == CMP_SSD) {
                        const float tmp = sv - tv;
                        score[q] += tmp * tmp;
                    } else {
                        score[q] += fabsf(sv - tv);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < FV_Q; q++)
                if (d0 + 64 * q + lane < D) out[d0 + 64 * q + lane] = score[q];
        }
    }
}

// Round 5: short feature vectors (up to FMAX features: the 17 superpixel means of the reference's GrPix17 compressors, 3x3 / 5x5 unfolds)
// with the TARGET record in registers.  The kernel above reads every target feature of every voxel from LDS -- 4 bytes per multiply-add:
// 45 GB at 1080p x 320 x 17, 0.57 ms of LDS bandwidth under 0.57 ms of vector issue, 1.8 ms measured.  Here a lane owns a target record
// (its F features in registers) and walks the source pixels of the tile that look at it -- d = record - pixel --, two at a time: a source
// feature is one broadcast LDS read for 64 voxels, and the multiply and the add of two voxels are one packed instruction each.  Same
// products added in the same order (c = 0, 1, 2, ...): same bits.
static inline int func_cmp(int func) { // the comparison of a float matching function
    switch (func) {
    case SVH_CC: case SVH_NCC: case SVH_ZCC: case SVH_ZNCC: return CMP_DOT;
    case SVH_SSD: case SVH_ZSSD: return CMP_SSD;
    case SVH_SAD: case SVH_ZSAD: return CMP_SAD;
    default: return -1;
    }
}
typedef float fv_f32x2 __attribute__((ext_vector_type(2)));
// EXACT: F == FMAX, known at compile time (17: the GrPix17 compressors; 9, 25: 3x3 and 5x5 unfolds; 27: 3x3 RGB) -- no guard per feature, so the
// source reads of a step are issued together instead of one wait per feature
template <int CMP, int FMAX, bool EXACT>
__global__ void __launch_bounds__(256) feature_volume_records_kernel(const float *__restrict__ ps, const float *__restrict__ pt, int H, int Ws, int Wt, int F_arg,
                                                                     int D, int sign, int disp_lower, int row_off, int64_t px_stride, int64_t out_off,
                                                                     float *__restrict__ cv) {
    extern __shared__ __attribute__((aligned(16))) float flds[];
    const int F = EXACT ? FMAX : F_arg;
    const int i = blockIdx.y, j0 = blockIdx.x * HV_TP;
    const int n_rec = HV_TP + D - 1, FS = F | 1; // record stride (odd: a wave's 64 records fall into different banks)
    const int it = i + row_off;
    const bool row_in = it >= 0 && it < H; // a target outside the image is the zero vector (cross_correlations.h:235)
    const float *trow = pt + (int64_t)(row_in ? it : 0) * Wt * F;
    float *lsrc = flds + ((n_rec * FS + 3) & ~3); // source features, feature-major: feature c of the tile's pixel u at lsrc[c * HV_TP + u] (16-byte rows)
    stage_target_records(flds, trow, row_in, n_rec, F, FS, sign > 0 ? 1 : -1, sign > 0 ? j0 + disp_lower : j0 + (HV_TP - 1) - disp_lower, Wt);
    const int n_px = min(HV_TP, Ws - j0);
    for (int e = threadIdx.x; e < HV_TP * F; e += blockDim.x) {
        const int u = e / F, c = e - u * F;
        lsrc[c * HV_TP + u] = u < n_px ? ps[((int64_t)i * Ws + j0) * F + e] : 0.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int y0 = 0; y0 < n_rec; y0 += 64) { // (block uniform) 64 target records at a time, one per lane
        const int y = y0 + lane;
        float t[FMAX];
#pragma unroll
        for (int c = 0; c < FMAX; c++) t[c] = ((EXACT || c < F) && y < n_rec) ? flds[min(y, n_rec - 1) * FS + c] : 0.0f;
        // the source pixels whose ranges touch these records -- pixel u's disparity 0 is record b(u) = u (sign > 0) or HV_TP - 1 - u --, four
        // at a time: this wave takes the groups wave, wave + 4, ...
        for (int u = 4 * wave; u < n_px; u += 16) {
            const int b_lo = sign > 0 ? u : HV_TP - 4 - u; // the smallest b of the group's four pixels
            if (y0 + 63 < b_lo || y0 >= b_lo + 3 + D) continue; // (wave uniform) none of the four windows touches the chunk
            const float4 *s4 = reinterpret_cast<const float4 *>(lsrc + u);
            fv_f32x2 acc01{0.0f, 0.0f}, acc23{0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < FMAX; c++) {
                if (EXACT || c < F) { // (uniform)
                    const float4 sv = s4[c * (HV_TP / 4)];
                    const fv_f32x2 s01{sv.x, sv.y}, s23{sv.z, sv.w}, tv{t[c], t[c]};
                    if (CMP == CMP_DOT) {
                        acc01 += s01 * tv;
                        acc23 += s23 * tv;
                    } else if (CMP == CMP_SSD) {
                        const fv_f32x2 d01 = s01 - tv, d23 = s23 - tv;
                        acc01 += d01 * d01;
                        acc23 += d23 * d23;
                    } else {
                        const fv_f32x2 d01 = s01 - tv, d23 = s23 - tv;
                        acc01 += fv_f32x2{fabsf(d01.x), fabsf(d01.y)};
                        acc23 += fv_f32x2{fabsf(d23.x), fabsf(d23.y)};
                    }
                }
            }
            const float res[4] = {acc01.x, acc01.y, acc23.x, acc23.y};
            float *o = cv + ((int64_t)i * Ws + j0 + u) * px_stride + out_off;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int d = y - (sign > 0 ? u + k : HV_TP - 1 - u - k);
                if (u + k < n_px && d >= 0 && d < D) o[k * px_stride + d] = res[k];
            }
        }
    }
}

// SVH_ERR_UNSUPPORTED (nothing launched): the records do not fit the LDS budget
static int cost_volume_features_tiled(svh_context *ctx, Scratch &scr, const CostVolumeArgs &a, const float *feat_src, const float *feat_tgt, int F, float *cv) {
    if (!ctx->feature_volume_tiled || a.H > 65535 || F < 1) return SVH_ERR_UNSUPPORTED;
    // the disparity range in chunks when the records of the whole range do not fit (long vectors: 9x9 unfolded features and more)
    const size_t rec = (size_t)(F | 1) * sizeof(float), budget = 60 * 1024;
    const int64_t max_rec = (int64_t)((budget - (size_t)HV_TP * F * sizeof(float)) / rec) - (HV_TP - 1); // disparities whose records fit beside the source vectors
    if ((size_t)HV_TP * F * sizeof(float) >= budget || max_rec < 16) return SVH_ERR_UNSUPPORTED;
    const int chunk = (int)std::min<int64_t>(a.D, max_rec >= 64 ? max_rec / 64 * 64 : max_rec);
    const int ppb = (int)std::min<size_t>(256, budget / rec);
    if (ppb < 32) return SVH_ERR_UNSUPPORTED;
    if ((int64_t)a.H * a.Ws * a.D == 0) return SVH_OK;
    const bool zm = func_zero_mean(a.func), nrm = func_normalized(a.func);
    const float *ps = feat_src, *pt = feat_tgt;
    if (zm || nrm) {
        const int64_t ns_px = (int64_t)a.H * a.Ws, nt_px = (int64_t)a.H * a.Wt;
        float *os = scr.get_n<float>((size_t)ns_px * F), *ot = scr.get_n<float>((size_t)nt_px * F);
        if (!os || !ot) return SVH_ERR_OUT_OF_MEMORY;
        const size_t pshmem = (size_t)ppb * rec;
#define SVH_PROCESS(ZMV, NRMV)                                                                                                                          \
    do {                                                                                                                                                \
        SVH_LAUNCH(ctx, "process_features", (process_features_kernel<ZMV, NRMV>), (int)((ns_px + ppb - 1) / ppb), 256, pshmem, feat_src, ns_px, F, os, ppb); \
        SVH_LAUNCH(ctx, "process_features", (process_features_kernel<ZMV, NRMV>), (int)((nt_px + ppb - 1) / ppb), 256, pshmem, feat_tgt, nt_px, F, ot, ppb); \
    } while (0)
        if (zm && nrm) SVH_PROCESS(true, true);
        else if (zm) SVH_PROCESS(true, false);
        else SVH_PROCESS(false, true);
#undef SVH_PROCESS
        ps = os;
        pt = ot;
    }
    const dim3 grid(ceil_div(a.Ws, HV_TP), a.H);
    const int sign = a.sign();
    for (int d0 = 0; d0 < a.D; d0 += chunk) {
        const int Dc = std::min(chunk, a.D - d0), lower = a.disp_lower + d0;
        const int64_t off = a.out_off + d0;
        if (F <= 32 && ctx->feature_volume_records) { // the target record in registers
            const size_t rshmem = ((((size_t)(HV_TP + Dc - 1) * (F | 1) + 3) & ~(size_t)3) + (size_t)HV_TP * F) * sizeof(float);
            const int cmp = func_cmp(a.func);
            if (rshmem <= budget + 4096 && cmp >= 0) {
#define SVH_FVR_AS(CMPV, FM, EX)                                                                                                                            \
    SVH_LAUNCH(ctx, "feature_volume_tiled", (feature_volume_records_kernel<CMPV, FM, EX>), grid, 256, rshmem, ps, pt, a.H, a.Ws, a.Wt, F, Dc, sign, lower, \
               a.tgt_row_off, a.px_stride(), off, cv)
#define SVH_FVR(CMPV)                                    \
    do {                                                 \
        if (F == 17) SVH_FVR_AS(CMPV, 17, true);         \
        else if (F == 25) SVH_FVR_AS(CMPV, 25, true);    \
        else if (F == 9) SVH_FVR_AS(CMPV, 9, true);      \
        else if (F == 27) SVH_FVR_AS(CMPV, 27, true);    \
        else if (F <= 20) SVH_FVR_AS(CMPV, 20, false);   \
        else SVH_FVR_AS(CMPV, 32, false);                \
    } while (0)
                if (cmp == CMP_DOT) SVH_FVR(CMP_DOT);
                else if (cmp == CMP_SSD) SVH_FVR(CMP_SSD);
                else SVH_FVR(CMP_SAD);
#undef SVH_FVR
#undef SVH_FVR_AS
                continue;
            }
        }
        const size_t shmem = ((size_t)(HV_TP + Dc - 1) * (F | 1) + (size_t)HV_TP * F) * sizeof(float);
        switch (a.func) {
        case SVH_CC: case SVH_NCC: case SVH_ZCC: case SVH_ZNCC:
            SVH_LAUNCH(ctx, "feature_volume_tiled", feature_volume_tiled_kernel<CMP_DOT>, grid, 256, shmem, ps, pt, a.H, a.Ws, a.Wt, F, Dc, sign, lower, a.tgt_row_off,
                       a.px_stride(), off, cv);
            break;
        case SVH_SSD: case SVH_ZSSD:
            SVH_LAUNCH(ctx, "feature_volume_tiled", feature_volume_tiled_kernel<CMP_SSD>, grid, 256, shmem, ps, pt, a.H, a.Ws, a.Wt, F, Dc, sign, lower, a.tgt_row_off,
                       a.px_stride(), off, cv);
            break;
        case SVH_SAD: case SVH_ZSAD:
            SVH_LAUNCH(ctx, "feature_volume_tiled", feature_volume_tiled_kernel<CMP_SAD>, grid, 256, shmem, ps, pt, a.H, a.Ws, a.Wt, F, Dc, sign, lower, a.tgt_row_off,
                       a.px_stride(), off, cv);
            break;
        default: return fail(ctx, SVH_ERR_UNSUPPORTED, "matching function %d is not available on the GPU path", a.func);
        }
    }
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

template <class A>
static int cost_volume_generic(svh_context *ctx, Scratch &scr, const CostVolumeArgs &a, A src, A tgt, float *cv) {
    if ((int64_t)a.H * a.Ws * a.D == 0) return SVH_OK;
    bool zm = func_zero_mean(a.func), nrm = func_normalized(a.func);
    float *ms = nullptr, *ns = nullptr, *mt = nullptr, *nt = nullptr;
    if (zm) {
        ms = scr.get_n<float>((size_t)a.H * a.Ws);
        mt = scr.get_n<float>((size_t)a.H * a.Wt);
        if (!ms || !mt) return SVH_ERR_OUT_OF_MEMORY;
    }
    if (nrm) {
        ns = scr.get_n<float>((size_t)a.H * a.Ws);
        nt = scr.get_n<float>((size_t)a.H * a.Wt);
        if (!ns || !nt) return SVH_ERR_OUT_OF_MEMORY;
    }
    if (zm || nrm) {
        SVH_TRY(launch_stats(ctx, src, a.H, a.Ws, zm, nrm, ms, ns));
        SVH_TRY(launch_stats(ctx, tgt, a.H, a.Wt, zm, nrm, mt, nt));
    }
    switch (a.func) {
    case SVH_CC: case SVH_NCC: case SVH_ZCC: case SVH_ZNCC:
        return launch_cv<A, CMP_DOT>(ctx, zm, nrm, src, tgt, ms, ns, mt, nt, a, cv);
    case SVH_SSD: case SVH_ZSSD:
        return launch_cv<A, CMP_SSD>(ctx, zm, nrm, src, tgt, ms, ns, mt, nt, a, cv);
    case SVH_SAD: case SVH_ZSAD:
        return launch_cv<A, CMP_SAD>(ctx, zm, nrm, src, tgt, ms, ns, mt, nt, a, cv);
    default:
        return fail(ctx, SVH_ERR_UNSUPPORTED, "matching function %d is not available on the GPU path", a.func);
    }
}

int dev_hamming_volume(svh_context *ctx, const CostVolumeArgs &a, const uint32_t *src_words, const uint32_t *tgt_words, int nWw,
                       float *cv) {
    int64_t n = (int64_t)a.H * a.Ws * a.D;
    if (n == 0) return SVH_OK;
    int sign = a.sign();
    if (nWw >= 1 && nWw <= 8 && (size_t)nWw * (2 * HV_TP + a.D - 1) * sizeof(uint32_t) <= 60 * 1024) {
        switch (nWw) {
        case 1: launch_hamming_tiled<1>(ctx, a, src_words, tgt_words, sign, cv); break;
        case 2: launch_hamming_tiled<2>(ctx, a, src_words, tgt_words, sign, cv); break;
        case 3: launch_hamming_tiled<3>(ctx, a, src_words, tgt_words, sign, cv); break;
        case 4: launch_hamming_tiled<4>(ctx, a, src_words, tgt_words, sign, cv); break;
        case 5: launch_hamming_tiled<5>(ctx, a, src_words, tgt_words, sign, cv); break;
        case 6: launch_hamming_tiled<6>(ctx, a, src_words, tgt_words, sign, cv); break;
        case 7: launch_hamming_tiled<7>(ctx, a, src_words, tgt_words, sign, cv); break;
        default: launch_hamming_tiled<8>(ctx, a, src_words, tgt_words, sign, cv); break;
        }
        SVH_CHECK_LAUNCH(ctx);
        if (a.minima_written) *a.minima_written = a.minima != nullptr;
        return SVH_OK;
    }
    if (!cv) return fail(ctx, SVH_ERR_HIP, "internal: the Hamming kernel for these records writes the volume");
    SVH_LAUNCH(ctx, "hamming_volume", hamming_volume_kernel, grid_for(n, 256, 65536), 256, 0, src_words, tgt_words, nWw, a.H, a.Ws,
               a.Wt, a.D, sign, a.disp_lower, a.tgt_row_off, a.px_stride(), a.out_off, cv);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

int dev_cost_volume_from_features(svh_context *ctx, Scratch &scr, const CostVolumeArgs &a, const float *feat_src,
                                  const float *feat_tgt, int F, float *cv) {
    if (func_census(a.func)) {
        int nWw = census_words_written(F);
        uint32_t *sw = scr.get_n<uint32_t>((size_t)a.H * a.Ws * (nWw ? nWw : 1));
        uint32_t *tw = scr.get_n<uint32_t>((size_t)a.H * a.Wt * (nWw ? nWw : 1));
        if (!sw || !tw) return SVH_ERR_OUT_OF_MEMORY;
        SVH_TRY(dev_census_from_features(ctx, feat_src, a.H, a.Ws, F, nWw, false, sw));
        SVH_TRY(dev_census_from_features(ctx, feat_tgt, a.H, a.Wt, F, nWw, true, tw));
        return dev_hamming_volume(ctx, a, sw, tw, nWw, cv);
    }
    const int st = cost_volume_features_tiled(ctx, scr, a, feat_src, feat_tgt, F, cv);
    if (st != SVH_ERR_UNSUPPORTED) return st;
    return cost_volume_generic(ctx, scr, a, FeatVolume{feat_src, a.Ws, F}, FeatVolume{feat_tgt, a.Wt, F}, cv);
}

int dev_cost_volume_from_images(svh_context *ctx, Scratch &scr, const CostVolumeArgs &a, ImageDesc src, ImageDesc tgt, int h_r,
                                int v_r, float *cv) {
    if (func_census(a.func)) {
        int F = (2 * h_r + 1) * (2 * v_r + 1) * src.C;
        int nWw = census_words_written(F);
        uint32_t *sw = scr.get_n<uint32_t>((size_t)a.H * a.Ws * (nWw ? nWw : 1));
        uint32_t *tw = scr.get_n<uint32_t>((size_t)a.H * a.Wt * (nWw ? nWw : 1));
        if (!sw || !tw) return SVH_ERR_OUT_OF_MEMORY;
        SVH_TRY(dev_census_pair_compact(ctx, src, tgt, h_r, v_r, nWw, sw, tw));
        return dev_hamming_volume(ctx, a, sw, tw, nWw, cv);
    }
    // Windows up to 15 wide on images of up to four channels: the column-sum kernel.  When its tile (window rows x channels x the whole
    // disparity range) is over the LDS budget it runs on chunks of the range -- sooner than the per-window kernel (grey images, at half
    // the rate) or the per-voxel kernel below (which walks and, for normalised functions, divides every sample of both windows of every
    // voxel: 50 - 80 ms where the chunks take about one).
    if (!a.literal && a.n_dh == 1 && a.row_count == 0 && !(a.reduce && a.reduce->mode) && !cost_volume_colsum_applies(ctx, a, src, tgt, h_r, v_r)) {
        CostVolumeArgs ac = a;
        ac.reduce = nullptr;
        int chunk = 0;
        for (int dc = (a.D - 1) / 16 * 16; dc >= 64; dc -= 16) {
            ac.D = dc;
            if (cost_volume_colsum_applies(ctx, ac, src, tgt, h_r, v_r)) {
                chunk = dc;
                break;
            }
        }
        if (chunk > 0) {
            WindowStatsCache stats; // means / norms / zero-target costs do not depend on the chunk: computed by the first
            stats.scr = &scr;
            if (!ac.stats) ac.stats = &stats;
            ac.out_px_stride = a.px_stride();
            const int n_chunks = ceil_div(a.D, chunk), even = ceil_div(ceil_div(a.D, n_chunks), 16) * 16; // (chunks of about equal length)
            for (int d0 = 0; d0 < a.D; d0 += even) {
                ac.D = std::min(even, a.D - d0);
                ac.disp_lower = a.disp_lower + d0;
                ac.out_off = a.out_off + d0;
                const int sc = dev_cost_volume_grey_tiled(ctx, scr, ac, src, tgt, h_r, v_r, cv);
                if (sc != SVH_OK) return sc == SVH_ERR_UNSUPPORTED ? fail(ctx, SVH_ERR_HIP, "internal: a chunk of the disparity range left the tiled kernel") : sc;
            }
            return SVH_OK;
        }
    }
    // grey images with windows up to 11x11: LDS-tiled, register-blocked kernels (svh_cost_volume_tiled.hip)
    const int st = a.literal ? SVH_ERR_UNSUPPORTED : dev_cost_volume_grey_tiled(ctx, scr, a, src, tgt, h_r, v_r, cv);
    if (st != SVH_ERR_UNSUPPORTED) return st;
    return cost_volume_generic(ctx, scr, a, FeatImage{src.data, src.H, src.W, src.C, h_r, v_r},
                               FeatImage{tgt.data, tgt.H, tgt.W, tgt.C, h_r, v_r}, cv);
}

} // namespace svh

using namespace svh;

extern "C" {

int svh_feature_cost_volume(svh_context *ctx, int match_func, int disp_direction, const svh_array *feat_l,
                            const svh_array *feat_r, int32_t disp_lower, int32_t disp_count, svh_array *cv) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, feat_l, "feat_l", SVH_F32, 3, 3));
    SVH_TRY(validate(ctx, feat_r, "feat_r", SVH_F32, 3, 3));
    SVH_TRY(validate(ctx, cv, "cv", SVH_F32, 3, 3));
    if (!func_supported(match_func)) return fail(ctx, SVH_ERR_UNSUPPORTED, "matching function %d", match_func);
    if (disp_direction != SVH_LEFT_TO_RIGHT && disp_direction != SVH_RIGHT_TO_LEFT)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad disparity direction");
    // aggregateCost: cross_correlations.h:209-211
    if (feat_l->shape[0] != feat_r->shape[0]) return fail(ctx, SVH_EMPTY_RESULT, "row counts differ");
    if (feat_l->shape[2] != feat_r->shape[2]) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "feature counts differ");
    if (disp_count <= 0) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "disp_count must be positive");
    int F = (int)feat_l->shape[2];
    if (func_census(match_func) && F <= 1) return fail(ctx, SVH_EMPTY_RESULT, "census needs at least two feature channels");
    bool r2l = disp_direction == SVH_RIGHT_TO_LEFT;
    const svh_array *src = r2l ? feat_r : feat_l, *tgt = r2l ? feat_l : feat_r; // condImgRef, correlation_base.h:829-878
    CostVolumeArgs a{match_func, disp_direction, (int)src->shape[0], (int)src->shape[1], (int)tgt->shape[1], disp_lower, disp_count};
    if (cv->shape[0] != a.H || cv->shape[1] != a.Ws || cv->shape[2] != a.D)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "cv must have shape (%d,%d,%d)", a.H, a.Ws, a.D);
    Scratch scr(ctx);
    void *ds, *dt;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *src, &ds));
    SVH_TRY(stage_in(ctx, scr, *tgt, &dt));
    SVH_TRY(stage_out(ctx, scr, *cv, &os));
    SVH_TRY(dev_cost_volume_from_features(ctx, scr, a, (const float *)ds, (const float *)dt, F, (float *)os.dptr));
    return finish_out(ctx, os);
}

} // extern "C"

static int unfold_cost_volume_impl(svh_context *ctx, int match_func, int disp_direction, const svh_array *img_l, const svh_array *img_r, int h_radius,
                                   int v_radius, int32_t disp_lower, int32_t disp_count, svh_array *cv, svh_array *minima, int *minima_written,
                                   svh_array *winner = nullptr, int *winner_written = nullptr) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    if (minima_written) *minima_written = 0;
    if (winner_written) *winner_written = 0;
    SVH_TRY(validate_image(ctx, img_l, "img_l", match_func));
    SVH_TRY(validate_image(ctx, img_r, "img_r", match_func));
    SVH_TRY(validate(ctx, cv, "cv", SVH_F32, 3, 3));
    if (!func_supported(match_func)) return fail(ctx, SVH_ERR_UNSUPPORTED, "matching function %d", match_func);
    if (disp_direction != SVH_LEFT_TO_RIGHT && disp_direction != SVH_RIGHT_TO_LEFT)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad disparity direction");
    if (h_radius < 0 || v_radius < 0 || h_radius > 255 || v_radius > 255)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "radii must be in [0,255] (uint8_t in the reference)");
    if (img_l->ndim != img_r->ndim) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "image ranks differ");
    // unfoldBasedCostVolume: cross_correlations.h:751-759
    if (img_l->shape[0] != img_r->shape[0]) return fail(ctx, SVH_EMPTY_RESULT, "row counts differ");
    int C = img_l->ndim == 3 ? (int)img_l->shape[2] : 1;
    if (img_l->ndim == 3 && img_l->shape[2] != img_r->shape[2]) return fail(ctx, SVH_EMPTY_RESULT, "channel counts differ");
    if (disp_count <= 0) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "disp_count must be positive");
    int F = (2 * h_radius + 1) * (2 * v_radius + 1) * C;
    if (func_census(match_func) && F <= 1) return fail(ctx, SVH_EMPTY_RESULT, "census needs at least two feature channels");
    bool r2l = disp_direction == SVH_RIGHT_TO_LEFT;
    const svh_array *src = r2l ? img_r : img_l, *tgt = r2l ? img_l : img_r;
    CostVolumeArgs a{match_func, disp_direction, (int)src->shape[0], (int)src->shape[1], (int)tgt->shape[1], disp_lower, disp_count};
    if (cv->shape[0] != a.H || cv->shape[1] != a.Ws || cv->shape[2] != a.D)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "cv must have shape (%d,%d,%d)", a.H, a.Ws, a.D);
    Scratch scr(ctx);
    void *ds, *dt;
    OutStage os, om;
    SVH_TRY(stage_image(ctx, scr, *src, &ds));
    SVH_TRY(stage_image(ctx, scr, *tgt, &dt));
    SVH_TRY(stage_out(ctx, scr, *cv, &os));
    int written = 0;
    CostReduce red;
    const ImageDesc isrc{(const float *)ds, a.H, a.Ws, C}, itgt{(const float *)dt, a.H, a.Wt, C};
    if (minima) {
        SVH_TRY(validate(ctx, minima, "minima", SVH_F32, 3, 3));
        if (minima->shape[0] != a.H || minima->shape[1] != a.Ws || minima->shape[2] != 2)
            return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "minima must have shape (%d,%d,2)", a.H, a.Ws);
    }
    if (minima && func_census(match_func)) {
        SVH_TRY(stage_out(ctx, scr, *minima, &om));
        a.minima = (float *)om.dptr;
        a.minima_written = &written;
    } else if (minima && func_strategy(match_func) == SVH_COST && ctx->cost_reduce_fused && C == 1 && cost_volume_colsum_applies(ctx, a, isrc, itgt, h_radius, v_radius)) {
        // float costs of a Cost-strategy function: the column-sum kernel leaves the regional minima (CostReduce mode 2) and says whether a
        // magnitude left the regime in which the Cost branch's recurrences may run on them alone (svh_sgm.hip)
        SVH_TRY(stage_out(ctx, scr, *minima, &om));
        red.mode = 2;
        red.minima = (float *)om.dptr;
        red.flag = scr.get_n<int>(64);
        if (!red.flag) return SVH_ERR_OUT_OF_MEMORY;
        SVH_HIP_CHECK(ctx, hipMemsetAsync(red.flag, 0, sizeof(int), ctx->stream));
        a.reduce = &red;
    }
    OutStage ow;
    if (winner) { // the column-sum kernel picks extractSelectedIndex's index while it holds a pixel's costs (CostReduce mode 1) and stores the volume too
        SVH_TRY(validate(ctx, winner, "winner", SVH_I32, 2, 2));
        if (winner->shape[0] != a.H || winner->shape[1] != a.Ws) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "winner must have shape (%d,%d)", a.H, a.Ws);
        if (!func_census(match_func) && ctx->cost_reduce_fused && C == 1 && cost_volume_colsum_applies(ctx, a, isrc, itgt, h_radius, v_radius)) { // (grey images: the reducing forms)
            SVH_TRY(stage_out(ctx, scr, *winner, &ow));
            red.mode = 1;
            red.score = func_strategy(match_func) != SVH_COST;
            red.idx = (int32_t *)ow.dptr;
            red.store = true;
            a.reduce = &red;
        }
    }
    SVH_TRY(dev_cost_volume_from_images(ctx, scr, a, isrc, itgt, h_radius, v_radius, (float *)os.dptr));
    if (red.mode == 1 && red.done) {
        SVH_TRY(finish_out(ctx, ow));
        if (winner_written) *winner_written = 1;
    }
    if (red.mode == 2 && red.done) {
        // one word back to the host: the statement is only made for a volume inside the regime (every finite |c| <= 1e30)
        int flag = 0;
        SVH_HIP_CHECK(ctx, hipMemcpyAsync(&flag, red.flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SVH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if ((flag & 2) == 0) written = 2;
    }
    if (written) SVH_TRY(finish_out(ctx, om));
    if (minima_written) *minima_written = written;
    return finish_out(ctx, os);
}

extern "C" {

int svh_unfold_cost_volume(svh_context *ctx, int match_func, int disp_direction, const svh_array *img_l, const svh_array *img_r, int h_radius,
                           int v_radius, int32_t disp_lower, int32_t disp_count, svh_array *cv) {
    return unfold_cost_volume_impl(ctx, match_func, disp_direction, img_l, img_r, h_radius, v_radius, disp_lower, disp_count, cv, nullptr, nullptr);
}

int svh_unfold_cost_volume_minima(svh_context *ctx, int match_func, int disp_direction, const svh_array *img_l, const svh_array *img_r, int h_radius,
                                  int v_radius, int32_t disp_lower, int32_t disp_count, svh_array *cv, svh_array *minima, int *minima_written) {
    return unfold_cost_volume_impl(ctx, match_func, disp_direction, img_l, img_r, h_radius, v_radius, disp_lower, disp_count, cv, minima, minima_written);
}

int svh_unfold_cost_volume_winner(svh_context *ctx, int match_func, int disp_direction, const svh_array *img_l, const svh_array *img_r, int h_radius,
                                  int v_radius, int32_t disp_lower, int32_t disp_count, svh_array *cv, svh_array *winner, int *winner_written) {
    return unfold_cost_volume_impl(ctx, match_func, disp_direction, img_l, img_r, h_radius, v_radius, disp_lower, disp_count, cv, nullptr, nullptr, winner,
                                   winner_written);
}

} // extern "C"

// ---- 2-D disparity volumes ----------------------------------------------------------
// unfoldBased2dDisparityCostVolume<matchFunc,...>(img_l, img_r, h_radius, v_radius, searchOffset<2>)
//                                                                        correlation/cross_correlations.h:794-822, :310-374
// cv (H, W, Dh, Dw): CV(i,j,dh,dw) = cmp(src(i,j,:), tgt(i + dh + lower0, j + dw + lower1, :)).  Built one vertical offset
// at a time with the 1-D kernels (target row offset, +1 column sign, strided output).
extern "C" int svh_unfold_cost_volume_2d(svh_context *ctx, int match_func, int disp_direction, const svh_array *img_l,
                                         const svh_array *img_r, int h_radius, int v_radius, int32_t lower0, int32_t upper0,
                                         int32_t lower1, int32_t upper1, svh_array *cv) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate_image(ctx, img_l, "img_l", match_func));
    SVH_TRY(validate_image(ctx, img_r, "img_r", match_func));
    SVH_TRY(validate(ctx, cv, "cv", SVH_F32, 4, 4));
    if (!func_supported(match_func)) return fail(ctx, SVH_ERR_UNSUPPORTED, "matching function %d", match_func);
    if (disp_direction != SVH_LEFT_TO_RIGHT && disp_direction != SVH_RIGHT_TO_LEFT)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad disparity direction");
    if (h_radius < 0 || v_radius < 0 || h_radius > 255 || v_radius > 255)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "radii must be in [0,255] (uint8_t in the reference)");
    if (img_l->ndim != img_r->ndim) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "image ranks differ");
    // cross_correlations.h:801-816: rows, columns and channels must agree; :338-340: non-empty ranges
    if (img_l->shape[0] != img_r->shape[0] || img_l->shape[1] != img_r->shape[1]) return fail(ctx, SVH_EMPTY_RESULT, "image sizes differ");
    const int C = img_l->ndim == 3 ? (int)img_l->shape[2] : 1;
    if (img_l->ndim == 3 && img_l->shape[2] != img_r->shape[2]) return fail(ctx, SVH_EMPTY_RESULT, "channel counts differ");
    const int Dh = upper0 - lower0 + 1, Dw = upper1 - lower1 + 1;
    if (Dh <= 0 || Dw <= 0) return fail(ctx, SVH_EMPTY_RESULT, "empty search range");
    const int F = (2 * h_radius + 1) * (2 * v_radius + 1) * C;
    if (func_census(match_func) && F <= 1) return fail(ctx, SVH_EMPTY_RESULT, "census needs at least two feature channels");
    const bool r2l = disp_direction == SVH_RIGHT_TO_LEFT;
    const svh_array *src = r2l ? img_r : img_l, *tgt = r2l ? img_l : img_r;
    const int H = (int)src->shape[0], W = (int)src->shape[1];
    if (cv->shape[0] != H || cv->shape[1] != W || cv->shape[2] != Dh || cv->shape[3] != Dw)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "cv must have shape (%d,%d,%d,%d)", H, W, Dh, Dw);
    Scratch scr(ctx);
    void *ds, *dt;
    OutStage os;
    SVH_TRY(stage_image(ctx, scr, *src, &ds));
    SVH_TRY(stage_image(ctx, scr, *tgt, &dt));
    SVH_TRY(stage_out(ctx, scr, *cv, &os));
    const ImageDesc isrc{(const float *)ds, H, W, C}, itgt{(const float *)dt, H, W, C};
    CostVolumeArgs a{match_func, disp_direction, H, W, W, lower1, Dw};
    a.force_sign = 1;
    a.out_px_stride = (int64_t)Dh * Dw;
    if (func_census(match_func)) { // census words once, one Hamming pass per vertical offset
        const int nWw = census_words_written(F);
        uint32_t *sw = scr.get_n<uint32_t>((size_t)H * W * (nWw ? nWw : 1));
        uint32_t *tw = scr.get_n<uint32_t>((size_t)H * W * (nWw ? nWw : 1));
        if (!sw || !tw) return SVH_ERR_OUT_OF_MEMORY;
        SVH_TRY(dev_census_pair_compact(ctx, isrc, itgt, h_radius, v_radius, nWw, sw, tw));
        if (dev_hamming_volume_2d(ctx, sw, tw, nWw, H, W, W, Dh, Dw, lower0, lower1, (float *)os.dptr)) {
            SVH_CHECK_LAUNCH(ctx);
            return finish_out(ctx, os);
        }
        for (int dh = 0; dh < Dh; dh++) {
            a.tgt_row_off = lower0 + dh;
            a.out_off = (int64_t)dh * Dw;
            SVH_TRY(dev_hamming_volume(ctx, a, sw, tw, nWw, (float *)os.dptr));
        }
    } else {
        WindowStatsCache stats; // means / norms / zero-target costs do not depend on the offset: computed by the first pass
        stats.scr = &scr;
        a.stats = &stats;
        // as many vertical offsets per launch as the column-sum kernel's tile holds (grey images, separable terms); one otherwise
        int group = 1;
        if (ctx->fold_2d_offsets) {
            for (int g = Dh; g > 1; g--) {
                a.n_dh = g;
                if (cost_volume_colsum_applies(ctx, a, isrc, itgt, h_radius, v_radius)) {
                    group = g;
                    break;
                }
            }
        }
        for (int dh = 0; dh < Dh; dh += group) {
            a.n_dh = std::min(group, Dh - dh);
            a.tgt_row_off = lower0 + dh;
            a.out_off = (int64_t)dh * Dw;
            Scratch inner(ctx); // the per-offset statistics maps are released after each pass
            SVH_TRY(dev_cost_volume_from_images(ctx, inner, a, isrc, itgt, h_radius, v_radius, (float *)os.dptr));
        }
    }
    return finish_out(ctx, os);
}

// featureVolume2CostVolume<matchFunc, ..., searchOffset<2>, dDir>(feature_vol_l, feature_vol_r, searchRange)
//                                                       correlation/cross_correlations.h:724-738 over aggregateCost :310-374
// Raw (H,W*,F) float feature volumes in, getFeatureVolumeForMatchFunc applied on the fly; only the row counts must agree
// (:324-326), the target may be narrower or wider than the source.
extern "C" int svh_feature_cost_volume_2d(svh_context *ctx, int match_func, int disp_direction, const svh_array *feat_l, const svh_array *feat_r,
                                          int32_t lower0, int32_t upper0, int32_t lower1, int32_t upper1, svh_array *cv) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, feat_l, "feat_l", SVH_F32, 3, 3));
    SVH_TRY(validate(ctx, feat_r, "feat_r", SVH_F32, 3, 3));
    SVH_TRY(validate(ctx, cv, "cv", SVH_F32, 4, 4));
    if (!func_supported(match_func)) return fail(ctx, SVH_ERR_UNSUPPORTED, "matching function %d", match_func);
    if (disp_direction != SVH_LEFT_TO_RIGHT && disp_direction != SVH_RIGHT_TO_LEFT)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad disparity direction");
    if (feat_l->shape[0] != feat_r->shape[0]) return fail(ctx, SVH_EMPTY_RESULT, "row counts differ");
    if (feat_l->shape[2] != feat_r->shape[2]) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "feature counts differ");
    const int Dh = upper0 - lower0 + 1, Dw = upper1 - lower1 + 1;
    if (Dh <= 0 || Dw <= 0) return fail(ctx, SVH_EMPTY_RESULT, "empty search range"); // :338-340
    const int F = (int)feat_l->shape[2];
    if (func_census(match_func) && F <= 1) return fail(ctx, SVH_EMPTY_RESULT, "census needs at least two feature channels");
    const bool r2l = disp_direction == SVH_RIGHT_TO_LEFT;
    const svh_array *src = r2l ? feat_r : feat_l, *tgt = r2l ? feat_l : feat_r;
    const int H = (int)src->shape[0], Ws = (int)src->shape[1], Wt = (int)tgt->shape[1];
    if (cv->shape[0] != H || cv->shape[1] != Ws || cv->shape[2] != Dh || cv->shape[3] != Dw)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "cv must have shape (%d,%d,%d,%d)", H, Ws, Dh, Dw);
    Scratch scr(ctx);
    void *ds, *dt;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *src, &ds));
    SVH_TRY(stage_in(ctx, scr, *tgt, &dt));
    SVH_TRY(stage_out(ctx, scr, *cv, &os));
    CostVolumeArgs a{match_func, disp_direction, H, Ws, Wt, lower1, Dw};
    a.force_sign = 1;
    a.out_px_stride = (int64_t)Dh * Dw;
    if (func_census(match_func)) { // census words once (target side rounded through float, rule E2), one Hamming pass per vertical offset
        const int nWw = census_words_written(F);
        uint32_t *sw = scr.get_n<uint32_t>((size_t)H * Ws * (nWw ? nWw : 1));
        uint32_t *tw = scr.get_n<uint32_t>((size_t)H * Wt * (nWw ? nWw : 1));
        if (!sw || !tw) return SVH_ERR_OUT_OF_MEMORY;
        SVH_TRY(dev_census_from_features(ctx, (const float *)ds, H, Ws, F, nWw, false, sw));
        SVH_TRY(dev_census_from_features(ctx, (const float *)dt, H, Wt, F, nWw, true, tw));
        if (dev_hamming_volume_2d(ctx, sw, tw, nWw, H, Ws, Wt, Dh, Dw, lower0, lower1, (float *)os.dptr)) {
            SVH_CHECK_LAUNCH(ctx);
            return finish_out(ctx, os);
        }
        for (int dh = 0; dh < Dh; dh++) {
            a.tgt_row_off = lower0 + dh;
            a.out_off = (int64_t)dh * Dw;
            SVH_TRY(dev_hamming_volume(ctx, a, sw, tw, nWw, (float *)os.dptr));
        }
    } else {
        for (int dh = 0; dh < Dh; dh++) {
            a.tgt_row_off = lower0 + dh;
            a.out_off = (int64_t)dh * Dw;
            Scratch inner(ctx); // per-pass mean / norm maps
            SVH_TRY(dev_cost_volume_from_features(ctx, inner, a, (const float *)ds, (const float *)dt, F, (float *)os.dptr));
        }
    }
    return finish_out(ctx, os);
}

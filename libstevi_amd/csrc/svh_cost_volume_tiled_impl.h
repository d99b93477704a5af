// LDS-tiled cost volumes for grey images (CC / NCC / SSD / SAD and their zero-mean variants).
//
// Two kernels.  cost_volume_colsum_kernel (round 3; everything but ZSAD) uses what the windows of neighbouring pixels share: the
// per-column sums  V_d(x) = sum_k f(S(i + k, x), T(i + k, x + d))  over the 2 v_r + 1 window rows belong to the 2 h_r + 1 windows that
// contain column x, so a voxel costs 2 v_r + 1 multiply-adds (its column) plus the horizontal sum of 2 h_r + 1 column sums, taken
// across the lanes with DPP shifts, instead of (2 h_r + 1)(2 v_r + 1) multiply-adds: 11 + ~6 vector instructions per voxel instead of
// 121 for 11 x 11 windows -- which leaves the 4 bytes the voxel is written with as the bound.  cost_volume_pxlane_kernel (round 1)
// evaluates every window on its own; ZSAD keeps it, because |(s - t) - (mean_s - mean_t)| depends on the window as a whole.
// No MFMA: the column sums are band-diagonal outer products of two image rows -- 2 v_r + 1 of them per voxel, not a contraction.
//
// pxlane: a block owns 64 consecutive pixels of one row and every disparity.  The (2v_r+1)-row source and target tiles are staged in
// LDS once; a lane is a pixel and keeps a block of 16 or 32 consecutive disparities in registers (see the kernel comment).
//
// Arithmetic relative to the reference (cross_correlations.h:416-594, matching_costs.h:59-156): the products /
// differences are accumulated per pixel in the reference's channel order (rows outer, columns inner); the per-element
// normalisations are applied algebraically after the sum,
//     NCC   sum (s/ns)(t/nt)            = (sum s t) / (ns nt)
//     ZCC   sum (s-ms)(t-mt)            = sum (s-c)(t-c) - F (ms-c)(mt-c)        (c: a block constant, keeps magnitudes small)
//     ZSSD  sum ((s-ms)-(t-mt))^2       = sum (s-t)^2 - F (ms-mt)^2
//     ZSAD  sum |(s-ms)-(t-mt)|         = sum |(s-t) - (ms-mt)|
// which changes results by rounding only (a few 1e-7 relative; the north-star tolerance for float costs is 1e-4).  The column-sum
// kernel also changes the ORDER of the window's terms (columns outer, rows inner; the reference: rows outer, cross_correlations.h
// via unfold.h:283): plain sums of the same terms, no running subtraction, so again rounding only.
// Means, norms and the all-zero-target cost (target column outside the image, cross_correlations.h:235) come from a
// per-pixel statistics kernel that follows the reference literally.
//
// This header holds the kernels and their launchers; one translation unit per comparison (svh_cost_volume_tiled_{dot,dotzm,ssd,sad}.hip) instantiates
// them, so that the four compile side by side (as one unit they were the five and a half minutes every clean build waited for).
#pragma once
#include "svh_compare.h"
#include "svh_internal.h"

namespace svh {

namespace {

enum { T_DOT = 0, T_SSD = 1, T_SAD = 2 };

__device__ __forceinline__ float image_or_zero1(const float *__restrict__ img, int H, int W, int i, int j) {
    return (i >= 0 && i < H && j >= 0 && j < W) ? img[(int64_t)i * W + j] : 0.0f;
}

// A block owns PL_TPX = 64 consecutive pixels of a row: lane u of every wave is pixel j0 + u, and each wave walks blocks of DB
// consecutive disparities held in registers (wave w takes blocks w, w + 4, ...).  For one window row a lane reads its h source
// samples and the h + DB - 1 target samples its DB windows cover (both stride-1 across the lanes: conflict-free LDS reads) and
// issues h DB multiply-adds on register operands only (v_pk_fma_f32): no cross-lane traffic, about h DB / (2 h + DB - 1)
// multiply-adds per LDS read, and no idle lanes when the disparity range is short (2-D volumes, the coarse levels of
// hierarchical matching).  A lane stores its DB costs as 16-byte pieces of its pixel's run.  The direction sign is a template
// parameter: as a runtime value it turns the operand selection into indirect register indexing.
// (The first version of this file put the disparities on the lanes and eight pixels per wave in registers, with the source
// samples broadcast by v_readlane: 2.45 ms for 1080p x 256 NCC 11x11 against 1.79 ms for this form.)
constexpr int PL_TPX = 64;

// four consecutive costs with only the 4-byte alignment every cost has: global_store_dwordx4 needs no more on gfx950, so the pieces of rows
// of any length (2-D volumes: 9 x 33 offsets make rows of 297 costs) are stored 16 bytes per lane like those of aligned rows.  (With the
// 16-byte alignment test the 2-D volumes took the store-per-cost path for every block: 3.4 ms for 1080p x 297 ZNCC 7x7.)
struct __attribute__((packed, aligned(4))) CostPiece4 {
    float x, y, z, w;
};

template <int CMP, bool ZM, int HR, int DB, int SIGN>
__global__ void __launch_bounds__(256) cost_volume_pxlane_kernel(const float *__restrict__ src, const float *__restrict__ tgt, int H, int Ws, int Wt, int v_r,
                                                                 int D, int disp_lower, const float *__restrict__ mean_s,
                                                                 const float *__restrict__ mean_t, const float *__restrict__ norm_s,
                                                                 const float *__restrict__ norm_t, const float *__restrict__ zcost, int row_off,
                                                                 int64_t px_stride, int64_t out_off, float *__restrict__ cv, int row0) {
    constexpr int h = 2 * HR + 1, NT = h + DB - 1;
    extern __shared__ float lds[];
    const int v = 2 * v_r + 1;
    const int n_blocks = (D + DB - 1) / DB;
    const int sw = PL_TPX + h - 1, tw = PL_TPX + h - 1 + n_blocks * DB - 1;
    float *stile = lds, *ttile = lds + v * sw;
    const int i = row0 + blockIdx.y, j0 = blockIdx.x * PL_TPX;
    const float c0 = (ZM && CMP == T_DOT) ? src[(int64_t)i * Ws + min(j0, Ws - 1)] : 0.0f; // keeps the zero-mean dot products small
    const int it = i + row_off; // target row (2-D disparity volumes); outside the image the target vector is zero
    const bool trow_in = it >= 0 && it < H;
    for (int k = 0; k < v; k++) {
        const int ii = i - v_r + k;
        for (int x = threadIdx.x; x < sw; x += blockDim.x) stile[k * sw + x] = image_or_zero1(src, H, Ws, ii, j0 - HR + x) - c0;
        for (int z = threadIdx.x; z < tw; z += blockDim.x) {
            // forward: tile index = column offset; LeftToRight: mirrored, so that the index still grows with d
            const int jt = SIGN > 0 ? j0 + disp_lower - HR + z : j0 + (PL_TPX - 1) - disp_lower + HR - z;
            ttile[k * tw + z] = image_or_zero1(tgt, H, Wt, it - v_r + k, jt) - c0;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = j0 + lane;
    const bool px_in = j < Ws;
    const int64_t px = (int64_t)i * Ws + min(j, Ws - 1);
    const int tlane = SIGN > 0 ? lane : PL_TPX - 1 - lane;
    const float Ff = (float)(h * v);
    const float ms = ZM ? mean_s[px] : 0.0f;
    for (int db = wave; db < n_blocks; db += 4) {
        const int d0 = db * DB;
        float acc[DB], delta[DB];
#pragma unroll
        for (int q = 0; q < DB; q++) {
            acc[q] = 0.0f;
            delta[q] = 0.0f;
        }
        if (ZM && CMP == T_SAD) {
#pragma unroll
            for (int q = 0; q < DB; q++) {
                const int jt = j + SIGN * (disp_lower + d0 + q);
                if (px_in && d0 + q < D && trow_in && jt >= 0 && jt < Wt) delta[q] = ms - mean_t[(int64_t)it * Wt + jt];
            }
        }
        for (int k = 0; k < v; k++) {
            float sv[h], tv[NT];
            const float *srow = stile + k * sw + lane, *trow = ttile + k * tw + tlane + d0;
#pragma unroll
            for (int l = 0; l < h; l++) sv[l] = srow[l];
#pragma unroll
            for (int m = 0; m < NT; m++) tv[m] = trow[m];
#pragma unroll
            for (int l = 0; l < h; l++) {
#pragma unroll
                for (int q = 0; q < DB; q++) {
                    const float s = sv[l], t = tv[SIGN > 0 ? l + q : (h - 1 - l) + q];
                    if (CMP == T_DOT) {
                        acc[q] = fmaf(s, t, acc[q]);
                    } else if (CMP == T_SSD) {
                        const float df = s - t;
                        acc[q] = fmaf(df, df, acc[q]);
                    } else {
                        acc[q] += fabsf((s - t) - delta[q]);
                    }
                }
            }
        }
        if (!px_in) continue;
        const float nsv = norm_s ? norm_s[px] : 1.0f, zc = zcost[px];
        float r[DB];
#pragma unroll
        for (int q = 0; q < DB; q++) {
            const int d = d0 + q;
            const int jt = j + SIGN * (disp_lower + d);
            r[q] = acc[q];
            if (d < D && trow_in && jt >= 0 && jt < Wt) {
                const int64_t pt = (int64_t)it * Wt + jt;
                if (ZM && CMP == T_DOT) r[q] -= Ff * (ms - c0) * (mean_t[pt] - c0);
                if (ZM && CMP == T_SSD) {
                    const float dm = ms - mean_t[pt];
                    r[q] -= Ff * dm * dm;
                }
                if (norm_s) r[q] /= nsv * norm_t[pt];
            } else {
                r[q] = zc;
            }
        }
        float *o = cv + px * px_stride + out_off + d0;
        if (d0 + DB <= D) {
#pragma unroll
            for (int q = 0; q < DB; q += 4) *reinterpret_cast<CostPiece4 *>(o + q) = CostPiece4{r[q], r[q + 1], r[q + 2], r[q + 3]};
        } else {
#pragma unroll
            for (int q = 0; q < DB; q++)
                if (d0 + q < D) o[q] = r[q];
        }
    }
}

// ---- column sums + horizontal DPP sums --------------------------------------------------------------------------------------
// A wave owns CS_COLS = 128 consecutive image columns of one row (lane u: columns 2 u and 2 u + 1 of the tile) and walks blocks of
// CS_DB consecutive disparities; the four waves of a block share the staged rows and take every fourth disparity block.  Per window
// row a lane reads its two source samples and the CS_DB + 1 target samples its two columns meet, and adds one term to each of its
// 2 CS_DB column sums.  Then the window sums: with P_n(u) = V(u) + P_{n-1}(u - 1) (one v_add_f32 with a wave_shr:1 DPP operand per
// step: P_n(u) = V(u) + ... + V(u - n)) on the even and the odd columns of the lanes,
//     window ending at column 2 u     = Peven_{h_r}(u) + Podd_{h_r - 1}(u - 1),
//     window ending at column 2 u + 1 = Podd_{h_r}(u)  + Peven_{h_r - 1}(u):   2 h_r + 2 additions for two voxels.
// Lanes u >= h_r hold complete windows: a tile yields 128 - 2 h_r output pixels (centre = window end - h_r).
constexpr int CS_COLS = 128, CS_DB = 16;
constexpr int CS_XP = 20; // floats per pixel in the transposition area: 16 costs + 4 of padding (16-byte pieces of 16 lanes then fall on different banks)

__device__ __forceinline__ float lane_below(float v) { // value of lane - 1 (wave_shr:1; lane 0 reads 0: its windows are incomplete anyway)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}

// what the kernel reduces over the disparity axis besides (or instead of) storing the costs: CostReduce of svh_internal.h as a kernel argument
struct ColsumReduce {
    int mode, score, store;
    int32_t *idx, *disp;
    int disp_sign, disp_offset;
    float2 *minima;
    int *flag;
    float big;
};

// RED: 0 none (the kernel of round 3, register for register: 119-126 VGPRs, two blocks of eight waves per CU), 1 winner, 2 regional
// minima.  A compile-time parameter because the reductions' running state (eight registers live across the whole disparity loop) took
// the kernel to 150 VGPRs and one block per CU when it was a run-time switch: every plain cost volume paid 0.19 ms of 0.54 for a feature
// it did not use.  The reducing forms are held to four waves per SIMD as well (second launch bound).
// FOLD (2-D disparity volumes, round 4f): the launch takes n_dh consecutive vertical offsets at once.  The target tile holds v + n_dh - 1
// rows; disparity block b belongs to vertical offset b / (blocks per offset) and reads the v rows that start at that offset.  One launch
// per vertical offset staged the v source rows once per offset for two or three blocks of work and left a wave of four without any.
// Search ranges are symmetric as a rule -- 2 r + 1 offsets, 17, 33, 65 -- so the last block of an offset would hold ONE disparity.
// The row loads of a block already bring the target samples of a seventeenth (the pairs t[16], t[17] of the last 8-byte read), so the
// last block of an ODD range is seventeen wide: two more single multiply-adds per window row instead of a block (16 k + 1 offsets
// run k blocks).  And the last block of a range that is no multiple of 16 (17 for odd ranges) starts early, at Dw - 16 (Dw - 17),
// instead of ending late: it recomputes what the block before it stores -- the same terms in the same order, the same bits -- and
// every block of a range of 16 or more takes the 64-byte store path.  Both starts are even: an odd start makes every 8-byte LDS read
// of the block misaligned (9 x 33 offsets with the last block at 17: 1.96 ms, against 1.27 for 9 x 34 with it at 18; a partial last
// block instead of the early start: 1.41).  Ranges shorter than 16 are one partial block per offset.
// MC (round 4f): images of C interleaved channels (RGB).  A window's terms are the sum over its rows AND channels of the same per-column
// terms, so the channels are staged as rows of their own -- tile row k C + c holds channel c of window row k -- and the kernel walks
// (2 v_r + 1) C rows.  (Colour images took the per-voxel kernel: NCC 7x7 at 1080p x 128 52.7 ms where the grey image takes 0.4.)
template <int CMP, bool ZM, bool NRM, int HR, int SIGN, int WAVES, int RED, bool FOLD = false, bool MC = false>
__global__ void __launch_bounds__(64 * WAVES, RED ? 4 : 1) cost_volume_colsum_kernel(const float *__restrict__ src, const float *__restrict__ tgt, int H, int Ws, int Wt, int v_r,
                                                                 int D, int disp_lower, const float *__restrict__ mean_s,
                                                                 const float *__restrict__ mean_t, const float *__restrict__ norm_s,
                                                                 const float *__restrict__ norm_t, const float *__restrict__ zcost, int row_off,
                                                                 int64_t px_stride, int64_t out_off, float *__restrict__ cv, int row0, ColsumReduce rd, int n_dh, int C) {
    constexpr int DB = CS_DB, OUT = CS_COLS - 2 * HR;
    extern __shared__ float lds[];
    const int v = MC ? (2 * v_r + 1) * C : 2 * v_r + 1; // rows the kernel walks (window rows x channels)
    const bool ext = FOLD && D > DB && (D & 1);       // the last block of an offset is seventeen wide
    const int nbw = (D - (ext ? 1 : 0) + DB - 1) / DB; // blocks of one vertical offset
    const int n_blocks = FOLD ? nbw * n_dh : nbw;
    const int n_off = FOLD ? n_dh : 1, tv = v + (n_off - 1) * (MC ? C : 1); // vertical offsets of the launch, target rows staged (MC: C tile rows per image row)
    const int tw = CS_COLS + nbw * DB; // target tile columns (even: 8-byte aligned pairs)
    float *stile = lds, *ttile = lds + v * CS_COLS, *tmean = ttile + tv * tw, *tinv = tmean + (ZM ? n_off * tw : 0); // (tmean / tinv: ZM / NRM only)
    float *xpose = tinv + (NRM ? n_off * tw : 0) + (threadIdx.x >> 6) * (64 * CS_XP); // this wave's area for turning 64 pixels x DB costs around
    const int i = row0 + blockIdx.y, p0 = blockIdx.x * OUT, xb = p0 - HR; // first output pixel, first column of the tile
    const float c0 = (ZM && CMP == T_DOT) ? src[((int64_t)i * Ws + min(p0, Ws - 1)) * (MC ? C : 1)] : 0.0f; // keeps the zero-mean dot products small
    const int it = i + row_off; // target row (2-D disparity volumes); outside the image the target vector is zero
    const bool trow_in = it >= 0 && it < H;
    // tile index z of the target tile <-> image column: the index grows with the disparity in both directions
    auto tcol = [&](int z) { return SIGN > 0 ? xb + disp_lower + z : xb + (CS_COLS - 1) - disp_lower - z; };
    // staging: a thread owns tile columns (source columns first, then target indices) and walks the window rows four at a time, all
    // eight loads issued before the first LDS store (one load, one wait, one store per sample made the staging a chain of
    // memory latencies as long as the arithmetic of the block)
    for (int xc = threadIdx.x; xc < CS_COLS + tw; xc += blockDim.x) {
        const bool is_src = xc < CS_COLS;
        const int col = is_src ? xb + xc : tcol(xc - CS_COLS);
        const float *img = is_src ? src : tgt;
        const int Wi = is_src ? Ws : Wt, r0 = (is_src ? i : it) - v_r;
        float *dst = is_src ? stile + xc : ttile + (xc - CS_COLS);
        const int pitch = is_src ? CS_COLS : tw;
        const bool col_in = col >= 0 && col < Wi;
        const int nr = (FOLD && !is_src) ? tv : v;
        for (int k0 = 0; k0 < nr; k0 += 4) {
            float val[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int wrow = MC ? (k0 + kk) / C : k0 + kk, ch = MC ? (k0 + kk) - wrow * C : 0; // (window row, channel) of tile row k0 + kk
                const int rr = r0 + wrow;
                val[kk] = (k0 + kk < nr && col_in && rr >= 0 && rr < H) ? img[MC ? ((int64_t)rr * Wi + col) * C + ch : (int64_t)rr * Wi + col] : 0.0f;
            }
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
                if (k0 + kk < nr) dst[(k0 + kk) * pitch] = val[kk] - c0;
        }
    }
    if (ZM || NRM) { // per target pixel: mean and 1 / norm at the window CENTRE column tcol(z) (FOLD: one row of them per vertical offset)
        for (int zz = threadIdx.x; zz < n_off * tw; zz += blockDim.x) {
            const int dh = FOLD ? zz / tw : 0, z = zz - dh * tw;
            const int jt = tcol(z), itd = it + dh;
            const bool in = itd >= 0 && itd < H && jt >= 0 && jt < Wt;
            if (ZM) tmean[zz] = in ? mean_t[(int64_t)itd * Wt + jt] : 0.0f;
            if (NRM) tinv[zz] = in ? 1.0f / norm_t[(int64_t)itd * Wt + jt] : 0.0f;
        }
    }
    __syncthreads();
    const int u = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // (scalar: the disparity block is wave uniform)
    const float Ff = (float)((2 * HR + 1) * v);
    // this lane's two output pixels (window ends at columns 2 u and 2 u + 1, centres HR to the left) and their per-pixel statistics
    const int pe[2] = {p0 + 2 * (u - HR), p0 + 2 * (u - HR) + 1};
    bool live[2];
    float ms[2], inv_ns[2], zc[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
        live[e] = u >= HR && pe[e] < Ws;
        const int64_t px = (int64_t)i * Ws + min(max(pe[e], 0), Ws - 1);
        ms[e] = ZM ? mean_s[px] : 0.0f;
        inv_ns[e] = NRM ? 1.0f / norm_s[px] : 1.0f;
        zc[e] = zcost[px];
    }
    // per-pixel reductions over the disparities this wave walks (ColsumReduce): the running winner (value, index; index -1: no candidate
    // yet) or the two regional minima, and the largest finite magnitude seen
    const float red0 = (RED == 1 && rd.score) ? -INFINITY : INFINITY;
    float red_a[2] = {red0, red0}, red_b[2] = {INFINITY, INFINITY}, amax = 0.0f;
    int red_d[2] = {-1, -1};
    bool nan_at_0[2] = {false, false};
    for (int db = wave; db < n_blocks; db += WAVES) {
        int d0 = db * DB, dh = 0;
        bool ext_b = false; // (wave uniform) this block carries the seventeenth disparity
        if constexpr (FOLD) {
            dh = db / nbw;
            d0 = (db - dh * nbw) * DB;
            if (D > DB && d0 + DB + (ext ? 1 : 0) >= D) { // the last block of the offset: it ends with the range
                d0 = D - DB - (ext ? 1 : 0);
                ext_b = ext;
            }
        }
        const float *ttile_b = FOLD ? ttile + dh * (MC ? C : 1) * tw : ttile, *tmean_b = FOLD ? tmean + dh * tw : tmean, *tinv_b = FOLD ? tinv + dh * tw : tinv;
        const bool trow_in_b = FOLD ? (it + dh >= 0 && it + dh < H) : trow_in;
        const int64_t out_b = FOLD ? out_off + (int64_t)dh * D : out_off;
        // Column sums as register PAIRS (v_pk_fma_f32 takes aligned pairs): the row's target samples t[0 .. DB + 1] arrive as the
        // pairs (t[2 m], t[2 m + 1]).  The column that meets t[q] pairs its disparities (2 m, 2 m + 1); the one that meets t[q + 1]
        // pairs (2 m + 1, 2 m + 2) -- the same register pairs, one to the right -- and keeps disparities 0 and DB - 1 as singles.
        using v2f = float __attribute__((ext_vector_type(2)));
        v2f A[DB / 2], Bp[DB / 2 - 1];
        float b_first = 0.0f, b_last = 0.0f;
        float x17a = 0.0f, x17b = 0.0f; // FOLD: disparity d0 + 16 of the even / the odd column (accumulated by every block, used by the last)
#pragma unroll
        for (int m = 0; m < DB / 2; m++) A[m] = (v2f){0.0f, 0.0f};
#pragma unroll
        for (int m = 0; m < DB / 2 - 1; m++) Bp[m] = (v2f){0.0f, 0.0f};
        // column 2 u + e meets target index z = (2 u + e) + d (forward) or (127 - 2 u - e) + d (mirrored)
        const int zb = (SIGN > 0 ? 2 * u : CS_COLS - 2 - 2 * u) + d0;
        struct Row {
            v2f s, t[DB / 2 + 1];
        };
        auto load_row = [&](Row &rw, int k) {
            const float2 sv = *reinterpret_cast<const float2 *>(stile + k * CS_COLS + 2 * u);
            rw.s = (v2f){sv.x, sv.y};
            const float2 *trow = reinterpret_cast<const float2 *>(ttile_b + k * tw + zb);
#pragma unroll
            for (int m = 0; m < DB / 2 + 1; m++) {
                const float2 tt = trow[m];
                rw.t[m] = (v2f){tt.x, tt.y};
            }
        };
        auto term2 = [&](v2f &acc, float sc, v2f t) {
            const v2f s2 = (v2f){sc, sc};
            if (CMP == T_DOT) {
                acc = __builtin_elementwise_fma(s2, t, acc);
            } else if (CMP == T_SSD) {
                const v2f df = s2 - t;
                acc = __builtin_elementwise_fma(df, df, acc);
            } else {
                acc += __builtin_elementwise_abs(s2 - t);
            }
        };
        auto term1 = [&](float &acc, float sc, float t) {
            if (CMP == T_DOT) acc = fmaf(sc, t, acc);
            else if (CMP == T_SSD) acc = fmaf(sc - t, sc - t, acc);
            else acc += fabsf(sc - t);
        };
        auto add_row = [&](const Row &rw) {
            const float sa = SIGN > 0 ? rw.s.x : rw.s.y, sb = SIGN > 0 ? rw.s.y : rw.s.x; // the column that meets t[q], the one that meets t[q + 1]
#pragma unroll
            for (int m = 0; m < DB / 2; m++) term2(A[m], sa, rw.t[m]);
#pragma unroll
            for (int m = 0; m < DB / 2 - 1; m++) term2(Bp[m], sb, rw.t[m + 1]);
            term1(b_first, sb, rw.t[0].y);
            term1(b_last, sb, rw.t[DB / 2].x);
            if constexpr (FOLD) {
                static_assert(!FOLD || SIGN > 0, "2-D volumes walk the target to the right");
                term1(x17a, sa, rw.t[DB / 2].x);
                term1(x17b, sb, rw.t[DB / 2].y);
            }
        };
        // two rows in flight: the LDS reads of the next row are issued before the arithmetic of the current one (the scheduling
        // barriers keep the compiler from sinking each read to its first use, which made every read a round trip of its own)
        Row ra, rb;
        load_row(ra, 0);
        int k = 0;
        for (; k + 1 < v; k += 2) {
            load_row(rb, k + 1);
            __builtin_amdgcn_sched_barrier(0);
            add_row(ra);
            __builtin_amdgcn_sched_barrier(0);
            if (k + 2 < v) load_row(ra, k + 2);
            __builtin_amdgcn_sched_barrier(0);
            add_row(rb);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (k < v) add_row(ra);
        float V0[DB], V1[DB]; // of the even, of the odd column
#pragma unroll
        for (int q = 0; q < DB; q++) {
            const float va = A[q / 2][q & 1];
            const float vb = q == 0 ? b_first : (q == DB - 1 ? b_last : Bp[(q - 1) / 2][(q - 1) & 1]);
            V0[q] = SIGN > 0 ? va : vb;
            V1[q] = SIGN > 0 ? vb : va;
        }
        // window sums: r[e][q] of the window that ENDS at column 2 u + e
        float r[2][DB];
#pragma unroll
        for (int q = 0; q < DB; q++) {
            float P0 = V0[q], P1 = V1[q], P0m = P0, P1m = P1;
#pragma unroll
            for (int l = 1; l <= HR; l++) {
                if (l == HR) {
                    P0m = P0;
                    P1m = P1;
                }
                P0 = V0[q] + lane_below(P0);
                P1 = V1[q] + lane_below(P1);
            }
            r[0][q] = P0 + lane_below(P1m);
            r[1][q] = P1 + P0m;
        }
        float r17[2] = {0.0f, 0.0f};
        if (FOLD && ext_b) {
            float P0 = x17a, P1 = x17b, P0m = P0, P1m = P1;
#pragma unroll
            for (int l = 1; l <= HR; l++) {
                if (l == HR) {
                    P0m = P0;
                    P1m = P1;
                }
                P0 = x17a + lane_below(P0);
                P1 = x17b + lane_below(P1);
            }
            r17[0] = P0 + lane_below(P1m);
            r17[1] = P1 + P0m;
        }
#pragma unroll
        for (int e = 0; e < 2; e++) {
            // (every lane goes through the exchange below; lanes without a pixel carry values nobody stores)
            // centre column of the window in the tile: 2 u + e - HR; its target statistics sit at the matching tile index
            const int zc0 = (SIGN > 0 ? 2 * u + e - HR : CS_COLS - 1 - (2 * u + e - HR)) + d0;
            // branch-free: the corrections read staged statistics (zeros outside the image), the select replaces what does not exist
            const int jt0 = pe[e] + SIGN * (disp_lower + d0); // target column of the block's first disparity
            const bool whole = d0 + DB <= D;                    // (wave uniform)
            float o_[DB];
            // Nearly every block lies inside the target image for all of its sixteen disparities and every pixel of the wave (wave uniform):
            // then no cost is the one against the all-zero vector and the test per voxel -- a compare, a select and the mask arithmetic, a
            // third of the block's vector instructions outside the window sums -- is one ballot per block.
            const int jt_last = jt0 + SIGN * (DB - 1 + ((FOLD && ext_b) ? 1 : 0));
            const bool interior = trow_in_b && whole && __all(!live[e] || ((unsigned)jt0 < (unsigned)Wt && (unsigned)jt_last < (unsigned)Wt));
#pragma unroll
            for (int q = 0; q < DB; q++) {
                float x = r[e][q];
                if (ZM && CMP == T_DOT) x -= Ff * (ms[e] - c0) * (tmean_b[zc0 + q] - c0);
                if (ZM && CMP == T_SSD) {
                    const float dm = ms[e] - tmean_b[zc0 + q];
                    x -= Ff * dm * dm;
                }
                if (NRM) x *= inv_ns[e] * tinv_b[zc0 + q];
                o_[q] = x;
            }
            if (!interior) {
#pragma unroll
                for (int q = 0; q < DB; q++) {
                    const bool there = trow_in_b && (unsigned)(jt0 + SIGN * q) < (unsigned)Wt && (whole || d0 + q < D);
                    o_[q] = there ? o_[q] : zc[e]; // no target pixel: the cost against the all-zero vector (cross_correlations.h:235)
                }
            }
            // The reductions must cost next to nothing per voxel (the kernel is bound by vector issue: a first form with explicit NaN and
            // candidate tests took 7 instructions per voxel and made it 65 % longer).
            if constexpr (RED == 1) { // winner so far: the sequential scan's rule on this wave's disparities, in increasing order
                // best starts at +inf (-inf for scores): `x <= best` holds for every non-NaN x until something smaller came, never for a
                // NaN, and an equal value later in the scan takes over (ties to the larger index): compare, min / max, select -- 3 per voxel.
                if (!whole) { // (wave uniform; the last block of a range that is no multiple of 16) indices past the range never win
#pragma unroll
                    for (int q = 0; q < DB; q++) o_[q] = d0 + q < D ? o_[q] : __uint_as_float(0x7FC00000u);
                }
                if (d0 == 0) nan_at_0[e] = isnan(o_[0]);
                if (rd.score) {
#pragma unroll
                    for (int q = 0; q < DB; q++) {
                        red_d[e] = o_[q] >= red_a[e] ? d0 + q : red_d[e];
                        red_a[e] = fmaxf(red_a[e], o_[q]);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < DB; q++) {
                        red_d[e] = o_[q] <= red_a[e] ? d0 + q : red_d[e];
                        red_a[e] = fminf(red_a[e], o_[q]);
                    }
                }
            } else if constexpr (RED == 2) { // regional minima: j + d < W / j + d >= W (sgm.h:285-289 applies Pout by the INDEX)
                // v_min ignores NaN and +inf never lowers a minimum; any infinity (the only way to a -inf) drives amax to +inf, which raises
                // the "outside the regime" bit and sends the aggregation to the kernels that read the volume: no finiteness test per voxel.
                const int first_oob = Ws - pe[e]; // this pixel's first index that looks past the right border
                float mn = INFINITY, mx = 0.0f;
                if (!whole) {
#pragma unroll
                    for (int q = 0; q < DB; q++) o_[q] = d0 + q < D ? o_[q] : __uint_as_float(0x7FC00000u);
                }
#pragma unroll
                for (int q = 0; q < DB; q++) mx = fmaxf(mx, fabsf(o_[q]));
                amax = fmaxf(amax, mx);
                const bool all_in = __all(d0 + DB <= first_oob || !live[e]), all_out = __all(d0 >= first_oob || !live[e]);
                if (all_in || all_out) { // (wave uniform) the usual case: one region for the whole block
#pragma unroll
                    for (int q = 0; q < DB; q++) mn = fminf(mn, o_[q]);
                    if (all_in) red_a[e] = fminf(red_a[e], mn);
                    else red_b[e] = fminf(red_b[e], mn);
                } else {
#pragma unroll
                    for (int q = 0; q < DB; q++) {
                        const bool oob = d0 + q >= first_oob;
                        red_a[e] = fminf(red_a[e], oob ? INFINITY : o_[q]);
                        red_b[e] = fminf(red_b[e], oob ? o_[q] : INFINITY);
                    }
                }
            }
            if (FOLD && ext_b && live[e]) { // the seventeenth cost: the lane's own pixel, one dword
                float x = r17[e];
                if (ZM && CMP == T_DOT) x -= Ff * (ms[e] - c0) * (tmean_b[zc0 + DB] - c0);
                if (ZM && CMP == T_SSD) {
                    const float dm = ms[e] - tmean_b[zc0 + DB];
                    x -= Ff * dm * dm;
                }
                if (NRM) x *= inv_ns[e] * tinv_b[zc0 + DB];
                const bool there = trow_in_b && (unsigned)(jt0 + DB) < (unsigned)Wt;
                cv[((int64_t)i * Ws + pe[e]) * px_stride + out_b + d0 + DB] = there ? x : zc[e];
            }
            if (RED && !rd.store) continue; // (kernel-uniform) nobody wants the volume
            if (whole || FOLD) { // (wave uniform)
                // A lane holds 64 bytes of its pixel's run; stored as they are, every 16-byte piece of a wave's store instruction would go to
                // a different pixel (1 KB apart): requests of 16 bytes, a quarter of what the memory side takes per request (measured: the
                // kernel then runs at 2.4 TB/s whatever the window).  Through LDS instead: lanes 4 a .. 4 a + 3 store the four pieces of
                // pixel 16 j + a, 64 contiguous bytes.
#pragma unroll
                for (int q = 0; q < DB; q += 4) *reinterpret_cast<float4 *>(xpose + u * CS_XP + q) = make_float4(o_[q], o_[q + 1], o_[q + 2], o_[q + 3]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (this wave's own writes; nobody else touches the area)
                const int a = u >> 2, c = u & 3;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int uu = 16 * j + a;                                 // the lane whose pixel this is
                    const int pp = p0 + 2 * (uu - HR) + e;                     // that pixel
                    const float4 piece = *reinterpret_cast<const float4 *>(xpose + uu * CS_XP + 4 * c);
                    float *dst = cv + ((int64_t)i * Ws + pp) * px_stride + out_b + d0 + 4 * c;
                    if (uu >= HR && pp < Ws) {
                        if (whole || d0 + 4 * c + 4 <= D) { // (FOLD: a partial block stores its whole pieces the same way)
                            *reinterpret_cast<CostPiece4 *>(dst) = CostPiece4{piece.x, piece.y, piece.z, piece.w};
                        } else {
                            if (d0 + 4 * c < D) dst[0] = piece.x;
                            if (d0 + 4 * c + 1 < D) dst[1] = piece.y;
                            if (d0 + 4 * c + 2 < D) dst[2] = piece.z;
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (read before the next round overwrites it)
            } else if (live[e]) {
                float *o = cv + ((int64_t)i * Ws + pe[e]) * px_stride + out_b + d0;
#pragma unroll
                for (int q = 0; q < DB; q++)
                    if (d0 + q < D) o[q] = o_[q];
            }
        }
    }
    if constexpr (RED == 0) return;
    // The waves of the block hold a pixel's range between them (disparity block b belongs to wave b mod WAVES): partial results go
    // through this wave's transposition area, wave 0 -- which walked block 0 and knows whether the cost at index 0 is NaN -- combines.
    float *part = xpose; // 64 lanes x 4 floats
    *reinterpret_cast<float4 *>(part + 4 * u) = RED == 1 ? make_float4(red_a[0], __int_as_float(red_d[0]), red_a[1], __int_as_float(red_d[1]))
                                                             : make_float4(red_a[0], red_b[0], red_a[1], red_b[1]);
    if (RED == 2 && rd.flag && __any(amax > rd.big) && u == 0) atomicOr(rd.flag, 2);
    __syncthreads();
    if (wave != 0) return;
    float *base = xpose; // wave 0's own area is the first
    for (int w = 1; w < WAVES; w++) {
        const float4 o = *reinterpret_cast<const float4 *>(base + w * (64 * CS_XP) + 4 * u);
        if (RED == 1) {
            const float ov[2] = {o.x, o.z};
            const int od[2] = {__float_as_int(o.y), __float_as_int(o.w)};
#pragma unroll
            for (int e = 0; e < 2; e++) { // extremum wins, ties go to the larger index (correlation_base.h:441-455)
                const bool take = od[e] >= 0 && (red_d[e] < 0 || (rd.score ? (ov[e] > red_a[e] || (ov[e] == red_a[e] && od[e] > red_d[e]))
                                                                          : (ov[e] < red_a[e] || (ov[e] == red_a[e] && od[e] > red_d[e]))));
                red_a[e] = take ? ov[e] : red_a[e];
                red_d[e] = take ? od[e] : red_d[e];
            }
        } else {
            red_a[0] = fminf(red_a[0], o.x);
            red_b[0] = fminf(red_b[0], o.y);
            red_a[1] = fminf(red_a[1], o.z);
            red_b[1] = fminf(red_b[1], o.w);
        }
    }
#pragma unroll
    for (int e = 0; e < 2; e++) {
        if (!live[e]) continue;
        const int64_t px = (int64_t)i * Ws + pe[e];
        if (RED == 1) {
            const int sel = (nan_at_0[e] || red_d[e] < 0) ? 0 : red_d[e];
            if (rd.idx) rd.idx[px] = sel;
            if (rd.disp) rd.disp[px] = rd.disp_sign * sel + rd.disp_offset;
        } else {
            rd.minima[px] = make_float2(red_a[e], red_b[e]);
        }
    }
}

// waves per block: eight share a staged tile when there are disparity blocks for all of them (two blocks of eight waves fit a CU's LDS
// where three of four would: 16 instead of 12 waves per CU), four otherwise
inline int colsum_waves(int D) { return (D + CS_DB - 1) / CS_DB >= 16 ? 8 : 4; }
inline int colsum_waves(int D, int n_dh) { return n_dh * ((D + CS_DB - 1) / CS_DB) >= 16 ? 8 : 4; }
inline size_t colsum_shmem(int v_r, int D, bool zm, bool nrm, int n_dh = 1, int C = 1) {
    const int v = (2 * v_r + 1) * C, nb = (D + CS_DB - 1) / CS_DB, tw = CS_COLS + nb * CS_DB;
    return (size_t)(v * CS_COLS + (v + (n_dh - 1) * C) * tw + (zm ? n_dh * tw : 0) + (nrm ? n_dh * tw : 0) + colsum_waves(D, n_dh) * 64 * CS_XP) * sizeof(float);
}

template <int CMP, bool ZM, bool NRM, int HR> void launch_colsum_n(svh_context *ctx, const CostVolumeArgs &a, const float *src, const float *tgt, int v_r, int sign,
                                                                   const float *ms, const float *mt, const float *ns, const float *nt, const float *zc, float *cv) {
    const size_t shmem = colsum_shmem(v_r, a.D, ZM, NRM, a.n_dh, a.C);
    ColsumReduce rd{};
    rd.store = 1;
    if (a.reduce && a.reduce->mode && (a.C == 1 || a.reduce->mode == 1) && a.tgt_row_off == 0 && a.out_off == 0 && a.row_count == 0 && a.n_dh == 1) { // (1-D volumes of whole images; colour images: the winner only)
        const CostReduce &r = *a.reduce;
        rd = ColsumReduce{r.mode, r.score ? 1 : 0, r.store ? 1 : 0, r.idx, r.disp, r.disp_sign, r.disp_offset, reinterpret_cast<float2 *>(r.minima), r.flag, r.big};
        a.reduce->done = true;
    }
    dim3 grid(ceil_div(a.Ws, CS_COLS - 2 * HR), a.row_count ? a.row_count : a.H);
#define SVH_CS_LAUNCH_RF(SG, WV, RD, FD)                                                                                                            \
    do {                                                                                                                                           \
        static int big_lds[64] = {}; /* (per instantiation and device) more than the default 64 KiB of dynamic LDS */                              \
        if (shmem > 64 * 1024 && !__atomic_load_n(&big_lds[ctx->device & 63], __ATOMIC_ACQUIRE)) {                                                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&cost_volume_colsum_kernel<CMP, ZM, NRM, HR, SG, WV, RD, FD>),                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                                     \
            __atomic_store_n(&big_lds[ctx->device & 63], 1, __ATOMIC_RELEASE);                                                                     \
        }                                                                                                                                          \
        SVH_LAUNCH(ctx, "cost_volume_tiled", (cost_volume_colsum_kernel<CMP, ZM, NRM, HR, SG, WV, RD, FD>), grid, 64 * WV, shmem, src, tgt, a.H,   \
                   a.Ws, a.Wt, v_r, a.D, a.disp_lower, ms, mt, ns, nt, zc, a.tgt_row_off, a.px_stride(), a.out_off, cv, a.row_begin, rd, a.n_dh, 1); \
    } while (0)
#define SVH_CS_LAUNCH_MCRF(SG, WV, RD, FD)                                                                                                          \
    do {                                                                                                                                           \
        static int big_lds[64] = {};                                                                                                               \
        if (shmem > 64 * 1024 && !__atomic_load_n(&big_lds[ctx->device & 63], __ATOMIC_ACQUIRE)) {                                                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&cost_volume_colsum_kernel<CMP, ZM, NRM, HR, SG, WV, RD, FD, true>),          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                                     \
            __atomic_store_n(&big_lds[ctx->device & 63], 1, __ATOMIC_RELEASE);                                                                     \
        }                                                                                                                                          \
        SVH_LAUNCH(ctx, "cost_volume_tiled", (cost_volume_colsum_kernel<CMP, ZM, NRM, HR, SG, WV, RD, FD, true>), grid, 64 * WV, shmem, src, tgt,  \
                   a.H, a.Ws, a.Wt, v_r, a.D, a.disp_lower, ms, mt, ns, nt, zc, a.tgt_row_off, a.px_stride(), a.out_off, cv, a.row_begin, rd,      \
                   a.n_dh, a.C);                                                                                                                   \
    } while (0)
#define SVH_CS_LAUNCH_MCF(SG, WV, FD) SVH_CS_LAUNCH_MCRF(SG, WV, 0, FD)
#define SVH_CS_LAUNCH_MC(SG, WV)                                                                                                                   \
    do { /* (colour images: the plain kernel or the winner) */                                                                                    \
        if (rd.mode == 1) SVH_CS_LAUNCH_MCRF(SG, WV, 1, false);                                                                                    \
        else SVH_CS_LAUNCH_MCRF(SG, WV, 0, false);                                                                                                 \
    } while (0)
#define SVH_CS_LAUNCH_R(SG, WV, RD) SVH_CS_LAUNCH_RF(SG, WV, RD, false)
#define SVH_CS_LAUNCH(SG, WV)                                                                                                                      \
    do {                                                                                                                                           \
        if (rd.mode == 1) SVH_CS_LAUNCH_R(SG, WV, 1);                                                                                              \
        else if (rd.mode == 2) SVH_CS_LAUNCH_R(SG, WV, 2);                                                                                         \
        else SVH_CS_LAUNCH_R(SG, WV, 0);                                                                                                           \
    } while (0)
    if (a.C > 1 && a.n_dh > 1) { // colour images, several vertical offsets of a 2-D volume in one launch (round 5; sign +1 there)
        if (colsum_waves(a.D, a.n_dh) == 8) SVH_CS_LAUNCH_MCF(1, 8, true);
        else SVH_CS_LAUNCH_MCF(1, 4, true);
    } else if (a.C > 1) { // interleaved channels: their own instantiations (the grey kernels stay as they are, register for register)
        if (colsum_waves(a.D) == 8) {
            if (sign > 0) SVH_CS_LAUNCH_MC(1, 8);
            else SVH_CS_LAUNCH_MC(-1, 8);
        } else {
            if (sign > 0) SVH_CS_LAUNCH_MC(1, 4);
            else SVH_CS_LAUNCH_MC(-1, 4);
        }
    } else if (a.n_dh > 1) { // several vertical offsets of a 2-D volume in one launch (sign +1 there, no reductions)
        if (colsum_waves(a.D, a.n_dh) == 8) SVH_CS_LAUNCH_RF(1, 8, 0, true);
        else SVH_CS_LAUNCH_RF(1, 4, 0, true);
    } else if (colsum_waves(a.D) == 8) {
        if (sign > 0) SVH_CS_LAUNCH(1, 8);
        else SVH_CS_LAUNCH(-1, 8);
    } else {
        if (sign > 0) SVH_CS_LAUNCH(1, 4);
        else SVH_CS_LAUNCH(-1, 4);
    }
#undef SVH_CS_LAUNCH
#undef SVH_CS_LAUNCH_R
#undef SVH_CS_LAUNCH_RF
#undef SVH_CS_LAUNCH_MC
#undef SVH_CS_LAUNCH_MCF
#undef SVH_CS_LAUNCH_MCRF
}
template <int CMP, bool ZM, int HR> void launch_colsum(svh_context *ctx, const CostVolumeArgs &a, const float *src, const float *tgt, int v_r, int sign,
                                                       const float *ms, const float *mt, const float *ns, const float *nt, const float *zc, float *cv) {
    if constexpr (CMP == T_DOT) { // (only the products are ever normalised: NCC, ZNCC)
        if (ns) return launch_colsum_n<CMP, ZM, true, HR>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv);
    }
    launch_colsum_n<CMP, ZM, false, HR>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv);
}

template <int CMP, bool ZM, int HR, int DB> void launch_pxlane_db(svh_context *ctx, const CostVolumeArgs &a, const float *src, const float *tgt, int v_r, int sign,
                                                                  const float *ms, const float *mt, const float *ns, const float *nt, const float *zc,
                                                                  float *cv, size_t shmem) {
    dim3 grid(ceil_div(a.Ws, PL_TPX), a.row_count ? a.row_count : a.H);
    if (sign > 0)
        SVH_LAUNCH(ctx, "cost_volume_tiled", (cost_volume_pxlane_kernel<CMP, ZM, HR, DB, 1>), grid, 256, shmem, src, tgt, a.H, a.Ws, a.Wt, v_r, a.D,
                   a.disp_lower, ms, mt, ns, nt, zc, a.tgt_row_off, a.px_stride(), a.out_off, cv, a.row_begin);
    else
        SVH_LAUNCH(ctx, "cost_volume_tiled", (cost_volume_pxlane_kernel<CMP, ZM, HR, DB, -1>), grid, 256, shmem, src, tgt, a.H, a.Ws, a.Wt, v_r, a.D,
                   a.disp_lower, ms, mt, ns, nt, zc, a.tgt_row_off, a.px_stride(), a.out_off, cv, a.row_begin);
}

// disparities per lane and block: 32 for wide windows over long ranges (fewer LDS reads per multiply-add: measured 5 % faster
// for 11x11 at D = 256, slower for 5x5), else 16
inline int pxlane_db(int h_r, int D) { return (D >= 128 && h_r >= 4) ? 32 : 16; }
inline size_t pxlane_shmem(int h_r, int v_r, int D) {
    const int h = 2 * h_r + 1, v = 2 * v_r + 1, db = pxlane_db(h_r, D), nb = (D + db - 1) / db;
    return (size_t)v * ((PL_TPX + h - 1) + (PL_TPX + h - 1 + nb * db - 1)) * sizeof(float);
}

template <int CMP, bool ZM, int HR> void launch_pxlane(svh_context *ctx, const CostVolumeArgs &a, const float *src, const float *tgt, int v_r, int sign,
                                                       const float *ms, const float *mt, const float *ns, const float *nt, const float *zc,
                                                       float *cv, size_t shmem) {
    if constexpr (HR >= 4) {
        if (pxlane_db(HR, a.D) == 32) return launch_pxlane_db<CMP, ZM, HR, 32>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem);
    }
    launch_pxlane_db<CMP, ZM, HR, 16>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem);
}

template <int CMP, bool ZM> bool dispatch_hr(svh_context *ctx, int h_r, const CostVolumeArgs &a, const float *src, const float *tgt, int v_r, int sign,
                                             const float *ms, const float *mt, const float *ns, const float *nt, const float *zc, float *cv,
                                             size_t shmem) {
    if constexpr (!(ZM && CMP == T_SAD)) { // separable terms: column sums shared by the windows that contain the column
        if (ctx->cost_volume_colsum && colsum_shmem(v_r, a.D, ZM, ns != nullptr, a.n_dh, a.C) <= 78 * 1024) {
            switch (h_r) {
            case 1: launch_colsum<CMP, ZM, 1>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv); return true;
            case 2: launch_colsum<CMP, ZM, 2>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv); return true;
            case 3: launch_colsum<CMP, ZM, 3>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv); return true;
            case 4: launch_colsum<CMP, ZM, 4>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv); return true;
            case 5: launch_colsum<CMP, ZM, 5>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv); return true;
            case 6: launch_colsum<CMP, ZM, 6>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv); return true; // (13 and 15 wide: the per-voxel
            case 7: launch_colsum<CMP, ZM, 7>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv); return true; //  kernel took 60 - 80 ms at 1080p x 128)
            default: return false;
            }
        }
    }
    if (shmem > 60 * 1024 || a.n_dh > 1 || a.C > 1) return false; // (only the column-sum kernel folds vertical offsets / takes channels: cost_volume_colsum_applies)
    switch (h_r) {
    case 1: launch_pxlane<CMP, ZM, 1>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); return true;
    case 2: launch_pxlane<CMP, ZM, 2>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); return true;
    case 3: launch_pxlane<CMP, ZM, 3>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); return true;
    case 4: launch_pxlane<CMP, ZM, 4>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); return true;
    case 5: launch_pxlane<CMP, ZM, 5>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); return true;
    case 6: // (13 and 15 wide: only ZSAD, which has no column-sum form, comes here with these)
        if constexpr (ZM && CMP == T_SAD) {
            launch_pxlane<CMP, ZM, 6>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem);
            return true;
        }
        return false;
    case 7:
        if constexpr (ZM && CMP == T_SAD) {
            launch_pxlane<CMP, ZM, 7>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem);
            return true;
        }
        return false;
    default: return false;
    }
}

} // namespace

// the per-comparison entry points (one translation unit each); false: nothing was launched (dispatch_hr)
#define SVH_TILED_ENTRY(NAME)                                                                                                                      \
    bool NAME(svh_context *ctx, int h_r, const CostVolumeArgs &a, const float *src, const float *tgt, int v_r, int sign, const float *ms, const float *mt, \
              const float *ns, const float *nt, const float *zc, float *cv, size_t shmem)
SVH_TILED_ENTRY(cost_tiled_dot);
SVH_TILED_ENTRY(cost_tiled_dot_zm);
SVH_TILED_ENTRY(cost_tiled_ssd);
SVH_TILED_ENTRY(cost_tiled_ssd_zm);
SVH_TILED_ENTRY(cost_tiled_sad);
SVH_TILED_ENTRY(cost_tiled_sad_zm);

} // namespace svh

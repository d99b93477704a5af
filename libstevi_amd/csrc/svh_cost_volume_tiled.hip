// LDS-tiled, register-blocked cost volumes for grey images (CC / NCC / SSD / SAD and their zero-mean variants).
//
// A block owns 64 consecutive pixels of one row and every disparity.  The (2v_r+1)-row source and target tiles are staged in
// LDS once; a lane is a pixel and keeps a block of 16 or 32 consecutive disparities in registers (see the kernel comment).
// No MFMA: there is no shared operand between the per-pixel dot products to contract over.
//
// Arithmetic relative to the reference (cross_correlations.h:416-594, matching_costs.h:59-156): the products /
// differences are accumulated per pixel in the reference's channel order (rows outer, columns inner); the per-element
// normalisations are applied algebraically after the sum,
//     NCC   sum (s/ns)(t/nt)            = (sum s t) / (ns nt)
//     ZCC   sum (s-ms)(t-mt)            = sum (s-c)(t-c) - F (ms-c)(mt-c)        (c: a block constant, keeps magnitudes small)
//     ZSSD  sum ((s-ms)-(t-mt))^2       = sum (s-t)^2 - F (ms-mt)^2
//     ZSAD  sum |(s-ms)-(t-mt)|         = sum |(s-t) - (ms-mt)|
// which changes results by rounding only (a few 1e-7 relative; the north-star tolerance for float costs is 1e-4).
// Means, norms and the all-zero-target cost (target column outside the image, cross_correlations.h:235) come from a
// per-pixel statistics kernel that follows the reference literally.
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

template <int CMP, bool ZM, int HR, int DB, int SIGN>
__global__ void __launch_bounds__(256) cost_volume_pxlane_kernel(const float *__restrict__ src, const float *__restrict__ tgt, int H, int Ws, int Wt, int v_r,
                                                                 int D, int disp_lower, const float *__restrict__ mean_s,
                                                                 const float *__restrict__ mean_t, const float *__restrict__ norm_s,
                                                                 const float *__restrict__ norm_t, const float *__restrict__ zcost, int row_off,
                                                                 int64_t px_stride, int64_t out_off, float *__restrict__ cv) {
    constexpr int h = 2 * HR + 1, NT = h + DB - 1;
    extern __shared__ float lds[];
    const int v = 2 * v_r + 1;
    const int n_blocks = (D + DB - 1) / DB;
    const int sw = PL_TPX + h - 1, tw = PL_TPX + h - 1 + n_blocks * DB - 1;
    float *stile = lds, *ttile = lds + v * sw;
    const int i = blockIdx.y, j0 = blockIdx.x * PL_TPX;
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
        if (d0 + DB <= D && (((px * px_stride + out_off + d0) & 3) == 0)) {
#pragma unroll
            for (int q = 0; q < DB; q += 4) *reinterpret_cast<float4 *>(o + q) = make_float4(r[q], r[q + 1], r[q + 2], r[q + 3]);
        } else {
#pragma unroll
            for (int q = 0; q < DB; q++)
                if (d0 + q < D) o[q] = r[q];
        }
    }
}

template <int CMP, bool ZM, int HR, int DB> void launch_pxlane_db(svh_context *ctx, const CostVolumeArgs &a, const float *src, const float *tgt, int v_r, int sign,
                                                                  const float *ms, const float *mt, const float *ns, const float *nt, const float *zc,
                                                                  float *cv, size_t shmem) {
    dim3 grid(ceil_div(a.Ws, PL_TPX), a.H);
    if (sign > 0)
        SVH_LAUNCH(ctx, "cost_volume_tiled", (cost_volume_pxlane_kernel<CMP, ZM, HR, DB, 1>), grid, 256, shmem, src, tgt, a.H, a.Ws, a.Wt, v_r, a.D,
                   a.disp_lower, ms, mt, ns, nt, zc, a.tgt_row_off, a.px_stride(), a.out_off, cv);
    else
        SVH_LAUNCH(ctx, "cost_volume_tiled", (cost_volume_pxlane_kernel<CMP, ZM, HR, DB, -1>), grid, 256, shmem, src, tgt, a.H, a.Ws, a.Wt, v_r, a.D,
                   a.disp_lower, ms, mt, ns, nt, zc, a.tgt_row_off, a.px_stride(), a.out_off, cv);
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
    switch (h_r) {
    case 1: launch_pxlane<CMP, ZM, 1>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); return true;
    case 2: launch_pxlane<CMP, ZM, 2>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); return true;
    case 3: launch_pxlane<CMP, ZM, 3>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); return true;
    case 4: launch_pxlane<CMP, ZM, 4>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); return true;
    case 5: launch_pxlane<CMP, ZM, 5>(ctx, a, src, tgt, v_r, sign, ms, mt, ns, nt, zc, cv, shmem); return true;
    default: return false;
    }
}

} // namespace

// Returns SVH_OK when the tiled kernel ran, SVH_ERR_UNSUPPORTED (without touching the context error) when the caller must
// use the generic kernel (multi-channel images, windows wider than 11, tiles beyond the LDS budget).
int dev_cost_volume_grey_tiled(svh_context *ctx, Scratch &scr, const CostVolumeArgs &a, ImageDesc src, ImageDesc tgt, int h_r, int v_r, float *cv) {
    if (src.C != 1 || tgt.C != 1 || h_r < 1 || h_r > 5 || func_census(a.func)) return SVH_ERR_UNSUPPORTED;
    const size_t shmem = pxlane_shmem(h_r, v_r, a.D);
    if (shmem > 60 * 1024) return SVH_ERR_UNSUPPORTED;
    if ((int64_t)a.H * a.Ws * a.D == 0) return SVH_OK;
    const bool zm = func_zero_mean(a.func), nrm = func_normalized(a.func);
    const int cmp = (a.func == SVH_SSD || a.func == SVH_ZSSD) ? T_SSD : (a.func == SVH_SAD || a.func == SVH_ZSAD) ? T_SAD : T_DOT;
    const size_t ns_px = (size_t)a.H * a.Ws, nt_px = (size_t)a.H * a.Wt;
    WindowStatsCache local;
    WindowStatsCache &st = a.stats ? *a.stats : local;
    if (!st.ready) {
        Scratch &owner = st.scr ? *st.scr : scr;
        st.zc = owner.get_n<float>(ns_px);
        if (!st.zc) return SVH_ERR_OUT_OF_MEMORY;
        if (zm) {
            st.ms = owner.get_n<float>(ns_px);
            st.mt = owner.get_n<float>(nt_px);
            if (!st.ms || !st.mt) return SVH_ERR_OUT_OF_MEMORY;
        }
        if (nrm) {
            st.ns = owner.get_n<float>(ns_px);
            st.nt = owner.get_n<float>(nt_px);
            if (!st.ns || !st.nt) return SVH_ERR_OUT_OF_MEMORY;
        }
        // means / norms follow the reference literally (channelsMean, channelsNorm, channelsZeroMeanNorm); zc = cost of the source
        // window against the all-zero target vector
        const int cmpk = cmp == T_DOT ? CMP_DOT : cmp == T_SSD ? CMP_SSD : CMP_SAD;
        SVH_TRY(launch_window_stats(ctx, FeatImage{src.data, src.H, src.W, 1, h_r, v_r}, zm, nrm, st.ms, st.ns, st.zc, cmpk));
        if (zm || nrm) SVH_TRY(launch_window_stats(ctx, FeatImage{tgt.data, tgt.H, tgt.W, 1, h_r, v_r}, zm, nrm, st.mt, st.nt, nullptr, cmpk));
        st.ready = true;
    }
    float *ms = st.ms, *mt = st.mt, *ns = st.ns, *nt = st.nt, *zc = st.zc;
    const int sign = a.sign();
    bool ok;
    if (cmp == T_DOT) ok = zm ? dispatch_hr<T_DOT, true>(ctx, h_r, a, src.data, tgt.data, v_r, sign, ms, mt, ns, nt, zc, cv, shmem)
                              : dispatch_hr<T_DOT, false>(ctx, h_r, a, src.data, tgt.data, v_r, sign, ms, mt, ns, nt, zc, cv, shmem);
    else if (cmp == T_SSD) ok = zm ? dispatch_hr<T_SSD, true>(ctx, h_r, a, src.data, tgt.data, v_r, sign, ms, mt, ns, nt, zc, cv, shmem)
                                   : dispatch_hr<T_SSD, false>(ctx, h_r, a, src.data, tgt.data, v_r, sign, ms, mt, ns, nt, zc, cv, shmem);
    else ok = zm ? dispatch_hr<T_SAD, true>(ctx, h_r, a, src.data, tgt.data, v_r, sign, ms, mt, ns, nt, zc, cv, shmem)
                 : dispatch_hr<T_SAD, false>(ctx, h_r, a, src.data, tgt.data, v_r, sign, ms, mt, ns, nt, zc, cv, shmem);
    if (!ok) return SVH_ERR_UNSUPPORTED;
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

} // namespace svh

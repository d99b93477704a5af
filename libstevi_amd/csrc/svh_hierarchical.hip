// Hierarchical (coarse-to-fine) matching: SURVEY.md section 8(f) rank 3.
//
//   averagePoolingDownsample                                   interpolation/downsampling.h:67-178
//   computeGuidedCV<matchFunc>(feat_l, feat_r, guide, r)        correlation/hierarchical.h:74-229
//   hiearchicalTruncatedCostVolume<matchFunc, depth>            correlation/hierarchical.h:232-319
//
// computeGuidedCV searches 2r+1 disparities around twice the bilinearly upsampled coarse estimate, keeps the first strict
// extremum d_r, and re-centres the truncated volume on it (:194-227).  After the re-centring every entry is
//     tcv(i, j, dd) = cmp(src(i,j,:), tgt(i, j + d_r + dirSign (dd - r), :)),
// a pure function of d_r.  Image inputs with r <= 3 take one pass (`guided_fused`: a lane per pixel evaluates the offsets
// d0 - 2r .. d0 + 2r once, picks d_r among the middle ones and copies the re-centred window out); feature volumes, census
// words and larger radii take two: a lane per pixel picks d_r (`guided_select`), then a lane per (pixel, dd) fills the volume
// (`guided_volume`).  All of them accumulate the terms of the literal comparison of svh_compare.h in the same order, so equal
// inputs give bit-equal costs and the strict '<' / '>' winner rule sees exactly the values the volume holds.
#include <climits>
#include <type_traits>

#include "svh_compare.h"
#include "svh_guided_wave.h"
#include "svh_internal.h"

namespace svh {

namespace {

// averagePoolingDownsample as written (downsampling.h:78-114, :131-176): the row loop of the window runs over the
// HORIZONTAL window size, the column offset is derived from the ROW remainder and vice versa; mean over the valid samples.
// (grid slice y = 1: the second image of a pair, same window)
__global__ void downsample_kernel(const float *__restrict__ img, int H, int W, int C, int win_h, int win_v, int Ho, int Wo, float *__restrict__ out,
                                  const float *__restrict__ img_b, int W_b, int Wo_b, float *__restrict__ out_b) {
    if (blockIdx.y == 1) {
        img = img_b;
        W = W_b;
        Wo = Wo_b;
        out = out_b;
    }
    const int64_t n = (int64_t)Ho * Wo * C;
    const int hRem = Ho * win_v - H, vRem = Wo * win_h - W;
    const int initialHOffset = hRem / 2, initialVOffset = vRem / 2;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(e % C);
        const int j = (int)((e / C) % Wo), i = (int)(e / ((int64_t)C * Wo));
        float val = 0.0f;
        int count = 0;
        for (int dv = 0; dv < win_h; dv++) {
            const int p_i = i * win_v - initialVOffset + dv;
            for (int dh = 0; dh < win_h; dh++) {
                const int p_j = j * win_h - initialHOffset + dh;
                if (p_i >= 0 && p_i < H && p_j >= 0 && p_j < W) {
                    val += img[((int64_t)p_i * W + p_j) * C + f];
                    count += 1;
                }
            }
        }
        out[e] = val / (float)count;
    }
}

// Hamming distance between exact words (the target is gathered as uint32 here, no float round trip: hierarchical.h:175-178)
struct WordVolume {
    const uint32_t *w;
    int W, nW;
};
__device__ __forceinline__ float compare_words(const WordVolume &src, const WordVolume &tgt, int i, int j, int jt) {
    const uint32_t *s = src.w + ((int64_t)i * src.W + j) * src.nW;
    const bool t_in = jt >= 0 && jt < tgt.W;
    const uint32_t *t = tgt.w + ((int64_t)i * tgt.W + (t_in ? jt : 0)) * tgt.nW;
    uint32_t score = 0;
    for (int k = 0; k < src.nW; k++) score += __popc(s[k] ^ (t_in ? t[k] : 0u));
    return (float)(uint16_t)score; // hamming_cv_t, matching_costs.h:234
}


template <class CmpFn>
__device__ __forceinline__ void guided_select_body(CmpFn cmp, int H, int Ws, const GuideArgs &g, int32_t *__restrict__ disp) {
    const int64_t npx = (int64_t)H * Ws;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npx; p += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(p % Ws), i = (int)(p / Ws);
        const int d0 = guided_base_disp(g.guide, g.Hg, g.Wg, H, Ws, i, j, g.dirSign);
        float score = g.cost ? INFINITY : -INFINITY;
        int d_r = d0;
        for (int delta_d = -g.radius; delta_d <= g.radius; delta_d++) { // :157-190: strict comparison, the first extremum wins
            const float c = cmp(i, j, j + d0 + delta_d);
            if (g.cost ? (c < score) : (c > score)) {
                score = c;
                d_r = d0 + delta_d;
            }
        }
        disp[p] = g.dirSign * d_r; // :192
    }
}

template <class CmpFn>
__device__ __forceinline__ void guided_volume_body(CmpFn cmp, int H, int Ws, const GuideArgs &g, const int32_t *__restrict__ disp,
                                                   float *__restrict__ tcv) {
    const int T = 2 * g.radius + 1;
    const int64_t n = (int64_t)H * Ws * T;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int dd = (int)(e % T);
        const int64_t p = e / T;
        const int j = (int)(p % Ws), i = (int)(p / Ws);
        const int d_r = g.dirSign * disp[p];
        tcv[e] = cmp(i, j, j + d_r + g.dirSign * (dd - g.radius));
    }
}

template <class A, int CMP, bool ZM, bool NORM>
__global__ void guided_select_kernel(A src, A tgt, const float *__restrict__ ms, const float *__restrict__ ns, const float *__restrict__ mt,
                                     const float *__restrict__ nt, int H, int Ws, int Wt, GuideArgs g, int32_t *__restrict__ disp) {
    guided_select_body([&](int i, int j, int jt) { return compare_features<A, CMP, ZM, NORM>(src, tgt, ms, ns, mt, nt, H, Ws, Wt, i, j, i, jt); }, H, Ws, g,
                       disp);
}
template <class A, int CMP, bool ZM, bool NORM>
__global__ void guided_volume_kernel(A src, A tgt, const float *__restrict__ ms, const float *__restrict__ ns, const float *__restrict__ mt,
                                     const float *__restrict__ nt, int H, int Ws, int Wt, GuideArgs g, const int32_t *__restrict__ disp,
                                     float *__restrict__ tcv) {
    guided_volume_body([&](int i, int j, int jt) { return compare_features<A, CMP, ZM, NORM>(src, tgt, ms, ns, mt, nt, H, Ws, Wt, i, j, i, jt); }, H, Ws, g,
                       disp, tcv);
}
__global__ void guided_select_words_kernel(WordVolume src, WordVolume tgt, int H, int Ws, GuideArgs g, int32_t *__restrict__ disp) {
    guided_select_body([&](int i, int j, int jt) { return compare_words(src, tgt, i, j, jt); }, H, Ws, g, disp);
}
__global__ void guided_volume_words_kernel(WordVolume src, WordVolume tgt, int H, int Ws, GuideArgs g, const int32_t *__restrict__ disp,
                                           float *__restrict__ tcv) {
    guided_volume_body([&](int i, int j, int jt) { return compare_words(src, tgt, i, j, jt); }, H, Ws, g, disp, tcv);
}

// One-pass form for images (the pyramid levels of hiearchicalTruncatedCostVolume) and radii up to 3: a lane evaluates the
// 4R + 1 offsets d0 - 2R .. d0 + 2R of its pixel in one walk over the window, sharing the processed source sample between them;
// the winner is searched over the middle 2R + 1 (strict comparison in increasing offset, as the reference does), and the
// re-centred volume d_r - R .. d_r + R always lies inside the evaluated span.  The per-offset sums see exactly the terms, in
// exactly the order, of compare_features, so the result equals the two-pass form bit for bit.
template <int CMP, bool ZM, bool NORM, int R>
__global__ void __launch_bounds__(256) guided_fused_kernel(FeatImage src, FeatImage tgt, const float *__restrict__ mean_s, const float *__restrict__ norm_s,
                                                           const float *__restrict__ mean_t, const float *__restrict__ norm_t, int H, int Ws, int Wt,
                                                           GuideArgs g, int32_t *__restrict__ disp, float *__restrict__ tcv) {
    const int64_t npx = (int64_t)H * Ws;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npx; p += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(p % Ws), i = (int)(p / Ws);
        const int d0 = guided_base_disp(g.guide, g.Hg, g.Wg, H, Ws, i, j, g.dirSign);
        guided_fused_px<CMP, ZM, NORM, R>(src, tgt, mean_s, norm_s, mean_t, norm_t, H, Ws, Wt, g, disp, tcv, p, i, j, d0);
    }
}

// The same for grey images with the processed target features SHARED by a block (round 4f).  The per-lane walk divides every target sample by
// its window's norm once per (pixel, offset) -- 4R + 1 times more often than the samples exist, and the true division the reference's
// normalised features ask for (cross_correlations.h:540) is ten instructions -- and waits for a global load per window sample: 0.82 of the
// 1.27 ms of the reference's own 1080p benchmark row.  A block owns 256 consecutive pixels of a row; the target windows its pixels look at
// are centred in one span of columns (the guide is smooth nearly everywhere).  The raw window rows of both images are staged in LDS once;
// then, window row by window row, every centre of the span has its 2 h_r + 1 samples processed ONCE (mean subtracted, divided by the norm)
// into a double-buffered strip, and a lane's 4R + 1 sums read them from there.  Same operations on the same operands in the same order
// as the per-lane walk: same bits.  A block whose guides spread wider than the staged span takes the per-lane walk.
// (First form: all F processed samples of a centre staged at once, 60 KB per block, samples straight from global memory: two blocks per CU
// waiting on a load per sample -- 0.93 ms, slower than the walk it replaced.)
template <int CMP, bool ZM, bool NORM, int R, int HR>
__global__ void __launch_bounds__(256) guided_shared_kernel(FeatImage src, FeatImage tgt, const float *__restrict__ mean_s, const float *__restrict__ norm_s,
                                                            const float *__restrict__ mean_t, const float *__restrict__ norm_t, int H, int Ws, int Wt,
                                                            GuideArgs g, int32_t *__restrict__ disp, float *__restrict__ tcv, int span_max) {
    constexpr int NC = 4 * R + 1, T = 2 * R + 1, h = 2 * HR + 1, SW = 256 + h - 1;
    extern __shared__ float lds[];
    __shared__ int red_lo[4], red_hi[4];
    const int i = blockIdx.y, j0 = blockIdx.x * 256, j = j0 + threadIdx.x;
    const int v = 2 * src.v_r + 1, TW = span_max + h - 1;
    float *stile = lds, *ttile = stile + v * SW, *feat = ttile + v * TW; // raw rows of both images; processed strip, two buffers of span_max x h
    const bool px = j < Ws;
    const int64_t p = (int64_t)i * Ws + (px ? j : Ws - 1);
    const int d0 = px ? guided_base_disp(g.guide, g.Hg, g.Wg, H, Ws, i, j, g.dirSign) : 0;
    // the span of target centres the block's pixels look at: [lo, hi]
    int lo = px ? j + d0 - 2 * R : INT_MAX, hi = px ? j + d0 + 2 * R : INT_MIN;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off));
        hi = max(hi, __shfl_xor(hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
        red_lo[threadIdx.x >> 6] = lo;
        red_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    lo = min(min(red_lo[0], red_lo[1]), min(red_lo[2], red_lo[3]));
    hi = max(max(red_hi[0], red_hi[1]), max(red_hi[2], red_hi[3]));
    const int64_t span64 = (int64_t)hi - lo + 1;
    if (span64 > span_max) { // (block uniform)
        if (px) guided_fused_px<CMP, ZM, NORM, R>(src, tgt, mean_s, norm_s, mean_t, norm_t, H, Ws, Wt, g, disp, tcv, p, i, j, d0);
        return;
    }
    const int span = (int)span64;
    // raw rows: a sample outside its image is 0 (before it is processed).  Eight window rows' loads are issued before the first is stored (a
    // row-by-row loop waits for memory once per row, and on the small grids this form is kept for a block is as long as its waits); a load
    // outside the image reads a clamped address and is replaced by 0
    for (int k0 = 0; k0 < v; k0 += 8) {
        float got_s[8][2], got_t[8][2];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int ii = i - src.v_r + k0 + r;
            const bool row_in = k0 + r < v && ii >= 0 && ii < H; // (block uniform)
            const int64_t ic = min(max(ii, 0), H - 1);
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int x = threadIdx.x + 256 * u, jj = j0 - HR + x, jc = lo - HR + x;
                const float gs = src.img[ic * Ws + min(max(jj, 0), Ws - 1)], gt = tgt.img[ic * Wt + min(max(jc, 0), Wt - 1)];
                got_s[r][u] = (row_in && jj >= 0 && jj < Ws) ? gs : 0.0f;
                got_t[r][u] = (row_in && jc >= 0 && jc < Wt) ? gt : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int x = threadIdx.x + 256 * u;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                if (k0 + r < v && x < SW) stile[(k0 + r) * SW + x] = got_s[r][u];
                if (k0 + r < v && x < span + h - 1) ttile[(k0 + r) * TW + x] = got_t[r][u];
            }
        }
    }
    // the centres this thread processes (span_max <= 512): a centre outside the image is the all-zero vector (hierarchical.h:175-178)
    float mt[2], nt[2];
    bool tin[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int jt = lo + threadIdx.x + 256 * u;
        tin[u] = (int)threadIdx.x + 256 * u < span && jt >= 0 && jt < Wt;
        const int64_t pt = (int64_t)i * Wt + (tin[u] ? jt : 0);
        mt[u] = (ZM && tin[u]) ? mean_t[pt] : 0.0f;
        nt[u] = (NORM && tin[u]) ? norm_t[pt] : 1.0f;
    }
    const float ms = ZM ? mean_s[p] : 0.0f, ns = NORM ? norm_s[p] : 1.0f;
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = 0.0f;
    const int qb = px ? j + d0 - 2 * R - lo : 0; // this pixel's first centre in the span
    __syncthreads();
    for (int k = 0; k < v; k++) {
        float *buf = feat + (k & 1) * span_max * h;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int q = threadIdx.x + 256 * u;
            if (q < span) {
#pragma unroll
                for (int l = 0; l < h; l++) {
                    float t = 0.0f;
                    if (tin[u]) {
                        t = ttile[k * TW + q + l];
                        if (ZM) t = t - mt[u];
                        if (NORM) t = t / nt[u];
                    }
                    buf[q * h + l] = t;
                }
            }
        }
        __syncthreads(); // (the strip of row k is complete; the other buffer was last read before the previous barrier)
#pragma unroll
        for (int l = 0; l < h; l++) {
            float s = stile[k * SW + threadIdx.x + l];
            if (ZM) s = s - ms;
            if (NORM) s = s / ns;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const float t = buf[(qb + c) * h + l];
                if (CMP == CMP_DOT) {
                    acc[c] += s * t;
                } else if (CMP == CMP_SSD) {
                    const float tmp = s - t;
                    acc[c] += tmp * tmp;
                } else {
                    acc[c] += fabsf(s - t);
                }
            }
        }
    }
    if (!px) return;
    float score = g.cost ? INFINITY : -INFINITY;
    int best = 0;
#pragma unroll
    for (int c = R; c <= 3 * R; c++) {
        if (g.cost ? (acc[c] < score) : (acc[c] > score)) {
            score = acc[c];
            best = c - 2 * R;
        }
    }
    disp[p] = g.dirSign * (d0 + best);
#pragma unroll
    for (int dd = 0; dd < T; dd++) {
        const int want = best + g.dirSign * (dd - R) + 2 * R;
        float val = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (c == want) val = acc[c];
        tcv[p * T + dd] = val;
    }
}

template <int CMP, bool ZM, bool NORM, int HR>
bool launch_guided_shared(svh_context *ctx, FeatImage src, FeatImage tgt, const float *ms, const float *ns, const float *mt, const float *nt, int H, int Ws,
                          int Wt, const GuideArgs &g, int32_t *disp, float *tcv) {
    // the staged span: a block's 256 pixels + the offsets + room for the guide to move (at most 512 centres: two per thread)
    constexpr int h = 2 * HR + 1, span_max = 384;
    const int v = 2 * src.v_r + 1;
    const size_t shmem = (size_t)(v * (256 + h - 1) + v * (span_max + h - 1) + 2 * span_max * h) * sizeof(float);
    if (shmem > 60 * 1024) return false;
    const dim3 grid(ceil_div(Ws, 256), H);
    switch (g.radius) {
    case 1: SVH_LAUNCH(ctx, "guided_fused", (guided_shared_kernel<CMP, ZM, NORM, 1, HR>), grid, 256, shmem, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv, span_max); return true;
    case 2: SVH_LAUNCH(ctx, "guided_fused", (guided_shared_kernel<CMP, ZM, NORM, 2, HR>), grid, 256, shmem, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv, span_max); return true;
    case 3: SVH_LAUNCH(ctx, "guided_fused", (guided_shared_kernel<CMP, ZM, NORM, 3, HR>), grid, 256, shmem, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv, span_max); return true;
    default: return false;
    }
}

template <int CMP, bool ZM, bool NORM>
bool launch_guided_fused(svh_context *ctx, FeatImage src, FeatImage tgt, const float *ms, const float *ns, const float *mt, const float *nt, int H, int Ws,
                         int Wt, const GuideArgs &g, int32_t *disp, float *tcv) {
    // grey images, windows up to 7 wide: the processed target features shared by a block; otherwise every lane on its own
    // RGB images: the wave form at every size (the alternative is the per-pixel walk; the blocks below take grey images only)
    if (ctx->guided_shared && src.C == 3 && tgt.C == 3 && ctx->guided_shared != 2 &&
        launch_guided_wave(ctx, CMP, ZM, NORM, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv))
        return true;
    if (ctx->guided_shared && src.C == 1 && tgt.C == 1 && H <= 65535 && src.h_r == tgt.h_r && src.v_r == tgt.v_r) {
        bool ran = false;
        // 1 (the default): a wave per 64 pixels where the grid fills the chip several times over (a wave per SIMD is 1 024 waves) and runs at the
        // rate of its instructions; a smaller grid -- 640 x 480, the coarse levels of a pyramid -- is as long as its slowest block, and
        // the blocks of four waves below with their wider span are shorter (measured crossover: 960 x 540).  3: a wave per 64 pixels always.
        const bool waves_fill_the_chip = (int64_t)ceil_div(Ws, 64) * H >= 8192;
        if (((ctx->guided_shared == 1 && waves_fill_the_chip) || ctx->guided_shared == 3) &&
            launch_guided_wave(ctx, CMP, ZM, NORM, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv))
            return true;
        switch (src.h_r) { // 2: a block of four waves per 256 pixels (round 4)
        case 1: ran = launch_guided_shared<CMP, ZM, NORM, 1>(ctx, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv); break;
        case 2: ran = launch_guided_shared<CMP, ZM, NORM, 2>(ctx, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv); break;
        case 3: ran = launch_guided_shared<CMP, ZM, NORM, 3>(ctx, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv); break;
        default: break;
        }
        if (ran) return true;
    }
    const int grid = grid_for((int64_t)H * Ws, 256, 16384);
    switch (g.radius) {
    case 1: SVH_LAUNCH(ctx, "guided_fused", (guided_fused_kernel<CMP, ZM, NORM, 1>), grid, 256, 0, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv); return true;
    case 2: SVH_LAUNCH(ctx, "guided_fused", (guided_fused_kernel<CMP, ZM, NORM, 2>), grid, 256, 0, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv); return true;
    case 3: SVH_LAUNCH(ctx, "guided_fused", (guided_fused_kernel<CMP, ZM, NORM, 3>), grid, 256, 0, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv); return true;
    default: return false;
    }
}

template <class A, int CMP, bool ZM, bool NORM>
int launch_guided(svh_context *ctx, A src, A tgt, const float *ms, const float *ns, const float *mt, const float *nt, int H, int Ws, int Wt,
                  const GuideArgs &g, int32_t *disp, float *tcv) {
    const int64_t npx = (int64_t)H * Ws;
    if constexpr (std::is_same_v<A, FeatImage>) {
        if (launch_guided_fused<CMP, ZM, NORM>(ctx, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv)) {
            SVH_CHECK_LAUNCH(ctx);
            return SVH_OK;
        }
    }
    SVH_LAUNCH(ctx, "guided_select", (guided_select_kernel<A, CMP, ZM, NORM>), grid_for(npx, 256, 16384), 256, 0, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g,
               disp);
    SVH_LAUNCH(ctx, "guided_volume", (guided_volume_kernel<A, CMP, ZM, NORM>), grid_for(npx * (2 * g.radius + 1), 256, 65536), 256, 0, src, tgt, ms, ns, mt,
               nt, H, Ws, Wt, g, disp, tcv);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

template <class A>
int guided_float(svh_context *ctx, Scratch &scr, int func, bool processed, A src, A tgt, int H, int Ws, int Wt, const GuideArgs &g, int32_t *disp,
                 float *tcv) {
    // `processed`: the features already went through getFeatureVolumeForMatchFunc (feature-volume entry point), the raw
    // comparison is all that is left; otherwise means / norms are applied on the fly like in the cost-volume kernels
    const bool zm = !processed && func_zero_mean(func), nrm = !processed && func_normalized(func);
    float *ms = nullptr, *ns = nullptr, *mt = nullptr, *nt = nullptr;
    if (zm) {
        ms = scr.get_n<float>((size_t)H * Ws);
        mt = scr.get_n<float>((size_t)H * Wt);
        if (!ms || !mt) return SVH_ERR_OUT_OF_MEMORY;
    }
    if (nrm) {
        ns = scr.get_n<float>((size_t)H * Ws);
        nt = scr.get_n<float>((size_t)H * Wt);
        if (!ns || !nt) return SVH_ERR_OUT_OF_MEMORY;
    }
    if (zm || nrm) {
        if constexpr (std::is_same_v<A, FeatImage>) {
            SVH_TRY(launch_stats_pair(ctx, src, tgt, zm, nrm, ms, ns, mt, nt)); // (both images in one launch)
        } else {
            SVH_TRY(launch_stats(ctx, src, H, Ws, zm, nrm, ms, ns));
            SVH_TRY(launch_stats(ctx, tgt, H, Wt, zm, nrm, mt, nt));
        }
    }
#define SVH_GUIDED(CMPV)                                                                                                     \
    do {                                                                                                                     \
        if (zm && nrm) return launch_guided<A, CMPV, true, true>(ctx, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv);    \
        if (zm) return launch_guided<A, CMPV, true, false>(ctx, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv);          \
        if (nrm) return launch_guided<A, CMPV, false, true>(ctx, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv);         \
        return launch_guided<A, CMPV, false, false>(ctx, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv);                 \
    } while (0)
    switch (func) {
    case SVH_CC: case SVH_NCC: case SVH_ZCC: case SVH_ZNCC: SVH_GUIDED(CMP_DOT);
    case SVH_SSD: case SVH_ZSSD: SVH_GUIDED(CMP_SSD);
    case SVH_SAD: case SVH_ZSAD: SVH_GUIDED(CMP_SAD);
    default: return fail(ctx, SVH_ERR_UNSUPPORTED, "matching function %d is not available on the GPU path", func);
    }
#undef SVH_GUIDED
}

int guided_words(svh_context *ctx, WordVolume src, WordVolume tgt, int H, int Ws, const GuideArgs &g, int32_t *disp, float *tcv) {
    const int64_t npx = (int64_t)H * Ws;
    SVH_LAUNCH(ctx, "guided_select", guided_select_words_kernel, grid_for(npx, 256, 16384), 256, 0, src, tgt, H, Ws, g, disp);
    SVH_LAUNCH(ctx, "guided_volume", guided_volume_words_kernel, grid_for(npx * (2 * g.radius + 1), 256, 65536), 256, 0, src, tgt, H, Ws, g, disp, tcv);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

int dev_downsample(svh_context *ctx, const float *img, int H, int W, int C, int win_h, int win_v, float *out) {
    const int Ho = (H + win_v - 1) / win_v, Wo = (W + win_h - 1) / win_h;
    const int64_t n = (int64_t)Ho * Wo * C;
    if (n == 0) return SVH_OK;
    SVH_LAUNCH(ctx, "average_pooling_downsample", downsample_kernel, grid_for(n, 256, 16384), 256, 0, img, H, W, C, win_h, win_v, Ho, Wo, out, img, W, Wo, out);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

// the two images of a pair (same height and channels) by halves in one launch
int dev_downsample_pair(svh_context *ctx, const float *img_a, int W_a, const float *img_b, int W_b, int H, int C, float *out_a, float *out_b) {
    const int Ho = (H + 1) / 2, Wo_a = (W_a + 1) / 2, Wo_b = (W_b + 1) / 2;
    const int64_t n = (int64_t)Ho * std::max(Wo_a, Wo_b) * C;
    if (n == 0) return SVH_OK;
    SVH_LAUNCH(ctx, "average_pooling_downsample", downsample_kernel, dim3(grid_for(n, 256, 16384), 2), 256, 0, img_a, H, W_a, C, 2, 2, Ho, Wo_a, out_a, img_b, W_b, Wo_b,
               out_b);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

// computeGuidedCV on the feature volumes of two images (unfold + getFeatureVolumeForMatchFunc evaluated on the fly)
int dev_guided_from_images(svh_context *ctx, Scratch &scr, int func, int ddir, ImageDesc img_l, ImageDesc img_r, int h_r, int v_r, const int32_t *guide,
                           int Hg, int Wg, int radius, float *tcv, int32_t *disp) {
    const bool r2l = ddir == SVH_RIGHT_TO_LEFT;
    const ImageDesc src = r2l ? img_r : img_l, tgt = r2l ? img_l : img_r;
    const GuideArgs g{guide, Hg, Wg, radius, r2l ? 1 : -1, func_strategy(func) == SVH_COST};
    if ((int64_t)src.H * src.W == 0) return SVH_OK;
    if (func_census(func)) {
        const int F = (2 * h_r + 1) * (2 * v_r + 1) * src.C;
        const int nWw = census_words_written(F); // the never-written trailing word is 0 on both sides (rule E1): it adds nothing
        if (nWw == 0) { // 3x3 and 5x5 grey windows: every distance is 0
            WordVolume none{nullptr, src.W, 0}, nonet{nullptr, tgt.W, 0};
            return guided_words(ctx, none, nonet, src.H, src.W, g, disp, tcv);
        }
        uint32_t *sw = scr.get_n<uint32_t>((size_t)src.H * src.W * nWw), *tw = scr.get_n<uint32_t>((size_t)tgt.H * tgt.W * nWw);
        if (!sw || !tw) return SVH_ERR_OUT_OF_MEMORY;
        SVH_TRY(dev_census_from_image(ctx, src, h_r, v_r, h_r, v_r, src.H, src.W, nWw, false, sw));
        SVH_TRY(dev_census_from_image(ctx, tgt, h_r, v_r, h_r, v_r, tgt.H, tgt.W, nWw, false, tw));
        return guided_words(ctx, WordVolume{sw, src.W, nWw}, WordVolume{tw, tgt.W, nWw}, src.H, src.W, g, disp, tcv);
    }
    return guided_float(ctx, scr, func, false, FeatImage{src.data, src.H, src.W, src.C, h_r, v_r}, FeatImage{tgt.data, tgt.H, tgt.W, tgt.C, h_r, v_r}, src.H,
                        src.W, tgt.W, g, disp, tcv);
}

// hiearchicalTruncatedCostVolume, hierarchical.h:232-294; radii arrays hold depth + 1 entries, coarsest level first
int dev_hierarchical(svh_context *ctx, Scratch &scr, int func, int ddir, int depth, ImageDesc img_l, ImageDesc img_r, const int32_t *h_radii,
                     const int32_t *v_radii, int disp_width, int radius, float *tcv, int32_t *disp) {
    const bool r2l = ddir == SVH_RIGHT_TO_LEFT;
    const int Hd = (img_l.H + 1) / 2, Wld = (img_l.W + 1) / 2, Wrd = (img_r.W + 1) / 2, C = img_l.C;
    float *dl = scr.get_n<float>((size_t)Hd * Wld * C), *dr = scr.get_n<float>((size_t)Hd * Wrd * C);
    if (!dl || !dr) return SVH_ERR_OUT_OF_MEMORY;
    if (img_l.H == img_r.H) {
        SVH_TRY(dev_downsample_pair(ctx, img_l.data, img_l.W, img_r.data, img_r.W, img_l.H, C, dl, dr)); // :248-249
    } else {
        SVH_TRY(dev_downsample(ctx, img_l.data, img_l.H, img_l.W, C, 2, 2, dl));
        SVH_TRY(dev_downsample(ctx, img_r.data, img_r.H, img_r.W, C, 2, 2, dr));
    }
    const ImageDesc sl{dl, Hd, Wld, C}, sr{dr, Hd, Wrd, C};
    const int Wsd = r2l ? Wrd : Wld;
    int32_t *guide = scr.get_n<int32_t>((size_t)Hd * Wsd);
    if (!guide) return SVH_ERR_OUT_OF_MEMORY;
    if (depth == 1) { // full search at the coarsest level, :253-260
        const int D0 = (disp_width + 1) / 2;
        CostVolumeArgs a{func, ddir, Hd, Wsd, r2l ? Wld : Wrd, 0, D0};
        a.literal = ctx->literal_cost_volumes;
        const ImageDesc csrc = r2l ? sr : sl, ctgt = r2l ? sl : sr;
        // the coarsest level wants the winner of every pixel and nothing else of its volume: where the column-sum kernel runs (grey
        // images, float functions) or the tiled Hamming kernel (census: up to eight words) it picks the winner while it holds the costs and
        // the volume is never written (CostReduce mode 1)
        bool winner_done = false;
        const int census_words = func_census(func) ? census_words_written((2 * h_radii[0] + 1) * (2 * v_radii[0] + 1) * C) : 0;
        const bool hamming_reduces = func_census(func) && census_words >= 1 && census_words <= 8 &&
                                     (size_t)census_words * (2 * 64 + D0 - 1) * sizeof(uint32_t) <= 60 * 1024; // (the tiled Hamming kernel: dev_hamming_volume)
        if (ctx->cost_reduce_fused && (hamming_reduces || (!func_census(func) && cost_volume_colsum_applies(ctx, a, csrc, ctgt, h_radii[0], v_radii[0])))) {
            CostReduce red;
            red.mode = 1;
            red.score = func_strategy(func) != SVH_COST;
            red.idx = guide;
            red.store = false;
            a.reduce = &red;
            SVH_TRY(dev_cost_volume_from_images(ctx, scr, a, csrc, ctgt, h_radii[0], v_radii[0], nullptr));
            winner_done = red.done;
            a.reduce = nullptr;
        }
        if (!winner_done) {
            float *cv = scr.get_n<float>((size_t)Hd * Wsd * D0);
            if (!cv) return SVH_ERR_OUT_OF_MEMORY;
            SVH_TRY(dev_cost_volume_from_images(ctx, scr, a, csrc, ctgt, h_radii[0], v_radii[0], cv));
            SVH_TRY(dev_extract_index(ctx, func_strategy(func), cv, (int64_t)Hd * Wsd, D0, guide, nullptr, 0, 0));
        }
    } else { // :268-286: the radii of the coarser levels are the leading entries
        float *tprev = scr.get_n<float>((size_t)Hd * Wsd * (2 * radius + 1));
        if (!tprev) return SVH_ERR_OUT_OF_MEMORY;
        SVH_TRY(dev_hierarchical(ctx, scr, func, ddir, depth - 1, sl, sr, h_radii, v_radii, (disp_width + 1) / 2, radius, tprev, guide));
    }
    return dev_guided_from_images(ctx, scr, func, ddir, img_l, img_r, h_radii[depth], v_radii[depth], guide, Hd, Wsd, radius, tcv, disp);
}

} // namespace

} // namespace svh

using namespace svh;

extern "C" int svh_average_pooling_downsample(svh_context *ctx, const svh_array *img, int win_horizontal, int win_vertical, svh_array *out) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, img, "img", SVH_F32, 2, 3));
    SVH_TRY(validate(ctx, out, "out", SVH_F32, 2, 3));
    if (win_horizontal < 1 || win_vertical < 1) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "window sizes must be positive");
    if (out->ndim != img->ndim) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "out must have the rank of img");
    const int H = (int)img->shape[0], W = (int)img->shape[1], C = img->ndim == 3 ? (int)img->shape[2] : 1;
    const int Ho = (H + win_vertical - 1) / win_vertical, Wo = (W + win_horizontal - 1) / win_horizontal; // downsampling.h:79-80
    if (out->shape[0] != Ho || out->shape[1] != Wo || (img->ndim == 3 && out->shape[2] != C))
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "out must have shape (%d,%d[,%d])", Ho, Wo, C);
    Scratch scr(ctx);
    void *di;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *img, &di));
    SVH_TRY(stage_out(ctx, scr, *out, &os));
    SVH_TRY(dev_downsample(ctx, (const float *)di, H, W, C, win_horizontal, win_vertical, (float *)os.dptr));
    return finish_out(ctx, os);
}

static int check_guided_outputs(svh_context *ctx, int H, int Ws, int radius, const svh_array *tcv, const svh_array *disp) {
    SVH_TRY(validate(ctx, tcv, "tcv", SVH_F32, 3, 3));
    SVH_TRY(validate(ctx, disp, "disp", SVH_I32, 2, 2));
    if (tcv->shape[0] != H || tcv->shape[1] != Ws || tcv->shape[2] != 2 * radius + 1)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "tcv must have shape (%d,%d,%d)", H, Ws, 2 * radius + 1);
    if (disp->shape[0] != H || disp->shape[1] != Ws) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "disp must have shape (%d,%d)", H, Ws);
    return SVH_OK;
}

extern "C" int svh_guided_cost_volume(svh_context *ctx, int match_func, int disp_direction, const svh_array *feat_l, const svh_array *feat_r,
                                      const svh_array *guide, int32_t upscale_disp_radius, svh_array *tcv, svh_array *disp) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    if (!func_supported(match_func)) return fail(ctx, SVH_ERR_UNSUPPORTED, "matching function %d", match_func);
    const int ft = func_census(match_func) ? SVH_U32 : SVH_F32; // FeatureType, matching_costs.h:742-788
    SVH_TRY(validate(ctx, feat_l, "feat_l", ft, 3, 3));
    SVH_TRY(validate(ctx, feat_r, "feat_r", ft, 3, 3));
    SVH_TRY(validate(ctx, guide, "guide", SVH_I32, 2, 2));
    if (disp_direction != SVH_LEFT_TO_RIGHT && disp_direction != SVH_RIGHT_TO_LEFT) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad disparity direction");
    if (upscale_disp_radius < 0 || upscale_disp_radius > 1024) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad search radius");
    if (feat_l->shape[0] != feat_r->shape[0]) return fail(ctx, SVH_EMPTY_RESULT, "row counts differ"); // hierarchical.h:90-92
    if (feat_l->shape[2] != feat_r->shape[2]) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "feature counts differ");
    if (guide->shape[0] < 2 || guide->shape[1] < 2) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "the guide must be at least 2x2 (bilinear taps, :108-137)");
    const bool r2l = disp_direction == SVH_RIGHT_TO_LEFT;
    const svh_array *src = r2l ? feat_r : feat_l, *tgt = r2l ? feat_l : feat_r;
    const int H = (int)src->shape[0], Ws = (int)src->shape[1], Wt = (int)tgt->shape[1], F = (int)src->shape[2];
    SVH_TRY(check_guided_outputs(ctx, H, Ws, upscale_disp_radius, tcv, disp));
    Scratch scr(ctx);
    void *ds, *dt, *dg;
    OutStage ot, od;
    SVH_TRY(stage_in(ctx, scr, *src, &ds));
    SVH_TRY(stage_in(ctx, scr, *tgt, &dt));
    SVH_TRY(stage_in(ctx, scr, *guide, &dg));
    SVH_TRY(stage_out(ctx, scr, *tcv, &ot));
    SVH_TRY(stage_out(ctx, scr, *disp, &od));
    const GuideArgs g{(const int32_t *)dg, (int)guide->shape[0], (int)guide->shape[1], upscale_disp_radius, r2l ? 1 : -1,
                      func_strategy(match_func) == SVH_COST};
    if ((int64_t)H * Ws > 0) {
        if (func_census(match_func))
            SVH_TRY(guided_words(ctx, WordVolume{(const uint32_t *)ds, Ws, F}, WordVolume{(const uint32_t *)dt, Wt, F}, H, Ws, g, (int32_t *)od.dptr,
                                 (float *)ot.dptr));
        else
            SVH_TRY(guided_float(ctx, scr, match_func, true, FeatVolume{(const float *)ds, Ws, F}, FeatVolume{(const float *)dt, Wt, F}, H, Ws, Wt, g,
                                 (int32_t *)od.dptr, (float *)ot.dptr));
    }
    SVH_TRY(finish_out(ctx, ot));
    return finish_out(ctx, od);
}

extern "C" int svh_hierarchical_truncated_cost_volume(svh_context *ctx, int match_func, int disp_direction, int depth, const svh_array *img_l,
                                                      const svh_array *img_r, const int32_t *h_radii, const int32_t *v_radii, int32_t disp_width,
                                                      int32_t upscale_disp_radius, svh_array *tcv, svh_array *disp) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, img_l, "img_l", SVH_F32, 2, 3));
    SVH_TRY(validate(ctx, img_r, "img_r", SVH_F32, 2, 3));
    if (!func_supported(match_func)) return fail(ctx, SVH_ERR_UNSUPPORTED, "matching function %d", match_func);
    if (disp_direction != SVH_LEFT_TO_RIGHT && disp_direction != SVH_RIGHT_TO_LEFT) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad disparity direction");
    if (depth < 1 || depth > 16) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "depth must be in [1,16] (static_assert depth > 0, hierarchical.h:243)");
    if (!h_radii || !v_radii) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "radii arrays must hold depth + 1 entries");
    for (int k = 0; k <= depth; k++)
        if (h_radii[k] < 0 || v_radii[k] < 0 || h_radii[k] > 255 || v_radii[k] > 255)
            return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "radii must be in [0,255] (uint8_t in the reference)");
    if (disp_width < 1) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "disp_width must be positive");
    if (upscale_disp_radius < 0 || upscale_disp_radius > 1024) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad search radius");
    if (img_l->ndim != img_r->ndim) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "image ranks differ");
    if (img_l->shape[0] != img_r->shape[0]) return fail(ctx, SVH_EMPTY_RESULT, "row counts differ"); // cross_correlations.h:751-753 at the coarsest level
    const int C = img_l->ndim == 3 ? (int)img_l->shape[2] : 1;
    if (img_l->ndim == 3 && img_l->shape[2] != img_r->shape[2]) return fail(ctx, SVH_EMPTY_RESULT, "channel counts differ");
    const bool r2l = disp_direction == SVH_RIGHT_TO_LEFT;
    const int H = (int)img_l->shape[0], Wl = (int)img_l->shape[1], Wr = (int)img_r->shape[1], Ws = r2l ? Wr : Wl;
    // every level must leave the bilinear guide at least 2x2 (hierarchical.h:108-137 reads outside it otherwise)
    if ((H >> depth) < 2 || (Wl >> depth) < 2 || (Wr >> depth) < 2)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "images too small for %d levels", depth);
    for (int k = 0; k <= depth; k++) {
        const int F = (2 * h_radii[k] + 1) * (2 * v_radii[k] + 1) * C;
        if (func_census(match_func) && F <= 1) return fail(ctx, SVH_EMPTY_RESULT, "census needs at least two feature channels");
    }
    SVH_TRY(check_guided_outputs(ctx, H, Ws, upscale_disp_radius, tcv, disp));
    Scratch scr(ctx);
    void *dl, *dr;
    OutStage ot, od;
    SVH_TRY(stage_in(ctx, scr, *img_l, &dl));
    SVH_TRY(stage_in(ctx, scr, *img_r, &dr));
    SVH_TRY(stage_out(ctx, scr, *tcv, &ot));
    SVH_TRY(stage_out(ctx, scr, *disp, &od));
    SVH_TRY(dev_hierarchical(ctx, scr, match_func, disp_direction, depth, ImageDesc{(const float *)dl, H, Wl, C}, ImageDesc{(const float *)dr, H, Wr, C},
                             h_radii, v_radii, disp_width, upscale_disp_radius, (float *)ot.dptr, (int32_t *)od.dptr));
    SVH_TRY(finish_out(ctx, ot));
    return finish_out(ctx, od);
}

// Feature accessors and the literal per-voxel comparison shared by the generic cost-volume kernel (svh_cost_volume.hip) and the
// guided cost volume of hierarchical matching (svh_hierarchical.hip).
#ifndef SVH_COMPARE_H
#define SVH_COMPARE_H

#include "svh_internal.h"

namespace svh {

// ---- feature accessors ----------------------------------------------------------------------------
struct FeatVolume { // dense (H, W, F)
    const float *f;
    int W, F;
    __device__ __forceinline__ int count() const { return F; }
};
struct FeatImage { // unfold on the fly: (H, W, C) image, window (2v_r+1) x (2h_r+1), auto padding
    const float *img;
    int H, W, C, h_r, v_r;
    __device__ __forceinline__ int count() const { return (2 * h_r + 1) * (2 * v_r + 1) * C; }
};

// sequential walk over the channels of pixel (i, j); `valid` false yields an all-zero raw vector
struct VolumeCursor {
    const float *p;
    __device__ __forceinline__ VolumeCursor(const FeatVolume &a, int i, int j) : p(a.f + ((int64_t)i * a.W + j) * a.F) {}
    __device__ __forceinline__ float next() { return *p++; }
};
struct ImageCursor {
    const FeatImage &a;
    int i0, j0, k = 0, l = 0, ch = 0;
    __device__ __forceinline__ ImageCursor(const FeatImage &acc, int i, int j) : a(acc), i0(i - acc.v_r), j0(j - acc.h_r) {}
    __device__ __forceinline__ float next() {
        int ii = i0 + k, jj = j0 + l;
        float v = (ii >= 0 && ii < a.H && jj >= 0 && jj < a.W) ? a.img[((int64_t)ii * a.W + jj) * a.C + ch] : 0.0f;
        if (++ch == a.C) {
            ch = 0;
            if (++l == 2 * a.h_r + 1) {
                l = 0;
                ++k;
            }
        }
        return v;
    }
};
template <class A> struct CursorOf;
template <> struct CursorOf<FeatVolume> { using type = VolumeCursor; };
template <> struct CursorOf<FeatImage> { using type = ImageCursor; };

// ---- per-pixel statistics (A7) ---------------------------------------------------------------------
// mean: channelsMean, correlation_base.h:1100-1136 (sequential sum, then * float(1./F))
// norm: channelsNorm cross_correlations.h:149-191 or channelsZeroMeanNorm :61-104 (sqrtf of sequential sum)
template <class A>
__global__ void stats_kernel(A acc, int H, int W, bool zero_mean, bool normalized, float *__restrict__ mean,
                             float *__restrict__ norm) {
    const int64_t npx = (int64_t)H * W;
    const int F = acc.count();
    const float scale = (float)(1. / (double)(float)F);
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npx; p += (int64_t)gridDim.x * blockDim.x) {
        int j = (int)(p % W), i = (int)(p / W);
        float m = 0.0f;
        if (zero_mean) {
            typename CursorOf<A>::type cur(acc, i, j);
            for (int c = 0; c < F; c++) m += cur.next();
            m *= scale;
            mean[p] = m;
        }
        if (normalized) {
            typename CursorOf<A>::type cur(acc, i, j);
            float n = 0.0f;
            for (int c = 0; c < F; c++) {
                float tmp = cur.next() - m; // m == 0 without zero-mean: x - 0 is exact
                n += tmp * tmp;
            }
            norm[p] = sqrtf(n);
        }
    }
}

template <class A>
inline int launch_stats(svh_context *ctx, A acc, int H, int W, bool zm, bool nrm, float *mean, float *norm) {
    int64_t npx = (int64_t)H * W;
    if (npx == 0) return SVH_OK;
    SVH_LAUNCH(ctx, "window_stats", (stats_kernel<A>), grid_for(npx, 256, 16384), 256, 0, acc, H, W, zm, nrm, mean, norm);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

// The same statistics of a dense (H, W, F) feature volume with the vectors staged in LDS: a block copies the F floats of `ppb` pixels in
// one flat, coalesced run (pixel stride F | 1: the per-pixel walks are then conflict free), a thread per pixel walks its vector exactly as
// stats_kernel does (same sums, same order).  A thread per pixel reading its vector straight from global memory touches a different
// cache line per lane and float: channelsMean of a 1080p x 49 volume took 1.6 ms for 406 MB.  mean_in (optional): the mean is given
// (channelsZeroMeanNorm with its second argument) instead of computed.
template <bool GIVEN> // (a template: the header is included by several translation units)
__global__ void __launch_bounds__(256) stats_staged_kernel(const float *__restrict__ f, int64_t npx, int F, bool zero_mean, bool normalized,
                                                           const float *__restrict__ mean_in, float *__restrict__ mean, float *__restrict__ norm, int ppb) {
    extern __shared__ float st_lds[];
    const int FS = F | 1;
    const int64_t p0 = (int64_t)blockIdx.x * ppb;
    const int n_px = (int)(npx - p0 < ppb ? npx - p0 : ppb), n = n_px * F;
    const float *src = f + p0 * F;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int pl = e / F;
        st_lds[pl * FS + (e - pl * F)] = src[e];
    }
    __syncthreads();
    if ((int)threadIdx.x >= n_px) return;
    const float *v = st_lds + threadIdx.x * FS;
    const int64_t p = p0 + threadIdx.x;
    const float scale = (float)(1. / (double)(float)F);
    float m = 0.0f;
    if (GIVEN) {
        m = mean_in[p];
    } else if (zero_mean) {
        for (int c = 0; c < F; c++) m += v[c];
        m *= scale;
        mean[p] = m;
    }
    if (normalized) {
        float acc = 0.0f;
        for (int c = 0; c < F; c++) {
            const float tmp = v[c] - m; // m == 0 without zero-mean: x - 0 is exact
            acc += tmp * tmp;
        }
        norm[p] = sqrtf(acc);
    }
}

// (non-template overload: dense feature volumes take the staged kernel when `ppb` >= 32 vectors fit 60 KB of LDS)
inline int launch_stats(svh_context *ctx, FeatVolume acc, int H, int W, bool zm, bool nrm, float *mean, float *norm, const float *mean_in = nullptr) {
    const int64_t npx = (int64_t)H * W;
    if (npx == 0 || (!zm && !nrm && !mean_in)) return SVH_OK;
    const size_t rec = (size_t)(acc.F | 1) * sizeof(float);
    const int ppb = (int)(60 * 1024 / rec < 256 ? 60 * 1024 / rec : 256);
    if (ppb < 32) {
        if (mean_in) return SVH_ERR_UNSUPPORTED; // (the caller keeps its own kernel for vectors this long)
        return launch_stats<FeatVolume>(ctx, acc, H, W, zm, nrm, mean, norm);
    }
    if (mean_in) SVH_LAUNCH(ctx, "window_stats", stats_staged_kernel<true>, (int)((npx + ppb - 1) / ppb), 256, (size_t)ppb * rec, acc.f, npx, acc.F, zm, nrm, mean_in, mean, norm, ppb);
    else SVH_LAUNCH(ctx, "window_stats", stats_staged_kernel<false>, (int)((npx + ppb - 1) / ppb), 256, (size_t)ppb * rec, acc.f, npx, acc.F, zm, nrm, mean_in, mean, norm, ppb);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

enum { CMP_DOT = 0, CMP_SSD = 1, CMP_SAD = 2 };

// Window statistics of an image (unfold on the fly, auto padding) from an LDS tile: a block owns WS_TP consecutive pixels of a
// row and stages the (2 v_r + 1) x (WS_TP + 2 h_r) x C samples its windows cover once; every lane then walks its own window two
// or three times exactly as stats_kernel does (same values, same order, same operations), reading LDS instead of issuing
// (2 h_r + 1)(2 v_r + 1) C bounds-checked global loads per walk.  zcost (optional): the cost of the pixel against the all-zero
// target vector, which aggregateCost uses where the target column leaves the image (cross_correlations.h:235).
constexpr int WS_TP = 256;

// WIN > 0: a square window of that radius on an image of WC channels (1 or 3), known at compile time (the window loops unrolled: 5x5 and 7x7, the
// windows of the reference's benchmark rows); 0: whatever the FeatImage says
// grid slice z = 1: the second image of a pair (same window; its own size and outputs, no zcost) -- the two images of a matching call in one launch
template <int CMP, int WIN, int WC>
__global__ void __launch_bounds__(WS_TP) window_stats_tiled_kernel(FeatImage a, bool zero_mean, bool normalized, float *__restrict__ mean,
                                                                   float *__restrict__ norm, float *__restrict__ zcost, int *__restrict__ odd_norm_flag,
                                                                   FeatImage b, float *__restrict__ mean_b, float *__restrict__ norm_b) {
    extern __shared__ float ws_tile[];
    if (blockIdx.z == 1) {
        a = b;
        mean = mean_b;
        norm = norm_b;
        zcost = nullptr;
    }
    if ((int)blockIdx.y >= a.H || (int)blockIdx.x * WS_TP >= a.W) return; // (the grid covers the larger image of a pair)
    if (WIN > 0) {
        a.h_r = WIN;
        a.v_r = WIN;
        a.C = WC;
    }
    constexpr int UNROLL_ROWS = WIN > 0 ? 2 * WIN + 1 : 1, UNROLL_COLS = WIN > 0 ? (2 * WIN + 1) * WC : 4;
    const int h = 2 * a.h_r + 1, v = 2 * a.v_r + 1, C = a.C, F = h * v * C;
    const int tw = (WS_TP + h - 1) * C; // floats per tile row
    const int i = blockIdx.y, j0 = blockIdx.x * WS_TP;
    // the tile: eight rows' loads are issued before any of them is stored (a row-by-row loop waits for memory once per window row: seven
    // waits in a row made this kernel 58 us for a 1080p image where its arithmetic is 15); a load outside the image reads a clamped
    // address and is replaced by 0
    for (int e0 = 0; e0 < tw; e0 += 2 * WS_TP) {
        for (int k0 = 0; k0 < v; k0 += 8) {
            float got[8][2];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int ii = i - a.v_r + k0 + r;
                const bool row_in = k0 + r < v && ii >= 0 && ii < a.H; // (block uniform)
                const float *row = a.img + (int64_t)min(max(ii, 0), a.H - 1) * a.W * C;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int e = e0 + threadIdx.x + WS_TP * u;
                    const int jj = j0 - a.h_r + e / C;
                    const int64_t at = (int64_t)(j0 - a.h_r) * C + e;
                    const float x = row[min(max(at, (int64_t)0), (int64_t)a.W * C - 1)];
                    got[r][u] = (row_in && e < tw && jj >= 0 && jj < a.W) ? x : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int e = e0 + threadIdx.x + WS_TP * u;
                if (e < tw) {
#pragma unroll
                    for (int r = 0; r < 8; r++)
                        if (k0 + r < v) ws_tile[(k0 + r) * tw + e] = got[r][u];
                }
            }
        }
    }
    __syncthreads();
    const int j = j0 + threadIdx.x;
    if (j >= a.W) return;
    const int64_t p = (int64_t)i * a.W + j;
    const float *win = ws_tile + threadIdx.x * C; // window sample (k, l, ch) at win[k * tw + l * C + ch]
    const int hc = h * C;
    const float scale = (float)(1. / (double)(float)F);
    float m = 0.0f, n = 1.0f;
    if (zero_mean) {
#pragma unroll UNROLL_ROWS
        for (int k = 0; k < v; k++) {
#pragma unroll UNROLL_COLS
            for (int q = 0; q < hc; q++) m += win[k * tw + q];
        }
        m *= scale;
        mean[p] = m;
    }
    if (normalized) {
        float acc = 0.0f;
#pragma unroll UNROLL_ROWS
        for (int k = 0; k < v; k++)
#pragma unroll UNROLL_COLS
            for (int q = 0; q < hc; q++) {
                const float tmp = win[k * tw + q] - m; // m == 0 without zero-mean: x - 0 is exact
                acc += tmp * tmp;
            }
        n = sqrtf(acc);
        norm[p] = n;
        if (odd_norm_flag && !(n > 0.0f && n < INFINITY)) atomicOr(odd_norm_flag, 1); // (FiniteCostsQuery: a zero, infinite or NaN norm)
    }
    if (zcost) {
        float acc = 0.0f;
        for (int k = 0; k < v; k++)
            for (int q = 0; q < hc; q++) {
                float s = win[k * tw + q];
                if (zero_mean) s = s - m;
                if (normalized) s = s / n;
                if (CMP == CMP_DOT) acc += s * 0.0f;
                else if (CMP == CMP_SSD) acc += s * s;
                else acc += fabsf(s);
            }
        zcost[p] = acc;
    }
}

// The same statistics of a grey image without zero-mean, from COLUMN SUMS shared by the windows that contain the column (the form
// cost_volume_colsum_kernel uses for the costs themselves): the sum of squares (norm, SSD's zero-target cost) and the zero-target term
// (DOT: sum of x * 0, which is 0 or NaN; SAD: sum of |x|) of the 2 v_r + 1 samples of a column are computed once per row and column,
// and a window adds the 2 h_r + 1 column sums it covers: 2 (v + h) additions per pixel instead of 2 v h.  The terms are the reference's;
// their ORDER is by column, not by row, so a norm may differ from the literal walk in its last bits -- as the costs of the column-sum
// kernel that consume it already do (both stay inside the 1e-4 the float cost volumes are held to; tests/test_gpu_parity.py).  The
// literal kernel above stays for the zero-mean functions (their terms are not separable), multi-channel images, and behind the
// "cost_volume_colsum" = 0 option.
template <int CMP>
__global__ void __launch_bounds__(WS_TP) window_stats_colsum_kernel(FeatImage a, bool normalized, float *__restrict__ norm, float *__restrict__ zcost,
                                                                    int *__restrict__ odd_norm_flag) {
    __shared__ float colA[WS_TP + 2 * 255], colZ[WS_TP + 2 * 255];
    const int h = 2 * a.h_r + 1, v = 2 * a.v_r + 1;
    const int i = blockIdx.y, j0 = blockIdx.x * WS_TP;
    for (int e = threadIdx.x; e < WS_TP + h - 1; e += WS_TP) {
        const int jj = j0 - a.h_r + e;
        float A = 0.0f, Z = 0.0f;
        if (jj >= 0 && jj < a.W) {
#pragma unroll 4
            for (int k = 0; k < v; k++) { // (clamped address, then 0 outside: four rows' loads in flight instead of a wait per row)
                const int ii = i - a.v_r + k;
                const float got = a.img[(int64_t)min(max(ii, 0), a.H - 1) * a.W + jj];
                const float x = (ii >= 0 && ii < a.H) ? got : 0.0f;
                A += x * x;
                if (CMP == CMP_DOT) Z += x * 0.0f;
                else if (CMP == CMP_SAD) Z += fabsf(x);
            }
        }
        colA[e] = A;
        colZ[e] = Z;
    }
    __syncthreads();
    const int j = j0 + threadIdx.x;
    if (j >= a.W) return;
    const int64_t p = (int64_t)i * a.W + j;
    float A = 0.0f, Z = 0.0f;
    for (int l = 0; l < h; l++) {
        A += colA[threadIdx.x + l];
        Z += colZ[threadIdx.x + l];
    }
    float n = 1.0f;
    if (normalized) {
        n = sqrtf(A);
        norm[p] = n;
        if (odd_norm_flag && !(n > 0.0f && n < INFINITY)) atomicOr(odd_norm_flag, 1); // (FiniteCostsQuery: a zero, infinite or NaN norm)
    }
    if (zcost) {
        // the literal loop's sum of ((x / n) * 0) [DOT], x * x [SSD], |x| [SAD] over the window (zero-mean off; only DOT is ever normalised)
        float z = CMP == CMP_SSD ? A : Z;
        if (CMP == CMP_DOT && normalized && n == 0.0f) z = __uint_as_float(0x7FC00000u); // 0 / 0 (or underflowed squares: x / 0 = inf, inf * 0)
        zcost[p] = z;
    }
}

// image statistics: the LDS-tiled kernel when the tile fits, the per-lane global walk otherwise; cmp only matters for zcost
inline int launch_window_stats(svh_context *ctx, FeatImage a, bool zm, bool nrm, float *mean, float *norm, float *zcost, int cmp, bool colsum = false,
                               int *odd_norm_flag = nullptr, const FeatImage *second = nullptr, float *mean_b = nullptr, float *norm_b = nullptr) {
    if ((int64_t)a.H * a.W == 0 || (!zm && !nrm && !zcost)) return SVH_OK;
    const FeatImage b = second ? *second : a;
    if (colsum && !zm && a.C == 1 && a.h_r <= 255 && !(nrm && cmp != CMP_DOT)) {
        dim3 grid(ceil_div(a.W, WS_TP), a.H);
        if (cmp == CMP_SSD) SVH_LAUNCH(ctx, "window_stats", window_stats_colsum_kernel<CMP_SSD>, grid, WS_TP, 0, a, nrm, norm, zcost, odd_norm_flag);
        else if (cmp == CMP_SAD) SVH_LAUNCH(ctx, "window_stats", window_stats_colsum_kernel<CMP_SAD>, grid, WS_TP, 0, a, nrm, norm, zcost, odd_norm_flag);
        else SVH_LAUNCH(ctx, "window_stats", window_stats_colsum_kernel<CMP_DOT>, grid, WS_TP, 0, a, nrm, norm, zcost, odd_norm_flag);
        SVH_CHECK_LAUNCH(ctx);
        return SVH_OK;
    }
    const size_t shmem = (size_t)(2 * a.v_r + 1) * (WS_TP + 2 * a.h_r) * a.C * sizeof(float);
    if (shmem <= 60 * 1024) {
        dim3 grid(ceil_div(second ? (a.W > b.W ? a.W : b.W) : a.W, WS_TP), second ? (a.H > b.H ? a.H : b.H) : a.H, second ? 2 : 1);
        const int win = ((a.C == 1 || a.C == 3) && a.h_r == a.v_r && (a.h_r == 2 || a.h_r == 3)) ? a.h_r : 0; // compile-time windows: 5x5, 7x7, grey and RGB
#define SVH_WS_AS(CMPV, WINV, WCV) SVH_LAUNCH(ctx, "window_stats", (window_stats_tiled_kernel<CMPV, WINV, WCV>), grid, WS_TP, shmem, a, zm, nrm, mean, norm, zcost, odd_norm_flag, b, mean_b, norm_b)
#define SVH_WS_TILED(CMPV)                                   \
    do {                                                     \
        if (win == 2 && a.C == 1) SVH_WS_AS(CMPV, 2, 1);     \
        else if (win == 3 && a.C == 1) SVH_WS_AS(CMPV, 3, 1); \
        else if (win == 2) SVH_WS_AS(CMPV, 2, 3);            \
        else if (win == 3) SVH_WS_AS(CMPV, 3, 3);            \
        else SVH_WS_AS(CMPV, 0, 1);                          \
    } while (0)
        if (cmp == CMP_SSD) SVH_WS_TILED(CMP_SSD);
        else if (cmp == CMP_SAD) SVH_WS_TILED(CMP_SAD);
        else SVH_WS_TILED(CMP_DOT);
#undef SVH_WS_TILED
#undef SVH_WS_AS
        SVH_CHECK_LAUNCH(ctx);
        return SVH_OK;
    }
    if (zcost) return SVH_ERR_UNSUPPORTED; // callers that need zcost only take windows that fit (the register-blocked cost volumes)
    return launch_stats<FeatImage>(ctx, a, a.H, a.W, zm, nrm, mean, norm);
}
inline int launch_stats(svh_context *ctx, FeatImage acc, int H, int W, bool zm, bool nrm, float *mean, float *norm) {
    (void)H;
    (void)W;
    if (!zm && !nrm) return SVH_OK;
    return launch_window_stats(ctx, acc, zm, nrm, mean, norm, nullptr, CMP_DOT);
}
// means and norms of the two images of a matching call in ONE launch where both take the LDS-tiled kernel (same window, same channels:
// the grid's third dimension is the image); two launches otherwise
inline int launch_stats_pair(svh_context *ctx, FeatImage a, FeatImage b, bool zm, bool nrm, float *mean_a, float *norm_a, float *mean_b, float *norm_b) {
    if (!zm && !nrm) return SVH_OK;
    const size_t shmem = (size_t)(2 * a.v_r + 1) * (WS_TP + 2 * a.h_r) * a.C * sizeof(float);
    const bool same = a.h_r == b.h_r && a.v_r == b.v_r && a.C == b.C && (int64_t)a.H * a.W > 0 && (int64_t)b.H * b.W > 0;
    if (same && shmem <= 60 * 1024) return launch_window_stats(ctx, a, zm, nrm, mean_a, norm_a, nullptr, CMP_DOT, false, nullptr, &b, mean_b, norm_b);
    const int st = launch_window_stats(ctx, a, zm, nrm, mean_a, norm_a, nullptr, CMP_DOT);
    if (st != SVH_OK) return st;
    return launch_window_stats(ctx, b, zm, nrm, mean_b, norm_b, nullptr, CMP_DOT);
}

// cmp(src(i,j,:), tgt(it,jt,:)) on the processed features (zero-mean / normalised per element, cross_correlations.h:416-594), with
// the reference's float operations in the reference's channel order; a target pixel outside the image is the zero vector
// (cross_correlations.h:235, :294, :359; hierarchical.h:177).
template <class A, int CMP, bool ZM, bool NORM>
__device__ __forceinline__ float compare_features(const A &src, const A &tgt, const float *__restrict__ mean_s, const float *__restrict__ norm_s,
                                                  const float *__restrict__ mean_t, const float *__restrict__ norm_t, int H, int Ws, int Wt, int i,
                                                  int j, int it, int jt) {
    const int F = src.count();
    const int64_t p = (int64_t)i * Ws + j;
    bool t_in = jt >= 0 && jt < Wt && it >= 0 && it < H;
    float ms = ZM ? mean_s[p] : 0.0f;
    float ns = NORM ? norm_s[p] : 1.0f;
    float mt = 0.0f, nt = 1.0f;
    if (t_in) {
        int64_t pt = (int64_t)it * Wt + jt;
        if (ZM) mt = mean_t[pt];
        if (NORM) nt = norm_t[pt];
    }
    typename CursorOf<A>::type cs(src, i, j);
    typename CursorOf<A>::type ct(tgt, t_in ? it : 0, t_in ? jt : 0);
    float score = 0.0f;
    for (int c = 0; c < F; c++) {
        float s = cs.next();
        float t = ct.next();
        if (ZM) s = s - ms;
        if (NORM) s = s / ns;
        if (t_in) {
            if (ZM) t = t - mt;
            if (NORM) t = t / nt;
        } else {
            t = 0.0f;
        }
        if (CMP == CMP_DOT) {
            score += s * t; // dotProduct, matching_costs.h:59-78
        } else if (CMP == CMP_SSD) {
            float tmp = s - t; // SumSquareDiff, :100-116
            score += tmp * tmp;
        } else {
            float tmp = s - t; // SumAbsDiff, :136-156
            score += fabsf(tmp);
        }
    }
    return score;
}

} // namespace svh

#endif // SVH_COMPARE_H

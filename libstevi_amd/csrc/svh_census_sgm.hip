// Census / Hamming specialisation of the SGM Cost branch ("pixel per lane" kernels).
//
// For census costs the volume is a pure function of two small word maps, so nothing voxel-sized has to live in
// HBM.  These kernels map one image pixel to one lane and walk the disparity axis sequentially inside the lane:
//   * a block handles 256 consecutive pixels of one row and stages the (256 + D - 1)-pixel window of target
//     census words in LDS (one plane per word: lanes read consecutive dwords, conflict free);
//   * the per-pixel reductions over d (minimum, winner with the reference's "<=" scan order) need no cross-lane
//     traffic at all, and the d loop is literally the reference's loop order, ties included.
//
// Exactness argument used by the fast path (checked on the host before it is taken, see census_exact_regime):
// Hamming costs are integers in [0, 32 nWw]; when Pout is an integer and 8 (2 cmax + |Pout|) (L + 2) < 2^24
// (L = longest line), every intermediate of sgm.h:257-300 is an integer below 2^24 in magnitude, so each float
// operation of the reference is exact.  Then the next pixel's min_p is
//     min_d [ c + ((c [+Pout]) - mp) ]  =  g(p) - mp,      g(p) = min_d [ c + (c [+Pout]) ]
// bit for bit, i.e. the sequential part of a pass collapses to the scalar recurrence mp' = g - mp along the line
// (line_scan_kernel), fed by one fully parallel sweep (census_gmin_kernel).  The apply step still evaluates the
// reference's expression per voxel.  Outside that regime the general wave-per-line kernels of svh_sgm.hip run.
#include "svh_internal.h"

namespace svh {

namespace {

constexpr int TJ = 256; // pixels (= threads) per block

struct CensusGeom {
    const uint32_t *sw, *tw; // compact words (H, Ws, nWw), (H, Wt, nWw); target pre-rounded through float (E2)
    int nWw, H, Ws, Wt, D, sign, disp_lower;
};

// Stage the target-word window of this block into LDS, planar: lds[w * win + x], x = column - x_base.
// sign > 0: x = tj + d;  sign < 0: x = tj + (D - 1 - d).
__device__ __forceinline__ void stage_target_window(const CensusGeom &g, int i, int j0, uint32_t *lds, int win) {
    const int x_base = g.sign > 0 ? j0 + g.disp_lower : j0 - g.disp_lower - (g.D - 1);
    const uint32_t *trow = g.tw + (int64_t)i * g.Wt * g.nWw;
    for (int e = threadIdx.x; e < win * g.nWw; e += blockDim.x) {
        int x = e / g.nWw, w = e - x * g.nWw; // consecutive threads read consecutive dwords of the row
        int jt = x_base + x;
        lds[w * win + x] = (jt >= 0 && jt < g.Wt) ? trow[(int64_t)jt * g.nWw + w] : 0u; // zero vector outside the image
    }
}

template <int NW> struct Words {
    uint32_t v[NW > 0 ? NW : 1];
};

template <int NW> __device__ __forceinline__ float hamming_at(const Words<NW> &s, const uint32_t *lds, int win, int x, int nWw) {
    uint32_t acc = 0;
    if constexpr (NW > 0) {
#pragma unroll
        for (int w = 0; w < NW; w++) acc += __popc(s.v[w] ^ lds[w * win + x]);
    } else {
        (void)s;
        (void)lds;
        (void)win;
        (void)x;
        (void)nWw;
    }
    return (float)acc;
}

template <int NW> __device__ __forceinline__ int hamming_int(const Words<NW> &s, const uint32_t *lds, int win, int x) {
    int acc = 0;
    if constexpr (NW > 0) {
#pragma unroll
        for (int w = 0; w < NW; w++) acc += __popc(s.v[w] ^ lds[w * win + x]);
    }
    return acc;
}

// g(p) = min over d of the first-pixel actual cost c + (c [+ Pout]) (sgm.h:287-294 with min_p = 0)
template <int NW, bool EXACT>
__global__ void __launch_bounds__(TJ) census_gmin_kernel(CensusGeom g, float Pout, float *__restrict__ gmap) {
    extern __shared__ uint32_t lds[];
    const int i = blockIdx.y, j0 = blockIdx.x * TJ, tj = threadIdx.x, j = j0 + tj;
    const int win = TJ + g.D - 1;
    stage_target_window(g, i, j0, lds, win);
    __syncthreads();
    if (j >= g.Ws) return;
    Words<NW> s;
    const uint32_t *sp = g.sw + ((int64_t)i * g.Ws + j) * g.nWw;
#pragma unroll
    for (int w = 0; w < NW; w++) s.v[w] = sp[w];
    float m = INFINITY;
    if (EXACT) {
        // integer-exact regime: c + (c + Pout) = 2c + Pout with integer Pout, evaluated in int32
        const int pout = (int)Pout;
        int mi = 0x7FFFFFFF;
        for (int d = 0; d < g.D; d++) {
            const int x = tj + (g.sign > 0 ? d : g.D - 1 - d);
            const int c2 = 2 * hamming_int<NW>(s, lds, win, x);
            mi = min(mi, (j + d >= g.Ws) ? c2 + pout : c2);
        }
        m = (float)mi;
    } else {
        for (int d = 0; d < g.D; d++) {
            const int x = tj + (g.sign > 0 ? d : g.D - 1 - d);
            const float c = hamming_at<NW>(s, lds, win, x, g.nWw);
            const float t = (j + d >= g.Ws) ? c + Pout : c;
            m = fminf(m, c + t);
        }
    }
    gmap[(int64_t)i * g.Ws + j] = m;
}

struct ScanGeom {
    int top, left, Hp, Wp, W;
};

// mp' = g - mp along every line of every pass (blockIdx.y = pass); one thread per line, batched loads
__global__ void __launch_bounds__(64) line_scan_kernel(const float *__restrict__ gmap, ScanGeom sg, int64_t npx, float *__restrict__ mmap) {
    const int q = blockIdx.y;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_lines = (q == 0 || q == 3 || q == 4) ? sg.Wp : sg.Hp;
    if (l >= n_lines) return;
    int i0, j0, di, dj, len;
    switch (q) {
    case 0: i0 = sg.top; j0 = sg.left + l; di = 1; dj = 0; len = sg.Hp; break;
    case 1: i0 = sg.top + l; j0 = sg.left; di = 0; dj = 1; len = sg.Wp; break;
    case 2: i0 = sg.top + l; j0 = sg.left; di = 1; dj = 1; len = min(sg.Hp - l, sg.Wp); break;
    case 3: i0 = sg.top; j0 = sg.left + l; di = 1; dj = 1; len = min(sg.Hp, sg.Wp - l); break;
    case 4: i0 = sg.top; j0 = sg.left + l; di = 1; dj = -1; len = min(sg.Hp, l + 1); break;
    default: i0 = sg.top + l; j0 = sg.left; di = -1; dj = 1; len = min(l + 1, sg.Wp); break;
    }
    float *out = mmap + (int64_t)q * npx;
    const int64_t step = (int64_t)di * sg.W + dj;
    int64_t p = (int64_t)i0 * sg.W + j0;
    float mp = 0.0f;
    constexpr int U = 8;
    int k = 0;
    for (; k + U <= len; k += U) {
        float gv[U];
#pragma unroll
        for (int u = 0; u < U; u++) gv[u] = gmap[p + u * step];
#pragma unroll
        for (int u = 0; u < U; u++) {
            out[p + u * step] = mp;
            mp = gv[u] - mp;
        }
        p += U * step;
    }
    for (; k < len; k++, p += step) {
        out[p] = mp;
        mp = gmap[p] - mp;
    }
}

struct SelectOut {
    int32_t *idx;
    float *taps;
    int taps_h_r, taps_v_r;
    unsigned long long *keys;
    int key_offset, key_total;
};

__device__ __forceinline__ uint32_t order_key_f(float v) {
    if (v == 0.0f) v = 0.0f;
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// S(p, d) per sgm.h:298-300 from the per-pass min_p values of this pixel
struct PixelPasses {
    float mp[6];
    unsigned vis; // bit q set: pass q visits the pixel and its min_p is finite
};

__device__ __forceinline__ float sgm_value(float c, float t, const PixelPasses &pp, int n_pass) {
    float s = c;
    const bool t_fin = fabsf(t) < INFINITY;
#pragma unroll
    for (int q = 0; q < 6; q++) {
        if (q < n_pass) {
            float act = c + (t - pp.mp[q]);
            float ns = s + (act - c);
            s = ((pp.vis >> q) & 1u) && t_fin ? ns : s;
        }
    }
    return s;
}

// apply + extractSelectedIndex (+ truncated taps, + reduction keys) without writing any volume
template <int NW, bool EXACT>
__global__ void __launch_bounds__(TJ) census_apply_select_kernel(CensusGeom g, ScanGeom sg, int n_pass, float Pout,
                                                                 const float *__restrict__ mmap, SelectOut out) {
    extern __shared__ uint32_t lds[];
    const int i = blockIdx.y, j0 = blockIdx.x * TJ, tj = threadIdx.x, j = j0 + tj;
    const int win = TJ + g.D - 1;
    stage_target_window(g, i, j0, lds, win);
    __syncthreads();
    if (j >= g.Ws) return;
    const int64_t npx = (int64_t)g.H * g.Ws, p = (int64_t)i * g.Ws + j;
    Words<NW> s;
    const uint32_t *sp = g.sw + p * g.nWw;
#pragma unroll
    for (int w = 0; w < NW; w++) s.v[w] = sp[w];
    PixelPasses pp;
    pp.vis = 0;
    const int ip = i - sg.top, jp = j - sg.left;
    const bool inside = ip >= 0 && ip < sg.Hp && jp >= 0 && jp < sg.Wp;
#pragma unroll
    for (int q = 0; q < 6; q++) {
        pp.mp[q] = 0.0f;
        bool v = inside && q < n_pass;
        if (q == 2) v = v && ip >= jp;
        if (q == 3) v = v && jp >= ip;
        if (q == 4) v = v && ip + jp < sg.Wp;
        if (q == 5) v = v && ip + jp < sg.Hp;
        if (v) {
            pp.mp[q] = mmap[(int64_t)q * npx + p];
            if (fabsf(pp.mp[q]) < INFINITY) pp.vis |= 1u << q;
        }
    }
    // sequential scan of extractSelectedIndex (correlation_base.h:441-455): '<=' keeps the last minimum
    float best = 0.0f;
    int bd = 0;
    // integer-exact regime: every term (c + (t - mp)) - c equals t - mp exactly, so
    // S = (1 + n) c + n Pout [oob] - sum(mp) in int32 carries the very same values as the float expression
    int n_vis = 0, k0 = 0, k1 = 0;
    if (EXACT) {
        int msum = 0;
#pragma unroll
        for (int q = 0; q < 6; q++)
            if ((pp.vis >> q) & 1u) {
                n_vis++;
                msum += (int)pp.mp[q];
            }
        k0 = -msum;
        k1 = n_vis * (int)Pout - msum;
        int besti = 0;
        const int mul = 1 + n_vis;
        for (int d = 0; d < g.D; d++) {
            const int x = tj + (g.sign > 0 ? d : g.D - 1 - d);
            const int v = mul * hamming_int<NW>(s, lds, win, x) + ((j + d >= g.Ws) ? k1 : k0);
            if (d == 0 || v <= besti) {
                besti = v;
                bd = d;
            }
        }
        best = (float)besti;
    } else {
        for (int d = 0; d < g.D; d++) {
            const int x = tj + (g.sign > 0 ? d : g.D - 1 - d);
            const float c = hamming_at<NW>(s, lds, win, x, g.nWw);
            const float t = (j + d >= g.Ws) ? c + Pout : c;
            const float v = sgm_value(c, t, pp, n_pass);
            if (d == 0 || v <= best) {
                best = v;
                bd = d;
            }
        }
    }
    if (out.idx) out.idx[p] = bd;
    if (out.keys) // census values are never NaN
        out.keys[p] = ((unsigned long long)order_key_f(best) << 32) | (uint32_t)(out.key_total - 1 - (out.key_offset + bd));
    if (out.taps) {
        const bool px_bad = j < out.taps_h_r || i < out.taps_v_r || i + out.taps_v_r >= g.H;
#pragma unroll
        for (int tap = 0; tap < 3; tap++) {
            const int pd = bd + tap - 1;
            float v = __uint_as_float(0x7FC00000u);
            if (!(px_bad || pd < 0 || pd >= g.D || j + pd + out.taps_h_r >= g.Ws)) {
                const int x = tj + (g.sign > 0 ? pd : g.D - 1 - pd);
                if (EXACT) {
                    v = (float)((1 + n_vis) * hamming_int<NW>(s, lds, win, x) + ((j + pd >= g.Ws) ? k1 : k0));
                } else {
                    const float c = hamming_at<NW>(s, lds, win, x, g.nWw);
                    const float t = (j + pd >= g.Ws) ? c + Pout : c;
                    v = sgm_value(c, t, pp, n_pass);
                }
            }
            out.taps[p * 3 + tap] = v;
        }
    }
}

template <int NW>
int launch_gmin(svh_context *ctx, const CensusGeom &g, float Pout, float *gmap) {
    dim3 grid(ceil_div(g.Ws, TJ), g.H);
    size_t shmem = (size_t)(NW > 0 ? NW : 1) * (TJ + g.D - 1) * sizeof(uint32_t);
    SVH_LAUNCH(ctx, "census_gmin", (census_gmin_kernel<NW, true>), grid, TJ, shmem, g, Pout, gmap);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

template <int NW>
int launch_apply_select(svh_context *ctx, const CensusGeom &g, const ScanGeom &sg, int n_pass, float Pout, const float *mmap,
                        const SelectOut &out, bool exact) {
    dim3 grid(ceil_div(g.Ws, TJ), g.H);
    size_t shmem = (size_t)(NW > 0 ? NW : 1) * (TJ + g.D - 1) * sizeof(uint32_t);
    if (exact)
        SVH_LAUNCH(ctx, "census_apply_select", (census_apply_select_kernel<NW, true>), grid, TJ, shmem, g, sg, n_pass, Pout, mmap, out);
    else
        SVH_LAUNCH(ctx, "census_apply_select", (census_apply_select_kernel<NW, false>), grid, TJ, shmem, g, sg, n_pass, Pout, mmap, out);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

} // namespace

// largest nWw the pixel-per-lane kernels are instantiated for (25x25 windows and smaller)
static constexpr int kMaxWords = 4;
// LDS budget: nWw planes of (256 + D - 1) dwords must fit the 64 KiB a block may take by default
static bool census_lane_kernels_fit(int nWw, int D) { return nWw <= kMaxWords && (size_t)(nWw ? nWw : 1) * (TJ + D - 1) * 4 <= 60 * 1024; }

bool census_exact_regime(const SgmArgs &a, int nWw) {
    if (!std::isfinite(a.Pout) || a.Pout != std::nearbyint(a.Pout)) return false;
    const double cmax = 32.0 * nWw, gmax = 2.0 * cmax + std::fabs((double)a.Pout);
    const double L = (double)std::max(a.H, a.W);
    return 8.0 * gmax * (L + 2.0) < 16777216.0;
}

int dev_census_minmaps_exact(svh_context *ctx, Scratch &scr, const SgmArgs &a, const CostSource &cs, float *mmap) {
    const int Hp = a.H - a.top - a.bottom, Wp = a.W - a.left - a.right;
    const int n_pass = a.n_dir >= 8 ? 6 : (a.n_dir >= 4 ? 2 : 0);
    if (n_pass == 0 || Hp <= 0 || Wp <= 0) return SVH_OK;
    float *gmap = scr.get_n<float>((size_t)a.H * a.W);
    if (!gmap) return SVH_ERR_OUT_OF_MEMORY;
    CensusGeom g{cs.src_words, cs.tgt_words, cs.nWw, a.H, a.W, cs.Wt, a.D, cs.sign, cs.disp_lower};
    switch (cs.nWw) {
    case 0: SVH_TRY(launch_gmin<0>(ctx, g, a.Pout, gmap)); break;
    case 1: SVH_TRY(launch_gmin<1>(ctx, g, a.Pout, gmap)); break;
    case 2: SVH_TRY(launch_gmin<2>(ctx, g, a.Pout, gmap)); break;
    case 3: SVH_TRY(launch_gmin<3>(ctx, g, a.Pout, gmap)); break;
    default: SVH_TRY(launch_gmin<4>(ctx, g, a.Pout, gmap)); break;
    }
    ScanGeom sg{a.top, a.left, Hp, Wp, a.W};
    dim3 grid(ceil_div(std::max(Hp, Wp), 64), n_pass);
    SVH_LAUNCH(ctx, "sgm_line_scan", line_scan_kernel, grid, 64, 0, gmap, sg, (int64_t)a.H * a.W, mmap);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

int dev_census_apply_select(svh_context *ctx, const SgmArgs &a, const CostSource &cs, const float *mmap, int32_t *out_idx,
                            float *out_taps, int taps_h_r, int taps_v_r, unsigned long long *out_keys, int key_index_offset,
                            int key_total_D) {
    const int Hp = a.H - a.top - a.bottom, Wp = a.W - a.left - a.right;
    const int n_pass = (Hp > 0 && Wp > 0) ? (a.n_dir >= 8 ? 6 : (a.n_dir >= 4 ? 2 : 0)) : 0;
    CensusGeom g{cs.src_words, cs.tgt_words, cs.nWw, a.H, a.W, cs.Wt, a.D, cs.sign, cs.disp_lower};
    ScanGeom sg{a.top, a.left, Hp > 0 ? Hp : 0, Wp > 0 ? Wp : 0, a.W};
    SelectOut out{out_idx, out_taps, taps_h_r, taps_v_r, out_keys, key_index_offset, key_total_D};
    const bool exact = census_exact_regime(a, cs.nWw); // then the min_p maps hold integers too, whichever kernel made them
    switch (cs.nWw) {
    case 0: return launch_apply_select<0>(ctx, g, sg, n_pass, a.Pout, mmap, out, exact);
    case 1: return launch_apply_select<1>(ctx, g, sg, n_pass, a.Pout, mmap, out, exact);
    case 2: return launch_apply_select<2>(ctx, g, sg, n_pass, a.Pout, mmap, out, exact);
    case 3: return launch_apply_select<3>(ctx, g, sg, n_pass, a.Pout, mmap, out, exact);
    default: return launch_apply_select<4>(ctx, g, sg, n_pass, a.Pout, mmap, out, exact);
    }
}

bool census_lane_kernels_available(int nWw, int D) { return census_lane_kernels_fit(nWw, D); }

} // namespace svh

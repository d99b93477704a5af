// computeGuidedCV (correlation/hierarchical.h:74-229) on grey images with windows up to 7 wide and search radii up to 3, a WAVE per 64
// pixels of a row.  Included by svh_guided_wave_h{1,2,3}.hip (one per window half-width).
//
// A pixel's result is a function of the 4R + 1 offsets around its guide d0: the winner among the middle 2R + 1 and the re-centred window
// around it (see svh_hierarchical.hip).  Every offset's cost walks the window rows, the samples of a row and compares the processed source
// sample -- (s - mean_s) / norm_s -- with the processed target sample -- (t - mean_t[centre]) / norm_t[centre] -- in the reference's
// order.  The processed target samples of a CENTRE are the same for every pixel that looks at it: the wave stages the raw rows of both
// images in LDS, then, window row by window row, processes each (centre, sample) of the span its pixels look at ONCE into a strip and
// every lane reads its 4R + 1 centres from there.
//  * One wave per block: the two meeting points per window row cost a wave nothing to wait at (the block form of round 4 -- four waves,
//    256 pixels -- had its first wave carry the second round of centres alone, and one patch of noise in the guide sent 256 pixels
//    to the per-lane walk).
//  * A lane processes the samples of the centres lane and 64 + lane of the span (means and reciprocal norms in registers); the strip
//    is sample-major so that a lane's 4R + 1 centres are consecutive words.
//  * Every division by a norm goes through div_by_shared: the norm's double reciprocal is formed once per centre / per source pixel.
//  * A wave whose guides point further apart than the staged span serves its pixels in PASSES -- a pass takes the lowest unserved centre
//    and every pixel whose offsets fit in the span behind it (the two sides of a parallax edge: two passes) -- and what a pass would
//    serve fewer than 16 pixels of (noise in the guide) walks per lane, five offsets at a time: their windows of a row overlap in h + 4 raw
//    samples, loaded once per row into registers; the source samples come from the staged rows.
#pragma once
#include <climits>

#include "svh_guided_wave.h"

namespace svh {

namespace {

// x / y for many x over one y, through the double reciprocal rd = 1.0 / (double)y: three instructions instead of the ten of a float division
// with its quarter-rate v_rcp_f32, and the same bits.  Why: the quotient of two 24-bit floats is never closer to a rounding boundary of the float
// format (a 25-bit midpoint) than 2^-49 of itself, and (double)x * rd carries at most 2^-52 (rd: 2^-53, the product: 2^-53), so rounding
// it to float rounds the exact quotient; signs of zeros, infinities, NaN, y = 0 and y = inf follow the same rules in the product as in the
// quotient.  Below the normal range the float grid is coarser than 24 bits and a quotient CAN sit exactly on a tie (x = y m for a
// midpoint m of few bits), where the product's 2^-52 decide the direction: a nonzero denormal result raises `bad` and the caller repeats
// the pixel with the division itself (a result of 0 is always right: the nearest other quotient is 2^-24 away from the 2^-150 boundary).
__device__ __forceinline__ float div_by_shared(float x, double rd, bool &bad) {
    const float q = (float)((double)x * rd);
    bad |= __builtin_amdgcn_classf(q, 0x090); // -denormal | +denormal
    return q;
}

constexpr int GW_PASSES = 3;      // passes before the rest walks
constexpr int GW_MIN_SERVED = 16; // a pass for fewer pixels than this is not worth its rows

template <int CMP>
__device__ __forceinline__ void accumulate(float &acc, float s, float t) {
    if (CMP == CMP_DOT) {
        acc += s * t;
    } else if (CMP == CMP_SSD) {
        const float tmp = s - t;
        acc += tmp * tmp;
    } else {
        acc += fabsf(s - t);
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int CMP>
__device__ __forceinline__ void accumulate2(f32x2 &acc, f32x2 s, f32x2 t) {
    if (CMP == CMP_DOT) {
        acc += s * t;
    } else if (CMP == CMP_SSD) {
        const f32x2 tmp = s - t;
        acc += tmp * tmp;
    } else {
        const f32x2 tmp = s - t;
        acc += f32x2{fabsf(tmp.x), fabsf(tmp.y)};
    }
}

// winner among the middle offsets (strict comparison in increasing offset, hierarchical.h:157-190) and the re-centred window (:194-227)
template <int R>
__device__ __forceinline__ void emit_pixel(const float (&acc)[4 * R + 1], const GuideArgs &g, int d0, int64_t p, int32_t *__restrict__ disp,
                                           float *__restrict__ tcv) {
    constexpr int NC = 4 * R + 1, T = 2 * R + 1;
    float score = g.cost ? INFINITY : -INFINITY;
    int best = 0; // offset of the winner relative to d0
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

// Rows of an image into LDS rows of PITCH words (outside the image: 0), in two halves so that a caller can put its other loads between them:
// `issue` loads the rows k0 .. k0 + 7 of `n_rows` from image row i0 + k0 on, `row_len` columns from column c0 -- a lane owns the columns
// lane + 64 u -- and `commit` stores them.  A wave of this kernel is as long as its dependent trips to memory (the guide; then the target rows
// the guide points at), so everything a trip can carry is issued before anything waits; every load reads a clamped address and is replaced
// by 0 afterwards where it was outside (no branch per load).
template <int NUX, int C> struct StagedRows { // (row_len, PITCH and the lanes' columns count floats: C per pixel; c0: the first PIXEL column)
    float val[8][NUX];
    template <int PITCH>
    __device__ __forceinline__ void issue(const float *__restrict__ img, int H, int W, int i0, int c0, int k0, int row_len, int lane) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int ii = i0 + k0 + r;
            const bool row_ok = ii >= 0 && ii < H; // (wave uniform)
            const float *rowp = img + (int64_t)min(max(ii, 0), H - 1) * W * C;
#pragma unroll
            for (int u = 0; u < NUX; u++) {
                const int x = lane + 64 * u, px = x / C, ch = x - px * C, jj = c0 + px;
                const float got = rowp[min(max(jj, 0), W - 1) * C + ch];
                val[r][u] = (row_ok && x < row_len && jj >= 0 && jj < W) ? got : 0.0f;
            }
        }
    }
    template <int PITCH> __device__ __forceinline__ void commit(float *dst, int n_rows, int k0, int row_len, int lane) const {
#pragma unroll
        for (int u = 0; u < NUX; u++) {
            if (lane + 64 * u < row_len) {
#pragma unroll
                for (int r = 0; r < 8; r++)
                    if (k0 + r < n_rows) dst[(k0 + r) * PITCH + lane + 64 * u] = val[r][u];
            }
        }
    }
};

// wave minimum / maximum of an int, in every lane (six DPP steps and a v_readlane instead of six ds_bpermute round trips)
template <bool MIN> __device__ __forceinline__ int wave_extremum_i32(int v) {
#define SVH_EXT_DPP(CTRL)                                                              \
    if constexpr (MIN) asm("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 " CTRL : "+v"(v));     \
    else asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " CTRL : "+v"(v))
    SVH_EXT_DPP("row_shr:1 row_mask:0xf bank_mask:0xf");
    SVH_EXT_DPP("row_shr:2 row_mask:0xf bank_mask:0xf");
    SVH_EXT_DPP("row_shr:4 row_mask:0xf bank_mask:0xf");
    SVH_EXT_DPP("row_shr:8 row_mask:0xf bank_mask:0xf");
    SVH_EXT_DPP("row_bcast:15 row_mask:0xa bank_mask:0xf");
    SVH_EXT_DPP("row_bcast:31 row_mask:0xc bank_mask:0xf");
#undef SVH_EXT_DPP
    asm("s_nop 0" : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// The per-lane walk of the offsets C0 .. C0 + G - 1 (a few at a time keeps the registers of the whole kernel at the passes' level): the G
// windows of a row overlap in h + G - 1 raw pixels, loaded once per row; the source samples come from the staged rows (`srow`: this
// lane's first sample of row 0).  C: interleaved channels (the feature index runs rows, columns, channels: unfold.h:180).
template <int CMP, bool ZM, bool NORM, int HR, int C, int NC, int C0>
__device__ __forceinline__ void walk_offsets(const FeatImage &tgt, const float *__restrict__ mean_t, const float *__restrict__ norm_t, int H, int Wt,
                                             const float *srow, int SW, int v, int v_r, int i, int first, float ms, double rs, float (&acc)[NC], bool &bad) {
    constexpr int h = 2 * HR + 1, GMAX = C == 1 ? 5 : 2, G = NC - C0 < GMAX ? NC - C0 : GMAX, NRAW = (h + G - 1) * C;
    float mt[G];
    double rd[G];
    bool tin[G];
#pragma unroll
    for (int c = 0; c < G; c++) {
        const int jt = first + C0 + c;
        tin[c] = jt >= 0 && jt < Wt;
        const int64_t pt = (int64_t)i * Wt + (tin[c] ? jt : 0);
        mt[c] = (ZM && tin[c]) ? mean_t[pt] : 0.0f;
        rd[c] = (NORM && tin[c]) ? 1.0 / (double)norm_t[pt] : 1.0;
    }
    float raw[NRAW], ahead[NRAW]; // (the row after is loaded while this one is worked on)
    auto load_row = [&](int k, float (&row)[NRAW]) {
        const int ii = i - v_r + k;
        const bool row_in = k < v && ii >= 0 && ii < H;
#pragma unroll
        for (int x = 0; x < NRAW; x++) {
            const int jc = first + C0 - HR + x / C;
            row[x] = (row_in && jc >= 0 && jc < Wt) ? tgt.img[((int64_t)ii * Wt + jc) * C + x % C] : 0.0f;
        }
    };
    load_row(0, ahead);
    for (int k = 0; k < v; k++) {
#pragma unroll
        for (int x = 0; x < NRAW; x++) raw[x] = ahead[x];
        load_row(k + 1, ahead);
#pragma unroll
        for (int m = 0; m < h * C; m++) { // (sample m: column m / C, channel m % C)
            float s = srow[k * SW + m];
            if (ZM) s = s - ms;
            if (NORM) s = div_by_shared(s, rs, bad);
#pragma unroll
            for (int c = 0; c < G; c++) {
                float t = 0.0f; // a centre outside the image is the all-zero vector
                if (tin[c]) {
                    t = raw[m + c * C];
                    if (ZM) t = t - mt[c];
                    if (NORM) t = div_by_shared(t, rd[c], bad);
                }
                accumulate<CMP>(acc[C0 + c], s, t);
            }
        }
    }
    if constexpr (C0 + G < NC) walk_offsets<CMP, ZM, NORM, HR, C, NC, C0 + G>(tgt, mean_t, norm_t, H, Wt, srow, SW, v, v_r, i, first, ms, rs, acc, bad);
}

// SPAN: the target centres a pass stages -- 64 pixels + 4R + 1 offsets + room for the guide to move
template <int CMP, bool ZM, bool NORM, int R, int HR, int SPAN, int C>
__global__ void __launch_bounds__(64) guided_wave_kernel(FeatImage src, FeatImage tgt, const float *__restrict__ mean_s, const float *__restrict__ norm_s,
                                                         const float *__restrict__ mean_t, const float *__restrict__ norm_t, int H, int Ws, int Wt,
                                                         GuideArgs g, int32_t *__restrict__ disp, float *__restrict__ tcv) {
    constexpr int NC = 4 * R + 1, NP = (NC + 1) / 2, h = 2 * HR + 1, hc = h * C, SW = (64 + h - 1) * C; // (floats: C interleaved channels per pixel)
    constexpr int TW = (SPAN + 2 * h - 1) * C;      // a staged target row: span + h - 1 raw pixels and h zero pixels (what a centre outside the image reads, hierarchical.h:175-178)
    constexpr int SP = SPAN + 1;                    // strip pitch (odd: a lane's centre q and sample l -> word l * SP + q)
    extern __shared__ float lds_d[];
    const int lane = threadIdx.x, i = blockIdx.y, j0 = blockIdx.x * 64, j = j0 + lane;
    const int v = 2 * src.v_r + 1;
    float *stile = lds_d, *ttile = stile + v * SW; // raw window rows of both images
    float *strip = ttile + v * TW;                           // processed samples of one window row: [sample l][centre q]
    const bool px = j < Ws;
    const int64_t p = (int64_t)i * Ws + (px ? j : Ws - 1);
    // first trip to memory: the source rows, the guide, this pixel's mean and norm
    StagedRows<(SW + 63) / 64, C> srows;
    srows.template issue<SW>(src.img, H, Ws, i - src.v_r, j0 - HR, 0, SW, lane);
    const int d0 = px ? guided_base_disp(g.guide, g.Hg, g.Wg, H, Ws, i, j, g.dirSign) : 0;
    const float ms = ZM ? mean_s[p] : 0.0f, ns = NORM ? norm_s[p] : 1.0f;
    srows.template commit<SW>(stile, v, 0, SW, lane);
    for (int k0 = 8; k0 < v; k0 += 8) { // (windows taller than 8 rows)
        srows.template issue<SW>(src.img, H, Ws, i - src.v_r, j0 - HR, k0, SW, lane);
        srows.template commit<SW>(stile, v, k0, SW, lane);
    }
    for (int x = lane; x < v * hc; x += 64) ttile[(x / hc) * TW + TW - hc + x % hc] = 0.0f;
    const int first = j + d0 - 2 * R; // this pixel's first target centre
    const double rs = 1.0 / (double)ns;
    bool todo = px, redo = false;
    float acc[NC];
    for (int pass = 0; pass < GW_PASSES; pass++) {
        if (__ballot(todo) == 0) break; // (wave uniform)
        const int lo = wave_extremum_i32<true>(todo ? first : INT_MAX);
        const bool served = todo && (int64_t)first + NC - 1 - lo < SPAN; // (the lane that owns lo always is)
        if ((int)__popcll(__ballot(served)) < GW_MIN_SERVED) break;
        const int span = wave_extremum_i32<false>(served ? first + NC - 1 : INT_MIN) - lo + 1;
        __syncthreads(); // (one wave: the readers of the pass before are done)
        // second trip: the target rows behind lo; the means and norms of the centres this lane processes -- lane, 64 + lane, ... of the span
        // (a centre outside the image reads the row's zeros)
        constexpr int NU = (SPAN + 63) / 64;
        StagedRows<((SPAN + h - 1) * C + 63) / 64, C> trows;
        trows.template issue<TW>(tgt.img, H, Wt, i - src.v_r, lo - HR, 0, (span + h - 1) * C, lane);
        float cm[NU], cn[NU];
        int cin[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int q = lane + 64 * u, jt = lo + q;
            const bool tin = q < span && jt >= 0 && jt < Wt;
            const int64_t pt = (int64_t)i * Wt + min(max(jt, 0), Wt - 1);
            const float got_m = ZM ? mean_t[pt] : 0.0f, got_n = NORM ? norm_t[pt] : 1.0f;
            cm[u] = tin ? got_m : 0.0f;
            cn[u] = tin ? got_n : 1.0f;
            cin[u] = tin ? q * C : TW - hc;
        }
        trows.template commit<TW>(ttile, v, 0, (span + h - 1) * C, lane);
        for (int k0 = 8; k0 < v; k0 += 8) {
            trows.template issue<TW>(tgt.img, H, Wt, i - src.v_r, lo - HR, k0, (span + h - 1) * C, lane);
            trows.template commit<TW>(ttile, v, k0, (span + h - 1) * C, lane);
        }
        double cr[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) cr[u] = 1.0 / (double)cn[u];
        __syncthreads();
        f32x2 acc2[NP]; // offsets 2c, 2c + 1 side by side: packed multiplies and adds (the same roundings, two per instruction)
#pragma unroll
        for (int c = 0; c < NP; c++) acc2[c] = f32x2{0.0f, 0.0f};
        const int qb = served ? first - lo : 0;
        bool bad_centre = false, bad_px = false;
        for (int k = 0; k < v; k++) {
            const float *trow = ttile + k * TW;
            float t[NU][hc];
#pragma unroll
            for (int u = 0; u < NU; u++) {
                if (u == 0 || span > 64 * u) { // (wave uniform)
#pragma unroll
                    for (int l = 0; l < hc; l++) {
                        t[u][l] = trow[cin[u] + l];
                        if (ZM) t[u][l] = t[u][l] - cm[u];
                        if (NORM) t[u][l] = div_by_shared(t[u][l], cr[u], bad_centre);
                    }
                }
            }
            __syncthreads(); // (the strip's readers of the row before are done)
#pragma unroll
            for (int u = 0; u < NU; u++) {
                if ((u == 0 || span > 64 * u) && lane + 64 * u < SPAN) {
#pragma unroll
                    for (int l = 0; l < hc; l++) strip[l * SP + 64 * u + lane] = t[u][l];
                }
            }
            __syncthreads();
#pragma unroll
            for (int l = 0; l < hc; l++) { // (sample l: column l / C, channel l % C)
                float s = stile[k * SW + lane * C + l];
                if (ZM) s = s - ms;
                if (NORM) s = div_by_shared(s, rs, bad_px);
                int row_at = l * SP + qb; // (opaque: one address per sample row and small offsets behind it, not a constant beyond the offset field per read)
                asm volatile("" : "+v"(row_at));
                const float *row = strip + row_at;
                const f32x2 s2{s, s};
#pragma unroll
                for (int c = 0; c < NP; c++) accumulate2<CMP>(acc2[c], s2, f32x2{row[2 * c], row[2 * c + 1]}); // (the word behind an odd count: unused)
            }
        }
#pragma unroll
        for (int c = 0; c < NC; c++) acc[c] = (c & 1) ? acc2[c / 2].y : acc2[c / 2].x;
        if (served) {
            if (__ballot(bad_centre) != 0 || bad_px) redo = true; // (a denormal quotient somewhere: this pixel again, with divisions)
            else emit_pixel<R>(acc, g, d0, p, disp, tcv);
            todo = false;
        }
    }
    if (todo) { // the walk of what is left: this pixel's offsets on their own, a few at a time
        bool bad = false;
#pragma unroll
        for (int c = 0; c < NC; c++) acc[c] = 0.0f;
        walk_offsets<CMP, ZM, NORM, HR, C, NC, 0>(tgt, mean_t, norm_t, H, Wt, stile + lane * C, SW, v, src.v_r, i, first, ms, rs, acc, bad);
        if (bad) redo = true;
        else emit_pixel<R>(acc, g, d0, p, disp, tcv);
    }
    if (redo) guided_fused_px<CMP, ZM, NORM, R>(src, tgt, mean_s, norm_s, mean_t, norm_t, H, Ws, Wt, g, disp, tcv, p, i, j, d0);
}

template <int CMP, bool ZM, bool NORM, int HR, int SPAN, int C>
bool launch_guided_wave_span(svh_context *ctx, FeatImage src, FeatImage tgt, const float *ms, const float *ns, const float *mt, const float *nt, int H, int Ws,
                             int Wt, const GuideArgs &g, int32_t *disp, float *tcv) {
    constexpr int h = 2 * HR + 1;
    const int v = 2 * src.v_r + 1;
    const size_t shmem = (size_t)(v * (64 + h - 1) * C + v * (SPAN + 2 * h - 1) * C + h * C * (SPAN + 1)) * sizeof(float);
    if (shmem > 60 * 1024) return false;
    const dim3 grid(ceil_div(Ws, 64), H);
    switch (g.radius) {
    case 1: SVH_LAUNCH(ctx, "guided_fused", (guided_wave_kernel<CMP, ZM, NORM, 1, HR, SPAN, C>), grid, 64, shmem, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv); return true;
    case 2: SVH_LAUNCH(ctx, "guided_fused", (guided_wave_kernel<CMP, ZM, NORM, 2, HR, SPAN, C>), grid, 64, shmem, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv); return true;
    case 3: SVH_LAUNCH(ctx, "guided_fused", (guided_wave_kernel<CMP, ZM, NORM, 3, HR, SPAN, C>), grid, 64, shmem, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv); return true;
    default: return false;
    }
}

// grey images: 96 staged centres; RGB: 76 or 80 (three times the rows and the strip in LDS: a CU holds eight or seven waves of it)
template <int CMP, bool ZM, bool NORM, int HR>
bool launch_guided_wave_radius(svh_context *ctx, FeatImage src, FeatImage tgt, const float *ms, const float *ns, const float *mt, const float *nt, int H,
                               int Ws, int Wt, const GuideArgs &g, int32_t *disp, float *tcv) {
    if (src.C == 3) { // (76 centres: eight waves per CU instead of seven; the thirteen offsets of radius 3 keep four centres of room)
        if (g.radius <= 2) return launch_guided_wave_span<CMP, ZM, NORM, HR, 76, 3>(ctx, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv);
        return launch_guided_wave_span<CMP, ZM, NORM, HR, 80, 3>(ctx, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv);
    }
    return launch_guided_wave_span<CMP, ZM, NORM, HR, 96, 1>(ctx, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv);
}

// the eight combinations the matching functions have: CC / NCC / ZCC / ZNCC, SSD / ZSSD, SAD / ZSAD
template <int HR>
bool launch_guided_wave_hr(svh_context *ctx, int cmp, bool zm, bool nrm, FeatImage src, FeatImage tgt, const float *ms, const float *ns, const float *mt,
                           const float *nt, int H, int Ws, int Wt, const GuideArgs &g, int32_t *disp, float *tcv) {
#define SVH_GW(CMPV, ZMV, NRMV) return launch_guided_wave_radius<CMPV, ZMV, NRMV, HR>(ctx, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv)
    if (cmp == CMP_DOT) {
        if (zm && nrm) SVH_GW(CMP_DOT, true, true);
        if (zm) SVH_GW(CMP_DOT, true, false);
        if (nrm) SVH_GW(CMP_DOT, false, true);
        SVH_GW(CMP_DOT, false, false);
    }
    if (nrm) return false;
    if (cmp == CMP_SSD) {
        if (zm) SVH_GW(CMP_SSD, true, false);
        SVH_GW(CMP_SSD, false, false);
    }
    if (cmp == CMP_SAD) {
        if (zm) SVH_GW(CMP_SAD, true, false);
        SVH_GW(CMP_SAD, false, false);
    }
#undef SVH_GW
    return false;
}

} // namespace

} // namespace svh

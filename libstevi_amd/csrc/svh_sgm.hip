// SGM path aggregation (A9), reproducing correlation/sgm.h as written.
//
// Effective passes of the reference for nDirections = 8, in accumulation order (sgm.h:379-388 with the start
// rule of :329-354; Down2Up, Right2Left and DownRight2UpLeft start one past the last valid index and do
// nothing):
//   0 Up2Down            lines from (top, j)   step (+1, 0)
//   1 Left2Right         lines from (i, left)  step ( 0,+1)
//   2 UpLeft2DownRight   lines from (i, left)  step (+1,+1)   (row starts)
//   3 UpLeft2DownRight   lines from (top, j)   step (+1,+1)   (column starts; the corner line runs twice)
//   4 UpRight2DownLeft   lines from (top, j)   step (+1,-1)
//   5 DownLeft2UpRight   lines from (i, left)  step (-1,+1)
// nDirections = 4 keeps passes 0-1.  Lines of one pass are disjoint, so a launch per pass is race free and the
// float accumulation order of the reference (pass after pass) is kept.
//
// Mapping: one wavefront per line, the 64 lanes span the disparity axis with R = ceil(D/64) consecutive
// disparities per lane, so a pixel's D costs are one contiguous 4*D-byte read (volumes are (row, col, disparity)
// with the disparity fastest).  No MFMA: there is no contraction here; the cross-lane work is a wave min / max
// and prefix/suffix maxima done with lane shuffles.
//
// Cost branch (sgm.h:257-296).  Because of `min_a_cost = c_score` (:281-283, F4) the recurrence couples
// disparities only through the per-pixel scalar min_p = min over finite previous costs:
//     actual(d) = c(d) + ((c(d) [+ Pout if j+d >= W]) - min_p)     when both terms are finite, else c(d)
// So a pass is fully described by the map pixel -> min_p.  The line kernel computes that map (the sequential
// part, with exactly the reference's float operations), and one "apply" kernel then rebuilds
//     S = C; S += actual_q - C for every pass q that visits the pixel, in pass order
// per voxel, again with the reference's operations and order, so S is bit-identical while the volume is
// written once instead of being read-modified-written per pass; the winner scan can be fused into it.
//
// Score branch (sgm.h:218-255) couples neighbouring disparities, so each pass is a read-modify-write sweep:
//     a(nd) = max over finite { prev[nd], prev[nd-1]-P1, prev[nd+1]-P1, max_{|od-nd|>1} prev[od] - P2 }
// The last term is evaluated exactly for any P1, P2 from exclusive prefix / suffix maxima (x -> fl(x - P2) is
// monotone, so the max of the differences is the difference of the max).
#include "svh_sgm_lines.h"

#include <type_traits>

namespace svh {

struct SrcCensus { // Hamming cost evaluated from compact census words (H, W*, nWw); target words pre-rounded (E2)
    const uint32_t *sw, *tw;
    int nWw, Ws, Wt, sign, disp_lower, D;
    template <int R> __device__ __forceinline__ void load(int i, int j, int lane, float (&c)[R]) const {
        const uint32_t *s = sw + ((int64_t)i * Ws + j) * nWw;
        const uint32_t *trow = tw + (int64_t)i * Wt * nWw;
#pragma unroll
        for (int k = 0; k < R; k++) {
            int d = lane * R + k;
            int jt = j + sign * (disp_lower + d);
            bool in = d < D && jt >= 0 && jt < Wt;
            uint32_t acc = 0;
            for (int w = 0; w < nWw; w++) acc += __popc(s[w] ^ (in ? trow[(int64_t)jt * nWw + w] : 0u));
            c[k] = (float)acc;
        }
    }
};

// ---- Cost branch: per-pass map pixel -> min_p ----------------------------------------------------------
// One step of sgm.h:259-295 for the R disparities of this lane; returns the lane's min over finite actual costs.
template <int R>
__device__ __forceinline__ float cost_step_lane_min(const float (&c)[R], int lane, int D, int j, int W, float Pout, float mp) {
    float lm = INFINITY;
    const bool mp_fin = finite_f(mp);
#pragma unroll
    for (int k = 0; k < R; k++) {
        int d = lane * R + k;
        float t = (j + d >= W) ? c[k] + Pout : c[k]; // min_a_cost = c_score (+ Pout), :281-289
        float act = c[k];
        if (mp_fin && finite_f(t)) act = c[k] + (t - mp); // :291-294
        if (d < D && finite_f(act)) lm = fminf(lm, act);  // next pixel's min over finite previous costs, :261-266
    }
    return lm;
}

// the lines of all the passes of one aggregation in one launch: pass q owns blocks [first_block[q], first_block[q + 1]) (four lines each)
struct PassSets {
    LineSet ls[6];
    int first_block[7];
    int n_pass;
};
template <class SRC, int R, int B>
__global__ void __launch_bounds__(256) sgm_cost_minmap_kernel(SRC src, PassSets sets, int D, int W, int64_t npx, float Pout,
                                                             float *__restrict__ mmap_base, const int *__restrict__ gate, int gate_mask) {
    if (gate && (*gate & gate_mask) == 0) return; // another route already produced the maps (exact-integer scans / the two-minima recurrences)
    const int lane = threadIdx.x & 63;
    int q = 0;
    while (q + 1 < sets.n_pass && (int)blockIdx.x >= sets.first_block[q + 1]) q++; // (block uniform)
    const LineSet ls = sets.ls[q];
    float *__restrict__ mmap = mmap_base + (int64_t)min_p_plane(q) * npx;
    const int l = ((int)blockIdx.x - sets.first_block[q]) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (l >= ls.n_lines) return;
    const Line L = line_of(ls, l);
    float mp = 0.0f; // previous_cost[] = 0 -> min over finite = 0 (sgm.h:206-208)
    // The loads do not depend on the recurrence.  Two register batches: while the serial part walks batch `cur`, the
    // loads of batch `nxt` are already in flight (a line is one wave, so nothing else hides the HBM latency).
    float cur[B][R], nxt[B][R];
    if constexpr (std::is_same<SRC, SrcVolume>::value && R % 4 == 0) {
        // 16-byte aligned volume with D = 64 R: whole batches with unconditional vector loads, so that the compiler's count of loads
        // in flight is exact and walking a batch does not wait for the loads of the next one (with a load under a branch every step
        // ended in s_waitcnt vmcnt(0)).  The map store is hidden from that count (inline asm): one dword per pixel.
        if (src.vec && D == 64 * R) {
            const int nb = L.len / B;
            auto px = [&](int t) { return src.cv + ((int64_t)(L.i0 + t * L.di) * W + (L.j0 + t * L.dj)) * D + lane * R; };
            auto load_full = [&](float (&c)[B][R], int k) {
                k = min(k, nb - 1); // (past the end: the last whole batch again, unused)
#pragma unroll
                for (int b = 0; b < B; b++) {
                    const float *p = px(k * B + b);
#pragma unroll
                    for (int q = 0; q < R / 4; q++) {
                        const float4 v = *reinterpret_cast<const float4 *>(p + 4 * q);
                        c[b][4 * q] = v.x; c[b][4 * q + 1] = v.y; c[b][4 * q + 2] = v.z; c[b][4 * q + 3] = v.w;
                    }
                }
            };
            auto one = [&](const float (&c)[R], int t) {
                const int ii = L.i0 + t * L.di, jj = L.j0 + t * L.dj;
                if (lane == 0) {
                    float *dst = mmap + (int64_t)ii * W + jj;
                    asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(mp) : "memory");
                }
                mp = wave_min(cost_step_lane_min<R>(c, lane, D, jj, W, Pout, mp));
            };
            auto run_full = [&](const float (&c)[B][R], int k) {
#pragma unroll
                for (int b = 0; b < B; b++) one(c[b], k * B + b);
            };
            if (nb > 0) {
                load_full(cur, 0);
                for (int k = 0; k < nb; k += 2) {
                    load_full(nxt, k + 1);
                    run_full(cur, k);
                    if (k + 1 >= nb) break;
                    load_full(cur, k + 2);
                    run_full(nxt, k + 1);
                }
            }
            for (int t = nb * B; t < L.len; t++) {
                src.template load<R>(L.i0 + t * L.di, L.j0 + t * L.dj, lane, cur[0]);
                one(cur[0], t);
            }
            return;
        }
    }
    auto load_batch = [&](float (&c)[B][R], int s0) {
#pragma unroll
        for (int b = 0; b < B; b++)
            if (s0 + b < L.len) src.template load<R>(L.i0 + (s0 + b) * L.di, L.j0 + (s0 + b) * L.dj, lane, c[b]);
    };
    auto run_batch = [&](const float (&c)[B][R], int s0) {
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (s0 + b < L.len) {
                const int ii = L.i0 + (s0 + b) * L.di, jj = L.j0 + (s0 + b) * L.dj;
                if (lane == 0) mmap[(int64_t)ii * W + jj] = mp;
                mp = wave_min(cost_step_lane_min<R>(c[b], lane, D, jj, W, Pout, mp));
            }
        }
    };
    load_batch(cur, 0);
    for (int s0 = 0; s0 < L.len; s0 += 2 * B) {
        load_batch(nxt, s0 + B);
        run_batch(cur, s0);
        load_batch(cur, s0 + 2 * B);
        run_batch(nxt, s0 + B);
    }
}

// order-preserving key of a float for unsigned comparison; -0 is folded onto +0 (they compare equal)
__device__ __forceinline__ uint32_t float_order_key(float v) {
    if (v == 0.0f) v = 0.0f;
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct Winner {
    float v;
    int d; // -1: no candidate yet
};

// combine rule of extractSelectedIndex (correlation_base.h:441-455) for two partial scans over disjoint
// disparity sets: the extremum wins, ties go to the larger index; NaN candidates never enter (handled by caller)
template <bool COST> __device__ __forceinline__ Winner better(Winner a, Winner b) {
    if (b.d < 0) return a;
    if (a.d < 0) return b;
    bool b_wins = COST ? (b.v < a.v || (b.v == a.v && b.d > a.d)) : (b.v > a.v || (b.v == a.v && b.d > a.d));
    return b_wins ? b : a;
}

// Winner over the non-NaN values of the wave (d = -1 when there is none) and whether the value at local index 0
// is NaN: the sequential scan of the reference starts from index 0 and never leaves it when that value is NaN.
// Wave extremum by DPP (v_min / v_max ignore NaN), a ballot of the lanes that hold a value EQUAL to it (a NaN equals nothing), the highest
// such lane and its largest such k: ties go to the larger index.  (The first form exchanged (value, index) pairs through six
// ds_bpermute rounds: the fused winner made sgm_cost_apply 35 % longer than the same kernel writing the whole volume.)
template <bool COST, int R>
__device__ __forceinline__ Winner wave_select(const float (&s)[R], int lane, int D, bool *first_is_nan) {
    const bool full = D == 64 * R; // (wave-uniform) every lane's R values exist: no per-value guard
    float A = COST ? INFINITY : -INFINITY;
    if (full) {
#pragma unroll
        for (int k = 0; k < R; k++) A = COST ? fminf(A, s[k]) : fmaxf(A, s[k]);
    } else {
#pragma unroll
        for (int k = 0; k < R; k++) {
            const float x = (lane * R + k < D) ? s[k] : __uint_as_float(0x7FC00000u);
            A = COST ? fminf(A, x) : fmaxf(A, x);
        }
    }
    const float M = COST ? wave_min(A) : wave_max_dpp(A); // +-inf when every value is NaN (then nothing equals it unless a real +-inf does)
    int kb = -1;
    if (full) {
#pragma unroll
        for (int k = 0; k < R; k++) kb = (s[k] == M) ? k : kb;
    } else {
#pragma unroll
        for (int k = 0; k < R; k++) kb = (lane * R + k < D && s[k] == M) ? k : kb;
    }
    const unsigned long long holders = __builtin_amdgcn_ballot_w64(kb >= 0);
    *first_is_nan = (__builtin_amdgcn_ballot_w64(isnan(s[0])) & 1ull) != 0;
    if (holders == 0ull) return Winner{0.0f, -1};
    const int top = 63 - __builtin_clzll(holders);
    return Winner{M, top * R + __builtin_amdgcn_readlane(kb, top)};
}

// (ordered value, index) key for a cross-shard min (COST) / max (SCORE) reduction that reproduces the tie rule
template <bool COST>
__device__ __forceinline__ unsigned long long winner_key(Winner w, bool first_is_nan, int key_offset, int key_total) {
    if (key_offset == 0 && first_is_nan) // global index 0 is NaN: it wins unconditionally
        return COST ? (unsigned long long)(uint32_t)(key_total - 1) : (0xFFFFFFFFull << 32);
    if (w.d < 0) return COST ? ~0ull : 0ull; // nothing comparable in this shard
    const uint32_t gd = (uint32_t)(key_offset + w.d);
    return ((unsigned long long)float_order_key(w.v) << 32) | (COST ? (uint32_t)(key_total - 1) - gd : gd);
}

// Stores the compiler does not count.  Its s_waitcnt bookkeeping treats a store under a branch as "maybe outstanding" and answers with
// vmcnt(0) at the next use of a load -- which also waits for the loads of the NEXT batch that were issued on purpose.  These outputs are never
// read back by the kernel, and an uncounted store only makes a counted wait more conservative (vmcnt counts stores too on gfx9).
__device__ __forceinline__ void store_uncounted(float *p, float v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v)); }
__device__ __forceinline__ void store_uncounted(int *p, int v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v)); }
__device__ __forceinline__ void store_uncounted(unsigned long long *p, unsigned long long v) { asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v)); }
__device__ __forceinline__ void store_uncounted(float4 *p, float4 v) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 r = {v.x, v.y, v.z, v.w};
    // (a store of more than 64 bits must not be followed at once by a VALU write of its data registers -- one wait state on gfx9, two on
    // gfx94x/95x; the compiler pads its own stores and knows nothing of this one: without the s_nop a few lanes stored the next address
    // computation instead of their four values)
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(r));
}

struct ApplyOut {
    float *sgm; // (H, W, D) or nullptr
    WinnerOut w;
    bool vec_store; // sgm rows 16-byte aligned, D % 4 == 0
};

template <class SRC, int R>
__global__ void __launch_bounds__(256) sgm_cost_apply_kernel(SRC src, int H, int W, int D, int top, int left, int Hp, int Wp,
                                                            int n_pass, float Pout, const float *__restrict__ mmap,
                                                            ApplyOut out, const int *__restrict__ regime_flag, int regime_vouched) {
    const int lane = threadIdx.x & 63;
    const int64_t npx = (int64_t)H * W;
    // In the regime the two-minima route established (no finite |c| above 1e30: bit 1 of the flag down) and with a finite Pout (then
    // |Pout| <= 1e30) `isfinite(t)` is redundant once mp is finite: for a non-finite c the sum c + (t - mp) IS c (same-signed infinities; NaN stays NaN)
    // and a finite c cannot overflow t.  Four operations per voxel and pass instead of eight: with the winner fused behind it this kernel
    // is bound by vector issue, not by its one read of the volume.
    // (regime_vouched: the host knows already -- small integer costs stated by the caller, svh_sgm_cost_volume_minima)
    const bool lean = (regime_vouched || (regime_flag && (*regime_flag & 2) == 0)) && finite_f(Pout);
    // (pixel indices are wave-uniform and fit 32 bits -- the library takes images of fewer than 2^31 pixels: kept scalar, one division per
    // APPLY_PB pixels instead of two 64-bit ones per pixel and lane)
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int npx32 = (int)npx;
    // The min_p values of a pixel are wave-uniform.  Loaded as scalars (24 of them per iteration, with the output descriptor and the pixels'
    // coordinates) they spilled 67 SGPRs, 440 v_readlane per iteration; and the loop over the passes, unrolled over 4 pixels x 6 passes x
    // {general, lean, lean at the border}, was 60 KB of code with some 110 scalar branches per pixel on its hot path.  Now LANE q of one
    // vector register per pixel holds the min_p of pass q (one dword load per pixel), the pass loop is a runtime loop that fetches its
    // min_p with v_readlane, and the body exists once per prefetched pixel.
    const int lq = min(lane, max(n_pass, 1) - 1);
    const float *__restrict__ mp_of_lane = mmap + (int64_t)min_p_plane(lq) * npx;
    const unsigned pass_mask = (1u << n_pass) - 1u;
    // APPLY_PB pixels per wave iteration: their cost rows and min_p values are all requested before the first one is used
    // (one pixel at a time leaves a single 1 KiB load in flight per wave and the kernel latency-bound).  Two such batches alternate, but the
    // loads of this general form sit under branches (the guarded tail of a row, the census source), so the compiler waits with vmcnt(0) and the
    // alternation buys nothing here: sgm_cost_apply_piped_kernel below is the form for aligned dense volumes.
    constexpr int APPLY_PB = R <= 4 ? 4 : R <= 8 ? 2 : 1;
    struct Batch {
        float cb[APPLY_PB][R], mpl[APPLY_PB];
    };
    const int64_t stride = (int64_t)nwaves * APPLY_PB;
    auto load_batch = [&](Batch &B, int64_t q0) {
      const int p0 = (int)min(q0, (int64_t)npx32 - 1); // (past the end: the last pixel again, unused)
      const int i0 = (int)((unsigned)p0 / (unsigned)W), j0 = p0 - i0 * W;
#pragma unroll
      for (int u = 0; u < APPLY_PB; u++) {
          const int pu = min(p0 + u, npx32 - 1);
          int jj = j0 + (pu - p0), ii = i0;
          while (jj >= W) { // (an image narrower than APPLY_PB pixels wraps more than once)
              jj -= W;
              ii++;
          }
          src.template load<R>(ii, jj, lane, B.cb[u]);
          B.mpl[u] = n_pass > 0 ? mp_of_lane[pu] : 0.0f;
      }
    };
    auto run_batch = [&](const Batch &B, int64_t q0) {
      if (q0 >= npx) return;
      const int p0 = (int)q0;
      const int i0 = (int)((unsigned)p0 / (unsigned)W), j0 = p0 - i0 * W;
      const auto &cb = B.cb;
      const auto &mpl = B.mpl;
#pragma unroll
      for (int u = 0; u < APPLY_PB; u++) {
        const int64_t p = (int64_t)p0 + u;
        if (p >= npx) break;
        int j = j0 + u, i = i0;
        while (j >= W) {
            j -= W;
            i++;
        }
        float c[R], s[R];
#pragma unroll
        for (int k = 0; k < R; k++) s[k] = c[k] = cb[u][k]; // sgm_cv := cv, sgm.h:371-377
        const int ip = i - top, jp = j - left;
        unsigned visits = 0u; // bit q: pass q visits this pixel (pass_visits(), svh_sgm_lines.h)
        if (ip >= 0 && ip < Hp && jp >= 0 && jp < Wp)
            visits = (3u | (ip >= jp ? 4u : 0u) | (jp >= ip ? 8u : 0u) | (ip + jp < Wp ? 16u : 0u) | (ip + jp < Hp ? 32u : 0u)) & pass_mask;
        const bool at_border = j + D > W;
        // The common pixel -- every disparity looks inside the image, the min_p of the passes that visit it finite, the lean regime -- takes the
        // passes that visit it (four on average: two always, one of the two diagonal halves, one or both anti-diagonal ones) in order, four packed
        // operations per pass and pair of disparities, with no other decision inside the loop.  (The one loop that decided regime, border and
        // finiteness per pass came out at 15 vector instructions and 17 scalar ones per pass, and the kernel was bound by them, not by its read.)
        unsigned todo = visits;
        if constexpr (R % 2 == 0) {
            const unsigned fin = (unsigned)__builtin_amdgcn_ballot_w64(finite_f(mpl[u])); // bit q: min_p of pass q finite (lane q holds it)
            if (lean && !at_border && (visits & ~fin) == 0u) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 c2[R / 2], s2[R / 2];
#pragma unroll
                for (int h = 0; h < R / 2; h++) {
                    c2[h] = f32x2{c[2 * h], c[2 * h + 1]};
                    s2[h] = c2[h];
                }
                while (todo) {
                    const int q = __builtin_ctz(todo);
                    todo &= todo - 1u;
                    const float mp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mpl[u]), q));
                    const f32x2 mp2 = {mp, mp};
#pragma unroll
                    for (int h = 0; h < R / 2; h++) s2[h] += (c2[h] + (c2[h] - mp2)) - c2[h]; // sgm.h:291-300 with t = c, both terms finite
                }
#pragma unroll
                for (int h = 0; h < R / 2; h++) {
                    s[2 * h] = s2[h][0];
                    s[2 * h + 1] = s2[h][1];
                }
            }
        }
        for (int q = 0; q < n_pass; q++) { // (every condition in this loop is wave-uniform)
            if (!((todo >> q) & 1u)) continue;
            const float mp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mpl[u]), q));
            const bool mp_fin = finite_f(mp);
            if (lean && mp_fin) {
                if (at_border) {
#pragma unroll
                    for (int k = 0; k < R; k++) {
                        const float t = (j + lane * R + k >= W) ? c[k] + Pout : c[k];
                        s[k] += (c[k] + (t - mp)) - c[k];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < R; k++) s[k] += (c[k] + (c[k] - mp)) - c[k];
                }
                continue;
            }
#pragma unroll
            for (int k = 0; k < R; k++) {
                int d = lane * R + k;
                float t = (j + d >= W) ? c[k] + Pout : c[k];
                float act = c[k];
                if (mp_fin && finite_f(t)) act = c[k] + (t - mp);
                s[k] += act - c[k]; // sgm.h:298-300
            }
        }
        if (out.sgm) {
            float *o = out.sgm + p * D + lane * R;
            bool stored = false;
            if constexpr (R % 4 == 0) {
                if (out.vec_store && lane * R + R <= D) {
#pragma unroll
                    for (int q = 0; q < R / 4; q++) store_uncounted(reinterpret_cast<float4 *>(o + 4 * q), make_float4(s[4 * q], s[4 * q + 1], s[4 * q + 2], s[4 * q + 3]));
                    stored = true;
                }
            }
            if (!stored) {
#pragma unroll
                for (int k = 0; k < R; k++)
                    if (lane * R + k < D) store_uncounted(o + k, s[k]);
            }
        }
        const WinnerOut &wo = out.w;
        if (wo.idx || wo.disp || wo.taps || wo.keys) {
            bool first_nan;
            const Winner w = wave_select<true, R>(s, lane, D, &first_nan);
            const int sel = (first_nan || w.d < 0) ? 0 : w.d;
            if (wo.idx && lane == 0) store_uncounted(wo.idx + p, sel);
            if (wo.disp && lane == 0) store_uncounted(wo.disp + p, wo.disp_sign * sel + wo.disp_offset);
            if (wo.keys && lane == 0) store_uncounted(wo.keys + p, winner_key<true>(w, first_nan, wo.key_offset, wo.key_total));
            if (wo.taps) {
                // truncatedCostVolume<Same>, radius 1 (correlation_base.h:601-613)
                const bool px_bad = j < wo.taps_h_r || i < wo.taps_v_r || i + wo.taps_v_r >= H;
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const int pd = sel + t - 1;
                    const bool bad = px_bad || pd < 0 || pd >= D || j + pd + wo.taps_h_r >= W;
                    if (bad) {
                        if (lane == 0) store_uncounted(wo.taps + p * 3 + t, __uint_as_float(0x7FC00000u));
                    } else {
#pragma unroll
                        for (int k = 0; k < R; k++)
                            if (lane * R + k == pd) store_uncounted(wo.taps + p * 3 + t, s[k]);
                    }
                }
            }
        }
      }
    };
    Batch A, B;
    load_batch(A, (int64_t)wave * APPLY_PB);
    for (int64_t q0 = (int64_t)wave * APPLY_PB; q0 < npx; q0 += 2 * stride) {
        load_batch(B, q0 + stride);
        run_batch(A, q0);
        load_batch(A, q0 + 2 * stride);
        run_batch(B, q0 + stride);
    }
}

// ---- the same for a dense volume with 16-byte aligned rows and D % 4 == 0 (R % 4 == 0): software-pipelined, scalar-lean --------------
// Counters of the general kernel above at 1920 x 1080 x 256 (rocprofv3 --pmc, per pixel): 105 vector, 136 scalar and 39 branch instructions.
// A CU issues one scalar instruction per cycle for its four SIMDs, so 136 scalar instructions per pixel are 0.46 ms of scalar issue alone
// -- more than the kernel's one read of the volume (0.42 ms); and every wait was an s_waitcnt vmcnt(0) (loads under a branch, stores under
// `if (lane == 0)`), so a wave had nothing in flight while it worked.  This form
//  * loads rows through buffer descriptors that cover exactly one pixel's row (a lane past D reads zeros, what the guarded loads give it):
//    every load is unconditional, the compiler's count of loads in flight is exact, and two batches alternate for real;
//  * issues its stores as inline asm the compiler does not count (they are never read back), addressed from a scalar base;
//  * computes what depends on the pixel's coordinates (which passes visit it, whether a disparity looks past the border, the limit of the
//    refinement taps) in the LANES of one register per batch -- lane u for pixel u -- next to the loads, and fetches it with one v_readlane;
//  * collects the winners of a batch in lanes and stores each output array once per batch, not once per pixel under its own exec mask.
template <int OFF> __device__ __forceinline__ void store_row_b128(const float *sbase, int voff, float a, float b, float c, float d) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 r = {a, b, c, d};
    // (the s_nop: see store_uncounted(float4 *, float4))
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(voff), "v"(r), "s"(sbase), "n"(OFF));
}
template <int R, int Q = 0> __device__ __forceinline__ void store_row(const float *sbase, int voff, const float (&s)[R], int lane, int D, bool full) {
    if constexpr (Q < R / 4) {
        if (full || lane * R + 4 * Q + 4 <= D) store_row_b128<16 * Q>(sbase, voff, s[4 * Q], s[4 * Q + 1], s[4 * Q + 2], s[4 * Q + 3]);
        store_row<R, Q + 1>(sbase, voff, s, lane, D, full);
    }
}

template <int R>
__global__ void __launch_bounds__(256) sgm_cost_apply_piped_kernel(const float *__restrict__ cv, int H, int W, int D, int top, int left, int Hp, int Wp,
                                                                  int n_pass, float Pout, const float *__restrict__ mmap, ApplyOut out,
                                                                  const int *__restrict__ regime_flag, int regime_vouched) {
    static_assert(R % 4 == 0, "");
    const int lane = threadIdx.x & 63;
    const int64_t npx = (int64_t)H * W;
    const int npx32 = (int)npx;
    const bool lean = (regime_vouched || (regime_flag && (*regime_flag & 2) == 0)) && finite_f(Pout); // (see sgm_cost_apply_kernel)
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int lq = min(lane, max(n_pass, 1) - 1);
    const float *__restrict__ mp_of_lane = mmap + (int64_t)min_p_plane(lq) * npx; // lane q: the plane of pass q
    const unsigned pass_mask = (1u << n_pass) - 1u;
    const bool full = D == 64 * R;
    const WinnerOut &wo = out.w;
    const bool want_winner = wo.idx || wo.disp || wo.taps || wo.keys;
    constexpr int PB = R <= 4 ? 4 : R <= 8 ? 2 : 1;
    constexpr unsigned BORDER = 64u;
    struct Batch {
        float cb[PB][R], mpl[PB];
        unsigned info; // lane u: bits 0..5 the passes that visit pixel u of the batch (pass_visits(), svh_sgm_lines.h), bit 6: j + D > W
        int j, limit;  // lane u: the pixel's column; the first disparity whose refinement tap is outside (truncatedCostVolume<Same>), -2: all are
    };
    const int64_t stride = (int64_t)nwaves * PB;
    auto load_batch = [&](Batch &B, int64_t q0) {
        const int p0 = (int)min(q0, (int64_t)npx32 - 1); // (past the end: the last pixel again, unused)
        const int i0 = (int)((unsigned)p0 / (unsigned)W), j0 = p0 - i0 * W;
#pragma unroll
        for (int u = 0; u < PB; u++) {
            const int pu = min(p0 + u, npx32 - 1);
            const __amdgpu_buffer_rsrc_t row = __builtin_amdgcn_make_buffer_rsrc((void *)(cv + (int64_t)pu * D), 0, D * 4, 0x00020000);
#pragma unroll
            for (int q = 0; q < R / 4; q++) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(row, (lane * R + 4 * q) * 4, 0, 0);
                // (through named words: __builtin_bit_cast applied to a vector element reads element 0 whatever the subscript)
                const uint32_t w0 = v[0], w1 = v[1], w2 = v[2], w3 = v[3];
                B.cb[u][4 * q] = __uint_as_float(w0);
                B.cb[u][4 * q + 1] = __uint_as_float(w1);
                B.cb[u][4 * q + 2] = __uint_as_float(w2);
                B.cb[u][4 * q + 3] = __uint_as_float(w3);
            }
            B.mpl[u] = mp_of_lane[pu]; // (the scratch always holds every plane)
        }
        int jj = j0 + min(lane, PB - 1), ii = i0; // lane u: pixel p0 + u (past the last pixel of the image: a row that does not exist, unused)
        while (jj >= W) {                       // (an image narrower than PB pixels wraps more than once)
            jj -= W;
            ii++;
        }
        const int ip = ii - top, jp = jj - left;
        unsigned vis = 0u;
        if (ip >= 0 && ip < Hp && jp >= 0 && jp < Wp)
            vis = (3u | (ip >= jp ? 4u : 0u) | (jp >= ip ? 8u : 0u) | (ip + jp < Wp ? 16u : 0u) | (ip + jp < Hp ? 32u : 0u)) & pass_mask;
        B.info = vis | (jj + D > W ? BORDER : 0u);
        B.j = jj;
        const bool px_bad = jj < wo.taps_h_r || ii < wo.taps_v_r || ii + wo.taps_v_r >= H;
        B.limit = px_bad ? -2 : W - wo.taps_h_r - jj;
    };
    auto run_batch = [&](const Batch &B, int64_t q0) {
        if (q0 >= npx) return;
        const int p0 = (int)q0;
        const int nvalid = min(PB, npx32 - p0);
        // the winners of the batch, lane u for pixel u: extremum, local index (-1: none), "the value at index 0 is NaN"; lanes 3u .. 3u + 2: the taps
        int Mv = 0, dv = 0, fnv = 0;
        float tapv = 0.0f;
#pragma unroll
        for (int u = 0; u < PB; u++) {
            if (u >= nvalid) break;
            const unsigned info = (unsigned)__builtin_amdgcn_readlane((int)B.info, u);
            float c[R], s[R];
#pragma unroll
            for (int k = 0; k < R; k++) s[k] = c[k] = B.cb[u][k]; // sgm_cv := cv, sgm.h:371-377
            const unsigned visits = info & 63u;
            const unsigned fin = (unsigned)__builtin_amdgcn_ballot_w64(finite_f(B.mpl[u])); // bit q: min_p of pass q finite (lane q holds it)
            unsigned todo = visits;
            if (lean && (info & BORDER) == 0u && (visits & ~fin) == 0u) {
                // the common pixel: every disparity looks inside the image, finite min_p, the lean regime -- the passes that visit it in order,
                // four packed operations per pass and pair of disparities (sgm.h:291-300 with t = c and both terms finite)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 c2[R / 2], s2[R / 2];
#pragma unroll
                for (int h = 0; h < R / 2; h++) {
                    c2[h] = f32x2{c[2 * h], c[2 * h + 1]};
                    s2[h] = c2[h];
                }
                while (todo) {
                    const int q = __builtin_ctz(todo);
                    todo &= todo - 1u;
                    const float mp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, B.mpl[u]), q));
                    const f32x2 mp2 = {mp, mp};
#pragma unroll
                    for (int h = 0; h < R / 2; h++) s2[h] += (c2[h] + (c2[h] - mp2)) - c2[h];
                }
#pragma unroll
                for (int h = 0; h < R / 2; h++) {
                    s[2 * h] = s2[h][0];
                    s[2 * h + 1] = s2[h][1];
                }
            }
            if (todo) { // a pixel at the border, a non-finite min_p, or not the lean regime: the step as written
                const int j = __builtin_amdgcn_readlane(B.j, u);
                for (int q = 0; q < n_pass; q++) { // (every condition in this loop is wave-uniform)
                    if (!((todo >> q) & 1u)) continue;
                    const float mp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, B.mpl[u]), q));
                    const bool mp_fin = finite_f(mp);
                    if (lean && mp_fin) {
#pragma unroll
                        for (int k = 0; k < R; k++) {
                            const float t = (j + lane * R + k >= W) ? c[k] + Pout : c[k];
                            s[k] += (c[k] + (t - mp)) - c[k];
                        }
                        continue;
                    }
#pragma unroll
                    for (int k = 0; k < R; k++) {
                        const int d = lane * R + k;
                        const float t = (j + d >= W) ? c[k] + Pout : c[k];
                        float act = c[k];
                        if (mp_fin && finite_f(t)) act = c[k] + (t - mp);
                        s[k] += act - c[k]; // sgm.h:298-300
                    }
                }
            }
            if (out.sgm) store_row<R>(out.sgm + (int64_t)(p0 + u) * D, lane * R * 4, s, lane, D, full);
            if (!want_winner) continue;
            // extractSelectedIndex (correlation_base.h:441-455): the extremum over the non-NaN values, ties to the larger index -- the wave
            // extremum by DPP, a ballot of the lanes holding a value EQUAL to it, the highest such lane and its largest such k
            float A = INFINITY;
            int kb = -1;
            if (full) {
#pragma unroll
                for (int k = 0; k < R; k++) A = fminf(A, s[k]);
            } else {
#pragma unroll
                for (int k = 0; k < R; k++) A = fminf(A, (lane * R + k < D) ? s[k] : INFINITY);
            }
            const float M = wave_min(A); // +inf when every value is NaN (then nothing equals it unless a real +inf does)
            if (full) {
#pragma unroll
                for (int k = 0; k < R; k++) kb = (s[k] == M) ? k : kb;
            } else {
#pragma unroll
                for (int k = 0; k < R; k++) kb = (lane * R + k < D && s[k] == M) ? k : kb;
            }
            const unsigned long long holders = __builtin_amdgcn_ballot_w64(kb >= 0);
            const int first_nan = (int)(__builtin_amdgcn_ballot_w64(isnan(s[0])) & 1ull);
            const int wl = holders ? 63 - __builtin_clzll(holders) : 0; // the winning lane
            const int wd = holders ? wl * R + __builtin_amdgcn_readlane(kb, wl) : -1;
            Mv = lane == u ? __builtin_bit_cast(int, M) : Mv;
            dv = lane == u ? wd : dv;
            fnv = lane == u ? first_nan : fnv;
            if (wo.taps) {
                // truncatedCostVolume<Same>, radius 1 (correlation_base.h:601-613) around the selected index: every lane builds the triple of ITS
                // candidate from its own registers and its neighbours' edge values (two DPP moves); the lane that holds the selected index is read
                const int sel = (first_nan || wd < 0) ? 0 : wd;
                const int sl = sel / R, sk = sel % R; // (R is a power of two)
                const float prevL = lane_shift_up(s[R - 1], 0.0f), nextR = lane_shift_down(s[0], 0.0f);
                float t0 = s[0], tm1 = prevL, tp1 = R > 1 ? s[R > 1 ? 1 : 0] : nextR;
#pragma unroll
                for (int k = 1; k < R; k++) {
                    const bool here = sk == k;
                    t0 = here ? s[k] : t0;
                    tm1 = here ? s[k - 1] : tm1;
                    tp1 = here ? (k + 1 < R ? s[k + 1 < R ? k + 1 : k] : nextR) : tp1;
                }
                const int limit = __builtin_amdgcn_readlane(B.limit, u); // valid: 0 <= pd < D and pd < limit
                const float nan = __uint_as_float(0x7FC00000u);
                const float r0 = (sel == 0 || sel - 1 >= limit) ? nan : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tm1), sl));
                const float r1 = (sel >= limit) ? nan : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t0), sl));
                const float r2 = (sel + 1 >= D || sel + 1 >= limit) ? nan : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tp1), sl));
                tapv = lane == 3 * u ? r0 : tapv;
                tapv = lane == 3 * u + 1 ? r1 : tapv;
                tapv = lane == 3 * u + 2 ? r2 : tapv;
            }
        }
        if (!want_winner) return;
        if (lane < nvalid) {
            const int64_t p = (int64_t)p0 + lane;
            const Winner w{__builtin_bit_cast(float, Mv), dv};
            const int sel = (fnv || dv < 0) ? 0 : dv;
            if (wo.idx) store_uncounted(wo.idx + p, sel);
            if (wo.disp) store_uncounted(wo.disp + p, wo.disp_sign * sel + wo.disp_offset);
            if (wo.keys) store_uncounted(wo.keys + p, winner_key<true>(w, fnv != 0, wo.key_offset, wo.key_total));
        }
        if (wo.taps && lane < 3 * nvalid) store_uncounted(wo.taps + (int64_t)p0 * 3 + lane, tapv);
    };
    Batch A, B;
    load_batch(A, (int64_t)wave * PB);
    for (int64_t q0 = (int64_t)wave * PB; q0 < npx; q0 += 2 * stride) {
        load_batch(B, q0 + stride);
        run_batch(A, q0);
        load_batch(A, q0 + 2 * stride);
        run_batch(B, q0 + stride);
    }
}

// ---- ... and for 128 disparities or fewer: several pixels per wave ----------------------------------------------------------------------
// With one or two disparities per lane a wave-per-pixel kernel issues a pixel's whole instruction stream for 64 or 128 costs: at 1080p the
// general kernel takes 0.56 ms for D = 64 and 0.57 ms for D = 128 -- the same as for 256 -- where the volume's read is 0.10 / 0.21 ms.
// Here a lane holds FOUR disparities of one pixel (a 16-byte load) and a pixel takes LPP = 16 (D <= 64) or 32 (D <= 128) lanes, so a wave
// works on four or two neighbouring pixels at once.  What was wave-uniform scalar state per pixel (coordinates, visiting passes, min_p) is
// per-lane vector state now; the common case -- the lean regime, no disparity past the border, every lane's pixel visited by the same
// passes with finite min_p -- is detected per group with one ballot and runs the same packed pass loop; everything else takes the step as
// written with per-lane predicates.  Reductions stay inside a pixel's lanes: row_ror DPP steps (an all-reduce within a row of 16) and one
// v_permlane16_swap for the two rows of a 32-lane pixel.  The lane that holds the winner writes the pixel's outputs.
template <int LPP>
__global__ void __launch_bounds__(256) sgm_cost_apply_packed_kernel(const float *__restrict__ cv, int H, int W, int D, int top, int left, int Hp, int Wp,
                                                                   int n_pass, float Pout, const float *__restrict__ mmap, ApplyOut out,
                                                                   const int *__restrict__ regime_flag, int regime_vouched) {
    static_assert(LPP == 16 || LPP == 32, "");
    constexpr int PPW = 64 / LPP, G = 2, PB = G * PPW; // pixels per wave and group; groups, pixels per batch
    const int lane = threadIdx.x & 63, sub = lane / LPP, dl = lane % LPP, d0 = dl * 4;
    const int64_t npx = (int64_t)H * W;
    const int npx32 = (int)npx;
    const bool lean = (regime_vouched || (regime_flag && (*regime_flag & 2) == 0)) && finite_f(Pout); // (see sgm_cost_apply_kernel)
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const unsigned pass_mask = (1u << n_pass) - 1u;
    const bool lane_on = d0 < D; // (D % 4 == 0: a lane's four disparities exist or none does)
    const WinnerOut &wo = out.w;
    const bool want_winner = wo.idx || wo.disp || wo.taps || wo.keys;
    // "the value at index 0 is NaN" makes index 0 the winner (correlation_base.h:441-455) but enters a reduction key only in the shard that
    // holds disparity 0 (winner_key): folded into the one integer reduction below when the two agree -- the host sends the other case
    // (keys of a later shard together with an index map) to the general kernel
    const bool fold_first_nan = !wo.keys || wo.key_offset == 0;
    struct Batch {
        float c[G][4], mp[G][MIN_P_PLANES];
    };
    const int64_t stride = (int64_t)nwaves * PB;
    auto load_batch = [&](Batch &B, int64_t q0) {
        const int p0 = (int)min(q0, (int64_t)npx32 - 1); // (past the end: the last pixel again, unused)
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int p = min(p0 + g * PPW + sub, npx32 - 1);
            const float4 v = *reinterpret_cast<const float4 *>(cv + (int64_t)p * D + (lane_on ? d0 : 0)); // (a lane past D: the row's first costs, unused)
            B.c[g][0] = v.x; B.c[g][1] = v.y; B.c[g][2] = v.z; B.c[g][3] = v.w;
#pragma unroll
            for (int pl = 0; pl < MIN_P_PLANES; pl++) B.mp[g][pl] = mmap[(int64_t)pl * npx + p]; // (the scratch always holds every plane)
        }
    };
    auto run_batch = [&](const Batch &B, int64_t q0) {
        if (q0 >= npx) return;
        const int p0 = (int)q0;
        const int i0 = (int)((unsigned)p0 / (unsigned)W), j0 = p0 - i0 * W;
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (p0 + g * PPW >= npx32) break; // (wave-uniform) no pixel of this group exists
            const int p = p0 + g * PPW + sub;
            const bool active = lane_on && p < npx32;
            int j = j0 + g * PPW + sub, i = i0;
            while (j >= W) { // (an image narrower than a batch wraps more than once)
                j -= W;
                i++;
            }
            const int ip = i - top, jp = j - left;
            unsigned vis = 0u; // bit q: pass q visits this lane's pixel (pass_visits(), svh_sgm_lines.h)
            if (ip >= 0 && ip < Hp && jp >= 0 && jp < Wp)
                vis = (3u | (ip >= jp ? 4u : 0u) | (jp >= ip ? 8u : 0u) | (ip + jp < Wp ? 16u : 0u) | (ip + jp < Hp ? 32u : 0u)) & pass_mask;
            float c[4], s[4];
#pragma unroll
            for (int k = 0; k < 4; k++) s[k] = c[k] = B.c[g][k]; // sgm_cv := cv, sgm.h:371-377
            // the largest |min_p| among the passes that visit the pixel (the plane of a pass that does not holds whatever the scratch held)
            float m = 0.0f;
#pragma unroll
            for (int q = 0; q < 6; q++) m = fmaxf(m, ((vis >> q) & 1u) ? fabsf(B.mp[g][min_p_plane(q)]) : 0.0f);
            const unsigned vis0 = (unsigned)__builtin_amdgcn_readfirstlane((int)vis);
            const bool odd = active && (j + D > W || vis != vis0 || !(m < INFINITY));
            if (lean && __builtin_amdgcn_ballot_w64(odd) == 0ull) {
                // every pixel of the group: inside the image for every disparity, the same visiting passes, finite min_p, the lean regime --
                // four operations per pass and cost (sgm.h:291-300 with t = c and both terms finite)
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    if (!((vis0 >> q) & 1u)) continue; // (wave-uniform)
                    const float mp = B.mp[g][min_p_plane(q)];
#pragma unroll
                    for (int k = 0; k < 4; k++) s[k] += (c[k] + (c[k] - mp)) - c[k];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    if (q >= n_pass) break;
                    const bool visits = (vis >> q) & 1u;
                    const float mp = B.mp[g][min_p_plane(q)];
                    const bool mp_fin = finite_f(mp);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float t = (j + d0 + k >= W) ? c[k] + Pout : c[k];
                        float act = c[k];
                        if (mp_fin && finite_f(t)) act = c[k] + (t - mp);
                        const float sum = s[k] + (act - c[k]); // sgm.h:298-300
                        s[k] = visits ? sum : s[k];
                    }
                }
            }
            if (out.sgm && active) store_uncounted(reinterpret_cast<float4 *>(out.sgm + (int64_t)p * D + d0), make_float4(s[0], s[1], s[2], s[3]));
            if (!want_winner) continue;
            // extractSelectedIndex (correlation_base.h:441-455) inside the pixel's lanes: the extremum over the non-NaN values, the largest
            // index that holds it, index 0 when the value there is NaN or nothing compares
            float A = INFINITY;
#pragma unroll
            for (int k = 0; k < 4; k++) A = fminf(A, s[k]);
            const float M = pixel_allreduce_min<LPP>(lane_on ? A : INFINITY);
            int kb = -1;
#pragma unroll
            for (int k = 0; k < 4; k++) kb = (s[k] == M) ? k : kb;
            constexpr int FIRST_NAN = 1 << 20;
            int key = (lane_on && kb >= 0) ? d0 + kb + 1 : 0; // 0: this lane holds no candidate
            const bool first_nan_here = dl == 0 && isnan(s[0]);
            if (fold_first_nan && first_nan_here) key = FIRST_NAN;
            const int best = pixel_allreduce_max<LPP>(key);
            const bool writer = active && (best == 0 ? dl == 0 : key == best);
            if (writer) {
                const bool first_nan = key == FIRST_NAN || (!fold_first_nan && first_nan_here);
                const int wd = (best == 0 || best == FIRST_NAN) ? -1 : best - 1; // (FIRST_NAN: unused by winner_key and by sel below)
                const int sel = (best == 0 || best == FIRST_NAN) ? 0 : best - 1;
                if (wo.idx) store_uncounted(wo.idx + p, sel);
                if (wo.disp) store_uncounted(wo.disp + p, wo.disp_sign * sel + wo.disp_offset);
                if (wo.keys) store_uncounted(wo.keys + p, winner_key<true>(Winner{M, wd}, first_nan, wo.key_offset, wo.key_total));
            }
            if (wo.taps) {
                // truncatedCostVolume<Same>, radius 1 (correlation_base.h:601-613) around the selected index, from the writer's own registers
                // and its neighbours' edge values (two DPP moves, executed by every lane)
                const float prevL = lane_shift_up(s[3], 0.0f), nextR = lane_shift_down(s[0], 0.0f);
                if (writer) {
                    const int ck = (best == 0 || best == FIRST_NAN) ? 0 : kb; // the writer's candidate (index 0: lane 0 of the pixel, k = 0)
                    const int sel = d0 + ck;
                    const float t0 = ck == 0 ? s[0] : ck == 1 ? s[1] : ck == 2 ? s[2] : s[3];
                    const float tm1 = ck == 0 ? prevL : ck == 1 ? s[0] : ck == 2 ? s[1] : s[2];
                    const float tp1 = ck == 0 ? s[1] : ck == 1 ? s[2] : ck == 2 ? s[3] : nextR;
                    const bool px_bad = j < wo.taps_h_r || i < wo.taps_v_r || i + wo.taps_v_r >= H;
                    const int limit = px_bad ? -2 : W - wo.taps_h_r - j; // valid: 0 <= pd < D and pd < limit
                    const float nan = __uint_as_float(0x7FC00000u);
                    float *o = wo.taps + (int64_t)p * 3;
                    store_uncounted(o, (sel == 0 || sel - 1 >= limit) ? nan : tm1);
                    store_uncounted(o + 1, (sel >= limit) ? nan : t0);
                    store_uncounted(o + 2, (sel + 1 >= D || sel + 1 >= limit) ? nan : tp1);
                }
            }
        }
    };
    Batch A, B;
    load_batch(A, (int64_t)wave * PB);
    for (int64_t q0 = (int64_t)wave * PB; q0 < npx; q0 += 2 * stride) {
        load_batch(B, q0 + stride);
        run_batch(A, q0);
        load_batch(A, q0 + 2 * stride);
        run_batch(B, q0 + stride);
    }
}

// Volume probe: ONE read of a float cost volume that replaces the per-pass sweeps of it.
//  * g(p) = min_d [c + (c [+ Pout])] and bit 0 of `flag`, raised as soon as one value is not an integer in [-limit, limit].  A volume
//    whose values are all small integers (a Hamming volume handed to the per-function API, say) is in the same exact regime as the
//    census path: every operation of sgm.h:257-300 is exact, the per-pass minima follow mp' = g - mp, and the scan kernels of
//    svh_census_sgm.hip produce the min_p maps from g.  (limit < 0: no exact route wanted, bit 0 is always raised.)
//  * minima(p) = (min of the finite costs with j + d < W, min of the finite costs with j + d >= W; +inf for an empty region) and bit 1
//    of `flag`, raised when a finite |c| exceeds SGM_SAFE_MAGNITUDE.  These two numbers are ALL the Cost branch's recurrence needs of
//    a pixel (sgm_cost_minmap_scalar_kernel below), for any float volume whose magnitudes cannot overflow along a line.
// The scan kernels run only while the flag is 0, the two-minima recurrences when bit 0 is up and bit 1 is not, the wave-per-line
// sweeps of the volume only when bit 1 is up: a device-side word gates all three, no host round trip.
constexpr float SGM_SAFE_MAGNITUDE = 1e30f;
// EXACT: the caller wants the exact-integer route tried (bit 0 is meaningful); otherwise bit 0 is raised unconditionally.
// g is not computed here any more: in the exact regime it is min(2 m_in, m_out + (m_out + Pout)) (gmap_from_minima_kernel, gated on
// the flag), one wave reduction less per pixel.  Most pixels have no disparity that looks past the border (j + D <= W, wave-uniform):
// one reduction instead of two.
template <int R, bool EXACT>
__global__ void __launch_bounds__(256) volume_minima_probe_kernel(SrcVolume src, int64_t npx, int D, int W, float limit, float2 *__restrict__ minima,
                                                                 int *__restrict__ flag) {
    const int lane = threadIdx.x & 63;
    // (pixel indices are wave-uniform: kept in scalar registers, one 32-bit division per eight pixels -- a 64-bit division per pixel
    // and lane cost more instructions than the whole reduction)
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int npx32 = (int)npx; // (the host refuses more than 2^31 - 1 pixels)
    bool bad = false;
    float amax = 0.0f; // largest finite |c| seen by this lane
    constexpr int PROBE_PB = 8; // pixels per wave iteration, loads issued together
    for (int p0 = wave * PROBE_PB; p0 < npx32; p0 += nwaves * PROBE_PB) {
        float cb[PROBE_PB][R];
        const int j0 = (int)((unsigned)p0 % (unsigned)W);
#pragma unroll
        for (int u = 0; u < PROBE_PB; u++) src.template load_flat<R>(min(p0 + u, npx32 - 1), lane, cb[u]);
#pragma unroll
        for (int u = 0; u < PROBE_PB; u++) {
            const int p = p0 + u;
            if (p >= npx32) break;
            int j = j0 + u;
            while (j >= W) j -= W;
            float cf[R];
#pragma unroll
            for (int k = 0; k < R; k++) {
                const float c = cb[u][k], ac = fabsf(c);
                const bool fin = ac < INFINITY && lane * R + k < D;
                if constexpr (EXACT) bad = bad || (lane * R + k < D && (!(ac <= limit) || c != rintf(c)));
                amax = fmaxf(amax, fin ? ac : 0.0f);
                cf[k] = fin ? c : INFINITY;
            }
            float m_in = INFINITY, m_out = INFINITY;
            if (j + D <= W) { // (wave-uniform) every disparity looks inside the image
#pragma unroll
                for (int k = 0; k < R; k++) m_in = fminf(m_in, cf[k]);
                m_in = wave_min(m_in);
            } else {
#pragma unroll
                for (int k = 0; k < R; k++) {
                    const bool oob = j + lane * R + k >= W;
                    m_in = fminf(m_in, oob ? INFINITY : cf[k]);
                    m_out = fminf(m_out, oob ? cf[k] : INFINITY);
                }
                m_in = wave_min(m_in);
                m_out = wave_min(m_out);
            }
            if (lane == 0) minima[p] = make_float2(m_in, m_out);
        }
    }
    const int raise = ((!EXACT || __any(bad)) ? 1 : 0) | (__any(amax > SGM_SAFE_MAGNITUDE) ? 2 : 0);
    // (on a volume of real numbers every wave raises bit 0: 32 768 atomics on one word took longer than the probe's read of the volume -- 0.41
    // of 0.46 ms at 1080p x 64; a wave that sees its bits already up has nothing to add)
    if (raise && lane == 0 && (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & raise) != raise) atomicOr(flag, raise);
}

// The same for aligned rows of at most 128 costs, a multiple of four: four costs per lane, a pixel on 16 or 32 lanes, four or two pixels per
// wave (see sgm_cost_apply_packed_kernel) -- the wave-per-pixel probe takes the same 0.29 - 0.44 ms at 1080p for 64 and 128 costs as for 256.
template <int LPP, bool EXACT>
__global__ void __launch_bounds__(256) volume_minima_probe_packed_kernel(const float *__restrict__ cv, int64_t npx, int D, int W, float limit,
                                                                        float2 *__restrict__ minima, int *__restrict__ flag) {
    constexpr int PPW = 64 / LPP, G = 4, PB = G * PPW;
    const int lane = threadIdx.x & 63, sub = lane / LPP, dl = lane % LPP, d0 = dl * 4;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int npx32 = (int)npx; // (the host refuses more than 2^31 - 1 pixels)
    const bool lane_on = d0 < D;
    bool bad = false;
    float amax = 0.0f; // largest finite |c| seen by this lane
    for (int64_t q0 = (int64_t)wave * PB; q0 < npx; q0 += (int64_t)nwaves * PB) {
        const int p0 = (int)q0;
        float4 v[G];
#pragma unroll
        for (int g = 0; g < G; g++) v[g] = *reinterpret_cast<const float4 *>(cv + (int64_t)min(p0 + g * PPW + sub, npx32 - 1) * D + (lane_on ? d0 : 0));
        const int j0 = (int)((unsigned)p0 % (unsigned)W);
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (p0 + g * PPW >= npx32) break; // (wave-uniform)
            const int p = p0 + g * PPW + sub;
            int j = j0 + g * PPW + sub;
            while (j >= W) j -= W;
            const float c4[4] = {v[g].x, v[g].y, v[g].z, v[g].w};
            float m_in = INFINITY, m_out = INFINITY;
            const bool inside = j + D <= W; // every disparity of this lane's pixel looks inside the image
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float c = c4[k], ac = fabsf(c);
                const bool fin = ac < INFINITY && lane_on;
                if constexpr (EXACT) bad = bad || (lane_on && p < npx32 && (!(ac <= limit) || c != rintf(c)));
                amax = fmaxf(amax, fin ? ac : 0.0f);
                const float cf = fin ? c : INFINITY;
                const bool oob = j + d0 + k >= W;
                m_in = fminf(m_in, oob ? INFINITY : cf);
                m_out = fminf(m_out, oob ? cf : INFINITY);
            }
            m_in = pixel_allreduce_min<LPP>(m_in);
            if (__builtin_amdgcn_ballot_w64(!inside) != 0ull) m_out = pixel_allreduce_min<LPP>(m_out); // (wave-uniform; +inf for an inside pixel either way)
            if (dl == 0 && p < npx32) minima[p] = make_float2(m_in, m_out);
        }
    }
    const int raise = ((!EXACT || __any(bad)) ? 1 : 0) | (__any(amax > SGM_SAFE_MAGNITUDE) ? 2 : 0);
    // (on a volume of real numbers every wave raises bit 0: 32 768 atomics on one word took longer than the probe's read of the volume -- 0.41
    // of 0.46 ms at 1080p x 64; a wave that sees its bits already up has nothing to add)
    if (raise && lane == 0 && (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & raise) != raise) atomicOr(flag, raise);
}

// g(p) from the probe's minima, exact route only (the scans that read it are gated the same way)
__global__ void gmap_from_minima_gated_kernel(const float2 *__restrict__ minima, int64_t npx, float Pout, float *__restrict__ gmap, const int *__restrict__ skip_if_nonzero) {
    if (*skip_if_nonzero != 0) return;
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p < npx) {
        const float2 m = minima[p];
        gmap[p] = fminf(m.x + m.x, m.y + (m.y + Pout));
    }
}

// ---- Cost branch on any float volume: the per-pass min_p maps from two numbers per pixel ----------------------------------------
// One step of sgm.h:259-295 at a pixel with previous-line minimum mp is, per disparity (cost_step_lane_min above),
//     t = c (+ Pout where j + d >= W);   act = (mp finite and t finite) ? c + (t - mp) : c;   mp' = min over the finite act.
// For fixed mp and Pout, c -> act is a composition of roundings of non-decreasing functions, hence non-decreasing in c within each of
// the two regions (j + d < W, j + d >= W); a non-finite c gives a non-finite act and never enters the minimum.  As long as nothing
// overflows -- every finite |c| <= SGM_SAFE_MAGNITUDE, Pout not finite or |Pout| <= SGM_SAFE_MAGNITUDE, lines of at most 10^5 pixels, so
// that |mp| stays below 10^36 -- finite c give finite act, and the minimum over a region is the step applied to the region's smallest
// finite cost.  mp' is therefore a function of the pixel's two regional minima alone: the six line recurrences run on an (H, W, 2) map
// (one lane per line, the reference's float operations in the reference's order, literally the per-disparity step on two values) and
// the volume is read once, by the probe, instead of once per pass.  The minima themselves are order-free (a minimum of floats is
// exact).  Outside the regime (bit 1 of the probe's flag, or a huge finite Pout) the wave-per-line kernels above run instead.
// Lane <-> line mapping: a lane owns a line, and the lanes of a wave touch CONSECUTIVE COLUMNS OF ONE ROW at every step, so that
// the loads of the minima and the stores of the map are whole cache lines: the wave walks the rows of the margin box (downwards;
// upwards for DownLeft2UpRight) and a lane is active on the rows its line crosses --
//     family 0  Up2Down            line l: column l                       every row
//     family 1  UpLeft2DownRight   line k = j - i in [-(Hp-1), Wp-1]       rows max(0, -k) .. : both start loops of the reference (passes 2
//                                  and 3 share a plane; the corner line runs twice there with the same values, once here)
//     family 2  UpRight2DownLeft   line l = i + j in [0, Wp-1]             rows 0 .. l
//     family 3  DownLeft2UpRight   line l = i + j in [0, Hp-1]             rows l .. 0, upwards
// -- while Left2Right (family 4) gives a lane a row, whose minima and map entries are contiguous in memory by themselves (eight steps:
// 64 bytes in, 32 out).  The chains of different lanes are independent, so none of this changes a value.
struct ScalarLines {
    int top, left, Hp, Wp;
    int first[6]; // lanes of family f: [first[f], first[f + 1]), each family padded to whole waves
    int n_dir;    // 4: families 0 and 4 only
};
// One step on two minima.  The kernel only runs in the regime the probe established (bit 1 of the flag down: every finite |c| <= 1e30, so
// a minimum is a finite number of that size or +inf for a region without a finite cost -- never NaN, never -inf -- and mp, a minimum of
// sums of such numbers, is finite or +inf and cannot overflow along a line of 10^5 pixels), which removes most of the per-disparity
// step's finiteness tests: an infinite minimum yields an infinite sum, which the final minimum ignores by itself.  What remains of
// sgm.h:259-295: act = (mp finite [and t finite]) ? c + (t - mp) : c, with t = c for the first region and c + Pout for the second
// (t is finite iff Pout is, for a finite c; for c = +inf either form gives +inf).
__device__ __forceinline__ float cost_step_two_minima(float2 m, float Pout, bool pout_fin, float mp, bool mp_fin) {
    const float a_in = mp_fin ? m.x + (m.x - mp) : m.x;
    const float a_out = (mp_fin && pout_fin) ? m.y + ((m.y + Pout) - mp) : m.y;
    float next; // (no NaN can reach this minimum: the plain instruction, without the canonicalising v_max pair fminf() puts on the chain)
    asm("v_min_f32 %0, %1, %2" : "=v"(next) : "v"(a_in), "v"(a_out));
    return next;
}
__global__ void __launch_bounds__(64) sgm_cost_minmap_scalar_kernel(const float2 *__restrict__ minima, ScalarLines g, int W, int64_t npx, float Pout,
                                                                    float *__restrict__ mmap, const int *__restrict__ gate) {
    {
        const int f = *gate;
        if (f == 0 || (f & 2)) return; // exact-integer scans made the maps / magnitudes outside the regime: the volume sweeps make them
    }
    // (families are padded to whole waves: the family is a function of the block alone, and written as one -- with the lane in it the
    // compiler took the buffer resources below for per-lane values and wrapped every access in a waterfall loop)
    const int block_first = blockIdx.x * 64;
    int fam = 0;
    while (fam < 4 && block_first >= g.first[fam + 1]) fam++;
    const int l = block_first - g.first[fam] + (int)threadIdx.x;
    const bool pout_fin = finite_f(Pout);
    // Addresses as 32-bit byte offsets into two buffer resources (the host keeps npx * 8 below 2^32): a step is one add per array, a
    // load past either end returns zeros and a store there is dropped, so a lane that is not on its line yet (or any more) needs no
    // branch -- its store offset is parked out of range.  The loads do not depend on the recurrence: two register batches of PF steps
    // alternate, the next batch's minima are in flight while the chain walks the current one.
    const __amdgpu_buffer_rsrc_t rs_min = __builtin_amdgcn_make_buffer_rsrc((void *)minima, 0, (int)(npx * 8), 0x00020000);
    constexpr int PF = 24;
    float mp = 0.0f; // previous_cost[] = 0 -> min over finite = 0 (sgm.h:206-208)
    bool mp_fin = true;
    if (fam == 4) { // Left2Right: lane = row; a lane's minima and map entries are contiguous in memory by themselves
        const __amdgpu_buffer_rsrc_t rs_map = __builtin_amdgcn_make_buffer_rsrc((void *)(mmap + (int64_t)min_p_plane(1) * npx), 0, (int)(npx * 4), 0x00020000);
        const bool mine = l < g.Hp;
        const uint32_t px0 = (uint32_t)((g.top + min(l, g.Hp - 1)) * W + g.left);
        // (lanes are rows here: every access of the wave touches 64 different cache lines, and the texture path takes them one at a
        // time -- with one 8-byte load and one 4-byte store per step that, not the recurrence, set the pace (170 cycles a step).  Two
        // steps per load and four per store: 16-byte accesses.)
        auto load = [&](float2 (&m)[PF], int t0) {
#pragma unroll
            for (int u = 0; u < PF; u += 2) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_min, (px0 + (uint32_t)(t0 + u)) * 8u, 0, 0); // (past the row: unused; past the map: zeros)
                m[u] = make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
                m[u + 1] = make_float2(__uint_as_float(v[2]), __uint_as_float(v[3]));
            }
        };
        auto walk = [&](const float2 (&m)[PF], int t0) {
#pragma unroll
            for (int u0 = 0; u0 < PF; u0 += 4) {
                if (t0 + u0 >= g.Wp) break; // (wave uniform)
                float out[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    out[u] = mp;
                    mp = cost_step_two_minima(m[u0 + u], Pout, pout_fin, mp, mp_fin);
                    mp_fin = mp < INFINITY;
                }
                const uint32_t off = mine ? (px0 + (uint32_t)(t0 + u0)) * 4u : 0xFFFFFFFFu;
                if (t0 + u0 + 4 <= g.Wp) { // (wave uniform)
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 ov = {__float_as_uint(out[0]), __float_as_uint(out[1]), __float_as_uint(out[2]), __float_as_uint(out[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(ov, rs_map, off, 0, 0);
                } else { // the row's last one to three pixels (the steps past them ran on values nobody uses)
                    for (int u = 0; t0 + u0 + u < g.Wp; u++) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out[u]), rs_map, mine ? off + 4u * u : off, 0, 0);
                }
            }
        };
        float2 ma[PF], mb[PF];
        load(ma, 0);
        for (int t0 = 0; t0 < g.Wp; t0 += 2 * PF) {
            load(mb, t0 + PF);
            walk(ma, t0);
            load(ma, t0 + 2 * PF);
            walk(mb, t0 + PF);
        }
        return;
    }
    // row-walking families: the wave walks rows, the lanes sit on consecutive columns of the row (see the table above)
    const int q = fam == 0 ? 0 : (fam == 1 ? 2 : (fam == 2 ? 4 : 5));
    const __amdgpu_buffer_rsrc_t rs_map = __builtin_amdgcn_make_buffer_rsrc((void *)(mmap + (int64_t)min_p_plane(q) * npx), 0, (int)(npx * 4), 0x00020000);
    const int k = fam == 1 ? l - (g.Hp - 1) : l; // family 1: k = j - i
    int r_first, r_last, lines;                  // rows of the line, in walking order r_first -> r_last
    switch (fam) {
    case 0: lines = g.Wp; r_first = 0; r_last = g.Hp - 1; break;
    case 1: lines = g.Hp + g.Wp - 1; r_first = max(0, -k); r_last = min(g.Hp - 1, g.Wp - 1 - k); break;
    case 2: lines = g.Wp; r_first = 0; r_last = min(g.Hp - 1, l); break;
    default: lines = g.Hp; r_first = l; r_last = max(0, l - (g.Wp - 1)); break;
    }
    const bool mine = l < lines;
    const int dir = fam == 3 ? -1 : 1;
    // the wave walks the union of its lanes' rows
    const int wl0 = l - (int)threadIdx.x, wl1 = min(wl0 + 63, lines - 1); // first / last line of the wave
    int w_first, w_last;
    switch (fam) {
    case 0: w_first = 0; w_last = g.Hp - 1; break;
    case 1: w_first = max(0, -(wl1 - (g.Hp - 1))); w_last = min(g.Hp - 1, g.Wp - 1 - (wl0 - (g.Hp - 1))); break;
    case 2: w_first = 0; w_last = min(g.Hp - 1, wl1); break;
    default: w_first = min(wl1, g.Hp - 1); w_last = max(0, wl0 - (g.Wp - 1)); break;
    }
    const int n_rows = (w_last - w_first) * dir + 1;
    const int col_first = fam == 0 ? l : (fam == 1 ? r_first + k : l - r_first); // the line's column on its first row
    const int dcol = fam == 0 ? 0 : (fam == 1 ? 1 : (fam == 2 ? -1 : 1));         // and how it moves per step
    const int stride = dir * W + dcol;                                             // pixel index step of the line
    const int s_begin = (r_first - w_first) * dir;                                 // the wave step at which this lane's line starts
    const int len = mine ? (r_last - r_first) * dir + 1 : 0;
    // pixel index at wave step s: px_first + (s - s_begin) * stride (two's-complement arithmetic: out of range when not on the line)
    const uint32_t px_at_0 = (uint32_t)((g.top + r_first) * W + g.left + col_first) - (uint32_t)s_begin * (uint32_t)stride;
    auto load = [&](float2 (&m)[PF], int s0) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs_min, (px_at_0 + (uint32_t)(s0 + u) * (uint32_t)stride) * 8u, 0, 0);
            m[u] = make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
        }
    };
    auto walk = [&](const float2 (&m)[PF], int s0) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int s = s0 + u;
            if (s < n_rows) { // (wave uniform)
                const unsigned rel = (unsigned)(s - s_begin);
                const bool on = rel < (unsigned)len;
                if (rel == 0u) { // the line's first pixel sees previous_cost[] = 0
                    mp = 0.0f;
                    mp_fin = true;
                }
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mp), rs_map, on ? (px_at_0 + (uint32_t)s * (uint32_t)stride) * 4u : 0xFFFFFFFFu, 0, 0);
                mp = cost_step_two_minima(m[u], Pout, pout_fin, mp, mp_fin);
                mp_fin = mp < INFINITY;
            }
        }
    };
    float2 ma[PF], mb[PF];
    load(ma, 0);
    for (int s0 = 0; s0 < n_rows; s0 += 2 * PF) {
        load(mb, s0 + PF);
        walk(ma, s0);
        load(ma, s0 + 2 * PF);
        walk(mb, s0 + PF);
    }
}

// g(p) = min_d [c + (c [+ Pout])] from the two regional minima of the pixel (an empty region holds +inf)
__global__ void gmap_from_minima_kernel(const float2 *__restrict__ minima, int64_t npx, float Pout, float *__restrict__ gmap) {
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p < npx) {
        const float2 m = minima[p];
        gmap[p] = fminf(m.x + m.x, m.y + (m.y + Pout));
    }
}

// ---- Score branch: read-modify-write sweep per pass ----------------------------------------------------
// FAR_IS_GLOBAL: P2 >= P1 >= 0.  Then fl(prev[od] - P2) <= fl(prev[od] - P1) <= prev[od] for every od (x -> fl(x - P) is
// monotone and P >= 0), so the three disparities excluded from the |od - nd| > 1 class are each dominated by a candidate
// that is present anyway, and max_{|od-nd|>1} (prev[od] - P2) may be replaced by max_p - P2 without changing a(nd).
// NEG: the pass runs on the negated costs and subtracts its contribution, which turns the max / -P recurrence into the
// textbook min / +P one bit for bit (negation is exact, max(-x) = -min(x), fl(-p - P) = -fl(p + P)); only used by the
// textbook mode: the reference's own Cost branch is the scalar recurrence of the kernels above (finding F4).
// DELTA: the pass writes its contribution act - c alone (the fused downward sweep below adds it in the reference's place).
// VEC: 16-byte aligned volumes and D = 64 R exactly: every load and store of the main loop is one unconditional vector access, so
// the compiler's count of outstanding memory operations is exact and consuming a batch does not wait for the loads of the next one
// (with the generic form's conditional accesses every step ended in s_waitcnt vmcnt(0): the two batches never overlapped).
// FIN (with VEC, FAR_IS_GLOBAL, not NEG / DELTA / FIRST): this pass is the last one to touch the pixels it visits; it writes their
// winner records (fin.records) and stores the aggregated costs only when somebody wants the volume (fin.store_all).
template <int R, int B, bool FIRST, bool FAR_IS_GLOBAL, bool NEG, bool DELTA = false, bool VEC = false, bool FIN = false, bool LEAN = false>
__global__ void __launch_bounds__(256) sgm_score_pass_kernel(const float *__restrict__ cv, float *__restrict__ sgm, LineSet ls,
                                                            int D, int W, float P1, float P2, float Pout, bool vec, ScoreFinish fin) {
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (l >= ls.n_lines) return;
    const Line L = line_of(ls, l);
    const SrcVolume src{cv, W, D, vec};
    const SrcVolume acc{sgm, W, D, vec};
    float prev[R];
#pragma unroll
    for (int k = 0; k < R; k++) prev[k] = 0.0f; // sgm.h:206-208
    // one pixel of the line: the reference's update of the R disparities of this lane
    auto step = [&](const float (&c_in)[R], const float (&sacc)[R], int ii, int jj) {
            if constexpr (VEC && FAR_IS_GLOBAL && !NEG) { // the branch-free form (svh_sgm_lines.h): a third of the instructions
                float act[R], outv[R];
                if (jj + D > W) score_step1_far_global<R, true, false, LEAN>(prev, c_in, jj, lane, D, W, P1, P2, Pout, act);
                else score_step1_far_global<R, false, false, LEAN>(prev, c_in, jj, lane, D, W, P1, P2, Pout, act);
#pragma unroll
                for (int k = 0; k < R; k++) {
                    const float base = FIRST ? c_in[k] : sacc[k];
                    outv[k] = DELTA ? act[k] - c_in[k] : base + (act[k] - c_in[k]); // :298-300
                    prev[k] = act[k];
                }
                if constexpr (FIN) wave_emit_record<false, R>(outv, lane, ii, jj, ls.Hp, W, fin.records, fin.taps_h_r, fin.taps_v_r, fin.d_valid > 0 ? fin.d_valid : 64 * R); // (whole image: ls.Hp = H)
                if (!FIN || fin.store_all) lds_put<R>(sgm + ((int64_t)ii * W + jj) * D + lane * R, outv);
                return;
            }
            float c[R];
#pragma unroll
            for (int k = 0; k < R; k++) c[k] = NEG ? -c_in[k] : c_in[k];
            // finite previous scores, -inf otherwise (isfinite filters of :224, :241)
            float pf[R];
            float A = -INFINITY, Ahead = -INFINITY, Atail = -INFINITY;
#pragma unroll
            for (int k = 0; k < R; k++) {
                pf[k] = (lane * R + k < D && finite_f(prev[k])) ? prev[k] : -INFINITY;
                A = fmaxf(A, pf[k]);
                if (k < R - 1) Ahead = fmaxf(Ahead, pf[k]);
                if (k > 0) Atail = fmaxf(Atail, pf[k]);
            }
            float max_p, PM = -INFINITY, SM = -INFINITY, far_l0 = -INFINITY, far_rl = -INFINITY;
            if (FAR_IS_GLOBAL) {
                max_p = wave_max_dpp(A); // max over finite previous scores, :220-227
            } else {
                // inclusive prefix / suffix maxima of A over the lanes
                const float pin = wave_prefix_max(A);
                float sin_ = A;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    float u = __shfl_down(sin_, off);
                    if (lane + off < 64) sin_ = fmaxf(sin_, u);
                }
                max_p = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pin), 63));
                PM = lane_shift_up(pin, -INFINITY);                       // lanes < lane
                SM = lane_shift_down(sin_, -INFINITY);                    // lanes > lane
                far_l0 = lane_shift_up(fmaxf(PM, Ahead), -INFINITY);      // disparities <= lane*R - 2
                far_rl = lane_shift_down(fmaxf(SM, Atail), -INFINITY);    // disparities >= lane*R + R + 1
            }
            const float prevL = lane_shift_up(pf[R - 1], -INFINITY);      // disparity lane*R - 1
            const float prevR = lane_shift_down(pf[0], -INFINITY);        // disparity lane*R + R
            const bool maxp_fin = finite_f(max_p);
            const float far_global = max_p - P2;
            float outv[R];
#pragma unroll
            for (int k = 0; k < R; k++) {
                float far;
                if (FAR_IS_GLOBAL) {
                    far = far_global;
                } else {
                    float fl;
                    if (k == 0) fl = far_l0;
                    else {
                        fl = PM;
#pragma unroll
                        for (int q = 0; q + 2 <= k; q++) fl = fmaxf(fl, pf[q]);
                    }
                    float fr;
                    if (k == R - 1) fr = far_rl;
                    else {
                        fr = SM;
#pragma unroll
                        for (int q = k + 2; q < R; q++) fr = fmaxf(fr, pf[q]);
                    }
                    far = fmaxf(fl, fr) - P2; // |od - nd| > 1, :239
                }
                const float lo = (k > 0 ? pf[k - 1] : prevL) - P1; // |od - nd| == 1, :238
                const float hi = (k < R - 1 ? pf[k + 1] : prevR) - P1;
                float a = fmaxf(fmaxf(pf[k], far), fmaxf(lo, hi));
                const int d = lane * R + k;
                if (jj + d >= W) a -= Pout; // :247-249
                float act = c[k];
                if (maxp_fin && finite_f(a)) act = c[k] + (a - max_p); // :251-254
                const float base = FIRST ? c_in[k] : sacc[k];
                outv[k] = DELTA ? act - c[k] : (NEG ? base - (act - c[k]) : base + (act - c[k])); // :298-300
                prev[k] = act;
            }
            float *o = sgm + ((int64_t)ii * W + jj) * D + lane * R;
            if constexpr (VEC) {
                lds_put<R>(o, outv);
                return;
            }
            if constexpr (R % 4 == 0) {
                if (vec && lane * R + R <= D) {
#pragma unroll
                    for (int q = 0; q < R / 4; q++) *reinterpret_cast<float4 *>(o + 4 * q) = make_float4(outv[4 * q], outv[4 * q + 1], outv[4 * q + 2], outv[4 * q + 3]);
                    return;
                }
            }
#pragma unroll
            for (int k = 0; k < R; k++)
                if (lane * R + k < D) o[k] = outv[k];
    };
    // two register batches: the loads of the next batch are in flight while the serial part walks the current one
    float c0[B][R], s0v[B][R], c1[B][R], s1v[B][R];
    if constexpr (VEC) {
        const int nb = L.len / B; // whole batches; the rest of the line one pixel at a time
        auto px = [&](const float *vol, int t) { return vol + ((int64_t)(L.i0 + t * L.di) * W + (L.j0 + t * L.dj)) * D + lane * R; };
        auto load_full = [&](float (&c)[B][R], float (&sv)[B][R], int k) {
            k = min(k, nb - 1); // (past the end: the last whole batch again, unused)
#pragma unroll
            for (int b = 0; b < B; b++) {
                lds_get<R>(px(cv, k * B + b), c[b]);
                if (!FIRST && !DELTA) lds_get<R>(px(sgm, k * B + b), sv[b]);
            }
        };
        auto run_full = [&](const float (&c)[B][R], const float (&sv)[B][R], int k) {
#pragma unroll
            for (int b = 0; b < B; b++) step(c[b], sv[b], L.i0 + (k * B + b) * L.di, L.j0 + (k * B + b) * L.dj);
        };
        if (nb > 0) {
            load_full(c0, s0v, 0);
            for (int k = 0; k < nb; k += 2) {
                load_full(c1, s1v, k + 1);
                run_full(c0, s0v, k);
                if (k + 1 >= nb) break;
                load_full(c0, s0v, k + 2);
                run_full(c1, s1v, k + 1);
            }
        }
        for (int t = nb * B; t < L.len; t++) {
            lds_get<R>(px(cv, t), c0[0]);
            if (!FIRST && !DELTA) lds_get<R>(px(sgm, t), s0v[0]);
            step(c0[0], s0v[0], L.i0 + t * L.di, L.j0 + t * L.dj);
        }
        return;
    }
    auto load_batch = [&](float (&c)[B][R], float (&sv)[B][R], int st) {
#pragma unroll
        for (int b = 0; b < B; b++)
            if (st + b < L.len) {
                src.template load<R>(L.i0 + (st + b) * L.di, L.j0 + (st + b) * L.dj, lane, c[b]);
                if (!FIRST && !DELTA) acc.template load<R>(L.i0 + (st + b) * L.di, L.j0 + (st + b) * L.dj, lane, sv[b]);
            }
    };
    auto run_batch = [&](const float (&c)[B][R], const float (&sv)[B][R], int st) {
#pragma unroll
        for (int b = 0; b < B; b++)
            if (st + b < L.len) step(c[b], sv[b], L.i0 + (st + b) * L.di, L.j0 + (st + b) * L.dj);
    };
    load_batch(c0, s0v, 0);
    for (int st = 0; st < L.len; st += 2 * B) {
        load_batch(c1, s1v, st + B);
        run_batch(c0, s0v, st);
        load_batch(c0, s0v, st + 2 * B);
        run_batch(c1, s1v, st + B);
    }
}


// ---- host side ------------------------------------------------------------------------------------------
// phase 1: the per-pass min_p maps (the sequential part); phase 2: rebuild S / pick the winner per pixel
template <class SRC, int R>
static int run_cost_branch(svh_context *ctx, const SgmArgs &a, const SRC &src, float *mmap, const ApplyOut *out, bool do_minmaps,
                           const int *gate = nullptr, int gate_mask = ~0, const int *regime_flag = nullptr, bool regime_vouched = false) {
    const int Hp = a.H - a.top - a.bottom, Wp = a.W - a.left - a.right;
    const int n_pass = a.n_dir >= 8 ? 6 : (a.n_dir >= 4 ? 2 : 0); // n_dir == 0: no aggregation, S = C
    constexpr int B = (R <= 4) ? 4 : (R == 8 ? 2 : 1);
    if (do_minmaps && Hp > 0 && Wp > 0 && n_pass > 0) {
        // one launch for the lines of every pass (they are independent: a pass's map depends on the volume alone)
        PassSets sets{};
        sets.n_pass = n_pass;
        for (int q = 0; q < n_pass; q++) {
            sets.ls[q] = LineSet{q, pass_lines(q, Hp, Wp), a.top, a.left, Hp, Wp};
            sets.first_block[q + 1] = sets.first_block[q] + ceil_div(sets.ls[q].n_lines, 4);
        }
        SVH_LAUNCH(ctx, "sgm_cost_minmap", (sgm_cost_minmap_kernel<SRC, R, B>), sets.first_block[n_pass], 256, 0, src, sets, a.D, a.W, (int64_t)a.H * a.W, a.Pout, mmap,
                   gate, gate_mask);
        SVH_CHECK_LAUNCH(ctx);
    }
    if (out) {
        const int64_t npx = (int64_t)a.H * a.W;
        int grid = grid_for(npx, 4, 256 * 8 * 4);
        bool piped = false;
        if constexpr (std::is_same<SRC, SrcVolume>::value && R <= 2) {
            // 128 disparities or fewer on an aligned dense volume: several pixels per wave.  (Not for reduction keys of a later shard together with
            // an index map: the first-NaN rule enters the two differently, see the kernel.)
            const bool keys_and_map = out->w.keys && out->w.key_offset != 0 && (out->w.idx || out->w.disp || out->w.taps);
            if (src.vec && (!out->sgm || out->vec_store) && a.D >= 4 && !keys_and_map) {
                piped = true;
                const int pb = a.D <= 64 ? 8 : 4; // pixels per wave and batch
                const int grid_p = grid_for(npx, 4 * pb, 256 * 8 * 4);
                if (a.D <= 64)
                    SVH_LAUNCH(ctx, "sgm_cost_apply", (sgm_cost_apply_packed_kernel<16>), grid_p, 256, 0, src.cv, a.H, a.W, a.D, a.top, a.left, Hp > 0 ? Hp : 0,
                               Wp > 0 ? Wp : 0, (Hp > 0 && Wp > 0) ? n_pass : 0, a.Pout, mmap, *out, regime_flag, regime_vouched ? 1 : 0);
                else
                    SVH_LAUNCH(ctx, "sgm_cost_apply", (sgm_cost_apply_packed_kernel<32>), grid_p, 256, 0, src.cv, a.H, a.W, a.D, a.top, a.left, Hp > 0 ? Hp : 0,
                               Wp > 0 ? Wp : 0, (Hp > 0 && Wp > 0) ? n_pass : 0, a.Pout, mmap, *out, regime_flag, regime_vouched ? 1 : 0);
            }
        }
        if constexpr (std::is_same<SRC, SrcVolume>::value && R % 4 == 0) {
            if (src.vec && (!out->sgm || out->vec_store)) {
                piped = true;
                SVH_LAUNCH(ctx, "sgm_cost_apply", (sgm_cost_apply_piped_kernel<R>), grid, 256, 0, src.cv, a.H, a.W, a.D, a.top, a.left, Hp > 0 ? Hp : 0,
                           Wp > 0 ? Wp : 0, (Hp > 0 && Wp > 0) ? n_pass : 0, a.Pout, mmap, *out, regime_flag, regime_vouched ? 1 : 0);
            }
        }
        if (!piped)
            SVH_LAUNCH(ctx, "sgm_cost_apply", (sgm_cost_apply_kernel<SRC, R>), grid, 256, 0, src, a.H, a.W, a.D, a.top, a.left, Hp > 0 ? Hp : 0,
                       Wp > 0 ? Wp : 0, (Hp > 0 && Wp > 0) ? n_pass : 0, a.Pout, mmap, *out, regime_flag, regime_vouched ? 1 : 0);
        SVH_CHECK_LAUNCH(ctx);
    }
    return SVH_OK;
}

template <class SRC>
static int dispatch_cost_branch(svh_context *ctx, const SgmArgs &a, const SRC &src, float *mmap, const ApplyOut *out, bool do_minmaps,
                                const int *gate = nullptr, int gate_mask = ~0, const int *regime_flag = nullptr, bool regime_vouched = false) {
    switch (pick_R(a.D)) {
    case 1: return run_cost_branch<SRC, 1>(ctx, a, src, mmap, out, do_minmaps, gate, gate_mask, regime_flag, regime_vouched);
    case 2: return run_cost_branch<SRC, 2>(ctx, a, src, mmap, out, do_minmaps, gate, gate_mask, regime_flag, regime_vouched);
    case 4: return run_cost_branch<SRC, 4>(ctx, a, src, mmap, out, do_minmaps, gate, gate_mask, regime_flag, regime_vouched);
    case 8: return run_cost_branch<SRC, 8>(ctx, a, src, mmap, out, do_minmaps, gate, gate_mask, regime_flag, regime_vouched);
    case 16: return run_cost_branch<SRC, 16>(ctx, a, src, mmap, out, do_minmaps, gate, gate_mask, regime_flag, regime_vouched);
    case 32: return run_cost_branch<SRC, 32>(ctx, a, src, mmap, out, do_minmaps, gate, gate_mask, regime_flag, regime_vouched); // (up to 2048 disparities: 32 per lane)
    default: return fail(ctx, SVH_ERR_UNSUPPORTED, "SGM supports at most 2048 disparities (got %d)", a.D);
    }
}

int dev_sgm_cost_branch(svh_context *ctx, Scratch &scr, const SgmArgs &a, const CostSource &cs, float *out_sgm, const WinnerOut &win) {
    if ((int64_t)a.H * a.W * a.D == 0) return SVH_OK;
    float *mmap = scr.get_n<float>((size_t)MIN_P_PLANES * a.H * a.W);
    if (!mmap) return SVH_ERR_OUT_OF_MEMORY;
    ApplyOut out{out_sgm, win, out_sgm && aligned16(out_sgm) && a.D % 4 == 0};
    if (cs.cv) {
        SrcVolume src{cs.cv, a.W, a.D, aligned16(cs.cv) && a.D % 4 == 0};
        const int Hp = a.H - a.top - a.bottom, Wp = a.W - a.left - a.right;
        const int n_pass = a.n_dir >= 8 ? 6 : (a.n_dir >= 4 ? 2 : 0);
        // largest |c| for which 8 (2 |c| + |Pout|) (L + 2) stays below 2^24 (same bound as census_exact_regime)
        const double L = (double)std::max(a.H, a.W);
        const double limit = std::floor((16777216.0 / (8.0 * (L + 2.0)) - std::fabs((double)a.Pout)) / 2.0) - 1.0;
        const bool try_exact = ctx->census_fast_path && (int64_t)a.H * a.W < (1ll << 31) && n_pass > 0 && Hp > 0 && Wp > 0 && std::isfinite(a.Pout) &&
                               a.Pout == std::nearbyint(a.Pout) && limit >= 1.0 && a.D <= 1024;
        // the two-minima recurrences (sgm_cost_minmap_scalar_kernel) need magnitudes that cannot overflow along a line
        const bool two_minima = ctx->sgm_cost_two_minima && (int64_t)a.H * a.W < (1ll << 28) && n_pass > 0 && Hp > 0 && Wp > 0 && (!std::isfinite(a.Pout) || std::fabs(a.Pout) <= SGM_SAFE_MAGNITUDE) &&
                                std::max(a.H, a.W) <= 100000 && a.D <= 1024;
        if (!try_exact && !two_minima) return dispatch_cost_branch(ctx, a, src, mmap, &out, true);
        const int64_t npx = (int64_t)a.H * a.W;
        if (try_exact && cs.minima && cs.max_abs <= (float)limit) {
            // the caller vouches for what the probe would find (svh_sgm_cost_volume_minima): g from the two regional minima, no read of C
            float *gmap = scr.get_n<float>((size_t)npx);
            if (!gmap) return SVH_ERR_OUT_OF_MEMORY;
            SVH_LAUNCH(ctx, "gmap_from_minima", gmap_from_minima_kernel, grid_for(npx, 256), 256, 0, reinterpret_cast<const float2 *>(cs.minima), npx, a.Pout, gmap);
            SVH_CHECK_LAUNCH(ctx);
            SVH_TRY(dev_census_scans(ctx, a, nullptr, gmap, true, mmap, nullptr));
            return dispatch_cost_branch(ctx, a, src, mmap, &out, false, nullptr, ~0, nullptr, true); // (small integers: inside the regime by the caller's statement)
        }
        if (two_minima && cs.float_minima && cs.float_flag) {
            // the kernel that wrote the volume left its regional minima (CostReduce mode 2): no read of C before the apply pass
            ScalarLines g{};
            g.top = a.top, g.left = a.left, g.Hp = Hp, g.Wp = Wp, g.n_dir = a.n_dir;
            const int fam_lines[5] = {Wp, n_pass > 2 ? Hp + Wp - 1 : 0, n_pass > 2 ? Wp : 0, n_pass > 2 ? Hp : 0, Hp};
            for (int f = 0; f < 5; f++) g.first[f + 1] = g.first[f] + ceil_div(fam_lines[f], 64) * 64;
            SVH_LAUNCH(ctx, "sgm_cost_minmap_scalar", sgm_cost_minmap_scalar_kernel, g.first[5] / 64, 64, 0, reinterpret_cast<const float2 *>(cs.float_minima), g, a.W, npx,
                       a.Pout, mmap, cs.float_flag);
            SVH_CHECK_LAUNCH(ctx);
            return dispatch_cost_branch(ctx, a, src, mmap, &out, true, cs.float_flag, 2, cs.float_flag);
        }
        float *gmap = try_exact ? scr.get_n<float>((size_t)npx) : nullptr;
        float2 *minima = scr.get_n<float2>((size_t)npx);
        int *flag = scr.get_n<int>(64);
        if ((try_exact && !gmap) || !minima || !flag) return SVH_ERR_OUT_OF_MEMORY;
        SVH_HIP_CHECK(ctx, hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
        const int grid = grid_for(npx, 8 * 4, 256 * 8 * 4);
#define SVH_PROBE(RV)                                                                                                                                   \
    do {                                                                                                                                                \
        if (try_exact) SVH_LAUNCH(ctx, "sgm_volume_probe", (volume_minima_probe_kernel<RV, true>), grid, 256, 0, src, npx, a.D, a.W, (float)limit, minima, flag); \
        else SVH_LAUNCH(ctx, "sgm_volume_probe", (volume_minima_probe_kernel<RV, false>), grid, 256, 0, src, npx, a.D, a.W, 0.0f, minima, flag);             \
    } while (0)
        if (src.vec && a.D >= 4 && a.D <= 128) { // several pixels per wave
            const int gridp = grid_for(npx, 4 * 4 * (a.D <= 64 ? 4 : 2), 256 * 8 * 4);
#define SVH_PROBE_P(L)                                                                                                                                  \
    do {                                                                                                                                                \
        if (try_exact) SVH_LAUNCH(ctx, "sgm_volume_probe", (volume_minima_probe_packed_kernel<L, true>), gridp, 256, 0, src.cv, npx, a.D, a.W, (float)limit, minima, flag); \
        else SVH_LAUNCH(ctx, "sgm_volume_probe", (volume_minima_probe_packed_kernel<L, false>), gridp, 256, 0, src.cv, npx, a.D, a.W, 0.0f, minima, flag);             \
    } while (0)
            if (a.D <= 64) SVH_PROBE_P(16);
            else SVH_PROBE_P(32);
#undef SVH_PROBE_P
        } else
        switch (pick_R(a.D)) {
        case 1: SVH_PROBE(1); break;
        case 2: SVH_PROBE(2); break;
        case 4: SVH_PROBE(4); break;
        case 8: SVH_PROBE(8); break;
        default: SVH_PROBE(16); break;
        }
#undef SVH_PROBE
        SVH_CHECK_LAUNCH(ctx);
        if (try_exact) {
            SVH_LAUNCH(ctx, "gmap_from_probe", gmap_from_minima_gated_kernel, grid_for(npx, 256), 256, 0, minima, npx, a.Pout, gmap, flag);
            SVH_CHECK_LAUNCH(ctx);
            SVH_TRY(dev_census_scans(ctx, a, nullptr, gmap, true, mmap, flag)); // run only while the flag is 0
        }
        if (two_minima) {
            ScalarLines g{};
            g.top = a.top, g.left = a.left, g.Hp = Hp, g.Wp = Wp, g.n_dir = a.n_dir;
            const int fam_lines[5] = {Wp, n_pass > 2 ? Hp + Wp - 1 : 0, n_pass > 2 ? Wp : 0, n_pass > 2 ? Hp : 0, Hp};
            for (int f = 0; f < 5; f++) g.first[f + 1] = g.first[f] + ceil_div(fam_lines[f], 64) * 64;
            SVH_LAUNCH(ctx, "sgm_cost_minmap_scalar", sgm_cost_minmap_scalar_kernel, g.first[5] / 64, 64, 0, minima, g, a.W, npx, a.Pout, mmap, flag);
            SVH_CHECK_LAUNCH(ctx);
            return dispatch_cost_branch(ctx, a, src, mmap, &out, true, flag, 2, flag); // the sweeps of the volume only when the probe saw magnitudes outside the regime
        }
        return dispatch_cost_branch(ctx, a, src, mmap, &out, true, flag);   // line kernels run only if it was raised
    }
    SrcCensus src{cs.src_words, cs.tgt_words, cs.nWw, a.W, cs.Wt, cs.sign, cs.disp_lower, a.D};
    // census specialisation (svh_census_sgm.hip): pixel-per-lane kernels.  In the integer-exact regime one sweep
    // yields the regional winner keys and g, parallel line scans turn g into the min_p maps, and a per-pixel kernel
    // finishes; otherwise the wave-per-line kernels make the min_p maps and the winner is evaluated in float.
    const bool lanes = ctx->census_fast_path && census_lane_kernels_available(cs.nWw, a.D);
    const bool exact = lanes && census_exact_regime(a, cs.nWw);
    const bool lane_winner = lanes && !out_sgm;
    // index / disparity maps alone: the winner does not depend on the min_p maps (census_finalize_kernel), so no g map and no scans
    // (refinement taps that are only ever subtracted from one another do not depend on them either: WinnerOut::taps_up_to_shift)
    const bool winner_only = exact && lane_winner && ctx->census_winner_shortcut && !win.keys && (!win.taps || win.taps_up_to_shift);
    uint2 *keys = nullptr;
    if (winner_only && !win.taps) return win.any() ? dev_census_winner(ctx, scr, a, cs, win) : SVH_OK;
    if (winner_only) { // taps: keys from the sweep, then the per-pixel kernel without maps
        keys = scr.get_n<uint2>((size_t)a.H * a.W);
        if (!keys) return SVH_ERR_OUT_OF_MEMORY;
        SVH_TRY(dev_census_sweep(ctx, a, cs, keys, nullptr));
        return dev_census_finalize(ctx, a, cs, nullptr, keys, win);
    }
    if (exact && lane_winner && win.any() && census_tiles_apply(ctx, a)) return dev_census_sweep_tiles(ctx, scr, a, cs, win); // no min_p maps at all
    if (exact) SVH_TRY(dev_census_sweep_and_scans(ctx, scr, a, cs, mmap, &keys));
    if (!exact || !lane_winner) SVH_TRY(dispatch_cost_branch(ctx, a, src, mmap, lane_winner ? nullptr : &out, !exact));
    if (lane_winner && win.any()) {
        if (exact) SVH_TRY(dev_census_finalize(ctx, a, cs, mmap, keys, win));
        else SVH_TRY(dev_census_apply_select(ctx, a, cs, mmap, win));
    }
    return SVH_OK;
}

// winner records from a finished aggregated volume (rows of 64 R floats): what the launch-per-pass form leaves to do when no pass is the last
// writer of every pixel (eight directions) -- one read of S instead of extract_index + truncatedCostVolume (+ the copy out of padded rows)
template <int R>
__global__ void __launch_bounds__(256) score_records_kernel(const float *__restrict__ sgm, int H, int W, ScoreFinish fin) {
    const int lane = threadIdx.x & 63;
    const int64_t npx = (int64_t)H * W;
    for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < npx; p += (int64_t)gridDim.x * 4) {
        float s[R];
        lds_get<R>(sgm + p * (64 * R) + lane * R, s);
        const int i = (int)(p / W), j = (int)(p - (int64_t)i * W);
        wave_emit_record<false, R>(s, lane, i, j, H, W, fin.records, fin.taps_h_r, fin.taps_v_r, fin.d_valid > 0 ? fin.d_valid : 64 * R);
    }
}

template <int R>
static int run_score_branch(svh_context *ctx, Scratch &scr, const SgmArgs &a, const float *cv, float *sgm, bool textbook = false, ScoreFinish *finish = nullptr) {
    const int Hp = a.H - a.top - a.bottom, Wp = a.W - a.left - a.right;
    const int n_pass = textbook ? (a.n_dir >= 8 ? 8 : 4) : (a.n_dir >= 8 ? 6 : 2);
    const int pass0 = textbook ? 6 : 0;
    const bool neg = textbook && a.strategy == SVH_COST;
    constexpr int B = (R <= 4) ? 4 : (R <= 8 ? 2 : 1);
    const bool vec = aligned16(cv) && aligned16(sgm) && a.D % 4 == 0;
    const bool whole = a.top == 0 && a.left == 0 && a.bottom == 0 && a.right == 0;
    const bool far_global = a.P2 >= a.P1 && a.P1 >= 0.0f; // also false for NaN penalties
    if (!whole || Hp <= 0 || Wp <= 0) {
        // pixels outside the margin box keep sgm = cv (sgm.h:371-377)
        ProfScope prof(ctx, "sgm_copy");
        SVH_HIP_CHECK(ctx, hipMemcpyAsync(sgm, cv, (size_t)a.H * a.W * a.D * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (Hp <= 0 || Wp <= 0) return SVH_OK;
    // The banded sweep moves 23 instead of 44 bytes per voxel but walks the rows one dependent step after the other (1.8 us + 10.5 ns per
    // disparity and row, whatever the width, until the strips outnumber the CUs); a launch per pass keeps every line of a pass in flight and
    // runs at what its bytes take (11.5 ns per thousand voxels) above a floor of 0.45 ms.  Images of up to about a megapixel at moderate
    // ranges are faster pass by pass -- the reference's own SGM benchmark rows (480x640 x 160: 1.49 -> 0.97 ms), 720p x 128 1.96 -> 1.50,
    // 1080p x 64 2.68 -> 2.05 -- larger volumes in bands (1080p x 256 4.9 against 6.4, C4).  The model picks (measured on 26 shapes; ties
    // go to the bands, which also carry the winner records).  Option "sgm_score_fused" 3 forces the bands, 0 the passes.
    bool bands = ctx->sgm_score_fused && !textbook && n_pass == 6 && whole && far_global;
    if (bands && ctx->sgm_score_fused == 1 && a.D >= 64 && (int64_t)a.H * a.W >= 20000) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        const int strip = ceil_div(a.W, 16) * 4 < cus * 3 ? 8 : 16, rounds = ceil_div(ceil_div(a.W, strip), cus);
        const double t_bands = (double)a.H * (1.8 + 0.0105 * a.D) * 1e-3 * rounds, t_passes = (double)a.H * a.W * a.D * 11.5e-9 + 0.45;
        if (t_passes < 0.95 * t_bands) bands = false;
    }
    if (bands) { // svh_sgm_sweep.hip: the four downward passes in one sweep
        bool ran = false;
        SVH_TRY(dev_sgm_score_sweep(ctx, scr, a, cv, sgm, vec, ctx->sgm_score_fused, &ran, finish));
        if (ran) return SVH_OK;
    }
    for (int q = 0; q < n_pass; q++) {
        LineSet ls{pass0 + q, pass_lines(pass0 + q, Hp, Wp), a.top, a.left, Hp, Wp};
        int grid = ceil_div(ls.n_lines, 4);
        const bool first = q == 0 && whole;
#define SVH_SCORE(FIRSTV, FARV, NEGV)                                                                                                    \
    SVH_LAUNCH(ctx, NEGV ? "sgm_textbook_pass" : "sgm_score_pass", (sgm_score_pass_kernel<R, B, FIRSTV, FARV, NEGV>), grid, 256, 0, cv, sgm, ls, a.D, a.W, \
               a.P1, a.P2, a.Pout, vec, ScoreFinish{})
        if (neg) {
            if (far_global) {
                if (first) SVH_SCORE(true, true, true); else SVH_SCORE(false, true, true);
            } else {
                if (first) SVH_SCORE(true, false, true); else SVH_SCORE(false, false, true);
            }
        } else if (far_global && vec && a.D == 64 * R && !textbook && n_pass == 2 && q == 1 && whole && finish && finish->records &&
                   (int64_t)a.H * a.W < (1ll << 29)) {
            // four directions: Left2Right is the last pass and visits every pixel of a whole-image aggregation -- the last writer of every
            // pixel: it emits the winner records (and stores the volume only when somebody wants it)
            const bool lean = a.costs_all_finite && std::isfinite(a.Pout);
            if (lean)
                SVH_LAUNCH(ctx, "sgm_score_pass", (sgm_score_pass_kernel<R, B, false, true, false, false, true, true, true>), grid, 256, 0, cv, sgm, ls, a.D, a.W, a.P1,
                           a.P2, a.Pout, vec, *finish);
            else
                SVH_LAUNCH(ctx, "sgm_score_pass", (sgm_score_pass_kernel<R, B, false, true, false, false, true, true, false>), grid, 256, 0, cv, sgm, ls, a.D, a.W, a.P1,
                           a.P2, a.Pout, vec, *finish);
            finish->done = true;
        } else if (far_global && vec && a.D == 64 * R) { // the usual case: exact prefetch form
            if (first)
                SVH_LAUNCH(ctx, "sgm_score_pass", (sgm_score_pass_kernel<R, B, true, true, false, false, true>), grid, 256, 0, cv, sgm, ls, a.D, a.W, a.P1, a.P2,
                           a.Pout, vec, ScoreFinish{});
            else
                SVH_LAUNCH(ctx, "sgm_score_pass", (sgm_score_pass_kernel<R, B, false, true, false, false, true>), grid, 256, 0, cv, sgm, ls, a.D, a.W, a.P1, a.P2,
                           a.Pout, vec, ScoreFinish{});
        } else if (far_global) {
            if (first) SVH_SCORE(true, true, false); else SVH_SCORE(false, true, false);
        } else {
            if (first) SVH_SCORE(true, false, false); else SVH_SCORE(false, false, false);
        }
#undef SVH_SCORE
        SVH_CHECK_LAUNCH(ctx);
    }
    if constexpr (R <= 8) {
        if (finish && finish->records && !finish->done && !textbook && whole && vec && a.D == 64 * R && (int64_t)a.H * a.W < (1ll << 29)) {
            SVH_LAUNCH(ctx, "sgm_score_records", score_records_kernel<R>, grid_for((int64_t)a.H * a.W, 4, 65536), 256, 0, sgm, a.H, a.W, *finish);
            SVH_CHECK_LAUNCH(ctx);
            finish->done = true;
        }
    }
    return SVH_OK;
}

// one pass of the whole-image 8-direction aggregation with P2 >= P1 >= 0 (what the fused sweeps leave to the line kernels):
// delta: write the pass's contribution instead of adding it to sgm
template <int R> static int score_line_pass(svh_context *ctx, const SgmArgs &a, const float *cv, float *sgm, int pass, bool delta, const ScoreFinish *finish) {
    // pixels per register batch (two batches: one in flight, one being walked).  Twice and four times as many measured the same at C4
    // (Left2Right + DownLeft2UpRight 5.57 / 5.57 / 5.60 ms): the passes are not waiting for their loads
    constexpr int B = (R <= 4) ? 4 : (R <= 8 ? 2 : 1);
    const bool vec = aligned16(cv) && aligned16(sgm) && a.D % 4 == 0;
    LineSet ls{pass, pass_lines(pass, a.H, a.W), 0, 0, a.H, a.W};
    if (ls.n_lines <= 0) return SVH_OK;
    const int grid = ceil_div(ls.n_lines, 4);
    const ScoreFinish none{};
#define SVH_LINE_L(DELTAV, VECV, FINV, LEANV) \
    SVH_LAUNCH(ctx, "sgm_score_pass", (sgm_score_pass_kernel<R, B, false, true, false, DELTAV, VECV, FINV, LEANV>), grid, 256, 0, cv, sgm, ls, a.D, a.W, a.P1, a.P2, a.Pout, \
               vec, FINV ? *finish : none)
#define SVH_LINE(DELTAV, VECV, FINV) SVH_LINE_L(DELTAV, VECV, FINV, false)
    const bool lean = a.costs_all_finite && std::isfinite(a.Pout);
    if (vec && a.D == 64 * R) {
        if (finish && !delta) {
            if (lean) SVH_LINE_L(false, true, true, true); else SVH_LINE(false, true, true);
        } else if (delta) {
            if (lean) SVH_LINE_L(true, true, false, true); else SVH_LINE(true, true, false);
        } else SVH_LINE(false, true, false);
    } else {
        if (finish) return fail(ctx, SVH_ERR_HIP, "internal: fused winner asked of the generic line pass");
        if (delta) SVH_LINE(true, false, false); else SVH_LINE(false, false, false);
    }
#undef SVH_LINE
#undef SVH_LINE_L
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

int dev_sgm_score_line_pass(svh_context *ctx, const SgmArgs &a, const float *cv, float *sgm, int pass, bool delta, const ScoreFinish *finish) {
    switch (pick_R_score(a.D)) {
    case 1: return score_line_pass<1>(ctx, a, cv, sgm, pass, delta, finish);
    case 2: return score_line_pass<2>(ctx, a, cv, sgm, pass, delta, finish);
    case 3: return score_line_pass<3>(ctx, a, cv, sgm, pass, delta, finish);
    case 4: return score_line_pass<4>(ctx, a, cv, sgm, pass, delta, finish);
    case 5: return score_line_pass<5>(ctx, a, cv, sgm, pass, delta, finish);
    case 6: return score_line_pass<6>(ctx, a, cv, sgm, pass, delta, finish);
    case 7: return score_line_pass<7>(ctx, a, cv, sgm, pass, delta, finish);
    case 8: return score_line_pass<8>(ctx, a, cv, sgm, pass, delta, finish);
    case 16: return score_line_pass<16>(ctx, a, cv, sgm, pass, delta, finish);
    default: return score_line_pass<32>(ctx, a, cv, sgm, pass, delta, finish);
    }
}

// rows of `count` floats at a pitch of `in_pitch` -> rows at a pitch of `out_pitch` floats, what lies past `count` filled with `fill`.  A block
// takes REPITCH_ROWS rows as one flat run of output elements: every store instruction of a wave is 256 contiguous bytes, every thread has
// a dozen independent loads in flight (a wave per row had three, and ran at 0.7 TB/s).
constexpr int REPITCH_ROWS = 16;
__global__ void __launch_bounds__(256) repitch_rows_kernel(const float *__restrict__ in, int64_t n_rows, int in_pitch, int count, float *__restrict__ out,
                                                          int out_pitch, float fill) {
    const int64_t r0 = (int64_t)blockIdx.x * REPITCH_ROWS;
    const int rows = (int)(n_rows - r0 < REPITCH_ROWS ? n_rows - r0 : REPITCH_ROWS), n = rows * out_pitch;
    const float *src = in + r0 * in_pitch;
    float *dst = out + r0 * out_pitch;
#pragma unroll 4
    for (int e = threadIdx.x; e < n; e += 256) {
        const int r = e / out_pitch, d = e - r * out_pitch;
        dst[e] = d < count ? src[r * in_pitch + d] : fill;
    }
}

__global__ void __launch_bounds__(256) fill_pads_kernel(float *__restrict__ cv, int64_t n_rows, int D, int pitch, float fill) {
    const int pads = pitch - D;
    const int64_t r0 = (int64_t)blockIdx.x * REPITCH_ROWS;
    const int rows = (int)(n_rows - r0 < REPITCH_ROWS ? n_rows - r0 : REPITCH_ROWS);
    for (int e = threadIdx.x; e < rows * pads; e += 256) {
        const int r = e / pads;
        cv[(r0 + r) * pitch + D + (e - r * pads)] = fill;
    }
}

int dev_sgm_fill_pads(svh_context *ctx, float *cv, int64_t n_rows, int D, int pitch) {
    if (n_rows <= 0 || pitch <= D) return SVH_OK;
    SVH_LAUNCH(ctx, "sgm_pad_rows", fill_pads_kernel, (int)((n_rows + REPITCH_ROWS - 1) / REPITCH_ROWS), 256, 0, cv, n_rows, D, pitch, -INFINITY);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

static int score_branch_dispatch(svh_context *ctx, Scratch &scr, const SgmArgs &a, const float *cv, float *out_sgm, bool textbook, ScoreFinish *finish);

int dev_sgm_score_branch(svh_context *ctx, Scratch &scr, const SgmArgs &a, const float *cv, float *out_sgm, bool textbook, ScoreFinish *finish) {
    if (finish) finish->done = false;
    if ((int64_t)a.H * a.W * a.D == 0) return SVH_OK;
    // 65 .. 511 disparities, no multiple of 64 (the reference's own benchmark rows use 160): the vector kernels -- unconditional 16-byte
    // accesses, the banded sweep, the winner records -- need rows of whole lanes.  Aggregate a copy padded to the next multiple of 64 with
    // -inf instead: a pad never enters a maximum (the isfinite filters of sgm.h:224, :241; v_max ignores it in the filter-free forms), its
    // own value stays cost + (a - max_p) = -inf along every line, and at a line's first pixel, where every previous score is 0, it only
    // repeats values the real neighbours already contribute.  Two more streaming passes (pad in, copy out) for a 1.5 x faster aggregation.
    const int DP = (a.D + 63) / 64 * 64;
    if (!textbook && ctx->sgm_score_pad && a.D > 64 && a.D < 512 && DP != a.D) {
        const int64_t npx = (int64_t)a.H * a.W;
        const bool prepadded = a.cv_pitch == DP; // (the caller's cost kernel wrote the padded layout and filled the pads)
        float *cvp = prepadded ? nullptr : scr.get_n<float>((size_t)npx * DP), *sgp = scr.get_n<float>((size_t)npx * DP);
        if ((!prepadded && !cvp) || !sgp) return SVH_ERR_OUT_OF_MEMORY;
        const int grid = (int)((npx + REPITCH_ROWS - 1) / REPITCH_ROWS);
        if (!prepadded) {
            SVH_LAUNCH(ctx, "sgm_pad_rows", repitch_rows_kernel, grid, 256, 0, cv, npx, a.D, a.D, cvp, DP, -INFINITY);
            SVH_CHECK_LAUNCH(ctx);
        }
        SgmArgs ap = a;
        ap.D = DP;
        ap.cv_pitch = 0;
        if (finish) finish->d_valid = a.D;
        SVH_TRY(score_branch_dispatch(ctx, scr, ap, prepadded ? cv : cvp, sgp, false, finish));
        if (!finish || finish->store_all || !finish->done) { // somebody reads the aggregated volume
            SVH_LAUNCH(ctx, "sgm_pad_rows", repitch_rows_kernel, grid, 256, 0, sgp, npx, DP, a.D, out_sgm, a.D, 0.0f);
            SVH_CHECK_LAUNCH(ctx);
        }
        return SVH_OK;
    }
    return score_branch_dispatch(ctx, scr, a, cv, out_sgm, textbook, finish);
}

static int score_branch_dispatch(svh_context *ctx, Scratch &scr, const SgmArgs &a, const float *cv, float *out_sgm, bool textbook, ScoreFinish *finish) {
    switch (pick_R_score(a.D)) {
    case 1: return run_score_branch<1>(ctx, scr, a, cv, out_sgm, textbook, finish);
    case 2: return run_score_branch<2>(ctx, scr, a, cv, out_sgm, textbook, finish);
    case 3: return run_score_branch<3>(ctx, scr, a, cv, out_sgm, textbook, finish);
    case 4: return run_score_branch<4>(ctx, scr, a, cv, out_sgm, textbook, finish);
    case 5: return run_score_branch<5>(ctx, scr, a, cv, out_sgm, textbook, finish);
    case 6: return run_score_branch<6>(ctx, scr, a, cv, out_sgm, textbook, finish);
    case 7: return run_score_branch<7>(ctx, scr, a, cv, out_sgm, textbook, finish);
    case 8: return run_score_branch<8>(ctx, scr, a, cv, out_sgm, textbook, finish);
    case 16: return run_score_branch<16>(ctx, scr, a, cv, out_sgm, textbook, finish);
    case 32: return run_score_branch<32>(ctx, scr, a, cv, out_sgm, textbook, finish); // (up to 2048 disparities: 32 per lane)
    default: return fail(ctx, SVH_ERR_UNSUPPORTED, "SGM supports at most 2048 disparities (got %d)", a.D);
    }
}

} // namespace svh

using namespace svh;

static int sgm_cost_volume_impl(svh_context *ctx, int n_directions, int strategy, const svh_array *cv, const svh_array *minima, float max_abs, float P1, float P2,
                                const int32_t margins[4], float Pout, svh_array *out, svh_array *winner_idx = nullptr, int *winner_written = nullptr,
                                int minima_kind = 1) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    if (winner_written) *winner_written = 0;
    if (winner_idx) {
        SVH_TRY(validate(ctx, winner_idx, "winner_idx", SVH_I32, 2, 2));
        if (!winner_written) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "winner_idx needs winner_written");
        if (winner_idx->shape[0] != cv->shape[0] || winner_idx->shape[1] != cv->shape[1]) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "winner_idx must have shape (H,W)");
    }
    SVH_TRY(validate_volume(ctx, cv, "cv")); // (T_CV: float or an integer type the reference casts to float as it reads, sgm.h:234, :273, :299)
    SVH_TRY(validate(ctx, out, "out", SVH_F32, 3, 3));
    if (minima && cv->dtype != SVH_F32) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "the minima statement goes with a float32 volume");
    if (n_directions == 16)
        return fail(ctx, SVH_ERR_UNSUPPORTED,
                    "16 directions: the reference's overlapping lines race on sgm_cv (sgm.h:299, :336), results are not defined");
    if (n_directions != 4 && n_directions != 8) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "n_directions must be 4 or 8");
    if (strategy != SVH_COST && strategy != SVH_SCORE) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad strategy");
    for (int k = 0; k < 3; k++)
        if (cv->shape[k] != out->shape[k]) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "out must have the shape of cv");
    static const int32_t zero[4] = {0, 0, 0, 0};
    const int32_t *m = margins ? margins : zero;
    for (int k = 0; k < 4; k++)
        if (m[k] < 0) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "margins must be non-negative");
    SgmArgs a{n_directions, strategy, (int)cv->shape[0], (int)cv->shape[1], (int)cv->shape[2], P1, P2, Pout, m[0], m[1], m[2], m[3]};
    Scratch scr(ctx);
    void *dcv;
    OutStage os;
    SVH_TRY(stage_volume_as_float(ctx, scr, *cv, &dcv));
    SVH_TRY(stage_out(ctx, scr, *out, &os));
    OutStage ow;
    if (winner_idx) SVH_TRY(stage_out(ctx, scr, *winner_idx, &ow));
    if (strategy == SVH_COST) {
        CostSource cs;
        cs.cv = (const float *)dcv;
        if (minima) {
            SVH_TRY(validate(ctx, minima, "minima", SVH_F32, 3, 3));
            if (minima->shape[0] != cv->shape[0] || minima->shape[1] != cv->shape[1] || minima->shape[2] != 2)
                return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "minima must have shape (H,W,2)");
            if (!(max_abs >= 0.0f)) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "max_abs must be a non-negative number");
            if (minima_kind != 1 && minima_kind != 2) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "minima_kind: 1 (integer costs) or 2 (float costs inside the regime)");
            void *dmin;
            SVH_TRY(stage_in(ctx, scr, *minima, &dmin));
            if (minima_kind == 1) {
                cs.minima = (const float *)dmin;
                cs.max_abs = max_abs;
            } else {
                // float costs the caller knows to lie inside the regime (svh_unfold_cost_volume_minima said 2): no probing read, the line
                // recurrences run on the minima (bit 0 of the flag word: no exact-integer route; bit 1 stays down)
                if (!(max_abs <= SGM_SAFE_MAGNITUDE)) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "minima_kind 2 states finite magnitudes up to 1e30");
                int *flag = scr.get_n<int>(64);
                if (!flag) return SVH_ERR_OUT_OF_MEMORY;
                SVH_HIP_CHECK(ctx, hipMemsetD32Async((hipDeviceptr_t)flag, 1, 1, ctx->stream)); // (a device-side fill, not a copy from the stack)
                cs.float_minima = (const float *)dmin;
                cs.float_flag = flag;
            }
        }
        // the kernel that writes S holds a pixel's aggregated costs in one wave: its winner is a by-product (extractSelectedIndex's rule)
        WinnerOut w;
        if (winner_idx) w.idx = (int32_t *)ow.dptr;
        SVH_TRY(dev_sgm_cost_branch(ctx, scr, a, cs, (float *)os.dptr, w));
        if (winner_idx) *winner_written = 1;
    } else {
        if (os.dptr == dcv) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "cv and out must not alias");
        // The winner of the Score branch: records left by the last writer of every pixel (the banded sweep) or a scan of the volume this call
        // has just written.  With the whole volume stored either way, the records only save the scan's re-read (4 B/voxel) and cost the
        // sweep and the last pass 8 - 12 % each, whatever D is: measured at 1080p, records / scan: D = 64 2.56 / 2.22 ms, D = 128 3.10 /
        // 2.86, D = 256 4.50 / 4.46 -- records from 256 disparities on, the scan below that and wherever the banded sweep does not run.
        ScoreFinish fin;
        const bool want_records = winner_idx && a.D >= 256;
        if (want_records) {
            fin.records = scr.get_n<float>((size_t)a.H * a.W * 4);
            if (!fin.records) return SVH_ERR_OUT_OF_MEMORY;
            fin.store_all = true;
        }
        SVH_TRY(dev_sgm_score_branch(ctx, scr, a, (const float *)dcv, (float *)os.dptr, false, want_records ? &fin : nullptr));
        if (winner_idx) {
            if (want_records && fin.done) SVH_TRY(dev_finish_records(ctx, fin.records, (int64_t)a.H * a.W, -1, 1, 0, (int32_t *)ow.dptr, nullptr, nullptr));
            else SVH_TRY(dev_extract_index(ctx, SVH_SCORE, (const float *)os.dptr, (int64_t)a.H * a.W, a.D, (int32_t *)ow.dptr, nullptr, 0, 0));
            *winner_written = 1;
        }
    }
    if (winner_idx) SVH_TRY(finish_out(ctx, ow));
    return finish_out(ctx, os);
}

extern "C" int svh_sgm_cost_volume(svh_context *ctx, int n_directions, int strategy, const svh_array *cv, float P1, float P2,
                                   const int32_t margins[4], float Pout, svh_array *out) {
    return sgm_cost_volume_impl(ctx, n_directions, strategy, cv, nullptr, 0.0f, P1, P2, margins, Pout, out);
}

extern "C" int svh_sgm_cost_volume_minima(svh_context *ctx, int n_directions, int strategy, const svh_array *cv, const svh_array *minima, float max_abs,
                                          float P1, float P2, const int32_t margins[4], float Pout, svh_array *out) {
    return sgm_cost_volume_impl(ctx, n_directions, strategy, cv, minima, max_abs, P1, P2, margins, Pout, out);
}

extern "C" int svh_sgm_cost_volume_winner(svh_context *ctx, int n_directions, int strategy, const svh_array *cv, const svh_array *minima, int minima_kind,
                                          float max_abs, float P1, float P2, const int32_t margins[4], float Pout, svh_array *out, svh_array *winner_idx,
                                          int *winner_written) {
    return sgm_cost_volume_impl(ctx, n_directions, strategy, cv, minima, max_abs, P1, P2, margins, Pout, out, winner_idx, winner_written, minima ? minima_kind : 1);
}

// "Textbook" semi-global matching: NOT the reference's behaviour -- what correlation/sgm.h
// evidently intends.  All 4 / 8 directions, every line of the margin box once (the reference skips three directions and half of
// two more: finding F5); Cost strategy a(nd) = min{prev[nd], prev[nd +- 1] + P1, min_far prev + P2} (the reference assigns the
// pixel's own cost there: finding F4); Score strategy as the reference's Score branch.  Defined by oracle/stevi_oracle.c
// so_sgm_textbook; same kernels as the Score branch (the Cost form runs on negated costs).
extern "C" int svh_sgm_cost_volume_textbook(svh_context *ctx, int n_directions, int strategy, const svh_array *cv, float P1, float P2,
                                            const int32_t margins[4], float Pout, svh_array *out) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate_volume(ctx, cv, "cv"));
    SVH_TRY(validate(ctx, out, "out", SVH_F32, 3, 3));
    if (n_directions != 4 && n_directions != 8) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "n_directions must be 4 or 8");
    if (strategy != SVH_COST && strategy != SVH_SCORE) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad strategy");
    for (int k = 0; k < 3; k++)
        if (cv->shape[k] != out->shape[k]) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "out must have the shape of cv");
    static const int32_t zero[4] = {0, 0, 0, 0};
    const int32_t *m = margins ? margins : zero;
    for (int k = 0; k < 4; k++)
        if (m[k] < 0) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "margins must be non-negative");
    SgmArgs a{n_directions, strategy, (int)cv->shape[0], (int)cv->shape[1], (int)cv->shape[2], P1, P2, Pout, m[0], m[1], m[2], m[3]};
    Scratch scr(ctx);
    void *dcv;
    OutStage os;
    SVH_TRY(stage_volume_as_float(ctx, scr, *cv, &dcv));
    SVH_TRY(stage_out(ctx, scr, *out, &os));
    if (os.dptr == dcv) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "cv and out must not alias");
    SVH_TRY(dev_sgm_score_branch(ctx, scr, a, (const float *)dcv, (float *)os.dptr, true));
    return finish_out(ctx, os);
}

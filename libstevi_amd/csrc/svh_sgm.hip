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
#include "svh_internal.h"

#include <type_traits>

namespace svh {

struct LineSet {
    int pass;   // 0..5 as listed above
    int n_lines;
    int top, left, Hp, Wp; // margin box origin and extent
};

struct Line {
    int i0, j0, di, dj, len;
};

__device__ __forceinline__ Line line_of(const LineSet &ls, int l) {
    Line L;
    switch (ls.pass) {
    case 0: L = {ls.top, ls.left + l, 1, 0, ls.Hp}; break;
    case 1: L = {ls.top + l, ls.left, 0, 1, ls.Wp}; break;
    case 2: L = {ls.top + l, ls.left, 1, 1, min(ls.Hp - l, ls.Wp)}; break;
    case 3: L = {ls.top, ls.left + l, 1, 1, min(ls.Hp, ls.Wp - l)}; break;
    case 4: L = {ls.top, ls.left + l, 1, -1, min(ls.Hp, l + 1)}; break;
    case 5: L = {ls.top + l, ls.left, -1, 1, min(l + 1, ls.Wp)}; break;
    // "textbook" line sets (svh_sgm_cost_volume_textbook): the eight directions, every line of the margin box exactly once
    case 6: L = {ls.top, ls.left + l, 1, 0, ls.Hp}; break;                  // Up2Down
    case 7: L = {ls.top + ls.Hp - 1, ls.left + l, -1, 0, ls.Hp}; break;     // Down2Up
    case 8: L = {ls.top + l, ls.left, 0, 1, ls.Wp}; break;                  // Left2Right
    case 9: L = {ls.top + l, ls.left + ls.Wp - 1, 0, -1, ls.Wp}; break;     // Right2Left
    case 10: case 11: {                                                    // diagonal j - i = l - (Hp - 1), forwards / backwards
        const int k = l - (ls.Hp - 1), i0 = k <= 0 ? -k : 0, j0 = k <= 0 ? 0 : k, len = min(ls.Hp - i0, ls.Wp - j0);
        if (ls.pass == 10) L = {ls.top + i0, ls.left + j0, 1, 1, len};
        else L = {ls.top + i0 + len - 1, ls.left + j0 + len - 1, -1, -1, len};
    } break;
    default: {                                                             // anti-diagonal i + j = l, downwards / upwards
        const int i0 = l < ls.Wp ? 0 : l - (ls.Wp - 1), j0 = l < ls.Wp ? l : ls.Wp - 1, len = min(ls.Hp - i0, j0 + 1);
        if (ls.pass == 12) L = {ls.top + i0, ls.left + j0, 1, -1, len};
        else L = {ls.top + i0 + len - 1, ls.left + j0 - (len - 1), -1, 1, len};
    } break;
    }
    return L;
}

// does pass q visit pixel (ip, jp) (coordinates relative to the margin box, already known to be inside it)?
__device__ __forceinline__ bool pass_visits(int q, int ip, int jp, int Hp, int Wp) {
    switch (q) {
    case 0: case 1: return true;
    case 2: return ip >= jp;
    case 3: return jp >= ip;
    case 4: return ip + jp < Wp;
    default: return ip + jp < Hp;
    }
}

__device__ __forceinline__ bool finite_f(float x) { return fabsf(x) < INFINITY; } // false for NaN and +-inf

__device__ __forceinline__ float wave_min(float v) { // DPP: row_shr 1,2,4,8, row_bcast:15, row_bcast:31, result in lane 63
#define SVH_DPP_MIN(CTRL, RM) v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x7F800000, __builtin_bit_cast(int, v), CTRL, RM, 0xF, false)))
    SVH_DPP_MIN(0x111, 0xF);
    SVH_DPP_MIN(0x112, 0xF);
    SVH_DPP_MIN(0x114, 0xF);
    SVH_DPP_MIN(0x118, 0xF);
    SVH_DPP_MIN(0x142, 0xA);
    SVH_DPP_MIN(0x143, 0xC);
#undef SVH_DPP_MIN
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// ---- cost sources -----------------------------------------------------------------------------------
struct SrcVolume { // dense (H, W, D) floats
    const float *cv;
    int W, D;
    bool vec; // 16-byte aligned rows, D % 4 == 0
    template <int R> __device__ __forceinline__ void load(int i, int j, int lane, float (&c)[R]) const {
        const float *p = cv + ((int64_t)i * W + j) * D + lane * R;
        if constexpr (R % 4 == 0) {
            if (vec && lane * R + R <= D) {
#pragma unroll
                for (int q = 0; q < R / 4; q++) {
                    float4 v = *reinterpret_cast<const float4 *>(p + 4 * q);
                    c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
                }
                return;
            }
        }
#pragma unroll
        for (int k = 0; k < R; k++) c[k] = (lane * R + k < D) ? p[k] : 0.0f;
    }
};

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

template <class SRC, int R, int B>
__global__ void __launch_bounds__(256) sgm_cost_minmap_kernel(SRC src, LineSet ls, int D, int W, float Pout,
                                                             float *__restrict__ mmap, const int *__restrict__ run_if_nonzero) {
    if (run_if_nonzero && *run_if_nonzero == 0) return; // the exact-integer route already produced the maps
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
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

template <bool COST> __device__ __forceinline__ Winner wave_winner(Winner w) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Winner o;
        o.v = __shfl_xor(w.v, off);
        o.d = __shfl_xor(w.d, off);
        w = better<COST>(w, o);
    }
    return w;
}

// Winner over the non-NaN values of the wave (d = -1 when there is none) and whether the value at local index 0
// is NaN: the sequential scan of the reference starts from index 0 and never leaves it when that value is NaN.
template <bool COST, int R>
__device__ __forceinline__ Winner wave_select(const float (&s)[R], int lane, int D, bool *first_is_nan) {
    Winner w{0.0f, -1};
#pragma unroll
    for (int k = 0; k < R; k++) {
        int d = lane * R + k;
        if (d < D && !isnan(s[k])) w = better<COST>(w, Winner{s[k], d});
    }
    w = wave_winner<COST>(w);
    *first_is_nan = isnan(__shfl(s[0], 0));
    return w;
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

struct ApplyOut {
    float *sgm; // (H, W, D) or nullptr
    WinnerOut w;
    bool vec_store; // sgm rows 16-byte aligned, D % 4 == 0
};

template <class SRC, int R>
__global__ void __launch_bounds__(256) sgm_cost_apply_kernel(SRC src, int H, int W, int D, int top, int left, int Hp, int Wp,
                                                            int n_pass, float Pout, const float *__restrict__ mmap,
                                                            ApplyOut out) {
    const int lane = threadIdx.x & 63;
    const int64_t npx = (int64_t)H * W;
    const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    // APPLY_PB pixels per wave iteration: their cost rows and min_p values are all requested before the first one is used
    // (one pixel at a time leaves a single 1 KiB load in flight per wave and the kernel latency-bound)
    constexpr int APPLY_PB = 4, MAX_PASS = 6;
    for (int64_t p0 = wave * APPLY_PB; p0 < npx; p0 += nwaves * APPLY_PB) {
      float cb[APPLY_PB][R], mpb[APPLY_PB][MAX_PASS];
#pragma unroll
      for (int u = 0; u < APPLY_PB; u++) {
          const int64_t pu = min(p0 + u, npx - 1);
          src.template load<R>((int)(pu / W), (int)(pu % W), lane, cb[u]);
#pragma unroll
          for (int q = 0; q < MAX_PASS; q++) mpb[u][q] = q < n_pass ? mmap[(int64_t)q * npx + pu] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < APPLY_PB; u++) {
        const int64_t p = p0 + u;
        if (p >= npx) break;
        const int j = (int)(p % W), i = (int)(p / W);
        float c[R], s[R];
#pragma unroll
        for (int k = 0; k < R; k++) s[k] = c[k] = cb[u][k]; // sgm_cv := cv, sgm.h:371-377
        const int ip = i - top, jp = j - left;
        if (ip >= 0 && ip < Hp && jp >= 0 && jp < Wp) {
#pragma unroll
            for (int q = 0; q < MAX_PASS; q++) {
                if (q >= n_pass || !pass_visits(q, ip, jp, Hp, Wp)) continue;
                const float mp = mpb[u][q];
                const bool mp_fin = finite_f(mp);
#pragma unroll
                for (int k = 0; k < R; k++) {
                    int d = lane * R + k;
                    float t = (j + d >= W) ? c[k] + Pout : c[k];
                    float act = c[k];
                    if (mp_fin && finite_f(t)) act = c[k] + (t - mp);
                    s[k] += act - c[k]; // sgm.h:298-300
                }
            }
        }
        if (out.sgm) {
            float *o = out.sgm + p * D + lane * R;
            bool stored = false;
            if constexpr (R % 4 == 0) {
                if (out.vec_store && lane * R + R <= D) {
#pragma unroll
                    for (int q = 0; q < R / 4; q++) *reinterpret_cast<float4 *>(o + 4 * q) = make_float4(s[4 * q], s[4 * q + 1], s[4 * q + 2], s[4 * q + 3]);
                    stored = true;
                }
            }
            if (!stored) {
#pragma unroll
                for (int k = 0; k < R; k++)
                    if (lane * R + k < D) o[k] = s[k];
            }
        }
        const WinnerOut &wo = out.w;
        if (wo.idx || wo.disp || wo.taps || wo.keys) {
            bool first_nan;
            const Winner w = wave_select<true, R>(s, lane, D, &first_nan);
            const int sel = (first_nan || w.d < 0) ? 0 : w.d;
            if (wo.idx && lane == 0) wo.idx[p] = sel;
            if (wo.disp && lane == 0) wo.disp[p] = wo.disp_sign * sel + wo.disp_offset;
            if (wo.keys && lane == 0) wo.keys[p] = winner_key<true>(w, first_nan, wo.key_offset, wo.key_total);
            if (wo.taps) {
                // truncatedCostVolume<Same>, radius 1 (correlation_base.h:601-613)
                const bool px_bad = j < wo.taps_h_r || i < wo.taps_v_r || i + wo.taps_v_r >= H;
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const int pd = sel + t - 1;
                    const bool bad = px_bad || pd < 0 || pd >= D || j + pd + wo.taps_h_r >= W;
                    if (bad) {
                        if (lane == 0) wo.taps[p * 3 + t] = __uint_as_float(0x7FC00000u);
                    } else {
#pragma unroll
                        for (int k = 0; k < R; k++)
                            if (lane * R + k == pd) wo.taps[p * 3 + t] = s[k];
                    }
                }
            }
        }
      }
    }
}

// Integer-volume probe.  A float cost volume whose values are all small integers (a Hamming volume handed to the
// per-function API, say) is in the same exact regime as the census path: every operation of sgm.h:257-300 is exact,
// the per-pass minima follow mp' = g - mp, and six wave-per-line sweeps of the volume collapse into this one read.
// The kernel writes g(p) = min_d [c + (c [+ Pout])] and raises `flag` as soon as one value is not an integer in
// [-limit, limit]; the scan kernels run only if the flag stayed 0 and the line kernels only if it was raised, so no host
// round trip is needed.
template <int R>
__global__ void __launch_bounds__(256) volume_gmin_probe_kernel(SrcVolume src, int64_t npx, int D, int W, float Pout, float limit,
                                                               float *__restrict__ gmap, int *__restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    bool bad = false;
    constexpr int PROBE_PB = 4; // pixels per wave iteration, loads issued together
    for (int64_t p0 = wave * PROBE_PB; p0 < npx; p0 += nwaves * PROBE_PB) {
        float cb[PROBE_PB][R];
#pragma unroll
        for (int u = 0; u < PROBE_PB; u++) {
            const int64_t pu = min(p0 + u, npx - 1);
            src.template load<R>((int)(pu / W), (int)(pu % W), lane, cb[u]);
        }
#pragma unroll
        for (int u = 0; u < PROBE_PB; u++) {
            const int64_t p = p0 + u;
            if (p >= npx) break;
            const int j = (int)(p % W);
            float m = INFINITY;
#pragma unroll
            for (int k = 0; k < R; k++) {
                const int d = lane * R + k;
                if (d < D) {
                    const float c = cb[u][k];
                    bad = bad || !(fabsf(c) <= limit) || c != rintf(c);
                    const float t = (j + d >= W) ? c + Pout : c;
                    m = fminf(m, c + t);
                }
            }
            m = wave_min(m);
            if (lane == 0) gmap[p] = m;
        }
    }
    if (__any(bad) && lane == 0) atomicOr(flag, 1);
}

// ---- Score branch: read-modify-write sweep per pass ----------------------------------------------------
// cross-lane primitives on the DPP path (no LDS round trip): whole-wave shifts by one lane and a max reduction
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ float dpp_move(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float lane_shift_up(float v, float fill) { return dpp_move<0x138>(fill, v); }   // wave_shr:1 -> value of lane-1
__device__ __forceinline__ float lane_shift_down(float v, float fill) { return dpp_move<0x130>(fill, v); } // wave_shl:1 -> value of lane+1
// inclusive prefix maximum over the lanes (lane 63 ends up with the wave maximum): row_shr 1,2,4,8 inside each row
// of 16, then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3
__device__ __forceinline__ float wave_prefix_max(float v) {
    v = fmaxf(v, dpp_move<0x111>(-INFINITY, v));
    v = fmaxf(v, dpp_move<0x112>(-INFINITY, v));
    v = fmaxf(v, dpp_move<0x114>(-INFINITY, v));
    v = fmaxf(v, dpp_move<0x118>(-INFINITY, v));
    v = fmaxf(v, dpp_move<0x142, 0xA>(-INFINITY, v));
    v = fmaxf(v, dpp_move<0x143, 0xC>(-INFINITY, v));
    return v;
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_prefix_max(v)), 63));
}

// R consecutive floats of a lane as 16 / 8 / 4-byte accesses (any address space)
template <int R> __device__ __forceinline__ void lds_get(const float *p, float (&v)[R]) {
    if constexpr (R % 4 == 0) {
#pragma unroll
        for (int q = 0; q < R / 4; q++) {
            const float4 x = *reinterpret_cast<const float4 *>(p + 4 * q);
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        }
    } else if constexpr (R == 2) {
        const float2 x = *reinterpret_cast<const float2 *>(p);
        v[0] = x.x; v[1] = x.y;
    } else {
#pragma unroll
        for (int k = 0; k < R; k++) v[k] = p[k];
    }
}
template <int R> __device__ __forceinline__ void lds_put(float *p, const float (&v)[R]) {
    if constexpr (R % 4 == 0) {
#pragma unroll
        for (int q = 0; q < R / 4; q++) *reinterpret_cast<float4 *>(p + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else if constexpr (R == 2) {
        *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
    } else {
#pragma unroll
        for (int k = 0; k < R; k++) p[k] = v[k];
    }
}

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
template <int R, int B, bool FIRST, bool FAR_IS_GLOBAL, bool NEG, bool DELTA = false, bool VEC = false>
__global__ void __launch_bounds__(256) sgm_score_pass_kernel(const float *__restrict__ cv, float *__restrict__ sgm, LineSet ls,
                                                            int D, int W, float P1, float P2, float Pout, bool vec) {
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


// ---- Score branch: the four downward passes in one sweep -----------------------------------------------------
// Passes 0, 2, 3 and 4 all step one row down: a pixel (i, j) takes its line state from (i-1, j), (i-1, j-1) and (i-1, j+1).  One
// sweep over the rows can therefore carry the three states, add the contributions in the reference's order
//     S = (((c + d0) + d1) [+ d2 if i >= j] [+ d3 if j >= i]) [+ d4 if i + j < W]        (d2 = d3: the same diagonal lines)
// in registers and touch the volume once (read c and d1, write S: 12 B/voxel) where the pass-per-launch form reads and writes S
// once per pass.  d1, the Left2Right contribution, is written beforehand by the DELTA form of the kernel above.
//
// Parallel form.  In the skewed coordinate u = i + j the three predecessors of (i, u) are (i-1, u-1), (i-1, u-2) and (i-1, u):
// none lies to the right.  The image is cut into strips of WB consecutive u (parallelograms leaning left), one block per strip,
// and a strip needs from its left neighbour, per row, the Up2Down state of its last cell and the diagonal states of its last two
// cells: 3 vectors of D floats.  Blocks form a one-directional pipeline: strip s writes row i's vectors to a global edge buffer
// (exporter wave) as 8-byte {value, tag} granules, agent-scope stores, the tag being the number of this launch; strip s + 1
// reads them before its row i + 1 (importer wave: agent-scope loads, again until every tag matches, then into the state rings in
// LDS).  A granule is written by one store instruction, so its tag vouches for its value and no flag, drain or fence is needed:
// a hop costs one store and one load.  The chain is as long as the number of strips (each row of strip s waits for the previous
// row of strip s - 1), so the sweep takes (rows x time per row) + (strips x hop), and the hop is what matters.  A strip takes
// its number from a ticket counter, so the strip it waits for always started before it: the pipeline cannot deadlock whatever
// the number of resident blocks.
//
// State in LDS, one slot per LINE, updated in place: Up2Down by column (ring of WB + 1), diagonal by j - i (ring of WB + 2),
// anti-diagonal by u (WB fixed slots); the spare slots receive next row's imports while this row still reads the leaving ones.
// One barrier per row; NCW compute waves (a cell = all three passes of one pixel, the 64 lanes span the disparities as in the
// kernel above), then the exporter and the importer wave.
// Three line states of one pixel at a time, written without per-lane branches: the sweep is bound by instruction issue, not by
// memory.  The three wave maxima run interleaved in hand-written DPP steps (v_max_f32_dpp with itself: one instruction per step
// and state, and two other instructions between a register's write and its next DPP read, which is the hazard distance; the
// compiler's form is v_mov, v_mov_dpp, a canonicalising v_max and the v_max).  The values are finite or -inf here, never NaN.
// POUT: some disparity of this pixel looks past the image border (wave-uniform; false for most pixels, which skip the term).
// TAIL: D < 64 R, the lanes past D are masked.
#define SVH_MAX3_DPP(CTRL)                                                                                                        \
    asm("v_max_f32_dpp %0, %0, %0 " CTRL "\n\tv_max_f32_dpp %1, %1, %1 " CTRL "\n\tv_max_f32_dpp %2, %2, %2 " CTRL : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]))
template <int R, bool POUT, bool TAIL>
__device__ __forceinline__ void score_step3_far_global(const float (&prev)[3][R], const float (&c)[R], int jj, int lane, int D, int W, float P1,
                                                       float P2, float Pout, float (&act)[3][R]) {
    float pf[3][R], A[3];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        A[s] = -INFINITY;
#pragma unroll
        for (int k = 0; k < R; k++) {
            const bool keep = TAIL ? ((lane * R + k < D) & finite_f(prev[s][k])) : finite_f(prev[s][k]); // isfinite filters of sgm.h:224, :241
            pf[s][k] = keep ? prev[s][k] : -INFINITY;
            A[s] = fmaxf(A[s], pf[s][k]);
        }
    }
    asm("s_nop 1" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]));
    SVH_MAX3_DPP("row_shr:1 row_mask:0xf bank_mask:0xf");
    SVH_MAX3_DPP("row_shr:2 row_mask:0xf bank_mask:0xf");
    SVH_MAX3_DPP("row_shr:4 row_mask:0xf bank_mask:0xf");
    SVH_MAX3_DPP("row_shr:8 row_mask:0xf bank_mask:0xf");
    SVH_MAX3_DPP("row_bcast:15 row_mask:0xa bank_mask:0xf");
    SVH_MAX3_DPP("row_bcast:31 row_mask:0xc bank_mask:0xf");
    asm("s_nop 0" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]));
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const float max_p = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, A[s]), 63)); // :220-227
        const float prevL = lane_shift_up(pf[s][R - 1], -INFINITY);                // disparity lane*R - 1
        const float prevR = lane_shift_down(pf[s][0], -INFINITY);                  // disparity lane*R + R
        const bool maxp_fin = finite_f(max_p);
        const float far = max_p - P2;                                              // :239 (FAR_IS_GLOBAL, see above)
#pragma unroll
        for (int k = 0; k < R; k++) {
            const float lo = (k > 0 ? pf[s][k - 1] : prevL) - P1;                  // :238
            const float hi = (k < R - 1 ? pf[s][k + 1] : prevR) - P1;
            float a = fmaxf(fmaxf(pf[s][k], far), fmaxf(lo, hi));
            if constexpr (POUT) {
                const float a_out = a - Pout;                                      // :247-249
                a = (jj + lane * R + k >= W) ? a_out : a;
            }
            const float moved = c[k] + (a - max_p);                                // :251-254
            act[s][k] = (maxp_fin & finite_f(a)) ? moved : c[k];
        }
    }
}
#undef SVH_MAX3_DPP

constexpr int kDownSpinCap = 1 << 21; // reloads of a neighbour's row before a block gives up (seconds; a healthy wait is microseconds)

template <int R, int WB, int NCW, bool VEC>
__global__ void __launch_bounds__((NCW + 2) * 64) sgm_score_down_kernel(const float *__restrict__ cv, float *sgm, int H, int W, int D, float P1,
                                                                        float P2, float Pout, uint64_t *edges, uint32_t tag, int *sync_words,
                                                                        unsigned long long *stamps) {
    constexpr int DP = 64 * R, MV = WB + 1, MD = WB + 2, CPW = WB / NCW;
    constexpr int NB = R <= 4 ? 4 : 3; // register sets of the row prefetch
    static_assert(WB % NCW == 0, "cells of a row are dealt evenly to the compute waves");
    extern __shared__ __attribute__((aligned(16))) float down_lds[];
    float *ringV = down_lds, *ringD = ringV + MV * DP, *ringA = ringD + MD * DP, *stage = ringA + WB * DP; // stage[2][3][DP]
    __shared__ int s_strip;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63; // (the wave index in a scalar register)
    int *ticket = sync_words, *error = sync_words + 1;
    if (threadIdx.x == 0) s_strip = atomicAdd(ticket, 1);
    __syncthreads();
    const int s = s_strip, n_strips = gridDim.x;
    const int u0 = s * WB;
    const int i_lo = max(0, u0 - (W - 1)), i_hi = min(H - 1, u0 + WB - 1);         // rows in which the strip meets the image
    const int l_lo = max(0, u0 - WB - (W - 1)), l_hi = min(H - 1, u0 - 1);          // the left neighbour's
    const int64_t strip_floats = (int64_t)H * 3 * DP;
    uint64_t *my_edges = edges + (int64_t)s * strip_floats;
    const uint64_t *left_edges = edges + (int64_t)(s - 1) * strip_floats;
    auto slotV = [&](int j) { return ((j % MV) + MV) % MV; };
    auto slotD = [&](int k) { return ((k % MD) + MD) % MD; };
    // VEC: 16-byte aligned volumes and D = 64 R exactly (every lane holds R disparities of the pixel): vector loads, no tail
    auto load_px = [&](const float *vol, int i, int j, float (&v)[R]) {
        const float *p = vol + ((int64_t)i * W + j) * D + lane * R;
        if constexpr (VEC) lds_get<R>(p, v); // (plain 16 / 8 / 4-byte loads: the helper does not care about the address space)
        else {
#pragma unroll
            for (int k = 0; k < R; k++) v[k] = (lane * R + k < D) ? p[k] : 0.0f;
        }
    };

    if (wave == NCW + 1) {
        // ---- importer: what row r needs from the left strip's row r - 1, into the spare ring slots.  The loads for row i + 2 are
        // issued while the compute waves are on row i and examined a row later; whatever tag does not match is loaded again.
        const bool has_left = s > 0;
        auto needs = [&](int r) { return has_left && r <= i_hi && r - 1 >= l_lo && r - 1 <= l_hi; };
        auto request = [&](int r, uint64_t (&g)[3][R]) {
            const uint64_t *e = left_edges + (int64_t)(r - 1) * 3 * DP + lane * R;
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int k = 0; k < R; k++) g[q][k] = __hip_atomic_load(e + q * DP + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto complete = [&](const uint64_t (&g)[3][R]) {
            bool ok = true;
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int k = 0; k < R; k++) ok &= (uint32_t)(g[q][k] >> 32) == tag;
            return __all(ok) != 0;
        };
        auto deliver = [&](int r, uint64_t (&g)[3][R]) {
            int spins = 0;
            while (!complete(g)) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kDownSpinCap) {
                    __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                request(r, g);
            }
            float v[3][R];
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int k = 0; k < R; k++) v[q][k] = __uint_as_float((uint32_t)g[q][k]);
            lds_put<R>(ringV + slotV(u0 - r) * DP + lane * R, v[0]);
            lds_put<R>(ringD + slotD(u0 - 2 * r) * DP + lane * R, v[1]);
            lds_put<R>(ringD + slotD(u0 - 2 * r + 1) * DP + lane * R, v[2]);
        };
        uint64_t g[3][R];
        if (needs(i_lo)) {
            request(i_lo, g);
            deliver(i_lo, g);
        }
        if (needs(i_lo + 1)) request(i_lo + 1, g);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int i = i_lo; i <= i_hi; i++) {
            if (needs(i + 1)) deliver(i + 1, g);
            if (needs(i + 2)) request(i + 2, g);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    if (wave == NCW) {
        // ---- exporter: the row just finished, from the staging slots to the edge buffer
        const bool has_right = s + 1 < n_strips;
        auto export_row = [&](int r) {
            if (!has_right || r + 1 > H - 1) return; // nobody reads it
            float v[3][R];
#pragma unroll
            for (int q = 0; q < 3; q++) lds_get<R>(stage + ((r & 1) * 3 + q) * DP + lane * R, v[q]);
            uint64_t *e = my_edges + (int64_t)r * 3 * DP + lane * R;
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
                for (int k = 0; k < R; k++)
                    __hip_atomic_store(e + q * DP + k, ((uint64_t)tag << 32) | __float_as_uint(v[q][k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int i = i_lo; i <= i_hi; i++) {
            if (i > i_lo) export_row(i - 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        export_row(i_hi);
        return;
    }

    // ---- compute waves: cells t = wave, wave + NCW, ... of the strip's row
    // one pixel: the three line states, the sum in the reference's order, the store.  INNER: i >= 1 and j >= 1 (no line starts here)
    auto cell = [&](auto pout_tag, auto inner_tag, const float (&c)[R], const float (&d1)[R], float *st, int t, int i, int j) {
        constexpr bool POUT = decltype(pout_tag)::value, INNER = decltype(inner_tag)::value;
        float prev[3][R], act[3][R];
        float *pv = ringV + slotV(j) * DP + lane * R;       // Up2Down (pass 0)
        float *pd = ringD + slotD(j - i) * DP + lane * R;   // UpLeft2DownRight (passes 2 and 3: lines from the left and from the top border)
        float *pa = ringA + t * DP + lane * R;              // UpRight2DownLeft (pass 4): only the lines that start on the top border (F5)
        const bool visA = i + j < W;
        lds_get<R>(pv, prev[0]);
        lds_get<R>(pd, prev[1]);
        lds_get<R>(pa, prev[2]);
        if constexpr (!INNER) {
            if (i == 0 || j == 0) { // (wave-uniform) a line's first pixel sees prev = 0 (sgm.h:206-208)
#pragma unroll
                for (int k = 0; k < R; k++) {
                    prev[1][k] = 0.0f;
                    if (i == 0) prev[0][k] = prev[2][k] = 0.0f;
                }
            }
        }
        score_step3_far_global<R, POUT, !VEC>(prev, c, j, lane, D, W, P1, P2, Pout, act);
        lds_put<R>(pv, act[0]);
        lds_put<R>(pd, act[1]);
        if (visA) lds_put<R>(pa, act[2]);
        if (t == WB - 1) {
            lds_put<R>(st, act[0]);
            lds_put<R>(st + 2 * DP, act[1]);
        }
        if (t == WB - 2) lds_put<R>(st + DP, act[1]);
        float outv[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            float S = c[k] + (act[0][k] - c[k]); // sgm.h:298-300, pass after pass
            S = S + d1[k];
            const float dD = act[1][k] - c[k];
            S = S + dD;                           // pass 2 (i >= j) or pass 3 (j >= i) ...
            if (i == j) S = S + dD;               // ... and both on the main diagonal: the corner line runs twice
            const float S4 = S + (act[2][k] - c[k]);
            outv[k] = visA ? S4 : S;
        }
        float *o = sgm + ((int64_t)i * W + j) * D + lane * R;
        if constexpr (VEC) lds_put<R>(o, outv);
        else {
#pragma unroll
            for (int k = 0; k < R; k++)
                if (lane * R + k < D) o[k] = outv[k];
        }
    };
    int stamp_row = i_lo;
    auto row_end = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (stamps) { // (diagnostic launches only: when each strip passed each row's barrier, 100 MHz ticks)
            if (threadIdx.x == 0 && stamp_row <= i_hi + 1) stamps[(int64_t)s * (H + 1) + stamp_row] = __builtin_amdgcn_s_memrealtime();
            stamp_row++;
        }
    };
    // Rows in which the strip enters or leaves the image (some cells outside), row 0 and the row that holds the strip's j == 0
    // pixel: a few per strip, loaded where they are used.
    auto edge_rows = [&](int lo, int hi) {
        for (int i = lo; i <= hi; i++) {
            float *st = stage + (i & 1) * 3 * DP + lane * R;
#pragma unroll
            for (int q = 0; q < CPW; q++) {
                const int t = wave + q * NCW, j = u0 + t - i;
                if (j < 0 || j >= W) continue;
                float c[R], d1[R];
                load_px(cv, i, j, c);
                load_px(sgm, i, j, d1);
                cell(std::true_type{}, std::false_type{}, c, d1, st, t, i, j);
            }
            row_end();
        }
    };
    // Rows with all WB cells inside the image and no line start: every load and store is unconditional (the compiler's count of
    // outstanding memory operations stays exact: a load under a branch makes it wait for the newest loads, which is the end of any
    // prefetch), NB register sets in rotation: the loads of rows i + 1 .. i + NB - 1 are in flight while row i is computed.
    auto full_rows = [&](auto pout_tag, int lo, int hi) {
        if (lo > hi) return;
        auto load_row = [&](float (&c)[CPW][R], float (&d1)[CPW][R], int i) {
            i = min(i, hi); // (past the end: the last row again, unused)
#pragma unroll
            for (int q = 0; q < CPW; q++) {
                const int j = u0 + wave + q * NCW - i;
                load_px(cv, i, j, c[q]);
                load_px(sgm, i, j, d1[q]);
            }
        };
        auto run_row = [&](const float (&c)[CPW][R], const float (&d1)[CPW][R], int i) {
            float *st = stage + (i & 1) * 3 * DP + lane * R;
#pragma unroll
            for (int q = 0; q < CPW; q++) cell(pout_tag, std::true_type{}, c[q], d1[q], st, wave + q * NCW, i, u0 + wave + q * NCW - i);
            row_end();
        };
        float cb[NB][CPW][R], eb[NB][CPW][R];
#pragma unroll
        for (int p = 0; p < NB - 1; p++) load_row(cb[p], eb[p], lo + p);
        for (int i = lo; i <= hi; i += NB) {
#pragma unroll
            for (int p = 0; p < NB; p++) {
                if (i + p > hi) break;
                load_row(cb[(p + NB - 1) % NB], eb[(p + NB - 1) % NB], i + p + NB - 1);
                run_row(cb[p], eb[p], i + p);
            }
        }
    };
    // full rows: j = u0 + t - i in [1, W) for t = 0 .. WB-1, i >= 1; the first of them may look past the right border (Pout)
    const int f_lo = max(max(i_lo, 1), u0 + WB - W), f_hi = min(i_hi, u0 - 1);
    const int p_hi = min(f_hi, u0 + WB - 2 + D - W); // rows in which the rightmost cell has j + D > W
    row_end(); // (the importer's prologue)
    stamp_row = i_lo + 1;
    if (f_lo > f_hi) edge_rows(i_lo, i_hi);
    else {
        edge_rows(i_lo, f_lo - 1);
        full_rows(std::true_type{}, f_lo, p_hi);
        full_rows(std::false_type{}, max(f_lo, p_hi + 1), f_hi);
        edge_rows(f_hi + 1, i_hi);
    }
}

// ---- Score branch: the four downward passes, a band of rows per launch -----------------------------------------
// The same fusion without any hand-off between blocks: a launch covers KB image rows, a block a strip of WB columns, and what a
// block would need from its neighbours during the band it computes itself -- the diagonal lines that enter its strip from the left
// (a triangle of at most KB - 1 columns, one column narrower every row) and the anti-diagonal lines that enter from the right.
// Line states cross from band to band through two global arrays (the previous band's last row, read; this band's last row,
// written: 3 x W vectors of D floats each), the launch boundary is the only synchronisation.  Per band and block this costs
// (KB - 1) KB / 2 extra single-pass pixels on either side against 3 WB KB pass-pixels of its own (+ 31 % at WB = KB = 16) and the
// carried states (+ 10 % of the band's bytes); nothing spins.
template <int R, bool POUT, bool TAIL>
__device__ __forceinline__ void score_step1_far_global(const float (&prev)[R], const float (&c)[R], int jj, int lane, int D, int W, float P1,
                                                       float P2, float Pout, float (&act)[R]) {
    float pf[R];
    float A = -INFINITY;
#pragma unroll
    for (int k = 0; k < R; k++) {
        const bool keep = TAIL ? ((lane * R + k < D) & finite_f(prev[k])) : finite_f(prev[k]);
        pf[k] = keep ? prev[k] : -INFINITY;
        A = fmaxf(A, pf[k]);
    }
    const float max_p = wave_max_dpp(A);
    const float prevL = lane_shift_up(pf[R - 1], -INFINITY);
    const float prevR = lane_shift_down(pf[0], -INFINITY);
    const bool maxp_fin = finite_f(max_p);
    const float far = max_p - P2;
#pragma unroll
    for (int k = 0; k < R; k++) {
        const float lo = (k > 0 ? pf[k - 1] : prevL) - P1;
        const float hi = (k < R - 1 ? pf[k + 1] : prevR) - P1;
        float a = fmaxf(fmaxf(pf[k], far), fmaxf(lo, hi));
        if constexpr (POUT) {
            const float a_out = a - Pout;
            a = (jj + lane * R + k >= W) ? a_out : a;
        }
        const float moved = c[k] + (a - max_p);
        act[k] = (maxp_fin & finite_f(a)) ? moved : c[k];
    }
}

template <int R, int WB, int KB, int NCW, bool VEC>
__global__ void __launch_bounds__(NCW * 64) sgm_score_band_kernel(const float *__restrict__ cv, float *sgm, int H, int W, int D, float P1, float P2,
                                                                  float Pout, int r0, int rows, const float *__restrict__ st_in,
                                                                  float *__restrict__ st_out) {
    constexpr int DP = 64 * R, CPW = WB / NCW, NL = WB + KB - 1, NH = 2 * (KB - 1), HS = (NH + NCW - 1) / NCW;
    constexpr int NB = R <= 4 ? 3 : 2; // register sets of the row prefetch (4: no faster)
    static_assert(WB % NCW == 0, "cells of a row are dealt evenly to the waves");
    extern __shared__ __attribute__((aligned(16))) float band_lds[];
    float *ringV = band_lds, *lineD = ringV + WB * DP, *lineA = lineD + NL * DP;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j0 = blockIdx.x * WB;
    const int64_t plane = (int64_t)W * DP; // st_in / st_out: [pass V, D, A][column][DP]
    auto row_end = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // ---- the line states of row r0 - 1 (nothing to load for the first band: every line starts inside it)
    if (r0 > 0) {
        float v[R];
        for (int slot = wave; slot < WB; slot += NCW) {
            const int j = j0 + slot;
            if (j < W) {
                lds_get<R>(st_in + (int64_t)j * DP + lane * R, v);
                lds_put<R>(ringV + slot * DP + lane * R, v);
            }
        }
        for (int slot = wave; slot < NL; slot += NCW) {
            const int jd = j0 - (rows - 1) + slot - 1; // predecessor (r0 - 1, jd) of the diagonal cell (r0, jd + 1) in slot `slot`
            if (jd >= 0 && jd < W) {
                lds_get<R>(st_in + plane + (int64_t)jd * DP + lane * R, v);
                lds_put<R>(lineD + slot * DP + lane * R, v);
            }
            const int ja = j0 + slot + 1; // predecessor (r0 - 1, ja) of the anti-diagonal cell (r0, ja - 1)
            if (ja < W) {
                lds_get<R>(st_in + 2 * plane + (int64_t)ja * DP + lane * R, v);
                lds_put<R>(lineA + slot * DP + lane * R, v);
            }
        }
    }
    row_end();
    auto load_px = [&](const float *vol, int i, int j, float (&v)[R]) {
        const float *p = vol + ((int64_t)i * W + j) * D + lane * R;
        if constexpr (VEC) lds_get<R>(p, v);
        else {
#pragma unroll
            for (int k = 0; k < R; k++) v[k] = (lane * R + k < D) ? p[k] : 0.0f;
        }
    };
    // halo slot hs of row r: the pixel, the line slot and whether it is needed (left: diagonal lines, right: anti-diagonal lines)
    auto halo_of = [&](int hs, int r, int &j, int &slot, bool &left) {
        left = hs < KB - 1;
        const int m = left ? hs + 1 : hs - (KB - 1) + 1;
        j = left ? j0 - m : j0 + WB - 1 + m;
        slot = left ? rows - 1 - r - m : WB - 1 + m + r;
        return m <= rows - 1 - r && j >= 0 && j < W && (left || (r0 + r) + j < W);
    };
    auto run = [&](auto full_tag, auto pout_tag) {
        constexpr bool FULL = decltype(full_tag)::value, POUT = decltype(pout_tag)::value; // FULL: the whole strip is inside the image
        auto load_row = [&](float (&c)[CPW][R], float (&d1)[CPW][R], float (&hc)[HS][R], int r) {
            r = min(r, rows - 1); // (past the band: the last row again, unused)
            const int i = r0 + r;
#pragma unroll
            for (int q = 0; q < CPW; q++) {
                const int j = FULL ? j0 + wave + q * NCW : min(j0 + wave + q * NCW, W - 1);
                load_px(cv, i, j, c[q]);
                load_px(sgm, i, j, d1[q]);
            }
#pragma unroll
            for (int q = 0; q < HS; q++) {
                int j, slot;
                bool left;
                (void)halo_of(wave + q * NCW, r, j, slot, left);
                load_px(cv, i, min(max(j, 0), W - 1), hc[q]); // (every slot loads, needed or not: the count of loads in flight stays exact)
            }
        };
        auto run_row = [&](const float (&c)[CPW][R], const float (&d1)[CPW][R], const float (&hc)[HS][R], int r) {
            const int i = r0 + r;
#pragma unroll
            for (int q = 0; q < CPW; q++) {
                const int t = wave + q * NCW, j = j0 + t;
                if (!FULL && j >= W) continue;
                float prev[3][R], act[3][R];
                float *pv = ringV + t * DP + lane * R, *pd = lineD + (t - r + rows - 1) * DP + lane * R, *pa = lineA + (t + r) * DP + lane * R;
                const bool visA = i + j < W;
                lds_get<R>(pv, prev[0]);
                lds_get<R>(pd, prev[1]);
                lds_get<R>(pa, prev[2]);
                if (i == 0 || j == 0) { // (wave-uniform) a line's first pixel sees prev = 0 (sgm.h:206-208)
#pragma unroll
                    for (int k = 0; k < R; k++) {
                        prev[1][k] = 0.0f;
                        if (i == 0) prev[0][k] = prev[2][k] = 0.0f;
                    }
                }
                score_step3_far_global<R, POUT, !VEC>(prev, c[q], j, lane, D, W, P1, P2, Pout, act);
                lds_put<R>(pv, act[0]);
                lds_put<R>(pd, act[1]);
                if (visA) lds_put<R>(pa, act[2]);
                float outv[R];
#pragma unroll
                for (int k = 0; k < R; k++) {
                    float S = c[q][k] + (act[0][k] - c[q][k]); // sgm.h:298-300, pass after pass
                    S = S + d1[q][k];
                    const float dD = act[1][k] - c[q][k];
                    S = S + dD;                               // pass 2 (i >= j) or pass 3 (j >= i) ...
                    if (i == j) S = S + dD;                   // ... and both on the main diagonal
                    const float S4 = S + (act[2][k] - c[q][k]);
                    outv[k] = visA ? S4 : S;
                }
                float *o = sgm + ((int64_t)i * W + j) * D + lane * R;
                if constexpr (VEC) lds_put<R>(o, outv);
                else {
#pragma unroll
                    for (int k = 0; k < R; k++)
                        if (lane * R + k < D) o[k] = outv[k];
                }
            }
            // the neighbours' pixels whose lines reach this strip before the band ends: their one pass, state only
#pragma unroll
            for (int q = 0; q < HS; q++) {
                int j, slot;
                bool left;
                if (!halo_of(wave + q * NCW, r, j, slot, left)) continue;
                float prev[R], act[R];
                float *ps = (left ? lineD : lineA) + slot * DP + lane * R;
                lds_get<R>(ps, prev);
                if (i == 0 || (left && j == 0)) {
#pragma unroll
                    for (int k = 0; k < R; k++) prev[k] = 0.0f;
                }
                score_step1_far_global<R, POUT, !VEC>(prev, hc[q], j, lane, D, W, P1, P2, Pout, act);
                lds_put<R>(ps, act);
            }
            row_end();
        };
        float cb[NB][CPW][R], eb[NB][CPW][R], hb[NB][HS][R];
#pragma unroll
        for (int p = 0; p < NB - 1; p++) load_row(cb[p], eb[p], hb[p], p);
        for (int r = 0; r < rows; r += NB) {
#pragma unroll
            for (int p = 0; p < NB; p++) {
                if (r + p >= rows) break;
                load_row(cb[(p + NB - 1) % NB], eb[(p + NB - 1) % NB], hb[(p + NB - 1) % NB], r + p + NB - 1);
                run_row(cb[p], eb[p], hb[p], r + p);
            }
        }
    };
    const bool full = j0 + WB <= W, pout = j0 + WB - 1 + (KB - 1) + D > W; // (block uniform; the Pout form is right for every pixel)
    if (full) {
        if (pout) run(std::true_type{}, std::true_type{});
        else run(std::true_type{}, std::false_type{});
    } else run(std::false_type{}, std::true_type{});
    // ---- this band's last row: the line states the next band starts from (own columns only)
    float v[R];
    for (int slot = wave; slot < WB; slot += NCW) {
        const int j = j0 + slot;
        if (j >= W) continue;
        lds_get<R>(ringV + slot * DP + lane * R, v);
        lds_put<R>(st_out + (int64_t)j * DP + lane * R, v);
        lds_get<R>(lineD + slot * DP + lane * R, v);
        lds_put<R>(st_out + plane + (int64_t)j * DP + lane * R, v);
        lds_get<R>(lineA + (slot + rows - 1) * DP + lane * R, v);
        lds_put<R>(st_out + 2 * plane + (int64_t)j * DP + lane * R, v);
    }
}

// ---- host side ------------------------------------------------------------------------------------------
static int pass_lines(int q, int Hp, int Wp) {
    if (q >= 10) return Hp + Wp - 1;
    if (q >= 6) return q < 8 ? Wp : Hp;
    return (q == 0 || q == 3 || q == 4) ? Wp : Hp;
}

static int pick_R(int D) {
    int R = 1;
    while (64 * R < D) R <<= 1;
    return R;
}

static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// phase 1: the per-pass min_p maps (the sequential part); phase 2: rebuild S / pick the winner per pixel
template <class SRC, int R>
static int run_cost_branch(svh_context *ctx, const SgmArgs &a, const SRC &src, float *mmap, const ApplyOut *out, bool do_minmaps,
                           const int *gate = nullptr) {
    const int Hp = a.H - a.top - a.bottom, Wp = a.W - a.left - a.right;
    const int n_pass = a.n_dir >= 8 ? 6 : (a.n_dir >= 4 ? 2 : 0); // n_dir == 0: no aggregation, S = C
    constexpr int B = (R <= 4) ? 4 : (R == 8 ? 2 : 1);
    if (do_minmaps && Hp > 0 && Wp > 0) {
        for (int q = 0; q < n_pass; q++) {
            LineSet ls{q, pass_lines(q, Hp, Wp), a.top, a.left, Hp, Wp};
            SVH_LAUNCH(ctx, "sgm_cost_minmap", (sgm_cost_minmap_kernel<SRC, R, B>), ceil_div(ls.n_lines, 4), 256, 0, src, ls, a.D, a.W,
                       a.Pout, mmap + (size_t)q * a.H * a.W, gate);
            SVH_CHECK_LAUNCH(ctx);
        }
    }
    if (out) {
        const int64_t npx = (int64_t)a.H * a.W;
        int grid = grid_for(npx, 4, 256 * 8 * 4);
        SVH_LAUNCH(ctx, "sgm_cost_apply", (sgm_cost_apply_kernel<SRC, R>), grid, 256, 0, src, a.H, a.W, a.D, a.top, a.left, Hp > 0 ? Hp : 0,
                   Wp > 0 ? Wp : 0, (Hp > 0 && Wp > 0) ? n_pass : 0, a.Pout, mmap, *out);
        SVH_CHECK_LAUNCH(ctx);
    }
    return SVH_OK;
}

template <class SRC>
static int dispatch_cost_branch(svh_context *ctx, const SgmArgs &a, const SRC &src, float *mmap, const ApplyOut *out, bool do_minmaps,
                                const int *gate = nullptr) {
    switch (pick_R(a.D)) {
    case 1: return run_cost_branch<SRC, 1>(ctx, a, src, mmap, out, do_minmaps, gate);
    case 2: return run_cost_branch<SRC, 2>(ctx, a, src, mmap, out, do_minmaps, gate);
    case 4: return run_cost_branch<SRC, 4>(ctx, a, src, mmap, out, do_minmaps, gate);
    case 8: return run_cost_branch<SRC, 8>(ctx, a, src, mmap, out, do_minmaps, gate);
    case 16: return run_cost_branch<SRC, 16>(ctx, a, src, mmap, out, do_minmaps, gate);
    default: return fail(ctx, SVH_ERR_UNSUPPORTED, "SGM supports at most 1024 disparities (got %d)", a.D);
    }
}

int dev_sgm_cost_branch(svh_context *ctx, Scratch &scr, const SgmArgs &a, const CostSource &cs, float *out_sgm, const WinnerOut &win) {
    if ((int64_t)a.H * a.W * a.D == 0) return SVH_OK;
    float *mmap = scr.get_n<float>((size_t)6 * a.H * a.W);
    if (!mmap) return SVH_ERR_OUT_OF_MEMORY;
    ApplyOut out{out_sgm, win, out_sgm && aligned16(out_sgm) && a.D % 4 == 0};
    if (cs.cv) {
        SrcVolume src{cs.cv, a.W, a.D, aligned16(cs.cv) && a.D % 4 == 0};
        const int Hp = a.H - a.top - a.bottom, Wp = a.W - a.left - a.right;
        const int n_pass = a.n_dir >= 8 ? 6 : (a.n_dir >= 4 ? 2 : 0);
        // largest |c| for which 8 (2 |c| + |Pout|) (L + 2) stays below 2^24 (same bound as census_exact_regime)
        const double L = (double)std::max(a.H, a.W);
        const double limit = std::floor((16777216.0 / (8.0 * (L + 2.0)) - std::fabs((double)a.Pout)) / 2.0) - 1.0;
        const bool try_exact = ctx->census_fast_path && n_pass > 0 && Hp > 0 && Wp > 0 && std::isfinite(a.Pout) &&
                               a.Pout == std::nearbyint(a.Pout) && limit >= 1.0 && a.D <= 1024;
        if (!try_exact) return dispatch_cost_branch(ctx, a, src, mmap, &out, true);
        const int64_t npx = (int64_t)a.H * a.W;
        float *gmap = scr.get_n<float>((size_t)npx);
        int *flag = scr.get_n<int>(64);
        if (!gmap || !flag) return SVH_ERR_OUT_OF_MEMORY;
        SVH_HIP_CHECK(ctx, hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
        const int grid = grid_for(npx, 4, 256 * 8 * 4);
#define SVH_PROBE(RV) SVH_LAUNCH(ctx, "sgm_volume_probe", volume_gmin_probe_kernel<RV>, grid, 256, 0, src, npx, a.D, a.W, a.Pout, (float)limit, gmap, flag)
        switch (pick_R(a.D)) {
        case 1: SVH_PROBE(1); break;
        case 2: SVH_PROBE(2); break;
        case 4: SVH_PROBE(4); break;
        case 8: SVH_PROBE(8); break;
        default: SVH_PROBE(16); break;
        }
#undef SVH_PROBE
        SVH_CHECK_LAUNCH(ctx);
        SVH_TRY(dev_census_scans(ctx, a, nullptr, gmap, true, mmap, flag)); // run only while the flag is 0
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
    if (exact) SVH_TRY(dev_census_sweep_and_scans(ctx, scr, a, cs, mmap, &keys));
    if (!exact || !lane_winner) SVH_TRY(dispatch_cost_branch(ctx, a, src, mmap, lane_winner ? nullptr : &out, !exact));
    if (lane_winner && win.any()) {
        if (exact) SVH_TRY(dev_census_finalize(ctx, a, cs, mmap, keys, win));
        else SVH_TRY(dev_census_apply_select(ctx, a, cs, mmap, win));
    }
    return SVH_OK;
}

// Passes 1 (contribution only), 0 + 2 + 3 + 4 (the downward sweep) and 5: 8 + 12 + 12 * coverage(5) bytes per voxel instead of
// 8 + 12 * (coverage of passes 1-5).  Whole-image aggregation, P2 >= P1 >= 0, up to 512 disparities; anything else takes the
// pass-per-launch form.  Returns SVH_OK with *ran = false when it does not apply.
template <int R>
static int run_score_branch_fused(svh_context *ctx, Scratch &scr, const SgmArgs &a, const float *cv, float *sgm, bool vec, bool *ran) {
    *ran = false;
    if constexpr (R > 8) return SVH_OK;
    else {
        constexpr int WB = 16, NCW = 8, DP = 64 * R;
        const int n_strips = ceil_div((int64_t)a.W + a.H - 1, WB);
        const size_t edge_bytes = (size_t)n_strips * a.H * 3 * DP * sizeof(uint64_t);
        if (edge_bytes > ((size_t)16 << 30)) return SVH_OK;
        const size_t shmem = (size_t)((WB + 1) + (WB + 2) + WB + 6) * DP * sizeof(float);
        static bool attr_set[64] = {};
        if (!attr_set[ctx->device & 63]) {
            SVH_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&sgm_score_down_kernel<R, WB, NCW, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            SVH_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&sgm_score_down_kernel<R, WB, NCW, false>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            attr_set[ctx->device & 63] = true;
        }
        if (ctx->sgm_edges_bytes < edge_bytes || ctx->sgm_edges_tag == 0xFFFFFFFFu) {
            // (a larger buffer, or the tag counter about to wrap: start from zeroed granules, which no launch number matches)
            if (ctx->sgm_edges_bytes < edge_bytes) {
                SVH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
                if (ctx->sgm_edges) (void)hipFree(ctx->sgm_edges);
                ctx->sgm_edges = nullptr;
                ctx->sgm_edges_bytes = 0;
                if (hipMalloc(&ctx->sgm_edges, edge_bytes) != hipSuccess) {
                    (void)hipGetLastError();
                    return fail(ctx, SVH_ERR_OUT_OF_MEMORY, "sgm: %zu bytes of strip edge buffer", edge_bytes);
                }
                ctx->sgm_edges_bytes = edge_bytes;
            }
            SVH_HIP_CHECK(ctx, hipMemsetAsync(ctx->sgm_edges, 0, ctx->sgm_edges_bytes, ctx->stream));
            ctx->sgm_edges_tag = 0;
        }
        const uint32_t tag = ++ctx->sgm_edges_tag;
        uint64_t *edges = static_cast<uint64_t *>(ctx->sgm_edges);
        int *sync_words = scr.get_n<int>(2);
        if (!sync_words) return SVH_ERR_OUT_OF_MEMORY;
        SVH_HIP_CHECK(ctx, hipMemsetAsync(sync_words, 0, 2 * sizeof(int), ctx->stream));
        {
            LineSet ls{1, pass_lines(1, a.H, a.W), 0, 0, a.H, a.W};
            // (a line per image row: few waves, each far from filling its share of the memory pipe with a short batch)
            constexpr int B = (R <= 4) ? 4 : (R == 8 ? 2 : 1);
            if (vec && a.D == 64 * R)
                SVH_LAUNCH(ctx, "sgm_score_pass", (sgm_score_pass_kernel<R, B, false, true, false, true, true>), ceil_div(ls.n_lines, 4), 256, 0, cv, sgm, ls,
                           a.D, a.W, a.P1, a.P2, a.Pout, vec);
            else
                SVH_LAUNCH(ctx, "sgm_score_pass", (sgm_score_pass_kernel<R, B, false, true, false, true>), ceil_div(ls.n_lines, 4), 256, 0, cv, sgm, ls, a.D,
                           a.W, a.P1, a.P2, a.Pout, vec);
            SVH_CHECK_LAUNCH(ctx);
        }
        // diagnostic: SVH_SGM_DOWN_STAMPS=<file> makes the launch record when each strip passed each row's barrier and dumps it
        // (tools/sgm_down_stamps_report.py reads the file)
        const char *stamp_file = getenv("SVH_SGM_DOWN_STAMPS");
        unsigned long long *stamps = nullptr;
        const size_t n_stamps = (size_t)n_strips * (a.H + 1);
        if (stamp_file) {
            stamps = scr.get_n<unsigned long long>(n_stamps);
            if (!stamps) return SVH_ERR_OUT_OF_MEMORY;
            SVH_HIP_CHECK(ctx, hipMemsetAsync(stamps, 0, n_stamps * 8, ctx->stream));
        }
        if (vec && a.D == DP)
            SVH_LAUNCH(ctx, "sgm_score_down", (sgm_score_down_kernel<R, WB, NCW, true>), n_strips, (NCW + 2) * 64, shmem, cv, sgm, a.H, a.W, a.D, a.P1,
                       a.P2, a.Pout, edges, tag, sync_words, stamps);
        else
            SVH_LAUNCH(ctx, "sgm_score_down", (sgm_score_down_kernel<R, WB, NCW, false>), n_strips, (NCW + 2) * 64, shmem, cv, sgm, a.H, a.W, a.D, a.P1,
                       a.P2, a.Pout, edges, tag, sync_words, stamps);
        if (stamps) {
            std::vector<unsigned long long> h(n_stamps);
            SVH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            SVH_HIP_CHECK(ctx, hipMemcpy(h.data(), stamps, n_stamps * 8, hipMemcpyDeviceToHost));
            if (FILE *f = fopen(stamp_file, "wb")) {
                const int hdr[4] = {n_strips, a.H, a.W, WB};
                fwrite(hdr, sizeof(int), 4, f);
                fwrite(h.data(), 8, n_stamps, f);
                fclose(f);
            }
        }
        SVH_CHECK_LAUNCH(ctx);
        {
            LineSet ls{5, pass_lines(5, a.H, a.W), 0, 0, a.H, a.W};
            constexpr int B = (R <= 4) ? 4 : (R == 8 ? 2 : 1);
            if (vec && a.D == 64 * R)
                SVH_LAUNCH(ctx, "sgm_score_pass", (sgm_score_pass_kernel<R, B, false, true, false, false, true>), ceil_div(ls.n_lines, 4), 256, 0, cv, sgm, ls,
                           a.D, a.W, a.P1, a.P2, a.Pout, vec);
            else
                SVH_LAUNCH(ctx, "sgm_score_pass", (sgm_score_pass_kernel<R, B, false, true, false>), ceil_div(ls.n_lines, 4), 256, 0, cv, sgm, ls, a.D, a.W,
                           a.P1, a.P2, a.Pout, vec);
            SVH_CHECK_LAUNCH(ctx);
        }
        *ran = true;
        return SVH_OK;
    }
}

// The same three stages with the downward sweep as one launch per band of KB rows (sgm_score_band_kernel).
template <int R, int KB, int WB>
static int run_score_branch_bands(svh_context *ctx, Scratch &scr, const SgmArgs &a, const float *cv, float *sgm, bool vec, bool *ran) {
    *ran = false;
    if constexpr (R > 8) return SVH_OK;
    else {
        constexpr int NCW = WB, DP = 64 * R; // a wave per own pixel (8 waves for 16 pixels: 13.8 ms at C4, 16 waves: 13.0)
        const size_t shmem = (size_t)(WB + 2 * (WB + KB - 1)) * DP * sizeof(float);
        static bool attr_set[64] = {};
        if (!attr_set[ctx->device & 63]) {
            SVH_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&sgm_score_band_kernel<R, WB, KB, NCW, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            SVH_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&sgm_score_band_kernel<R, WB, KB, NCW, false>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            attr_set[ctx->device & 63] = true;
        }
        const size_t state_floats = (size_t)3 * a.W * DP;
        float *st[2] = {scr.get_n<float>(state_floats), scr.get_n<float>(state_floats)};
        if (!st[0] || !st[1]) return SVH_ERR_OUT_OF_MEMORY;
        constexpr int B = (R <= 4) ? 4 : (R == 8 ? 2 : 1);
        {
            LineSet ls{1, pass_lines(1, a.H, a.W), 0, 0, a.H, a.W};
            if (vec && a.D == 64 * R)
                SVH_LAUNCH(ctx, "sgm_score_pass", (sgm_score_pass_kernel<R, B, false, true, false, true, true>), ceil_div(ls.n_lines, 4), 256, 0, cv, sgm, ls,
                           a.D, a.W, a.P1, a.P2, a.Pout, vec);
            else
                SVH_LAUNCH(ctx, "sgm_score_pass", (sgm_score_pass_kernel<R, B, false, true, false, true>), ceil_div(ls.n_lines, 4), 256, 0, cv, sgm, ls, a.D,
                           a.W, a.P1, a.P2, a.Pout, vec);
            SVH_CHECK_LAUNCH(ctx);
        }
        const int strips = ceil_div(a.W, WB);
        {
            ProfScope prof(ctx, "sgm_score_bands"); // (one bracket around all the band launches)
            for (int r0 = 0, b = 0; r0 < a.H; r0 += KB, b++) {
                const int rows = std::min(KB, a.H - r0);
                if (vec && a.D == DP)
                    hipLaunchKernelGGL((sgm_score_band_kernel<R, WB, KB, NCW, true>), strips, NCW * 64, shmem, ctx->stream, cv, sgm, a.H, a.W, a.D, a.P1, a.P2,
                                       a.Pout, r0, rows, st[b & 1], st[(b + 1) & 1]);
                else
                    hipLaunchKernelGGL((sgm_score_band_kernel<R, WB, KB, NCW, false>), strips, NCW * 64, shmem, ctx->stream, cv, sgm, a.H, a.W, a.D, a.P1, a.P2,
                                       a.Pout, r0, rows, st[b & 1], st[(b + 1) & 1]);
            }
        }
        SVH_CHECK_LAUNCH(ctx);
        {
            LineSet ls{5, pass_lines(5, a.H, a.W), 0, 0, a.H, a.W};
            if (vec && a.D == 64 * R)
                SVH_LAUNCH(ctx, "sgm_score_pass", (sgm_score_pass_kernel<R, B, false, true, false, false, true>), ceil_div(ls.n_lines, 4), 256, 0, cv, sgm, ls,
                           a.D, a.W, a.P1, a.P2, a.Pout, vec);
            else
                SVH_LAUNCH(ctx, "sgm_score_pass", (sgm_score_pass_kernel<R, B, false, true, false>), ceil_div(ls.n_lines, 4), 256, 0, cv, sgm, ls, a.D, a.W,
                           a.P1, a.P2, a.Pout, vec);
            SVH_CHECK_LAUNCH(ctx);
        }
        *ran = true;
        return SVH_OK;
    }
}

template <int R>
static int run_score_branch(svh_context *ctx, Scratch &scr, const SgmArgs &a, const float *cv, float *sgm, bool textbook = false) {
    const int Hp = a.H - a.top - a.bottom, Wp = a.W - a.left - a.right;
    const int n_pass = textbook ? (a.n_dir >= 8 ? 8 : 4) : (a.n_dir >= 8 ? 6 : 2);
    const int pass0 = textbook ? 6 : 0;
    const bool neg = textbook && a.strategy == SVH_COST;
    constexpr int B = (R <= 4) ? 4 : (R == 8 ? 2 : 1);
    const bool vec = aligned16(cv) && aligned16(sgm) && a.D % 4 == 0;
    const bool whole = a.top == 0 && a.left == 0 && a.bottom == 0 && a.right == 0;
    const bool far_global = a.P2 >= a.P1 && a.P1 >= 0.0f; // also false for NaN penalties
    if (!whole || Hp <= 0 || Wp <= 0) {
        // pixels outside the margin box keep sgm = cv (sgm.h:371-377)
        ProfScope prof(ctx, "sgm_copy");
        SVH_HIP_CHECK(ctx, hipMemcpyAsync(sgm, cv, (size_t)a.H * a.W * a.D * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (Hp <= 0 || Wp <= 0) return SVH_OK;
    if (ctx->sgm_score_fused && !textbook && n_pass == 6 && whole && far_global) { // 1 (3: 16-column strips whatever the width): bands of rows per launch, 2: strips handed over in one launch
        bool ran = false;
        if (ctx->sgm_score_fused == 2) SVH_TRY(run_score_branch_fused<R>(ctx, scr, a, cv, sgm, vec, &ran));
        else { // 16-column strips, 16 rows per band (8 / 12 / 16 rows: the same time at C4, 24: + 8 %); 8 x 8 where 16 columns would leave CUs
               // without a strip (1080p: 8 / 16 / 32 rows per band 2.56 / 2.78 / 4.05 ms)
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
            if (ctx->sgm_score_fused != 3 && ceil_div(a.W, 16) * 4 < cus * 3) SVH_TRY((run_score_branch_bands<R, 8, 8>(ctx, scr, a, cv, sgm, vec, &ran)));
            else SVH_TRY((run_score_branch_bands<R, 16, 16>(ctx, scr, a, cv, sgm, vec, &ran)));
        }
        if (ran) return SVH_OK;
    }
    for (int q = 0; q < n_pass; q++) {
        LineSet ls{pass0 + q, pass_lines(pass0 + q, Hp, Wp), a.top, a.left, Hp, Wp};
        int grid = ceil_div(ls.n_lines, 4);
        const bool first = q == 0 && whole;
#define SVH_SCORE(FIRSTV, FARV, NEGV)                                                                                                    \
    SVH_LAUNCH(ctx, NEGV ? "sgm_textbook_pass" : "sgm_score_pass", (sgm_score_pass_kernel<R, B, FIRSTV, FARV, NEGV>), grid, 256, 0, cv, sgm, ls, a.D, a.W, \
               a.P1, a.P2, a.Pout, vec)
        if (neg) {
            if (far_global) {
                if (first) SVH_SCORE(true, true, true); else SVH_SCORE(false, true, true);
            } else {
                if (first) SVH_SCORE(true, false, true); else SVH_SCORE(false, false, true);
            }
        } else if (far_global && vec && a.D == 64 * R) { // the usual case: exact prefetch form
            if (first)
                SVH_LAUNCH(ctx, "sgm_score_pass", (sgm_score_pass_kernel<R, B, true, true, false, false, true>), grid, 256, 0, cv, sgm, ls, a.D, a.W, a.P1, a.P2,
                           a.Pout, vec);
            else
                SVH_LAUNCH(ctx, "sgm_score_pass", (sgm_score_pass_kernel<R, B, false, true, false, false, true>), grid, 256, 0, cv, sgm, ls, a.D, a.W, a.P1, a.P2,
                           a.Pout, vec);
        } else if (far_global) {
            if (first) SVH_SCORE(true, true, false); else SVH_SCORE(false, true, false);
        } else {
            if (first) SVH_SCORE(true, false, false); else SVH_SCORE(false, false, false);
        }
#undef SVH_SCORE
        SVH_CHECK_LAUNCH(ctx);
    }
    return SVH_OK;
}

int dev_sgm_score_branch(svh_context *ctx, Scratch &scr, const SgmArgs &a, const float *cv, float *out_sgm, bool textbook) {
    if ((int64_t)a.H * a.W * a.D == 0) return SVH_OK;
    switch (pick_R(a.D)) {
    case 1: return run_score_branch<1>(ctx, scr, a, cv, out_sgm, textbook);
    case 2: return run_score_branch<2>(ctx, scr, a, cv, out_sgm, textbook);
    case 4: return run_score_branch<4>(ctx, scr, a, cv, out_sgm, textbook);
    case 8: return run_score_branch<8>(ctx, scr, a, cv, out_sgm, textbook);
    case 16: return run_score_branch<16>(ctx, scr, a, cv, out_sgm, textbook);
    default: return fail(ctx, SVH_ERR_UNSUPPORTED, "SGM supports at most 1024 disparities (got %d)", a.D);
    }
}

} // namespace svh

using namespace svh;

extern "C" int svh_sgm_cost_volume(svh_context *ctx, int n_directions, int strategy, const svh_array *cv, float P1, float P2,
                                   const int32_t margins[4], float Pout, svh_array *out) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, cv, "cv", SVH_F32, 3, 3));
    SVH_TRY(validate(ctx, out, "out", SVH_F32, 3, 3));
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
    SVH_TRY(stage_in(ctx, scr, *cv, &dcv));
    SVH_TRY(stage_out(ctx, scr, *out, &os));
    if (strategy == SVH_COST) {
        CostSource cs;
        cs.cv = (const float *)dcv;
        SVH_TRY(dev_sgm_cost_branch(ctx, scr, a, cs, (float *)os.dptr, WinnerOut()));
    } else {
        if (os.dptr == dcv) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "cv and out must not alias");
        SVH_TRY(dev_sgm_score_branch(ctx, scr, a, (const float *)dcv, (float *)os.dptr, false));
    }
    return finish_out(ctx, os);
}

// "Textbook" semi-global matching: NOT the reference's behaviour -- what correlation/sgm.h
// evidently intends.  All 4 / 8 directions, every line of the margin box once (the reference skips three directions and half of
// two more: finding F5); Cost strategy a(nd) = min{prev[nd], prev[nd +- 1] + P1, min_far prev + P2} (the reference assigns the
// pixel's own cost there: finding F4); Score strategy as the reference's Score branch.  Defined by oracle/stevi_oracle.c
// so_sgm_textbook; same kernels as the Score branch (the Cost form runs on negated costs).
extern "C" int svh_sgm_cost_volume_textbook(svh_context *ctx, int n_directions, int strategy, const svh_array *cv, float P1, float P2,
                                            const int32_t margins[4], float Pout, svh_array *out) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, cv, "cv", SVH_F32, 3, 3));
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
    SVH_TRY(stage_in(ctx, scr, *cv, &dcv));
    SVH_TRY(stage_out(ctx, scr, *out, &os));
    if (os.dptr == dcv) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "cv and out must not alias");
    SVH_TRY(dev_sgm_score_branch(ctx, scr, a, (const float *)dcv, (float *)os.dptr, true));
    return finish_out(ctx, os);
}

// Line geometry of the reference's SGM passes and the wave-level primitives the SGM kernels share (svh_sgm.hip: the per-pass
// kernels of both branches; svh_sgm_sweep.hip: the fused downward sweeps of the Score branch).
#pragma once

#include "svh_internal.h"

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

// Wave extremum, result in every lane's return value (read from lane 63): row_shr 1,2,4,8, row_bcast:15, row_bcast:31 as hand-written DPP steps, one
// instruction each -- the compiler's form of `v = fminf(v, update_dpp(inf, v, ...))` is a v_mov of the fill value, the v_mov_dpp, a canonicalising
// v_max and the v_min, and the kernels that reduce per pixel are bound by instruction issue.  A lane whose DPP source does not exist is not
// written and keeps its value (what the +-inf fill gave); a quiet NaN operand yields the other operand, as fminf / fmaxf do.  The operand
// must not be a signalling NaN (every caller passes the result of an fminf / fmaxf chain, which is canonical).
template <bool MIN> __device__ __forceinline__ float wave_extremum_dpp(float v) {
#define SVH_EXT_DPP(CTRL)                                                                  \
    if constexpr (MIN) asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 " CTRL : "+v"(v));     \
    else asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " CTRL : "+v"(v))
    SVH_EXT_DPP("row_shr:1 row_mask:0xf bank_mask:0xf");
    SVH_EXT_DPP("row_shr:2 row_mask:0xf bank_mask:0xf");
    SVH_EXT_DPP("row_shr:4 row_mask:0xf bank_mask:0xf");
    SVH_EXT_DPP("row_shr:8 row_mask:0xf bank_mask:0xf");
    SVH_EXT_DPP("row_bcast:15 row_mask:0xa bank_mask:0xf");
    SVH_EXT_DPP("row_bcast:31 row_mask:0xc bank_mask:0xf");
#undef SVH_EXT_DPP
    asm("s_nop 0" : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_min(float v) { return wave_extremum_dpp<true>(v); }
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
    // the same by flat pixel index (row * W + column)
    template <int R> __device__ __forceinline__ void load_flat(int64_t px, int lane, float (&c)[R]) const {
        const float *p = cv + px * D + lane * R;
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

// cross-lane primitives on the DPP path (no LDS round trip): whole-wave shifts by one lane and a max reduction
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
__device__ __forceinline__ float wave_max_dpp(float v) { return wave_extremum_dpp<false>(v); }

// R consecutive floats of a lane as 16 / 8 / 4-byte accesses (any address space).  R = 3, 5, 6, 7 (192, 320, 384, 448 disparities: a lane's
// run starts at a multiple of 4 R bytes, not of 16): pieces of four / two floats that claim only the 4-byte alignment every float has --
// global memory takes a dwordx4 at any dword (as the cost kernels' CostPiece4 stores), LDS accesses are split as the alignment demands.
struct __attribute__((packed, aligned(4))) LanePiece4 {
    float x, y, z, w;
};
struct __attribute__((packed, aligned(4))) LanePiece2 {
    float x, y;
};
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
        constexpr int R4 = R / 4 * 4;
#pragma unroll
        for (int q = 0; q < R4; q += 4) {
            const LanePiece4 x = *reinterpret_cast<const LanePiece4 *>(p + q);
            v[q] = x.x; v[q + 1] = x.y; v[q + 2] = x.z; v[q + 3] = x.w;
        }
        if constexpr (R - R4 >= 2) {
            const LanePiece2 x = *reinterpret_cast<const LanePiece2 *>(p + R4);
            v[R4] = x.x; v[R4 + 1] = x.y;
        }
        if constexpr ((R - R4) & 1) v[R - 1] = p[R - 1];
    }
}
template <int R> __device__ __forceinline__ void lds_put(float *p, const float (&v)[R]) {
    if constexpr (R % 4 == 0) {
#pragma unroll
        for (int q = 0; q < R / 4; q++) *reinterpret_cast<float4 *>(p + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else if constexpr (R == 2) {
        *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
    } else {
        constexpr int R4 = R / 4 * 4;
#pragma unroll
        for (int q = 0; q < R4; q += 4) *reinterpret_cast<LanePiece4 *>(p + q) = LanePiece4{v[q], v[q + 1], v[q + 2], v[q + 3]};
        if constexpr (R - R4 >= 2) *reinterpret_cast<LanePiece2 *>(p + R4) = LanePiece2{v[R4], v[R4 + 1]};
        if constexpr ((R - R4) & 1) p[R - 1] = v[R - 1];
    }
}

// One step of the Score branch's recurrence (sgm.h:218-255) for the R disparities of this lane when P2 >= P1 >= 0 (the far term is
// max_p - P2: see svh_sgm.hip), written without per-lane branches.  POUT: some disparity of this pixel looks past the image border
// (wave-uniform; false for most pixels, which skip the term).  TAIL: D < 64 R, the lanes past D are masked.
// LEAN: every cost is finite and of magnitude about one and Pout is finite (SgmArgs::costs_all_finite): then every line state is finite
// and the isfinite filters of sgm.h:224, :241, :251 change nothing -- left out, a sixth of the step's instructions.
template <int R, bool POUT, bool TAIL, bool LEAN = false>
__device__ __forceinline__ void score_step1_far_global(const float (&prev)[R], const float (&c)[R], int jj, int lane, int D, int W, float P1,
                                                       float P2, float Pout, float (&act)[R]) {
    float pf[R];
    float A = -INFINITY;
#pragma unroll
    for (int k = 0; k < R; k++) {
        const bool keep = TAIL ? ((lane * R + k < D) & (LEAN || finite_f(prev[k]))) : (LEAN || finite_f(prev[k]));
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
        act[k] = (LEAN || (maxp_fin & finite_f(a))) ? moved : c[k];
    }
}

// ---- the winner of a pixel from the wave that holds its D aggregated values (R per lane) -------------------------------------------
// extractSelectedIndex (correlation_base.h:427-464): the sequential scan keeps the extremum, ties go to the LARGER index, a NaN never
// wins unless it sits at index 0 (then the index stays 0).
// One 16-byte RECORD per pixel -- (tap(-1), tap(0), tap(+1), index as bits): the three truncatedCostVolume<Same>(S, idx, h_r,
// v_r, 1) values around the winner (correlation_base.h:579-613: NaN where the window of the shifted disparity leaves the image) and the
// winner -- written from inside a streaming kernel whose prefetch depends on the compiler's exact count of memory operations in
// flight (conditional accesses end in s_waitcnt vmcnt(0): DESIGN.md 4.2a).  A first form with branches and `if (lane == 0)` stores
// per output array put 32 such waits into the sweep (4 without records) and made it 30 % longer; this one is straight-line code with a
// single 16-byte store.  D = 64 R (the vector form).  finish_records_kernel (svh_select_refine.hip) turns records into index /
// disparity / refined maps.
// Instruction budget: the host kernels are bound by instruction issue.  The wave extremum runs as hand-written DPP steps (one
// instruction per step: the compiler's form is v_mov, v_mov_dpp, a canonicalising v_max and the v_max); NaN never enters it (v_max /
// v_min return the other operand), a lane finds the largest k whose value EQUALS the extremum (a NaN equals nothing; +-inf do), the
// highest such lane holds the winner, and every lane builds the record of ITS OWN candidate from its registers and its two
// neighbours' edge values (two DPP moves): no cross-lane gather, almost no scalar state.  The record is stored by the winning lane.
// Dv <= 64 R: rows padded to a whole number of lanes (dev_sgm_score_branch; Score: the pads hold -inf and stay -inf along every line):
// only the first Dv values exist for the caller.
template <bool COST, int R>
__device__ __forceinline__ void wave_emit_record(const float (&s)[R], int lane, int i, int j, int H, int W, float *__restrict__ rec, int h_r, int v_r,
                                                 int Dv = 64 * R) {
    constexpr int D = 64 * R;
    float A = s[0];
#pragma unroll
    for (int k = 1; k < R; k++) A = COST ? fminf(A, s[k]) : fmaxf(A, s[k]);
    if constexpr (COST) {
#define SVH_EXT_DPP(CTRL) asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 " CTRL : "+v"(A))
        SVH_EXT_DPP("row_shr:1 row_mask:0xf bank_mask:0xf");
        SVH_EXT_DPP("row_shr:2 row_mask:0xf bank_mask:0xf");
        SVH_EXT_DPP("row_shr:4 row_mask:0xf bank_mask:0xf");
        SVH_EXT_DPP("row_shr:8 row_mask:0xf bank_mask:0xf");
        SVH_EXT_DPP("row_bcast:15 row_mask:0xa bank_mask:0xf");
        SVH_EXT_DPP("row_bcast:31 row_mask:0xc bank_mask:0xf");
#undef SVH_EXT_DPP
    } else {
#define SVH_EXT_DPP(CTRL) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " CTRL : "+v"(A))
        SVH_EXT_DPP("row_shr:1 row_mask:0xf bank_mask:0xf");
        SVH_EXT_DPP("row_shr:2 row_mask:0xf bank_mask:0xf");
        SVH_EXT_DPP("row_shr:4 row_mask:0xf bank_mask:0xf");
        SVH_EXT_DPP("row_shr:8 row_mask:0xf bank_mask:0xf");
        SVH_EXT_DPP("row_bcast:15 row_mask:0xa bank_mask:0xf");
        SVH_EXT_DPP("row_bcast:31 row_mask:0xc bank_mask:0xf");
#undef SVH_EXT_DPP
    }
    asm("s_nop 0" : "+v"(A));
    const float M = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, A), 63)); // NaN iff every value is NaN
    int kb = -1;
#pragma unroll
    for (int k = 0; k < R; k++) kb = (s[k] == M) ? k : kb; // ties: the larger index (correlation_base.h:441-455)
    if (Dv < D && M == (COST ? INFINITY : -INFINITY)) { // (wave uniform, no memory operation inside) a pad equals the extremum only when every real value is infinite or NaN
        kb = -1;
#pragma unroll
        for (int k = 0; k < R; k++) kb = (s[k] == M && lane * R + k < Dv) ? k : kb;
    }
    const unsigned long long holders = __builtin_amdgcn_ballot_w64(kb >= 0);
    const unsigned long long nan_at_0 = __builtin_amdgcn_ballot_w64(isnan(s[0])); // bit 0: the value at disparity 0
    const bool none = holders == 0ull || (nan_at_0 & 1ull);                         // -> index 0
    const int writer = none ? 0 : 63 - __builtin_clzll(holders | 1ull);
    const int c = none ? 0 : kb; // (lanes without a candidate build a record nobody stores)
    const float prevL = lane_shift_up(s[R - 1], 0.0f), prevR = lane_shift_down(s[0], 0.0f);
    float t0 = s[0], tm1 = prevL, tp1 = R > 1 ? s[R > 1 ? 1 : 0] : prevR;
#pragma unroll
    for (int k = 1; k < R; k++) {
        const bool here = c == k;
        t0 = here ? s[k] : t0;
        tm1 = here ? s[k - 1] : tm1;
        tp1 = here ? (k + 1 < R ? s[k + 1 < R ? k + 1 : k] : prevR) : tp1;
    }
    // truncatedCostVolume<Same>, radius 1 (correlation_base.h:601-613): NaN where the pixel's window, or the window of the shifted
    // disparity, leaves the image, or the disparity its range
    const bool px_bad = j < h_r || i < v_r || i + v_r >= H;
    const int limit = px_bad ? -2 : W - h_r - j; // valid: 0 <= pd < D and pd < limit
    const int d = lane * R + c;
    const float nan = __uint_as_float(0x7FC00000u);
    float4 r;
    r.x = (d == 0 || d - 1 >= limit) ? nan : tm1;
    r.y = (d >= limit) ? nan : t0;
    r.z = (d + 1 >= Dv || d + 1 >= limit) ? nan : tp1;
    r.w = __int_as_float(d);
    if (lane == writer) *reinterpret_cast<float4 *>(rec + (unsigned)(i * W + j) * 4u) = r;
}

// ---- reductions inside the lanes of one pixel (several pixels per wave: svh_sgm.hip's packed apply kernel, svh_select_refine.hip) ----------
template <bool MIN> __device__ __forceinline__ float row_allreduce_f32(float v) { // every lane: the extremum of its row of 16
#define SVH_ROR_DPP(CTRL)                                                                                                   \
    if constexpr (MIN) asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(v));          \
    else asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(v))
    SVH_ROR_DPP("row_ror:1");
    SVH_ROR_DPP("row_ror:2");
    SVH_ROR_DPP("row_ror:4");
    SVH_ROR_DPP("row_ror:8");
#undef SVH_ROR_DPP
    asm("s_nop 0" : "+v"(v));
    return v;
}
__device__ __forceinline__ int row_allreduce_max_i32(int v) {
#define SVH_ROR_DPP(CTRL) asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(v))
    SVH_ROR_DPP("row_ror:1");
    SVH_ROR_DPP("row_ror:2");
    SVH_ROR_DPP("row_ror:4");
    SVH_ROR_DPP("row_ror:8");
#undef SVH_ROR_DPP
    asm("s_nop 0" : "+v"(v));
    return v;
}
// all-reduce over the LPP = 16 / 32 / 64 lanes of a pixel: after v_permlane16_swap of a register with itself every lane of a row pair holds
// both rows' values (one in each result), after v_permlane32_swap every lane both halves' (through named words: see the note on
// __builtin_bit_cast of vector elements in svh_sgm.hip)
template <int LPP, bool MIN> __device__ __forceinline__ float pixel_allreduce_f32(float v) {
    v = row_allreduce_f32<MIN>(v);
    if constexpr (LPP >= 32) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        const uint32_t a = r[0], b = r[1];
        v = MIN ? fminf(__uint_as_float(a), __uint_as_float(b)) : fmaxf(__uint_as_float(a), __uint_as_float(b));
    }
    if constexpr (LPP == 64) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        const uint32_t a = r[0], b = r[1];
        v = MIN ? fminf(__uint_as_float(a), __uint_as_float(b)) : fmaxf(__uint_as_float(a), __uint_as_float(b));
    }
    return v;
}
template <int LPP> __device__ __forceinline__ float pixel_allreduce_min(float v) { return pixel_allreduce_f32<LPP, true>(v); }
template <int LPP> __device__ __forceinline__ int pixel_allreduce_max(int v) {
    v = row_allreduce_max_i32(v);
    if constexpr (LPP >= 32) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        const uint32_t a = r[0], b = r[1];
        v = max((int)a, (int)b);
    }
    if constexpr (LPP == 64) {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        const uint32_t a = r[0], b = r[1];
        v = max((int)a, (int)b);
    }
    return v;
}

// ---- host helpers
inline int pass_lines(int q, int Hp, int Wp) {
    if (q >= 10) return Hp + Wp - 1;
    if (q >= 6) return q < 8 ? Wp : Hp;
    return (q == 0 || q == 3 || q == 4) ? Wp : Hp;
}

inline int pick_R(int D) {
    int R = 1;
    while (64 * R < D) R <<= 1;
    return R;
}
// Score branch: every count up to eight per lane (192, 320, 384, 448 disparities run the vector forms at full width instead of the masked
// forms of the next power of two: the reference's own benchmark rows use 160 and 320); powers of two beyond
inline int pick_R_score(int D) { return D <= 512 ? (D + 63) / 64 : pick_R(D); }

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

} // namespace svh

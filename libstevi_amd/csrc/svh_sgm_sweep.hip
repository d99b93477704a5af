// SGM Score branch (correlation/sgm.h:218-255, :329-389): the four downward passes of the reference's 8-direction aggregation as ONE
// sweep of the volume instead of one read-modify-write sweep per pass, bit-identical to the per-pass kernels of svh_sgm.hip:
//   sgm_score_band_kernel   a launch per band of rows, every block recomputing the lines that enter its strip of columns.
// (Round 2 also carried a one-launch form that handed strips from block to block through tagged granules in global memory; it was
// bit-identical too and no faster than a launch per pass -- the hand-off chain is as long as the number of strips -- and has been
// removed; DESIGN.md section 4.2a keeps its measurements.)
#include "svh_sgm_lines.h"

#include <type_traits>
#include <vector>

namespace svh {

// ---- Score branch: the four downward passes in one sweep -----------------------------------------------------
// Passes 0, 2, 3 and 4 all step one row down: a pixel (i, j) takes its line state from (i-1, j), (i-1, j-1) and (i-1, j+1).  One
// sweep over the rows can therefore carry the three states, add the contributions in the reference's order
//     S = (((c + d0) + d1) [+ d2 if i >= j] [+ d3 if j >= i]) [+ d4 if i + j < W]        (d2 = d3: the same diagonal lines)
// in registers and touch the volume once (read c and d1, write S: 12 B/voxel) where the pass-per-launch form reads and writes S
// once per pass.  d1, the Left2Right contribution, is written beforehand by the DELTA form of the kernel above.
//
// Three line states of one pixel at a time, written without per-lane branches: the sweep is bound by instruction issue, not by
// memory.  The three wave maxima run interleaved in hand-written DPP steps (v_max_f32_dpp with itself: one instruction per step
// and state, and two other instructions between a register's write and its next DPP read, which is the hazard distance; the
// compiler's form is v_mov, v_mov_dpp, a canonicalising v_max and the v_max).  The values are finite or -inf here, never NaN.
// POUT: some disparity of this pixel looks past the image border (wave-uniform; false for most pixels, which skip the term).
// TAIL: D < 64 R, the lanes past D are masked.
#define SVH_MAX3_DPP(CTRL)                                                                                                        \
    asm("v_max_f32_dpp %0, %0, %0 " CTRL "\n\tv_max_f32_dpp %1, %1, %1 " CTRL "\n\tv_max_f32_dpp %2, %2, %2 " CTRL : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]))
template <int R, bool POUT, bool TAIL, bool LEAN = false>
__device__ __forceinline__ void score_step3_far_global(const float (&prev)[3][R], const float (&c)[R], int jj, int lane, int D, int W, float P1,
                                                       float P2, float Pout, float (&act)[3][R]) {
    float pf[3][R], A[3];
#pragma unroll
    for (int s = 0; s < 3; s++) {
        A[s] = -INFINITY;
#pragma unroll
        for (int k = 0; k < R; k++) {
            const bool keep = TAIL ? ((lane * R + k < D) & (LEAN || finite_f(prev[s][k]))) : (LEAN || finite_f(prev[s][k])); // isfinite filters of sgm.h:224, :241
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
            act[s][k] = (LEAN || (maxp_fin & finite_f(a))) ? moved : c[k];
        }
    }
}
#undef SVH_MAX3_DPP
// ---- Score branch: the four downward passes, a band of rows per launch -----------------------------------------
// The same fusion without any hand-off between blocks: a launch covers KB image rows, a block a strip of WB columns, and what a
// block would need from its neighbours during the band it computes itself -- the diagonal lines that enter its strip from the left
// (a triangle of at most KB - 1 columns, one column narrower every row) and the anti-diagonal lines that enter from the right.
// Line states cross from band to band through two global arrays (the previous band's last row, read; this band's last row,
// written: 3 x W vectors of D floats each), the launch boundary is the only synchronisation.  Per band and block this costs
// (KB - 1) KB / 2 extra single-pass pixels on either side against 3 WB KB pass-pixels of its own (+ 31 % at WB = KB = 16) and the
// carried states (+ 10 % of the band's bytes); nothing spins.
// FIN (VEC only): the sweep is the last writer of the pixels DownLeft2UpRight does not visit (i + j >= H).  It writes a winner record
// for EVERY pixel (unconditionally: the later pass overwrites the records of the pixels it visits) and, when nobody wants the volume
// (fin.store_all false), stores the aggregated costs only where that pass will read them.
// LEAN (with FIN, VEC): all costs finite (SgmArgs::costs_all_finite), the finiteness filters of the steps are left out.
template <int R, int WB, int KB, int NCW, bool VEC, bool FIN = false, bool LEAN = false>
__global__ void __launch_bounds__(NCW * 64) sgm_score_band_kernel(const float *__restrict__ cv, float *sgm, int H, int W, int D, float P1, float P2,
                                                                  float Pout, int r0, int rows, const float *__restrict__ st_in,
                                                                  float *__restrict__ st_out, ScoreFinish fin) {
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
        // A cell's column does not change from row to row: its offset inside a row of the volume (32 bits: W D < 2^31, checked by the host) is
        // computed once, and a row costs one 64-bit scalar product for the row's base.  (Per load the sweep spent some fourteen scalar
        // instructions on ((i W + j) D + lane R) in 64 bits: 70 of the 116 scalar instructions a wave issued per row, in a kernel bound by
        // instruction issue -- profiles/r04d_sq_counters_c4.jsonl.)
        unsigned own_off[CPW], halo_off[HS];
#pragma unroll
        for (int q = 0; q < CPW; q++) own_off[q] = (unsigned)((FULL ? j0 + wave + q * NCW : min(j0 + wave + q * NCW, W - 1)) * D + lane * R);
#pragma unroll
        for (int q = 0; q < HS; q++) {
            int j, slot;
            bool left;
            (void)halo_of(wave + q * NCW, 0, j, slot, left); // (the column of a halo slot does not depend on the row)
            halo_off[q] = (unsigned)(min(max(j, 0), W - 1) * D + lane * R);
        }
        auto load_at = [&](const float *row, unsigned off, float (&v)[R]) {
            if constexpr (VEC) lds_get<R>(row + off, v);
            else {
#pragma unroll
                for (int k = 0; k < R; k++) v[k] = (lane * R + k < D) ? row[off + k] : 0.0f;
            }
        };
        auto load_row = [&](float (&c)[CPW][R], float (&d1)[CPW][R], float (&hc)[HS][R], int r) {
            r = min(r, rows - 1); // (past the band: the last row again, unused)
            const int64_t row_off = (int64_t)(r0 + r) * W * D;
            const float *crow = cv + row_off, *srow = sgm + row_off;
#pragma unroll
            for (int q = 0; q < CPW; q++) {
                load_at(crow, own_off[q], c[q]);
                load_at(srow, own_off[q], d1[q]);
            }
#pragma unroll
            for (int q = 0; q < HS; q++) load_at(crow, halo_off[q], hc[q]); // (every slot loads, needed or not: the count of loads in flight stays exact)
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
                score_step3_far_global<R, POUT, !VEC, LEAN>(prev, c[q], j, lane, D, W, P1, P2, Pout, act);
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
                float *o = sgm + (int64_t)i * W * D + own_off[q]; // (own_off: the cell's column, see load_row)
                if constexpr (FIN) {
                    wave_emit_record<false, R>(outv, lane, i, j, H, W, fin.records, fin.taps_h_r, fin.taps_v_r, fin.d_valid > 0 ? fin.d_valid : 64 * R);
                    if (fin.store_all || i + j < H) lds_put<R>(o, outv); // (wave-uniform) DownLeft2UpRight visits i + j < H
                } else if constexpr (VEC) lds_put<R>(o, outv);
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
                score_step1_far_global<R, POUT, !VEC, LEAN>(prev, hc[q], j, lane, D, W, P1, P2, Pout, act);
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

// The same three stages with the downward sweep as one launch per band of KB rows (sgm_score_band_kernel).
template <int R, int KB, int WB>
static int run_score_branch_bands(svh_context *ctx, Scratch &scr, const SgmArgs &a, const float *cv, float *sgm, bool vec, bool *ran, ScoreFinish *finish) {
    *ran = false;
    if constexpr (R > 8) return SVH_OK;
    else if ((int64_t)a.W * a.D >= (1ll << 31)) return SVH_OK; // (a cell's offset inside a row of the volume is kept in 32 bits: a launch per pass then)
    else {
        constexpr int NCW = WB, DP = 64 * R; // a wave per own pixel (8 waves for 16 pixels: 13.8 ms at C4, 16 waves: 13.0)
        const size_t shmem = (size_t)(WB + 2 * (WB + KB - 1)) * DP * sizeof(float);
        static int attr_set[64] = {}; // (per instantiation and device; concurrent per-thread contexts: atomics)
        if (!__atomic_load_n(&attr_set[ctx->device & 63], __ATOMIC_ACQUIRE)) {
            SVH_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&sgm_score_band_kernel<R, WB, KB, NCW, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            SVH_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&sgm_score_band_kernel<R, WB, KB, NCW, true, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            SVH_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&sgm_score_band_kernel<R, WB, KB, NCW, true, true, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            SVH_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&sgm_score_band_kernel<R, WB, KB, NCW, false>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            __atomic_store_n(&attr_set[ctx->device & 63], 1, __ATOMIC_RELEASE);
        }
        const size_t state_floats = (size_t)3 * a.W * DP;
        float *st[2] = {scr.get_n<float>(state_floats), scr.get_n<float>(state_floats)};
        if (!st[0] || !st[1]) return SVH_ERR_OUT_OF_MEMORY;
        // the winner rides on the last writer of every pixel: the vector form only, and not for cross-shard keys
        const bool fused_finish = finish && finish->records && vec && a.D == DP && (int64_t)a.H * a.W < (1ll << 29);
        const ScoreFinish none{};
        const bool lean = fused_finish && a.costs_all_finite && std::isfinite(a.Pout); // (the fused call's volumes: SgmArgs::costs_all_finite)
        SVH_TRY(dev_sgm_score_line_pass(ctx, a, cv, sgm, 1, true)); // Left2Right: its contribution alone
        const int strips = ceil_div(a.W, WB);
        {
            ProfScope prof(ctx, "sgm_score_bands"); // (one bracket around all the band launches)
            for (int r0 = 0, b = 0; r0 < a.H; r0 += KB, b++) {
                const int rows = std::min(KB, a.H - r0);
                if (fused_finish && lean)
                    hipLaunchKernelGGL((sgm_score_band_kernel<R, WB, KB, NCW, true, true, true>), strips, NCW * 64, shmem, ctx->stream, cv, sgm, a.H, a.W, a.D,
                                       a.P1, a.P2, a.Pout, r0, rows, st[b & 1], st[(b + 1) & 1], *finish);
                else if (fused_finish)
                    hipLaunchKernelGGL((sgm_score_band_kernel<R, WB, KB, NCW, true, true>), strips, NCW * 64, shmem, ctx->stream, cv, sgm, a.H, a.W, a.D, a.P1,
                                       a.P2, a.Pout, r0, rows, st[b & 1], st[(b + 1) & 1], *finish);
                else if (vec && a.D == DP)
                    hipLaunchKernelGGL((sgm_score_band_kernel<R, WB, KB, NCW, true>), strips, NCW * 64, shmem, ctx->stream, cv, sgm, a.H, a.W, a.D, a.P1, a.P2,
                                       a.Pout, r0, rows, st[b & 1], st[(b + 1) & 1], none);
                else
                    hipLaunchKernelGGL((sgm_score_band_kernel<R, WB, KB, NCW, false>), strips, NCW * 64, shmem, ctx->stream, cv, sgm, a.H, a.W, a.D, a.P1, a.P2,
                                       a.Pout, r0, rows, st[b & 1], st[(b + 1) & 1], none);
            }
        }
        SVH_CHECK_LAUNCH(ctx);
        SVH_TRY(dev_sgm_score_line_pass(ctx, a, cv, sgm, 5, false, fused_finish ? finish : nullptr)); // DownLeft2UpRight: read-modify-write
        if (fused_finish) finish->done = true;
        *ran = true;
        return SVH_OK;
    }
}

int dev_sgm_score_sweep(svh_context *ctx, Scratch &scr, const SgmArgs &a, const float *cv, float *sgm, bool vec, int form, bool *ran, ScoreFinish *finish) {
    *ran = false;
    const auto run = [&](auto r_tag) -> int {
        constexpr int R = decltype(r_tag)::value;
        // 16-column strips, 16 rows per band (8 / 12 / 16 rows: the same time at C4, 24: + 8 %); 8 x 8 where 16 columns would leave CUs
        // without a strip (1080p: 8 / 16 / 32 rows per band 2.56 / 2.78 / 4.05 ms)
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        if (form != 3 && ceil_div(a.W, 16) * 4 < cus * 3) return run_score_branch_bands<R, 8, 8>(ctx, scr, a, cv, sgm, vec, ran, finish);
        return run_score_branch_bands<R, 16, 16>(ctx, scr, a, cv, sgm, vec, ran, finish);
    };
    switch (pick_R_score(a.D)) {
    case 1: return run(std::integral_constant<int, 1>{});
    case 2: return run(std::integral_constant<int, 2>{});
    case 3: return run(std::integral_constant<int, 3>{});
    case 4: return run(std::integral_constant<int, 4>{});
    case 5: return run(std::integral_constant<int, 5>{});
    case 6: return run(std::integral_constant<int, 6>{});
    case 7: return run(std::integral_constant<int, 7>{});
    case 8: return run(std::integral_constant<int, 8>{});
    default: return SVH_OK; // more than 512 disparities: a launch per pass
    }
}

} // namespace svh

// SGM Score branch (correlation/sgm.h:218-255, :329-389): the four downward passes of the reference's 8-direction aggregation as ONE
// sweep of the volume instead of one read-modify-write sweep per pass.  Two forms, both bit-identical to the per-pass kernels of
// svh_sgm.hip:
//   sgm_score_band_kernel   a launch per band of rows, every block recomputing the lines that enter its strip of columns (default)
//   sgm_score_down_kernel   one launch, strips of the skewed image handed from block to block through global memory (option)
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
        SVH_TRY(dev_sgm_score_line_pass(ctx, a, cv, sgm, 1, true)); // Left2Right: its contribution alone
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
        SVH_TRY(dev_sgm_score_line_pass(ctx, a, cv, sgm, 5, false)); // DownLeft2UpRight: read-modify-write
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
        SVH_TRY(dev_sgm_score_line_pass(ctx, a, cv, sgm, 1, true)); // Left2Right: its contribution alone
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
        SVH_TRY(dev_sgm_score_line_pass(ctx, a, cv, sgm, 5, false)); // DownLeft2UpRight: read-modify-write
        *ran = true;
        return SVH_OK;
    }
}

int dev_sgm_score_sweep(svh_context *ctx, Scratch &scr, const SgmArgs &a, const float *cv, float *sgm, bool vec, int form, bool *ran) {
    *ran = false;
    const auto run = [&](auto r_tag) -> int {
        constexpr int R = decltype(r_tag)::value;
        if (form == 2) return run_score_branch_fused<R>(ctx, scr, a, cv, sgm, vec, ran);
        // 16-column strips, 16 rows per band (8 / 12 / 16 rows: the same time at C4, 24: + 8 %); 8 x 8 where 16 columns would leave CUs
        // without a strip (1080p: 8 / 16 / 32 rows per band 2.56 / 2.78 / 4.05 ms)
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        if (form != 3 && ceil_div(a.W, 16) * 4 < cus * 3) return run_score_branch_bands<R, 8, 8>(ctx, scr, a, cv, sgm, vec, ran);
        return run_score_branch_bands<R, 16, 16>(ctx, scr, a, cv, sgm, vec, ran);
    };
    switch (pick_R(a.D)) {
    case 1: return run(std::integral_constant<int, 1>{});
    case 2: return run(std::integral_constant<int, 2>{});
    case 4: return run(std::integral_constant<int, 4>{});
    case 8: return run(std::integral_constant<int, 8>{});
    default: return SVH_OK; // more than 512 disparities: a launch per pass
    }
}

} // namespace svh

// The RightToLeft specialisation of the FP4 matrix-core census sweep: kernel and launch templates, shared by the two translation units
// that instantiate them (svh_census_sweep_rl.hip: one to four census words, and the RAGGED form; svh_census_sweep_rl_wide.hip: five to
// eight words -- windows of 13x13 and 15x15, 9x9 on colour images).
#pragma once

#include <type_traits>

#include <cstddef>

#include "svh_census_sweep_fp4.h"

namespace svh {

namespace {

using namespace fp4sweep;

constexpr int RL_WAVES = 4;

// blocks per CU the register budget is set for: three (168 registers) while the window buffers of three blocks fit the LDS
__host__ __device__ constexpr int rl_win(int nt, int ct) { return RL_WAVES * ct * 32 + 32 * (nt - 1); }
__host__ __device__ constexpr size_t rl_lds_bytes(int nw, int nt, int ct) { return (size_t)2 * 2 * ((nw + 1) / 2) * rl_win(nt, ct) * 16; }
__host__ __device__ constexpr int rl_blocks_per_cu(int nw, int nt, int ct) { return 3 * rl_lds_bytes(nw, nt, ct) <= 160 * 1024 ? 3 : (2 * rl_lds_bytes(nw, nt, ct) <= 160 * 1024 ? 2 : 1); }

// the kernel's arguments as the kernel argument segment lays them out (by-value arguments at their natural alignment, in order)
struct RlKernelArgs {
    CensusGeom g;
    float Pout;
    uint2 *keys;
    float *gmap;
    SweepPlan plan;
    SweepWinner sw;
};
static_assert(offsetof(RlKernelArgs, keys) % 8 == 0 && offsetof(RlKernelArgs, sw) % 8 == 0 && alignof(SweepWinner) == 8 && alignof(CensusGeom) == 8,
              "the WINNER epilogue reads `sw` from the kernel argument segment at offsetof(RlKernelArgs, sw)");

template <int N> __device__ __forceinline__ void buffer_load_words(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, uint32_t (&out)[N]) {
    static_assert(N >= 1 && N <= 8, "");
    if constexpr (N == 1) {
        out[0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, byte_off, 0, 0);
    } else if constexpr (N == 2) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, byte_off, 0, 0);
        out[0] = v[0];
        out[1] = v[1];
    } else if constexpr (N == 3) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b96(rsrc, byte_off, 0, 0);
        out[0] = v[0];
        out[1] = v[1];
        out[2] = v[2];
    } else {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
        out[0] = v[0];
        out[1] = v[1];
        out[2] = v[2];
        out[3] = v[3];
        if constexpr (N > 4) { // five to eight words: the rest of the record behind the first 16 bytes
            uint32_t rest[N - 4];
            buffer_load_words<N - 4>(rsrc, byte_off + 16, rest);
#pragma unroll
            for (int q = 0; q < N - 4; q++) out[4 + q] = rest[q];
        }
    }
}

// RAGGED: the disparity count g.D is no multiple of 32; the launch covers D = 32 (NT - 1) >= g.D disparities and the cells of the last
// two row tiles that lie past g.D are pushed out of reach by their start patterns, like the cells with d < 0 of tile 0 (until round 5 the
// matrix cores took the leading multiple of 32 and a second launch of the vector-ALU sweep the 1 .. 31 disparities behind it).
template <int NW, int NT, int CT, bool WINNER, int BPC = rl_blocks_per_cu(NW, NT, CT), bool RAGGED = false>
__global__ void __launch_bounds__(64 * RL_WAVES, BPC)
    census_sweep_rl_kernel(CensusGeom g, float Pout, uint2 *__restrict__ keys, float *__restrict__ gmap, SweepPlan plan, SweepWinner sw) {
    constexpr int WAVES = RL_WAVES;
    constexpr int PXB = WAVES * CT * 32;  // source pixels per item
    constexpr int D = 32 * (NT - 1);      // disparities this launch has cells for (RAGGED: g.D of them count)
    constexpr int WIN = PXB + D;          // window pixels: the last column tile's last row tile ends at PXB - 32 + 32 NT
    constexpr int SL = WIN / WAVES;       // window pixels fetched and expanded by each wave
    constexpr int PB = (SL + 63) / 64;    // ... in rounds of one pixel per lane
    constexpr int NG = (NW + 1) / 2;      // MFMAs per tile: 64 bits of census record each
    constexpr int B = 32 * NW;            // written census bits: cost = (B - dot) / 2
    constexpr int BUF = 2 * NG * WIN;     // records per window buffer
    static_assert(WIN % WAVES == 0 && NT >= 3, "");
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[]; // [2][2 NG planes][WIN] FP4 records

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (slot >= plan.items) return; // (block uniform, before any barrier)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;

    // the word maps as buffers: a load past either end returns zeros (first and last rows of the image, the padding pixels of the
    // last item of a row), whatever it returns for a column outside the image is replaced during the expansion
    const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc((void *)g.tw, 0, g.H * g.Wt * NW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_s = __builtin_amdgcn_make_buffer_rsrc((void *)g.sw, 0, g.H * g.Ws * NW * 4, 0x00020000);
    const uint32_t t_lane = (uint32_t)((wave * SL + lane) * NW * 4);            // this lane's pixel of the wave's window slice, round 0
    const uint32_t s_lane = (uint32_t)((r * NW + h) * 4);                        // word h of this lane's source pixel in a column tile

    // cell (row tile k, row, col) of column tile c is disparity 32 k + row - col of source pixel j = j0 + 32 c + col; it counts
    // (region without Pout) while j + d < R, R = Ws - d_offset: row < R - j0 - 32 c - 32 k.  R = 32 Rq + nv.
    const int R = g.Ws - g.d_offset, Rq = R >> 5, nv = R & 31;
    // the Pout region's key of a pixel: (|s|, last index) when j + last >= Ws (disparity shards: the last index of the whole range)
    const int last = g.region1_global_last >= 0 ? g.region1_global_last : g.d_offset + (RAGGED ? g.D : D) - 1;

    // Which column tiles (32 source pixels each, WAVES CT per item) a wave takes.  Away from the right image border every column tile
    // has its NT row tiles and a wave takes CT neighbours (their row tiles share window fragments).  In an item AT the border the
    // column tiles get shorter from left to right (kq below: NT, ..., NT, NT - 1, NT - 2, ...), the item's barrier makes every wave
    // wait for the one with the most tiles, and that one sits on the same SIMD in every block: there the waves take the column tiles
    // in serpentine order -- w, 2 WAVES - 1 - w, 2 WAVES + w, ... -- so that each gets long and short ones (1080p x 256, last item
    // of a row: 18 / 18 / 18 / 18 tiles instead of 27 / 24 / 15 / 6).
    auto border_item = [&](int j0) { return Rq - (j0 >> 5) - (WAVES * CT - 1) < NT; }; // (block uniform) its last column tile lacks row tiles
    const bool serpentine = plan.nbuf != 3; // (nbuf == 3: A/B switch of the development tools, neighbouring column tiles everywhere)
    auto ct_of = [&](int cc, bool border) { return border && serpentine ? cc * WAVES + ((cc & 1) ? WAVES - 1 - wave : wave) : wave * CT + cc; };
    uint32_t tword[PB][NW], snext[CT][NG], sword[CT][NG];
    auto issue_loads = [&](int i, int j0) {
        if (i >= g.H) return; // (padding rows of the last group of eight)
        const bool border = border_item(j0);
        const uint32_t t_off = (uint32_t)((i * g.Wt + j0 + g.disp_lower) * NW * 4) + t_lane;
#pragma unroll
        for (int b = 0; b < PB; b++) buffer_load_words<NW>(rsrc_t, t_off + (uint32_t)(b * 64 * NW * 4), tword[b]);
        const uint32_t s_off = (uint32_t)((i * g.Ws + j0) * NW * 4) + s_lane;
#pragma unroll
        for (int cc = 0; cc < CT; cc++)
#pragma unroll
            for (int gq = 0; gq < NG; gq++)
                snext[cc][gq] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_s, s_off + (uint32_t)((ct_of(cc, border) * 32 * NW + 2 * gq) * 4), 0, 0);
    };
    // registers -> FP4 records of window buffer `buf` (this wave's SL pixels); the source words of the item move into place
    const int wr_lane = wave * SL + lane; // window pixel of this lane, round 0
    auto expand = [&](int i, int j0, int buf) {
        if (i >= g.H) return;
        uint4 *const win = lds4 + buf * BUF;
        const int col0 = j0 + g.disp_lower + wave * SL;             // target column of the slice's first pixel
        const bool interior = col0 >= 0 && col0 + SL <= g.Wt;       // (wave uniform) every column of the slice is inside the image
        if (!interior) {
#pragma unroll
            for (int b = 0; b < PB; b++) {
                const int col = col0 + 64 * b + lane;
                const bool inside = col >= 0 && col < g.Wt;         // outside: the all-zero vector (cross_correlations.h:235)
#pragma unroll
                for (int w = 0; w < NW; w++) tword[b][w] = inside ? tword[b][w] : 0u;
            }
        }
#pragma unroll
        for (int b = 0; b < PB; b++) {
            const bool live = 64 * (b + 1) <= SL || lane < SL - 64 * b;
            if (live) {
#pragma unroll
                for (int w = 0; w < NW; w++) win[w * WIN + wr_lane + 64 * b] = word_record(tword[b][w]);
            }
        }
#pragma unroll
        for (int cc = 0; cc < CT; cc++)
#pragma unroll
            for (int gq = 0; gq < NG; gq++) sword[cc][gq] = snext[cc][gq];
    };

    // ---- once per block: the missing plane of an odd word count (0.0 operands: contributes nothing)
    if constexpr (NW & 1) {
        for (int b = 0; b < 2; b++)
            for (int y = threadIdx.x; y < WIN; y += 64 * WAVES) lds4[b * BUF + NW * WIN + y] = make_uint4(0u, 0u, 0u, 0u);
    }
    // the walk over this block's items: s = slot, slot + step, ... as (q, jb) = (s / bpr, s % bpr): row 8 q + xcd, source pixel PXB jb
    int s = slot, q = slot / plan.bpr, jb = slot - q * plan.bpr;
    auto advance = [&](int &qq, int &jj) {
        qq += plan.step_q;
        jj += plan.step_j;
        if (jj >= plan.bpr) {
            jj -= plan.bpr;
            qq++;
        }
    };
    issue_loads(q * 8 + xcd, jb * PXB);
    // accumulator starts: a cell is 512 dot + its ROW in the tile (+ 32 per row tile, in the frame of the current tile: the running
    // maximum drops by 32 per tile) -- svh_census_sweep_pm.hip starts from the register number instead, which orders the cells of a
    // lane but not those of the two lane halves; with the row in the cell a packed maximum is 1024 (B - cost) + (d + col) - 512 B
    // whichever lane took it.  The two triangular tiles push their invalid cells out of reach.  Opaque so that the patterns live in
    // registers and go into the MFMA as its C operand
    float opaque_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
    // the fragment-major interior path below (with a fourth start pattern for its odd row tiles): where it was measured to pay -- 256
    // disparities (1080p: 31.4 -> 31.0 us); at 512 disparities (8192 x 4320, two blocks per CU) it ran 2.41 ms against 1.36 ms column-major
    constexpr bool FRAGMAJOR = NT == 9 && NW != 1 && !RAGGED;
    constexpr bool ODD = FRAGMAJOR;
    // RAGGED: g.D = 32 (NT - 2) + e, 0 < e < 32: tile NT - 2 holds d = 32 (NT - 2) + row - col < g.D where row - col < e, the last tile
    // d = 32 (NT - 1) + row - col < g.D where row - col < e - 32 (the aligned form is e = 32: tile NT - 2 whole)
    const int e_rag = RAGGED ? g.D - 32 * (NT - 2) : 32;
    v16f full_tile, first_tile, last_tile, odd_tile; // (odd_tile: the fragment-major path's odd row tiles; RAGGED: the start pattern of tile NT - 2)
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        const int rr = row_of_reg(reg) + 4 * h;
        full_tile[reg] = (float)rr + opaque_zero;
        if constexpr (RAGGED) odd_tile[reg] = (float)rr - (rr - r < e_rag ? 0.0f : BIG_CELL) + opaque_zero;
        else odd_tile[reg] = (float)(rr + (ODD ? 32 : 0)) + opaque_zero; // an odd row tile starts one tile tag higher: its frame shift is paid by the next even tile
        first_tile[reg] = (float)rr - (rr >= r ? 0.0f : BIG_CELL) + opaque_zero; // tile 0: d = row - col >= 0
        last_tile[reg] = (float)rr - (rr - r < e_rag - 32 ? 0.0f : BIG_CELL) + opaque_zero; // tile D/32: d = D + row - col < g.D
    }
    // start pattern of row tile k (k > 0; compile-time k after unrolling, wave-uniform otherwise)
    auto start_of = [&](int k) -> const v16f & { return k == NT - 1 ? last_tile : ((RAGGED && k == NT - 2) ? odd_tile : full_tile); };
    expand(q * 8 + xcd, jb * PXB, 0);
    int q_n = q, jb_n = jb; // the item after the current one
    advance(q_n, jb_n);
    if (s + plan.step < plan.items) issue_loads(q_n * 8 + xcd, jb_n * PXB);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the window writes are done ...
    __builtin_amdgcn_s_barrier();                       // ... raw: a __syncthreads() would also wait for the loads just issued

    const uint32_t frag_lane = (uint32_t)(h * WIN + r); // record of (row tile 0, group 0) of the item's column tile 0 in a buffer

    int buf = 0;
    for (;;) {
        const int i = q * 8 + xcd, j0 = jb * PXB;
        if (i < g.H) {
            const uint4 *const arow = lds4 + buf * BUF + frag_lane;
            const bool border = border_item(j0);
            const int64_t row = (int64_t)i * g.Ws;
            struct Frags {
                v8i w[NG];
            };
            float Gabs[CT]; // per column tile: the packed maximum over its row tiles, of this lane's rows (4 h + ...), tile tags absolute
            bool all_interior = true;
            // row tiles of column tile ct that hold cells of the region without Pout: all of tiles 0 .. kq - 1, rows below nv of tile kq,
            // kq = Rq - j0 / 32 - ct; the item's last column tile has the smallest: border_item()
            if (FRAGMAJOR && !border && plan.nbuf != 1) { // (nbuf == 1: A/B switch of the development tools, column-major tiles everywhere)
                // ---- every row tile of every column tile of this wave counts: all items but the ones at the right image border.
                // FRAGMENT-MAJOR: window group p (32 target columns, one LDS read) is row tile p - c of column tile c, so one fragment
                // serves up to CT MFMAs: NT + CT - 1 reads for CT NT tiles instead of one read per tile.  A column tile's row tiles still
                // arrive in increasing order, so its running maximum works as before; the MFMA of tile t + 1 is issued before the maximum of
                // tile t is taken.  Odd row tiles start from odd_tile (one tag up): H = max3(G, t6, t7) needs no frame shift, the even tile
                // after it pays both, max3(H - 64, t6, t7) -- half the shifts.
                v8i bfr[CT][NG];
#pragma unroll
                for (int cc = 0; cc < CT; cc++)
#pragma unroll
                    for (int gq = 0; gq < NG; gq++) {
                        const uint4 rec = word_record(sword[cc][gq]);
                        const bool present = h == 0 || 2 * gq + 1 < NW;
                        bfr[cc][gq] = (v8i){present ? (int)rec.x : 0, present ? (int)rec.y : 0, present ? (int)rec.z : 0, present ? (int)rec.w : 0, 0, 0, 0, 0};
                    }
                const uint4 *const arow_w = arow + wave * CT * 32; // (the wave's CT neighbouring column tiles)
                auto load_group = [&](int p) { // window group p: arow_w[2 gq WIN + 32 p]
                    Frags f;
#pragma unroll
                    for (int gq = 0; gq < NG; gq++) {
                        const uint4 t = arow_w[2 * gq * WIN + 32 * p];
                        f.w[gq] = (v8i){(int)t.x, (int)t.y, (int)t.z, (int)t.w, 0, 0, 0, 0};
                    }
                    return f;
                };
                auto mfma_tile = [&](const Frags &f, int c, int k) {
                    v16f a = k == 0 ? first_tile : (k == NT - 1 ? last_tile : ((k & 1) ? odd_tile : full_tile));
#pragma unroll
                    for (int gq = 0; gq < NG; gq++) a = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(f.w[gq], bfr[c][gq], a, 4, 4, 0, 132, 0, 131);
                    return a;
                };
                float G[CT];
#pragma unroll
                for (int cc = 0; cc < CT; cc++) G[cc] = -BIG_G;
                auto take_max = [&](const v16f &a, int c, int k) {
                    const float t1 = max3f(a[0], a[1], a[2]), t2 = max3f(a[3], a[4], a[5]), t3 = max3f(a[6], a[7], a[8]), t4 = max3f(a[9], a[10], a[11]),
                                t5 = max3f(a[12], a[13], a[14]);
                    const float t6 = max3f(t1, t2, t3), t7 = max3f(t4, t5, a[15]);
                    if (ODD) G[c] = (k & 1) ? max3f(G[c], t6, t7) : max3f(G[c] - 64.0f, t6, t7);
                    else G[c] = max3f(G[c] - 32.0f, t6, t7);
                };
                constexpr int NP = NT + CT - 1;
                Frags fr[2];
                v16f acc[2];
                fr[0] = load_group(0);
                fr[1] = load_group(1);
                acc[0] = mfma_tile(fr[0], 0, 0);
                int t = 0, pc = 0, pk = 0; // tiles issued so far (compile-time after unrolling); the tile whose maximum is pending
#pragma unroll
                for (int p = 0; p < NP; p++) {
#pragma unroll
                    for (int c = 0; c < CT; c++) {
                        const int k = p - c;
                        if (k < 0 || k >= NT) continue;
                        if (t > 0) { // (tile 0 was issued ahead of the loop)
                            acc[t & 1] = mfma_tile(fr[p & 1], c, k);
                            take_max(acc[(t - 1) & 1], pc, pk);
                        }
                        pc = c;
                        pk = k;
                        t++;
                    }
                    if (p + 2 < NP) fr[p & 1] = load_group(p + 2); // (every MFMA of group p has been issued)
                }
                take_max(acc[(t - 1) & 1], pc, pk);
#pragma unroll
                for (int cc = 0; cc < CT; cc++) Gabs[cc] = G[cc] + (float)(32 * (NT - 1));
            } else
#pragma unroll
            for (int cc = 0; cc < CT; cc++) {
                const int ct = ct_of(cc, border);
                const uint4 *const arow_c = arow + ct * 32;
                // source fragments: this lane's word of every 64-bit group as +-1.0 nibbles (an absent odd word: 0.0)
                v8i bfrag[NG];
#pragma unroll
                for (int gq = 0; gq < NG; gq++) {
                    const uint4 rec = word_record(sword[cc][gq]);
                    const bool present = h == 0 || 2 * gq + 1 < NW;
                    bfrag[gq] = (v8i){present ? (int)rec.x : 0, present ? (int)rec.y : 0, present ? (int)rec.z : 0, present ? (int)rec.w : 0, 0, 0, 0, 0};
                }
                auto load_frags = [&](int k) { // row tile k, group gq: arow[2 gq WIN + 32 (ct + k)]
                    Frags f;
#pragma unroll
                    for (int gq = 0; gq < NG; gq++) {
                        const uint4 t = arow_c[2 * gq * WIN + 32 * k];
                        f.w[gq] = (v8i){(int)t.x, (int)t.y, (int)t.z, (int)t.w, 0, 0, 0, 0};
                    }
                    return f;
                };
                auto tile = [&](const Frags &f, const v16f &start) {
                    v16f a = start;
#pragma unroll
                    for (int gq = 0; gq < NG; gq++) // FP4 x FP4, block scales 2^5 x 2^4 (E8M0 132, 131): an equal bit adds 512, a differing one -512
                        a = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(f.w[gq], bfrag[gq], a, 4, 4, 0, 132, 0, 131);
                    return a;
                };
                float G = -BIG_G; // running maximum, frame of the current tile
                auto epilogue = [&](const v16f &a) {
                    const float t1 = max3f(a[0], a[1], a[2]), t2 = max3f(a[3], a[4], a[5]), t3 = max3f(a[6], a[7], a[8]), t4 = max3f(a[9], a[10], a[11]),
                                t5 = max3f(a[12], a[13], a[14]);
                    const float t6 = max3f(t1, t2, t3), t7 = max3f(t4, t5, a[15]);
                    G = max3f(G - 32.0f, t6, t7); // frame of tile k: max over the tiles so far of (tile maximum - 32 (k - tile))
                };
                // row tiles of this column tile that hold cells of the region without Pout: all of tiles 0 .. kq - 1, rows below nv of tile kq
                const int kq = Rq - (j0 >> 5) - ct; // (wave uniform)
                if (kq >= NT) {
                    // every row tile counts -- every column tile but the ones at the right image border.  Straight-line: the fragments
                    // of tile k + 2 requested, the MFMA of tile k + 1 in flight while the maximum of tile k is taken
                    Frags f_a = load_frags(0), f_b = load_frags(1);
                    v16f acc_a = tile(f_a, first_tile), acc_b;
#pragma unroll
                    for (int k = 0; k < NT; k++) {
                        if (k + 1 < NT) {
                            if (k & 1) acc_a = tile(f_a, start_of(k + 1));
                            else acc_b = tile(f_b, start_of(k + 1));
                        }
                        if (k + 2 < NT) {
                            if (k & 1) f_b = load_frags(k + 2);
                            else f_a = load_frags(k + 2);
                        }
                        epilogue((k & 1) ? acc_b : acc_a);
                    }
                    Gabs[cc] = G + (float)(32 * (NT - 1));
                } else {
                    // near the right border: tiles 0 .. kq - 1 whole, tile kq up to row nv - 1; nothing further right counts
                    all_interior = false;
                    const int k_end = kq < 0 ? 0 : (nv > 0 ? kq + 1 : kq); // (<= NT: kq < NT)
                    for (int k = 0; k < k_end; k++) {
                        v16f start = k == 0 ? first_tile : start_of(k);
                        if (k == kq) {
#pragma unroll
                            for (int reg = 0; reg < 16; reg++) start[reg] = row_of_reg(reg) + 4 * h < nv ? start[reg] : -BIG_CELL;
                        }
                        Frags f;
#pragma unroll
                        for (int gq = 0; gq < NG; gq++) {
                            const uint4 t = arow_c[2 * gq * WIN + 32 * k];
                            f.w[gq] = (v8i){(int)t.x, (int)t.y, (int)t.z, (int)t.w, 0, 0, 0, 0};
                        }
                        epilogue(tile(f, start));
                    }
                    Gabs[cc] = G + (float)(32 * (k_end - 1));
                }
            }
            // ---- per pixel: keys (or the winner) from the packed maxima.  A lane half holds rows 4 h + {0..3, 8..11, ...} of its pixel's
            // cells; packed maxima compare across the halves as they are (the row is in them), so the halves merge BEFORE the decode,
            // and two column tiles merge into one wave of 64 different pixels: v_permlane32_swap puts tile c's two halves into the lower
            // lanes of (x, y), tile c + 1's into the upper lanes.  A last odd column tile takes the lower lanes alone.
            auto finish = [&](int c0, bool pair) {
                SweepWinner w;
                if constexpr (WINNER) {
                    // `sw` is read from the kernel argument segment HERE, once per finish (an opaque pointer: the loads cannot be hoisted into
                    // the tile code, which has no scalar registers to spare), and first used a few dozen instructions further down
                    typedef __attribute__((address_space(4))) const char *KernArg;
                    KernArg ka = (KernArg)__builtin_amdgcn_kernarg_segment_ptr();
                    asm volatile("" : "+s"(ka));
                    typedef __attribute__((address_space(4))) const uint32_t *KernArgWords;
                    KernArgWords kw = (KernArgWords)(ka + offsetof(RlKernelArgs, sw));
                    static_assert(sizeof(SweepWinner) % 4 == 0, "read as dwords");
                    uint32_t raw[sizeof(SweepWinner) / 4];
#pragma unroll
                    for (unsigned qq = 0; qq < sizeof(SweepWinner) / 4; qq++) raw[qq] = kw[qq];
                    __builtin_memcpy(&w, raw, sizeof w);
                }
                float x = Gabs[c0], y = pair ? Gabs[c0 + 1] : Gabs[c0];
                int ones_x = 0, ones_y = 0; // |s|: each lane half holds its words of the pixel
#pragma unroll
                for (int gq = 0; gq < NG; gq++) {
                    const bool present = 2 * gq + 1 < NW || h == 0;
                    ones_x += present ? __popc(sword[c0][gq]) : 0;
                    ones_y += present ? __popc(sword[pair ? c0 + 1 : c0][gq]) : 0;
                }
                const auto sg = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
                const float gf = __builtin_fmaxf(__uint_as_float(sg[0]), __uint_as_float(sg[1]));
                const int ct_x = ct_of(c0, border), ct_y = pair ? ct_of(c0 + 1, border) : ct_x; // (wave uniform)
                const int jp = j0 + ((pair && h) ? ct_y : ct_x) * 32 + r;     // this lane's pixel
                // gf = 1024 (B - cost) + 32 tile + row - 512 B, and 32 tile + row = d + col (col = lane & 31)
                const int gv = (int)gf + 512 * B, cost0 = B - (gv >> 10);
                int a0 = (int)make_key(cost0, g.d_offset + (gv & 1023) - r), a1 = (int)KEY_NONE;
                int g2 = 2 * cost0; // g = min over the disparities of 2 c (+ Pout), svh_census_keys.h: g_from_keys
                if (!all_interior) { // (wave uniform) a pixel right of the region's edge has no cell at all
                    const bool any = gf > G_VALID_ABOVE;
                    a0 = any ? a0 : (int)KEY_NONE;
                    g2 = any ? g2 : (1 << 24);
                }
                if (j0 + max(ct_x, ct_y) * 32 + 31 + last >= g.Ws) { // (wave uniform) some pixel here has disparities that pay Pout
                    const auto so = __builtin_amdgcn_permlane32_swap((unsigned)ones_x, (unsigned)ones_y, false, false);
                    const int ones = (int)(so[0] + so[1]);
                    const bool pays = jp + last >= g.Ws;
                    a1 = pays ? (int)make_key(ones, last) : (int)KEY_NONE;
                    g2 = min(g2, pays ? 2 * ones + (int)Pout : (1 << 24));
                }
                if ((pair || h == 0) && jp < g.Ws) {
                    if constexpr (WINNER) { // the winner itself: census_keys.h
                        const int bd = winner_index((uint32_t)a0, (uint32_t)a1, passes_visiting(w, i, jp), w.pout);
                        if ((unsigned)(i - w.row_first) < (unsigned)w.row_count) {
                            const int64_t o = (int64_t)(i - w.row_first) * g.Ws + jp;
                            if (w.idx) w.idx[o] = bd;
                            if (w.disp) w.disp[o] = w.disp_sign * bd + w.disp_offset;
                        }
                    } else {
                        (keys + row)[(uint32_t)jp] = make_uint2((uint32_t)a0, (uint32_t)a1);
                        if (gmap) (gmap + row)[(uint32_t)jp] = (float)g2;
                    }
                }
            };
#pragma unroll
            for (int c0 = 0; c0 + 1 < CT; c0 += 2) finish(c0, true);
            if constexpr (CT & 1) finish(CT - 1, false);
        }
        s += plan.step;
        if (s >= plan.items) break; // (block uniform)
        q = q_n;
        jb = jb_n;
        advance(q_n, jb_n);
        expand(q * 8 + xcd, jb * PXB, buf ^ 1); // (waits for the words requested before the tiles)
        if (s + plan.step < plan.items) issue_loads(q_n * 8 + xcd, jb_n * PXB);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        buf ^= 1;
    }
}

template <int NW, int NT, int CT, bool RAGGED = false, int BPC = rl_blocks_per_cu(NW, NT, CT)> int launch_rl(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, const SweepWinner &sw) {
    constexpr int PXB = RL_WAVES * CT * 32;
    static int n_cu[64] = {};
    const int dev = ctx->device & 63;
    int n = __atomic_load_n(&n_cu[dev], __ATOMIC_RELAXED);
    if (!n) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || n <= 0) n = 256;
        __atomic_store_n(&n_cu[dev], n, __ATOMIC_RELAXED);
    }
    constexpr size_t shmem = rl_lds_bytes(NW, NT, CT);
    int grid;
    const SweepPlan plan = make_walk(g.H, g.Ws, PXB, n, BPC, ctx->census_sweep_rl == 2 ? 1 : (ctx->census_sweep_rl == 3 ? 3 : 2), &grid);
    static int attr_set[64] = {}; // (per instantiation and device)
    if (!__atomic_load_n(&attr_set[dev], __ATOMIC_ACQUIRE)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&census_sweep_rl_kernel<NW, NT, CT, false, BPC, RAGGED>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&census_sweep_rl_kernel<NW, NT, CT, true, BPC, RAGGED>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return fail(ctx, SVH_ERR_HIP, "census_sweep (fp4, right-to-left): cannot raise the dynamic LDS limit");
        __atomic_store_n(&attr_set[dev], 1, __ATOMIC_RELEASE);
    }
    if (sw.on()) SVH_LAUNCH(ctx, "census_sweep_rl", (census_sweep_rl_kernel<NW, NT, CT, true, BPC, RAGGED>), grid, 64 * RL_WAVES, shmem, g, Pout, keys, gmap, plan, sw);
    else SVH_LAUNCH(ctx, "census_sweep_rl", (census_sweep_rl_kernel<NW, NT, CT, false, BPC, RAGGED>), grid, 64 * RL_WAVES, shmem, g, Pout, keys, gmap, plan, sw);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SVH_OK : fail(ctx, SVH_ERR_HIP, "census_sweep (fp4, right-to-left): %s", hipGetErrorString(e));
}

// does the configuration exist: two window buffers of one block within the CU's LDS
template <int NW, int NT, int CT> constexpr bool rl_fits() { return rl_lds_bytes(NW, NT, CT) <= 160 * 1024; }

template <int NW, int NT> bool launch_rl_width(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, int *status, const SweepWinner &sw) {
    if constexpr (NW > 4) {
        // five to eight census words (three or four MFMAs' worth of fragments per tile): two column tiles per wave, 256-pixel items
        if constexpr (rl_fits<NW, NT, 2>()) {
            *status = launch_rl<NW, NT, 2>(ctx, g, Pout, keys, gmap, sw);
            return true;
        } else return false; // (the general kernel: one window buffer)
    } else {
        // item width: 384 source pixels, or 512 where that pads the row less (and its window buffers fit)
        constexpr bool wide_fits = 2 * rl_lds_bytes(NW, NT, 4) <= 160 * 1024;
        if constexpr (wide_fits) {
            if ((int64_t)ceil_div(g.Ws, 512) * 512 < (int64_t)ceil_div(g.Ws, 384) * 384) {
                *status = launch_rl<NW, NT, 4>(ctx, g, Pout, keys, gmap, sw);
                return true;
            }
        }
        *status = launch_rl<NW, NT, 3>(ctx, g, Pout, keys, gmap, sw);
        return true;
    }
}

template <int NW> bool launch_rl_words(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, int *status, const SweepWinner &sw) {
    if (g.D & 31) { // 33 .. 511 disparities, no multiple of 32: the RAGGED form of the next multiple (384-pixel items; up to four words)
        if constexpr (NW <= 4) {
            switch ((g.D + 31) >> 5) {
#define SVH_RL_RAGGED(M) case M: *status = launch_rl<NW, M + 1, 3, true>(ctx, g, Pout, keys, gmap, sw); return true;
                SVH_RL_RAGGED(2) SVH_RL_RAGGED(3) SVH_RL_RAGGED(4) SVH_RL_RAGGED(5) SVH_RL_RAGGED(6) SVH_RL_RAGGED(7) SVH_RL_RAGGED(8) SVH_RL_RAGGED(9)
                SVH_RL_RAGGED(10) SVH_RL_RAGGED(11) SVH_RL_RAGGED(12) SVH_RL_RAGGED(13) SVH_RL_RAGGED(14) SVH_RL_RAGGED(15) SVH_RL_RAGGED(16)
#undef SVH_RL_RAGGED
            default: return false;
            }
        }
        return false;
    }
    switch (g.D) { // every multiple of 32 from 64 to 512: the row tiles of a column tile are straight-line code (NT = D / 32 + 1)
#define SVH_RL_D(NTV) case 32 * (NTV - 1): return launch_rl_width<NW, NTV>(ctx, g, Pout, keys, gmap, status, sw);
        SVH_RL_D(3) SVH_RL_D(4) SVH_RL_D(5) SVH_RL_D(6) SVH_RL_D(7) SVH_RL_D(8) SVH_RL_D(9) SVH_RL_D(10) SVH_RL_D(11) SVH_RL_D(12) SVH_RL_D(13) SVH_RL_D(14)
        SVH_RL_D(15) SVH_RL_D(16) SVH_RL_D(17)
#undef SVH_RL_D
    default: return false;
    }
}

// the geometry the specialisation takes (launch_sweep_rl / launch_sweep_rl_wide)
inline bool rl_geometry_ok(const CensusGeom &g, const SweepWinner &sw) {
    if (sw.on() && g.region1_global_last >= 0) return false; // (disparity shards exchange keys)
    if (g.sign <= 0) return false;                            // RightToLeft only
    // the disparities that pay Pout (j + global index >= Ws) are exactly the ones whose target column j + disp_lower + d is outside
    if (g.Wt - g.disp_lower != g.Ws - g.d_offset) return false;
    // buffer descriptors and offsets are 32-bit
    if ((int64_t)g.H * g.Wt * g.nWw * 4 >= (int64_t)1 << 31 || (int64_t)g.H * g.Ws * g.nWw * 4 >= (int64_t)1 << 31) return false;
    if (g.d_offset + g.D > (int)KEY_IDX_MASK + 1) return false;
    return true;
}

} // namespace

} // namespace svh

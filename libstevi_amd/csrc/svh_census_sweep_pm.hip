// census_sweep on the matrix cores: FP4 operands +-1, persistent blocks with the next item's operands in flight.
//
// The one voxel-sized kernel of the census + SGM path (svh_census_sgm.hip has the algebra: per pixel the two regional winner
// keys (c << 12 | 4095 - d) and g = min_d [2 c (+ Pout)]).  Hamming distance as a dot product on
// v_mfma_scale_f32_32x32x64_f8f6f4:
//
//     bit 0 -> +1.0, bit 1 -> -1.0 on BOTH sides (svh_nibble_tables.h):   dot = B - 2 c   (B = written census bits)
//     32 target columns (rows of the tile, operand A) x 32 source pixels (columns, operand B) per MFMA, 64 bits of record each
//     block scales 2^4 x 2^4: a cell is 256 dot + reg (+ 16 k in the frame of row tile k), an exact float (integers below 2^17)
//
// The winner wants the SMALLEST cost and, among equals, the LARGEST disparity (extractSelectedIndex's '<=' scan,
// correlation_base.h:441-455): both are "largest cell" once the accumulator does not start at 0 but at its register number --
// in the C/D layout a lane holds one column (source pixel) and 16 rows, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5),
// increasing with reg -- and row tile k carries the tag 16 k.  The tile maximum is a tree of eight v_max3_f32; across row tiles
// the running maximum lives in the frame of the current tile, G <- max3(G - 16, t6, t7): one add per tile.  Column tile c
// (32 source pixels from J) meets row tiles k = 0 .. D/32 starting at target column J + lower + 32 k, d = 32 k + row - col: tile
// 0 is valid on and below its diagonal, tile D/32 strictly above it; those two start from a pattern that holds -2^22 in the
// invalid cells (both patterns live in registers for the life of the block).  With +-1 on both sides |s| drops out of the cost.
//
// Structure.  An ITEM is PXB = 32 CT WAVES source pixels of one image row; the first FP4 kernel (round 1) ran one item per
// block, and in-kernel stamps (tools/sweep_phases.hip, profiles/r02_sweep_phases_*.json) showed what that costs: a wave spent
// ~40 % of its life between kernel entry and the first MFMA (kernel arguments -> addresses -> loads -> expansion -> barrier, a
// chain of dependent latencies paid by every block) with only ~27 tiles of work behind it.  Here a block is PERSISTENT: a few
// per CU, each walking a static list of items, and the chain runs one item ahead of the arithmetic:
//
//     item n:    MFMA tiles out of window buffer n & 1                           (matrix pipe + VALU)
//     item n+1:  its census words land in the wave's own LDS staging slice       (LDS-DMA, issued before item n's tiles)
//                after item n's tiles: staging -> FP4 nibbles -> window buffer (n + 1) & 1   (byte table, 10 VALU per word)
//     item n+2:  DMA issued, one workgroup barrier, next round
//
// Every wave fetches and expands exactly a quarter of the window and the source words of its own column tiles, so the only
// synchronisation per item is that one barrier; no global load returns into a register inside the loop.  Compact words in
// memory (8 bytes per pixel) -- writing the target as FP4 records from the census kernel was tried first and lost: 4x the
// bytes through a CU's ~30 B/clk fill path cost more than the expansion it saved.
// Items are numbered so that the blocks of one XCD walk the same image rows (their windows overlap: one L2).
//
// Also takes LeftToRight (target column j - lower - d): window and lanes mirrored so that the window position still grows with d;
// the Pout boundary then advances by two cells per lane and is handled by masking the (up to three) row tiles it crosses.
//
// SVH_SWEEP_STAMPS (tools/sweep_phases.hip): s_memtime stamps per wave and phase into a side buffer; never in the product build.
#include <type_traits>

#include <cstddef>

#include "svh_census_sweep_fp4.h"
#include "svh_nibble_tables.h"

namespace svh {

namespace {

using namespace fp4sweep;

#ifdef SVH_SWEEP_STAMPS
#define SVH_STAMP(k) stamp[k] = __builtin_amdgcn_s_memtime()
#define SVH_STAMP_PARAM , unsigned long long *__restrict__ stamp_out
#else
#define SVH_STAMP(k)
#define SVH_STAMP_PARAM
#endif

__host__ __device__ inline int staging_chunks(int pixels, int nw) { return (pixels * nw * 4 + 15) / 16 + 1; } // 16-byte pieces of a slice (+1: a slice may start mid-piece)

inline size_t pm_lds_bytes(int nw, int waves, int ct, int D, int nbuf) {
    const int pxb = waves * ct * 32, win = pxb + D, sl = win / waves;
    return (size_t)nbuf * 2 * ((nw + 1) / 2) * win * 16 + (size_t)waves * (staging_chunks(sl, nw) + staging_chunks(ct * 32, nw)) * 16 + 16; // + dump record
}

// blocks per CU the register budget is set for (waves per SIMD with one wave per SIMD and block): three, or two where the
// fragments of four census words or four column tiles per wave want more than 168 registers
constexpr int pm_blocks_per_cu(int nw, int ct) { return (ct >= 4 || nw > 2) ? 2 : 3; }

// the kernel's arguments as the kernel argument segment lays them out (by-value arguments at their natural alignment, in order)
struct SweepKernelArgs {
    CensusGeom g;
    float Pout;
    uint2 *keys;
    float *gmap;
    SweepPlan plan;
    SweepWinner sw;
};

// WINNER: the index / disparity maps instead of keys and g (a separate instantiation: the fields of `sw` would otherwise sit in
// scalar registers through the tile loop of the key-writing form too)
template <int NW, int WAVES, int CT, bool WINNER = false>
__global__ void __launch_bounds__(64 * WAVES, pm_blocks_per_cu(NW, CT)) census_sweep_pm_kernel(CensusGeom g, float Pout, uint2 *__restrict__ keys, float *__restrict__ gmap,
                                                                     SweepPlan plan, SweepWinner sw SVH_STAMP_PARAM) {
    constexpr int PXB = WAVES * CT * 32; // source pixels per item
    constexpr int NG = (NW + 1) / 2;     // MFMAs per tile: 64 bits of census record each
    constexpr int B = 32 * NW;           // written census bits: cost = (B - dot) / 2
    constexpr bool AFFINE = 64 % NW == 0; // the expansion's lane -> (pixel, word) map repeats every 64 units: addresses are lane base + immediate
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
#ifdef SVH_SWEEP_STAMPS
    unsigned long long stamp[8] = {};
    int n_items_done = 0;
#endif
    SVH_STAMP(0);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (slot >= plan.items) return; // (block uniform, before any barrier)
    const int NT = g.D / 32 + 1;            // row tiles per column tile
    const int WIN = PXB + g.D;              // window pixels: the last column tile's last row tile ends at PXB - 32 + 32 NT
    const int SL = WIN / WAVES;             // window pixels fetched and expanded by each wave (WIN is a multiple of 32)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const bool fwd = g.sign > 0;       // RightToLeft: target column j + lower + d; LeftToRight: j - lower - d, everything mirrored
    const int xoff = fwd ? r : 31 - r; // source pixel of this lane inside its column tile (the lane index stays the tile's `col`)
    // ---- LDS: [nbuf][2 NG planes][WIN] FP4 records | per wave: target staging, source staging (compact words) | dump record
    const int t_chunks = staging_chunks(SL, NW), s_chunks = staging_chunks(CT * 32, NW);
    uint4 *const stage_t4 = lds4 + plan.nbuf * 2 * NG * WIN + wave * (t_chunks + s_chunks), *const stage_s4 = stage_t4 + t_chunks;
    uint4 *const dump_lds4 = lds4 + plan.nbuf * 2 * NG * WIN + WAVES * (t_chunks + s_chunks); // one record nobody reads
    const uint32_t *const stage_t = reinterpret_cast<const uint32_t *>(stage_t4), *const stage_s = reinterpret_cast<const uint32_t *>(stage_s4);
    // sizes of the word maps rounded up to whole 16-byte pieces (the workspace pool hands out multiples of 256 bytes, so the last
    // piece is readable; < 2^31: checked on the host)
    const int t_last = (int)((((int64_t)g.H * g.Wt * NW * 4 + 15) & ~(int64_t)15) - 16), s_last = (int)((((int64_t)g.H * g.Ws * NW * 4 + 15) & ~(int64_t)15) - 16);
    const char *const tw_bytes = reinterpret_cast<const char *>(g.tw), *const sw_bytes = reinterpret_cast<const char *>(g.sw);

    // first target column (ascending memory order) of this wave's window slice for an item that starts at source pixel j0
    const int slice_col_off = fwd ? g.disp_lower + wave * SL : PXB - 1 - g.disp_lower - wave * SL - (SL - 1);
    // ---- LDS-DMA of an item's compact words into this wave's staging slices.  Pieces are 16 bytes from 16-byte aligned offsets
    // of the word maps, clamped into the maps (what lies outside the image is replaced during the expansion, never read).
    auto issue_dma = [&](int i, int j0) {
        if (i >= g.H) return; // (padding rows of the last group of eight: nothing to fetch)
        const int t_base = ((i * g.Wt + j0 + slice_col_off) * NW * 4) & ~15; // (may be negative on the first row)
        for (int c0 = 0; c0 < t_chunks; c0 += 64) {                         // (one or two wave-instructions)
            const int off = min(max(t_base + 16 * (c0 + lane), 0), t_last);
            if (c0 + lane < t_chunks) glds16(tw_bytes + (uint32_t)off, stage_t4 + c0);
        }
        const int s_base = ((i * g.Ws + j0 + wave * CT * 32) * NW * 4) & ~15;
        for (int c0 = 0; c0 < s_chunks; c0 += 64) {
            const int off = min(max(s_base + 16 * (c0 + lane), 0), s_last);
            if (c0 + lane < s_chunks) glds16(sw_bytes + (uint32_t)off, stage_s4 + c0);
        }
    };
    // ---- staging -> FP4 records of window buffer `buf` (this wave's SL pixels), and this lane's source words of the item.
    // Window pixel y is target column j0 + lower + y (LeftToRight: j0 + PXB - 1 - lower - y, so that y still grows with the
    // disparity); a column outside the image is the all-zero word (cross_correlations.h:235), whose record is table entry 0.
    // Unit u = 64 b + lane of a round is word w of slice pixel yy, u = yy NW + w.  When NW divides 64 the pair (yy - (64 / NW) b, w)
    // depends on the lane only: the staging and window addresses of a unit are a per-lane base plus a compile-time multiple of b.
    constexpr int UB = 5; // words per lane and round: all their staging reads, then all their table reads, then the stores
    const int yy_lane = AFFINE ? lane / NW : 0, w_lane = AFFINE ? lane % NW : 0;
    const int rd_lane = fwd ? lane : (SL - 1 - yy_lane) * NW + w_lane;       // staging dword of unit `lane` (b = 0) behind the slice start
    const int wr_lane = w_lane * WIN + wave * SL + yy_lane;                  // its record in the window buffer
    uint32_t sword[CT][NG];
    auto expand = [&](int i, int j0, int buf) {
        if (i >= g.H) return;
        uint4 *win = lds4 + buf * 2 * NG * WIN;
        const int dump = (int)(dump_lds4 - win); // the dump record, as an index into this window buffer
        const int col0 = j0 + slice_col_off;
        const int mis_t = (((i * g.Wt + col0) * NW * 4) & 15) >> 2; // dwords between the first piece's start and the slice
        const bool interior = col0 >= 0 && col0 + SL <= g.Wt;        // (wave uniform) every column of the slice is inside the image
        // Straight-line rounds of UB units: every staging read, then the expansions, then every store (with per-unit branches the
        // compiler serialised the units: five dependent LDS round trips per round, ~3,000 cycles per item in the stamps).  FULL: all
        // 64 UB units of the round exist; otherwise a unit past the slice reads staging dword 0 and stores into the dump record.
        // INTERIOR: no column test.  In a FULL INTERIOR round of the AFFINE layout every address is a per-item base plus a constant.
        auto round = [&](int u0, auto full_c, auto interior_c) {
            constexpr bool FULL = decltype(full_c)::value, INTERIOR = decltype(interior_c)::value;
            uint32_t v[UB];
            int dst[UB];
#pragma unroll
            for (int b = 0; b < UB; b++) {
                const int u = u0 + 64 * b + lane;
                const bool live = FULL || u < SL * NW;
                int rd, yy;
                if constexpr (AFFINE) {
                    rd = mis_t + (fwd ? u0 : -u0) + rd_lane + (fwd ? 64 * b : -64 * b);
                    yy = u0 / NW + (64 / NW) * b + yy_lane;
                    dst[b] = live ? wr_lane + u0 / NW + (64 / NW) * b : dump;
                } else {
                    yy = u / NW;
                    const int w = u - yy * NW;
                    rd = mis_t + (fwd ? u : (SL - 1 - yy) * NW + w);
                    dst[b] = live ? w * WIN + wave * SL + yy : dump;
                }
                const uint32_t staged = stage_t[live ? rd : 0];
                if constexpr (INTERIOR) {
                    v[b] = live ? staged : 0u;
                } else { // the clamp in issue_dma shifts a piece only when it would leave the map: columns outside the image, replaced here
                    const int col = fwd ? col0 + yy : col0 + (SL - 1 - yy);
                    v[b] = (live && col >= 0 && col < g.Wt) ? staged : 0u;
                }
            }
#pragma unroll
            for (int b = 0; b < UB; b++) win[dst[b]] = word_record(v[b]);
        };
        const int n_units = SL * NW, n_full = n_units / (64 * UB);
        if (interior) {
            for (int rr = 0; rr < n_full; rr++) round(rr * 64 * UB, std::true_type{}, std::true_type{});
            if (n_full * 64 * UB < n_units) round(n_full * 64 * UB, std::false_type{}, std::true_type{});
        } else {
            for (int u0 = 0; u0 < n_units; u0 += 64 * UB) round(u0, std::false_type{}, std::false_type{});
        }
        const int mis_s = (((i * g.Ws + j0 + wave * CT * 32) * NW * 4) & 15) >> 2;
#pragma unroll
        for (int cc = 0; cc < CT; cc++) { // per-column state is indexed by the slot cc of the tile loop: column tile cc, or CT - 1 - cc mirrored
            const int c = fwd ? cc : CT - 1 - cc;
#pragma unroll
            for (int gq = 0; gq < NG; gq++)
                sword[cc][gq] = (2 * gq + 1 < NW || h == 0) ? stage_s[mis_s + (c * 32 + xoff) * NW + min(2 * gq + h, NW - 1)] : 0u;
        }
    };

    // ---- once per block: the missing plane of an odd word count, the start patterns
    if constexpr (NW & 1) {
        for (int b = 0; b < plan.nbuf; b++)
            for (int y = threadIdx.x; y < WIN; y += 64 * WAVES) lds4[(b * 2 * NG + NW) * WIN + y] = make_uint4(0u, 0u, 0u, 0u);
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
    issue_dma(q * 8 + xcd, jb * PXB);
    // accumulator starts: a full tile starts at its register number, the two triangular tiles push their invalid cells out of
    // reach.  (The zero is opaque to the compiler on purpose: a start pattern it can rematerialise it rebuilds in the accumulator
    // registers before every tile, eight v_mov_b64; one that lives in registers goes into the MFMA as its C operand.)
    float opaque_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
    v16f full_tile, first_tile, last_tile;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        const int rr = row_of_reg(reg) + 4 * h;
        full_tile[reg] = (float)reg + opaque_zero;
        first_tile[reg] = (float)reg - (rr >= r ? 0.0f : BIG_CELL) + opaque_zero; // tile 0: d = row - col >= 0
        last_tile[reg] = (float)reg - (rr < r ? 0.0f : BIG_CELL) + opaque_zero;   // tile D/32: d = D + row - col < D
    }
    const int pout = (int)Pout;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's staging slices have landed
    expand(q * 8 + xcd, jb * PXB, 0);
    int q_n = q, jb_n = jb; // the item after the current one
    advance(q_n, jb_n);
    if (s + plan.step < plan.items) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the staging reads above have returned
        issue_dma(q_n * 8 + xcd, jb_n * PXB);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier(); // raw: a __syncthreads() would also wait for the DMA just issued
    SVH_STAMP(1);

    const float tag_last = (float)(16 * (NT - 1)); // out of the frame of the last tile: the absolute tag of tile k is 16 k
    const int rbase = 4 * h - r;
    auto decode = [&](float gf) { // (tile, cell) -> key
        const int gv = (int)gf + 256 * B; // exact: 512 (B - cost) + 16 tile + register
        const int reg = gv & 15;
        const int d_local = 2 * (gv & 0x1F0) + (reg & 3) + 2 * (reg & 12) + rbase; // 32 tile + row - col
        const int cost = B - (gv >> 9);
        return gf > G_VALID_ABOVE ? (int)make_key(cost, g.d_offset + d_local) : (int)KEY_NONE;
    };
    auto col_of = [&](int cc) { return fwd ? cc : CT - 1 - cc; }; // column tile of slot cc (wave uniform; address arithmetic only)

    int buf = 0;
    for (;;) {
        const int i = q * 8 + xcd, j0 = jb * PXB;
        const int64_t row = (int64_t)i * g.Ws;
        if (i < g.H) {
            const uint4 *win = lds4 + buf * 2 * NG * WIN;
            struct Frags {
                v8i w[NG];
            };
            // smallest `below` (Ws - d_offset - first pixel of a column tile [- 31 mirrored]: the row-tile position from which cells pay
            // Pout) among this wave's column tiles: when even that one keeps every row tile free of paying cells, the wave runs the plain
            // path -- every item but the ones at the right image border
            const int below_min = g.Ws - g.d_offset - j0 - (wave * CT + CT - 1) * 32 - (fwd ? 0 : 31);
            const bool wave_plain = below_min >= 32 * NT;
            // body of one column tile; PLAIN: no cell pays Pout (one region, no test per tile, one decode)
            auto column_tile = [&](int cc, auto plain) {
                constexpr bool PLAIN = decltype(plain)::value;
                const int ct = wave * CT + col_of(cc);
                const int j = j0 + ct * 32 + xoff;
                // source fragments: this lane's word of every 64-bit group as +-1.0 nibbles
                v8i bfrag[NG];
#pragma unroll
                for (int gq = 0; gq < NG; gq++) {
                    const uint4 rec = word_record(sword[cc][gq]);
                    const bool present = h == 0 || 2 * gq + 1 < NW; // an absent odd word contributes nothing: all-zero nibbles
                    bfrag[gq] = (v8i){present ? (int)rec.x : 0, present ? (int)rec.y : 0, present ? (int)rec.z : 0, present ? (int)rec.w : 0, 0, 0, 0, 0};
                }
                const uint4 *arow = win + h * WIN + (fwd ? ct * 32 : PXB - 32 * (ct + 1)) + r; // row tile k, group gq: arow[2 gq WIN + 32 k]
                auto load_frags = [&](int k) {
                    Frags f;
#pragma unroll
                    for (int gq = 0; gq < NG; gq++) {
                        const uint4 t = arow[2 * gq * WIN + 32 * k];
                        f.w[gq] = (v8i){(int)t.x, (int)t.y, (int)t.z, (int)t.w, 0, 0, 0, 0};
                    }
                    return f;
                };
                auto tile = [&](const Frags &f, const v16f &start) {
                    v16f a = start;
#pragma unroll
                    for (int gq = 0; gq < NG; gq++) // FP4 x FP4, block scales 2^4 (E8M0 131) on both sides: an equal bit adds 256, a differing one -256
                        a = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(f.w[gq], bfrag[gq], a, 4, 4, 0, 131, 0, 131);
                    return a;
                };
                // Region bookkeeping (wave uniform; general path only).  Cell (row tile k, row, col) is disparity 32 k + row - col of its
                // lane's pixel and pays Pout when pixel + global index >= Ws:  32 k + row >= below + slope col, with slope 0 for RightToLeft
                // (the boundary is a target column: a row of the tile) and 2 for LeftToRight (pixel and target column move apart).  Tiles
                // before k_lo hold no paying cell, tiles from k_hi on only paying cells, the tiles between (at most one, LeftToRight
                // three) are masked per cell.
                const int slope = fwd ? 0 : 2;
                const int below = PLAIN ? 0 : g.Ws - g.d_offset - j0 - ct * 32 - (fwd ? 0 : 31);
                const int k_lo = PLAIN ? NT : (below <= 0 ? 0 : min(below >> 5, NT));
                const int all_from = below + 31 * slope; // 32 k >= this: every cell pays
                const int k_hi = PLAIN ? NT : (all_from <= 0 ? 0 : min((all_from + 31) >> 5, NT));
                const bool none_pays = PLAIN || k_lo >= NT, all_pay = !PLAIN && k_hi <= 0; // (wave uniform)
                float G = -BIG_G, G0 = -BIG_G, Hpay = -BIG_G; // running maximum (frame of the current tile); closed no-Pout region; Pout region inside the masked tiles
                // `chk`: this column tile has the Pout boundary inside its row tiles (only near the right image border); the others
                // run the loop without the per-tile region test
                auto epilogue = [&](const v16f &a, int k, auto chk) {
                    if constexpr (decltype(chk)::value) {
                        if (k >= k_lo && k <= k_hi) { // around the boundary: a few tiles per column tile
                            asm volatile("; region boundary" ::: "memory"); // keep this a branch: if-converted it costs selects in every tile
                            if (k < k_hi) {
                                float m0 = -BIG_CELL, m1 = -BIG_CELL;
                                int below_lane = below + slope * r;
                                asm volatile("; masked tile" : "+v"(below_lane)); // or the 16 compares below are hoisted in front of every column tile
#pragma unroll
                                for (int reg = 0; reg < 16; reg++) {
                                    const bool pays = 32 * k + row_of_reg(reg) + 4 * h >= below_lane;
                                    m0 = __builtin_fmaxf(m0, pays ? -BIG_CELL : a[reg]);
                                    m1 = __builtin_fmaxf(m1, pays ? a[reg] : -BIG_CELL);
                                }
                                G = __builtin_fmaxf(G - 16.0f, m0);       // both regions keep their running maximum, frame of tile k
                                Hpay = __builtin_fmaxf(Hpay - 16.0f, m1);
                                return;
                            }
                            // k == k_hi: the first tile that pays everywhere.  The no-Pout region is closed (frame of tile k - 1: absolute
                            // tag); the Pout region goes on from what the masked tiles found
                            G0 = G + (float)(16 * (k - 1));
                            G = Hpay;
                        }
                    }
                    const float t1 = max3f(a[0], a[1], a[2]), t2 = max3f(a[3], a[4], a[5]), t3 = max3f(a[6], a[7], a[8]), t4 = max3f(a[9], a[10], a[11]),
                                t5 = max3f(a[12], a[13], a[14]);
                    const float t6 = max3f(t1, t2, t3), t7 = max3f(t4, t5, a[15]);
                    G = max3f(G - 16.0f, t6, t7); // frame of tile k: max over the tiles so far of (tile maximum - 16 (k - tile))
                };
                // NT >= 2: tile 0 and tile NT - 1 are the triangular ones, the loop body only meets full tiles.  Software pipeline:
                // the fragments of tile k + 1 are in registers and the MFMAs of tile k in flight while the epilogue of tile k - 1
                // runs, so neither the LDS latency nor the matrix pipe's sits between two epilogues of a wave.
                auto sweep_tiles = [&](auto chk) {
                    Frags f_next = load_frags(0);
                    v16f acc_a = tile(f_next, first_tile), acc_b;
                    f_next = load_frags(1);
                    int k = 0; // invariant: acc_a = tile k (in flight), f_next = fragments of tile k + 1
                    for (; k + 4 < NT - 1; k += 4) { // (four tiles per trip: half the loop bookkeeping of two)
                        acc_b = tile(f_next, full_tile);
                        f_next = load_frags(k + 2);
                        epilogue(acc_a, k, chk);
                        acc_a = tile(f_next, full_tile);
                        f_next = load_frags(k + 3);
                        epilogue(acc_b, k + 1, chk);
                        acc_b = tile(f_next, full_tile);
                        f_next = load_frags(k + 4);
                        epilogue(acc_a, k + 2, chk);
                        acc_a = tile(f_next, full_tile);
                        f_next = load_frags(k + 5);
                        epilogue(acc_b, k + 3, chk);
                    }
                    for (; k + 2 < NT - 1; k += 2) {
                        acc_b = tile(f_next, full_tile);
                        f_next = load_frags(k + 2);
                        epilogue(acc_a, k, chk);
                        acc_a = tile(f_next, full_tile);
                        f_next = load_frags(k + 3);
                        epilogue(acc_b, k + 1, chk);
                    }
                    if (k + 2 == NT - 1) {
                        acc_b = tile(f_next, full_tile);
                        f_next = load_frags(k + 2);
                        epilogue(acc_a, k, chk);
                        acc_a = tile(f_next, last_tile);
                        epilogue(acc_b, k + 1, chk);
                        epilogue(acc_a, k + 2, chk);
                    } else { // k + 1 == NT - 1
                        acc_b = tile(f_next, last_tile);
                        epilogue(acc_a, k, chk);
                        epilogue(acc_b, k + 1, chk);
                    }
                };
                int a0, a1;
                if constexpr (PLAIN) {
                    sweep_tiles(std::false_type{});
                    a0 = decode(G + tag_last);
                    a0 = min(a0, __shfl_xor(a0, 32)); // the two lane halves hold different rows of the same pixel
                    a1 = (int)KEY_NONE;
                } else {
                    if (none_pays || all_pay) sweep_tiles(std::false_type{}); // one region: no test inside
                    else sweep_tiles(std::true_type{});
                    float G1;
                    if (none_pays) { // everything ran into G
                        G0 = G + tag_last;
                        G1 = -BIG_G;
                    } else if (k_hi >= NT) { // the masked tiles reach the end of the range: both regions still open
                        G0 = G + tag_last;
                        G1 = Hpay + tag_last;
                    } else { // G0 was closed at k_hi (all_pay: it is still -BIG_G), G has been the Pout region since
                        G1 = G + tag_last;
                    }
                    a0 = decode(G0);
                    a1 = decode(G1);
                    a0 = min(a0, __shfl_xor(a0, 32));
                    a1 = min(a1, __shfl_xor(a1, 32));
                }
                if (g.region1_global_last >= 0) { // disparity shards (svh_internal.h, CostSource): the Pout region's winner over all shards
                    int ones = 0; // |s|: each lane half holds its words of the pixel
#pragma unroll
                    for (int gq = 0; gq < NG; gq++) ones += (2 * gq + 1 < NW || h == 0) ? __popc(sword[cc][gq]) : 0;
                    ones += __shfl_xor(ones, 32);
                    a1 = (int)global_region1_key(g, j, ones);
                }
                if (h == 0 && j < g.Ws) {
                    if constexpr (WINNER) { // the winner itself: census_keys.h
                        // `sw` is read from the kernel argument segment HERE (an opaque pointer: the loads cannot be hoisted), not
                        // kept in a dozen scalar registers through the tile loop, which has none to spare
                        typedef __attribute__((address_space(4))) const char *KernArg;
                        KernArg ka = (KernArg)__builtin_amdgcn_kernarg_segment_ptr();
                        asm volatile("" : "+s"(ka));
                        typedef __attribute__((address_space(4))) const uint32_t *KernArgWords;
                        KernArgWords kw = (KernArgWords)(ka + offsetof(SweepKernelArgs, sw));
                        static_assert(sizeof(SweepWinner) % 4 == 0, "read as dwords");
                        uint32_t raw[sizeof(SweepWinner) / 4];
#pragma unroll
                        for (unsigned q = 0; q < sizeof(SweepWinner) / 4; q++) raw[q] = kw[q];
                        SweepWinner w;
                        __builtin_memcpy(&w, raw, sizeof w);
                        const int bd = winner_index((uint32_t)a0, (uint32_t)a1, passes_visiting(w, i, j), w.pout);
                        if ((unsigned)(i - w.row_first) < (unsigned)w.row_count) {
                            const int64_t o = (int64_t)(i - w.row_first) * g.Ws + j;
                            if (w.idx) w.idx[o] = bd;
                            if (w.disp) w.disp[o] = w.disp_sign * bd + w.disp_offset;
                        }
                    } else {
                        keys[row + j] = make_uint2((uint32_t)a0, (uint32_t)a1);
                        if (gmap) gmap[row + j] = g_from_keys((uint32_t)a0, (uint32_t)a1, pout);
                    }
                }
            };
            if (wave_plain) {
#pragma unroll
                for (int cc = 0; cc < CT; cc++) column_tile(cc, std::true_type{});
            } else {
#pragma unroll
                for (int cc = 0; cc < CT; cc++) column_tile(cc, std::false_type{});
            }
        }
#ifdef SVH_SWEEP_STAMPS
        if (n_items_done == 1) SVH_STAMP(3);
#endif
        s += plan.step;
        if (s >= plan.items) break; // (block uniform)
        q = q_n;
        jb = jb_n;
        advance(q_n, jb_n);
        // the next item: its words landed while the tiles ran (the wait also covers this wave's key stores)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SVH_SWEEP_STAMPS
        if (n_items_done == 1) SVH_STAMP(4);
#endif
        if (plan.nbuf == 1) __builtin_amdgcn_s_barrier(); // one window buffer: everybody is done reading it
        expand(q * 8 + xcd, jb * PXB, plan.nbuf == 2 ? buf ^ 1 : 0);
#ifdef SVH_SWEEP_STAMPS
        if (n_items_done == 1) SVH_STAMP(5);
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // staging reads returned (the slices are about to be overwritten), window writes done
#ifdef SVH_SWEEP_STAMPS
        if (n_items_done == 1) SVH_STAMP(6);
#endif
        if (s + plan.step < plan.items) issue_dma(q_n * 8 + xcd, jb_n * PXB);
#ifdef SVH_SWEEP_STAMPS
        if (n_items_done == 1) SVH_STAMP(7);
#endif
        __builtin_amdgcn_s_barrier();
#ifdef SVH_SWEEP_STAMPS
        if (n_items_done == 0) SVH_STAMP(2);
        n_items_done++;
#endif
        if (plan.nbuf == 2) buf ^= 1;
    }
#ifdef SVH_SWEEP_STAMPS
    if (lane == 0) {
        unsigned long long *o = stamp_out + ((int64_t)blockIdx.x * WAVES + wave) * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) o[k] = stamp[k];
    }
#endif
}

// blocks per CU and grid: persistent blocks, as many as fit (LDS, 3 waves per SIMD by registers), a multiple of 8 so that every
// XCD list gets the same number of walkers
inline SweepPlan make_plan(int H, int Ws, int pxb, int nw, int waves, int ct, int D, int n_cu, size_t *shmem, int *grid) {
    SweepPlan p;
    p.bpr = ceil_div(Ws, pxb);
    p.items = ceil_div(H, 8) * p.bpr;
    p.nbuf = pm_lds_bytes(nw, waves, ct, D, 2) <= 160 * 1024 ? 2 : 1;
    *shmem = pm_lds_bytes(nw, waves, ct, D, p.nbuf);
    int per_cu = (int)std::min<size_t>(pm_blocks_per_cu(nw, ct), (160 * 1024) / *shmem);
    if (per_cu < 1) per_cu = 1;
    int g = n_cu * per_cu;
    g = std::min(g, p.items * 8);
    g = std::max(8, (g / 8) * 8);
    p.step = g / 8;
    p.step_q = p.step / p.bpr;
    p.step_j = p.step % p.bpr;
    *grid = g;
    return p;
}

#ifndef SVH_SWEEP_STAMPS
template <int NW, int WAVES, int CT> int launch_config(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, const SweepWinner &sw) {
    constexpr int PXB = WAVES * CT * 32;
    static int n_cu[64] = {};
    const int dev = ctx->device & 63;
    if (!n_cu[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || n <= 0) n = 256;
        n_cu[dev] = n;
    }
    size_t shmem;
    int grid;
    const SweepPlan plan = make_plan(g.H, g.Ws, PXB, NW, WAVES, CT, g.D, n_cu[dev], &shmem, &grid);
    static int attr_set_dev[64] = {}; // (per instantiation and device; concurrent per-thread contexts: atomics)
    if (!__atomic_load_n(&attr_set_dev[dev], __ATOMIC_ACQUIRE)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&census_sweep_pm_kernel<NW, WAVES, CT, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&census_sweep_pm_kernel<NW, WAVES, CT, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return fail(ctx, SVH_ERR_HIP, "census_sweep (mfma, fp4 +-1): cannot raise the dynamic LDS limit");
        __atomic_store_n(&attr_set_dev[dev], 1, __ATOMIC_RELEASE);
    }
    if (sw.on()) SVH_LAUNCH(ctx, "census_sweep_pm", (census_sweep_pm_kernel<NW, WAVES, CT, true>), grid, 64 * WAVES, shmem, g, Pout, keys, gmap, plan, sw);
    else SVH_LAUNCH(ctx, "census_sweep_pm", (census_sweep_pm_kernel<NW, WAVES, CT, false>), grid, 64 * WAVES, shmem, g, Pout, keys, gmap, plan, sw);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SVH_OK : fail(ctx, SVH_ERR_HIP, "census_sweep (mfma, fp4 +-1): %s", hipGetErrorString(e));
}

template <int NW> bool launch_for_words(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, int *status, const SweepWinner &sw) {
    // item width (4 waves x CT column tiles of 32 pixels): the one that pads the row least, the wider on a tie
    int best_ct = 0;
    int64_t best_pad = 0;
    // (five to eight census words: three or four MFMAs' worth of fragments per column tile -- two column tiles per wave keep them in registers)
    for (int ct = NW > 4 ? 2 : 4; ct >= 2; ct--) {
        const int pxb = 4 * ct * 32;
        if (pm_lds_bytes(NW, 4, ct, g.D, 1) > 160 * 1024) continue;
        const int64_t pad = (int64_t)ceil_div(g.Ws, pxb) * pxb;
        if (!best_ct || pad < best_pad) {
            best_ct = ct;
            best_pad = pad;
        }
    }
    if (!best_ct) return false;
    if constexpr (NW > 4) {
        *status = launch_config<NW, 4, 2>(ctx, g, Pout, keys, gmap, sw);
    } else {
        switch (best_ct) {
        case 4: *status = launch_config<NW, 4, 4>(ctx, g, Pout, keys, gmap, sw); break;
        case 3: *status = launch_config<NW, 4, 3>(ctx, g, Pout, keys, gmap, sw); break;
        default: *status = launch_config<NW, 4, 2>(ctx, g, Pout, keys, gmap, sw); break;
        }
    }
    return true;
}
#endif

} // namespace

#ifndef SVH_SWEEP_STAMPS
bool launch_sweep_pm(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, int *status, const SweepWinner *winner) {
    if (ctx->census_sweep_rl && launch_sweep_rl(ctx, g, Pout, keys, gmap, status, winner)) return true;
    const SweepWinner sw = winner ? *winner : SweepWinner();
    if (sw.on() && g.region1_global_last >= 0) return false; // (disparity shards exchange keys)
    if (g.D < 32 || (g.D & 31) != 0 || g.D > 992 || g.nWw < 1 || g.nWw > 8) return false;
    // the staging DMA addresses the word maps with byte offsets held in 32-bit-safe ranges
    if ((int64_t)g.H * g.Wt * g.nWw * 4 >= (int64_t)1 << 31 || (int64_t)g.H * g.Ws * g.nWw * 4 >= (int64_t)1 << 31) return false;
    if ((int64_t)g.H * g.Wt * g.nWw * 4 < 16 || (int64_t)g.H * g.Ws * g.nWw * 4 < 16) return false;
    switch (g.nWw) {
    case 1: return launch_for_words<1>(ctx, g, Pout, keys, gmap, status, sw);
    case 2: return launch_for_words<2>(ctx, g, Pout, keys, gmap, status, sw);
    case 3: return launch_for_words<3>(ctx, g, Pout, keys, gmap, status, sw);
    case 4: return launch_for_words<4>(ctx, g, Pout, keys, gmap, status, sw);
    case 5: return launch_for_words<5>(ctx, g, Pout, keys, gmap, status, sw);
    case 6: return launch_for_words<6>(ctx, g, Pout, keys, gmap, status, sw);
    case 7: return launch_for_words<7>(ctx, g, Pout, keys, gmap, status, sw);
    default: return launch_for_words<8>(ctx, g, Pout, keys, gmap, status, sw);
    }
}
#endif

} // namespace svh

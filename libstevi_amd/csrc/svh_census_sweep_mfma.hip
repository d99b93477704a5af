// census_sweep on the matrix cores.
//
// The sweep of svh_census_sgm.hip evaluates  c(j, d) = popcount(s_j xor t_{j + lower + d})  for every pixel j of a row and every
// disparity d, and keeps per pixel the minimum key (c << 12 | 4095 - d) of the two disparity regions (before / from the column
// where the reference adds Pout, sgm.h:287-289).  On the vector ALU that is 2 v_xor + 2 v_bcnt + v_lshl_or + 1/2 v_min3 per 64
// voxels of 64-bit census records: 21.4 issue cycles per SIMD, and the kernel sits at that issue bound, far
// from HBM (67 MB per launch).  The Hamming distance of two bit strings is also a dot product:
//
//     popcount(s xor t) = |s| + sum_b t_b (1 - 2 s_b)
//
// so with the target bits as 0 / 4 bytes (operand A: rows = 32 target columns) and the source bits as +64 / -64 bytes (operand B:
// columns = 32 source pixels) one v_mfma_i32_32x32x32_i8 per 32 bits of census record yields 32 x 32 costs (times 256, minus
// |s|, a per-lane constant).  In the accumulator layout a lane owns ONE source pixel and 16 target columns, so the running minimum over
// disparities stays inside the lane, and with the cell index riding in the accumulator's start value (see the kernel) the
// minimum of 1024 voxels is eight v_min3 and one add per tile instead of 16 x 5.5 VALU instructions; the matrix pipe runs beside
// them.  Integer arithmetic throughout: results are the VALU sweep's, bit for bit (the parity tests run both,
// svh_context_set_option("census_sweep", ...)).  This is the int8 form (engine 2); svh_census_sweep_mfma4.hip is the same kernel with
// FP4 operands (engine 3, the default: half the matrix-pipe time, half the LDS window).
//
// Geometry.  A wave owns CT column tiles (of 32 source pixels) and takes them one after the other; column tile c meets the row
// tiles k = 0 .. D/32 of the target window, row tile k starting at target column  J0c + lower + 32 k, i.e. d = 32 k + row - col: tile 0 is valid on and below its
// diagonal, tile D/32 strictly above it, the tiles between completely.  The two triangular tiles start their accumulator from a
// pattern that holds 2^20 in the invalid cells (they can then never win).  The Pout boundary
// j + d_global >= Ws is the TARGET column Ws + lower - d_offset: a row boundary, so a tile lies in one region or (rarely)
// straddles it.  Covers sign > 0 (RightToLeft), 1..4 census words, D a multiple of 32 up to 480; everything else runs the VALU sweep.
#include <type_traits>

#include "svh_census_keys.h"

namespace svh {

namespace {

using v4i = int __attribute__((ext_vector_type(4)));
using v16i = int __attribute__((ext_vector_type(16)));

constexpr int BIG_CELL = 1 << 20;       // accumulator start of an invalid cell: never the minimum of a tile that has a valid one
constexpr int BIG_G = 1 << 28;          // "no candidate yet"
constexpr int G_VALID_BELOW = 1 << 19;  // valid packed minima stay below 2^15

// bits 4q .. 4q+3 of w as four 0/1 bytes (bit b of the nibble in byte b), and the same as 0/4 bytes (the largest scale whose
// multiplier still fits v_mul_u32_u24)
__device__ __forceinline__ uint32_t nibble_bytes(uint32_t w, int q) { return (((w >> (4 * q)) & 0xFu) * 0x00204081u) & 0x01010101u; }
__device__ __forceinline__ uint32_t nibble_bytes_x4(uint32_t w, int q) { return (((w >> (4 * q)) & 0xFu) * 0x00810204u) & 0x04040404u; }

// min of three, written so that it selects v_min3_i32.  Deliberately NOT inline asm: the tree below is the first reader of the
// MFMA result, and the compiler only pads the MFMA -> VALU read hazard for instructions it can see (an asm reader got stale cells).
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }

__device__ __forceinline__ constexpr int row_of_reg(int reg) { return (reg & 3) + 8 * (reg >> 2); } // + 4 * (lane >> 5)

// Cell value.  The target bytes are 0 / 4, the source bytes +64 / -64 and the accumulator starts at 15 - reg, so a cell comes out
// of the matrix pipe as
//     a[reg] = 256 (c - |s|) + 15 - reg
// which already orders the 16 cells of a lane the way the reference breaks ties (the larger row, i.e. the larger disparity, wins):
// the tile minimum is a plain min tree, no per-cell instruction.  Row tile k carries the tag 16 (15 - k) (bits 4..7, between the
// cost and the cell: cost first, then the later tile, then the later cell); the running minimum lives in the frame of the current
// tile (G <- min(G + 16, tile minimum): one add per tile) and is decoded once per pixel at the end.  Four tag bits: D <= 480.
template <int NW, int WAVES, int CT>
__global__ void __launch_bounds__(64 * WAVES) census_sweep_mfma_kernel(CensusGeom g, float Pout, uint2 *__restrict__ keys, float *__restrict__ gmap) {
    constexpr int PXB = WAVES * CT * 32; // source pixels per block
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[]; // [2 NW chunks of 16 bits][WIN window pixels] x 16 bytes
    const int i = blockIdx.y, j0 = blockIdx.x * PXB;
    const int NT = g.D / 32 + 1;  // row tiles per column tile
    const int WIN = PXB + g.D;    // window pixels: the last column tile's last row tile ends at PXB - 32 + 32 NT
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int64_t row = (int64_t)i * g.Ws;
    // ---- every global load of the kernel up front: the source words of this wave's column tiles, then the target window
    uint32_t sword[CT][NW];
#pragma unroll
    for (int c = 0; c < CT; c++) {
        const int j = j0 + (wave * CT + c) * 32 + r;
        const uint32_t *sp = g.sw + (row + min(j, g.Ws - 1)) * NW;
#pragma unroll
        for (int m = 0; m < NW; m++) sword[c][m] = sp[m];
    }
    // ---- byte -> eight bytes tables: bit b of the byte as 0 / 4 (target) and as +64 / -64 (source).  Expanding a word through
    // them costs 8 VALU instructions and 4 LDS reads instead of 24 VALU instructions.
    uint2 *lut_t = reinterpret_cast<uint2 *>(lds4 + 2 * NW * WIN + 512), *lut_s = lut_t + 256;
    for (int b = threadIdx.x; b < 256; b += 64 * WAVES) {
        lut_t[b] = make_uint2(nibble_bytes_x4(b, 0), nibble_bytes_x4(b, 1));
        lut_s[b] = make_uint2(__builtin_amdgcn_perm(0u, 0x0000C040u, nibble_bytes(b, 0)), __builtin_amdgcn_perm(0u, 0x0000C040u, nibble_bytes(b, 1)));
    }
    // ---- the target window as 0/4 bytes; window pixel y is target column j0 + lower + y (zero vector outside the image).  A thread
    // takes whole pixels (all NW words: one address, one bounds test), PPT of them with every load issued before the first table
    // look-up; the barrier that completes the tables sits under those loads.
    {
        const uint32_t *trow = g.tw + (int64_t)i * g.Wt * NW;
        constexpr int PPT = 4; // pixels per thread and round
        for (int y0 = threadIdx.x; y0 < WIN; y0 += PPT * 64 * WAVES) { // (first round: every thread enters, WIN > 64 WAVES)
            uint32_t w[PPT][NW];
#pragma unroll
            for (int u = 0; u < PPT; u++) {
                const int y = y0 + u * 64 * WAVES, jt = j0 + g.disp_lower + y;
                const bool inside = y < WIN && jt >= 0 && jt < g.Wt;
                const uint32_t *tp = trow + (int64_t)(inside ? jt : 0) * NW;
#pragma unroll
                for (int m = 0; m < NW; m++) w[u][m] = inside ? tp[m] : 0u;
            }
            if (y0 == (int)threadIdx.x) __syncthreads(); // first round (uniform): the tables are complete
#pragma unroll
            for (int u = 0; u < PPT; u++) {
                const int y = y0 + u * 64 * WAVES;
                if (y < WIN) {
#pragma unroll
                    for (int m = 0; m < NW; m++) {
                        const uint32_t v = w[u][m];
                        const uint2 b0 = lut_t[v & 0xFFu], b1 = lut_t[(v >> 8) & 0xFFu], b2 = lut_t[(v >> 16) & 0xFFu], b3 = lut_t[v >> 24];
                        lds4[(2 * m) * WIN + y] = make_uint4(b0.x, b0.y, b1.x, b1.y);
                        lds4[(2 * m + 1) * WIN + y] = make_uint4(b2.x, b2.y, b3.x, b3.y);
                    }
                }
            }
        }
    }
    // ---- accumulator starts: 15 - reg for full tiles (in registers), and for the two triangular tiles the same with the invalid
    // cells pushed out of reach (in LDS behind the window: read twice per column tile)
    // (the zero is opaque to the compiler on purpose: a start pattern it can rematerialise it rebuilds in the accumulator registers
    // before every tile, eight v_mov_b64; one that lives in registers goes into the MFMA as its C operand)
    int opaque_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
    v16i full_tile;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) full_tile[reg] = 15 - reg + opaque_zero;
    uint4 *edge = lds4 + 2 * NW * WIN; // [first, last][4 register quads][64 lanes]
    for (int q = wave; q < 8; q += WAVES) { // eight register quads (first tile 0..3, last tile 4..7) shared out over the waves
        int v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int reg = 4 * (q & 3) + e, rr = row_of_reg(reg) + 4 * h;
            const bool valid = q < 4 ? rr >= r   // tile 0: d = row - col >= 0
                                     : rr < r;   // tile D/32: d = D + row - col < D
            v[e] = 15 - reg + (valid ? 0 : BIG_CELL);
        }
        edge[q * 64 + lane] = make_uint4(v[0], v[1], v[2], v[3]);
    }
    auto edge_tile = [&](int which) {
        v16i a;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 t = edge[(4 * which + q) * 64 + lane];
            a[4 * q] = (int)t.x;
            a[4 * q + 1] = (int)t.y;
            a[4 * q + 2] = (int)t.z;
            a[4 * q + 3] = (int)t.w;
        }
        return a;
    };
    const int thr = g.Ws + g.disp_lower - g.d_offset; // target columns from here on pay Pout
    const int pout = (int)Pout;
    __syncthreads();

    // one column tile after the other; the MFMAs of row tile k + 1 are issued before the epilogue of row tile k
    for (int c = 0; c < CT; c++) {
        const int ct = wave * CT + c; // column tile of the block (wave uniform)
        const int j = j0 + ct * 32 + r;
        // source fragment (+64 / -64 bytes of this lane's half of every word) and |s|
        v4i bfrag[NW];
        int ones = 0;
#pragma unroll
        for (int m = 0; m < NW; m++) {
            uint32_t w = sword[0][m];
#pragma unroll
            for (int cc = 1; cc < CT; cc++) w = c == cc ? sword[cc][m] : w; // c is wave uniform: scalar selects, no indexed registers
            ones += __popc(w);
            const uint32_t half = w >> (16 * h);
            const uint2 lo = lut_s[half & 0xFFu], hi = lut_s[(half >> 8) & 0xFFu];
            bfrag[m] = (v4i){(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
        }
        const uint4 *arow = lds4 + h * WIN + ct * 32 + r; // row tile k, word m: arow[2 m WIN + 32 k]
        struct Frags {
            v4i w[NW];
        };
        auto load_frags = [&](int k) {
            Frags f;
#pragma unroll
            for (int m = 0; m < NW; m++) {
                const uint4 t = arow[2 * m * WIN + 32 * k];
                f.w[m] = (v4i){(int)t.x, (int)t.y, (int)t.z, (int)t.w};
            }
            return f;
        };
        auto tile = [&](const Frags &f, const v16i &start) {
            v16i a = start;
#pragma unroll
            for (int m = 0; m < NW; m++) a = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.w[m], bfrag[m], a, 0, 0, 0);
            return a;
        };
        // Region bookkeeping (wave uniform).  Row tile k covers target columns t0 + 32 k ...; tiles below k_switch lie in the
        // region without Pout; tile k_switch straddles the boundary when `straddle`, else it is the first tile that pays.
        const int t0 = j0 + g.disp_lower + ct * 32;
        const int below = thr - t0; // target rows (relative to tile 0) that do not pay
        const int k_switch = below <= 0 ? 0 : min(below >> 5, NT);
        const bool straddle = below > 0 && (below & 31) != 0 && k_switch < NT;
        int G = BIG_G, G0 = BIG_G;
        // `checked`: this column tile has the Pout boundary inside its row tiles (only near the right image border); the others
        // run the loop without the per-tile region test
        auto epilogue = [&](const v16i &a, int k, auto checked) {
            if constexpr (decltype(checked)::value) {
                if (k == k_switch) { // once per column tile
                    asm volatile("; region switch" ::: "memory"); // keep this a branch: if-converted it costs selects in every tile
                    if (straddle) {
                        int m0 = BIG_CELL, m1 = BIG_CELL;
                        int below_here = below;
                        asm volatile("; straddling tile" : "+s"(below_here)); // or the 16 compares below are hoisted in front of every column tile
#pragma unroll
                        for (int reg = 0; reg < 16; reg++) {
                            const bool pays = 32 * k + row_of_reg(reg) + 4 * h >= below_here;
                            m0 = min(m0, pays ? BIG_CELL : a[reg]);
                            m1 = min(m1, pays ? a[reg] : BIG_CELL);
                        }
                        G0 = min(G + 16, m0) + 16 * (15 - k); // leaves the frame: absolute tile tag
                        G = m1;
                        return;
                    }
                    G0 = G + 16 * (15 - (k - 1)); // (frame of tile k - 1; k_switch = 0 never gets here with anything but BIG_G)
                    G = BIG_G;
                }
            }
            const int t1 = min3i(a[0], a[1], a[2]), t2 = min3i(a[3], a[4], a[5]), t3 = min3i(a[6], a[7], a[8]), t4 = min3i(a[9], a[10], a[11]),
                      t5 = min3i(a[12], a[13], a[14]);
            const int t6 = min3i(t1, t2, t3), t7 = min3i(t4, t5, a[15]);
            G = min3i(G + 16, t6, t7); // frame of tile k: min over the tiles so far of (tile minimum + 16 (k - tile))
        };
        // NT >= 2: tile 0 and tile NT - 1 are the triangular ones, the loop body only meets full tiles.  Software pipeline: the
        // fragments of tile k + 1 are in registers and the MFMAs of tile k in flight while the epilogue of tile k - 1 runs, so
        // neither the LDS latency nor the matrix pipe's sits between two epilogues of a wave.
        auto sweep_tiles = [&](auto checked) {
            Frags f_next = load_frags(0);
            v16i acc_a = tile(f_next, edge_tile(0)), acc_b;
            f_next = load_frags(1);
            int k = 0; // invariant: acc_a = tile k (in flight), f_next = fragments of tile k + 1
            for (; k + 2 < NT - 1; k += 2) {
                acc_b = tile(f_next, full_tile);
                f_next = load_frags(k + 2);
                epilogue(acc_a, k, checked);
                acc_a = tile(f_next, full_tile);
                f_next = load_frags(k + 3);
                epilogue(acc_b, k + 1, checked);
            }
            if (k + 2 == NT - 1) {
                acc_b = tile(f_next, full_tile);
                f_next = load_frags(k + 2);
                epilogue(acc_a, k, checked);
                acc_a = tile(f_next, edge_tile(1));
                epilogue(acc_b, k + 1, checked);
                epilogue(acc_a, k + 2, checked);
            } else { // k + 1 == NT - 1
                acc_b = tile(f_next, edge_tile(1));
                epilogue(acc_a, k, checked);
                epilogue(acc_b, k + 1, checked);
            }
        };
        if (k_switch >= NT || (k_switch == 0 && !straddle)) sweep_tiles(std::false_type{}); // one region: no test inside
        else sweep_tiles(std::true_type{});
        // out of the frame (tile NT - 1): the absolute tag of tile k is 16 (15 - k)
        const int G_abs = G + 16 * (15 - (NT - 1));
        const bool none_pays = k_switch >= NT, all_pay = k_switch == 0 && !straddle; // (wave uniform)
        const int G1 = none_pays ? BIG_G : G_abs;
        if (none_pays) G0 = G_abs;
        // ---- decode (tile, cell) -> key; the two lane halves hold different rows of the same pixel: merge, store
        auto decode = [&](int gv) {
            const int kk = 15 - ((gv >> 4) & 15);
            const int reg = 15 - (gv & 15), cost = ones + (gv >> 8);
            const int d_local = 32 * kk + (reg & 3) + 8 * (reg >> 2) + 4 * h - r;
            return gv < G_VALID_BELOW ? (int)make_key(cost, g.d_offset + d_local) : (int)KEY_NONE;
        };
        int a0 = all_pay ? (int)KEY_NONE : decode(G0), a1 = none_pays ? (int)KEY_NONE : decode(G1);
        a0 = min(a0, __shfl_xor(a0, 32));
        a1 = min(a1, __shfl_xor(a1, 32));
        if (g.region1_global_last >= 0) a1 = (int)global_region1_key(g, j, ones); // disparity shards (svh_internal.h, CostSource)
        if (h == 0 && j < g.Ws) {
            keys[row + j] = make_uint2((uint32_t)a0, (uint32_t)a1);
            if (gmap) gmap[row + j] = g_from_keys((uint32_t)a0, (uint32_t)a1, pout);
        }
    }
}

template <int NW, int WAVES, int CT> int launch_config(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, size_t shmem) {
    constexpr int PXB = WAVES * CT * 32;
    dim3 grid(ceil_div(g.Ws, PXB), g.H);
    SVH_LAUNCH(ctx, "census_sweep_mfma", (census_sweep_mfma_kernel<NW, WAVES, CT>), grid, 64 * WAVES, shmem, g, Pout, keys, gmap);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SVH_OK : fail(ctx, SVH_ERR_HIP, "census_sweep (mfma): %s", hipGetErrorString(e));
}

template <int NW> bool launch_for_words(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, int *status) {
    // block width (4 waves x CT column tiles of 32 pixels): the widest that fits 64 KB of LDS, preferring less row padding
    int best_ct = 0;
    int64_t best_pad = 0;
    for (int ct = 4; ct >= 2; ct--) {
        const int pxb = 4 * ct * 32;
        if (((size_t)2 * NW * (pxb + g.D) + 512 + 256) * sizeof(uint4) > 64 * 1024) continue;
        const int64_t pad = (int64_t)ceil_div(g.Ws, pxb) * pxb;
        if (!best_ct || pad < best_pad) {
            best_ct = ct;
            best_pad = pad;
        }
    }
    if (!best_ct) return false;
    const size_t shmem = ((size_t)2 * NW * (4 * best_ct * 32 + g.D) + 512 + 256) * sizeof(uint4); // window + the two edge patterns + the byte tables
    switch (best_ct) {
    case 4: *status = launch_config<NW, 4, 4>(ctx, g, Pout, keys, gmap, shmem); break;
    case 3: *status = launch_config<NW, 4, 3>(ctx, g, Pout, keys, gmap, shmem); break;
    default: *status = launch_config<NW, 4, 2>(ctx, g, Pout, keys, gmap, shmem); break;
    }
    return true;
}

} // namespace

bool launch_sweep_mfma(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, int *status) {
    if (g.sign <= 0 || g.D < 32 || (g.D & 31) != 0 || g.D > 480 || g.nWw < 1 || g.nWw > 4) return false;
    switch (g.nWw) {
    case 1: return launch_for_words<1>(ctx, g, Pout, keys, gmap, status);
    case 2: return launch_for_words<2>(ctx, g, Pout, keys, gmap, status);
    case 3: return launch_for_words<3>(ctx, g, Pout, keys, gmap, status);
    default: return launch_for_words<4>(ctx, g, Pout, keys, gmap, status);
    }
}

} // namespace svh

// census_sweep on the matrix cores, 4-bit operands.
//
// Same kernel as svh_census_sweep_mfma.hip (read that file's header first: Hamming distance as a dot product, the band of row
// tiles, the cell index in the accumulator's start value, the moving frame of the tile tag), with the bits expanded to FP4 (e2m1)
// nibbles instead of int8 bytes and one v_mfma_scale_f32_32x32x64_f8f6f4 per 64 bits of census record:
//
//     target bit t -> 0 / +1.0 (nibble 0x0 / 0x2),   source bit s -> +1.0 / -1.0 (0x2 / 0xA),   block scales 2^3 and 2^6
//     cell = 512 (c - |s|) + 15 - reg      as an exact float (all values are integers far below 2^24)
//
// (the block scales make any power of two free, so the tile tag gets five bits here: 16 (31 - k), D up to 992)
//
// What it buys: the LDS window is half the size (16 bytes per pixel and census word: four blocks per CU instead of three), a
// tile needs one MFMA and one ds_read_b128 instead of two each, and the expansion writes half the bytes.  The minimum tree works
// on floats (v_min3_f32); the decode converts once per pixel.  Results are the int8 kernel's and the vector-ALU sweep's bit for
// bit (tests/test_gpu_sweep_engines.py).  The operand layout (lane half h holds the k-subset [32 h, 32 h + 32) of its row or
// column as nibble e of dword q = element 8 q + e) was pinned with exact integer data before use.
//
// Unlike the int8 form this kernel also takes LeftToRight (target column j - lower - d): the window is staged mirrored and a lane's
// pixel is the mirror image of its column index, so that the window position still grows with d and everything above holds; only
// the Pout boundary changes -- it advances by two cells per lane instead of being a row of the tile -- and is handled by masking
// the (up to three) row tiles it crosses per cell.
#include <type_traits>

#include "svh_census_keys.h"

namespace svh {

namespace {

using v8i = int __attribute__((ext_vector_type(8)));
using v16f = float __attribute__((ext_vector_type(16)));

constexpr float BIG_CELL = 4194304.0f;       // accumulator start of an invalid cell: never the minimum of a tile that has a valid one
constexpr float BIG_G = 268435456.0f;          // "no candidate yet"
constexpr float G_VALID_BELOW = 1048576.0f; // valid packed minima stay below 2^16

// the eight bits of a byte as eight FP4 nibbles: 0 / +1.0 for target bits, +1.0 / -1.0 for source bits
__device__ __forceinline__ uint32_t byte_nibbles(uint32_t b, uint32_t clear, uint32_t set) {
    uint32_t v = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) v |= (((b >> e) & 1u) ? set : clear) << (4 * e);
    return v;
}

// min of three (v_min3_f32).  Not inline asm: the tree is the first reader of the MFMA result and the compiler only pads the
// MFMA -> VALU read hazard for instructions it can see.
__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

__device__ __forceinline__ constexpr int row_of_reg(int reg) { return (reg & 3) + 8 * (reg >> 2); } // + 4 * (lane >> 5)

template <int NW, int WAVES, int CT>
__global__ void __launch_bounds__(64 * WAVES) census_sweep_mfma4_kernel(CensusGeom g, float Pout, uint2 *__restrict__ keys, float *__restrict__ gmap) {
    constexpr int PXB = WAVES * CT * 32; // source pixels per block
    constexpr int NG = (NW + 1) / 2; // MFMAs per tile: 64 bits of census record each
    extern __shared__ __attribute__((aligned(16))) uint4 lds4[]; // [2 NG words (a missing odd word is a zero chunk)][WIN window pixels] x 16 bytes
    const int i = blockIdx.y, j0 = blockIdx.x * PXB;
    const int NT = g.D / 32 + 1;  // row tiles per column tile
    const int WIN = PXB + g.D;    // window pixels: the last column tile's last row tile ends at PXB - 32 + 32 NT
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const bool fwd = g.sign > 0;     // RightToLeft: target column j + lower + d; LeftToRight: j - lower - d, everything mirrored
    const int xoff = fwd ? r : 31 - r; // source pixel of this lane inside its column tile (the lane index stays the tile's `col`)
    const int64_t row = (int64_t)i * g.Ws;
    // ---- every global load of the kernel up front: the source words of this wave's column tiles, then the target window
    uint32_t sword[CT][NW];
#pragma unroll
    for (int c = 0; c < CT; c++) {
        const int j = j0 + (wave * CT + c) * 32 + xoff;
        const uint32_t *sp = g.sw + (row + min(j, g.Ws - 1)) * NW;
#pragma unroll
        for (int m = 0; m < NW; m++) sword[c][m] = sp[m];
    }
    // ---- byte -> eight nibbles tables (one dword per byte): target bits 0 / +1.0, source bits +1.0 / -1.0
    uint32_t *lut_t = reinterpret_cast<uint32_t *>(lds4 + 2 * NG * WIN + 512), *lut_s = lut_t + 256;
    for (int b = threadIdx.x; b < 256; b += 64 * WAVES) {
        lut_t[b] = byte_nibbles(b, 0x0u, 0x2u);
        lut_s[b] = byte_nibbles(b, 0x2u, 0xAu);
    }
    // ---- the target window as nibbles; window pixel y is target column j0 + lower + y (LeftToRight: j0 + PXB - 1 - lower - y, so that
    // y still grows with the disparity; zero vector outside the image).  A thread
    // takes whole pixels (all NW words: one address, one bounds test), PPT of them with every load issued before the first table
    // look-up; the barrier that completes the tables sits under those loads.
    {
        const uint32_t *trow = g.tw + (int64_t)i * g.Wt * NW;
        constexpr int PPT = 4; // pixels per thread and round
        for (int y0 = threadIdx.x; y0 < WIN; y0 += PPT * 64 * WAVES) { // (first round: every thread enters, WIN > 64 WAVES)
            uint32_t w[PPT][NW];
#pragma unroll
            for (int u = 0; u < PPT; u++) {
                const int y = y0 + u * 64 * WAVES, jt = fwd ? j0 + g.disp_lower + y : j0 + PXB - 1 - g.disp_lower - y;
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
                    for (int m = 0; m < 2 * NG; m++) {
                        const uint32_t v = m < NW ? w[u][m < NW ? m : 0] : 0u;
                        lds4[m * WIN + y] = m < NW ? make_uint4(lut_t[v & 0xFFu], lut_t[(v >> 8) & 0xFFu], lut_t[(v >> 16) & 0xFFu], lut_t[v >> 24])
                                                   : make_uint4(0u, 0u, 0u, 0u);
                    }
                }
            }
        }
    }
    // ---- accumulator starts: 15 - reg for full tiles (in registers), and for the two triangular tiles the same with the invalid
    // cells pushed out of reach (in LDS behind the window: read twice per column tile)
    // (the zero is opaque to the compiler on purpose: a start pattern it can rematerialise it rebuilds in the accumulator registers
    // before every tile, eight v_mov_b64; one that lives in registers goes into the MFMA as its C operand)
    float opaque_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque_zero));
    v16f full_tile;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) full_tile[reg] = (float)(15 - reg) + opaque_zero;
    float4 *edge = reinterpret_cast<float4 *>(lds4 + 2 * NG * WIN); // [first, last][4 register quads][64 lanes]
    for (int q = wave; q < 8; q += WAVES) { // eight register quads (first tile 0..3, last tile 4..7) shared out over the waves
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int reg = 4 * (q & 3) + e, rr = row_of_reg(reg) + 4 * h;
            const bool valid = q < 4 ? rr >= r   // tile 0: d = row - col >= 0
                                     : rr < r;   // tile D/32: d = D + row - col < D
            v[e] = (float)(15 - reg) + (valid ? 0.0f : BIG_CELL);
        }
        edge[q * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
    auto edge_tile = [&](int which) {
        v16f a;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 t = edge[(4 * which + q) * 64 + lane];
            a[4 * q] = t.x;
            a[4 * q + 1] = t.y;
            a[4 * q + 2] = t.z;
            a[4 * q + 3] = t.w;
        }
        return a;
    };
    const int thr = g.Ws + g.disp_lower - g.d_offset; // target columns from here on pay Pout
    const int pout = (int)Pout;
    __syncthreads();

    // one column tile after the other; the MFMAs of row tile k + 1 are issued before the epilogue of row tile k
    for (int c = 0; c < CT; c++) {
        const int ct = wave * CT + c; // column tile of the block (wave uniform)
        const int j = j0 + ct * 32 + xoff;
        // source fragments (this lane's word of every 64-bit group as +1.0 / -1.0 nibbles) and |s|
        v8i bfrag[NG];
        int ones = 0;
#pragma unroll
        for (int m = 0; m < NW; m++) {
            uint32_t w = sword[0][m];
#pragma unroll
            for (int cc = 1; cc < CT; cc++) w = c == cc ? sword[cc][m] : w; // c is wave uniform: scalar selects, no indexed registers
            ones += __popc(w);
        }
#pragma unroll
        for (int gq = 0; gq < NG; gq++) {
            uint32_t w0 = sword[0][2 * gq], w1 = 2 * gq + 1 < NW ? sword[0][2 * gq + 1 < NW ? 2 * gq + 1 : 0] : 0u;
#pragma unroll
            for (int cc = 1; cc < CT; cc++) {
                w0 = c == cc ? sword[cc][2 * gq] : w0;
                if (2 * gq + 1 < NW) w1 = c == cc ? sword[cc][2 * gq + 1 < NW ? 2 * gq + 1 : 0] : w1;
            }
            const bool present = h == 0 || 2 * gq + 1 < NW; // an absent odd word contributes nothing: all-zero nibbles
            const uint32_t w = h ? w1 : w0;
            bfrag[gq] = (v8i){present ? (int)lut_s[w & 0xFFu] : 0, present ? (int)lut_s[(w >> 8) & 0xFFu] : 0, present ? (int)lut_s[(w >> 16) & 0xFFu] : 0,
                              present ? (int)lut_s[w >> 24] : 0, 0, 0, 0, 0};
        }
        const uint4 *arow = lds4 + h * WIN + (fwd ? ct * 32 : PXB - 32 * (ct + 1)) + r; // row tile k, group gq: arow[2 gq WIN + 32 k]
        struct Frags {
            v8i w[NG];
        };
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
            for (int gq = 0; gq < NG; gq++) // FP4 x FP4, block scales 2^3 (E8M0 130) and 2^6 (133): a differing bit adds 512
                a = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(f.w[gq], bfrag[gq], a, 4, 4, 0, 130, 0, 133);
            return a;
        };
        // Region bookkeeping (wave uniform).  Cell (row tile k, row, col) is disparity 32 k + row - col of its lane's pixel and pays
        // Pout when pixel + global index >= Ws:  32 k + row >= below + slope col, with slope 0 for RightToLeft (the boundary is a
        // target column: a row of the tile) and 2 for LeftToRight (pixel and target column move apart).  Tiles before k_lo hold no
        // paying cell, tiles from k_hi on only paying cells, the tiles between (at most one, LeftToRight three) are masked per cell.
        const int below = g.Ws - g.d_offset - j0 - ct * 32 - (fwd ? 0 : 31), slope = fwd ? 0 : 2;
        const int k_lo = below <= 0 ? 0 : min(below >> 5, NT);
        const int all_from = below + 31 * slope; // 32 k >= this: every cell pays
        const int k_hi = all_from <= 0 ? 0 : min((all_from + 31) >> 5, NT);
        const bool none_pays = k_lo >= NT, all_pay = k_hi <= 0; // (wave uniform)
        float G = BIG_G, G0 = BIG_G, Hpay = BIG_G; // running minimum (frame of the current tile); closed no-Pout region; Pout region inside the masked tiles
        // `checked`: this column tile has the Pout boundary inside its row tiles (only near the right image border); the others
        // run the loop without the per-tile region test
        auto epilogue = [&](const v16f &a, int k, auto checked) {
            if constexpr (decltype(checked)::value) {
                if (k >= k_lo && k <= k_hi) { // around the boundary: a few tiles per column tile
                    asm volatile("; region boundary" ::: "memory"); // keep this a branch: if-converted it costs selects in every tile
                    if (k < k_hi) {
                        float m0 = BIG_CELL, m1 = BIG_CELL;
                        int below_lane = below + slope * r;
                        asm volatile("; masked tile" : "+v"(below_lane)); // or the 16 compares below are hoisted in front of every column tile
#pragma unroll
                        for (int reg = 0; reg < 16; reg++) {
                            const bool pays = 32 * k + row_of_reg(reg) + 4 * h >= below_lane;
                            m0 = __builtin_fminf(m0, pays ? BIG_CELL : a[reg]);
                            m1 = __builtin_fminf(m1, pays ? a[reg] : BIG_CELL);
                        }
                        G = __builtin_fminf(G + 16.0f, m0);       // both regions keep their running minimum, frame of tile k
                        Hpay = __builtin_fminf(Hpay + 16.0f, m1);
                        return;
                    }
                    // k == k_hi: the first tile that pays everywhere.  The no-Pout region is closed (frame of tile k - 1: absolute tag);
                    // the Pout region goes on from what the masked tiles found
                    G0 = G + (float)(16 * (31 - (k - 1)));
                    G = Hpay;
                }
            }
            const float t1 = min3f(a[0], a[1], a[2]), t2 = min3f(a[3], a[4], a[5]), t3 = min3f(a[6], a[7], a[8]), t4 = min3f(a[9], a[10], a[11]),
                        t5 = min3f(a[12], a[13], a[14]);
            const float t6 = min3f(t1, t2, t3), t7 = min3f(t4, t5, a[15]);
            G = min3f(G + 16.0f, t6, t7); // frame of tile k: min over the tiles so far of (tile minimum + 16 (k - tile))
        };
        // NT >= 2: tile 0 and tile NT - 1 are the triangular ones, the loop body only meets full tiles.  Software pipeline: the
        // fragments of tile k + 1 are in registers and the MFMAs of tile k in flight while the epilogue of tile k - 1 runs, so
        // neither the LDS latency nor the matrix pipe's sits between two epilogues of a wave.
        auto sweep_tiles = [&](auto checked) {
            Frags f_next = load_frags(0);
            v16f acc_a = tile(f_next, edge_tile(0)), acc_b;
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
        if (none_pays || all_pay) sweep_tiles(std::false_type{}); // one region: no test inside
        else sweep_tiles(std::true_type{});
        // out of the frame (tile NT - 1): the absolute tag of tile k is 16 (31 - k)
        const float tag_last = (float)(16 * (31 - (NT - 1)));
        float G1;
        if (none_pays) { // everything ran into G
            G0 = G + tag_last;
            G1 = BIG_G;
        } else if (k_hi >= NT) { // the masked tiles reach the end of the range: both regions still open
            G0 = G + tag_last;
            G1 = Hpay + tag_last;
        } else { // G0 was closed at k_hi (all_pay: it is still BIG_G), G has been the Pout region since
            G1 = G + tag_last;
        }
        // ---- decode (tile, cell) -> key; the two lane halves hold different rows of the same pixel: merge, store
        auto decode = [&](float gf) {
            const int gv = (int)gf; // exact: an integer below 2^24
            const int kk = 31 - ((gv >> 4) & 31);
            const int reg = 15 - (gv & 15), cost = ones + (gv >> 9);
            const int d_local = 32 * kk + (reg & 3) + 8 * (reg >> 2) + 4 * h - r;
            return gf < G_VALID_BELOW ? (int)make_key(cost, g.d_offset + d_local) : (int)KEY_NONE;
        };
        int a0 = decode(G0), a1 = decode(G1);
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
    SVH_LAUNCH(ctx, "census_sweep_mfma4", (census_sweep_mfma4_kernel<NW, WAVES, CT>), grid, 64 * WAVES, shmem, g, Pout, keys, gmap);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SVH_OK : fail(ctx, SVH_ERR_HIP, "census_sweep (mfma, fp4): %s", hipGetErrorString(e));
}

template <int NW> bool launch_for_words(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, int *status) {
    // block width (4 waves x CT column tiles of 32 pixels): the widest that fits 64 KB of LDS, preferring less row padding
    int best_ct = 0;
    int64_t best_pad = 0;
    for (int ct = 4; ct >= 2; ct--) {
        const int pxb = 4 * ct * 32;
        if (((size_t)2 * ((NW + 1) / 2) * (pxb + g.D) + 512 + 128) * sizeof(uint4) > 64 * 1024) continue;
        const int64_t pad = (int64_t)ceil_div(g.Ws, pxb) * pxb;
        if (!best_ct || pad < best_pad) {
            best_ct = ct;
            best_pad = pad;
        }
    }
    if (!best_ct) return false;
    const size_t shmem = ((size_t)2 * ((NW + 1) / 2) * (4 * best_ct * 32 + g.D) + 512 + 128) * sizeof(uint4); // window + the two edge patterns + the byte tables
    switch (best_ct) {
    case 4: *status = launch_config<NW, 4, 4>(ctx, g, Pout, keys, gmap, shmem); break;
    case 3: *status = launch_config<NW, 4, 3>(ctx, g, Pout, keys, gmap, shmem); break;
    default: *status = launch_config<NW, 4, 2>(ctx, g, Pout, keys, gmap, shmem); break;
    }
    return true;
}

} // namespace

bool launch_sweep_mfma4(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, int *status) {
    if (g.D < 32 || (g.D & 31) != 0 || g.D > 992 || g.nWw < 1 || g.nWw > 4) return false;
    switch (g.nWw) {
    case 1: return launch_for_words<1>(ctx, g, Pout, keys, gmap, status);
    case 2: return launch_for_words<2>(ctx, g, Pout, keys, gmap, status);
    case 3: return launch_for_words<3>(ctx, g, Pout, keys, gmap, status);
    default: return launch_for_words<4>(ctx, g, Pout, keys, gmap, status);
    }
}

} // namespace svh

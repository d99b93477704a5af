// Census / Hamming specialisation of the SGM Cost branch ("pixel per lane" kernels).
//
// For census costs the volume is a pure function of two small word maps, so nothing voxel-sized has to live in
// HBM.  These kernels map one image pixel to one lane and walk the disparity axis sequentially inside the lane:
//   * a block handles 256 consecutive pixels of one row and stages the (256 + D - 1)-pixel window of target
//     census words in LDS as an array of NW-word records, stored so that lane tj reads record ybase(tj) + d:
//     64 lanes read 64 consecutive records (ds_read_b32 / b64 / b128 with an immediate offset, conflict free);
//   * the per-pixel reductions over d (minimum, winner with the reference's "<=" scan order) need no cross-lane
//     traffic at all.
//
// Exactness argument used by the fast path (checked on the host before it is taken, see census_exact_regime):
// Hamming costs are integers in [0, 32 nWw]; when Pout is an integer and 8 (2 cmax + |Pout|) (L + 2) < 2^24
// (L = longest line), every intermediate of sgm.h:257-300 is an integer below 2^24 in magnitude, so each float
// operation of the reference is exact and integer arithmetic carries the very same values.  Then
//   * the next pixel's min_p is  min_d [ c + ((c [+Pout]) - mp) ] = g(p) - mp,  g(p) = min_d [ 2c [+Pout] ]:
//     the sequential part of a pass is the scalar recurrence mp' = g - mp along the line (scan kernels), fed by one
//     fully parallel sweep (census_gmin_kernel);
//   * S(p,d) = c + sum over visiting passes of ((c + (t - mp_q)) - c) = (1 + n) c + n Pout [j+d >= W] - sum mp_q,
//     and the "<=" winner scan over d is a min over keys (value, D-1-d).
// Outside that regime (non-integer or huge Pout) the float expressions of the reference are evaluated literally
// (EXACT = false variants) on min_p maps produced by the general wave-per-line kernels of svh_sgm.hip.
#include <type_traits>

#include "svh_internal.h"
#include "svh_census_keys.h"

namespace svh {

namespace {

constexpr int TJ = 256; // pixels (= threads) per block
constexpr int UNROLL = 8;


template <int NW> struct Words {
    uint32_t v[NW > 0 ? NW : 1];
};

// Stage the target-word window of this block into LDS: record y of lane tj's disparity d is y = ybase(tj) + d with
// ybase = tj for sign > 0 (target column j + lower + d) and TJ-1-tj for sign < 0 (target column j - lower - d, the
// window is stored mirrored so that y still grows with d).  Columns outside the target image hold the zero vector
// (cross_correlations.h:235).
template <int NW> __device__ __forceinline__ void stage_target_window(const CensusGeom &g, int i, int j0, uint32_t *lds) {
    if constexpr (NW > 0) {
        const int win = TJ + g.D - 1;
        const uint32_t *trow = g.tw + (int64_t)i * g.Wt * NW;
        for (int e = threadIdx.x; e < win * NW; e += TJ) {
            const int y = e / NW, w = e - y * NW;
            const int jt = g.sign > 0 ? j0 + g.disp_lower + y : j0 + (TJ - 1) - g.disp_lower - y;
            lds[e] = (jt >= 0 && jt < g.Wt) ? trow[(int64_t)jt * NW + w] : 0u;
        }
    }
}

template <int NW> __device__ __forceinline__ Words<NW> load_source_words(const CensusGeom &g, int64_t p) {
    Words<NW> s;
    s.v[0] = 0;
    if constexpr (NW > 0) {
        const uint32_t *sp = g.sw + p * NW;
#pragma unroll
        for (int w = 0; w < NW; w++) s.v[w] = sp[w];
    }
    return s;
}

// Hamming distance between the source words and LDS record `rec`
template <int NW> __device__ __forceinline__ int hamming_rec(const Words<NW> &s, const uint32_t *rec) {
    if constexpr (NW == 0) {
        return 0;
    } else if constexpr (NW == 2) {
        const uint2 t = *reinterpret_cast<const uint2 *>(rec);
        return __popc(s.v[0] ^ t.x) + __popc(s.v[1] ^ t.y);
    } else if constexpr (NW == 4) {
        const uint4 t = *reinterpret_cast<const uint4 *>(rec);
        return __popc(s.v[0] ^ t.x) + __popc(s.v[1] ^ t.y) + __popc(s.v[2] ^ t.z) + __popc(s.v[3] ^ t.w);
    } else {
        int acc = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) acc += __popc(s.v[w] ^ rec[w]);
        return acc;
    }
}

// Calls f(d, c) for d = 0..D-1 in increasing order, c = Hamming cost of (pixel, d); LDS reads are issued UNROLL
// records ahead of their use.
template <int NW, class F> __device__ __forceinline__ void for_each_disparity(const Words<NW> &s, const uint32_t *base, int D, F f) {
    int d0 = 0;
    for (; d0 + UNROLL <= D; d0 += UNROLL) {
        int c[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) c[u] = hamming_rec<NW>(s, base + (d0 + u) * NW);
#pragma unroll
        for (int u = 0; u < UNROLL; u++) f(d0 + u, c[u]);
    }
    for (; d0 < D; d0++) f(d0, hamming_rec<NW>(s, base + d0 * NW));
}

// The one voxel sweep of the exact regime.  Per pixel it keeps, for the two disparity regions d < dsplit (no Pout)
// and d >= dsplit (Pout applies, sgm.h:287-289), the minimum of the keys (c << 12 | 4095 - d): the regional minimum
// cost with the reference's tie rule (last index) built in.  From them
//   g(p) = min_d [ c + (c [+ Pout]) ] = min(2 c0, 2 c1 + Pout)          (first-pixel actual cost, min_p = 0)
// feeds the line recurrences, and census_finalize_kernel picks the winner once the min_p maps exist.
// Keys are positive int32 values (cost <= 32 nWw <= 256 needs 9 bits, the index 12), so an int32 MIN all-reduce across
// disparity shards is the cross-GPU winner reduction.  The index part holds the GLOBAL disparity index.
// Two neighbouring pixels per lane.  With the records stored as 2-record pairs P_m = (rec 2m, rec 2m+1), the "even"
// pixel E of a lane (record base 2 m0) and its "odd" neighbour O (base 2 m0 + 1) need, for the disparity pair (2e, 2e+1),
//   E: P_{m0+e}.lo, P_{m0+e}.hi        O: P_{m0+e}.hi, P_{m0+e+1}.lo
// i.e. ONE new 16-byte-aligned pair (ds_read_b128 for 2-word records) per four voxels.
constexpr int PX = 2 * TJ; // pixels per block

template <int NW> struct Rec {
    uint32_t w[NW > 0 ? NW : 1];
};

template <int NW> __device__ __forceinline__ void load_pair(const uint32_t *lds, int m, Rec<NW> &lo, Rec<NW> &hi) {
    if constexpr (NW == 2) {
        const uint4 t = *reinterpret_cast<const uint4 *>(lds + 4 * m);
        lo.w[0] = t.x; lo.w[1] = t.y; hi.w[0] = t.z; hi.w[1] = t.w;
    } else if constexpr (NW == 1) {
        const uint2 t = *reinterpret_cast<const uint2 *>(lds + 2 * m);
        lo.w[0] = t.x; hi.w[0] = t.y;
    } else if constexpr (NW == 4) {
        const uint4 a = *reinterpret_cast<const uint4 *>(lds + 8 * m), b = *reinterpret_cast<const uint4 *>(lds + 8 * m + 4);
        lo.w[0] = a.x; lo.w[1] = a.y; lo.w[2] = a.z; lo.w[3] = a.w; hi.w[0] = b.x; hi.w[1] = b.y; hi.w[2] = b.z; hi.w[3] = b.w;
    } else if constexpr (NW == 3) {
        const uint2 a = *reinterpret_cast<const uint2 *>(lds + 6 * m), b = *reinterpret_cast<const uint2 *>(lds + 6 * m + 2),
                    c = *reinterpret_cast<const uint2 *>(lds + 6 * m + 4);
        lo.w[0] = a.x; lo.w[1] = a.y; lo.w[2] = b.x; hi.w[0] = b.y; hi.w[1] = c.x; hi.w[2] = c.y;
    } else if constexpr (NW > 4) {
        // pairs of 2 NW words start on 8-byte boundaries: NW reads of two words
        const uint2 *p = reinterpret_cast<const uint2 *>(lds + 2 * NW * m);
        uint32_t t[2 * NW];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint2 q = p[w];
            t[2 * w] = q.x; t[2 * w + 1] = q.y;
        }
#pragma unroll
        for (int w = 0; w < NW; w++) { lo.w[w] = t[w]; hi.w[w] = t[NW + w]; }
    } else {
        lo.w[0] = 0; hi.w[0] = 0;
    }
}

template <int NW> __device__ __forceinline__ int hamming_words(const Words<NW> &s, const Rec<NW> &r) {
    int acc = 0;
    if constexpr (NW > 0) {
#pragma unroll
        for (int w = 0; w < NW; w++) acc += __popc(s.v[w] ^ r.w[w]);
    }
    return acc;
}

// window of PX + D - 1 records (+ padding to whole pairs), same record order as stage_target_window
template <int NW, int TJV> __device__ __forceinline__ void stage_target_window2(const CensusGeom &g, int i, int j0, uint32_t *lds, int n_rec) {
    constexpr int TJ = TJV, PX = 2 * TJV; // the sweep's own block width (shadows the file-wide default)
    if constexpr (NW > 0) {
        const uint32_t *trow = g.tw + (int64_t)i * g.Wt * NW;
        const int n = n_rec * NW;
        constexpr int CH = 8; // loads in flight per thread before the first LDS write
        for (int e0 = threadIdx.x; e0 < n; e0 += CH * TJ) {
            uint32_t v[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const int e = e0 + c * TJ;
                const int y = e / NW, w = e - y * NW;
                const int jt = g.sign > 0 ? j0 + g.disp_lower + y : j0 + (PX - 1) - g.disp_lower - y;
                v[c] = (e < n && jt >= 0 && jt < g.Wt && y < PX + g.D - 1) ? trow[(int64_t)jt * NW + w] : 0u;
            }
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const int e = e0 + c * TJ;
                if (e < n) lds[e] = v[c];
            }
        }
    }
}

__host__ __device__ inline int sweep_records(int D, int px = PX) { return (px + D - 1 + 3) & ~1; } // whole pairs, one spare pair for the look-ahead

// TJV lanes = 2 TJV pixels per block: 256 by default, 192 when that tiles the row with less padding (1920 = 5 x 384)
template <int NW, int TJV>
__global__ void __launch_bounds__(TJV) census_sweep_kernel(CensusGeom g, float Pout, uint2 *__restrict__ keys, float *__restrict__ gmap, int merge) {
    constexpr int TJ = TJV, PX = 2 * TJV;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int i = blockIdx.y, j0 = blockIdx.x * PX, t = threadIdx.x;
    stage_target_window2<NW, TJV>(g, i, j0, lds, sweep_records(g.D, PX));
    __syncthreads();
    const bool fwd = g.sign > 0;
    const int m0 = fwd ? t : TJ - 1 - t;                       // pair index of the even pixel at d = 0
    const int jE = j0 + (fwd ? 2 * t : 2 * t + 1), jO = j0 + (fwd ? 2 * t + 1 : 2 * t);
    const int64_t row = (int64_t)i * g.Ws;
    Words<NW> sE = load_source_words<NW>(g, row + min(jE, g.Ws - 1)), sO = load_source_words<NW>(g, row + min(jO, g.Ws - 1));
    const int d_off = g.d_offset, D = g.D;
    // region limits: a pixel pays Pout for d >= dsplit = W - j - d_off.  Over the 128 pixels of this wave the split moves
    // by 127, so d < lo is "nobody pays" and d >= hi "everybody pays", both wave uniform.
    const int wave_j0 = __builtin_amdgcn_readfirstlane(j0 + 2 * (t & ~63)); // first pixel of this wave (SGPR: scalar branches below)
    const int lo = min(max(g.Ws - d_off - (wave_j0 + 127), 0), D), hi = min(max(g.Ws - d_off - wave_j0, 0), D);
    const int splitE = min(max(g.Ws - d_off - jE, 0), D), splitO = min(max(g.Ws - d_off - jO, 0), D);
    uint32_t kE0 = KEY_NONE, kE1 = KEY_NONE, kO0 = KEY_NONE, kO1 = KEY_NONE;
    Rec<NW> cur_lo, cur_hi, nxt_lo, nxt_hi;
    load_pair<NW>(lds, m0, cur_lo, cur_hi);
    const int n_pairs = D >> 1;
    // three branch-free loops instead of a three-way test per iteration: pairs entirely below `lo` (nobody pays Pout), the
    // at most 64 pairs between `lo` and `hi` (per-lane split), pairs entirely from `hi` on (everybody pays).  The limits are
    // wave uniform: 2e + 1 < lo <=> e < lo >> 1, and 2e >= hi <=> e >= (hi + 1) >> 1.
    const int e_lo = min(lo >> 1, n_pairs), e_hi = min((hi + 1) >> 1, n_pairs);
    auto step = [&](int e, auto mode) {
        constexpr int MODE = decltype(mode)::value;
        load_pair<NW>(lds, m0 + e + 1, nxt_lo, nxt_hi);
        const int d = 2 * e;
        const uint32_t kEa = make_key(hamming_words<NW>(sE, cur_lo), d_off + d), kOa = make_key(hamming_words<NW>(sO, cur_hi), d_off + d);
        const uint32_t kEb = make_key(hamming_words<NW>(sE, cur_hi), d_off + d + 1), kOb = make_key(hamming_words<NW>(sO, nxt_lo), d_off + d + 1);
        if constexpr (MODE == 0) {
            kE0 = min(kE0, min(kEa, kEb));
            kO0 = min(kO0, min(kOa, kOb));
        } else if constexpr (MODE == 2) {
            kE1 = min(kE1, min(kEa, kEb));
            kO1 = min(kO1, min(kOa, kOb));
        } else {
            kE0 = min(kE0, min(d >= splitE ? KEY_NONE : kEa, d + 1 >= splitE ? KEY_NONE : kEb));
            kE1 = min(kE1, min(d >= splitE ? kEa : KEY_NONE, d + 1 >= splitE ? kEb : KEY_NONE));
            kO0 = min(kO0, min(d >= splitO ? KEY_NONE : kOa, d + 1 >= splitO ? KEY_NONE : kOb));
            kO1 = min(kO1, min(d >= splitO ? kOa : KEY_NONE, d + 1 >= splitO ? kOb : KEY_NONE));
        }
        cur_lo = nxt_lo;
        cur_hi = nxt_hi;
    };
    // (the min-reductions are associative, so the loop vectoriser would otherwise interleave 16 iterations and spill the
    // register file: 256 VGPRs instead of 40)
#pragma clang loop vectorize(disable) interleave(disable) unroll_count(4)
    for (int e = 0; e < e_lo; e++) step(e, std::integral_constant<int, 0>{});
#pragma clang loop vectorize(disable) interleave(disable) unroll_count(2)
    for (int e = e_lo; e < e_hi; e++) step(e, std::integral_constant<int, 1>{});
#pragma clang loop vectorize(disable) interleave(disable) unroll_count(4)
    for (int e = e_hi; e < n_pairs; e++) step(e, std::integral_constant<int, 2>{});
    if (D & 1) { // last (even) disparity
        const int d = D - 1;
        const uint32_t kEa = make_key(hamming_words<NW>(sE, cur_lo), d_off + d), kOa = make_key(hamming_words<NW>(sO, cur_hi), d_off + d);
        kE0 = min(kE0, d >= splitE ? KEY_NONE : kEa);
        kE1 = min(kE1, d >= splitE ? kEa : KEY_NONE);
        kO0 = min(kO0, d >= splitO ? KEY_NONE : kOa);
        kO1 = min(kO1, d >= splitO ? kOa : KEY_NONE);
    }
    const int pout = (int)Pout;
    if (g.region1_global_last >= 0) { // disparity shards: the Pout region's winner over all shards (svh_internal.h, CostSource)
        int onesE = 0, onesO = 0;
        if constexpr (NW > 0) {
#pragma unroll
            for (int w = 0; w < NW; w++) {
                onesE += __popc(sE.v[w]);
                onesO += __popc(sO.v[w]);
            }
        }
        kE1 = global_region1_key(g, jE, onesE);
        kO1 = global_region1_key(g, jO, onesO);
    }
    // merge != 0: `keys` already holds the keys of the disparities below this call's (launch_sweep: the matrix-core sweep took the leading
    // multiple of 32); keys of disjoint ranges combine by MIN, as they do across disparity shards
    if (jE < g.Ws) {
        if (merge) {
            const uint2 o = keys[row + jE];
            kE0 = min(kE0, o.x);
            kE1 = min(kE1, o.y);
        }
        keys[row + jE] = make_uint2(kE0, kE1);
        if (gmap) gmap[row + jE] = g_from_keys(kE0, kE1, pout);
    }
    if (jO < g.Ws) {
        if (merge) {
            const uint2 o = keys[row + jO];
            kO0 = min(kO0, o.x);
            kO1 = min(kO1, o.y);
        }
        keys[row + jO] = make_uint2(kO0, kO1);
        if (gmap) gmap[row + jO] = g_from_keys(kO0, kO1, pout);
    }
}

// the same g from already reduced keys (disparity-sharded runs)
__global__ void gmap_from_keys_kernel(const uint2 *__restrict__ keys, int64_t npx, float Pout, float *__restrict__ gmap) {
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p < npx) gmap[p] = g_from_keys(keys[p].x, keys[p].y, (int)Pout);
}

struct ScanGeom {
    int top, left, Hp, Wp, W;
};

// ---- the recurrence mp' = g - mp in parallel form ------------------------------------------------------------
// Over a run of n pixels mp' = g - mp is the affine map mp_out = A + (-1)^n mp_in (A = the run started from 0),
// and such maps compose, so a line can be cut into segments that are summarised independently, chained, and
// replayed.  All sums are of integers below 2^24 (exact regime), so the split does not change a single bit.
//
// scan_cols_kernel: passes whose lines cross the rows (0, 2, 3, 4, 5).  Lines are indexed by a "virtual column" v
// with pixel column = v + s * row (s = 0 vertical, +1 diagonal, -1 anti-diagonal), so that at every row the 64
// lanes of a wave (64 neighbouring lines) touch 64 consecutive pixels: every access is a coalesced 256-byte row
// segment.  The rows are cut into SCAN_SEGS chunks handled by the threadIdx.y slices of the block.
constexpr int SCAN_SEGS = 16;
constexpr int SCAN_MAXSEG = 72; // segments up to this many rows are kept in registers (a launch covers up to 1152 rows)
constexpr int TILE_SUB = 8;     // tile rows per segment (tiles mode: TileMaps, tile_finalize_kernel)
// rows per segment: a multiple of TILE_SUB (and of four: the lean form's lengths)
__host__ __device__ __forceinline__ int scan_chunk(int Hp) { return ((Hp + SCAN_SEGS - 1) / SCAN_SEGS + TILE_SUB - 1) / TILE_SUB * TILE_SUB; }
// Tiles mode on images taller than one launch covers (round 4f): the rows are cut into BANDS of SCAN_SEGS segments, one launch per band,
// and a line's min_p crosses from band to band through one float per line and pass (ScanBands::carry: the affine maps of the bands
// compose like those of the segments).  The downward passes take the bands top to bottom, pass 5 bottom to top, so launch k holds band k
// of the former and band n - 1 - k of the latter.
struct ScanBands {
    int n = 1, chunk = 0; // bands, rows per segment (every band: SCAN_SEGS * chunk rows, the last one what is left)
    // n > 1: 5 x stride floats each; a launch reads what the one before it wrote (zeros for the first) and writes the other array --
    // two arrays, so that no wave's read of an entry can meet the write of the block's last slice
    const float *carry_in = nullptr;
    float *carry_out = nullptr;
    int stride = 0;
    int band_dn = 0, band_up = 0; // the bands of this launch
    __host__ __device__ int rows() const { return SCAN_SEGS * chunk; }
};
inline ScanBands scan_bands_for(int Hp) {
    ScanBands b;
    b.n = (Hp + SCAN_SEGS * SCAN_MAXSEG - 1) / (SCAN_SEGS * SCAN_MAXSEG);
    if (b.n < 1) b.n = 1;
    b.chunk = ((Hp + SCAN_SEGS * b.n - 1) / (SCAN_SEGS * b.n) + TILE_SUB - 1) / TILE_SUB * TILE_SUB;
    return b;
}

struct ColLines {
    int s, v_lo, n_lines, dir;
};

__device__ __forceinline__ ColLines col_lines(int q, const ScanGeom &sg) {
    switch (q) {
    case 0: return {0, 0, sg.Wp, 1};
    case 2: return {1, -(sg.Hp - 1), sg.Hp, 1};
    case 3: return {1, 0, sg.Wp, 1};
    case 4: return {-1, 0, sg.Wp, 1};
    default: return {-1, 0, sg.Hp, -1};
    }
}

// rows [r0, r1) (relative to the margin box) visited by line v of pass q
__device__ __forceinline__ void col_line_rows(int q, int v, const ScanGeom &sg, int &r0, int &r1) {
    switch (q) {
    case 0: r0 = 0; r1 = sg.Hp; break;
    case 2: r0 = -v; r1 = min(sg.Hp, sg.Wp - v); break;
    case 3: r0 = 0; r1 = min(sg.Hp, sg.Wp - v); break;
    case 4: r0 = 0; r1 = min(sg.Hp, v + 1); break;
    default: r0 = max(0, v - sg.Wp + 1); r1 = v + 1; break;
    }
}

// The register form of scan_cols_kernel for a wave whose 64 lines all have the same LEN rows in its segment (or none: lane_off out
// of range).  The general form spends 26 instructions per pixel -- 64-bit addresses, a per-lane `k < n` around every load, update and
// store -- and the launch is bound by instruction issue, not by its 100 MB (SQ counters, profiles/r03a: 5.4 M wave instructions,
// 0.49 M branches, on 192 busy CUs).  Here the row count is a compile-time constant, the pixel of traversal step k is buffer base +
// lane offset + a scalar offset carried from step to step: no vector address arithmetic, no test per pixel; lanes without a line
// load zeros and their stores are dropped (range-checked buffer accesses).  The arithmetic per pixel is the same subtraction in the
// same order as in the general form: same bits.
// What scan_cols_kernel<true> leaves instead of the six min_p maps, for tile_finalize_kernel: the min_p values on the EDGES of the
// tiles of R rows x 64 columns (R = an eighth of a segment) -- a ninth of the maps at 1080p.  Families f: 0 vertical (pass 0),
// 1 diagonal (passes 2 and 3: one plane), 2 anti-diagonal downwards (pass 4), 3 anti-diagonal upwards (pass 5).
//   E_f[y][col]   min_p of family f at pixel (row, col), row = the first row of tile row y in the family's direction of travel (its top
//                 row R y; for f = 3 its bottom row): what the lines bring along when they enter tile row y
//   S_f[tx][row]  f = 1, 3: min_p at pixel (row, 64 tx); f = 2: at (row, 64 tx + 63): what a line brings along when it enters tile
//                 column tx through its side
//   C1[row][tx]   pass 1: the signed prefix sum of the row over the tile columns before tx (scan_one_row)
// Entries of pixels a pass does not visit are never written and never used.
struct TileMaps {
    float *base = nullptr;
    int Hp = 0, Wp = 0, R = 1;
    __host__ __device__ int ny() const { return (Hp + R - 1) / R; }
    __host__ __device__ int tx() const { return (Wp + 63) / 64; }
    __host__ __device__ int64_t e_size() const { return (int64_t)ny() * Wp; }
    __host__ __device__ int64_t s_size() const { return (int64_t)tx() * Hp; }
    __host__ __device__ int64_t e_off(int f) const { return f * e_size(); }
    __host__ __device__ int64_t s_off(int f) const { return 4 * e_size() + (f - 1) * s_size(); } // f = 1, 2, 3
    __host__ __device__ int64_t c1_off() const { return 4 * e_size() + 3 * s_size(); }
    __host__ __device__ int64_t total() const { return 4 * e_size() + 4 * s_size(); }
};
__host__ __device__ __forceinline__ int tile_family(int q) { return q == 0 ? 0 : (q <= 3 ? 1 : q - 2); }

template <bool CARRY>
__device__ __forceinline__ void scan_one_row(const float *__restrict__ gmap, const ScanGeom &sg, int64_t npx, float *__restrict__ mmap, int row,
                                             int lane, const TileMaps &tm);

struct LeanArgs {
    const float *gmap;
    float *out;
    int64_t npx;
    uint32_t st4;      // bytes from a line's pixel to its pixel one row down
    uint32_t lane_off; // byte offset of this lane's pixel in the segment's first (lowest) row; 0xFFFFFFFF: no line
    int dir, seg, lane, n;
    // tiles mode
    int q, s, vwave, nb, n_u; // pass, column step per row, line of lane 0, first row and row count of the segment
    int v_lo, n_lines;        // the pass's lines
    int kb;                   // first traversal step of this lane's line in the segment (its steps: [kb, kb + n))
    int ty0;                  // tile row of the band's first row
    const float *carry_in;    // several bands: this lane's entries of ScanBands::carry_in / carry_out (nullptr: one band, lines start from 0)
    float *carry_out;
};
// FULL: every lane's line crosses all LEN rows of the segment or none (and, in tiles mode, LEN = TILE_SUB R: a whole segment)
// BANDS: the image is taller than one launch covers (ScanBands): lines start from what the band before left and leave their own
template <int LEN, bool CARRY, bool FULL, bool BANDS = false>
__device__ __forceinline__ void scan_lean(const LeanArgs &a, float (*seg_a)[64], int (*seg_n)[64], const TileMaps &tm) {
    static_assert(CARRY || FULL, "the maps mode has its own general form");
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)a.gmap, 0, (int)(a.npx * 4), 0x00020000);
    float gv[LEN];
    float mp = 0.0f, carry_in = 0.0f;
    if constexpr (BANDS) carry_in = *a.carry_in; // what the bands before this one leave on this lane's line
    // traversal step k is the segment's row k (downwards) or n_u - 1 - k (pass 5, upwards)
    if constexpr (FULL) {
        const uint32_t s0 = a.dir > 0 ? 0u : (uint32_t)(LEN - 1) * a.st4, sd = a.dir > 0 ? a.st4 : 0u - a.st4;
        uint32_t so = s0;
#pragma unroll
        for (int k = 0; k < LEN; k++) {
            gv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, a.lane_off, so, 0));
            so += sd;
        }
#pragma unroll
        for (int k = 0; k < LEN; k++) mp = gv[k] - mp;
    } else {
        // lines that start or end inside the segment, a last segment shorter than LEN: a pixel outside the lane's steps [kb, kb + n)
        // loads 0.  Zeros before the line starts leave min_p at 0; the zeros after its end flip the sign of the sum once each.
        const uint32_t first_off = a.lane_off + (a.dir > 0 ? 0u : (uint32_t)(a.n_u - 1) * a.st4), sd = a.dir > 0 ? a.st4 : 0u - a.st4;
#pragma unroll
        for (int k = 0; k < LEN; k++) {
            const bool on = (unsigned)(k - a.kb) < (unsigned)a.n;
            gv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, on ? first_off + (uint32_t)k * sd : 0xFFFFFFFFu, 0, 0));
        }
#pragma unroll
        for (int k = 0; k < LEN; k++) mp = gv[k] - mp;
        mp = ((LEN - (a.kb + a.n)) & 1) ? -mp : mp;
        mp = a.n ? mp : 0.0f;
    }
    seg_a[a.seg][a.lane] = mp;
    seg_n[a.seg][a.lane] = a.n;
    __syncthreads();
    mp = carry_in;
    if (a.dir > 0) {
        for (int t = 0; t < a.seg; t++) mp = seg_a[t][a.lane] + ((seg_n[t][a.lane] & 1) ? -mp : mp);
    } else {
        for (int t = SCAN_SEGS - 1; t > a.seg; t--) mp = seg_a[t][a.lane] + ((seg_n[t][a.lane] & 1) ? -mp : mp);
    }
    if constexpr (CARRY) {
        if (BANDS && a.seg == (a.dir > 0 ? SCAN_SEGS - 1 : 0)) { // the last segment in traversal order: the whole band's map, for the next band
            const float out_c = seg_a[a.seg][a.lane] + ((seg_n[a.seg][a.lane] & 1) ? -mp : mp);
            if (a.v_lo + a.n_lines > a.vwave + a.lane) *a.carry_out = out_c;
        }
        // The replay keeps what lies on tile edges: E at the first row of every tile row in the direction of travel (a full-wave
        // store), S where a lane's pixel sits on a tile's side: once every 64 rows, at steps known beforehand -- a compare and a
        // select per row park the value, one store per lane at the end writes it.  (A line that has not started yet carries 0
        // through zeros; after its end nothing is kept.)
        constexpr int RF = LEN / TILE_SUB > 0 ? LEN / TILE_SUB : 1; // (FULL: LEN = TILE_SUB R)
        const int f = tile_family(a.q), T = a.q == 4 ? 63 : 0, R = FULL ? RF : tm.R;
        const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void *)(tm.base + tm.e_off(f)), 0, (int)(tm.e_size() * 4), 0x00020000);
        const int v = a.vwave + a.lane;
        // (general form) the traversal steps at which a tile row begins, as a bit mask in scalar registers -- the test per step is one
        // s_bitcmp -- and the tile row of the first of them; every further one is the next tile row in the direction of travel.
        // Downwards the segment starts on a tile row (nb is a multiple of R): steps 0, R, 2 R, ...; upwards the tile rows' bottom rows
        // are the steps n_u - R m > 0 and step 0 (the segment's last row, whatever tile row it lies in)
        unsigned long long em_lo = 0;
        unsigned em_hi = 0;
        int ty_next = (BANDS ? a.ty0 : 0) + TILE_SUB * a.seg;
        if constexpr (!FULL) {
            int k0 = 0;
            if (a.dir < 0) {
                int left = a.n_u - 1;
                for (; left >= R; left -= R) ty_next++;
                k0 = left + 1 == R ? 0 : left + 1; // n_u mod R
                em_lo = 1;
            }
            for (int kk = k0; kk < a.n_u; kk += R) {
                if (kk < 64) em_lo |= 1ull << kk;
                else em_hi |= 1u << (kk - 64);
            }
            em_lo = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(em_lo >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)em_lo);
            em_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)em_hi);
            ty_next = __builtin_amdgcn_readfirstlane(ty_next);
        }
        // S: a lane's pixel sits on a tile's side (column = T mod 64) once every 64 rows, at traversal steps khit and khit + 64
        const int row0 = a.dir > 0 ? a.nb : a.nb + a.n_u - 1, sd1 = a.s * a.dir;
        const int khit = ((T - v - a.s * row0) * sd1) & 63;
        float hit0 = 0.0f, hit1 = 0.0f;
#pragma unroll
        for (int k = 0; k < LEN; k++) {
            const int row = a.dir > 0 ? a.nb + k : a.nb + a.n_u - 1 - k; // (wave uniform)
            bool edge; // (wave uniform) the first row of a tile row in the direction of travel
            int ty;
            if constexpr (FULL) {
                edge = k % RF == 0;
                ty = (BANDS ? a.ty0 : 0) + TILE_SUB * a.seg + (a.dir > 0 ? k / RF : TILE_SUB - 1 - k / RF);
            } else {
                edge = k < 64 ? (em_lo >> k) & 1 : (em_hi >> (k - 64)) & 1;
                ty = ty_next;
            }
            if (edge) {
                const bool on = FULL ? a.lane_off != 0xFFFFFFFFu : (unsigned)(k - a.kb) < (unsigned)a.n;
                const uint32_t eo = on ? (uint32_t)((ty * tm.Wp + v + a.s * row) * 4) : 0xFFFFFFFFu;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mp), re, eo, 0, 0);
                if constexpr (!FULL) ty_next += a.dir;
            }

            if (f > 0) {
                if (k < 64) hit0 = khit == k ? mp : hit0;
                else hit1 = khit == k - 64 ? mp : hit1;
            }
            mp = gv[k] - mp;
        }
        if (f > 0) {
            float *const sp = tm.base + tm.s_off(f);
#pragma unroll
            for (int hh = 0; hh < (LEN + 63) / 64; hh++) {
                const int k = khit + 64 * hh, row = row0 + a.dir * k, col = v + a.s * row;
                const bool on = FULL ? (a.lane_off != 0xFFFFFFFFu && k < LEN) : (unsigned)(k - a.kb) < (unsigned)a.n;
                if (on && col >= 0 && col < tm.Wp) sp[(int64_t)(col >> 6) * tm.Hp + row] = hh ? hit1 : hit0;
            }
        }
        return;
    }
    if constexpr (FULL) {
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)a.out, 0, (int)(a.npx * 4), 0x00020000);
        const uint32_t s0 = a.dir > 0 ? 0u : (uint32_t)(LEN - 1) * a.st4, sd = a.dir > 0 ? a.st4 : 0u - a.st4;
        uint32_t so = s0;
#pragma unroll
        for (int k = 0; k < LEN; k++) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mp), ro, a.lane_off, so, 0);
            so += sd;
            mp = gv[k] - mp;
        }
    }
}

template <bool CARRY, bool BANDS = false>
__global__ void __launch_bounds__(64 * SCAN_SEGS) scan_cols_kernel(const float *__restrict__ gmap, ScanGeom sg, int64_t npx, int n_pass,
                                                                   float *__restrict__ mmap, const int *__restrict__ skip_if_nonzero, TileMaps tm, ScanBands bands) {
    if (skip_if_nonzero && *skip_if_nonzero != 0) return; // the integer-volume probe failed: the line kernels make the maps
    // slice (which pass) and block within the slice: blockIdx.y / blockIdx.x of a two-dimensional grid whose slices are as wide as the
    // widest (blocks past a pass's lines leave at once), or -- tiles mode -- a one-dimensional grid of exactly the blocks that have
    // work, the diagonal passes (the longest blocks) first: pass 2, 3, 4, 5, 0, the rows
    int by = blockIdx.y, bx = blockIdx.x;
    if (CARRY && gridDim.y == 1) {
        const int cw = (sg.Wp + 63) / 64, ch = (sg.Hp + 63) / 64;
        const int band_rows = BANDS ? min(bands.rows(), sg.Hp - bands.band_dn * bands.rows()) : sg.Hp; // (pass 1: the rows of this launch's downward band)
        const int count[6] = {ch, cw, cw, ch, cw, (band_rows + SCAN_SEGS - 1) / SCAN_SEGS}, slice[6] = {1, 2, 3, 4, 0, 5};
        int rem = blockIdx.x;
        by = -1;
#pragma unroll
        for (int t = 0; t < 6; t++) {
            if (by < 0 && rem < count[t]) {
                by = slice[t];
                bx = rem;
            }
            rem -= count[t];
        }
        if (by < 0) return;
    }
    if (by == (n_pass <= 2 ? 1 : 5)) { // last slice of the grid: pass 1, one wave per row
        scan_one_row<CARRY>(gmap, sg, npx, mmap, (BANDS ? bands.band_dn * bands.rows() : 0) + bx * SCAN_SEGS + threadIdx.y, threadIdx.x, tm);
        return;
    }
    __shared__ float seg_a[SCAN_SEGS][64];
    __shared__ int seg_n[SCAN_SEGS][64];
    const int q = (n_pass <= 2) ? 0 : (by == 0 ? 0 : by + 1); // slice -> pass 0, 2, 3, 4, 5
    const ColLines cl = col_lines(q, sg);
    const int lane = threadIdx.x, seg = threadIdx.y;
    const int v = cl.v_lo + bx * 64 + lane;
    if (bx * 64 >= cl.n_lines) return; // whole block out of range (uniform)
    int r0 = 0, r1 = 0;
    if (v < cl.v_lo + cl.n_lines) col_line_rows(q, v, sg, r0, r1);
    const int chunk = CARRY ? bands.chunk : scan_chunk(sg.Hp);
    const int band0 = BANDS ? (q == 5 ? bands.band_up : bands.band_dn) * bands.rows() : 0; // first row of this launch's band of the pass
    const int b = max(r0, band0 + seg * chunk), e = min(r1, band0 + (seg + 1) * chunk);
    const int n = max(e - b, 0);
    // pixel index of relative row r on this line: (top + r) * W + left + v + s r, advanced by dir * (W + s) per step
    const int first = cl.dir > 0 ? b : e - 1;
    const int64_t p0 = (int64_t)(sg.top + first) * sg.W + sg.left + v + cl.s * first;
    const int64_t step = (int64_t)cl.dir * (sg.W + cl.s);
    float *out = CARRY ? nullptr : mmap + (int64_t)min_p_plane(q) * npx;
    float mp = 0.0f;
    if (CARRY && chunk <= SCAN_MAXSEG && npx <= ((int64_t)1 << 29)) { // (with the maps' stores behind it the lean form measured slower: 24 against 19 us)
        // LEAN FORM (scan_lean below) for the waves whose 64 lines all cross every row of the segment or none of it -- all but the
        // waves at the ragged ends of the diagonal passes -- when the segment's row count is one of the compile-time lengths
        const int segu = __builtin_amdgcn_readfirstlane(seg);
        const int nb = band0 + segu * chunk, ne = min(sg.Hp, nb + chunk), n_u = ne - nb; // rows [nb, ne) of the margin box
        const bool ragged = n != 0 && (b != nb || e != ne);
        const bool full = n_u > 0 && (n_u & 3) == 0 && (!CARRY || n_u == chunk) && __builtin_amdgcn_ballot_w64(ragged) == 0;
        if (full || CARRY) {
            // lane_off: the lane's pixel in row nb (in the general form possibly not a pixel of its line: only used inside its steps)
            const uint32_t lane_off = (full && n == 0) ? 0xFFFFFFFFu : (uint32_t)(((int64_t)(sg.top + nb) * sg.W + sg.left + v + cl.s * nb) * 4);
            const LeanArgs la{gmap, out, npx, (uint32_t)(sg.W + cl.s) * 4u, lane_off, cl.dir, segu, lane, n,
                              q, cl.s, cl.v_lo + bx * 64, nb, n_u, cl.v_lo, cl.n_lines, cl.dir > 0 ? b - nb : ne - e, BANDS ? band0 / max(tm.R, 1) : 0,
                              BANDS ? bands.carry_in + (int64_t)by * bands.stride + min(bx * 64 + lane, bands.stride - 1) : nullptr,
                              BANDS ? bands.carry_out + (int64_t)by * bands.stride + min(bx * 64 + lane, bands.stride - 1) : nullptr};
            if (n_u > 0) {
                switch ((n_u + 3) >> 2) {
#define SVH_LEAN(Q)                                                     \
    case Q:                                                             \
        if (full) scan_lean<4 * Q, CARRY, true, BANDS>(la, seg_a, seg_n, tm);  \
        else if constexpr (CARRY) scan_lean<4 * Q, true, false, BANDS>(la, seg_a, seg_n, tm); \
        return;
                    SVH_LEAN(1) SVH_LEAN(2) SVH_LEAN(3) SVH_LEAN(4) SVH_LEAN(5) SVH_LEAN(6) SVH_LEAN(7) SVH_LEAN(8) SVH_LEAN(9)
                    SVH_LEAN(10) SVH_LEAN(11) SVH_LEAN(12) SVH_LEAN(13) SVH_LEAN(14) SVH_LEAN(15) SVH_LEAN(16) SVH_LEAN(17) SVH_LEAN(18)
#undef SVH_LEAN
                default: break;
                }
            }
        }
    }
    if constexpr (CARRY) { // (tiles mode is only launched where scan_lean applies: what arrives here is a segment past the last row)
        seg_a[seg][lane] = 0.0f; // the identity for the chains that walk over it
        seg_n[seg][lane] = 0;
        __syncthreads();
        return;
    }
    if (chunk <= SCAN_MAXSEG) {
        // the whole segment lives in registers: one round of loads (all in flight together), summary, chain, replay
        float gv[SCAN_MAXSEG];
#pragma unroll
        for (int k = 0; k < SCAN_MAXSEG; k++) gv[k] = k < n ? gmap[p0 + k * step] : 0.0f;
#pragma unroll
        for (int k = 0; k < SCAN_MAXSEG; k++)
            if (k < n) mp = gv[k] - mp;
        seg_a[seg][lane] = mp;
        seg_n[seg][lane] = n;
        __syncthreads();
        mp = 0.0f;
        if (cl.dir > 0) {
            for (int t = 0; t < seg; t++) mp = seg_a[t][lane] + ((seg_n[t][lane] & 1) ? -mp : mp);
        } else {
            for (int t = SCAN_SEGS - 1; t > seg; t--) mp = seg_a[t][lane] + ((seg_n[t][lane] & 1) ? -mp : mp);
        }
#pragma unroll
        for (int k = 0; k < SCAN_MAXSEG; k++)
            if (k < n) {
                out[p0 + k * step] = mp;
                mp = gv[k] - mp;
            }
        return;
    }
    {
        int64_t p = p0;
        int k = 0;
        for (; k + 4 <= n; k += 4, p += 4 * step) {
            const float g0 = gmap[p], g1 = gmap[p + step], g2 = gmap[p + 2 * step], g3 = gmap[p + 3 * step];
            mp = g3 - (g2 - (g1 - (g0 - mp)));
        }
        for (; k < n; k++, p += step) mp = gmap[p] - mp;
    }
    seg_a[seg][lane] = mp;
    seg_n[seg][lane] = n;
    __syncthreads();
    // chain the segments that come before this one in traversal order
    mp = 0.0f;
    if (cl.dir > 0) {
        for (int t = 0; t < seg; t++) mp = seg_a[t][lane] + ((seg_n[t][lane] & 1) ? -mp : mp);
    } else {
        for (int t = SCAN_SEGS - 1; t > seg; t--) mp = seg_a[t][lane] + ((seg_n[t][lane] & 1) ? -mp : mp);
    }
    {
        int64_t p = p0;
        int k = 0;
        for (; k + 4 <= n; k += 4, p += 4 * step) {
            const float g0 = gmap[p], g1 = gmap[p + step], g2 = gmap[p + 2 * step], g3 = gmap[p + 3 * step];
            const float m1 = g0 - mp, m2 = g1 - m1, m3 = g2 - m2;
            out[p] = mp;
            out[p + step] = m1;
            out[p + 2 * step] = m2;
            out[p + 3 * step] = m3;
            mp = g3 - m3;
        }
        for (; k < n; k++, p += step) {
            out[p] = mp;
            mp = gmap[p] - mp;
        }
    }
}

// scan_one_row: pass 1 (Left2Right), run by the last grid slice of scan_cols_kernel; one wavefront per image row,
// 64 consecutive pixels per step (coalesced 256-byte loads and stores).  Every pixel is the map x -> g - x, so the min_p seen
// by pixel k is the alternating sum  g_{k-1} - g_{k-2} + ... :  with h_t = (-1)^t g_t and the inclusive prefix sum P of h,
// min_p(k) = (-1)^(k-1) P_{k-1}.  The prefix sum inside a 64-pixel chunk is six DPP adds; exact (integers below 2^24).
template <int CTRL, int ROW_MASK = 0xF> __device__ __forceinline__ float dpp_add_from(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_prefix_sum(float v) {
    v = dpp_add_from<0x111>(v);      // row_shr:1
    v = dpp_add_from<0x112>(v);      // row_shr:2
    v = dpp_add_from<0x114>(v);      // row_shr:4
    v = dpp_add_from<0x118>(v);      // row_shr:8
    v = dpp_add_from<0x142, 0xA>(v); // row_bcast:15 into rows 1 and 3
    v = dpp_add_from<0x143, 0xC>(v); // row_bcast:31 into rows 2 and 3
    return v;
}

constexpr int SCAN_ROW_CHUNKS = 32; // rows up to 2048 pixels are loaded in one go

template <bool CARRY>
__device__ __forceinline__ void scan_one_row(const float *__restrict__ gmap, const ScanGeom &sg, int64_t npx, float *__restrict__ mmap, int row,
                                             int lane, const TileMaps &tm) {
    if (row >= sg.Hp) return;
    const int64_t base = (int64_t)(sg.top + row) * sg.W + sg.left;
    float *out = CARRY ? nullptr : mmap + npx; // pass 1
    float *const crow = CARRY ? tm.base + tm.c1_off() + (int64_t)row * tm.tx() : nullptr; // C1: the prefix sum before each 64-pixel chunk
    const float sgn = (lane & 1) ? -1.0f : 1.0f; // (-1)^lane; chunks start at even pixel indices
    float carry = 0.0f;                            // prefix sum of h over the previous chunks
    auto chunk = [&](float gval, int k) {
        if constexpr (CARRY) {
            if (lane == 0) crow[k >> 6] = carry;
            const float tot = wave_prefix_sum(sgn * gval);
            carry += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), 63));
            return;
        }
        const float P = wave_prefix_sum(sgn * gval) + carry; // inclusive over pixels 0..k
        // min_p(k) = (-1)^(k-1) P_{k-1}: take the neighbour's inclusive sum (lane 0: the carry)
        float Pm1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, P), 0x138, 0xF, 0xF, false)); // wave_shr:1
        if (lane == 0) Pm1 = carry;
        const float mp = -sgn * Pm1;
        if (k < sg.Wp) out[base + k] = mp == 0.0f ? 0.0f : mp; // the reference's first min_p is +0, never -0
        carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, P), 63));
    };
    if (sg.Wp <= 64 * SCAN_ROW_CHUNKS) {
        float gv[SCAN_ROW_CHUNKS];
#pragma unroll
        for (int c = 0; c < SCAN_ROW_CHUNKS; c++) gv[c] = (c * 64 + lane < sg.Wp) ? gmap[base + c * 64 + lane] : 0.0f;
#pragma unroll
        for (int c = 0; c < SCAN_ROW_CHUNKS; c++)
            if (c * 64 < sg.Wp) chunk(gv[c], c * 64 + lane);
    } else {
        for (int c0 = 0; c0 < sg.Wp; c0 += 64) chunk((c0 + lane < sg.Wp) ? gmap[base + c0 + lane] : 0.0f, c0 + lane);
    }
}

__device__ __forceinline__ uint32_t order_key_f(float v) {
    if (v == 0.0f) v = 0.0f;
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// per-pass min_p values of one pixel
struct PixelPasses {
    float mp[6];
    unsigned vis; // bit q set: pass q visits the pixel and its min_p is finite
};

__device__ __forceinline__ PixelPasses load_pixel_passes(const ScanGeom &sg, int n_pass, int i, int j, const float *__restrict__ mmap,
                                                         int64_t npx, int64_t p) {
    PixelPasses pp;
    pp.vis = 0;
    const int ip = i - sg.top, jp = j - sg.left;
    const bool inside = ip >= 0 && ip < sg.Hp && jp >= 0 && jp < sg.Wp;
#pragma unroll
    for (int q = 0; q < 6; q++) {
        pp.mp[q] = 0.0f;
        bool v = inside && q < n_pass;
        if (q == 2) v = v && ip >= jp;
        if (q == 3) v = v && jp >= ip;
        if (q == 4) v = v && ip + jp < sg.Wp;
        if (q == 5) v = v && ip + jp < sg.Hp;
        if (v) {
            pp.mp[q] = mmap[(int64_t)min_p_plane(q) * npx + p];
            if (fabsf(pp.mp[q]) < INFINITY) pp.vis |= 1u << q;
        }
    }
    return pp;
}

// S(p, d) per sgm.h:298-300, literal float form
__device__ __forceinline__ float sgm_value(float c, float t, const PixelPasses &pp, int n_pass) {
    float s = c;
    const bool t_fin = fabsf(t) < INFINITY;
#pragma unroll
    for (int q = 0; q < 6; q++) {
        if (q < n_pass) {
            float act = c + (t - pp.mp[q]);
            float ns = s + (act - c);
            s = ((pp.vis >> q) & 1u) && t_fin ? ns : s;
        }
    }
    return s;
}

// Hamming cost of (pixel, d) straight from the word maps (used for the three refinement taps only)
template <int NW> __device__ __forceinline__ int hamming_global(const CensusGeom &g, const Words<NW> &s, int i, int j, int d) {
    if constexpr (NW == 0) {
        return 0;
    } else {
        const int jt = j + g.sign * (g.disp_lower + d);
        const bool in = jt >= 0 && jt < g.Wt;
        const uint32_t *t = g.tw + ((int64_t)i * g.Wt + (in ? jt : 0)) * NW;
        int acc = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) acc += __popc(s.v[w] ^ (in ? t[w] : 0u));
        return acc;
    }
}

// Exact regime, per pixel: S = (1 + n) c + n Pout [d >= dsplit] - sum of the visiting passes' min_p.  The winner is
// the better of the two regional winners (the later region wins ties, as the reference's '<=' scan does).
//
// The sum of the min_p is the same for every disparity of a pixel, so it moves S without moving the winner: where only the
// index / disparity maps are wanted (MAPS = false) neither the g map, the line scans nor the six min_p maps are needed, and n,
// the number of passes that visit the pixel, follows from its position (sgm.h:329-354, finding F5).  What the reference's Cost
// branch hands from pixel to pixel (finding F4: one scalar per pass) never reaches its own argmin.  The S values themselves
// (refinement taps, cross-shard value keys) still take the maps.
template <int NW>
__device__ __forceinline__ void finalize_px(const CensusGeom &g, const WinnerOut &out, int i, int j, int64_t p, int msum, int n_vis, float Pout, uint2 k);

template <int NW, bool MAPS>
__global__ void __launch_bounds__(256) census_finalize_kernel(CensusGeom g, ScanGeom sg, int n_pass, float Pout, const float *__restrict__ mmap,
                                                              const uint2 *__restrict__ keys, WinnerOut out) {
    const int64_t npx = (int64_t)g.H * g.Ws;
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y; // (a row of blocks per image row: no 64-bit division per pixel)
    if (j >= g.Ws) return;
    const int64_t p = (int64_t)i * g.Ws + j;
    int msum = 0, n_vis = 0;
    if constexpr (MAPS) {
        const PixelPasses pp = load_pixel_passes(sg, n_pass, i, j, mmap, npx, p);
#pragma unroll
        for (int q = 0; q < 6; q++)
            if ((pp.vis >> q) & 1u) {
                n_vis++;
                msum += (int)pp.mp[q];
            }
    } else {
        const int ip = i - sg.top, jp = j - sg.left;
        if (ip >= 0 && ip < sg.Hp && jp >= 0 && jp < sg.Wp) {
#pragma unroll
            for (int q = 0; q < 6; q++) n_vis += q < n_pass && pass_visits_px(q, ip, jp, sg.Hp, sg.Wp);
        }
    }
    finalize_px<NW>(g, out, i, j, p, msum, n_vis, Pout, keys[p]);
}

template <int NW>
__device__ __forceinline__ void finalize_px(const CensusGeom &g, const WinnerOut &out, int i, int j, int64_t p, int msum, int n_vis, float Pout, uint2 k) {
    const int mul = 1 + n_vis, k0 = -msum, k1 = n_vis * (int)Pout - msum;
    const int v0 = mul * key_cost(k.x) + k0, d0 = key_index(k.x);
    const int v1 = mul * key_cost(k.y) + k1, d1 = key_index(k.y);
    const bool take1 = k.y != KEY_NONE && (k.x == KEY_NONE || v1 <= v0);
    const int best = take1 ? v1 : v0, bd = take1 ? d1 : d0;
    if (out.idx) out.idx[p] = bd;
    if (out.disp) out.disp[p] = out.disp_sign * bd + out.disp_offset;
    if (out.keys) out.keys[p] = ((unsigned long long)order_key_f((float)best) << 32) | (uint32_t)(out.key_total - 1 - (out.key_offset + bd));
    if (out.taps) { // truncatedCostVolume<Same>, radius 1 (correlation_base.h:601-613)
        const Words<NW> s = load_source_words<NW>(g, p);
        const int dsplit = min(max(g.Ws - j, 0), g.D);
        const bool px_bad = j < out.taps_h_r || i < out.taps_v_r || i + out.taps_v_r >= g.H;
#pragma unroll
        for (int tap = 0; tap < 3; tap++) {
            const int pd = bd + tap - 1;
            float v = __uint_as_float(0x7FC00000u);
            if (!(px_bad || pd < 0 || pd >= g.D || j + pd + out.taps_h_r >= g.Ws))
                v = (float)(mul * hamming_global<NW>(g, s, i, j, pd) + (pd >= dsplit ? k1 : k0));
            out.taps[p * 3 + tap] = v;
        }
    }
}

// ---- the per-pixel kernel without the min_p maps (exact regime, 8 directions, no margins, segments that fit the registers) ----------
// The six maps cost the scans 50 MB of stores and the per-pixel kernel 50 MB of loads (1080p), and both kernels run at what the memory
// system gives for that; the recurrences themselves are a few instructions per pixel.  So the maps are not written: scan_cols_kernel<true>
// keeps the min_p values on the edges of tiles of R rows x 64 columns (TileMaps, 3 % of the maps) and this kernel, a wave per tile,
// REPLAYS the recurrences inside its tile -- the same subtraction per pixel in the same order as the scans -- with the lanes as the
// tile's columns for every family: going one row down, the min_p a diagonal line carries moves one lane to the right (v_mov_dpp
// wave_shr:1), an anti-diagonal one to the left, a vertical one stays; the lane a line enters the tile through takes its value from S
// (or 0 where the line starts: the image border).  Pass 5 walks the rows upwards.  The sum of the visiting passes' min_p per pixel
// stays in registers, then keys -> winner -> outputs as in census_finalize_kernel.  g is read once.
constexpr int TILE_RMAX = SCAN_MAXSEG / TILE_SUB;
__device__ __forceinline__ float dpp_wave_shr1(float src, float old) { // lane l <- lane l - 1; lane 0 <- old
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_wave_shl1(float src, float old) { // lane l <- lane l + 1; lane 63 <- old
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x130, 0xF, 0xF, false));
}
template <int NW>
__global__ void __launch_bounds__(256) tile_finalize_kernel(CensusGeom g, ScanGeom sg, float Pout, const float *__restrict__ gmap, TileMaps tm,
                                                            const uint2 *__restrict__ keys, WinnerOut out) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Hp = sg.Hp, Wp = sg.Wp, W = sg.W, R = tm.R;
    const int txi = blockIdx.x, y = blockIdx.y * 4 + wave;
    const int top = y * R;
    if (top >= Hp) return;
    const int n_r = min(R, Hp - top), c0 = txi * 64, c = c0 + lane;
    const bool col_in = c < Wp;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)gmap, 0, (int)((int64_t)g.H * W * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)tm.base, 0, (int)(tm.total() * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void *)keys, 0, (int)((int64_t)g.H * g.Ws * 8), 0x00020000);
    auto ld = [&](const __amdgpu_buffer_rsrc_t &r, int64_t index, bool on) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, on ? (uint32_t)(index * 4) : 0xFFFFFFFFu, 0, 0));
    };
    // everything the tile needs, requested before anything is used.  32-bit byte offsets, one add per row; a lane past the last column
    // starts at 2^31 (out of range for every row of the tile: the maps are below 2^30 bytes), a row past the last row is out of range
    // by itself: both read zeros
    float gv[TILE_RMAX];
    uint2 kv[TILE_RMAX];
    const uint32_t g0 = col_in ? (uint32_t)((top * W + c) * 4) : 0x80000000u, k0 = col_in ? (uint32_t)((top * g.Ws + c) * 8) : 0x80000000u;
#pragma unroll
    for (int k = 0; k < TILE_RMAX; k++) {
        gv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, g0 + (uint32_t)(k * W * 4), 0, 0));
        const auto kk = __builtin_amdgcn_raw_buffer_load_b64(rk, k0 + (uint32_t)(k * g.Ws * 8), 0, 0);
        kv[k] = make_uint2(kk[0], kk[1]);
    }
    const int bot = top + n_r - 1;
    float mp0 = ld(rt, tm.e_off(0) + (int64_t)y * Wp + c, col_in), mp1d = ld(rt, tm.e_off(1) + (int64_t)y * Wp + c, col_in),
          mp4 = ld(rt, tm.e_off(2) + (int64_t)y * Wp + c, col_in), mp5 = ld(rt, tm.e_off(3) + (int64_t)y * Wp + c, col_in);
    // the values lines bring in through the tile's sides, one per row transition, laid out so that the lane that takes the next one
    // holds it after a rotation by one lane per step:
    //   diagonal (enters lane 0 at row top + 1 + k after step k): lane k <- S1[tx][top + 1 + k], rotated left
    //   pass 4 (enters lane 63 at row top + 1 + k): lane 63 - k <- S2[tx][top + 1 + k], rotated right
    //   pass 5 (upwards; enters lane 0 at row bot - 1 - k after step k): lane k <- S3[tx][bot - 1 - k], rotated left
    // A line that starts at the image border brings 0.
    const bool left_nb = c0 > 0, right_nb = c0 + 64 < Wp;
    float s1 = ld(rt, tm.s_off(1) + (int64_t)txi * Hp + top + 1 + lane, left_nb && lane < TILE_RMAX && top + 1 + lane < Hp);
    float s2 = ld(rt, tm.s_off(2) + (int64_t)txi * Hp + top + 1 + (63 - lane), right_nb && 63 - lane < TILE_RMAX && top + 1 + (63 - lane) < Hp);
    float s3 = ld(rt, tm.s_off(3) + (int64_t)txi * Hp + bot - 1 - lane, left_nb && lane < TILE_RMAX && bot - 1 - lane >= 0);
    const float c1 = ld(rt, tm.c1_off() + (int64_t)(top + lane) * tm.tx() + txi, lane < TILE_RMAX && top + lane < Hp); // lane k: row top + k
    const float sgn = (lane & 1) ? -1.0f : 1.0f;
    float acc[TILE_RMAX];
    // downwards: vertical, diagonal (counted twice on the line both pass 2 and pass 3 run: finding F5), pass 4 where it visits; and the rows
#pragma unroll
    for (int k = 0; k < TILE_RMAX; k++) {
        const int i = top + k;
        float a = mp0 + mp1d;
        a += (i == c) ? mp1d : 0.0f;
        a += (i + c < Wp) ? mp4 : 0.0f;
        // pass 1: min_p(col) = (-1)^(col - 1) P_{col - 1}, P the inclusive prefix sum of (-1)^col g over the row (scan_one_row)
        const float carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c1), k));
        const float P = wave_prefix_sum(sgn * gv[k]) + carry;
        a += -sgn * dpp_wave_shr1(P, carry);
        acc[k] = a;
        mp0 = gv[k] - mp0;
        mp1d = dpp_wave_shr1(gv[k] - mp1d, s1);
        s1 = dpp_wave_shl1(s1, 0.0f);
        mp4 = dpp_wave_shl1(gv[k] - mp4, s2);
        s2 = dpp_wave_shr1(s2, 0.0f);
    }
    // upwards: pass 5 where it visits
#pragma unroll
    for (int k = TILE_RMAX - 1; k >= 0; k--) {
        if (k < n_r) { // (wave uniform)
            const int i = top + k;
            acc[k] += (i + c < Hp) ? mp5 : 0.0f;
            mp5 = dpp_wave_shr1(gv[k] - mp5, s3);
            s3 = dpp_wave_shl1(s3, 0.0f);
        }
    }
    const int p0 = top * g.Ws + c; // (pixel index: the image is below 2^28 pixels)
#pragma unroll
    for (int k = 0; k < TILE_RMAX; k++) {
        if (k < n_r && col_in) {
            const int i = top + k, sdiag = i + c;
            const int n_vis = 3 + (int)(i == c) + (int)(sdiag < Wp) + (int)(sdiag < Hp); // passes 0, 1, one of 2 / 3 (both on the corner diagonal), 4, 5
            finalize_px<NW>(g, out, i, c, p0 + k * g.Ws, (int)acc[k], n_vis, Pout, kv[k]);
        }
    }
}

// apply + extractSelectedIndex (+ truncated taps, + reduction keys) without writing any volume: literal float
// evaluation of sgm.h:287-300 per voxel, for census costs outside the exact regime
template <int NW>
__global__ void __launch_bounds__(TJ) census_apply_select_kernel(CensusGeom g, ScanGeom sg, int n_pass, float Pout,
                                                                 const float *__restrict__ mmap, WinnerOut out) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int i = blockIdx.y, j0 = blockIdx.x * TJ, tj = threadIdx.x, j = j0 + tj;
    stage_target_window<NW>(g, i, j0, lds);
    __syncthreads();
    if (j >= g.Ws) return;
    const int64_t npx = (int64_t)g.H * g.Ws, p = (int64_t)i * g.Ws + j;
    const Words<NW> s = load_source_words<NW>(g, p);
    const uint32_t *base = lds + (g.sign > 0 ? tj : TJ - 1 - tj) * NW;
    const PixelPasses pp = load_pixel_passes(sg, n_pass, i, j, mmap, npx, p);
    const int dsplit = min(max(g.Ws - j, 0), g.D); // d >= dsplit pays Pout
    float best = 0.0f; // S at the winner
    int bd = 0;        // winner: sequential '<=' scan of extractSelectedIndex (correlation_base.h:441-455)
    for_each_disparity<NW>(s, base, g.D, [&](int d, int ci) {
        const float c = (float)ci;
        const float t = (d >= dsplit) ? c + Pout : c;
        const float v = sgm_value(c, t, pp, n_pass);
        if (d == 0 || v <= best) {
            best = v;
            bd = d;
        }
    });
    if (out.idx) out.idx[p] = bd;
    if (out.disp) out.disp[p] = out.disp_sign * bd + out.disp_offset;
    if (out.keys) { // (ordered value, total-1-global index): a min-reduction keeps ties on the largest index
        unsigned long long key = ((unsigned long long)order_key_f(best) << 32) | (uint32_t)(out.key_total - 1 - (out.key_offset + bd));
        if (isnan(best)) key = (out.key_offset == 0 && bd == 0) ? (unsigned long long)(uint32_t)(out.key_total - 1) : ~0ull;
        out.keys[p] = key;
    }
    if (out.taps) { // truncatedCostVolume<Same>, radius 1 (correlation_base.h:601-613)
        const bool px_bad = j < out.taps_h_r || i < out.taps_v_r || i + out.taps_v_r >= g.H;
#pragma unroll
        for (int tap = 0; tap < 3; tap++) {
            const int pd = bd + tap - 1;
            float v = __uint_as_float(0x7FC00000u);
            if (!(px_bad || pd < 0 || pd >= g.D || j + pd + out.taps_h_r >= g.Ws)) {
                const float c = (float)hamming_rec<NW>(s, base + pd * NW);
                v = sgm_value(c, (pd >= dsplit) ? c + Pout : c, pp, n_pass);
            }
            out.taps[p * 3 + tap] = v;
        }
    }
}

size_t lds_bytes(int nWw, int D) { return (size_t)(nWw ? nWw : 1) * (TJ + D - 1) * sizeof(uint32_t); }

template <int NW> static int launch_sweep_valu(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, int merge) {
    // block width: the one that pads the row less (the lanes of a partial last block idle through the whole disparity loop);
    // 1080p: 5 blocks of 384 pixels instead of 3.75 of 512 -> 100 us instead of 106
    const int64_t pad512 = (int64_t)ceil_div(g.Ws, 512) * 512, pad384 = (int64_t)ceil_div(g.Ws, 384) * 384;
    if (pad384 < pad512) {
        dim3 grid(ceil_div(g.Ws, 384), g.H);
        const size_t shmem = (size_t)(NW ? NW : 1) * sweep_records(g.D, 384) * sizeof(uint32_t);
        SVH_LAUNCH(ctx, "census_sweep", (census_sweep_kernel<NW, 192>), grid, 192, shmem, g, Pout, keys, gmap, merge);
    } else {
        dim3 grid(ceil_div(g.Ws, PX), g.H);
        const size_t shmem = (size_t)(NW ? NW : 1) * sweep_records(g.D) * sizeof(uint32_t);
        SVH_LAUNCH(ctx, "census_sweep", (census_sweep_kernel<NW, TJ>), grid, TJ, shmem, g, Pout, keys, gmap, merge);
    }
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

template <int NW> int launch_sweep(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap) {
    // engines: 0 = automatic (FP4 matrix-core sweep where it applies, else vector ALU), 1 = vector ALU, 3 = FP4 matrix cores
    if (ctx->census_sweep_mode != 1) {
        int status = SVH_OK;
        if (launch_sweep_pm(ctx, g, Pout, keys, gmap, &status)) return status;
        // a disparity count that is no multiple of 32: the matrix cores take the leading multiple, the vector ALU the D % 32 disparities
        // behind it as a second disparity shard (global indices in the keys, MIN-merged into the first part's: D = 250 at 1080p 0.13 -> 0.09 ms)
        const int D32 = std::min(g.D & ~31, 992); // (also ranges longer than the matrix-core sweep's 992)
        if (D32 >= 32 && D32 < g.D) {
            CensusGeom ga = g;
            ga.D = D32;
            if (launch_sweep_pm(ctx, ga, Pout, keys, nullptr, &status)) {
                if (status != SVH_OK) return status;
                CensusGeom gb = g;
                gb.D = g.D - D32;
                gb.d_offset = g.d_offset + D32;
                gb.disp_lower = g.disp_lower + D32;
                return launch_sweep_valu<NW>(ctx, gb, Pout, keys, gmap, 1);
            }
        }
    }
    return launch_sweep_valu<NW>(ctx, g, Pout, keys, gmap, 0);
}

template <int NW>
int launch_finalize(svh_context *ctx, const CensusGeom &g, const ScanGeom &sg, int n_pass, float Pout, const float *mmap, const uint2 *keys,
                    const WinnerOut &out) {
    if (g.H > 65535) return fail(ctx, SVH_ERR_UNSUPPORTED, "images taller than 65535 rows");
    const dim3 grid(ceil_div(g.Ws, 256), g.H);
    if (mmap) SVH_LAUNCH(ctx, "census_finalize", (census_finalize_kernel<NW, true>), grid, 256, 0, g, sg, n_pass, Pout, mmap, keys, out);
    else if ((out.taps && !out.taps_up_to_shift) || out.keys) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "census finalize: S values need the min_p maps");
    else SVH_LAUNCH(ctx, "census_finalize", (census_finalize_kernel<NW, false>), grid, 256, 0, g, sg, n_pass, Pout, mmap, keys, out);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

template <int NW>
int launch_apply_select(svh_context *ctx, const CensusGeom &g, const ScanGeom &sg, int n_pass, float Pout, const float *mmap,
                        const WinnerOut &out) {
    dim3 grid(ceil_div(g.Ws, TJ), g.H);
    SVH_LAUNCH(ctx, "census_apply_select", census_apply_select_kernel<NW>, grid, TJ, lds_bytes(NW, g.D), g, sg, n_pass, Pout, mmap, out);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

} // namespace

// largest nWw the pixel-per-lane kernels are instantiated for (15x15 windows and smaller; 9x9 -> 2 words, 13x13 -> 5, 15x15 -> 7).  The
// matrix-core sweeps take every count up to it (svh_census_sweep_pm.hip; the RightToLeft specialisation since round 5 too).
// Key invariant (svh_census_keys.h): cost <= 32 nWw <= 256, i.e. 9 cost bits above the 12 index bits -- a positive int32 with room to
// spare, which is what lets the cross-shard exchange be an int32 MIN all-reduce.
static constexpr int kMaxWords = 8;
static_assert(32 * kMaxWords < (1 << (31 - KEY_IDX_BITS)), "the largest census cost must fit above the index bits of a positive int32 key");

// LDS budget (nWw records of 256 + D - 1 pixels within the 64 KiB a block gets by default), at most 1024 disparities per
// call, 4096 over all shards (12 index bits in the winner key), costs <= 32 nWw <= 256
bool census_lane_kernels_available(int nWw, int D) {
    return nWw <= kMaxWords && D <= 1024 && (size_t)(nWw ? nWw : 1) * sweep_records(D) * sizeof(uint32_t) <= 60 * 1024;
}
int census_max_total_disparities() { return 1 << KEY_IDX_BITS; }

bool census_exact_regime(const SgmArgs &a, int nWw) {
    if (!std::isfinite(a.Pout) || a.Pout != std::nearbyint(a.Pout)) return false;
    const double cmax = 32.0 * nWw, gmax = 2.0 * cmax + std::fabs((double)a.Pout);
    const double L = (double)std::max(a.H, a.W);
    return 8.0 * gmax * (L + 2.0) < 16777216.0;
}

#define SVH_NW_DISPATCH(nWw, CALL)                                                              \
    switch (nWw) {                                                                              \
    case 0: return CALL(0);                                                                     \
    case 1: return CALL(1);                                                                     \
    case 2: return CALL(2);                                                                     \
    case 3: return CALL(3);                                                                     \
    case 4: return CALL(4);                                                                     \
    case 5: return CALL(5);                                                                     \
    case 6: return CALL(6);                                                                     \
    case 7: return CALL(7);                                                                     \
    case 8: return CALL(8);                                                                     \
    default: return fail(ctx, SVH_ERR_UNSUPPORTED, "census lane kernels: %d words", nWw);       \
    }

static int sweep_dispatch(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap) {
#define CALL(N) launch_sweep<N>(ctx, g, Pout, keys, gmap)
    SVH_NW_DISPATCH(g.nWw, CALL)
#undef CALL
}

int dev_census_sweep(svh_context *ctx, const SgmArgs &a, const CostSource &cs, uint2 *keys, float *gmap) {
    CensusGeom g{cs.src_words, cs.tgt_words, cs.nWw, a.H, a.W, cs.Wt, a.D, cs.sign, cs.disp_lower, cs.d_offset, cs.region1_global_last};
    return sweep_dispatch(ctx, g, a.Pout, keys, gmap);
}

// index / disparity maps alone in the exact regime: the matrix-core sweep writes them itself; other geometries go through keys
int dev_census_winner(svh_context *ctx, Scratch &scr, const SgmArgs &a, const CostSource &cs, const WinnerOut &win) {
    const bool band = a.full_H > 0;
    const int Hp = (band ? a.full_H : a.H) - a.top - a.bottom, Wp = a.W - a.left - a.right;
    const int n_pass = (Hp > 0 && Wp > 0) ? (a.n_dir >= 8 ? 6 : (a.n_dir >= 4 ? 2 : 0)) : 0;
    CensusGeom g{cs.src_words, cs.tgt_words, cs.nWw, a.H, a.W, cs.Wt, a.D, cs.sign, cs.disp_lower, cs.d_offset, cs.region1_global_last};
    if (ctx->census_sweep_mode == 0 || ctx->census_sweep_mode == 3) {
        SweepWinner sw;
        sw.idx = win.idx;
        sw.disp = win.disp;
        sw.disp_sign = win.disp_sign;
        sw.disp_offset = win.disp_offset;
        sw.top = a.top - a.row_origin;
        if (band) {
            sw.row_first = a.store_row0;
            sw.row_count = a.store_rows;
        }
        sw.left = a.left;
        sw.Hp = Hp > 0 ? Hp : 0;
        sw.Wp = Wp > 0 ? Wp : 0;
        sw.n_pass = n_pass;
        sw.pout = (int)a.Pout;
        int status = SVH_OK;
        if (launch_sweep_pm(ctx, g, a.Pout, nullptr, nullptr, &status, &sw)) return status;
    }
    if (band) return fail(ctx, SVH_ERR_UNSUPPORTED, "row bands need the matrix-core sweep (disparity count a multiple of 32 up to 992, or 33 .. 512 for a RightToLeft range that ends at the image edge; at most 8 census words)");
    uint2 *keys = scr.get_n<uint2>((size_t)a.H * a.W);
    if (!keys) return SVH_ERR_OUT_OF_MEMORY;
    SVH_TRY(sweep_dispatch(ctx, g, a.Pout, keys, nullptr));
    return dev_census_finalize(ctx, a, cs, nullptr, keys, win);
}

int dev_census_scans(svh_context *ctx, const SgmArgs &a, const uint2 *keys, float *gmap, bool gmap_ready, float *mmap, const int *skip_if_nonzero) {
    const int Hp = a.H - a.top - a.bottom, Wp = a.W - a.left - a.right;
    const int n_pass = (Hp > 0 && Wp > 0) ? (a.n_dir >= 8 ? 6 : (a.n_dir >= 4 ? 2 : 0)) : 0;
    if (n_pass == 0) return SVH_OK;
    const int64_t npx = (int64_t)a.H * a.W;
    if (!gmap_ready) {
        SVH_LAUNCH(ctx, "gmap_from_keys", gmap_from_keys_kernel, grid_for(npx, 256), 256, 0, keys, npx, a.Pout, gmap);
        SVH_CHECK_LAUNCH(ctx);
    }
    ScanGeom sg{a.top, a.left, Hp, Wp, a.W};
    // one launch: grid slices 0..4 (or 0) are the passes that cross the rows, the last slice is pass 1 (one wave per row)
    dim3 cgrid(std::max(ceil_div(std::max(Hp, Wp), 64), ceil_div(Hp, SCAN_SEGS)), (n_pass == 6 ? 5 : 1) + 1), cblock(64, SCAN_SEGS);
    SVH_LAUNCH(ctx, "sgm_line_scans", scan_cols_kernel<false>, cgrid, cblock, 0, gmap, sg, npx, n_pass, mmap, skip_if_nonzero, TileMaps{}, ScanBands{});
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

template <int NW>
static int launch_tile_finalize(svh_context *ctx, dim3 tgrid, const CensusGeom &g, const ScanGeom &sg, float Pout, const float *gmap, const TileMaps &tm,
                                const uint2 *keys, const WinnerOut &win) {
    SVH_LAUNCH(ctx, "census_finalize", tile_finalize_kernel<NW>, tgrid, 256, 0, g, sg, Pout, gmap, tm, keys, win);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

// sweep -> tile-edge values of the line scans -> replay + per-pixel kernel, no min_p maps (tile_finalize_kernel): whether the geometry allows it
bool census_tiles_apply(const svh_context *ctx, const SgmArgs &a) {
    const bool whole = a.top == 0 && a.left == 0 && a.bottom == 0 && a.right == 0 && a.row_origin == 0 && a.full_H == 0 && a.store_rows == 0;
    // (taller than one launch's 1152 rows: bands of rows, ScanBands; tile_finalize_kernel addresses keys with 32-bit byte offsets)
    return ctx->census_tiles && whole && a.n_dir >= 8 && a.n_dir != 16 && a.H > 0 && a.W > 0 && a.H <= 65535 && (int64_t)a.H * a.W <= ((int64_t)1 << 27);
}

// keys (and g, unless the caller has it already) -> tile-edge values of the line scans -> replay + per-pixel kernel
int dev_census_tiles_from_keys(svh_context *ctx, Scratch &scr, const SgmArgs &a, const CostSource &cs, const uint2 *keys, float *gmap, bool gmap_ready,
                               const WinnerOut &win) {
    const int64_t npx = (int64_t)a.H * a.W;
    ScanBands bands = scan_bands_for(a.H);
    TileMaps tm;
    tm.Hp = a.H;
    tm.Wp = a.W;
    tm.R = bands.chunk / TILE_SUB;
    tm.base = scr.get_n<float>((size_t)tm.total());
    if (!tm.base) return SVH_ERR_OUT_OF_MEMORY;
    if (!gmap_ready) {
        SVH_LAUNCH(ctx, "gmap_from_keys", gmap_from_keys_kernel, grid_for(npx, 256), 256, 0, keys, npx, a.Pout, gmap);
        SVH_CHECK_LAUNCH(ctx);
    }
    float *carry[2] = {nullptr, nullptr};
    if (bands.n > 1) {
        bands.stride = ceil_div(std::max(a.H, a.W), 64) * 64;
        carry[0] = scr.get_n<float>((size_t)10 * bands.stride);
        if (!carry[0]) return SVH_ERR_OUT_OF_MEMORY;
        carry[1] = carry[0] + (size_t)5 * bands.stride;
        SVH_HIP_CHECK(ctx, hipMemsetAsync(carry[0], 0, (size_t)5 * bands.stride * sizeof(float), ctx->stream));
    }
    ScanGeom sg{0, 0, a.H, a.W, a.W};
    // one block per 64 lines of a pass (3 x over the columns, 2 x over the rows) and per 16 rows of the row pass, nothing else
    // (against six slices as wide as the widest, 408 blocks of which 216 leave at once: 16.4 -> 16.2 us, rocprofv3, same box)
    for (int k = 0; k < bands.n; k++) {
        bands.band_dn = k;
        bands.band_up = bands.n - 1 - k;
        bands.carry_in = carry[k & 1];
        bands.carry_out = carry[(k + 1) & 1];
        const int band_rows = std::min(bands.rows(), a.H - k * bands.rows());
        dim3 cgrid(3 * ceil_div(a.W, 64) + 2 * ceil_div(a.H, 64) + ceil_div(band_rows, SCAN_SEGS)), cblock(64, SCAN_SEGS);
        if (bands.n > 1) SVH_LAUNCH(ctx, "sgm_line_scans", (scan_cols_kernel<true, true>), cgrid, cblock, 0, gmap, sg, npx, 6, nullptr, nullptr, tm, bands);
        else SVH_LAUNCH(ctx, "sgm_line_scans", (scan_cols_kernel<true, false>), cgrid, cblock, 0, gmap, sg, npx, 6, nullptr, nullptr, tm, bands);
        SVH_CHECK_LAUNCH(ctx);
    }
    CensusGeom g{cs.src_words, cs.tgt_words, cs.nWw, a.H, a.W, cs.Wt, a.D, cs.sign, cs.disp_lower, cs.d_offset, cs.region1_global_last};
    dim3 tgrid(tm.tx(), ceil_div(tm.ny(), 4));
#define CALL(N) launch_tile_finalize<N>(ctx, tgrid, g, sg, a.Pout, gmap, tm, keys, win)
    SVH_NW_DISPATCH(cs.nWw, CALL)
#undef CALL
}

int dev_census_sweep_tiles(svh_context *ctx, Scratch &scr, const SgmArgs &a, const CostSource &cs, const WinnerOut &win) {
    const int64_t npx = (int64_t)a.H * a.W;
    uint2 *keys = scr.get_n<uint2>((size_t)npx);
    float *gmap = scr.get_n<float>((size_t)npx);
    if (!keys || !gmap) return SVH_ERR_OUT_OF_MEMORY;
    SVH_TRY(dev_census_sweep(ctx, a, cs, keys, gmap));
    return dev_census_tiles_from_keys(ctx, scr, a, cs, keys, gmap, true, win);
}

int dev_census_sweep_and_scans(svh_context *ctx, Scratch &scr, const SgmArgs &a, const CostSource &cs, float *mmap, uint2 **keys_out) {
    const int64_t npx = (int64_t)a.H * a.W;
    uint2 *keys = scr.get_n<uint2>((size_t)npx);
    float *gmap = scr.get_n<float>((size_t)npx);
    if (!keys || !gmap) return SVH_ERR_OUT_OF_MEMORY;
    *keys_out = keys;
    SVH_TRY(dev_census_sweep(ctx, a, cs, keys, gmap));
    return dev_census_scans(ctx, a, keys, gmap, true, mmap, nullptr);
}

int dev_census_finalize(svh_context *ctx, const SgmArgs &a, const CostSource &cs, const float *mmap, const uint2 *keys, const WinnerOut &win) {
    const int Hp = a.H - a.top - a.bottom, Wp = a.W - a.left - a.right;
    const int n_pass = (Hp > 0 && Wp > 0) ? (a.n_dir >= 8 ? 6 : (a.n_dir >= 4 ? 2 : 0)) : 0;
    CensusGeom g{cs.src_words, cs.tgt_words, cs.nWw, a.H, a.W, cs.Wt, a.D, cs.sign, cs.disp_lower, cs.d_offset, cs.region1_global_last};
    ScanGeom sg{a.top, a.left, Hp > 0 ? Hp : 0, Wp > 0 ? Wp : 0, a.W};
#define CALL(N) launch_finalize<N>(ctx, g, sg, n_pass, a.Pout, mmap, keys, win)
    SVH_NW_DISPATCH(cs.nWw, CALL)
#undef CALL
}

int dev_census_apply_select(svh_context *ctx, const SgmArgs &a, const CostSource &cs, const float *mmap, const WinnerOut &win) {
    const int Hp = a.H - a.top - a.bottom, Wp = a.W - a.left - a.right;
    const int n_pass = (Hp > 0 && Wp > 0) ? (a.n_dir >= 8 ? 6 : (a.n_dir >= 4 ? 2 : 0)) : 0;
    CensusGeom g{cs.src_words, cs.tgt_words, cs.nWw, a.H, a.W, cs.Wt, a.D, cs.sign, cs.disp_lower, cs.d_offset, cs.region1_global_last};
    ScanGeom sg{a.top, a.left, Hp > 0 ? Hp : 0, Wp > 0 ? Wp : 0, a.W};
#define CALL(N) launch_apply_select<N>(ctx, g, sg, n_pass, a.Pout, mmap, win)
    SVH_NW_DISPATCH(cs.nWw, CALL)
#undef CALL
}

} // namespace svh

// Shared by the census + SGM kernels of the exact regime (svh_census_sgm.hip) and the matrix-core sweep
// (svh_census_sweep_pm.hip, svh_census_sweep_rl.hip): the geometry of a sweep and the winner keys.
#pragma once

#include "svh_internal.h"

namespace svh {

struct CensusGeom {
    const uint32_t *sw, *tw; // compact words (H, Ws, nWw), (H, Wt, nWw); target pre-rounded through float (E2)
    int nWw, H, Ws, Wt, D, sign, disp_lower;
    int d_offset; // global index of local disparity 0 (disparity shards); disp_lower already includes it
    int region1_global_last = -1; // CostSource::region1_global_last
};


constexpr uint32_t KEY_NONE = 0x7FFFFFFFu;
constexpr int KEY_IDX_BITS = 12;
constexpr uint32_t KEY_IDX_MASK = (1u << KEY_IDX_BITS) - 1u;
__device__ __forceinline__ uint32_t make_key(int c, int d_global) { return ((uint32_t)c << KEY_IDX_BITS) | (KEY_IDX_MASK - (uint32_t)d_global); }
// the Pout region's key of pixel j when CensusGeom::region1_global_last >= 0 (ones = popcount of the pixel's source census words)
__device__ __forceinline__ uint32_t global_region1_key(const CensusGeom &g, int j, int ones) {
    return j + g.region1_global_last >= g.Ws ? make_key(ones, g.region1_global_last) : KEY_NONE;
}
__device__ __forceinline__ int key_cost(uint32_t k) { return (int)(k >> KEY_IDX_BITS); }
__device__ __forceinline__ int key_index(uint32_t k) { return (int)(KEY_IDX_MASK - (k & KEY_IDX_MASK)); }

__device__ __forceinline__ float g_from_keys(uint32_t key0, uint32_t key1, int pout) {
    const int g0 = key0 == KEY_NONE ? (1 << 24) : 2 * key_cost(key0);
    const int g1 = key1 == KEY_NONE ? (1 << 24) : 2 * key_cost(key1) + pout;
    return (float)min(g0, g1);
}

// does pass q (0..5, svh_sgm.hip) visit pixel (ip, jp) of the margin box? (sgm.h:329-354, finding F5)
__host__ __device__ __forceinline__ bool pass_visits_px(int q, int ip, int jp, int Hp, int Wp) {
    return q < 2 || (q == 2 && ip >= jp) || (q == 3 && jp >= ip) || (q == 4 && ip + jp < Wp) || (q == 5 && ip + jp < Hp);
}

// The winner of a pixel from its two regional keys when only the index / disparity maps are wanted (exact regime):
// S = (1 + n) c + n Pout [Pout region] - (a sum that is the same for every disparity of the pixel), n = passes visiting the pixel;
// the later region wins ties (the reference's '<=' scan).  Written by the sweep itself (no keys, no finalize launch) or by
// census_finalize_kernel from stored keys.
struct SweepWinner {
    int32_t *idx = nullptr, *disp = nullptr;
    int disp_sign = 1, disp_offset = 0;
    int top = 0, left = 0, Hp = 0, Wp = 0, n_pass = 0, pout = 0; // (top may be negative: row bands, see SgmArgs)
    int row_first = 0, row_count = 0x7fffffff;                  // rows stored: [row_first, row_first + row_count), to output row i - row_first
    __host__ __device__ bool on() const { return idx || disp; }
};
__device__ __forceinline__ int passes_visiting(const SweepWinner &w, int i, int j) { // (n_pass is 0, 2 or 6; branch-free)
    const int ip = i - w.top, jp = j - w.left, s = ip + jp;
    const bool inside = (unsigned)ip < (unsigned)w.Hp && (unsigned)jp < (unsigned)w.Wp;
    const int n6 = 2 + (int)(ip >= jp) + (int)(jp >= ip) + (int)(s < w.Wp) + (int)(s < w.Hp);
    return inside ? (w.n_pass == 6 ? n6 : w.n_pass) : 0;
}
__device__ __forceinline__ int winner_index(uint32_t k0, uint32_t k1, int n_vis, int pout) {
    const int v0 = (1 + n_vis) * key_cost(k0), v1 = (1 + n_vis) * key_cost(k1) + n_vis * pout;
    const bool take1 = k1 != KEY_NONE && (k0 == KEY_NONE || v1 <= v0);
    return take1 ? key_index(k1) : key_index(k0);
}

// census_sweep on the matrix cores: 4-bit (FP4) operands +-1, persistent blocks (svh_census_sweep_pm.hip); false when the geometry is
// outside what the kernel covers (the caller then runs the vector-ALU sweep)
// (winner: write the index / disparity maps instead of keys and g)
bool launch_sweep_pm(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, int *status, const SweepWinner *winner = nullptr);
// the same for RightToLeft geometries whose Pout region is "target column outside the image", a multiple of 32 from 64 to 512
// disparities, up to four census words (svh_census_sweep_rl.hip); launch_sweep_pm tries it first unless the "census_sweep_rl" option is 0
bool launch_sweep_rl(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, int *status, const SweepWinner *winner);
// (svh_census_sweep_rl_wide.hip) the same specialisation for records of five to eight words, geometry already checked
bool launch_sweep_rl_wide(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, int *status, const SweepWinner &sw);

} // namespace svh

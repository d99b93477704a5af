// census_sweep on the matrix cores, the form the benchmark geometry takes: RightToLeft, the disparities that pay Pout are exactly
// the ones that look past the target image's right edge, disparity count known at compile time.
//
// Same arithmetic as svh_census_sweep_pm.hip (that file has the encoding: +-1 FP4 operands on v_mfma_scale_f32_32x32x64_f8f6f4, cell =
// 256 dot + register number + 16 tile, running maximum in the frame of the current tile) and the same persistent walk over items
// (an item = 32 CT WAVES source pixels of one image row, the blocks of an XCD on neighbouring rows).  What differs is everything
// around the MFMAs, because that -- not the matrix pipe -- was what the launch spent its time on: round 2's kernel issued 23 M wave
// instructions per 1080p x 256 launch, of which the tiles themselves (MFMA + fragment read + the nine-instruction maximum) are 7 M;
// a third of the stream was scalar bookkeeping and branches (profiles/r02b_sq_counters.json).  Here
//
//   * the disparity count is a template parameter: a column tile's D/32 + 1 row tiles are straight-line code (no loop counter, no
//     remainder loops, fragment addresses are immediates);
//   * with the geometry condition  Wt - disp_lower == Ws - d_offset  the Pout region of a pixel is the set of disparities whose
//     target column lies outside the image: those columns hold the all-zero census vector (cross_correlations.h:235), their cost is
//     |s| for every such disparity, and the reference's '<=' scan (correlation_base.h:441-455) leaves the LAST index of the range as
//     that region's winner.  So the second regional key is  (popcount(s), last index)  when the pixel has such disparities, and no
//     tile has to be computed, masked or decoded for them: a column tile near the right edge simply runs fewer row tiles (the one
//     that straddles the edge starts from a pattern that holds -2^22 in its rows past it);
//   * operands come through registers, not through an LDS staging area: buffer loads (range-checked by the hardware: no clamps,
//     nothing to compute per lane and item but one add) fetch the next item's compact words while the tiles of the current one run,
//     the expansion writes the FP4 records of the window; LDS holds the two window buffers and nothing else;
//   * the two lane halves merge with v_permlane32_swap instead of a ds_bpermute round trip.
//
// Geometries outside this (LeftToRight, a search range that does not end at the image edge, fewer than 33 or more than 512 disparities)
// keep svh_census_sweep_pm.hip; tests/test_gpu_sweep_engines.py holds the two kernels and the vector-ALU sweep to the same keys.
#include "svh_census_sweep_rl_impl.h"

namespace svh {

bool launch_sweep_rl(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, int *status, const SweepWinner *winner) {
    const SweepWinner sw = winner ? *winner : SweepWinner();
    if (!rl_geometry_ok(g, sw)) return false;
    switch (g.nWw) {
    case 1: return launch_rl_words<1>(ctx, g, Pout, keys, gmap, status, sw);
    case 2: return launch_rl_words<2>(ctx, g, Pout, keys, gmap, status, sw);
    case 3: return launch_rl_words<3>(ctx, g, Pout, keys, gmap, status, sw);
    case 4: return launch_rl_words<4>(ctx, g, Pout, keys, gmap, status, sw);
    case 5: case 6: case 7: case 8: return launch_sweep_rl_wide(ctx, g, Pout, keys, gmap, status, sw);
    default: return false;
    }
}

} // namespace svh

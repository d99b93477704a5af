// census_sweep on the matrix cores, RightToLeft specialisation (svh_census_sweep_rl_impl.h), for census records of five to eight words:
// 13x13 and 15x15 windows (census.h:80-108: (F - 1) / 32 + 1 words for F window samples), 9x9 windows on colour images.  Three or four
// MFMAs per tile, two column tiles per wave, 256-pixel items.  Until round 5 these ran the general kernel (13x13 at 1080p x 256: 0.27 ms).
#include "svh_census_sweep_rl_impl.h"

namespace svh {

bool launch_sweep_rl_wide(svh_context *ctx, const CensusGeom &g, float Pout, uint2 *keys, float *gmap, int *status, const SweepWinner &sw) {
    switch (g.nWw) {
    case 5: return launch_rl_words<5>(ctx, g, Pout, keys, gmap, status, sw);
    case 6: return launch_rl_words<6>(ctx, g, Pout, keys, gmap, status, sw);
    case 7: return launch_rl_words<7>(ctx, g, Pout, keys, gmap, status, sw);
    case 8: return launch_rl_words<8>(ctx, g, Pout, keys, gmap, status, sw);
    default: return false;
    }
}

} // namespace svh

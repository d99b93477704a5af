// computeGuidedCV, a wave per 64 pixels: windows 5 wide (svh_guided_wave_impl.h)
#include "svh_guided_wave_impl.h"

namespace svh {

bool launch_guided_wave_h2(svh_context *ctx, int cmp, bool zm, bool nrm, FeatImage src, FeatImage tgt, const float *ms, const float *ns, const float *mt,
                           const float *nt, int H, int Ws, int Wt, const GuideArgs &g, int32_t *disp, float *tcv) {
    return launch_guided_wave_hr<2>(ctx, cmp, zm, nrm, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv);
}

} // namespace svh

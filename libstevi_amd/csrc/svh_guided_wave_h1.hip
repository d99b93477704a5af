// computeGuidedCV, a wave per 64 pixels: windows 3 wide (svh_guided_wave_impl.h)
#include "svh_guided_wave_impl.h"

namespace svh {

bool launch_guided_wave_h1(svh_context *ctx, int cmp, bool zm, bool nrm, FeatImage src, FeatImage tgt, const float *ms, const float *ns, const float *mt,
                           const float *nt, int H, int Ws, int Wt, const GuideArgs &g, int32_t *disp, float *tcv) {
    return launch_guided_wave_hr<1>(ctx, cmp, zm, nrm, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv);
}

bool launch_guided_wave(svh_context *ctx, int cmp, bool zm, bool nrm, FeatImage src, FeatImage tgt, const float *ms, const float *ns, const float *mt,
                        const float *nt, int H, int Ws, int Wt, const GuideArgs &g, int32_t *disp, float *tcv) {
    if ((src.C != 1 && src.C != 3) || tgt.C != src.C || src.h_r != tgt.h_r || src.v_r != tgt.v_r || H > 65535) return false;
    switch (src.h_r) {
    case 1: return launch_guided_wave_h1(ctx, cmp, zm, nrm, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv);
    case 2: return launch_guided_wave_h2(ctx, cmp, zm, nrm, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv);
    case 3: return launch_guided_wave_h3(ctx, cmp, zm, nrm, src, tgt, ms, ns, mt, nt, H, Ws, Wt, g, disp, tcv);
    default: return false;
    }
}

} // namespace svh

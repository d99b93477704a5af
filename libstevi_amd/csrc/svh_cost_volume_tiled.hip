// Register-blocked float cost volumes of grey and colour images: the dispatch.  The kernels live in svh_cost_volume_tiled_impl.h and are instantiated
// by one translation unit per comparison (svh_cost_volume_tiled_{dot,dotzm,ssd,sad}.hip).
#include "svh_cost_volume_tiled_impl.h"

namespace svh {

bool cost_volume_colsum_applies(const svh_context *ctx, const CostVolumeArgs &a, ImageDesc src, ImageDesc tgt, int h_r, int v_r) {
    if (a.literal || src.C != tgt.C || src.C < 1 || src.C > 4 || h_r < 1 || h_r > 7 || func_census(a.func) || a.func == SVH_ZSAD || !ctx->cost_volume_colsum) return false;
    return colsum_shmem(v_r, a.D, func_zero_mean(a.func), func_normalized(a.func), a.n_dh, src.C) <= 78 * 1024 && (int64_t)a.H * a.Ws * a.D > 0;
}

// Returns SVH_OK when the tiled kernel ran, SVH_ERR_UNSUPPORTED (without touching the context error) when the caller must
// use the generic kernel (multi-channel images, windows wider than 11, tiles beyond the LDS budget).
int dev_cost_volume_grey_tiled(svh_context *ctx, Scratch &scr, const CostVolumeArgs &a_in, ImageDesc src, ImageDesc tgt, int h_r, int v_r, float *cv) {
    if (h_r < 1 || h_r > 7 || func_census(a_in.func)) return SVH_ERR_UNSUPPORTED;
    CostVolumeArgs a = a_in; // (+ the channel count: colour images take the column-sum kernel with the channels as rows, or nothing here)
    a.C = src.C;
    if (src.C != 1 || tgt.C != 1 || (h_r > 5 && a.func != SVH_ZSAD)) { // (colour images, windows 13 or 15 wide: the column-sum kernel or nothing here; ZSAD on grey images: the per-window kernel)
        if (!cost_volume_colsum_applies(ctx, a, src, tgt, h_r, v_r)) return SVH_ERR_UNSUPPORTED;
    }
    const size_t shmem = pxlane_shmem(h_r, v_r, a.D);
    const bool zm = func_zero_mean(a.func), nrm = func_normalized(a.func);
    if (a.n_dh > 1 && !cost_volume_colsum_applies(ctx, a, src, tgt, h_r, v_r)) return SVH_ERR_UNSUPPORTED;
    if (shmem > 60 * 1024 && (a.func == SVH_ZSAD || !ctx->cost_volume_colsum || colsum_shmem(v_r, a.D, zm, nrm, a.n_dh, a.C) > 78 * 1024)) return SVH_ERR_UNSUPPORTED;
    if ((int64_t)a.H * a.Ws * a.D == 0) return SVH_OK;
    const int cmp = (a.func == SVH_SSD || a.func == SVH_ZSSD) ? T_SSD : (a.func == SVH_SAD || a.func == SVH_ZSAD) ? T_SAD : T_DOT;
    const size_t ns_px = (size_t)a.H * a.Ws, nt_px = (size_t)a.H * a.Wt;
    WindowStatsCache local;
    WindowStatsCache &st = a.stats ? *a.stats : local;
    if (!st.ready) {
        Scratch &owner = st.scr ? *st.scr : scr;
        st.zc = owner.get_n<float>(ns_px);
        if (!st.zc) return SVH_ERR_OUT_OF_MEMORY;
        if (zm) {
            st.ms = owner.get_n<float>(ns_px);
            st.mt = owner.get_n<float>(nt_px);
            if (!st.ms || !st.mt) return SVH_ERR_OUT_OF_MEMORY;
        }
        if (nrm) {
            st.ns = owner.get_n<float>(ns_px);
            st.nt = owner.get_n<float>(nt_px);
            if (!st.ns || !st.nt) return SVH_ERR_OUT_OF_MEMORY;
        }
        // means / norms follow the reference literally (channelsMean, channelsNorm, channelsZeroMeanNorm); zc = cost of the source
        // window against the all-zero target vector
        const int cmpk = cmp == T_DOT ? CMP_DOT : cmp == T_SSD ? CMP_SSD : CMP_SAD;
        // (without zero-mean the terms are separable: column sums shared between windows, like the cost kernel that reads these maps)
        const bool sep = ctx->cost_volume_colsum;
        // FiniteCostsQuery: the costs of a normalised function are finite and of magnitude about one iff every norm is positive and finite
        int *odd = nullptr;
        if (a.finite_query && a.finite_query->asked && nrm) {
            odd = owner.get_n<int>(64);
            if (!odd) return SVH_ERR_OUT_OF_MEMORY;
            SVH_HIP_CHECK(ctx, hipMemsetAsync(odd, 0, sizeof(int), ctx->stream));
        }
        SVH_TRY(launch_window_stats(ctx, FeatImage{src.data, src.H, src.W, src.C, h_r, v_r}, zm, nrm, st.ms, st.ns, st.zc, cmpk, sep, odd));
        if (zm || nrm) SVH_TRY(launch_window_stats(ctx, FeatImage{tgt.data, tgt.H, tgt.W, tgt.C, h_r, v_r}, zm, nrm, st.mt, st.nt, nullptr, cmpk, sep, odd));
        st.ready = true;
        if (odd) { // one word back, before the cost kernel is launched: the wait is for the two statistics kernels
            int h = 1;
            SVH_HIP_CHECK(ctx, hipMemcpyAsync(&h, odd, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            SVH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            a.finite_query->known = true;
            a.finite_query->all_finite = h == 0;
        }
    }
    float *ms = st.ms, *mt = st.mt, *ns = st.ns, *nt = st.nt, *zc = st.zc;
    const int sign = a.sign();
    bool ok;
    if (cmp == T_DOT) ok = zm ? cost_tiled_dot_zm(ctx, h_r, a, src.data, tgt.data, v_r, sign, ms, mt, ns, nt, zc, cv, shmem)
                              : cost_tiled_dot(ctx, h_r, a, src.data, tgt.data, v_r, sign, ms, mt, ns, nt, zc, cv, shmem);
    else if (cmp == T_SSD) ok = zm ? cost_tiled_ssd_zm(ctx, h_r, a, src.data, tgt.data, v_r, sign, ms, mt, ns, nt, zc, cv, shmem)
                                   : cost_tiled_ssd(ctx, h_r, a, src.data, tgt.data, v_r, sign, ms, mt, ns, nt, zc, cv, shmem);
    else ok = zm ? cost_tiled_sad_zm(ctx, h_r, a, src.data, tgt.data, v_r, sign, ms, mt, ns, nt, zc, cv, shmem)
                 : cost_tiled_sad(ctx, h_r, a, src.data, tgt.data, v_r, sign, ms, mt, ns, nt, zc, cv, shmem);
    if (!ok) return SVH_ERR_UNSUPPORTED;
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

} // namespace svh

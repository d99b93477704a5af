// Fused stereo pipeline: the reference benchmark's call chain kept on the device.
//   unfoldBasedCostVolume -> [sgmCostVolume] -> extractSelectedIndex -> selectedIndexToDisp
//   (test/benchmarks/benchmarkCrossCorrelationAlgorithms.cpp:92-96, :288-294)
//   -> [truncatedCostVolume -> refineDispCostInterpolation]   (examples/stereo_refine_test/main.cpp:367-384)
// Every stage is the same kernel the stand-alone entry points use; what changes is what touches HBM:
//   * census / Hamming: the cost volume is a pure function of two 8-byte-per-pixel word maps, so the SGM line
//     kernels and the apply/winner kernel evaluate it on the fly and neither C nor S is written unless the
//     caller asks for them;
//   * float costs: C is materialised once (scratch or the caller's array); the Cost branch still writes S at
//     most once, the Score branch sweeps S per pass.
#include "svh_internal.h"

using namespace svh;

extern "C" int svh_stereo_match(svh_context *ctx, const svh_stereo_params *prm, const svh_array *img_l, const svh_array *img_r,
                                svh_array *disp, svh_array *refined, svh_array *cv, svh_array *sgm_cv, svh_array *keys) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    if (!prm) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "null parameters");
    SVH_TRY(validate_image(ctx, img_l, "img_l", prm->match_func));
    SVH_TRY(validate_image(ctx, img_r, "img_r", prm->match_func));
    const int func = prm->match_func;
    if (!func_supported(func)) return fail(ctx, SVH_ERR_UNSUPPORTED, "matching function %d", func);
    if (prm->disp_direction != SVH_LEFT_TO_RIGHT && prm->disp_direction != SVH_RIGHT_TO_LEFT)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad disparity direction");
    if (prm->h_radius < 0 || prm->v_radius < 0 || prm->h_radius > 255 || prm->v_radius > 255)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "radii must be in [0,255]");
    if (prm->disp_count <= 0) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "disp_count must be positive");
    if (prm->sgm_directions != 0 && prm->sgm_directions != 4 && prm->sgm_directions != 8)
        return fail(ctx, prm->sgm_directions == 16 ? SVH_ERR_UNSUPPORTED : SVH_ERR_INVALID_ARGUMENT, "sgm_directions must be 0, 4 or 8");
    if (img_l->ndim != img_r->ndim) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "image ranks differ");
    if (img_l->shape[0] != img_r->shape[0]) return fail(ctx, SVH_EMPTY_RESULT, "row counts differ");
    const int C = img_l->ndim == 3 ? (int)img_l->shape[2] : 1;
    if (img_l->ndim == 3 && img_l->shape[2] != img_r->shape[2]) return fail(ctx, SVH_EMPTY_RESULT, "channel counts differ");
    const int F = (2 * prm->h_radius + 1) * (2 * prm->v_radius + 1) * C;
    if (func_census(func) && F <= 1) return fail(ctx, SVH_EMPTY_RESULT, "census needs at least two feature channels");
    const bool want_refine = prm->refine_kernel >= 0;
    if (want_refine && prm->refine_kernel > SVH_GAUSSIAN) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad interpolation kernel");
    if (want_refine && !refined) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "refinement requested without an output array");
    if (!disp && !refined && !keys && !cv && !sgm_cv) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "no output requested");

    const bool r2l = prm->disp_direction == SVH_RIGHT_TO_LEFT;
    const svh_array *src = r2l ? img_r : img_l, *tgt = r2l ? img_l : img_r;
    const int H = (int)src->shape[0], Ws = (int)src->shape[1], Wt = (int)tgt->shape[1];
    const int Dtot = prm->disp_count;
    const int sb = prm->shard_count > 0 ? prm->shard_begin : 0;
    const int D = prm->shard_count > 0 ? prm->shard_count : Dtot;
    if (sb < 0 || sb + D > Dtot) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "disparity shard outside [0, disp_count)");
    const bool sharded = D != Dtot;
    const int strategy = func_strategy(func);
    const bool sgm = prm->sgm_directions != 0;
    if (sharded && sgm)
        return fail(ctx, SVH_ERR_UNSUPPORTED, "SGM over a disparity shard needs the per-pixel minimum of every shard (see DESIGN.md, multi-GPU)");
    if (sharded && want_refine) return fail(ctx, SVH_ERR_UNSUPPORTED, "refinement over a disparity shard");
    for (int k = 0; k < 4; k++)
        if (prm->margins[k] < 0) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "margins must be non-negative");

    if (disp) {
        SVH_TRY(validate(ctx, disp, "disp", SVH_I32, 2, 2));
        if (disp->shape[0] != H || disp->shape[1] != Ws) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "disp must have shape (%d,%d)", H, Ws);
    }
    if (refined) {
        SVH_TRY(validate(ctx, refined, "refined", SVH_F32, 2, 2));
        if (refined->shape[0] != H || refined->shape[1] != Ws) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "refined must have shape (%d,%d)", H, Ws);
        if (!want_refine) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "refined output given but refine_kernel < 0");
    }
    if (keys) {
        SVH_TRY(validate(ctx, keys, "keys", SVH_U64, 2, 2));
        if (keys->shape[0] != H || keys->shape[1] != Ws) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "keys must have shape (%d,%d)", H, Ws);
    }
    for (svh_array *vol : {cv, sgm_cv}) {
        if (!vol) continue;
        SVH_TRY(validate(ctx, vol, "volume output", SVH_F32, 3, 3));
        if (vol->shape[0] != H || vol->shape[1] != Ws || vol->shape[2] != D)
            return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "volume outputs must have shape (%d,%d,%d)", H, Ws, D);
    }
    if (sgm_cv && !sgm) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "sgm_cv output given but sgm_directions == 0");

    Scratch scr(ctx);
    void *dsrc, *dtgt;
    SVH_TRY(stage_image(ctx, scr, *src, &dsrc));
    SVH_TRY(stage_image(ctx, scr, *tgt, &dtgt));
    OutStage o_disp, o_ref, o_cv, o_sgm, o_keys;
    if (disp) SVH_TRY(stage_out(ctx, scr, *disp, &o_disp));
    if (refined) SVH_TRY(stage_out(ctx, scr, *refined, &o_ref));
    if (cv) SVH_TRY(stage_out(ctx, scr, *cv, &o_cv));
    if (sgm_cv) SVH_TRY(stage_out(ctx, scr, *sgm_cv, &o_sgm));
    if (keys) SVH_TRY(stage_out(ctx, scr, *keys, &o_keys));

    const int64_t npx = (int64_t)H * Ws;
    const size_t nvox = (size_t)npx * D;
    // the Cost-branch winner kernels write the disparity map themselves; the index map is only materialised for the
    // stages that read it back (refinement, the generic extract -> index_to_disp chain)
    const bool cost_fused = func_census(func) || (sgm && strategy == SVH_COST);
    const bool need_idx = want_refine || (disp && !cost_fused);
    int32_t *d_idx = need_idx ? scr.get_n<int32_t>((size_t)npx) : nullptr;
    if (need_idx && !d_idx) return SVH_ERR_OUT_OF_MEMORY;
    const int disp_sign = r2l ? 1 : -1;
    const int disp_offset = disp_sign * (prm->disp_lower + sb); // selectedIndexToDisp(idx, first searched offset); shards report global indices
    bool disp_written = false, refined_written = false; // (from the winner records of the Score branch's fused finish)
    WinnerOut win;
    win.idx = d_idx;
    win.disp = (disp && cost_fused) ? (int32_t *)o_disp.dptr : nullptr;
    win.disp_sign = disp_sign;
    win.disp_offset = disp_offset;
    float *d_taps = want_refine ? scr.get_n<float>((size_t)npx * 3) : nullptr;
    if (want_refine && !d_taps) return SVH_ERR_OUT_OF_MEMORY;
    unsigned long long *d_keys = keys ? (unsigned long long *)o_keys.dptr : nullptr;
    win.taps = d_taps;
    win.taps_h_r = prm->refine_h_radius;
    win.taps_v_r = prm->refine_v_radius;
    win.taps_up_to_shift = want_refine && (prm->refine_kernel == SVH_PARABOLA || prm->refine_kernel == SVH_EQUIANGULAR);
    win.keys = d_keys;
    win.key_offset = sb;
    win.key_total = Dtot;

    const CostVolumeArgs cva{func, prm->disp_direction, H, Ws, Wt, prm->disp_lower + sb, D};
    const ImageDesc isrc{(const float *)dsrc, H, Ws, C}, itgt{(const float *)dtgt, H, Wt, C};
    SgmArgs sa{prm->sgm_directions, strategy, H, Ws, D, prm->P1, prm->P2, prm->Pout, prm->margins[0], prm->margins[1], prm->margins[2],
               prm->margins[3]};

    if (func_census(func)) {
        const int nWw = census_words_written(F);
        uint32_t *sw = scr.get_n<uint32_t>((size_t)H * Ws * (nWw ? nWw : 1));
        uint32_t *tw = scr.get_n<uint32_t>((size_t)H * Wt * (nWw ? nWw : 1));
        if (!sw || !tw) return SVH_ERR_OUT_OF_MEMORY;
        SVH_TRY(dev_census_pair_compact(ctx, isrc, itgt, prm->h_radius, prm->v_radius, nWw, sw, tw));
        if (cv) SVH_TRY(dev_hamming_volume(ctx, cva, sw, tw, nWw, (float *)o_cv.dptr));
        if (win.any() || sgm_cv) {
            // Cost branch with the Hamming cost evaluated on the fly; sgm_directions == 0 degenerates to S = C
            CostSource cs;
            cs.src_words = sw;
            cs.tgt_words = tw;
            cs.nWw = nWw;
            cs.Wt = Wt;
            cs.sign = r2l ? 1 : -1;
            cs.disp_lower = cva.disp_lower;
            SVH_TRY(dev_sgm_cost_branch(ctx, scr, sa, cs, sgm_cv ? (float *)o_sgm.dptr : nullptr, win));
        }
    } else {
        // What the cost kernel can do while it holds a pixel's costs (CostReduce; the column-sum kernel of grey images):
        //   no SGM, disparity map only: the winner -- the volume is then never written, and nothing reads it back;
        //   Cost-branch SGM: the two regional minima the line recurrences run on -- no probing read of the volume.
        CostReduce red;
        CostVolumeArgs cva_r = cva;
        const bool colsum = ctx->cost_reduce_fused && cost_volume_colsum_applies(ctx, cva, isrc, itgt, prm->h_radius, prm->v_radius);
        const bool winner_in_cost = colsum && !sgm && !cv && !want_refine && !keys && disp && !sharded; // (grey and colour images)
        const bool minima_in_cost = colsum && isrc.C == 1 && sgm && strategy == SVH_COST && ctx->sgm_cost_two_minima && !sharded; // (grey images)
        if (winner_in_cost) {
            red.mode = 1;
            red.score = strategy != SVH_COST;
            red.disp = (int32_t *)o_disp.dptr;
            red.disp_sign = disp_sign;
            red.disp_offset = disp_offset;
            red.store = false;
            cva_r.reduce = &red;
            SVH_TRY(dev_cost_volume_from_images(ctx, scr, cva_r, isrc, itgt, prm->h_radius, prm->v_radius, nullptr));
            if (!red.done) return fail(ctx, SVH_ERR_HIP, "internal: the cost kernel that ran does not reduce");
            SVH_TRY(finish_out(ctx, o_disp));
            return SVH_OK;
        }
        // Score-branch SGM on a disparity count that is no multiple of 64 aggregates rows padded to whole lanes (dev_sgm_score_branch): when
        // nobody wants the cost volume itself the cost kernel writes that layout directly (its output pitch) and only the pads are filled
        const int DP = (D + 63) / 64 * 64;
        const bool padded_cv = sgm && strategy != SVH_COST && !cv && ctx->sgm_score_pad && D > 64 && D < 512 && DP != D;
        float *d_cv = cv ? (float *)o_cv.dptr : scr.get_n<float>(padded_cv ? (size_t)npx * DP : nvox);
        if (!d_cv) return SVH_ERR_OUT_OF_MEMORY;
        if (padded_cv) cva_r.out_px_stride = DP;
        float *d_s = nullptr;
        if (sgm && strategy != SVH_COST) {
            d_s = sgm_cv ? (float *)o_sgm.dptr : scr.get_n<float>(nvox);
            if (!d_s) return SVH_ERR_OUT_OF_MEMORY;
        }
        // Score-branch SGM behind a normalised function: ask the statistics kernels whether every norm is a positive finite number (then
        // every cost is finite and the aggregation's finiteness filters are no-ops: SgmArgs::costs_all_finite).  One word and a wait for the
        // statistics kernels: only for volumes where the aggregation is long enough to notice.
        FiniteCostsQuery fq;
        if (sgm && strategy != SVH_COST && func_normalized(func) && ctx->sgm_score_finish_fused && nvox >= ((size_t)1 << 26)) {
            fq.asked = true;
            cva_r.finite_query = &fq;
        }
        if (minima_in_cost) {
            red.mode = 2;
            red.minima = scr.get_n<float>((size_t)npx * 2);
            red.flag = scr.get_n<int>(64);
            if (!red.minima || !red.flag) return SVH_ERR_OUT_OF_MEMORY;
            // bit 0: no exact-integer route (the kernel does not test integrality); it raises bit 1 itself.  (A device-side fill: an
            // asynchronous copy from a stack variable may be read after the variable is gone.)
            SVH_HIP_CHECK(ctx, hipMemsetD32Async((hipDeviceptr_t)red.flag, 1, 1, ctx->stream));
            cva_r.reduce = &red;
        }
        SVH_TRY(dev_cost_volume_from_images(ctx, scr, cva_r, isrc, itgt, prm->h_radius, prm->v_radius, d_cv));
        if (sgm && strategy == SVH_COST) {
            CostSource cs;
            cs.cv = d_cv;
            if (minima_in_cost && red.done) {
                cs.float_minima = red.minima;
                cs.float_flag = red.flag;
            }
            SVH_TRY(dev_sgm_cost_branch(ctx, scr, sa, cs, sgm_cv ? (float *)o_sgm.dptr : nullptr, win));
        } else {
            const float *d_final = d_cv;
            bool winner_done = false;
            if (sgm) {
                // the winner (index, disparity, refinement taps) rides on the launches that write a pixel's final aggregated costs where the
                // Score branch can do that; the volume itself is only stored in full when the caller asked for it
                ScoreFinish fin;
                const bool try_fused = ctx->sgm_score_finish_fused && (need_idx || disp) && !keys;
                if (try_fused) {
                    fin.records = scr.get_n<float>((size_t)npx * 4);
                    if (!fin.records) return SVH_ERR_OUT_OF_MEMORY;
                }
                fin.taps_h_r = prm->refine_h_radius;
                fin.taps_v_r = prm->refine_v_radius;
                fin.store_all = sgm_cv != nullptr;
                sa.costs_all_finite = fq.known && fq.all_finite;
                if (padded_cv) {
                    SVH_TRY(dev_sgm_fill_pads(ctx, d_cv, npx, D, DP));
                    sa.cv_pitch = DP;
                }
                SVH_TRY(dev_sgm_score_branch(ctx, scr, sa, d_cv, d_s, false, try_fused ? &fin : nullptr));
                winner_done = try_fused && fin.done;
                if (winner_done) {
                    SVH_TRY(dev_finish_records(ctx, fin.records, npx, want_refine ? prm->refine_kernel : -1, disp_sign, disp_offset, nullptr,
                                               disp ? (int32_t *)o_disp.dptr : nullptr, want_refine ? (float *)o_ref.dptr : nullptr));
                    disp_written = disp != nullptr;
                    refined_written = want_refine;
                }
                d_final = d_s;
            }
            if (!winner_done) {
                if (need_idx || keys) SVH_TRY(dev_extract_index(ctx, strategy, d_final, npx, D, d_idx, d_keys, sb, Dtot));
                if (want_refine)
                    SVH_TRY(dev_truncated_cv(ctx, SVH_TCV_SAME, prm->disp_direction, d_final, d_idx, H, Ws, D, prm->refine_h_radius,
                                             prm->refine_v_radius, 1, d_taps));
            }
        }
    }
    if (want_refine && !refined_written) SVH_TRY(dev_refine(ctx, prm->refine_kernel, d_taps, d_idx, npx, 3, (float *)o_ref.dptr));
    if (disp && !cost_fused && !disp_written) SVH_TRY(dev_index_to_disp(ctx, prm->disp_direction, d_idx, npx, disp_offset, (int32_t *)o_disp.dptr));

    if (disp) SVH_TRY(finish_out(ctx, o_disp));
    if (refined) SVH_TRY(finish_out(ctx, o_ref));
    if (cv) SVH_TRY(finish_out(ctx, o_cv));
    if (sgm_cv) SVH_TRY(finish_out(ctx, o_sgm));
    if (keys) SVH_TRY(finish_out(ctx, o_keys));
    return SVH_OK;
}

// ---- disparity-sharded census (+ SGM) across GPUs ---------------------------------------------------------------
// Each rank owns the disparity indices [shard_begin, shard_begin + shard_count) of the D-wide range and calls
//   1. svh_census_shard_keys   -> (H, W, 2) int32 regional winner keys of its shard (global indices inside)
//   2. an int32 MIN all-reduce of the keys over the ranks (RCCL; the only exchange of the whole pipeline)
//   3. svh_census_shard_finish -> the reduced keys give g = min_d [2c (+Pout)] over ALL disparities, hence the same
//      min_p maps on every rank (line scans), and the finalize kernel emits the disparity map (replicated).
// Valid in the integer-exact regime only (census / Hamming costs, integer Pout): there the Cost branch couples the
// disparities through per-pixel scalars alone, which is what makes the disparity axis shardable.
namespace {

struct ShardSetup {
    const svh_array *src, *tgt;
    int H, Ws, Wt, C, F, nWw, Dtot, sb, D, sign;
};

// `per_shard_kernels`: the call runs the voxel sweep over [shard_begin, shard_begin + shard_count) and so needs the lane kernels
// for that many disparities (svh_census_shard_keys).  svh_census_shard_finish only touches per-pixel maps of the WHOLE range: its
// limit is the 12 index bits of the keys (census_max_total_disparities), not the per-launch one.
int shard_setup(svh_context *ctx, const svh_stereo_params *prm, const svh_array *img_l, const svh_array *img_r, bool per_shard_kernels,
                ShardSetup *s) {
    if (!prm) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "null parameters");
    SVH_TRY(validate_image(ctx, img_l, "img_l", prm->match_func));
    SVH_TRY(validate_image(ctx, img_r, "img_r", prm->match_func));
    if (!func_census(prm->match_func)) return fail(ctx, SVH_ERR_UNSUPPORTED, "disparity sharding with SGM needs census / Hamming costs");
    if (prm->disp_direction != SVH_LEFT_TO_RIGHT && prm->disp_direction != SVH_RIGHT_TO_LEFT)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad disparity direction");
    if (prm->h_radius < 0 || prm->v_radius < 0 || prm->h_radius > 255 || prm->v_radius > 255)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "radii must be in [0,255]");
    if (prm->sgm_directions != 0 && prm->sgm_directions != 4 && prm->sgm_directions != 8)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "sgm_directions must be 0, 4 or 8");
    if (img_l->ndim != img_r->ndim) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "image ranks differ");
    if (img_l->shape[0] != img_r->shape[0]) return fail(ctx, SVH_EMPTY_RESULT, "row counts differ");
    s->C = img_l->ndim == 3 ? (int)img_l->shape[2] : 1;
    if (img_l->ndim == 3 && img_l->shape[2] != img_r->shape[2]) return fail(ctx, SVH_EMPTY_RESULT, "channel counts differ");
    s->F = (2 * prm->h_radius + 1) * (2 * prm->v_radius + 1) * s->C;
    if (s->F <= 1) return fail(ctx, SVH_EMPTY_RESULT, "census needs at least two feature channels");
    s->nWw = census_words_written(s->F);
    const bool r2l = prm->disp_direction == SVH_RIGHT_TO_LEFT;
    s->sign = r2l ? 1 : -1;
    s->src = r2l ? img_r : img_l;
    s->tgt = r2l ? img_l : img_r;
    s->H = (int)s->src->shape[0];
    s->Ws = (int)s->src->shape[1];
    s->Wt = (int)s->tgt->shape[1];
    s->Dtot = prm->disp_count;
    s->sb = prm->shard_count > 0 ? prm->shard_begin : 0;
    s->D = prm->shard_count > 0 ? prm->shard_count : s->Dtot;
    if (s->Dtot <= 0 || s->sb < 0 || s->sb + s->D > s->Dtot) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "disparity shard outside [0, disp_count)");
    if (s->Dtot > census_max_total_disparities()) return fail(ctx, SVH_ERR_UNSUPPORTED, "at most %d disparities over all shards", census_max_total_disparities());
    for (int k = 0; k < 4; k++)
        if (prm->margins[k] < 0) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "margins must be non-negative");
    SgmArgs sa{prm->sgm_directions, SVH_COST, s->H, s->Ws, s->D, prm->P1, prm->P2, prm->Pout, prm->margins[0], prm->margins[1], prm->margins[2],
               prm->margins[3]};
    if (!census_lane_kernels_available(s->nWw, per_shard_kernels ? s->D : 1) || !census_exact_regime(sa, s->nWw))
        return fail(ctx, SVH_ERR_UNSUPPORTED,
                    "disparity sharding needs the integer-exact regime (integer Pout, at most 8 census words = windows up to 15x15, <= 1024 disparities per shard)");
    return SVH_OK;
}

} // namespace

// Row bands: in the exact regime the winner of a pixel depends on its own costs and on its position only (svh_census_keys.h,
// SweepWinner), so rows are independent once the census words of their window rows exist: v_radius halo rows on either side.
extern "C" int svh_census_band_match(svh_context *ctx, const svh_stereo_params *prm, const svh_array *img_l, const svh_array *img_r,
                                     int32_t row_begin, int32_t row_count, svh_array *disp_band) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    ShardSetup s;
    SVH_TRY(shard_setup(ctx, prm, img_l, img_r, true, &s));
    if (prm->shard_count > 0) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "row bands take the whole disparity range");
    if (prm->refine_kernel >= 0) return fail(ctx, SVH_ERR_UNSUPPORTED, "row bands produce the disparity map only");
    if (!ctx->census_fast_path) return fail(ctx, SVH_ERR_UNSUPPORTED, "row bands need the census fast path");
    if (row_begin < 0 || row_count < 0 || row_begin + (int64_t)row_count > s.H) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "row band outside the image");
    SVH_TRY(validate(ctx, disp_band, "disp_band", SVH_I32, 2, 2));
    if (disp_band->shape[0] != row_count || disp_band->shape[1] != s.Ws)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "disp_band must have shape (%d,%d)", row_count, s.Ws);
    if (row_count == 0 || s.Ws == 0) return SVH_EMPTY_RESULT;
    Scratch scr(ctx);
    void *dsrc, *dtgt;
    OutStage od;
    SVH_TRY(stage_image(ctx, scr, *s.src, &dsrc));
    SVH_TRY(stage_image(ctx, scr, *s.tgt, &dtgt));
    SVH_TRY(stage_out(ctx, scr, *disp_band, &od));
    // the rows whose census words the band's windows read; the words of the halo rows themselves are wrong where the halo is cut
    // from the middle of the image (their windows would need rows further out) and are never used
    const int a = std::max(0, row_begin - prm->v_radius), b = std::min(s.H, row_begin + row_count + prm->v_radius), Hs = b - a;
    uint32_t *sw = scr.get_n<uint32_t>((size_t)Hs * s.Ws * (s.nWw ? s.nWw : 1));
    uint32_t *tw = scr.get_n<uint32_t>((size_t)Hs * s.Wt * (s.nWw ? s.nWw : 1));
    if (!sw || !tw) return SVH_ERR_OUT_OF_MEMORY;
    const float *bsrc = (const float *)dsrc + (size_t)a * s.Ws * s.C, *btgt = (const float *)dtgt + (size_t)a * s.Wt * s.C;
    SVH_TRY(dev_census_pair_compact(ctx, {bsrc, Hs, s.Ws, s.C}, {btgt, Hs, s.Wt, s.C}, prm->h_radius, prm->v_radius, s.nWw, sw, tw));
    SgmArgs sa{prm->sgm_directions, SVH_COST, Hs, s.Ws, s.D, prm->P1, prm->P2, prm->Pout, prm->margins[0], prm->margins[1], prm->margins[2],
               prm->margins[3]};
    sa.row_origin = a;
    sa.full_H = s.H;
    sa.store_row0 = row_begin - a;
    sa.store_rows = row_count;
    CostSource cs;
    cs.src_words = sw;
    cs.tgt_words = tw;
    cs.nWw = s.nWw;
    cs.Wt = s.Wt;
    cs.sign = s.sign;
    cs.disp_lower = prm->disp_lower;
    WinnerOut win;
    win.disp = (int32_t *)od.dptr;
    win.disp_sign = s.sign;
    win.disp_offset = s.sign * prm->disp_lower;
    SVH_TRY(dev_census_winner(ctx, scr, sa, cs, win));
    return finish_out(ctx, od);
}

extern "C" int svh_census_shard_region1_is_global(const svh_stereo_params *prm, const svh_array *img_l, const svh_array *img_r) {
    if (!prm || !img_l || !img_r || img_l->ndim < 2 || img_r->ndim < 2) return 0;
    if (prm->disp_direction != SVH_RIGHT_TO_LEFT) return 0;
    const int64_t Ws = img_r->shape[1], Wt = img_l->shape[1]; // RightToLeft: source = right image, target = left image
    return Ws + prm->disp_lower >= Wt ? 1 : 0;
}

extern "C" int svh_census_shard_keys(svh_context *ctx, const svh_stereo_params *prm, const svh_array *img_l, const svh_array *img_r,
                                     svh_array *keys) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    ShardSetup s;
    SVH_TRY(shard_setup(ctx, prm, img_l, img_r, true, &s));
    SVH_TRY(validate(ctx, keys, "keys", SVH_I32, 3, 3));
    if (keys->shape[0] != s.H || keys->shape[1] != s.Ws || keys->shape[2] != 2)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "keys must have shape (%d,%d,2)", s.H, s.Ws);
    Scratch scr(ctx);
    void *dsrc, *dtgt;
    OutStage ok;
    SVH_TRY(stage_image(ctx, scr, *s.src, &dsrc));
    SVH_TRY(stage_image(ctx, scr, *s.tgt, &dtgt));
    SVH_TRY(stage_out(ctx, scr, *keys, &ok));
    uint32_t *sw = scr.get_n<uint32_t>((size_t)s.H * s.Ws * (s.nWw ? s.nWw : 1));
    uint32_t *tw = scr.get_n<uint32_t>((size_t)s.H * s.Wt * (s.nWw ? s.nWw : 1));
    if (!sw || !tw) return SVH_ERR_OUT_OF_MEMORY;
    SVH_TRY(dev_census_pair_compact(ctx, {(const float *)dsrc, s.H, s.Ws, s.C}, {(const float *)dtgt, s.H, s.Wt, s.C}, prm->h_radius, prm->v_radius,
                                    s.nWw, sw, tw)); // both images in one launch
    SgmArgs sa{prm->sgm_directions, SVH_COST, s.H, s.Ws, s.D, prm->P1, prm->P2, prm->Pout, prm->margins[0], prm->margins[1], prm->margins[2],
               prm->margins[3]};
    CostSource cs;
    cs.src_words = sw;
    cs.tgt_words = tw;
    cs.nWw = s.nWw;
    cs.Wt = s.Wt;
    cs.sign = s.sign;
    cs.disp_lower = prm->disp_lower + s.sb;
    cs.d_offset = s.sb;
    // Disparity index d pays Pout when j + d >= Ws, and looks at target column j + lower + d: with Ws + lower >= Wt every paying
    // disparity of the WHOLE range sees the zero vector, the Pout region's winner is (|s|, last index) on every shard, and the
    // second key plane is global as written (libstevi_amd/sharded.py then exchanges the first plane only).
    if (svh_census_shard_region1_is_global(prm, img_l, img_r) == 1) cs.region1_global_last = s.Dtot - 1;
    SVH_TRY(dev_census_sweep(ctx, sa, cs, (uint2 *)ok.dptr, nullptr));
    return finish_out(ctx, ok);
}

extern "C" int svh_census_shard_finish(svh_context *ctx, const svh_stereo_params *prm, const svh_array *img_l, const svh_array *img_r,
                                       const svh_array *keys, svh_array *disp, svh_array *refined) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    ShardSetup s;
    SVH_TRY(shard_setup(ctx, prm, img_l, img_r, false, &s));
    SVH_TRY(validate(ctx, keys, "keys", SVH_I32, 3, 3));
    if (keys->shape[0] != s.H || keys->shape[1] != s.Ws || keys->shape[2] != 2)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "keys must have shape (%d,%d,2)", s.H, s.Ws);
    const bool want_refine = prm->refine_kernel >= 0;
    if (want_refine && (prm->refine_kernel > SVH_GAUSSIAN || !refined)) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad refinement request");
    if (!disp && !refined) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "no output requested");
    if (disp) {
        SVH_TRY(validate(ctx, disp, "disp", SVH_I32, 2, 2));
        if (disp->shape[0] != s.H || disp->shape[1] != s.Ws) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "disp must have shape (%d,%d)", s.H, s.Ws);
    }
    if (refined) {
        SVH_TRY(validate(ctx, refined, "refined", SVH_F32, 2, 2));
        if (refined->shape[0] != s.H || refined->shape[1] != s.Ws) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "refined must have shape (%d,%d)", s.H, s.Ws);
        if (!want_refine) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "refined output given but refine_kernel < 0");
    }
    Scratch scr(ctx);
    void *dkeys;
    SVH_TRY(stage_in(ctx, scr, *keys, &dkeys));
    OutStage o_disp, o_ref;
    if (disp) SVH_TRY(stage_out(ctx, scr, *disp, &o_disp));
    if (refined) SVH_TRY(stage_out(ctx, scr, *refined, &o_ref));
    const int64_t npx = (int64_t)s.H * s.Ws;
    // from here on everything is about the whole disparity range
    SgmArgs sa{prm->sgm_directions, SVH_COST, s.H, s.Ws, s.Dtot, prm->P1, prm->P2, prm->Pout, prm->margins[0], prm->margins[1], prm->margins[2],
               prm->margins[3]};
    // the disparity map alone does not depend on the min_p maps (census_finalize_kernel): no scans then
    const bool shift_ok = want_refine && (prm->refine_kernel == SVH_PARABOLA || prm->refine_kernel == SVH_EQUIANGULAR);
    const bool winner_only = ctx->census_winner_shortcut && (!want_refine || shift_ok);
    float *mmap = nullptr, *gmap = nullptr;
    // (the recurrences run) tile-edge values + replay where the geometry allows it, as in the one-GPU call; else the six min_p maps
    const bool tiles = !winner_only && census_tiles_apply(ctx, sa);
    if (!winner_only) {
        gmap = scr.get_n<float>((size_t)npx);
        if (!gmap) return SVH_ERR_OUT_OF_MEMORY;
        if (!tiles) {
            mmap = scr.get_n<float>((size_t)MIN_P_PLANES * npx);
            if (!mmap) return SVH_ERR_OUT_OF_MEMORY;
            SVH_TRY(dev_census_scans(ctx, sa, (const uint2 *)dkeys, gmap, false, mmap, nullptr));
        }
    }
    CostSource cs;
    cs.nWw = s.nWw;
    cs.Wt = s.Wt;
    cs.sign = s.sign;
    cs.disp_lower = prm->disp_lower;
    WinnerOut win;
    win.disp = disp ? (int32_t *)o_disp.dptr : nullptr;
    win.disp_sign = s.sign;
    win.disp_offset = s.sign * prm->disp_lower;
    int32_t *d_idx = nullptr;
    float *d_taps = nullptr;
    if (want_refine) { // the three taps are re-evaluated from the census words
        void *dsrc, *dtgt;
        SVH_TRY(stage_image(ctx, scr, *s.src, &dsrc));
        SVH_TRY(stage_image(ctx, scr, *s.tgt, &dtgt));
        uint32_t *sw = scr.get_n<uint32_t>((size_t)s.H * s.Ws * (s.nWw ? s.nWw : 1));
        uint32_t *tw = scr.get_n<uint32_t>((size_t)s.H * s.Wt * (s.nWw ? s.nWw : 1));
        d_idx = scr.get_n<int32_t>((size_t)npx);
        d_taps = scr.get_n<float>((size_t)npx * 3);
        if (!sw || !tw || !d_idx || !d_taps) return SVH_ERR_OUT_OF_MEMORY;
        SVH_TRY(dev_census_from_image(ctx, {(const float *)dsrc, s.H, s.Ws, s.C}, prm->h_radius, prm->v_radius, prm->h_radius, prm->v_radius, s.H,
                                      s.Ws, s.nWw, false, sw));
        SVH_TRY(dev_census_from_image(ctx, {(const float *)dtgt, s.H, s.Wt, s.C}, prm->h_radius, prm->v_radius, prm->h_radius, prm->v_radius, s.H,
                                      s.Wt, s.nWw, true, tw));
        cs.src_words = sw;
        cs.tgt_words = tw;
        win.idx = d_idx;
        win.taps = d_taps;
        win.taps_h_r = prm->refine_h_radius;
        win.taps_v_r = prm->refine_v_radius;
        win.taps_up_to_shift = shift_ok;
    }
    if (tiles) SVH_TRY(dev_census_tiles_from_keys(ctx, scr, sa, cs, (const uint2 *)dkeys, gmap, false, win));
    else SVH_TRY(dev_census_finalize(ctx, sa, cs, mmap, (const uint2 *)dkeys, win));
    if (want_refine) SVH_TRY(dev_refine(ctx, prm->refine_kernel, d_taps, d_idx, npx, 3, (float *)o_ref.dptr));
    if (disp) SVH_TRY(finish_out(ctx, o_disp));
    if (refined) SVH_TRY(finish_out(ctx, o_ref));
    return SVH_OK;
}

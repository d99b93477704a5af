// Per-pixel feature statistics and the feature-volume transforms of getFeatureVolumeForMatchFunc as stand-alone entry points
// (SURVEY.md section 8a rows A7 / A8).  The cost-volume kernels evaluate the same arithmetic on the fly and never write these
// volumes; callers that use the functions directly (examples/stereo_refine_test/main.cpp:386-398) get them here.
//
//   channelsMean                      correlation/correlation_base.h:1100-1136   sequential sum, then * float(1. / f)
//   channelsNorm                      correlation/cross_correlations.h:149-191    sqrtf(sequential sum of squares)
//   channelsZeroMeanNorm              :61-122                                      same on (v - mean)
//   zeromeanFeatureVolume             :570-594      v - mean
//   normalizedFeatureVolume           :504-550      v / norm              (true division; norm 0 -> NaN / inf)
//   zeromeanNormalizedFeatureVolume   :416-462      (v - mean) / norm
//   getFeatureVolumeForMatchFunc      :645-722      dispatch on the matching function's traits; census words for CENSUS / HAMMING
#include "svh_compare.h"
#include "svh_internal.h"

namespace svh {

namespace {

__global__ void channels_norm_given_kernel(const float *__restrict__ feat, const float *__restrict__ mean, int64_t npx, int F, float *__restrict__ norm) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npx; p += (int64_t)gridDim.x * blockDim.x) {
        const float *f = feat + p * F;
        const float m = mean[p];
        float n = 0.0f;
        for (int c = 0; c < F; c++) {
            const float tmp = f[c] - m;
            n += tmp * tmp;
        }
        norm[p] = sqrtf(n);
    }
}

// out = (v - mean) / norm, either map optional; one lane per element, the channel axis fastest
__global__ void affine_features_kernel(const float *__restrict__ feat, const float *__restrict__ mean, const float *__restrict__ norm, int64_t n, int F,
                                       float *__restrict__ out) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = e / F;
        float v = feat[e];
        if (mean) v = v - mean[p];
        if (norm) v = v / norm[p];
        out[e] = v;
    }
}

int dev_affine(svh_context *ctx, const float *feat, const float *mean, const float *norm, int64_t npx, int F, float *out) {
    const int64_t n = npx * F;
    if (n == 0) return SVH_OK;
    SVH_LAUNCH(ctx, "feature_transform", affine_features_kernel, grid_for(n, 256, 65536), 256, 0, feat, mean, norm, n, F, out);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

int check_map(svh_context *ctx, const svh_array *feat, const svh_array *map, const char *what) {
    SVH_TRY(validate(ctx, map, what, SVH_F32, 2, 2));
    if (map->shape[0] != feat->shape[0] || map->shape[1] != feat->shape[1])
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "%s must have shape (%lld,%lld)", what, (long long)feat->shape[0], (long long)feat->shape[1]);
    return SVH_OK;
}

// which: 0 mean, 1 norm, 2 zero-mean norm (mean given or computed)
int channel_statistic(svh_context *ctx, int which, const svh_array *feat, const svh_array *mean, svh_array *out) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, feat, "feat", SVH_F32, 3, 3));
    SVH_TRY(check_map(ctx, feat, out, "out"));
    if (mean) SVH_TRY(check_map(ctx, feat, mean, "mean"));
    const int H = (int)feat->shape[0], W = (int)feat->shape[1], F = (int)feat->shape[2];
    const int64_t npx = (int64_t)H * W;
    Scratch scr(ctx);
    void *df, *dm = nullptr;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *feat, &df));
    if (mean) SVH_TRY(stage_in(ctx, scr, *mean, &dm));
    SVH_TRY(stage_out(ctx, scr, *out, &os));
    if (npx > 0) {
        const FeatVolume fv{(const float *)df, W, F};
        if (which == 0) {
            SVH_TRY(launch_stats(ctx, fv, H, W, true, false, (float *)os.dptr, (float *)nullptr));
        } else if (which == 1) {
            SVH_TRY(launch_stats(ctx, fv, H, W, false, true, (float *)nullptr, (float *)os.dptr));
        } else if (dm) {
            const int st = launch_stats(ctx, fv, H, W, false, true, (float *)nullptr, (float *)os.dptr, (const float *)dm); // (the mean is given)
            if (st == SVH_ERR_UNSUPPORTED) {
                SVH_LAUNCH(ctx, "window_stats", channels_norm_given_kernel, grid_for(npx, 256, 16384), 256, 0, (const float *)df, (const float *)dm, npx, F,
                           (float *)os.dptr);
                SVH_CHECK_LAUNCH(ctx);
            } else if (st != SVH_OK) {
                return st;
            }
        } else {
            float *m = scr.get_n<float>((size_t)npx);
            if (!m) return SVH_ERR_OUT_OF_MEMORY;
            SVH_TRY(launch_stats(ctx, fv, H, W, true, true, m, (float *)os.dptr));
        }
    }
    return finish_out(ctx, os);
}

int transform_volume(svh_context *ctx, const svh_array *feat, const svh_array *mean, const svh_array *norm, svh_array *out) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, feat, "feat", SVH_F32, 3, 3));
    SVH_TRY(validate(ctx, out, "out", SVH_F32, 3, 3));
    if (mean) SVH_TRY(check_map(ctx, feat, mean, "mean"));
    if (norm) SVH_TRY(check_map(ctx, feat, norm, "norm"));
    for (int k = 0; k < 3; k++)
        if (out->shape[k] != feat->shape[k]) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "out must have the shape of feat");
    Scratch scr(ctx);
    void *df, *dm = nullptr, *dn = nullptr;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *feat, &df));
    if (mean) SVH_TRY(stage_in(ctx, scr, *mean, &dm));
    if (norm) SVH_TRY(stage_in(ctx, scr, *norm, &dn));
    SVH_TRY(stage_out(ctx, scr, *out, &os));
    SVH_TRY(dev_affine(ctx, (const float *)df, (const float *)dm, (const float *)dn, feat->shape[0] * feat->shape[1], (int)feat->shape[2], (float *)os.dptr));
    return finish_out(ctx, os);
}

} // namespace

} // namespace svh

using namespace svh;

namespace svh {
// getFeatureVolumeForMatchFunc for a float function on device arrays (dense (H, W, F)): v, v - mean, v / norm or (v - mean) / norm
int dev_feature_volume_for_match_func(svh_context *ctx, Scratch &scr, int match_func, const float *feat, int H, int W, int F, float *out) {
    const int64_t npx = (int64_t)H * W;
    if (npx * F == 0) return SVH_OK;
    const bool zm = func_zero_mean(match_func), nrm = func_normalized(match_func);
    float *m = nullptr, *n = nullptr;
    if (zm && !(m = scr.get_n<float>((size_t)npx))) return SVH_ERR_OUT_OF_MEMORY;
    if (nrm && !(n = scr.get_n<float>((size_t)npx))) return SVH_ERR_OUT_OF_MEMORY;
    if (zm || nrm) SVH_TRY(launch_stats(ctx, FeatVolume{feat, W, F}, H, W, zm, nrm, m, n));
    return dev_affine(ctx, feat, m, n, npx, F, out); // plain copy (cast) when neither applies, cross_correlations.h:716-720
}
} // namespace svh

extern "C" {

int svh_channels_mean(svh_context *ctx, const svh_array *feat, svh_array *mean) { return channel_statistic(ctx, 0, feat, nullptr, mean); }
int svh_channels_norm(svh_context *ctx, const svh_array *feat, svh_array *norm) { return channel_statistic(ctx, 1, feat, nullptr, norm); }
int svh_channels_zero_mean_norm(svh_context *ctx, const svh_array *feat, const svh_array *mean, svh_array *norm) {
    return channel_statistic(ctx, 2, feat, mean, norm);
}
int svh_zeromean_feature_volume(svh_context *ctx, const svh_array *feat, const svh_array *mean, svh_array *out) {
    if (!mean) return ctx ? fail(ctx, SVH_ERR_INVALID_ARGUMENT, "mean is required") : SVH_ERR_INVALID_ARGUMENT;
    return transform_volume(ctx, feat, mean, nullptr, out);
}
int svh_normalized_feature_volume(svh_context *ctx, const svh_array *feat, const svh_array *norm, svh_array *out) {
    if (!norm) return ctx ? fail(ctx, SVH_ERR_INVALID_ARGUMENT, "norm is required") : SVH_ERR_INVALID_ARGUMENT;
    return transform_volume(ctx, feat, nullptr, norm, out);
}
int svh_zeromean_normalized_feature_volume(svh_context *ctx, const svh_array *feat, const svh_array *mean, const svh_array *norm, svh_array *out) {
    if (!mean || !norm) return ctx ? fail(ctx, SVH_ERR_INVALID_ARGUMENT, "mean and norm are required") : SVH_ERR_INVALID_ARGUMENT;
    return transform_volume(ctx, feat, mean, norm, out);
}

int svh_feature_volume_for_match_func(svh_context *ctx, int match_func, const svh_array *feat, svh_array *out) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, feat, "feat", SVH_F32, 3, 3));
    if (!func_supported(match_func)) return fail(ctx, SVH_ERR_UNSUPPORTED, "matching function %d", match_func);
    const int H = (int)feat->shape[0], W = (int)feat->shape[1], F = (int)feat->shape[2];
    const int64_t npx = (int64_t)H * W;
    const bool census = func_census(match_func);
    SVH_TRY(validate(ctx, out, "out", census ? SVH_U32 : SVH_F32, 3, 3)); // FeatureType, matching_costs.h:742-788
    if (census && F <= 1) return fail(ctx, SVH_EMPTY_RESULT, "census needs at least two feature channels"); // census.h:76-78
    const int Fo = census ? census_words(F) : F;
    if (out->shape[0] != H || out->shape[1] != W || out->shape[2] != Fo) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "out must have shape (%d,%d,%d)", H, W, Fo);
    Scratch scr(ctx);
    void *df;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *feat, &df));
    SVH_TRY(stage_out(ctx, scr, *out, &os));
    if (census) {
        SVH_TRY(dev_census_from_features(ctx, (const float *)df, H, W, F, Fo, false, (uint32_t *)os.dptr));
    } else if (npx * F > 0) {
        SVH_TRY(dev_feature_volume_for_match_func(ctx, scr, match_func, (const float *)df, H, W, F, (float *)os.dptr));
    }
    return finish_out(ctx, os);
}

} // extern "C"

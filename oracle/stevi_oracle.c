/*
 * stevi_oracle.c -- CPU restatement of LibStevi's correlation/ hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the CPU baseline.  The HIP library (libstevi_amd/csrc) never links it.
 *
 * PARITY STATUS (see DESIGN.md "Oracle"):
 *   - the reference cannot be built in this image (MultidimArrays, StatusOptional, Eigen, FFTW
 *     are absent; CMakeLists.txt:62-76,:92,:95) and its tests hold no golden vectors (all inputs
 *     come from std::random_device, test/unittests/testCorrelationFilters.cpp:76-79);
 *   - rows pinned by the reference's own property tests, restated in tests/test_oracle_pins.py:
 *     unfold (testCorrelationFilters.cpp:384-445), ZCC/ZNCC volumes (:264-370,:462-500),
 *     NCC/SSD/ZSSD/SAD/ZSAD arithmetic (testCorrelation2d.cpp:75-127 via
 *     test/test_correlation_utils.h:9-310), parabola refinement (testCostRefinement.cpp:33-57);
 *   - rows with NO reference test or fixture -- census, Hamming, SGM, extractSelectedIndex tie
 *     rule, truncatedCostVolume: **parity unpinned**; they follow the cited lines operation by
 *     operation and are cross-checked only against hand-computed cases.
 *
 * All arrays are dense, row-major, last index fastest:
 *   images  [H][W][C]      feature volumes [H][W][F]      census words [H][W][nW]
 *   cost volumes [H][W][D]  index / disparity maps [H][W]
 * Each function cites the reference file:line it restates (paths relative to the LibStevi tree).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* enum values follow correlation/matching_costs.h:38-53 */
enum { SO_CC = 0, SO_NCC = 1, SO_SSD = 2, SO_SAD = 3, SO_ZCC = 4, SO_ZNCC = 5, SO_ZSSD = 6,
       SO_ZSAD = 7, SO_HAMMING = 10, SO_CENSUS = 11 };
/* correlation/correlation_base.h:31-45 */
enum { SO_COST = 0, SO_SCORE = 1 };
enum { SO_LEFT_TO_RIGHT = 0, SO_RIGHT_TO_LEFT = 1 };
enum { SO_TCV_SAME = 0, SO_TCV_REVERSED = 1, SO_TCV_BOTH = 2 };
/* correlation/cost_based_refinement.h:30-35 */
enum { SO_EQUIANGULAR = 0, SO_PARABOLA = 1, SO_GAUSSIAN = 2 };

int so_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void so_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---- traits: correlation/matching_costs.h:419-685 ------------------------------------------- */
static int func_zero_mean(int f) { return f == SO_ZCC || f == SO_ZNCC || f == SO_ZSSD || f == SO_ZSAD; }
static int func_normalized(int f) { return f == SO_NCC || f == SO_ZNCC; }
static int func_census(int f) { return f == SO_HAMMING || f == SO_CENSUS; }
int so_func_supported(int f) {
    return f == SO_CC || f == SO_NCC || f == SO_SSD || f == SO_SAD || func_zero_mean(f) || func_census(f);
}
/* extractionStrategy of each trait class */
int so_func_strategy(int f) {
    return (f == SO_CC || f == SO_NCC || f == SO_ZCC || f == SO_ZNCC) ? SO_SCORE : SO_COST;
}

/* number of census words allocated / written: correlation/census.h:80 and :103-108 */
int so_census_words(int F) { return (F - 1) / 32 + 1; }
int so_census_words_written(int F) { return (F - 1) / 32; }

/* ---- A1: unfold, correlation/unfold.h:247-344 (Rotate0 only) -------------------------------- *
 * pad[4] = left, top, right, bottom; pad == NULL means PaddingMargins() "auto" = (h_r, v_r).     *
 * Output shape (unfold.h:269-270): Ho = H - v + pt + pb + 1, Wo = W - h + pl + pr + 1.           */
void so_unfold_shape(int H, int W, int C, int h_r, int v_r, const int *pad, int *Ho, int *Wo, int *F) {
    int pl = pad ? pad[0] : h_r, pt = pad ? pad[1] : v_r, pr = pad ? pad[2] : h_r, pb = pad ? pad[3] : v_r;
    int h = 2 * h_r + 1, v = 2 * v_r + 1;
    *Ho = H - v + pt + pb + 1;
    *Wo = W - h + pl + pr + 1;
    *F = h * v * C;
}

void so_unfold(const float *img, int H, int W, int C, int h_r, int v_r, const int *pad, float *out) {
    int pl = pad ? pad[0] : h_r, pt = pad ? pad[1] : v_r;
    int Ho, Wo, F;
    so_unfold_shape(H, W, C, h_r, v_r, pad, &Ho, &Wo, &F);
    int h = 2 * h_r + 1, v = 2 * v_r + 1;
#pragma omp parallel for
    for (int i = 0; i < Ho; i++) {
        for (int j = 0; j < Wo; j++) {
            float *o = out + ((size_t)i * Wo + j) * F;
            for (int k = 0; k < v; k++) {
                for (int l = 0; l < h; l++) {
                    for (int ch = 0; ch < C; ch++) {
                        /* channelFromCord, unfold.h:180 */
                        int c = C * h * k + C * l + ch;
                        int ii = i - pt + k, jj = j - pl + l;
                        /* valueOrAlt(..., 0), unfold.h:284 / :335 */
                        o[c] = (ii >= 0 && ii < H && jj >= 0 && jj < W) ? img[((size_t)ii * W + jj) * C + ch] : 0.0f;
                    }
                }
            }
        }
    }
}

/* ---- A2: censusFeatures, correlation/census.h:69-115 ---------------------------------------- *
 * ref = channel 0 (:89); bit b of the running word := ref > val (:98-101); the word is stored    *
 * only when 32 bits are filled (:103-108), so the trailing partial word is never written.        *
 * Unwritten words are defined as 0 here (rule E1 of SURVEY.md section 8a).                       */
int so_census_features(const float *feat, int H, int W, int F, uint32_t *words) {
    if (F <= 1) return 1; /* census.h:76-78: empty array */
    int nW = so_census_words(F);
    memset(words, 0, (size_t)H * W * nW * sizeof(uint32_t));
#pragma omp parallel for
    for (int i = 0; i < H; i++) {
        for (int j = 0; j < W; j++) {
            const float *f = feat + ((size_t)i * W + j) * F;
            uint32_t *o = words + ((size_t)i * W + j) * nW;
            float ref = f[0];
            uint32_t d = 0;
            unsigned b = 0;
            int census_channel = 0;
            for (int c = 1; c < F; c++) {
                uint32_t g = (ref > f[c]) ? 1u : 0u;
                d |= g << b;
                b++;
                if (b >= 32) {
                    o[census_channel] = d;
                    census_channel++;
                    d = 0;
                    b = 0;
                }
            }
        }
    }
    return 0;
}

/* A3: censusTransform2D, correlation/census.h:117-131 = unfold then censusFeatures. */
int so_census_transform(const float *img, int H, int W, int C, int h_r, int v_r, const int *pad, uint32_t *words) {
    int Ho, Wo, F;
    so_unfold_shape(H, W, C, h_r, v_r, pad, &Ho, &Wo, &F);
    if (Ho <= 0 || Wo <= 0) return 1;
    float *feat = (float *)malloc((size_t)Ho * Wo * F * sizeof(float));
    if (!feat) return 2;
    so_unfold(img, H, W, C, h_r, v_r, pad, feat);
    int rc = so_census_features(feat, Ho, Wo, F, words);
    free(feat);
    return rc;
}

/* Rule E2: `float t = word; word' = t;` of correlation/cross_correlations.h:235-236.            *
 * uint32 -> float is round-to-nearest-even; float -> uint32 of 2^32 (words >= 0xFFFFFF80) is UB  *
 * in C++ and the reference's own builds differ (checked by running both in the container,        *
 * tests/test_oracle_semantics.py::test_e2_overflow_matches_this_hosts_conversions):              *
 *   mode 0, saturate -> 0xFFFFFFFF: vcvttss2usi, i.e. the Release build (-march=native,          *
 *           CMakeLists.txt:41) on a host with AVX-512;                                           *
 *   mode 1, zero -> 0: vcvttss2si r64 and the low 32 bits, i.e. x86-64 code generation without   *
 *           AVX-512 (-mavx -mavx2 -mfma, CMakeLists.txt:44-58; every Debug build).               *
 * The mode is process-wide test state (so_set_float_overflow), default 0.                        */
static int g_float_overflow_zero = 0;
void so_set_float_overflow(int zero) { g_float_overflow_zero = zero != 0; }
int so_get_float_overflow(void) { return g_float_overflow_zero; }
uint32_t so_round_word_through_float(uint32_t w) {
    float t = (float)w;
    if (t >= 4294967296.0f) return g_float_overflow_zero ? 0u : 0xFFFFFFFFu;
    return (uint32_t)t;
}

/* ---- A7: per-pixel channel statistics ------------------------------------------------------- */
/* channelsMean, correlation/correlation_base.h:1100-1136: sequential sum then * float(1./f) */
static void channels_mean(const float *feat, int H, int W, int F, float *mean) {
    float scale = (float)(1. / (double)(float)F);
#pragma omp parallel for
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            const float *f = feat + ((size_t)i * W + j) * F;
            float m = 0;
            for (int c = 0; c < F; c++) m += f[c];
            mean[(size_t)i * W + j] = m * scale;
        }
}
/* channelsZeroMeanNorm, correlation/cross_correlations.h:61-104 (float branch) */
static void channels_zeromean_norm(const float *feat, const float *mean, int H, int W, int F, float *norm) {
#pragma omp parallel for
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            const float *f = feat + ((size_t)i * W + j) * F;
            float m = mean[(size_t)i * W + j];
            float n = 0;
            for (int c = 0; c < F; c++) {
                float tmp = f[c] - m;
                n += tmp * tmp;
            }
            norm[(size_t)i * W + j] = sqrtf(n);
        }
}
/* channelsNorm, correlation/cross_correlations.h:149-191 (float branch) */
static void channels_norm(const float *feat, int H, int W, int F, float *norm) {
#pragma omp parallel for
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            const float *f = feat + ((size_t)i * W + j) * F;
            float n = 0;
            for (int c = 0; c < F; c++) {
                float tmp = f[c];
                n += tmp * tmp;
            }
            norm[(size_t)i * W + j] = sqrtf(n);
        }
}

void so_channels_mean(const float *feat, int H, int W, int F, float *mean) { channels_mean(feat, H, W, F, mean); }
void so_channels_norm(const float *feat, int H, int W, int F, float *norm) { channels_norm(feat, H, W, F, norm); }
void so_channels_zeromean_norm(const float *feat, int H, int W, int F, float *norm) {
    float *mean = (float *)malloc((size_t)H * W * sizeof(float));
    channels_mean(feat, H, W, F, mean);
    channels_zeromean_norm(feat, mean, H, W, F, norm);
    free(mean);
}

/* getFeatureVolumeForMatchFunc, correlation/cross_correlations.h:645-722, float input, non-census.  *
 * Writes the processed float feature volume in place of `out` (same shape as feat).                  */
static void processed_features(int func, const float *feat, int H, int W, int F, float *out) {
    size_t npx = (size_t)H * W;
    if (func_zero_mean(func) && func_normalized(func)) { /* :680-689, zeromeanNormalizedFeatureVolume :416-462 */
        float *mean = (float *)malloc(npx * sizeof(float));
        float *sigma = (float *)malloc(npx * sizeof(float));
        channels_mean(feat, H, W, F, mean);
        channels_zeromean_norm(feat, mean, H, W, F, sigma);
#pragma omp parallel for
        for (long p = 0; p < (long)npx; p++)
            for (int c = 0; c < F; c++) out[p * F + c] = (feat[p * F + c] - mean[p]) / sigma[p];
        free(mean);
        free(sigma);
    } else if (func_zero_mean(func)) { /* :691-699, zeromeanFeatureVolume :570-594 */
        float *mean = (float *)malloc(npx * sizeof(float));
        channels_mean(feat, H, W, F, mean);
#pragma omp parallel for
        for (long p = 0; p < (long)npx; p++)
            for (int c = 0; c < F; c++) out[p * F + c] = feat[p * F + c] - mean[p];
        free(mean);
    } else if (func_normalized(func)) { /* :701-709, normalizedFeatureVolume :504-550 */
        float *norm = (float *)malloc(npx * sizeof(float));
        channels_norm(feat, H, W, F, norm);
#pragma omp parallel for
        for (long p = 0; p < (long)npx; p++)
            for (int c = 0; c < F; c++) out[p * F + c] = feat[p * F + c] / norm[p];
        free(norm);
    } else { /* :716-720 cast only */
        memcpy(out, feat, npx * F * sizeof(float));
    }
}

/* ---- A4/A5: comparison kernels, correlation/matching_costs.h:59-156, :236-263 --------------- */
static float cmp_float(int func, const float *s, const float *t, int F) {
    float score = 0;
    switch (func) {
    case SO_CC: case SO_NCC: case SO_ZCC: case SO_ZNCC: /* dotProduct :59-78 */
        for (int c = 0; c < F; c++) score += s[c] * t[c];
        break;
    case SO_SSD: case SO_ZSSD: /* SumSquareDiff :100-116 */
        for (int c = 0; c < F; c++) {
            float tmp = s[c] - t[c];
            score += tmp * tmp;
        }
        break;
    default: /* SumAbsDiff :136-156 */
        for (int c = 0; c < F; c++) {
            float tmp = s[c] - t[c];
            score += fabsf(tmp);
        }
        break;
    }
    return score;
}

static float cmp_hamming(const uint32_t *s, const uint32_t *t, int nW) {
    uint16_t score = 0; /* hamming_cv_t, :234 */
    for (int w = 0; w < nW; w++) score += (uint16_t)__builtin_popcount(s[w] ^ t[w]);
    return (float)score; /* traits return float, :664/:681 */
}

/* ---- A6: aggregateCost, correlation/cross_correlations.h:194-308 ---------------------------- *
 * CV(i,j,d) = cmp(src(i,j,:), tgt(i, j + sign*(disp_lower + d), :)); the target vector is        *
 * gathered element by element through a `float` temporary, 0 where the column is outside         *
 * [0, Wt) (:234-237 / :293-296).  disp_lower = 0 is the disp_t overload, otherwise searchOffset. */
static void aggregate_float(int func, const float *src, const float *tgt, int H, int Ws, int Wt, int F,
                            int ddir, int disp_lower, int D, float *cv) {
    int sign = (ddir == SO_RIGHT_TO_LEFT) ? 1 : -1;
#pragma omp parallel
    {
        float *tv = (float *)malloc((size_t)F * sizeof(float));
#pragma omp for
        for (int i = 0; i < H; i++) {
            for (int j = 0; j < Ws; j++) {
                const float *sv = src + ((size_t)i * Ws + j) * F;
                for (int d = 0; d < D; d++) {
                    int jt = j + sign * (disp_lower + d);
                    for (int c = 0; c < F; c++) {
                        float t = (jt >= 0 && jt < Wt) ? tgt[((size_t)i * Wt + jt) * F + c] : 0.0f;
                        tv[c] = t;
                    }
                    cv[((size_t)i * Ws + j) * D + d] = cmp_float(func, sv, tv, F);
                }
            }
        }
        free(tv);
    }
}

static void aggregate_hamming(const uint32_t *src, const uint32_t *tgt, int H, int Ws, int Wt, int nW,
                              int ddir, int disp_lower, int D, float *cv) {
    int sign = (ddir == SO_RIGHT_TO_LEFT) ? 1 : -1;
#pragma omp parallel
    {
        uint32_t *tv = (uint32_t *)malloc((size_t)nW * sizeof(uint32_t));
#pragma omp for
        for (int i = 0; i < H; i++) {
            for (int j = 0; j < Ws; j++) {
                const uint32_t *sv = src + ((size_t)i * Ws + j) * nW;
                for (int d = 0; d < D; d++) {
                    int jt = j + sign * (disp_lower + d);
                    for (int w = 0; w < nW; w++) {
                        /* float t = valueOrAlt(...); target(c) = t;  -> rule E2 */
                        uint32_t raw = (jt >= 0 && jt < Wt) ? tgt[((size_t)i * Wt + jt) * nW + w] : 0u;
                        tv[w] = so_round_word_through_float(raw);
                    }
                    cv[((size_t)i * Ws + j) * D + d] = cmp_hamming(sv, tv, nW);
                }
            }
        }
        free(tv);
    }
}

/* featureVolume2CostVolume, correlation/cross_correlations.h:724-738.                           *
 * feat_l [H][Wl][F], feat_r [H][Wr][F] float.  Source/target selection: condImgRef,             *
 * correlation_base.h:829-878 (RightToLeft: source = right, target = left).                      *
 * cv is [H][Ws][D] with Ws the source width.  Returns non-zero where the reference returns an    *
 * empty array.                                                                                   */
int so_feature_cost_volume(int func, const float *feat_l, const float *feat_r, int H, int Wl, int Wr, int F,
                           int ddir, int disp_lower, int D, float *cv) {
    if (!so_func_supported(func) || D <= 0) return 1;
    const float *src = (ddir == SO_RIGHT_TO_LEFT) ? feat_r : feat_l;
    const float *tgt = (ddir == SO_RIGHT_TO_LEFT) ? feat_l : feat_r;
    int Ws = (ddir == SO_RIGHT_TO_LEFT) ? Wr : Wl;
    int Wt = (ddir == SO_RIGHT_TO_LEFT) ? Wl : Wr;
    if (func_census(func)) {
        if (F <= 1) return 1;
        int nW = so_census_words(F);
        uint32_t *ws = (uint32_t *)malloc((size_t)H * Ws * nW * sizeof(uint32_t));
        uint32_t *wt = (uint32_t *)malloc((size_t)H * Wt * nW * sizeof(uint32_t));
        so_census_features(src, H, Ws, F, ws);
        so_census_features(tgt, H, Wt, F, wt);
        aggregate_hamming(ws, wt, H, Ws, Wt, nW, ddir, disp_lower, D, cv);
        free(ws);
        free(wt);
    } else {
        float *ps = (float *)malloc((size_t)H * Ws * F * sizeof(float));
        float *pt = (float *)malloc((size_t)H * Wt * F * sizeof(float));
        processed_features(func, src, H, Ws, F, ps);
        processed_features(func, tgt, H, Wt, F, pt);
        aggregate_float(func, ps, pt, H, Ws, Wt, F, ddir, disp_lower, D, cv);
        free(ps);
        free(pt);
    }
    return 0;
}

/* unfoldBasedCostVolume, correlation/cross_correlations.h:740-765: unfold both images with auto  *
 * padding, then featureVolume2CostVolume.  Images [H][W*][C].                                    */
int so_unfold_cost_volume(int func, const float *img_l, const float *img_r, int Hl, int Wl, int Hr, int Wr, int C,
                          int h_r, int v_r, int ddir, int disp_lower, int D, float *cv) {
    if (Hl != Hr) return 1; /* :751-753 */
    int F = (2 * h_r + 1) * (2 * v_r + 1) * C;
    float *fl = (float *)malloc((size_t)Hl * Wl * F * sizeof(float));
    float *fr = (float *)malloc((size_t)Hr * Wr * F * sizeof(float));
    if (!fl || !fr) { free(fl); free(fr); return 2; }
    so_unfold(img_l, Hl, Wl, C, h_r, v_r, NULL, fl);
    so_unfold(img_r, Hr, Wr, C, h_r, v_r, NULL, fr);
    int rc = so_feature_cost_volume(func, fl, fr, Hl, Wl, Wr, F, ddir, disp_lower, D, cv);
    free(fl);
    free(fr);
    return rc;
}

/* ---- A9: SGM, correlation/sgm.h ------------------------------------------------------------- */
/* directionTraits, sgm.h:57-155, in the order of enum sgmDirections (:29-46) */
static const int SGM_STEPS_V[16][2] = {{1, 1}, {-1, -1}, {0, 0}, {0, 0}, {1, 1}, {-1, -1}, {1, 1}, {-1, -1},
                                       {0, 1}, {0, -1}, {0, 1}, {0, -1}, {1, 1}, {-1, -1}, {1, 1}, {-1, -1}};
static const int SGM_STEPS_H[16][2] = {{0, 0}, {0, 0}, {1, 1}, {-1, -1}, {1, 1}, {-1, -1}, {-1, -1}, {1, 1},
                                       {1, 1}, {-1, -1}, {-1, -1}, {1, 1}, {0, 1}, {0, -1}, {0, -1}, {0, 1}};
enum { SGM_NOSTART = 0, SGM_ZEROPOS = 1, SGM_ENDPOS = 2 };
/* startPostInfos, sgm.h:162-184 */
static int start_pos(const int s[2]) {
    if (s[0] == 0 && s[1] == 0) return SGM_NOSTART;
    if (s[0] >= 0 && s[1] >= 0) return SGM_ZEROPOS;
    return SGM_ENDPOS;
}

/* traverseLine, sgm.h:186-311, literal O(D^2) loops. */
static void traverse_line_faithful(int dir, int strategy, long start_i, long start_j, const float *cv, float *sgm,
                                   int H, int W, int D, float P1, float P2, const int m[4], float Pout) {
    const int *sv = SGM_STEPS_V[dir], *sh = SGM_STEPS_H[dir];
    int left = m[0], top = m[1], right = m[2], bottom = m[3];
    float *previous_cost = (float *)malloc((size_t)D * sizeof(float));
    float *actual_cost = (float *)malloc((size_t)D * sizeof(float));
    for (int d = 0; d < D; d++) previous_cost[d] = 0.0f;
    int c, i, j;
    for (c = 0, i = (int)start_i, j = (int)start_j;
         (i >= top && i < H - bottom) && (j >= left && j < W - right);
         i += sv[c % 2], j += sh[c % 2], c++) {
        const float *cvp = cv + ((size_t)i * W + j) * D;
        if (strategy == SO_SCORE) { /* :218-255 */
            float max_p_cost = -INFINITY;
            for (int d = 0; d < D; d++) {
                float p_score = previous_cost[d];
                if (p_score > max_p_cost && isfinite(p_score)) max_p_cost = previous_cost[d];
            }
            for (int nd = 0; nd < D; nd++) {
                float max_a_cost = -INFINITY;
                float c_score = cvp[nd];
                for (int od = 0; od < D; od++) {
                    float p_score = previous_cost[od];
                    if (abs(od - nd) == 1) p_score -= P1;
                    if (abs(od - nd) > 1) p_score -= P2;
                    if (p_score > max_a_cost && isfinite(p_score)) max_a_cost = p_score;
                }
                if (j + nd >= W) max_a_cost -= Pout;
                actual_cost[nd] = c_score;
                if (isfinite(max_a_cost) && isfinite(max_p_cost)) actual_cost[nd] += max_a_cost - max_p_cost;
            }
        } else { /* :257-296 */
            float min_p_cost = INFINITY;
            for (int d = 0; d < D; d++) {
                float p_score = previous_cost[d];
                if (p_score < min_p_cost && isfinite(p_score)) min_p_cost = previous_cost[d];
            }
            for (int nd = 0; nd < D; nd++) {
                float min_a_cost = INFINITY;
                float c_score = cvp[nd];
                for (int od = 0; od < D; od++) {
                    float p_score = previous_cost[od];
                    if (abs(od - nd) == 1) p_score += P1;
                    if (abs(od - nd) > 1) p_score += P2;
                    if (p_score < min_a_cost && isfinite(p_score)) min_a_cost = c_score; /* sic, :281-283 */
                }
                if (j + nd >= W) min_a_cost += Pout;
                actual_cost[nd] = c_score;
                if (isfinite(min_a_cost) && isfinite(min_p_cost)) actual_cost[nd] += min_a_cost - min_p_cost;
            }
        }
        float *sp = sgm + ((size_t)i * W + j) * D;
        for (int d = 0; d < D; d++) sp[d] += actual_cost[d] - cvp[d]; /* :298-300 */
        float *tmp = previous_cost;
        previous_cost = actual_cost;
        actual_cost = tmp;
    }
    free(previous_cost);
    free(actual_cost);
}

/* Same recurrence in O(D) per pixel; must agree bitwise with traverse_line_faithful whenever      *
 * prev[od] -/+ P stays finite for finite prev[od] (no float overflow).                            *
 * Score: max over od of fl(prev[od] - pen(od,nd)) = fl(max prev[od] - pen) per penalty class       *
 * because x -> fl(x - P) is monotone; the |od-nd|>1 class uses exclusive prefix/suffix maxima.     *
 * Cost: min_a_cost is c_score as soon as one penalised prev is finite and smaller than the running *
 * value -- reproduced with the same comparisons on the three penalty classes.                      */
static void traverse_line_linear(int dir, int strategy, long start_i, long start_j, const float *cv, float *sgm,
                                 int H, int W, int D, float P1, float P2, const int m[4], float Pout) {
    const int *sv = SGM_STEPS_V[dir], *sh = SGM_STEPS_H[dir];
    int left = m[0], top = m[1], right = m[2], bottom = m[3];
    float *prev = (float *)malloc((size_t)D * sizeof(float));
    float *act = (float *)malloc((size_t)D * sizeof(float));
    float *pre = (float *)malloc((size_t)(D + 2) * sizeof(float)); /* pre[k] = ext over finite prev[0..k-1] */
    float *suf = (float *)malloc((size_t)(D + 2) * sizeof(float)); /* suf[k] = ext over finite prev[k..D-1] */
    for (int d = 0; d < D; d++) prev[d] = 0.0f;
    int c, i, j;
    for (c = 0, i = (int)start_i, j = (int)start_j;
         (i >= top && i < H - bottom) && (j >= left && j < W - right);
         i += sv[c % 2], j += sh[c % 2], c++) {
        const float *cvp = cv + ((size_t)i * W + j) * D;
        if (strategy == SO_SCORE) {
            pre[0] = -INFINITY;
            for (int d = 0; d < D; d++) pre[d + 1] = (isfinite(prev[d]) && prev[d] > pre[d]) ? prev[d] : pre[d];
            suf[D] = -INFINITY;
            suf[D + 1] = -INFINITY;
            for (int d = D - 1; d >= 0; d--) suf[d] = (isfinite(prev[d]) && prev[d] > suf[d + 1]) ? prev[d] : suf[d + 1];
            float max_p = pre[D];
            for (int nd = 0; nd < D; nd++) {
                float a = -INFINITY;
                float far_l = (nd >= 2) ? pre[nd - 1] : -INFINITY; /* prev[0..nd-2] */
                float far_r = suf[nd + 2 <= D ? nd + 2 : D];       /* prev[nd+2..] */
                float far = far_l > far_r ? far_l : far_r;
                float cand;
                cand = prev[nd];
                if (cand > a && isfinite(cand)) a = cand;
                if (nd >= 1) { cand = prev[nd - 1] - P1; if (cand > a && isfinite(cand)) a = cand; }
                if (nd + 1 < D) { cand = prev[nd + 1] - P1; if (cand > a && isfinite(cand)) a = cand; }
                cand = far - P2;
                if (cand > a && isfinite(cand)) a = cand;
                if (j + nd >= W) a -= Pout;
                act[nd] = cvp[nd];
                if (isfinite(a) && isfinite(max_p)) act[nd] += a - max_p;
            }
        } else {
            float min_p = INFINITY;
            int nfinite = 0;
            for (int d = 0; d < D; d++)
                if (isfinite(prev[d])) { nfinite++; if (prev[d] < min_p) min_p = prev[d]; }
            for (int nd = 0; nd < D; nd++) {
                float c_score = cvp[nd];
                /* min_a becomes c_score iff some penalised prev is finite (every finite value is < +inf) */
                float a = nfinite > 0 ? c_score : INFINITY;
                if (j + nd >= W) a += Pout;
                act[nd] = c_score;
                if (isfinite(a) && isfinite(min_p)) act[nd] += a - min_p;
            }
        }
        float *sp = sgm + ((size_t)i * W + j) * D;
        for (int d = 0; d < D; d++) sp[d] += act[d] - cvp[d];
        float *tmp = prev;
        prev = act;
        act = tmp;
    }
    free(prev);
    free(act);
    free(pre);
    free(suf);
}

typedef void (*line_fn)(int, int, long, long, const float *, float *, int, int, int, float, float, const int *, float);

/* addDirectionalCost, sgm.h:313-356.  `serial` runs the lines of one start loop in index order      *
 * (needed for the overlapping 16-direction lines, where the reference itself is a data race).       */
static void add_directional_cost(line_fn fn, int dir, int strategy, const float *cv, float *sgm, int H, int W, int D,
                                 float P1, float P2, const int m[4], float Pout, int serial) {
    int colStart = start_pos(SGM_STEPS_V[dir]); /* from the vertical steps, sgm.h:167-173 */
    int rowStart = start_pos(SGM_STEPS_H[dir]); /* from the horizontal steps, :175-181 */
    int left = m[0], top = m[1], right = m[2], bottom = m[3];
    if (rowStart != SGM_NOSTART) { /* :329-341 */
        long start_j = (rowStart == SGM_ZEROPOS) ? left : W - right;
        long bi = top, ei = H - bottom;
#pragma omp parallel for if (!serial)
        for (long start_i = bi; start_i < ei; start_i++) fn(dir, strategy, start_i, start_j, cv, sgm, H, W, D, P1, P2, m, Pout);
    }
    if (colStart != SGM_NOSTART) { /* :343-354 */
        long start_i = (colStart == SGM_ZEROPOS) ? top : H - bottom;
        long bj = left, ej = W - right;
#pragma omp parallel for if (!serial)
        for (long start_j = bj; start_j < ej; start_j++) fn(dir, strategy, start_i, start_j, cv, sgm, H, W, D, P1, P2, m, Pout);
    }
}

/* sgmCostVolume, sgm.h:360-404.  margins = left, top, right, bottom.  variant 0 = literal O(D^2),    *
 * 1 = O(D).                                                                                          */
int so_sgm(int n_dir, int strategy, const float *cv, int H, int W, int D, float P1, float P2, const int margins[4],
           float Pout, float *sgm, int variant) {
    if (n_dir != 4 && n_dir != 8 && n_dir != 16) return 1;
    line_fn fn = variant ? traverse_line_linear : traverse_line_faithful;
    memcpy(sgm, cv, (size_t)H * W * D * sizeof(float)); /* :371-377 */
    static const int order4[4] = {0, 1, 2, 3};          /* :379-382 */
    static const int order8[4] = {4, 5, 6, 7};          /* :385-388 */
    static const int order16[8] = {12, 13, 14, 15, 8, 9, 10, 11}; /* :392-400 */
    for (int k = 0; k < 4; k++) add_directional_cost(fn, order4[k], strategy, cv, sgm, H, W, D, P1, P2, margins, Pout, 0);
    if (n_dir >= 8)
        for (int k = 0; k < 4; k++) add_directional_cost(fn, order8[k], strategy, cv, sgm, H, W, D, P1, P2, margins, Pout, 0);
    if (n_dir >= 16)
        for (int k = 0; k < 8; k++) add_directional_cost(fn, order16[k], strategy, cv, sgm, H, W, D, P1, P2, margins, Pout, 1);
    return 0;
}

/* one direction alone, accumulated into a caller-initialised sgm volume (test helper) */
int so_sgm_add_direction(int dir, int strategy, const float *cv, int H, int W, int D, float P1, float P2,
                         const int margins[4], float Pout, float *sgm, int variant) {
    if (dir < 0 || dir >= 16) return 1;
    add_directional_cost(variant ? traverse_line_linear : traverse_line_faithful, dir, strategy, cv, sgm, H, W, D, P1, P2,
                         margins, Pout, dir >= 8);
    return 0;
}

/* ---- A10: extractSelectedIndex, correlation/correlation_base.h:427-464 ----------------------- */
void so_extract_index(int strategy, const float *cv, int H, int W, int D, int32_t *idx) {
#pragma omp parallel for
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            const float *p = cv + ((size_t)i * W + j) * D;
            float sel = p[0];
            int32_t sd = 0;
            for (int d = 1; d < D; d++) {
                if (strategy == SO_COST) {
                    if (p[d] <= sel) { sel = p[d]; sd = d; }
                } else {
                    if (p[d] >= sel) { sel = p[d]; sd = d; }
                }
            }
            idx[(size_t)i * W + j] = sd;
        }
}

/* selectedIndexToDisp, correlation_base.h:511-532 */
void so_index_to_disp(int ddir, const int32_t *idx, int H, int W, int32_t offset, int32_t *disp) {
    int32_t sign = (ddir == SO_RIGHT_TO_LEFT) ? 1 : -1;
    for (size_t p = 0; p < (size_t)H * W; p++) disp[p] = sign * idx[p] + offset;
}

/* selectedCost, correlation_base.h:557-577 */
void so_selected_cost(const float *cv, const int32_t *idx, int H, int W, int D, float *out) {
    for (size_t p = 0; p < (size_t)H * W; p++) out[p] = cv[p * D + (uint32_t)idx[p]];
}

/* ---- A11: truncatedCostVolume, correlation_base.h:579-674 ------------------------------------ */
int so_truncated_cv_depth(int sdir, int r) { return sdir == SO_TCV_BOTH ? 4 * r + 1 : 2 * r + 1; }

void so_truncated_cost_volume(int sdir, int ddir, const float *cv, const int32_t *idx, int H, int W, int D, int h_r,
                              int v_r, int r, float *tcv) {
    int T = so_truncated_cv_depth(sdir, r);
#pragma omp parallel for
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            float *o = tcv + ((size_t)i * W + j) * T;
            int32_t sel = idx[(size_t)i * W + j];
            for (int32_t d = 0; d <= 2 * r; d++) {
                int32_t p = sel + d - r;
                if (sdir == SO_TCV_SAME) { /* :601-613 */
                    if (p < 0 || p >= D || j < h_r || j + p + h_r >= W || i < v_r || i + v_r >= H)
                        o[d] = nanf("");
                    else
                        o[d] = cv[((size_t)i * W + j) * D + p];
                } else if (sdir == SO_TCV_REVERSED) { /* :615-630 */
                    int32_t sgn = (ddir == SO_RIGHT_TO_LEFT) ? -1 : 1;
                    int32_t jp = j + sgn * (d - r);
                    int32_t mn = jp < j ? jp : j, mx = jp > j ? jp : j;
                    if (p < 0 || p >= D || mn < h_r || mx + h_r >= W || i < v_r || i + v_r >= H)
                        o[d] = nanf("");
                    else
                        o[d] = cv[((size_t)i * W + jp) * D + p];
                } else { /* Both, :632-667 */
                    int32_t sgn = (ddir == SO_RIGHT_TO_LEFT) ? -1 : 1;
                    int32_t jp = j + sgn * (d - r);
                    int32_t d_d = 2 * d, d_r = 2 * d + 1;
                    if (d == r) jp = -1;
                    if (d > r) { d_d -= 1; d_r -= 1; }
                    if (p < 0 || p >= D || j < h_r || j + p + h_r >= W || i < v_r || i + v_r >= H)
                        o[d_d] = nanf("");
                    else
                        o[d_d] = cv[((size_t)i * W + j) * D + p];
                    int32_t mn = jp < j ? jp : j, mx = jp > j ? jp : j;
                    /* at d == r the reference writes tcv(i,j,d_r) with d_r == 2r+1 == slot of d=r+1's d_d;
                       it is NaN there (jp = -1 < h_radius unless h_radius == 0 ... then value(i,-1,p)) and is
                       overwritten by the next iteration's d_d store, so skip the out-of-bounds read. */
                    if (d == r) continue;
                    if (p < 0 || p >= D || mn < h_r || mx + h_r >= W || i < v_r || i + v_r >= H)
                        o[d_r] = nanf("");
                    else
                        o[d_r] = cv[((size_t)i * W + jp) * D + p];
                }
            }
        }
}

/* ---- A12: refineCostTriplet / refineDispCostInterpolation, cost_based_refinement.h:43-69, :128-163 */
float so_refine_triplet(int kernel, float cm1, float c0, float c1) {
    float val = 0;
    switch (kernel) {
    case SO_EQUIANGULAR: {
        float alpha = copysignf(1.f, c0 - cm1);
        alpha *= fmaxf(fabsf(c0 - cm1), fabsf(c1 - c0));
        val = (c1 - cm1) / (2 * alpha);
    } break;
    case SO_PARABOLA:
        val = (cm1 - c1) / (2 * (c1 - 2 * c0 + cm1));
        break;
    case SO_GAUSSIAN:
        val = (logf(cm1) - logf(c1)) / (2 * (logf(c1) - 2 * logf(c0) + logf(cm1)));
        break;
    }
    return val;
}

int so_refine_disp(int kernel, const float *tcv, const int32_t *raw, int H, int W, int T, float *refined) {
    int cv_radius = (T - 1) / 2;
    if (cv_radius < 1 || 2 * cv_radius + 1 != T) return 1; /* :141-143 */
#pragma omp parallel for
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            const float *t = tcv + ((size_t)i * W + j) * T;
            float delta = so_refine_triplet(kernel, t[cv_radius - 1], t[cv_radius], t[cv_radius + 1]);
            refined[(size_t)i * W + j] = (float)raw[(size_t)i * W + j] + delta;
        }
    return 0;
}

/* ===== 2-D disparity (optical-flow style) volumes: SURVEY.md section 8(f) rank 2 ================================= */

/* aggregateCost(searchOffset<2>), correlation/cross_correlations.h:310-374:
 * CV(i,j,dh,dw) = cmp(src(i,j,:), tgt(i + dh + lower0, j + dw + lower1, :)), zero target vector outside the image
 * (:359).  No direction sign here: dDir only selects which image is the source.  cv is [H][Ws][Dh][Dw]. */
static void aggregate_float_2d(int func, const float *src, const float *tgt, int H, int Ws, int Wt, int F, int lower0, int Dh,
                               int lower1, int Dw, float *cv) {
#pragma omp parallel
    {
        float *tv = (float *)malloc((size_t)F * sizeof(float));
#pragma omp for
        for (int i = 0; i < H; i++)
            for (int j = 0; j < Ws; j++) {
                const float *sv = src + ((size_t)i * Ws + j) * F;
                for (int dh = 0; dh < Dh; dh++)
                    for (int dw = 0; dw < Dw; dw++) {
                        int it = i + dh + lower0, jt = j + dw + lower1;
                        int in = it >= 0 && it < H && jt >= 0 && jt < Wt;
                        for (int c = 0; c < F; c++) tv[c] = in ? tgt[((size_t)it * Wt + jt) * F + c] : 0.0f;
                        cv[(((size_t)i * Ws + j) * Dh + dh) * Dw + dw] = cmp_float(func, sv, tv, F);
                    }
            }
        free(tv);
    }
}

static void aggregate_hamming_2d(const uint32_t *src, const uint32_t *tgt, int H, int Ws, int Wt, int nW, int lower0, int Dh,
                                 int lower1, int Dw, float *cv) {
#pragma omp parallel
    {
        uint32_t *tv = (uint32_t *)malloc((size_t)nW * sizeof(uint32_t));
#pragma omp for
        for (int i = 0; i < H; i++)
            for (int j = 0; j < Ws; j++) {
                const uint32_t *sv = src + ((size_t)i * Ws + j) * nW;
                for (int dh = 0; dh < Dh; dh++)
                    for (int dw = 0; dw < Dw; dw++) {
                        int it = i + dh + lower0, jt = j + dw + lower1;
                        int in = it >= 0 && it < H && jt >= 0 && jt < Wt;
                        for (int w = 0; w < nW; w++)
                            tv[w] = so_round_word_through_float(in ? tgt[((size_t)it * Wt + jt) * nW + w] : 0u);
                        cv[(((size_t)i * Ws + j) * Dh + dh) * Dw + dw] = cmp_hamming(sv, tv, nW);
                    }
            }
        free(tv);
    }
}

/* unfoldBased2dDisparityCostVolume, cross_correlations.h:794-822 (rows AND columns must agree, :801-810) */
int so_unfold_cost_volume_2d(int func, const float *img_l, const float *img_r, int Hl, int Wl, int Hr, int Wr, int C, int h_r,
                             int v_r, int ddir, int lower0, int upper0, int lower1, int upper1, float *cv) {
    if (Hl != Hr || Wl != Wr) return 1;
    int Dh = upper0 - lower0 + 1, Dw = upper1 - lower1 + 1;
    if (Dh <= 0 || Dw <= 0 || !so_func_supported(func)) return 1; /* :338-340 */
    int H = Hl, W = Wl, F = (2 * h_r + 1) * (2 * v_r + 1) * C;
    float *fl = (float *)malloc((size_t)H * W * F * sizeof(float));
    float *fr = (float *)malloc((size_t)H * W * F * sizeof(float));
    if (!fl || !fr) { free(fl); free(fr); return 2; }
    so_unfold(img_l, H, W, C, h_r, v_r, NULL, fl);
    so_unfold(img_r, H, W, C, h_r, v_r, NULL, fr);
    const float *src = (ddir == SO_RIGHT_TO_LEFT) ? fr : fl, *tgt = (ddir == SO_RIGHT_TO_LEFT) ? fl : fr;
    if (func_census(func)) {
        if (F <= 1) { free(fl); free(fr); return 1; }
        int nW = so_census_words(F);
        uint32_t *ws = (uint32_t *)malloc((size_t)H * W * nW * sizeof(uint32_t));
        uint32_t *wt = (uint32_t *)malloc((size_t)H * W * nW * sizeof(uint32_t));
        so_census_features(src, H, W, F, ws);
        so_census_features(tgt, H, W, F, wt);
        aggregate_hamming_2d(ws, wt, H, W, W, nW, lower0, Dh, lower1, Dw, cv);
        free(ws);
        free(wt);
    } else {
        float *ps = (float *)malloc((size_t)H * W * F * sizeof(float));
        float *pt = (float *)malloc((size_t)H * W * F * sizeof(float));
        processed_features(func, src, H, W, F, ps);
        processed_features(func, tgt, H, W, F, pt);
        aggregate_float_2d(func, ps, pt, H, W, W, F, lower0, Dh, lower1, Dw, cv);
        free(ps);
        free(pt);
    }
    free(fl);
    free(fr);
    return 0;
}

/* extractSelected2dIndex, correlation_base.h:466-509: scan (d1, d2) in row-major order with <= / >=, starting from
 * cv(i,j,0,0): the last extremum wins.  idx is [H][W][2]. */
void so_extract_index_2d(int strategy, const float *cv, int H, int W, int D1, int D2, int32_t *idx) {
#pragma omp parallel for
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            const float *p = cv + ((size_t)i * W + j) * D1 * D2;
            float sel = p[0];
            int32_t s1 = 0, s2 = 0;
            for (int d1 = 0; d1 < D1; d1++)
                for (int d2 = 0; d2 < D2; d2++) {
                    float v = p[d1 * D2 + d2];
                    if (strategy == SO_COST ? (v <= sel) : (v >= sel)) { sel = v; s1 = d1; s2 = d2; }
                }
            idx[((size_t)i * W + j) * 2] = s1;
            idx[((size_t)i * W + j) * 2 + 1] = s2;
        }
}

/* selected2dIndexToDisp, correlation_base.h:534-555 */
void so_index_2d_to_disp(const int32_t *idx, int H, int W, int lower0, int lower1, int32_t *disp) {
    for (size_t p = 0; p < (size_t)H * W; p++) {
        disp[2 * p] = idx[2 * p] + lower0;
        disp[2 * p + 1] = idx[2 * p + 1] + lower1;
    }
}

/* truncatedBidirectionaCostVolume, correlation_base.h:677-725 (explicit radii); tcv is [H][W][2r0+1][2r1+1] */
void so_truncated_bidirectional_cv(const float *cv, const int32_t *idx, int H, int W, int D1, int D2, int r0, int r1, float *tcv) {
    int T0 = 2 * r0 + 1, T1 = 2 * r1 + 1;
#pragma omp parallel for
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            size_t px = (size_t)i * W + j;
            for (int d0 = 0; d0 < T0; d0++) {
                int p0 = idx[2 * px] + d0 - r0;
                for (int d1 = 0; d1 < T1; d1++) {
                    int p1 = idx[2 * px + 1] + d1 - r1;
                    tcv[(px * T0 + d0) * T1 + d1] = (p0 >= 0 && p0 < D1 && p1 >= 0 && p1 < D2) ? cv[(px * D1 + p0) * D2 + p1] : nanf("");
                }
            }
        }
}

/* ===== 2-D cost-based refinement: SURVEY.md section 8(f) rank 1 (the refinement stereo-match --refine calls) ========= */

#define SO_ISOTROPIC 0
#define SO_ANISOTROPIC 1

/* refineCostPatch<Parabola|Gaussian>, cost_based_refinement.h:71-126: least-squares quadric through the 3x3 patch
 * (parameters: v^2, v h, h^2, v, h, 1), fitted = ((A^T A)^-1 A^T) L in float, then the stationary point M^-1 v.
 * The 6x6 inverse is a float Gauss-Jordan elimination with partial pivoting (Eigen's inverse() of a 6x6 goes through a
 * partial-pivoting LU; A^T A holds small integers, so the two differ by rounding only); the 2x2 inverse is the
 * adjugate times 1/det like Eigen's fixed-size 2x2 path. */
static void so_refine_patch(int kernel, const float c[9], float delta[2]) {
    float L[9];
    for (int k = 0; k < 9; k++) L[k] = kernel == SO_GAUSSIAN ? logf(c[k]) : c[k]; /* :119-121 */
    static const float vd[9] = {-1, -1, -1, 0, 0, 0, 1, 1, 1}, hd[9] = {-1, 0, 1, -1, 0, 1, -1, 0, 1}; /* :91-94 */
    float A[9][6];
    for (int i = 0; i < 9; i++) { /* :98-105 */
        A[i][0] = vd[i] * vd[i];
        A[i][1] = vd[i] * hd[i];
        A[i][2] = hd[i] * hd[i];
        A[i][3] = vd[i];
        A[i][4] = hd[i];
        A[i][5] = 1;
    }
    float N[6][12];
    for (int r = 0; r < 6; r++)
        for (int q = 0; q < 6; q++) {
            float s = 0;
            for (int i = 0; i < 9; i++) s += A[i][r] * A[i][q];
            N[r][q] = s;
            N[r][6 + q] = r == q ? 1.0f : 0.0f;
        }
    for (int col = 0; col < 6; col++) {
        int piv = col;
        for (int r = col + 1; r < 6; r++)
            if (fabsf(N[r][col]) > fabsf(N[piv][col])) piv = r;
        if (piv != col)
            for (int q = 0; q < 12; q++) { float t = N[col][q]; N[col][q] = N[piv][q]; N[piv][q] = t; }
        float inv = 1.0f / N[col][col];
        for (int q = 0; q < 12; q++) N[col][q] *= inv;
        for (int r = 0; r < 6; r++) {
            if (r == col) continue;
            float f = N[r][col];
            for (int q = 0; q < 12; q++) N[r][q] -= f * N[col][q];
        }
    }
    float P[6][9]; /* (A^T A)^-1 A^T */
    for (int r = 0; r < 6; r++)
        for (int i = 0; i < 9; i++) {
            float s = 0;
            for (int q = 0; q < 6; q++) s += N[r][6 + q] * A[i][q];
            P[r][i] = s;
        }
    float f[6];
    for (int r = 0; r < 6; r++) {
        float s = 0;
        for (int i = 0; i < 9; i++) s += P[r][i] * L[i];
        f[r] = s;
    }
    float m00 = 2 * f[0], m01 = f[1], m10 = f[1], m11 = 2 * f[2]; /* :109-111 */
    float v0 = -f[3], v1 = -f[4];                                  /* :113-114 */
    float invdet = 1.0f / (m00 * m11 - m10 * m01);
    float i00 = m11 * invdet, i01 = -m01 * invdet, i10 = -m10 * invdet, i11 = m00 * invdet;
    delta[0] = i00 * v0 + i01 * v1;
    delta[1] = i10 * v0 + i11 * v1;
}

/* refineDisp2dCostInterpolation<kernel, isotropy>, cost_based_refinement.h:165-376.  tcv [H][W][T0][T1], raw and refined
 * [H][W][2].  Returns 1 for the shapes the reference answers with an empty array (:180-182). */
int so_refine_disp_2d(int kernel, int isotropy, const float *tcv, const int32_t *raw, int H, int W, int T0, int T1, float *refined) {
    int r0 = (T0 - 1) / 2, r1 = (T1 - 1) / 2;
    if (r0 < 1 || r1 < 1 || 2 * r0 + 1 != T0 || 2 * r1 + 1 != T1) return 1;
#define TCV(i, j, a, b) tcv[(((size_t)(i) * W + (j)) * T0 + (a)) * T1 + (b)]
    int is_score = 0; /* :184-203: probed once at the centre pixel; every comparison with a NaN is false */
    if (H > 0 && W > 0) {
        int ic = H / 2, jc = W / 2;
        float v0 = TCV(ic, jc, r0, r1);
        if (v0 > TCV(ic, jc, r0 + 1, r1)) is_score = 1;
        if (v0 > TCV(ic, jc, r0 - 1, r1)) is_score = 1;
        if (v0 > TCV(ic, jc, r0, r1 + 1)) is_score = 1;
        if (v0 > TCV(ic, jc, r0, r1 - 1)) is_score = 1;
    }
#pragma omp parallel for
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            float delta0 = 0, delta1 = 0;
            if (isotropy == SO_ISOTROPIC) { /* :257-268 */
                delta0 = so_refine_triplet(kernel, TCV(i, j, r0 - 1, r1), TCV(i, j, r0, r1), TCV(i, j, r0 + 1, r1));
                delta1 = so_refine_triplet(kernel, TCV(i, j, r0, r1 - 1), TCV(i, j, r0, r1), TCV(i, j, r0, r1 + 1));
            } else {
                /* extremum along axis 0 in column `col` of axis 1 (argminForRow, :205-225) and along axis 1 in row `row`
                 * of axis 0 (argminForCol, :227-247); last extremum wins, NaN never selected, default 0 */
                int am[4];
                for (int which = 0; which < 4; which++) {
                    int fixed = which == 0 ? r1 - 1 : which == 1 ? r1 + 1 : which == 2 ? r0 - 1 : r0 + 1;
                    int n = which < 2 ? T0 : T1;
                    float hat = is_score ? -INFINITY : INFINITY;
                    int arg = 0;
                    for (int a = 0; a < n; a++) {
                        float v = which < 2 ? TCV(i, j, a, fixed) : TCV(i, j, fixed, a);
                        if (is_score ? (v >= hat) : (v <= hat)) { hat = v; arg = a; }
                    }
                    am[which] = arg;
                }
                float delta0_1 = so_refine_triplet(kernel, TCV(i, j, r0 - 1, r1), TCV(i, j, r0, r1), TCV(i, j, r0 + 1, r1)); /* :274-278 */
                float delta0_0 = delta0_1, delta0_2 = delta0_1;
                if (am[0] > 0 && am[0] < T0 - 1) /* :285-293 */
                    delta0_0 = am[0] - r0 + so_refine_triplet(kernel, TCV(i, j, am[0] - 1, r1 - 1), TCV(i, j, am[0], r1 - 1), TCV(i, j, am[0] + 1, r1 - 1));
                if (am[1] > 0 && am[1] < T0 - 1) /* :300-308 */
                    delta0_2 = am[1] - r0 + so_refine_triplet(kernel, TCV(i, j, am[1] - 1, r1 + 1), TCV(i, j, am[1], r1 + 1), TCV(i, j, am[1] + 1, r1 + 1));
                float a0 = (delta0_2 - delta0_0) / 2; /* :311-312 */
                float b0 = (delta0_0 + delta0_1 + delta0_2) / 3;
                float delta1_1 = so_refine_triplet(kernel, TCV(i, j, r0, r1 - 1), TCV(i, j, r0, r1), TCV(i, j, r0, r1 + 1)); /* :316-320 */
                float delta1_0 = delta1_1, delta1_2 = delta1_1;
                if (am[2] > 0 && am[2] < T1 - 1) /* :327-335 */
                    delta1_0 = am[2] - r1 + so_refine_triplet(kernel, TCV(i, j, r0 - 1, am[2] - 1), TCV(i, j, r0 - 1, am[2]), TCV(i, j, r0 - 1, am[2] + 1));
                if (am[3] > 0 && am[3] < T1 - 1) /* :341-349 */
                    delta1_2 = am[3] - r1 + so_refine_triplet(kernel, TCV(i, j, r0 + 1, am[3] - 1), TCV(i, j, r0 + 1, am[3]), TCV(i, j, r0 + 1, am[3] + 1));
                float a1 = (delta1_2 - delta1_0) / 2; /* :352-353 */
                float b1 = (delta1_0 + delta1_1 + delta1_2) / 3;
                delta0 = (a0 * b1 + b0) / (1 - a0 * a1); /* :357-358 */
                delta1 = (a1 * b0 + b1) / (1 - a0 * a1);
            }
            if (fabsf(delta0) > 1 || fabsf(delta1) > 1 || isnan(delta0) || isnan(delta1)) { delta0 = 0; delta1 = 0; } /* :362-366 */
            size_t px = (size_t)i * W + j;
            refined[2 * px] = (float)raw[2 * px] + delta0;
            refined[2 * px + 1] = (float)raw[2 * px + 1] + delta1;
        }
    return 0;
}

/* refineDisp2dCostPatchInterpolation<Parabola|Gaussian>, cost_based_refinement.h:378-436 */
int so_refine_disp_2d_patch(int kernel, const float *tcv, const int32_t *raw, int H, int W, int T0, int T1, float *refined) {
    int r0 = (T0 - 1) / 2, r1 = (T1 - 1) / 2;
    if (r0 < 1 || r1 < 1 || 2 * r0 + 1 != T0 || 2 * r1 + 1 != T1) return 1;
    if (kernel != SO_PARABOLA && kernel != SO_GAUSSIAN) return 2; /* static_assert, :83 */
#pragma omp parallel for
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            float c[9], d[2];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) c[3 * a + b] = TCV(i, j, r0 - 1 + a, r1 - 1 + b); /* :405-415 */
            so_refine_patch(kernel, c, d);
            if (fabsf(d[0]) > 1 || fabsf(d[1]) > 1 || isnan(d[0]) || isnan(d[1])) { d[0] = 0; d[1] = 0; } /* :424-428 */
            size_t px = (size_t)i * W + j;
            refined[2 * px] = (float)raw[2 * px] + d[0];
            refined[2 * px + 1] = (float)raw[2 * px + 1] + d[1];
        }
    return 0;
#undef TCV
}

/* ===== hierarchical matching: SURVEY.md section 8(f) rank 3 ========================================================= */

/* Interpolation::averagePoolingDownsample, interpolation/downsampling.h:67-178, as written: the inner row loop runs over
 * windows.horizontal() (:92, :147), and the column offset is derived from the ROW remainder and vice versa (:82-86,
 * :136-140).  With the 2x2 windows hierarchical.h uses, both offsets are 0 and odd sizes give a last window of one valid
 * row / column whose mean is taken over the valid samples only.  img [H][W][C] -> out [Ho][Wo][C], float accumulation. */
void so_downsample_shape(int H, int W, int win_h, int win_v, int *Ho, int *Wo) {
    *Ho = (H + (win_v - 1)) / win_v;
    *Wo = (W + (win_h - 1)) / win_h;
}

void so_average_pooling_downsample(const float *img, int H, int W, int C, int win_h, int win_v, float *out) {
    int Ho, Wo;
    so_downsample_shape(H, W, win_h, win_v, &Ho, &Wo);
    int hRem = Ho * win_v - H, vRem = Wo * win_h - W;
    int initialHOffset = hRem / 2, initialVOffset = vRem / 2;
#pragma omp parallel for
    for (int i = 0; i < Ho; i++)
        for (int j = 0; j < Wo; j++)
            for (int f = 0; f < C; f++) {
                float val = 0;
                int count = 0;
                for (int dv = 0; dv < win_h; dv++) {
                    int p_i = i * win_v - initialVOffset + dv;
                    for (int dh = 0; dh < win_h; dh++) {
                        int p_j = j * win_h - initialHOffset + dh;
                        if (p_i >= 0 && p_i < H && p_j >= 0 && p_j < W) {
                            val += img[((size_t)p_i * W + p_j) * C + f];
                            count += 1;
                        }
                    }
                }
                val /= count;
                out[((size_t)i * Wo + j) * C + f] = val;
            }
}

/* computeGuidedCV<matchFunc>, correlation/hierarchical.h:74-229.  feat_l / feat_r are the feature volumes
 * getFeatureVolumeForMatchFunc returns: float [H][W*][F] (is_census = 0) or uint32 words [H][W*][nW] (is_census = 1; the
 * target words are gathered as uint32, NOT through a float: :175-178).  guide [Hg][Wg] int32, Hg, Wg >= 2 (the bilinear
 * taps of :108-137 read outside the array otherwise).  Outputs: tcv [H][Ws][2r+1], disp [H][Ws].
 * After the re-centring of :200-228, tcv(i,j,dd) = cmp(src(i,j), tgt(i, j + d_r + dirSign (dd - r))) for every dd, computed
 * here exactly as the reference does (first pass around d0, shift, fill the uncovered entries). */
int so_guided_cv(int func, int is_census, const void *feat_l, const void *feat_r, int H, int Wl, int Wr, int F, int ddir, const int32_t *guide,
                 int Hg, int Wg, int radius, float *tcv, int32_t *disp) {
    if (Hg < 2 || Wg < 2 || radius < 0) return 1;
    const int dirSign = ddir == SO_RIGHT_TO_LEFT ? 1 : -1;
    const void *src = ddir == SO_RIGHT_TO_LEFT ? feat_r : feat_l, *tgt = ddir == SO_RIGHT_TO_LEFT ? feat_l : feat_r;
    const int w = ddir == SO_RIGHT_TO_LEFT ? Wr : Wl, wt = ddir == SO_RIGHT_TO_LEFT ? Wl : Wr, h = H;
    const int depth = 2 * radius + 1;
    const int cost = so_func_strategy(func) == SO_COST;
#pragma omp parallel
    {
        void *tv = malloc((size_t)F * 4);
#define GUIDED_CMP(col, res)                                                                                                       \
    do {                                                                                                                           \
        int c_ = (col);                                                                                                            \
        if (is_census) {                                                                                                           \
            for (int c = 0; c < F; c++) ((uint32_t *)tv)[c] = (c_ >= 0 && c_ < wt) ? ((const uint32_t *)tgt)[((size_t)i * wt + c_) * F + c] : 0u; \
            res = cmp_hamming((const uint32_t *)src + ((size_t)i * w + j) * F, (const uint32_t *)tv, F);                           \
        } else {                                                                                                                   \
            for (int c = 0; c < F; c++) ((float *)tv)[c] = (c_ >= 0 && c_ < wt) ? ((const float *)tgt)[((size_t)i * wt + c_) * F + c] : 0.0f; \
            res = cmp_float(func, (const float *)src + ((size_t)i * w + j) * F, (const float *)tv, F);                             \
        }                                                                                                                          \
    } while (0)
#pragma omp for
        for (int i = 0; i < h; i++) {
            float v_pos = (float)(i * (Hg - 1)) / (h - 1); /* :108 */
            int v0 = (int)floorf(v_pos), v1 = (int)ceilf(v_pos);
            if (v0 == v1) v1 += 1;
            if (v1 == Hg) { v0 -= 1; v1 -= 1; }
            for (int j = 0; j < w; j++) {
                float h_pos = (float)(j * (Wg - 1)) / (w - 1); /* :124 */
                int h0 = (int)floorf(h_pos), h1 = (int)ceilf(h_pos);
                if (h0 == h1) h1 += 1;
                if (h1 == Wg) { h0 -= 1; h1 -= 1; }
                float interp = 0; /* :138-147 */
                interp += (v_pos - v0) * (h_pos - h0) * guide[(size_t)v1 * Wg + h1];
                interp += (v1 - v_pos) * (h_pos - h0) * guide[(size_t)v0 * Wg + h1];
                interp += (v_pos - v0) * (h1 - h_pos) * guide[(size_t)v1 * Wg + h0];
                interp += (v1 - v_pos) * (h1 - h_pos) * guide[(size_t)v0 * Wg + h0];
                interp *= 2;
                int32_t d0 = dirSign * (int32_t)roundf(interp); /* :150 */
                float *o = tcv + ((size_t)i * w + j) * depth;
                float score = cost ? INFINITY : -INFINITY;
                int32_t d_r = d0;
                for (int delta_d = -radius; delta_d <= radius; delta_d++) { /* :157-190 */
                    float cmp;
                    GUIDED_CMP(j + d0 + delta_d, cmp);
                    o[dirSign * delta_d + radius] = cmp;
                    if (cost ? (cmp < score) : (cmp > score)) { score = cmp; d_r = d0 + delta_d; }
                }
                disp[(size_t)i * w + j] = dirSign * d_r; /* :192 */
                if (d_r != d0) {                         /* :194-227 */
                    int delta = dirSign * (d0 - d_r), startempty, endempty;
                    if (delta > 0) {
                        for (int dd = depth - 1; dd >= delta; dd--) o[dd] = o[dd - delta];
                        startempty = 0;
                        endempty = delta;
                    } else {
                        for (int dd = 0; dd < depth + delta; dd++) o[dd] = o[dd - delta];
                        startempty = depth + delta;
                        endempty = depth;
                    }
                    for (int dd = startempty; dd < endempty; dd++) GUIDED_CMP(j + d_r + dirSign * (dd - radius), o[dd]);
                }
            }
        }
        free(tv);
    }
#undef GUIDED_CMP
    return 0;
}

/* feature volume of one image for a matching function: unfold (auto padding) + getFeatureVolumeForMatchFunc
 * (cross_correlations.h:645-722); returns a malloc'ed float [H][W][F] or uint32 [H][W][nW] volume and the channel count */
static void *match_features(int func, const float *img, int H, int W, int C, int h_r, int v_r, int *nchan) {
    int F = (2 * h_r + 1) * (2 * v_r + 1) * C;
    float *raw = (float *)malloc((size_t)H * W * F * sizeof(float));
    so_unfold(img, H, W, C, h_r, v_r, NULL, raw);
    if (func_census(func)) {
        int nW = so_census_words(F);
        uint32_t *words = (uint32_t *)malloc((size_t)H * W * nW * sizeof(uint32_t));
        so_census_features(raw, H, W, F, words);
        free(raw);
        *nchan = nW;
        return words;
    }
    float *out = (float *)malloc((size_t)H * W * F * sizeof(float));
    processed_features(func, raw, H, W, F, out);
    free(raw);
    *nchan = F;
    return out;
}

/* hiearchicalTruncatedCostVolume<matchFunc, depth>, correlation/hierarchical.h:232-294: recursion over 2x2 average-pooled
 * images down to `depth` levels, full search with ceil(disp_width / 2^depth) disparities at the coarsest level
 * (unfoldBasedCostVolume + extractSelectedIndex, :253-260), then one computeGuidedCV per level on the way up.
 * h_radii / v_radii have depth + 1 entries, coarsest first (:256, :262 use [0] and [1]; :289 uses back()).
 * Outputs at the resolution of the source image: tcv [H][Ws][2r+1], disp [H][Ws]. */
int so_hierarchical_truncated_cv(int func, int depth, const float *img_l, const float *img_r, int H, int Wl, int Wr, int C,
                                 const int *h_radii, const int *v_radii, int disp_width, int radius, int ddir, float *tcv, int32_t *disp) {
    if (depth < 1 || !so_func_supported(func)) return 1;
    int Hd, Wld, Wrd;
    so_downsample_shape(H, Wl, 2, 2, &Hd, &Wld);
    so_downsample_shape(H, Wr, 2, 2, &Hd, &Wrd);
    float *dl = (float *)malloc((size_t)Hd * Wld * C * sizeof(float)), *dr = (float *)malloc((size_t)Hd * Wrd * C * sizeof(float));
    so_average_pooling_downsample(img_l, H, Wl, C, 2, 2, dl);
    so_average_pooling_downsample(img_r, H, Wr, C, 2, 2, dr);
    int Wsd = ddir == SO_RIGHT_TO_LEFT ? Wrd : Wld;
    int32_t *guide = (int32_t *)malloc((size_t)Hd * Wsd * sizeof(int32_t));
    int rc = 0;
    if (depth == 1) {
        int D0 = (disp_width + 1) / 2;
        float *cv = (float *)malloc((size_t)Hd * Wsd * D0 * sizeof(float));
        rc = so_unfold_cost_volume(func, dl, dr, Hd, Wld, Hd, Wrd, C, h_radii[0], v_radii[0], ddir, 0, D0, cv);
        if (rc == 0) so_extract_index(so_func_strategy(func), cv, Hd, Wsd, D0, guide);
        free(cv);
    } else {
        float *tprev = (float *)malloc((size_t)Hd * Wsd * (2 * radius + 1) * sizeof(float));
        rc = so_hierarchical_truncated_cv(func, depth - 1, dl, dr, Hd, Wld, Wrd, C, h_radii, v_radii, (disp_width + 1) / 2, radius, ddir, tprev, guide);
        free(tprev);
    }
    if (rc == 0) {
        int nl, nr;
        void *fl = match_features(func, img_l, H, Wl, C, h_radii[depth], v_radii[depth], &nl);
        void *fr = match_features(func, img_r, H, Wr, C, h_radii[depth], v_radii[depth], &nr);
        rc = so_guided_cv(func, func_census(func), fl, fr, H, Wl, Wr, nl, ddir, guide, Hd, Wsd, radius, tcv, disp);
        free(fl);
        free(fr);
    }
    free(dl);
    free(dr);
    free(guide);
    return rc;
}

/* the feature volumes computeGuidedCV is given in hierarchical.h:262-265 / :288-291, exposed for the tests */
int so_match_features(int func, const float *img, int H, int W, int C, int h_r, int v_r, void *out) {
    int n;
    void *f = match_features(func, img, H, W, C, h_r, v_r, &n);
    memcpy(out, f, (size_t)H * W * n * 4);
    free(f);
    return n;
}

/* featureVolume2CostVolume<matchFunc, ..., searchOffset<2>> on raw feature volumes, cross_correlations.h:724-738 over
 * aggregateCost :310-374: only the row counts must agree (:324-326).  cv [H][Ws][Dh][Dw]. */
int so_feature_cost_volume_2d(int func, const float *feat_l, const float *feat_r, int H, int Wl, int Wr, int F, int ddir, int lower0, int upper0,
                              int lower1, int upper1, float *cv) {
    int Dh = upper0 - lower0 + 1, Dw = upper1 - lower1 + 1;
    if (Dh <= 0 || Dw <= 0 || !so_func_supported(func)) return 1;
    const float *src = ddir == SO_RIGHT_TO_LEFT ? feat_r : feat_l, *tgt = ddir == SO_RIGHT_TO_LEFT ? feat_l : feat_r;
    int Ws = ddir == SO_RIGHT_TO_LEFT ? Wr : Wl, Wt = ddir == SO_RIGHT_TO_LEFT ? Wl : Wr;
    if (func_census(func)) {
        if (F <= 1) return 1;
        int nW = so_census_words(F);
        uint32_t *ws = (uint32_t *)malloc((size_t)H * Ws * nW * sizeof(uint32_t)), *wt = (uint32_t *)malloc((size_t)H * Wt * nW * sizeof(uint32_t));
        so_census_features(src, H, Ws, F, ws);
        so_census_features(tgt, H, Wt, F, wt);
        aggregate_hamming_2d(ws, wt, H, Ws, Wt, nW, lower0, Dh, lower1, Dw, cv);
        free(ws);
        free(wt);
    } else {
        float *ps = (float *)malloc((size_t)H * Ws * F * sizeof(float)), *pt = (float *)malloc((size_t)H * Wt * F * sizeof(float));
        processed_features(func, src, H, Ws, F, ps);
        processed_features(func, tgt, H, Wt, F, pt);
        aggregate_float_2d(func, ps, pt, H, Ws, Wt, F, lower0, Dh, lower1, Dw, cv);
        free(ps);
        free(pt);
    }
    return 0;
}

/* ---- A7 / A8 as stand-alone functions (callers such as examples/stereo_refine_test/main.cpp:386-398 use them directly) ---- */
/* channelsZeroMeanNorm with an explicit mean map, cross_correlations.h:61-104 */
void so_channels_zeromean_norm_given(const float *feat, const float *mean, int H, int W, int F, float *norm) {
    channels_zeromean_norm(feat, mean, H, W, F, norm);
}
/* zeromeanFeatureVolume :570-594 (norm == NULL), normalizedFeatureVolume :504-550 (mean == NULL),
 * zeromeanNormalizedFeatureVolume :416-462 (both): (v - mean) / norm with exactly those operations */
void so_affine_feature_volume(const float *feat, const float *mean, const float *norm, int H, int W, int F, float *out) {
#pragma omp parallel for
    for (long p = 0; p < (long)H * W; p++)
        for (int c = 0; c < F; c++) {
            float v = feat[p * F + c];
            if (mean) v = v - mean[p];
            if (norm) v = v / norm[p];
            out[p * F + c] = v;
        }
}
/* getFeatureVolumeForMatchFunc on a feature volume, :645-722: float [H][W][F] or census words [H][W][nW]; returns the
 * channel count of the result */
int so_feature_volume_for_match_func(int func, const float *feat, int H, int W, int F, void *out) {
    if (func_census(func)) {
        so_census_features(feat, H, W, F, (uint32_t *)out);
        return so_census_words(F);
    }
    processed_features(func, feat, H, W, F, (float *)out);
    return F;
}

/* ===== UnFoldCompressor: SURVEY.md section 8(f) rank 4 (second half) ================================================ */

/* UnFoldCompressor(mask), correlation/unfold.h:47-121: positive labels of an odd- or even-sized int mask are superpixels;
 * offsets are relative to (height/2, width/2); features are numbered in increasing label order; the index list is ordered by
 * feature, then row-major inside the mask; weight = float(1. / pixels of the superpixel); the bounding box always contains the
 * centre (min / max start at 0, :57-60).  Fills at most mh*mw entries; returns the entry count. */
typedef struct { int v, h, f; float w; } so_pix_index;
static int compressor_build(const int32_t *mask, int mh, int mw, so_pix_index *idx, int *n_features, int *box /* minH,maxH,minW,maxW */) {
    int v_off = mh / 2, h_off = mw / 2, minH = 0, maxH = 0, minW = 0, maxW = 0, nf = 0, n = 0;
    int32_t *labels = (int32_t *)malloc((size_t)mh * mw * sizeof(int32_t));
    int *counts = (int *)malloc((size_t)mh * mw * sizeof(int));
    for (int i = 0; i < mh; i++)
        for (int j = 0; j < mw; j++) {
            int32_t feat = mask[i * mw + j];
            if (feat <= 0) continue;
            if (i - v_off < minH) minH = i - v_off;
            if (i - v_off > maxH) maxH = i - v_off;
            if (j - h_off < minW) minW = j - h_off;
            if (j - h_off > maxW) maxW = j - h_off;
            int k = 0;
            while (k < nf && labels[k] != feat) k++;
            if (k == nf) { labels[nf] = feat; counts[nf] = 0; nf++; }
            counts[k]++;
        }
    for (int a = 0; a < nf; a++) /* std::sort of the labels, :103 */
        for (int b = a + 1; b < nf; b++)
            if (labels[b] < labels[a]) { int32_t t = labels[a]; labels[a] = labels[b]; labels[b] = t; int c = counts[a]; counts[a] = counts[b]; counts[b] = c; }
    for (int f = 0; f < nf; f++)
        for (int i = 0; i < mh; i++)
            for (int j = 0; j < mw; j++)
                if (mask[i * mw + j] == labels[f]) {
                    idx[n].v = i - v_off; idx[n].h = j - h_off; idx[n].f = f; idx[n].w = (float)(1. / counts[f]);
                    n++;
                }
    *n_features = nf;
    box[0] = minH; box[1] = maxH; box[2] = minW; box[3] = maxW;
    free(labels);
    free(counts);
    return n;
}

/* unfold(compressor, img, padding), unfold.h:346-471; pad = {left, top, right, bottom} or NULL for auto (= compressor.margins()).
 * img [H][W][C] -> out [Ho][Wo][C * nFeatures], channel-major feature index in_c * nFeatures + f (:455). */
void so_unfold_compressed_shape(int H, int W, int C, const int32_t *mask, int mh, int mw, const int *pad, int *Ho, int *Wo, int *F) {
    so_pix_index *idx = (so_pix_index *)malloc((size_t)mh * mw * sizeof(so_pix_index));
    int nf, box[4];
    compressor_build(mask, mh, mw, idx, &nf, box);
    int pl = pad ? pad[0] : -box[2], pt = pad ? pad[1] : -box[0], pr = pad ? pad[2] : box[3], pb = pad ? pad[3] : box[1];
    *Ho = H - (box[1] - box[0] + 1) + pt + pb + 1;
    *Wo = W - (box[3] - box[2] + 1) + pl + pr + 1;
    *F = C * nf;
    free(idx);
}

void so_unfold_compressed(const float *img, int H, int W, int C, const int32_t *mask, int mh, int mw, const int *pad, float *out) {
    so_pix_index *idx = (so_pix_index *)malloc((size_t)mh * mw * sizeof(so_pix_index));
    int nf, box[4];
    int n = compressor_build(mask, mh, mw, idx, &nf, box);
    int top = -box[0], left = -box[2];
    int pl = pad ? pad[0] : left, pt = pad ? pad[1] : top, pr = pad ? pad[2] : box[3], pb = pad ? pad[3] : box[1];
    int Ho = H - (box[1] - box[0] + 1) + pt + pb + 1, Wo = W - (box[3] - box[2] + 1) + pl + pr + 1, F = C * nf;
    if (Ho > 0 && Wo > 0) {
        memset(out, 0, (size_t)Ho * Wo * F * sizeof(float));
        for (int e = 0; e < n; e++) /* :395-411 / :444-464: one pass over the output per index entry (and channel) */
            for (int c = 0; c < C; c++) {
#pragma omp parallel for
                for (int i = 0; i < Ho; i++) {
                    int in_i = i + idx[e].v + top - pt;
                    for (int j = 0; j < Wo; j++) {
                        int in_j = j + idx[e].h + left - pl;
                        float v = (in_i >= 0 && in_i < H && in_j >= 0 && in_j < W) ? img[((size_t)in_i * W + in_j) * C + c] : 0.0f;
                        out[((size_t)i * Wo + j) * F + c * nf + idx[e].f] += idx[e].w * v;
                    }
                }
            }
    }
    free(idx);
}

/* ---- unfold with a patch orientation, correlation/unfold.h:139-191, :247-344 ---------------------------------------------
 * channelFromCord(k, l, ch, hSize, vSize, channels, orientation) places window sample (k, l, ch) at the position it has after
 * rotating the patch by 0 / 90 / 180 / 270 degrees; the samples themselves are read exactly as for Rotate0. */
static int channel_from_cord(int vertical, int horizontal, int channel, int hSize, int vSize, int channels, int orientation) {
    switch (orientation) {
    case 0: return channels * hSize * vertical + channels * horizontal + channel;
    case 1: return channels * vSize * (hSize - horizontal - 1) + channels * vertical + channel;
    case 2: return channels * hSize * (vSize - vertical - 1) + channels * (hSize - horizontal - 1) + channel;
    case 3: return channels * vSize * horizontal + channels * (vSize - vertical - 1) + channel;
    }
    return -1;
}

void so_unfold_oriented(const float *img, int H, int W, int C, int h_r, int v_r, const int *pad, int orientation, float *out) {
    int pl = pad ? pad[0] : h_r, pt = pad ? pad[1] : v_r;
    int Ho, Wo, F;
    so_unfold_shape(H, W, C, h_r, v_r, pad, &Ho, &Wo, &F);
    int h = 2 * h_r + 1, v = 2 * v_r + 1;
#pragma omp parallel for
    for (int i = 0; i < Ho; i++)
        for (int j = 0; j < Wo; j++) {
            float *o = out + ((size_t)i * Wo + j) * F;
            for (int k = 0; k < v; k++)
                for (int l = 0; l < h; l++)
                    for (int ch = 0; ch < C; ch++) {
                        int ii = i - pt + k, jj = j - pl + l;
                        o[channel_from_cord(k, l, ch, h, v, C, orientation)] = (ii >= 0 && ii < H && jj >= 0 && jj < W) ? img[((size_t)ii * W + jj) * C + ch] : 0.0f;
                    }
        }
}

/* ===== "textbook" SGM: SURVEY.md section 8(f) rank 4 (first half) ===================================================
 * NOT the reference's behaviour (that is so_sgm, findings F4 / F5): what correlation/sgm.h evidently intends.
 *   - all 8 (or 4) directions are traversed, every line exactly once, in the call order of sgmCostVolume (sgm.h:379-388):
 *     Up2Down, Down2Up, Left2Right, Right2Left, UpLeft2DownRight, DownRight2UpLeft, UpRight2DownLeft, DownLeft2UpRight;
 *   - Cost strategy:  a(nd) = min over finite { prev[nd], prev[nd-1] + P1, prev[nd+1] + P1, min_{|od-nd|>1} prev[od] + P2 }
 *     (the penalised neighbour, not the pixel's own cost: the fix of sgm.h:281-283), + Pout where j + nd >= W;
 *     Score strategy: the mirror image with max and -P (sgm.h:230-255 as written, it has no such slip);
 *   - act(nd) = c(nd) + (a(nd) - ext) when a and ext = min / max over finite prev are finite, else c(nd); S += act - c;
 *     prev = 0 at the start of every line; pixels outside the margin box keep S = C.
 * This function is the definition the device mode is checked against (there is nothing in the reference to pin it to). */
static void textbook_line(int strategy, long i0, long j0, int di, int dj, int len, const float *cv, float *sgm, int W, int D, float P1, float P2,
                          float Pout, float *prev, float *act, float *pre, float *suf) {
    const int cost = strategy == SO_COST;
    const float worst = cost ? INFINITY : -INFINITY;
    for (int d = 0; d < D; d++) prev[d] = 0.0f;
    for (int s = 0; s < len; s++) {
        long i = i0 + (long)s * di, j = j0 + (long)s * dj;
        const float *c = cv + ((size_t)i * W + j) * D;
        float *S = sgm + ((size_t)i * W + j) * D;
        /* exclusive prefix / suffix extrema over the finite previous values: pre[d] covers od < d, suf[d] covers od > d */
        float run = worst;
        for (int d = 0; d < D; d++) { pre[d] = run; if (isfinite(prev[d])) run = cost ? fminf(run, prev[d]) : fmaxf(run, prev[d]); }
        float ext = run;
        run = worst;
        for (int d = D - 1; d >= 0; d--) { suf[d] = run; if (isfinite(prev[d])) run = cost ? fminf(run, prev[d]) : fmaxf(run, prev[d]); }
        for (int nd = 0; nd < D; nd++) {
            float a = worst;
#define TB_TAKE(v) do { float v_ = (v); if (isfinite(v_)) a = cost ? fminf(a, v_) : fmaxf(a, v_); } while (0)
            TB_TAKE(prev[nd]);
            if (nd > 0) TB_TAKE(cost ? prev[nd - 1] + P1 : prev[nd - 1] - P1);
            if (nd + 1 < D) TB_TAKE(cost ? prev[nd + 1] + P1 : prev[nd + 1] - P1);
            float far = worst; /* |od - nd| > 1: od <= nd - 2 or od >= nd + 2 */
            if (nd >= 1) far = cost ? fminf(far, pre[nd - 1]) : fmaxf(far, pre[nd - 1]);
            if (nd + 1 < D) far = cost ? fminf(far, suf[nd + 1]) : fmaxf(far, suf[nd + 1]);
            TB_TAKE(cost ? far + P2 : far - P2);
#undef TB_TAKE
            if (j + nd >= W) a = cost ? a + Pout : a - Pout;
            act[nd] = c[nd];
            if (isfinite(a) && isfinite(ext)) act[nd] = c[nd] + (a - ext);
        }
        for (int d = 0; d < D; d++) { S[d] += act[d] - c[d]; prev[d] = act[d]; }
    }
}

int so_sgm_textbook(int n_dir, int strategy, const float *cv, int H, int W, int D, float P1, float P2, const int margins[4], float Pout, float *out) {
    if (n_dir != 4 && n_dir != 8) return 1;
    memcpy(out, cv, (size_t)H * W * D * sizeof(float));
    int left = margins[0], top = margins[1], Hp = H - margins[1] - margins[3], Wp = W - margins[0] - margins[2];
    if (Hp <= 0 || Wp <= 0 || D <= 0) return 0;
    int n_pass = n_dir == 8 ? 8 : 4;
    for (int q = 0; q < n_pass; q++) {
        int n_lines = q < 2 ? Wp : q < 4 ? Hp : Hp + Wp - 1;
#pragma omp parallel
        {
            float *buf = (float *)malloc((size_t)4 * D * sizeof(float));
#pragma omp for schedule(dynamic, 4)
            for (int l = 0; l < n_lines; l++) {
                long i0, j0;
                int di, dj, len;
                switch (q) {
                case 0: i0 = 0; j0 = l; di = 1; dj = 0; len = Hp; break;
                case 1: i0 = Hp - 1; j0 = l; di = -1; dj = 0; len = Hp; break;
                case 2: i0 = l; j0 = 0; di = 0; dj = 1; len = Wp; break;
                case 3: i0 = l; j0 = Wp - 1; di = 0; dj = -1; len = Wp; break;
                case 4: case 5: { /* diagonal k = j - i = l - (Hp - 1) */
                    int k = l - (Hp - 1);
                    i0 = k <= 0 ? -k : 0; j0 = k <= 0 ? 0 : k;
                    len = (Hp - i0) < (Wp - j0) ? (int)(Hp - i0) : (int)(Wp - j0);
                    di = 1; dj = 1;
                    if (q == 5) { i0 += len - 1; j0 += len - 1; di = -1; dj = -1; }
                } break;
                default: { /* anti-diagonal s = i + j = l */
                    i0 = l < Wp ? 0 : l - (Wp - 1); j0 = l < Wp ? l : Wp - 1;
                    len = (Hp - i0) < (j0 + 1) ? (int)(Hp - i0) : (int)(j0 + 1);
                    di = 1; dj = -1;
                    if (q == 7) { i0 += len - 1; j0 -= len - 1; di = -1; dj = 1; }
                } break;
                }
                textbook_line(strategy, top + i0, left + j0, di, dj, len, cv, out, W, D, P1, P2, Pout, buf, buf + D, buf + 2 * D, buf + 3 * D);
            }
            free(buf);
        }
    }
    return 0;
}

/* ===== on-demand (cacheless) cost volumes and PatchMatch: SURVEY.md section 8(f) rank 1, what examples/stereo-match runs ==========
 *
 * Features: OnDemandDecoratedFeaturesVolume<ZNFeaturesVolumeDecorator<ZeroMean, Normalized>, float, 3, ..., 2>
 * (correlation/on_demand_features_volume.h:34-214) over the window list examples/stereo-match/main.cpp:150-164 builds: offsets
 * (di, dj) in [-r, r]^2 row-major, channels innermost.  Samples outside the image are CLAMPED to the border (:128-133), not
 * zero; the decorator divides by the feature count: mean = (sum v) / nF, v -= mean, norm = sqrt((sum v^2) / nF), v /= norm
 * (:168-214).  So these are not the features of the dense path (zero padding, sqrt of the plain sum).
 * Cost: CachelessOnDemandCostVolume::costValue (correlation/on_demand_cost_volume.h:409-468): no value when a disparity is
 * outside the search range or the target position is outside the target image, else featureComparison of the two vectors. */
typedef struct {
    int func, search_dims, h_r, v_r, lower[2], upper[2];
    int Hs, Ws, Ht, Wt, C, nF;
    const float *fs, *ft; /* decorated feature volumes [H][W][nF] */
} od_volume;

static void on_demand_features(int func, const float *img, int H, int W, int C, int h_r, int v_r, float *out) {
    const int nF = (2 * v_r + 1) * (2 * h_r + 1) * C;
    const int zm = func_zero_mean(func), nrm = func_normalized(func);
#pragma omp parallel for
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            float *v = out + ((size_t)i * W + j) * nF;
            int f = 0;
            for (int di = -v_r; di <= v_r; di++)
                for (int dj = -h_r; dj <= h_r; dj++)
                    for (int c = 0; c < C; c++) {
                        int ii = i + di, jj = j + dj; /* constant border condition, :128-133 */
                        ii = ii < 0 ? 0 : (ii > H - 1 ? H - 1 : ii);
                        jj = jj < 0 ? 0 : (jj > W - 1 ? W - 1 : jj);
                        v[f++] = img[((size_t)ii * W + jj) * C + c];
                    }
            if (zm) { /* :183-192 */
                float mean = 0;
                for (int k = 0; k < nF; k++) mean += v[k];
                mean /= nF;
                for (int k = 0; k < nF; k++) v[k] -= mean;
            }
            if (nrm) { /* :194-207 */
                float norm = 0;
                for (int k = 0; k < nF; k++) norm += v[k] * v[k];
                norm /= nF;
                norm = sqrtf(norm);
                for (int k = 0; k < nF; k++) v[k] /= norm;
            }
        }
}

/* costValue, on_demand_cost_volume.h:409-468; returns 0 when there is no value.  disp[0] = rows, disp[1] = columns for the flow
 * volume; disp[0] = columns for the stereo volume. */
static int od_cost(const od_volume *o, int i, int j, const int *disp, float *cost) {
    int ti = i, tj = j;
    if (i < 0 || i >= o->Hs || j < 0 || j >= o->Ws) return 0;
    for (int s = 0; s < o->search_dims; s++) {
        if (disp[s] < o->lower[s] || disp[s] > o->upper[s]) return 0;
    }
    if (o->search_dims == 2) { ti += disp[0]; tj += disp[1]; } else tj += disp[0];
    if (ti < 0 || ti >= o->Ht || tj < 0 || tj >= o->Wt) return 0;
    *cost = cmp_float(o->func, o->fs + ((size_t)i * o->Ws + j) * o->nF, o->ft + ((size_t)ti * o->Wt + tj) * o->nF, o->nF);
    return 1;
}

static int od_setup(od_volume *o, int func, int search_dims, const float *img_s, int Hs, int Ws, const float *img_t, int Ht, int Wt, int C, int h_r,
                    int v_r, const int *range /* lower0, upper0, lower1, upper1 */) {
    if (!so_func_supported(func) || func_census(func) || (search_dims != 1 && search_dims != 2)) return 1;
    if (search_dims == 1 && Hs != Ht) return 1; /* patchmatch.h:587-591 */
    o->func = func; o->search_dims = search_dims; o->h_r = h_r; o->v_r = v_r;
    o->lower[0] = range[0]; o->upper[0] = range[1]; o->lower[1] = range[2]; o->upper[1] = range[3];
    o->Hs = Hs; o->Ws = Ws; o->Ht = Ht; o->Wt = Wt; o->C = C; o->nF = (2 * v_r + 1) * (2 * h_r + 1) * C;
    float *fs = (float *)malloc((size_t)Hs * Ws * o->nF * sizeof(float)), *ft = (float *)malloc((size_t)Ht * Wt * o->nF * sizeof(float));
    if (!fs || !ft) { free(fs); free(ft); return 2; }
    on_demand_features(func, img_s, Hs, Ws, C, h_r, v_r, fs);
    on_demand_features(func, img_t, Ht, Wt, C, h_r, v_r, ft);
    o->fs = fs; o->ft = ft;
    return 0;
}
static void od_free(od_volume *o) { free((void *)o->fs); free((void *)o->ft); }

/* the decorated features themselves, for the tests */
void so_on_demand_features(int func, const float *img, int H, int W, int C, int h_r, int v_r, float *out) { on_demand_features(func, img, H, W, C, h_r, v_r, out); }

/* CachelessOnDemandCostVolume::truncatedCostVolume(disp, radius, defaultVal), on_demand_cost_volume.h:474-596, AS WRITTEN: the
 * value handed to costValue as a disparity is `tap - radius + disp2idx(disparity)`, an INDEX (:513-514, :285-286), so the window
 * is centred on disparity - lowerOffset, which is the disparity only when the range starts at 0.  Entries without a value hold
 * defaultCvValForMatchFunc (matching_costs.h:706-713): FLT_MAX for costs, numeric_limits<float>::min() = FLT_MIN for scores.
 * range = {lower0, upper0, lower1, upper1}; for search_dims == 1 only lower1 / upper1 (columns) are used.
 * disp [H][W][search_dims] -> tcv [H][W][(2 radius + 1)^search_dims]. */
int so_on_demand_truncated_cv(int func, int search_dims, const float *img_s, int Hs, int Ws, const float *img_t, int Ht, int Wt, int C, int h_r,
                              int v_r, const int *range, const int32_t *disp, int radius, float *tcv) {
    od_volume o;
    int r2[4] = {range[0], range[1], range[2], range[3]};
    if (search_dims == 1) { r2[0] = range[2]; r2[1] = range[3]; }
    int rc = od_setup(&o, func, search_dims, img_s, Hs, Ws, img_t, Ht, Wt, C, h_r, v_r, r2);
    if (rc) return rc;
    const int T = 2 * radius + 1;
    const float def = so_func_strategy(func) == SO_COST ? FLT_MAX : FLT_MIN;
#pragma omp parallel for
    for (int i = 0; i < Hs; i++)
        for (int j = 0; j < Ws; j++) {
            const int32_t *d = disp + ((size_t)i * Ws + j) * search_dims;
            if (search_dims == 2) {
                for (int a = 0; a < T; a++)
                    for (int b = 0; b < T; b++) {
                        int arg[2] = {a - radius + (d[0] - o.lower[0]), b - radius + (d[1] - o.lower[1])};
                        float c;
                        tcv[(((size_t)i * Ws + j) * T + a) * T + b] = od_cost(&o, i, j, arg, &c) ? c : def;
                    }
            } else {
                for (int a = 0; a < T; a++) {
                    int arg[1] = {a - radius + (d[0] - o.lower[0])};
                    float c;
                    tcv[((size_t)i * Ws + j) * T + a] = od_cost(&o, i, j, arg, &c) ? c : def;
                }
            }
        }
    od_free(&o);
    return 0;
}

/* ---- cachelessPatchMatch, correlation/patchmatch.h:61-160 (init), :162-224 (test), :226-363 (search), :365-445 (propagate),
 * :447-493 (iterations), :560-621 (entry).  The reference seeds one std::default_random_engine per OpenMP thread from
 * std::random_device (:76-92, :243-258), so its output is not reproducible even by itself; here every draw is a pure function
 * of (seed, iteration, pixel, draw index, dimension) -- a counter-based stream in the role of the reference's NumbersCache
 * branch, mapped into the range exactly as that branch does: setValueInRange(v) = |v % range| + lower (correlation_base.h:377-384).
 * Everything else follows the reference statement by statement, including the std::optional comparisons of :203-211:
 * a candidate without a value is dropped; with a score function a candidate beats a current solution that has no value
 * (`v >= nullopt` is true), with a cost function it never does (`v <= nullopt` is false). */
static inline int32_t pm_random(uint64_t seed, uint32_t iter, uint32_t i, uint32_t j, uint32_t k, uint32_t dim) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (((uint64_t)iter << 40) ^ ((uint64_t)i << 20) ^ (uint64_t)j ^ ((uint64_t)k << 56) ^ ((uint64_t)dim << 60) ^ 0x632BE59BD9B4E019ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; /* splitmix64 finaliser */
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (int32_t)(uint32_t)(z >> 32);
}
int32_t so_pm_random(uint64_t seed, uint32_t iter, uint32_t i, uint32_t j, uint32_t k, uint32_t dim) { return pm_random(seed, iter, i, j, k, dim); }
static inline int32_t pm_in_range(int32_t v, int lower, int upper) {
    int range = upper - lower + 1;
    int m = v % range;
    return (m < 0 ? -m : m) + lower;
}

static int pm_test(const od_volume *o, int32_t *sol, int i, int j, const int *cand) { /* patchMatchTestCost, :162-224 */
    const int nd = o->search_dims;
    int32_t *cur = sol + ((size_t)i * o->Ws + j) * nd;
    int dc[2] = {cand[0], nd == 2 ? cand[1] : 0}, da[2] = {cur[0], nd == 2 ? cur[1] : 0};
    float c_new, c_old;
    if (!od_cost(o, i, j, dc, &c_new)) return 0;
    int has_old = od_cost(o, i, j, da, &c_old);
    int keep;
    if (so_func_strategy(o->func) == SO_SCORE) keep = has_old ? (c_new >= c_old) : 1;
    else keep = has_old ? (c_new <= c_old) : 0;
    if (keep) { cur[0] = cand[0]; if (nd == 2) cur[1] = cand[1]; }
    return keep;
}

/* patchMatchImpl, :447-493, on a set-up volume: initial solution (the draw, or `init`), then the iterations */
static int pm_run(const od_volume *op, int n_iter, int n_random, uint64_t seed, const int32_t *init, int32_t *sol) {
    const int nd = op->search_dims, Hs = op->Hs, Ws = op->Ws;
    for (int i = 0; i < Hs; i++) /* randomDispInit, the NumbersCache branch with a search offset: :110-118 */
        for (int j = 0; j < Ws; j++)
            for (int s = 0; s < nd; s++) {
                const size_t e = ((size_t)i * Ws + j) * nd + s;
                sol[e] = init ? init[e] : pm_in_range(pm_random(seed, 0xFFFFFFFFu, i, j, 0, s), op->lower[s], op->upper[s]); /* (init: the `initializer` callback's map, :598-605) */
            }
    int it = 0;
    for (; it < n_iter; it++) {
        long changes = 0;
        const int inc0 = (it % 4) < 2 ? 1 : -1, inc1 = (it % 2) == 0 ? 1 : -1; /* propagation_direction.h:64-86, patchmatch.h:462-479 */
        /* line scans, :387-410: rows are independent, a row is sequential */
#pragma omp parallel for reduction(+ : changes)
        for (int i = 0; i < Hs; i++)
            for (int j = inc1 > 0 ? 0 : Ws - 1; inc1 > 0 ? j < Ws : j > 0; j += inc1) { /* `j != final` with final = Ws or 0: the last column is skipped going left */
                int pj = j - inc1;
                if (pj < 0 || pj >= Ws) continue;
                const int32_t *pv = sol + ((size_t)i * Ws + pj) * nd;
                int cand[2] = {pv[0], nd == 2 ? pv[1] : 0};
                changes += pm_test(op, sol, i, j, cand);
            }
        /* column scans, :412-437 */
#pragma omp parallel for reduction(+ : changes)
        for (int j = 0; j < Ws; j++)
            for (int i = inc0 > 0 ? 0 : Hs - 1; inc0 > 0 ? i < Hs : i > 0; i += inc0) {
                int pi = i - inc0;
                if (pi < 0 || pi >= Hs) continue;
                const int32_t *pv = sol + ((size_t)pi * Ws + j) * nd;
                int cand[2] = {pv[0], nd == 2 ? pv[1] : 0};
                changes += pm_test(op, sol, i, j, cand);
            }
        /* random search, :226-363 */
#pragma omp parallel for reduction(+ : changes)
        for (int i = 0; i < Hs; i++)
            for (int j = 0; j < Ws; j++) {
                const int32_t *cur = sol + ((size_t)i * Ws + j) * nd;
                const int base_i = nd == 2 ? cur[0] : 0, base_j = nd == 2 ? cur[1] : cur[0];
                int n_chang = 0;
                for (int k = 0; k < n_random; k++) {
                    int disp_i = 0, disp_j;
                    if (nd == 1) disp_j = pm_in_range(pm_random(seed, it, i, j, k, 0), op->lower[0], op->upper[0]);
                    else {
                        disp_i = pm_in_range(pm_random(seed, it, i, j, k, 0), op->lower[0], op->upper[0]);
                        disp_j = pm_in_range(pm_random(seed, it, i, j, k, 1), op->lower[1], op->upper[1]);
                    }
                    int delta_i = disp_i - base_i, delta_j = disp_j - base_j; /* :320-331: exploration shrunk towards the solution */
                    delta_j *= k + 1;
                    delta_j /= n_random + 1;
                    if (nd == 2) { delta_i *= k + 1; delta_i /= n_random + 1; }
                    disp_i = base_i + delta_i;
                    disp_j = base_j + delta_j;
                    if (nd == 1) { if (disp_j == base_j) disp_j = base_j + 1; }
                    else if (disp_i == base_i && disp_j == base_j) { disp_i = base_i + 1; disp_j = base_j + 1; }
                    int cand[2] = {nd == 2 ? disp_i : disp_j, disp_j};
                    n_chang = pm_test(op, sol, i, j, cand); /* `=`, not `+=` (:345): only the last draw's outcome is counted */
                }
                changes += n_chang;
            }
        if (changes == 0) { it++; break; } /* :486-488 */
    }
    return it;
}

int so_cacheless_patch_match_init(int func, int search_dims, const float *img_s, int Hs, int Ws, const float *img_t, int Ht, int Wt, int C, int h_r,
                                  int v_r, const int *range, int n_iter, int n_random, uint64_t seed, const int32_t *init, int32_t *sol,
                                  int *iterations_run) {
    od_volume o;
    int r2[4] = {range[0], range[1], range[2], range[3]};
    if (search_dims == 1) { r2[0] = range[2]; r2[1] = range[3]; }
    int rc = od_setup(&o, func, search_dims, img_s, Hs, Ws, img_t, Ht, Wt, C, h_r, v_r, r2);
    if (rc) return rc;
    const int it = pm_run(&o, n_iter, n_random, seed, init, sol);
    if (iterations_run) *iterations_run = it;
    od_free(&o);
    return 0;
}

int so_cacheless_patch_match(int func, int search_dims, const float *img_s, int Hs, int Ws, const float *img_t, int Ht, int Wt, int C, int h_r,
                             int v_r, const int *range, int n_iter, int n_random, uint64_t seed, int32_t *sol, int *iterations_run) {
    return so_cacheless_patch_match_init(func, search_dims, img_s, Hs, Ws, img_t, Ht, Wt, C, h_r, v_r, range, n_iter, n_random, seed, NULL, sol, iterations_run);
}

/* patchMatch, correlation/patchmatch.h:496-558: the same iteration on feature volumes the caller built, through the cached cost volume
 * (on_demand_cost_volume.h:35-327: costValue :105-178 has the rules of the cacheless one -- range, target inside -- and caches what
 * featureComparison returns).  The vectors pass getFeatureVolumeForMatchFunc TWICE for zero-mean / normalised functions: in the entry
 * point (:522-523) and again in the cost volume's constructor (on_demand_cost_volume.h:62-67).  feat_s [Hs][Ws][F], feat_t [Ht][Wt][F]. */
int so_patch_match(int func, int search_dims, const float *feat_s, int Hs, int Ws, const float *feat_t, int Ht, int Wt, int F, const int *range,
                   int n_iter, int n_random, uint64_t seed, const int32_t *init, int32_t *sol, int *iterations_run) {
    if (!so_func_supported(func) || func_census(func) || (search_dims != 1 && search_dims != 2)) return 1;
    if (search_dims == 1 && Hs != Ht) return 1; /* :533-537 */
    od_volume o;
    o.func = func; o.search_dims = search_dims; o.h_r = 0; o.v_r = 0;
    if (search_dims == 1) { o.lower[0] = range[2]; o.upper[0] = range[3]; o.lower[1] = 0; o.upper[1] = 0; }
    else { o.lower[0] = range[0]; o.upper[0] = range[1]; o.lower[1] = range[2]; o.upper[1] = range[3]; }
    o.Hs = Hs; o.Ws = Ws; o.Ht = Ht; o.Wt = Wt; o.C = 1; o.nF = F;
    float *fs = (float *)malloc((size_t)Hs * Ws * F * sizeof(float)), *ft = (float *)malloc((size_t)Ht * Wt * F * sizeof(float));
    float *tmp = (float *)malloc((size_t)(Hs * Ws > Ht * Wt ? Hs * Ws : Ht * Wt) * F * sizeof(float));
    if (!fs || !ft || !tmp) { free(fs); free(ft); free(tmp); return 2; }
    const int twice = func_zero_mean(func) || func_normalized(func);
    so_feature_volume_for_match_func(func, feat_s, Hs, Ws, F, twice ? (void *)tmp : (void *)fs);
    if (twice) so_feature_volume_for_match_func(func, tmp, Hs, Ws, F, fs);
    so_feature_volume_for_match_func(func, feat_t, Ht, Wt, F, twice ? (void *)tmp : (void *)ft);
    if (twice) so_feature_volume_for_match_func(func, tmp, Ht, Wt, F, ft);
    free(tmp);
    o.fs = fs; o.ft = ft;
    const int it = pm_run(&o, n_iter, n_random, seed, init, sol);
    if (iterations_run) *iterations_run = it;
    od_free(&o);
    return 0;
}

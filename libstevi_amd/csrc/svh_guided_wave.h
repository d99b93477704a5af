// computeGuidedCV on grey images, a wave per 64 pixels of a row (svh_guided_wave_impl.h; one translation unit per window width so that the
// instantiations compile side by side) + what it shares with svh_hierarchical.hip.
#pragma once
#include <cstdint>

#include "svh_compare.h"
#include "svh_internal.h"

namespace svh {

struct GuideArgs {
    const int32_t *guide;
    int Hg, Wg, radius, dirSign;
    bool cost;
};

// hierarchical.h:106-150: bilinear upsampling of the integer guide (taps clamped as written), times two, rounded half away
// from zero; float operations in the reference's order
__device__ __forceinline__ int guided_base_disp(const int32_t *__restrict__ guide, int Hg, int Wg, int h, int w, int i, int j, int dirSign) {
    const float v_pos = (float)(i * (Hg - 1)) / (float)(h - 1);
    int v0 = (int)floorf(v_pos), v1 = (int)ceilf(v_pos);
    if (v0 == v1) v1 += 1;
    if (v1 == Hg) {
        v0 -= 1;
        v1 -= 1;
    }
    const float h_pos = (float)(j * (Wg - 1)) / (float)(w - 1);
    int h0 = (int)floorf(h_pos), h1 = (int)ceilf(h_pos);
    if (h0 == h1) h1 += 1;
    if (h1 == Wg) {
        h0 -= 1;
        h1 -= 1;
    }
    float interp = 0.0f;
    interp += (v_pos - (float)v0) * (h_pos - (float)h0) * (float)guide[(int64_t)v1 * Wg + h1];
    interp += ((float)v1 - v_pos) * (h_pos - (float)h0) * (float)guide[(int64_t)v0 * Wg + h1];
    interp += (v_pos - (float)v0) * ((float)h1 - h_pos) * (float)guide[(int64_t)v1 * Wg + h0];
    interp += ((float)v1 - v_pos) * ((float)h1 - h_pos) * (float)guide[(int64_t)v0 * Wg + h0];
    interp *= 2.0f;
    return dirSign * (int)roundf(interp);
}


// one pixel of the one-pass form, everything from global memory (the per-lane walk; also what a block of guided_shared_kernel falls back
// to when its pixels' guides point too far apart for the staged target features)
template <int CMP, bool ZM, bool NORM, int R>
__device__ __forceinline__ void guided_fused_px(const FeatImage &src, const FeatImage &tgt, const float *__restrict__ mean_s, const float *__restrict__ norm_s,
                                                const float *__restrict__ mean_t, const float *__restrict__ norm_t, int H, int Ws, int Wt, const GuideArgs &g,
                                                int32_t *__restrict__ disp, float *__restrict__ tcv, int64_t p, int i, int j, int d0) {
    constexpr int NC = 4 * R + 1, T = 2 * R + 1;
    const int C = src.C, h = 2 * src.h_r + 1, v = 2 * src.v_r + 1;
    const float ms = ZM ? mean_s[p] : 0.0f, ns = NORM ? norm_s[p] : 1.0f;
    float mt[NC], nt[NC], acc[NC];
    bool tin[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int jt = j + d0 + c - 2 * R;
        tin[c] = jt >= 0 && jt < Wt;
        const int64_t pt = (int64_t)i * Wt + (tin[c] ? jt : 0);
        mt[c] = (ZM && tin[c]) ? mean_t[pt] : 0.0f;
        nt[c] = (NORM && tin[c]) ? norm_t[pt] : 1.0f;
        acc[c] = 0.0f;
    }
    for (int k = 0; k < v; k++) {
        const int ii = i - src.v_r + k;
        const bool row_in = ii >= 0 && ii < H;
        for (int l = 0; l < h; l++) {
            const int jj = j - src.h_r + l;
            for (int ch = 0; ch < C; ch++) {
                float s = (row_in && jj >= 0 && jj < Ws) ? src.img[((int64_t)ii * Ws + jj) * C + ch] : 0.0f;
                if (ZM) s = s - ms;
                if (NORM) s = s / ns;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const int jc = jj + d0 + c - 2 * R;
                    float t = 0.0f;
                    if (tin[c]) {
                        t = (row_in && jc >= 0 && jc < Wt) ? tgt.img[((int64_t)ii * Wt + jc) * C + ch] : 0.0f;
                        if (ZM) t = t - mt[c];
                        if (NORM) t = t / nt[c];
                    }
                    if (CMP == CMP_DOT) {
                        acc[c] += s * t;
                    } else if (CMP == CMP_SSD) {
                        const float tmp = s - t;
                        acc[c] += tmp * tmp;
                    } else {
                        acc[c] += fabsf(s - t);
                    }
                }
            }
        }
    }
    float score = g.cost ? INFINITY : -INFINITY;
    int best = 0; // offset of the winner relative to d0
#pragma unroll
    for (int c = R; c <= 3 * R; c++) {
        if (g.cost ? (acc[c] < score) : (acc[c] > score)) {
            score = acc[c];
            best = c - 2 * R;
        }
    }
    disp[p] = g.dirSign * (d0 + best);
#pragma unroll
    for (int dd = 0; dd < T; dd++) {
        const int want = best + g.dirSign * (dd - R) + 2 * R;
        float val = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (c == want) val = acc[c];
        tcv[p * T + dd] = val;
    }
}

// false: not a case of this form (window wider than 7, search radius above 3, a window too tall for the staged rows, or a combination of
// comparison / zero mean / norm no matching function has) -- nothing was launched
bool launch_guided_wave(svh_context *ctx, int cmp, bool zm, bool nrm, FeatImage src, FeatImage tgt, const float *ms, const float *ns, const float *mt,
                        const float *nt, int H, int Ws, int Wt, const GuideArgs &g, int32_t *disp, float *tcv);
bool launch_guided_wave_h1(svh_context *ctx, int cmp, bool zm, bool nrm, FeatImage src, FeatImage tgt, const float *ms, const float *ns, const float *mt,
                           const float *nt, int H, int Ws, int Wt, const GuideArgs &g, int32_t *disp, float *tcv);
bool launch_guided_wave_h2(svh_context *ctx, int cmp, bool zm, bool nrm, FeatImage src, FeatImage tgt, const float *ms, const float *ns, const float *mt,
                           const float *nt, int H, int Ws, int Wt, const GuideArgs &g, int32_t *disp, float *tcv);
bool launch_guided_wave_h3(svh_context *ctx, int cmp, bool zm, bool nrm, FeatImage src, FeatImage tgt, const float *ms, const float *ns, const float *mt,
                           const float *nt, int H, int Ws, int Wt, const GuideArgs &g, int32_t *disp, float *tcv);

} // namespace svh

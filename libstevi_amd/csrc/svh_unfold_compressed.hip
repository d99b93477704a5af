// UnFoldCompressor features (second half): superpixel means over a window mask.
//
//   UnFoldCompressor(mask)                       correlation/unfold.h:47-121
//   unfold(compressor, img2D | img3D, padding)   correlation/unfold.h:346-471
//
// The mask is a small host-side parameter (7x7 / 9x9 in the reference's generators): the index list is built on the host
// exactly as the constructor does (positive labels, features in increasing label order, entries ordered by feature then
// row-major, weight = float(1. / pixel count), bounding box always containing the centre) and uploaded as a few hundred
// bytes.  One lane per output feature accumulates `weight * sample` over its entries in list order: the products are
// rounded before the add, as in the reference's `out += static_cast<T_O>(weight * value)`.
#include <algorithm>
#include <map>
#include <vector>

#include "svh_internal.h"

namespace svh {

namespace {

struct PixelIndex {
    int32_t v, h;
    float w;
};

struct Compressor {
    int n_features = 0, minH = 0, maxH = 0, minW = 0, maxW = 0;
    std::vector<PixelIndex> entries;   // ordered by feature
    std::vector<int32_t> first;        // entries of feature f: [first[f], first[f + 1])
    int height() const { return maxH - minH + 1; }
    int width() const { return maxW - minW + 1; }
};

Compressor build_compressor(const int32_t *mask, int mh, int mw) {
    Compressor c;
    const int v_off = mh / 2, h_off = mw / 2;
    std::map<int32_t, int> count;
    for (int i = 0; i < mh; i++)
        for (int j = 0; j < mw; j++) {
            const int32_t feat = mask[i * mw + j];
            if (feat <= 0) continue;
            c.minH = std::min(c.minH, i - v_off);
            c.maxH = std::max(c.maxH, i - v_off);
            c.minW = std::min(c.minW, j - h_off);
            c.maxW = std::max(c.maxW, j - h_off);
            count[feat]++;
        }
    c.n_features = (int)count.size();
    for (auto const &kv : count) { // std::map iterates in increasing label order (= the sorted feats of unfold.h:103)
        c.first.push_back((int32_t)c.entries.size());
        for (int i = 0; i < mh; i++)
            for (int j = 0; j < mw; j++)
                if (mask[i * mw + j] == kv.first) c.entries.push_back({i - v_off, j - h_off, (float)(1. / kv.second)});
    }
    c.first.push_back((int32_t)c.entries.size());
    return c;
}

void compressed_geometry(const Compressor &c, int H, int W, const int32_t pad[4], int *pl, int *pt, int *Ho, int *Wo) {
    const int left = -c.minW, top = -c.minH; // compressor.margins(), unfold.h:98
    *pl = pad ? pad[0] : left;
    *pt = pad ? pad[1] : top;
    const int pr = pad ? pad[2] : c.maxW, pb = pad ? pad[3] : c.maxH;
    *Ho = H - c.height() + *pt + pb + 1;
    *Wo = W - c.width() + *pl + pr + 1;
}

// out(i, j, in_c * nF + f) = sum over the entries of feature f, in order, of fl(weight * img(i + v + top - pad_top, j + h + left - pad_left, in_c))
// (a block row per output row, 32-bit index arithmetic inside the row: four 64-bit divisions per output element made this kernel 0.67 ms
// for two 1080p images and 17 features)
__global__ void __launch_bounds__(256) unfold_compressed_kernel(const float *__restrict__ img, int H, int W, int C, const PixelIndex *__restrict__ entries,
                                                                const int32_t *__restrict__ first, int nF, int di, int dj, int Ho, int Wo,
                                                                float *__restrict__ out) {
    const int F = C * nF, row_n = Wo * F;
    for (int i = blockIdx.y; i < Ho; i += gridDim.y) {
        float *orow = out + (int64_t)i * row_n;
        for (int e = blockIdx.x * 256 + threadIdx.x; e < row_n; e += gridDim.x * 256) {
            const int j = e / F, fo = e - j * F;
            const int in_c = fo / nF, f = fo - in_c * nF;
            float acc = 0.0f;
            for (int k = first[f]; k < first[f + 1]; k++) {
                const int in_i = i + entries[k].v + di, in_j = j + entries[k].h + dj;
                const float v = (in_i >= 0 && in_i < H && in_j >= 0 && in_j < W) ? img[((int64_t)in_i * W + in_j) * C + in_c] : 0.0f;
                acc += entries[k].w * v;
            }
            orow[e] = acc;
        }
    }
}

// Round 5: the same on an LDS tile.  The kernel above walks its feature's entries through two dependent global loads per term (the entry,
// then the sample): 0.34 - 0.55 ms per 1080p image for 17 features, which are 141 MB of output.  Here a block owns UC_TPX pixels of an
// output row; the rows of the image its windows cover are staged once (samples outside the image: 0, as above), the entries as (offset in the
// tile, weight) pairs next to them, and a thread per output element adds `weight * sample` over its feature's entries in list order --
// the same products, rounded before the add, in the same order.
constexpr int UC_TPX = 64;
__global__ void __launch_bounds__(256) unfold_compressed_tiled_kernel(const float *__restrict__ img, int H, int W, int C, const PixelIndex *__restrict__ entries,
                                                                      const int32_t *__restrict__ first, int nF, int n_entries, int di, int dj, int minH, int minW,
                                                                      int hh, int ww, int Ho, int Wo, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float uc_lds[];
    const int TW = (UC_TPX + ww - 1) * C; // floats per tile row
    float *tile = uc_lds, *wgt = tile + hh * TW;
    int *off = reinterpret_cast<int *>(wgt + n_entries), *fst = off + n_entries;
    const int i = blockIdx.y, j0 = blockIdx.x * UC_TPX, n_px = min(UC_TPX, Wo - j0);
    const int n_cols = (n_px + ww - 1) * C, n_tile = hh * n_cols;
    const int col0 = j0 + minW + dj; // image column of tile column 0
#pragma unroll 4
    for (int x = threadIdx.x; x < n_tile; x += 256) {
        const int r = x / n_cols, q = x - r * n_cols, px = q / C, ch = q - px * C;
        const int in_i = i + minH + r + di, in_j = col0 + px;
        const float got = img[((int64_t)min(max(in_i, 0), H - 1) * W + min(max(in_j, 0), W - 1)) * C + ch];
        tile[r * TW + q] = (in_i >= 0 && in_i < H && in_j >= 0 && in_j < W) ? got : 0.0f;
    }
    for (int k = threadIdx.x; k < n_entries; k += 256) {
        off[k] = (entries[k].v - minH) * TW + (entries[k].h - minW) * C;
        wgt[k] = entries[k].w;
    }
    for (int f = threadIdx.x; f <= nF; f += 256) fst[f] = first[f];
    __syncthreads();
    const int F = C * nF, n = n_px * F;
    float *orow = out + ((int64_t)i * Wo + j0) * F;
    // e = j F + fo is advanced by 256 at a time without a division per element (two of them were two thirds of this loop's instructions)
    const int dq = 256 / F, dr = 256 - dq * F;
    int j = (int)threadIdx.x / F, fo = (int)threadIdx.x - j * F;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int in_c = C == 1 ? 0 : fo / nF, f = fo - in_c * nF;
        const float *at = tile + j * C + in_c;
        float acc = 0.0f;
        for (int k = fst[f]; k < fst[f + 1]; k++) acc += wgt[k] * at[off[k]];
        orow[e] = acc;
        j += dq;
        fo += dr;
        if (fo >= F) {
            fo -= F;
            j++;
        }
    }
}

int check_mask(svh_context *ctx, const int32_t *mask, int mh, int mw) {
    if (!mask || mh < 1 || mw < 1 || mh > 255 || mw > 255) {
        if (ctx) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "mask must be a host array of 1..255 x 1..255 labels");
        return SVH_ERR_INVALID_ARGUMENT;
    }
    return SVH_OK;
}

} // namespace

} // namespace svh

using namespace svh;

extern "C" int svh_unfold_compressed_shape(const svh_array *img, const int32_t *mask, int mask_h, int mask_w, const int32_t pad[4], int64_t out_shape[3]) {
    if (!img || !out_shape || img->ndim < 2 || img->ndim > 3 || check_mask(nullptr, mask, mask_h, mask_w) != SVH_OK) return SVH_ERR_INVALID_ARGUMENT;
    const Compressor c = build_compressor(mask, mask_h, mask_w);
    int pl, pt, Ho, Wo;
    compressed_geometry(c, (int)img->shape[0], (int)img->shape[1], pad, &pl, &pt, &Ho, &Wo);
    out_shape[0] = Ho;
    out_shape[1] = Wo;
    out_shape[2] = (img->ndim == 3 ? img->shape[2] : 1) * c.n_features;
    return SVH_OK;
}

extern "C" int svh_unfold_compressed(svh_context *ctx, const svh_array *img, const int32_t *mask, int mask_h, int mask_w, const int32_t pad[4],
                                     svh_array *out) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate_image(ctx, img, "img", -1));
    SVH_TRY(validate(ctx, out, "out", SVH_F32, 3, 3));
    SVH_TRY(check_mask(ctx, mask, mask_h, mask_w));
    const int H = (int)img->shape[0], W = (int)img->shape[1], C = img->ndim == 3 ? (int)img->shape[2] : 1;
    const Compressor c = build_compressor(mask, mask_h, mask_w);
    if (c.n_features == 0) return fail(ctx, SVH_EMPTY_RESULT, "the mask holds no positive label");
    int pl, pt, Ho, Wo;
    compressed_geometry(c, H, W, pad, &pl, &pt, &Ho, &Wo);
    if (Ho <= 0 || Wo <= 0) return fail(ctx, SVH_EMPTY_RESULT, "unfold output is empty");
    const int F = C * c.n_features;
    if (out->shape[0] != Ho || out->shape[1] != Wo || out->shape[2] != F) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "out must have shape (%d,%d,%d)", Ho, Wo, F);
    Scratch scr(ctx);
    void *dimg;
    OutStage os;
    SVH_TRY(stage_image(ctx, scr, *img, &dimg));
    SVH_TRY(stage_out(ctx, scr, *out, &os));
    PixelIndex *d_entries = scr.get_n<PixelIndex>(c.entries.size());
    int32_t *d_first = scr.get_n<int32_t>(c.first.size());
    if (!d_entries || !d_first) return SVH_ERR_OUT_OF_MEMORY;
    // pageable sources: the copies have left the host buffers when the calls return, the vectors may go out of scope afterwards
    SVH_HIP_CHECK(ctx, hipMemcpyAsync(d_entries, c.entries.data(), c.entries.size() * sizeof(PixelIndex), hipMemcpyHostToDevice, ctx->stream));
    SVH_HIP_CHECK(ctx, hipMemcpyAsync(d_first, c.first.data(), c.first.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    if ((int64_t)Wo * F >= (1ll << 31)) return fail(ctx, SVH_ERR_UNSUPPORTED, "unfold: an output row of %lld floats", (long long)Wo * F);
    const size_t tile_bytes = ((size_t)c.height() * (UC_TPX + c.width() - 1) * C + 2 * c.entries.size() + c.first.size()) * sizeof(float);
    if (tile_bytes <= 56 * 1024 && Ho <= 65535) { // the rows a block's windows cover in LDS
        const dim3 tgrid(ceil_div(Wo, UC_TPX), Ho);
        SVH_LAUNCH(ctx, "unfold_compressed", unfold_compressed_tiled_kernel, tgrid, 256, tile_bytes, (const float *)dimg, H, W, C, d_entries, d_first, c.n_features,
                   (int)c.entries.size(), -c.minH - pt, -c.minW - pl, c.minH, c.minW, c.height(), c.width(), Ho, Wo, (float *)os.dptr);
    } else {
        const dim3 grid(std::min(ceil_div(Wo * F, 256), 65535), std::min(Ho, 65535));
        SVH_LAUNCH(ctx, "unfold_compressed", unfold_compressed_kernel, grid, 256, 0, (const float *)dimg, H, W, C, d_entries, d_first, c.n_features,
                   -c.minH - pt, -c.minW - pl, Ho, Wo, (float *)os.dptr);
    }
    SVH_CHECK_LAUNCH(ctx);
    return finish_out(ctx, os);
}

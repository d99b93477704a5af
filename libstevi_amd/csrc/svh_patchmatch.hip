// On-demand (cacheless) cost volumes and PatchMatch: SURVEY.md section 8(f) rank 1 -- what examples/stereo-match runs
// (main.cpp:166-210): ZNCC cachelessPatchMatch on on-demand zero-mean-normalised features, then, with --refine, the truncated
// on-demand cost volume around the result (fed to refineDisp2dCostInterpolation, svh_select_refine.hip).
//
//   OnDemandDecoratedFeaturesVolume<ZNFeaturesVolumeDecorator<ZM, N>, ...>     correlation/on_demand_features_volume.h:34-214
//   CachelessOnDemandCostVolume::costValue / truncatedCostVolume               correlation/on_demand_cost_volume.h:345-612
//   cachelessPatchMatch / patchMatchImpl / patchMatchPropagate / patchMatchSearch / patchMatchTestCost
//                                                                              correlation/patchmatch.h:61-621
//
// These are NOT the features of the dense path: window samples outside the image are clamped to the border instead of being
// zero, and the decorator divides by the feature count (mean = sum / nF, norm = sqrt(sum of squares / nF)).  "Cacheless" in the
// reference means every cost is recomputed from the two images; here the decorated feature vectors of both images are formed
// once per call (H W nF floats each, in HBM) and a cost is the reference's featureComparison of two of them -- same values,
// same order, without redoing the decoration for every candidate.
//
// PatchMatch follows the reference statement by statement: a propagation pass is a row sweep (rows independent, a row
// sequential: one lane per row) followed by a column sweep (one lane per column), then the random search (one lane per pixel);
// the std::optional comparisons of patchmatch.h:203-211 are reproduced (a candidate without a value is dropped; with a score
// function a candidate beats a current solution that has no value, with a cost function it never does); going left / up the
// sweeps stop before index 0 (`j != final`, propagation_direction.h:40-55).  The one thing that cannot be reproduced is the random
// stream: the reference seeds a std::default_random_engine per OpenMP thread from std::random_device (patchmatch.h:76-92,
// :243-258), so its own output differs from run to run.  Here every draw is a pure function of (seed, iteration, pixel, draw,
// dimension), mapped into the range as the reference's NumbersCache branch does: |v % range| + lower (correlation_base.h:377-384).
#include <algorithm>
#include <cfloat>

#include "svh_internal.h"

namespace svh {

namespace {

struct OdVolume {
    const float *fs, *ft; // decorated features (H, W, nF)
    int func, nd;         // matching function, number of search dimensions
    int lower[2], upper[2];
    int Hs, Ws, Ht, Wt, nF;
    bool score;
    // the target image itself and the decorator's two numbers per target pixel (mean, norm): a target feature is
    // (sample - mean) / norm of a window sample -- the random search forms it again from these instead of fetching it (pm_search_image_kernel)
    const float *timg = nullptr, *tstat = nullptr; // tstat: (Ht, Wt, 2)
    int C = 1, h_r = 0, v_r = 0;
    bool zm = false, nrm = false;
};

// A block takes 256 consecutive pixels: a thread per pixel walks its window once or twice for the mean and the norm (nested loops in the
// order of the feature index: no division per sample), then the block writes the 256 x nF decorated features as one flat, coalesced run --
// a thread per pixel storing its nF floats nF * 4 bytes apart, with three divisions per sample and pass, took 5.2 ms for two 1080p RGB
// images and 7x7 windows (2.4 GB at 0.47 TB/s).  Same operations in the same order.
constexpr int ODF_PX = 256;
__global__ void __launch_bounds__(256) on_demand_features_kernel(const float *__restrict__ img, int H, int W, int C, int h_r, int v_r, bool zm, bool nrm,
                                                                 float *__restrict__ out, float2 *__restrict__ stats) {
    __shared__ float s_mean[ODF_PX], s_norm[ODF_PX];
    const int64_t npx = (int64_t)H * W, p0 = (int64_t)blockIdx.x * ODF_PX;
    const int h = 2 * h_r + 1, v = 2 * v_r + 1, nF = v * h * C;
    const int n_px = (int)(npx - p0 < ODF_PX ? npx - p0 : ODF_PX);
    if ((int)threadIdx.x < n_px) {
        const int64_t p = p0 + threadIdx.x;
        const int j = (int)(p % W), i = (int)(p / W);
        float mean = 0.0f, norm = 1.0f;
        if (zm) { // :183-192
            for (int k = 0; k < v; k++) {
                const int ii = min(H - 1, max(0, i + k - v_r)); // constant border condition, features_volume.h:128-133
                for (int l = 0; l < h; l++) {
                    const float *px = img + ((int64_t)ii * W + min(W - 1, max(0, j + l - h_r))) * C;
                    for (int c = 0; c < C; c++) mean += px[c];
                }
            }
            mean /= (float)nF;
        }
        if (nrm) { // :194-207
            float acc = 0.0f;
            for (int k = 0; k < v; k++) {
                const int ii = min(H - 1, max(0, i + k - v_r));
                for (int l = 0; l < h; l++) {
                    const float *px = img + ((int64_t)ii * W + min(W - 1, max(0, j + l - h_r))) * C;
                    for (int c = 0; c < C; c++) {
                        float x = px[c];
                        if (zm) x -= mean;
                        acc += x * x;
                    }
                }
            }
            acc /= (float)nF;
            norm = sqrtf(acc);
        }
        s_mean[threadIdx.x] = mean;
        s_norm[threadIdx.x] = norm;
        if (stats) stats[p] = make_float2(mean, norm);
    }
    __syncthreads();
    // a wave per pixel, the lanes over the feature index (a pixel's nF floats are one contiguous run): which window sample a lane's
    // features are is the same for every pixel -- decoded once (three divisions per feature made this loop the whole kernel)
    const int hc = h * C, j_blk = (int)(p0 % W), i_blk = (int)(p0 / W);
    if (nF < 48) { // short vectors would leave most of a wave's lanes idle: one flat run of elements instead (a few divisions per element)
        float *o = out + p0 * nF;
        for (int e = threadIdx.x; e < n_px * nF; e += 256) {
            const int q = e / nF, f = e - q * nF;
            const int k = f / hc, r = f - k * hc, l = r / C, c = r - l * C;
            const int t = j_blk + q, i = i_blk + t / W, j = t - (t / W) * W;
            const int ii = min(H - 1, max(0, i + k - v_r)), jj = min(W - 1, max(0, j + l - h_r));
            float x = img[((int64_t)ii * W + jj) * C + c];
            if (zm) x -= s_mean[q];
            if (nrm) x /= s_norm[q];
            o[e] = x;
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int MAXCH = 4;
    int dk[MAXCH], dl[MAXCH], dc[MAXCH];
#pragma unroll
    for (int m = 0; m < MAXCH; m++) {
        const int f = min(lane + 64 * m, nF - 1);
        const int k = f / hc, r = f - k * hc, l = r / C;
        dk[m] = k - v_r;
        dl[m] = l - h_r;
        dc[m] = r - l * C;
    }
    for (int q = wave; q < n_px; q += 4) {
        const int t = j_blk + q, i = i_blk + t / W, j = t - (t / W) * W; // (wave uniform)
        const float mean = s_mean[q], norm = s_norm[q];
        float *o = out + (p0 + q) * nF;
#pragma unroll
        for (int m = 0; m < MAXCH; m++) {
            const int f = lane + 64 * m;
            if (f < nF) {
                const int ii = min(H - 1, max(0, i + dk[m])), jj = min(W - 1, max(0, j + dl[m]));
                float x = img[((int64_t)ii * W + jj) * C + dc[m]];
                if (zm) x -= mean;
                if (nrm) x /= norm;
                o[f] = x;
            }
        }
        for (int f = lane + 64 * MAXCH; f < nF; f += 64) { // (more than 256 features: decoded on the spot)
            const int k = f / hc, r = f - k * hc, l = r / C, c = r - l * C;
            const int ii = min(H - 1, max(0, i + k - v_r)), jj = min(W - 1, max(0, j + l - h_r));
            float x = img[((int64_t)ii * W + jj) * C + c];
            if (zm) x -= mean;
            if (nrm) x /= norm;
            o[f] = x;
        }
    }
}

// x / y through the double reciprocal rd = 1.0 / (double)y of a divisor many samples share (svh_guided_wave_impl.h, div_by_shared, has the
// argument: the float nearest to (double)x * rd is the correctly rounded quotient whenever that is 0 or a normal number): a fifth of a float
// division's cycles.  A nonzero denormal result takes the division itself.
__device__ __forceinline__ float div_by_shared_or_divide(float x, float y, double rd) {
    float q = (float)((double)x * rd);
    if (__builtin_amdgcn_classf(q, 0x090)) q = x / y;
    return q;
}

// Round 5: the same with the window rows of the block's pixels in LDS.  The kernel above walks every window through the caches twice (mean,
// norm: a dependent load per sample) and a third time for the features, and divides every sample by the norm: 3.1 ms for two 1080p RGB
// images and 7x7 windows, whose 2.4 GB of features take 0.4 ms to write.  Here a block owns up to 256 consecutive pixels of ONE image row;
// the v window rows of those pixels (256 + h - 1 pixels each, border pixels repeated: features_volume.h:128-133) are copied to LDS once --
// one contiguous run per row away from the image's left and right borders -- and the three walks read them there; the divisions go
// through the norm's double reciprocal.  Same operations on the same operands in the same order.
__global__ void __launch_bounds__(256) on_demand_features_tiled_kernel(const float *__restrict__ img, int H, int W, int C, int h_r, int v_r, bool zm, bool nrm,
                                                                       float *__restrict__ out, float2 *__restrict__ stats) {
    extern __shared__ float tile[]; // [v][(256 + h - 1) C]
    __shared__ float s_mean[ODF_PX], s_norm[ODF_PX];
    __shared__ double s_rd[ODF_PX];
    const int i = blockIdx.y, j0 = blockIdx.x * ODF_PX, n_px = min(ODF_PX, W - j0);
    const int h = 2 * h_r + 1, v = 2 * v_r + 1, hc = h * C, nF = v * hc, TWc = (ODF_PX + h - 1) * C;
    const int n_cols = (n_px + h - 1) * C; // what of a tile row this block reads
    const bool inside = j0 - h_r >= 0 && j0 + n_px - 1 + h_r < W; // (block uniform)
    // four window rows' loads before the first store (a row-by-row copy waits for memory once per window row)
    for (int r0 = 0; r0 < n_cols; r0 += 256) {
        const int r = r0 + threadIdx.x;
        const int x = r / C, c = r - x * C;
        const int64_t at = inside ? (int64_t)(j0 - h_r) * C + r : (int64_t)min(W - 1, max(0, j0 - h_r + x)) * C + c; // (clamped: the border pixels repeat)
        for (int k0 = 0; k0 < v; k0 += 4) {
            float got[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float *row = img + (int64_t)min(H - 1, max(0, i + min(k0 + q, v - 1) - v_r)) * W * C;
                got[q] = row[r < n_cols ? at : 0];
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (k0 + q < v && r < n_cols) tile[(k0 + q) * TWc + r] = got[q];
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < n_px) {
        const float *win = tile + threadIdx.x * C; // this pixel's window: rows k of hc consecutive floats (feature index order: k, l, c)
        float mean = 0.0f, norm = 1.0f;
        if (zm) { // :183-192
            for (int k = 0; k < v; k++) {
#pragma unroll 4
                for (int m = 0; m < hc; m++) mean += win[k * TWc + m];
            }
            mean /= (float)nF;
        }
        if (nrm) { // :194-207
            float acc = 0.0f;
            for (int k = 0; k < v; k++) {
#pragma unroll 4
                for (int m = 0; m < hc; m++) {
                    float x = win[k * TWc + m];
                    if (zm) x -= mean;
                    acc += x * x;
                }
            }
            acc /= (float)nF;
            norm = sqrtf(acc);
        }
        s_mean[threadIdx.x] = mean;
        s_norm[threadIdx.x] = norm;
        s_rd[threadIdx.x] = 1.0 / (double)norm;
        if (stats) stats[(int64_t)i * W + j0 + threadIdx.x] = make_float2(mean, norm);
    }
    __syncthreads();
    float *o_blk = out + ((int64_t)i * W + j0) * nF;
    if (nF < 48) { // short vectors would leave most of a wave's lanes idle: one flat run of elements instead (a few divisions per element)
        for (int e = threadIdx.x; e < n_px * nF; e += 256) {
            const int q = e / nF, f = e - q * nF;
            const int k = f / hc, m = f - k * hc;
            float x = tile[k * TWc + q * C + m];
            if (zm) x -= s_mean[q];
            if (nrm) x = div_by_shared_or_divide(x, s_norm[q], s_rd[q]);
            o_blk[e] = x;
        }
        return;
    }
    // a wave per pixel, the lanes over the feature index (a pixel's nF floats are one contiguous run): where in the tile a lane's features
    // sit relative to the pixel is the same for every pixel -- decoded once
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int MAXCH = 4;
    int at[MAXCH];
#pragma unroll
    for (int m = 0; m < MAXCH; m++) {
        const int f = min(lane + 64 * m, nF - 1);
        const int k = f / hc;
        at[m] = k * TWc + (f - k * hc);
    }
    for (int q = wave; q < n_px; q += 4) {
        const float mean = s_mean[q], norm = s_norm[q];
        const double rd = s_rd[q];
        const float *win = tile + q * C;
        float *o = o_blk + (int64_t)q * nF;
#pragma unroll
        for (int m = 0; m < MAXCH; m++) {
            const int f = lane + 64 * m;
            if (f < nF) {
                float x = win[at[m]];
                if (zm) x -= mean;
                if (nrm) x = div_by_shared_or_divide(x, norm, rd);
                o[f] = x;
            }
        }
        for (int f = lane + 64 * MAXCH; f < nF; f += 64) { // (more than 256 features: decoded on the spot)
            const int k = f / hc;
            float x = win[k * TWc + (f - k * hc)];
            if (zm) x -= mean;
            if (nrm) x = div_by_shared_or_divide(x, norm, rd);
            o[f] = x;
        }
    }
}

// costValue, on_demand_cost_volume.h:409-468.  disp[0] = rows, disp[1] = columns (flow); disp[0] = columns (stereo).
__device__ __forceinline__ bool od_cost(const OdVolume &o, int i, int j, int d0, int d1, float *cost) {
    int ti = i, tj = j;
    if (o.nd == 2) {
        if (d0 < o.lower[0] || d0 > o.upper[0] || d1 < o.lower[1] || d1 > o.upper[1]) return false;
        ti += d0;
        tj += d1;
    } else {
        if (d0 < o.lower[0] || d0 > o.upper[0]) return false;
        tj += d0;
    }
    if (ti < 0 || ti >= o.Ht || tj < 0 || tj >= o.Wt) return false;
    const float *s = o.fs + ((int64_t)i * o.Ws + j) * o.nF, *t = o.ft + ((int64_t)ti * o.Wt + tj) * o.nF;
    float acc = 0.0f;
    if (o.func == SVH_SSD || o.func == SVH_ZSSD) {
        for (int f = 0; f < o.nF; f++) {
            const float tmp = s[f] - t[f];
            acc += tmp * tmp;
        }
    } else if (o.func == SVH_SAD || o.func == SVH_ZSAD) {
        for (int f = 0; f < o.nF; f++) acc += fabsf(s[f] - t[f]);
    } else {
        for (int f = 0; f < o.nF; f++) acc += s[f] * t[f];
    }
    *cost = acc;
    return true;
}

// truncatedCostVolume as written (on_demand_cost_volume.h:474-596): the value handed to costValue as a disparity is
// tap - radius + disp2idx(disparity), i.e. the window is centred on disparity - lowerOffset
__global__ void od_truncated_kernel(OdVolume o, const int32_t *__restrict__ disp, int radius, float def, float *__restrict__ tcv) {
    const int T = 2 * radius + 1, TT = o.nd == 2 ? T * T : T;
    const int64_t n = (int64_t)o.Hs * o.Ws * TT;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int tap = (int)(e % TT);
        const int64_t p = e / TT;
        const int j = (int)(p % o.Ws), i = (int)(p / o.Ws);
        const int32_t *d = disp + p * o.nd;
        float c;
        bool ok;
        if (o.nd == 2) ok = od_cost(o, i, j, tap / T - radius + (d[0] - o.lower[0]), tap % T - radius + (d[1] - o.lower[1]), &c);
        else ok = od_cost(o, i, j, tap - radius + (d[0] - o.lower[0]), 0, &c);
        tcv[e] = ok ? c : def;
    }
}

// ---- PatchMatch ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t pm_random(uint64_t seed, uint32_t iter, uint32_t i, uint32_t j, uint32_t k, uint32_t dim) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (((uint64_t)iter << 40) ^ ((uint64_t)i << 20) ^ (uint64_t)j ^ ((uint64_t)k << 56) ^ ((uint64_t)dim << 60) ^ 0x632BE59BD9B4E019ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; // splitmix64 finaliser
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (int32_t)(uint32_t)(z >> 32);
}
__device__ __forceinline__ int32_t pm_in_range(int32_t v, int lower, int upper) { // setValueInRange, correlation_base.h:377-384
    const int range = upper - lower + 1;
    const int m = v % range;
    return (m < 0 ? -m : m) + lower;
}

// One wavefront evaluates one cost: the lanes fetch the two feature vectors coalesced (64 consecutive floats per access) and form
// the per-feature terms in parallel -- exactly the products / differences featureComparison forms -- park them in LDS, and the
// running sum is then taken in the reference's order (f = 0, 1, 2, ...) by every lane redundantly, so all lanes hold the same,
// bit-identical value.  (A lane per cost, as the truncated-volume kernel does, leaves the sweeps waiting on one dependent load
// after the other: 45 us per pixel step at 1080p RGB 7x7; this form takes well under one.)  Blocks are single waves.
__device__ __forceinline__ bool wave_cost(const OdVolume &o, int i, int j, int d0, int d1, float *buf, float *cost) {
    int ti = i, tj = j;
    if (o.nd == 2) {
        if (d0 < o.lower[0] || d0 > o.upper[0] || d1 < o.lower[1] || d1 > o.upper[1]) return false;
        ti += d0;
        tj += d1;
    } else {
        if (d0 < o.lower[0] || d0 > o.upper[0]) return false;
        tj += d0;
    }
    if (ti < 0 || ti >= o.Ht || tj < 0 || tj >= o.Wt) return false;
    const float *s = o.fs + ((int64_t)i * o.Ws + j) * o.nF, *t = o.ft + ((int64_t)ti * o.Wt + tj) * o.nF;
    const int lane = threadIdx.x;
    __syncthreads(); // the previous cost's readers are done with buf
    for (int f0 = 0; f0 < o.nF; f0 += 256) { // four rounds of 64 features per wait: eight loads in flight (a load, a wait, a store per round: three waits at 7x7 RGB)
        float a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            a[r] = b[r] = 0.0f;
            if (f0 + 64 * r < o.nF) { // (wave uniform: short vectors load one round)
                const int f = min(f0 + 64 * r + lane, o.nF - 1);
                a[r] = s[f];
                b[r] = t[f];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int f = f0 + 64 * r + lane;
            float term;
            if (o.func == SVH_SSD || o.func == SVH_ZSSD) {
                const float tmp = a[r] - b[r];
                term = tmp * tmp;
            } else if (o.func == SVH_SAD || o.func == SVH_ZSAD) {
                term = fabsf(a[r] - b[r]);
            } else {
                term = a[r] * b[r];
            }
            if (f < o.nF) buf[f] = term;
        }
    }
    __syncthreads();
    float acc = 0.0f;
    int f = 0;
    for (; f + 16 <= o.nF; f += 16) { // four 16-byte reads in flight per wait (one read, one wait, four additions per round was a third of an evaluation)
        const float4 q0 = *reinterpret_cast<const float4 *>(buf + f), q1 = *reinterpret_cast<const float4 *>(buf + f + 4);
        const float4 q2 = *reinterpret_cast<const float4 *>(buf + f + 8), q3 = *reinterpret_cast<const float4 *>(buf + f + 12);
        acc += q0.x; acc += q0.y; acc += q0.z; acc += q0.w;
        acc += q1.x; acc += q1.y; acc += q1.z; acc += q1.w;
        acc += q2.x; acc += q2.y; acc += q2.z; acc += q2.w;
        acc += q3.x; acc += q3.y; acc += q3.z; acc += q3.w;
    }
    for (; f + 4 <= o.nF; f += 4) {
        const float4 q = *reinterpret_cast<const float4 *>(buf + f);
        acc += q.x;
        acc += q.y;
        acc += q.z;
        acc += q.w;
    }
    for (; f < o.nF; f++) acc += buf[f];
    *cost = acc;
    return true;
}

// ---- 64 costs at once: lane e owns the pair (source pixel spx, target pixel tpx) -- the evaluation of pm_search_chunked_kernel below, shared
// with the sweeps' run batches.  The vectors are fetched coalesced, 8 pairs x 128 bytes per load instruction (lane = (pair of the group,
// 16-byte piece)), 32 features at a time, their per-feature terms parked in a 64 x PMC_PITCH float table in LDS, and lane e adds up row e in
// the reference's order (f = 0, 1, 2, ...); the nF % 4 trailing features are fetched by the pair's own lane.  Same products / differences and
// the same order as wave_cost: the same bits.  `has`: the lane's pair exists (its result is meaningless otherwise).  Single-wave blocks.
struct __attribute__((packed, aligned(4))) Feat4 { // four consecutive features; global_load_dwordx4 only needs 4-byte alignment
    float x, y, z, w;
};
constexpr int PMC_PITCH = 36; // (32 features per chunk) row pitch of the table in floats (16-byte rows, conflict-free 16-byte column reads)
__device__ __forceinline__ float pm_costs64(const OdVolume &o, int spx, int tpx, bool has, float *tab) {
    const int lane = threadIdx.x;
    const int cg = lane >> 3, piece = lane & 7; // candidate of the group, 16-byte piece of the chunk
    const int nF = o.nF, nq = nF >> 2, n_chunks = (nq + 7) >> 3;
    const int func = o.func;
    auto term_of = [&](float a, float b) {
        if (func == SVH_SSD || func == SVH_ZSSD) {
            const float tmp = a - b;
            return tmp * tmp;
        }
        if (func == SVH_SAD || func == SVH_ZSAD) return fabsf(a - b);
        return a * b;
    };
    const unsigned long long has_mask = __ballot(has);
    int spx_g[8], tpx_g[8]; // the pairs this lane fetches pieces of (one per group of eight)
#pragma unroll
    for (int g = 0; g < 8; g++) {
        spx_g[g] = __shfl(spx, 8 * g + cg);
        tpx_g[g] = __shfl(tpx, 8 * g + cg);
    }
    float c_new = 0.0f;
    for (int ch = 0; ch < n_chunks; ch++) {
        const int q = 8 * ch + piece; // this lane's 16-byte piece of every vector
        Feat4 sa[8], ta[8];
#pragma unroll
        for (int g = 0; g < 8; g++) {
            sa[g] = ta[g] = Feat4{0.f, 0.f, 0.f, 0.f};
            if (q < nq && ((has_mask >> (8 * g + cg)) & 1ull)) {
                sa[g] = *reinterpret_cast<const Feat4 *>(o.fs + (int64_t)spx_g[g] * nF + 4 * q);
                ta[g] = *reinterpret_cast<const Feat4 *>(o.ft + (int64_t)tpx_g[g] * nF + 4 * q);
            }
        }
        __syncthreads(); // the previous chunk's readers are done with the table
#pragma unroll
        for (int g = 0; g < 8; g++)
            *reinterpret_cast<float4 *>(tab + (8 * g + cg) * PMC_PITCH + 4 * piece) =
                make_float4(term_of(sa[g].x, ta[g].x), term_of(sa[g].y, ta[g].y), term_of(sa[g].z, ta[g].z), term_of(sa[g].w, ta[g].w));
        __syncthreads();
        const int quads = min(8, nq - 8 * ch); // (uniform)
        const float4 *row = reinterpret_cast<const float4 *>(tab + lane * PMC_PITCH);
        if (quads == 8) { // (every chunk but the last) the eight reads in flight together, then the 32 additions in order
            float4 t[8];
#pragma unroll
            for (int qq = 0; qq < 8; qq++) t[qq] = row[qq];
#pragma unroll
            for (int qq = 0; qq < 8; qq++) {
                c_new += t[qq].x;
                c_new += t[qq].y;
                c_new += t[qq].z;
                c_new += t[qq].w;
            }
        } else {
            for (int qq = 0; qq < quads; qq++) {
                const float4 t = row[qq];
                c_new += t.x;
                c_new += t.y;
                c_new += t.z;
                c_new += t.w;
            }
        }
    }
    if (has) { // the nF % 4 trailing features
        const float *sv = o.fs + (int64_t)spx * nF, *tv = o.ft + (int64_t)tpx * nF;
        for (int f = 4 * nq; f < nF; f++) c_new += term_of(sv[f], tv[f]);
    }
    __syncthreads(); // (the table may be rewritten by the caller's next use)
    return c_new;
}

// The cost of the current solution of every pixel is kept next to it (value + "has a value"), updated whenever the solution
// changes: the reference recomputes it for every test (it is "cacheless"), which gives the same number each time.
struct PmState {
    int32_t *sol;
    float *cost;
    uint8_t *valid;
};

// patchMatchTestCost, patchmatch.h:162-224, for the whole wave (uniform control flow); returns 1 when the candidate was kept
__device__ __forceinline__ int pm_test(const OdVolume &o, const PmState &st, int i, int j, int c0, int c1, float *buf) {
    const int64_t p = (int64_t)i * o.Ws + j;
    float c_new;
    if (!wave_cost(o, i, j, c0, c1, buf, &c_new)) return 0;
    const bool has_old = st.valid[p] != 0;
    const float c_old = st.cost[p];
    bool keep;
    if (o.score) keep = has_old ? (c_new >= c_old) : true;  // `v >= nullopt` is true
    else keep = has_old ? (c_new <= c_old) : false;         // `v <= nullopt` is false
    if (keep && threadIdx.x == 0) {
        st.sol[p * o.nd] = c0;
        if (o.nd == 2) st.sol[p * o.nd + 1] = c1;
        st.cost[p] = c_new;
        st.valid[p] = 1;
    }
    return keep ? 1 : 0;
}

// randomDispInit, :61-160 (NumbersCache branch), plus the cost of the initial solution; a wave per pixel
// init != nullptr: the caller's initial solution (the reference's `initializer` callback, patchmatch.h:538-545 / :598-605) instead of the draw
__global__ void __launch_bounds__(64) pm_init_kernel(OdVolume o, uint64_t seed, PmState st, const int32_t *__restrict__ init) {
    extern __shared__ __attribute__((aligned(16))) float pm_buf[];
    const int64_t npx = (int64_t)o.Hs * o.Ws;
    for (int64_t p = blockIdx.x; p < npx; p += gridDim.x) {
        const int j = (int)(p % o.Ws), i = (int)(p / o.Ws);
        int d[2] = {0, 0};
        for (int s = 0; s < o.nd; s++) d[s] = init ? init[p * o.nd + s] : pm_in_range(pm_random(seed, 0xFFFFFFFFu, i, j, 0, s), o.lower[s], o.upper[s]);
        float c = 0.0f;
        const bool ok = wave_cost(o, i, j, d[0], d[1], pm_buf, &c);
        if (threadIdx.x == 0) {
            for (int s = 0; s < o.nd; s++) st.sol[p * o.nd + s] = d[s];
            st.cost[p] = c;
            st.valid[p] = ok ? 1 : 0;
        }
    }
}

// the target pixel of candidate (d0, d1) at (i, j) as wave_cost finds it: false when the candidate leaves the search range or the image
__device__ __forceinline__ bool pm_target_of(const OdVolume &o, int i, int j, int d0, int d1, int *tpx) {
    int ti = i, tj = j;
    if (o.nd == 2) {
        if (d0 < o.lower[0] || d0 > o.upper[0] || d1 < o.lower[1] || d1 > o.upper[1]) return false;
        ti += d0;
        tj += d1;
    } else {
        if (d0 < o.lower[0] || d0 > o.upper[0]) return false;
        tj += d0;
    }
    if (ti < 0 || ti >= o.Ht || tj < 0 || tj >= o.Wt) return false;
    *tpx = ti * o.Wt + tj; // (pixels < 2^31: checked by the host)
    return true;
}

// The same 64 pixels at a time (round 5): a lane per pixel draws its solution, pm_costs64 evaluates the 64 costs together (coalesced
// 128-byte pieces, 32 features at a time through LDS: the evaluation of the random search) -- a wave per pixel took 1.57 ms at 1080p RGB
// for what one round of the search does in 0.17.  Same draws, same costs (pm_costs64 adds in wave_cost's order).
__global__ void __launch_bounds__(64) pm_init64_kernel(OdVolume o, uint64_t seed, PmState st, const int32_t *__restrict__ init) {
    __shared__ __attribute__((aligned(16))) float tab[64 * PMC_PITCH];
    const int64_t npx = (int64_t)o.Hs * o.Ws;
    const int lane = threadIdx.x;
    for (int64_t g0 = (int64_t)blockIdx.x * 64; g0 < npx; g0 += (int64_t)gridDim.x * 64) { // (wave uniform)
        const int64_t p = g0 + lane;
        const bool mine = p < npx;
        int d[2] = {0, 0}, tpx = 0;
        bool ok = false;
        if (mine) {
            const int j = (int)(p % o.Ws), i = (int)(p / o.Ws);
            for (int s = 0; s < o.nd; s++) d[s] = init ? init[p * o.nd + s] : pm_in_range(pm_random(seed, 0xFFFFFFFFu, i, j, 0, s), o.lower[s], o.upper[s]);
            ok = pm_target_of(o, i, j, d[0], d[1], &tpx);
        }
        const float c = pm_costs64(o, (int)min(p, npx - 1), tpx, ok, tab);
        if (mine) {
            for (int s = 0; s < o.nd; s++) st.sol[p * o.nd + s] = d[s];
            st.cost[p] = ok ? c : 0.0f;
            st.valid[p] = ok ? 1 : 0;
        }
    }
}

// row sweep of patchMatchPropagate (:387-410): a wave per row, the row is sequential; the candidate is the (possibly just
// updated) solution of the previous pixel, carried in registers
__global__ void __launch_bounds__(64) pm_rows_kernel(OdVolume o, int inc, PmState st, int *__restrict__ changes) {
    extern __shared__ __attribute__((aligned(16))) float pm_buf[];
    const int i = blockIdx.x;
    int n = 0;
    const int jfirst = inc > 0 ? 0 : o.Ws - 1;
    int64_t pp = (int64_t)i * o.Ws + jfirst;
    int p0 = st.sol[pp * o.nd], p1 = o.nd == 2 ? st.sol[pp * o.nd + 1] : 0; // solution of the sweep's first pixel (it has no predecessor)
    for (int j = jfirst + inc; inc > 0 ? j < o.Ws : j > 0; j += inc) { // `j != final`: going left stops before column 0
        const int kept = pm_test(o, st, i, j, p0, p1, pm_buf);
        n += kept;
        if (!kept) { // the next pixel's candidate is this pixel's solution
            const int64_t p = (int64_t)i * o.Ws + j;
            p0 = st.sol[p * o.nd];
            p1 = o.nd == 2 ? st.sol[p * o.nd + 1] : 0;
        }
    }
    if (n && threadIdx.x == 0) atomicAdd(changes, n);
}

// column sweep (:412-437): a wave per column
__global__ void __launch_bounds__(64) pm_cols_kernel(OdVolume o, int inc, PmState st, int *__restrict__ changes) {
    extern __shared__ __attribute__((aligned(16))) float pm_buf[];
    const int j = blockIdx.x;
    int n = 0;
    const int ifirst = inc > 0 ? 0 : o.Hs - 1;
    int64_t pp = (int64_t)ifirst * o.Ws + j;
    int p0 = st.sol[pp * o.nd], p1 = o.nd == 2 ? st.sol[pp * o.nd + 1] : 0;
    for (int i = ifirst + inc; inc > 0 ? i < o.Hs : i > 0; i += inc) {
        const int kept = pm_test(o, st, i, j, p0, p1, pm_buf);
        n += kept;
        if (!kept) {
            const int64_t p = (int64_t)i * o.Ws + j;
            p0 = st.sol[p * o.nd];
            p1 = o.nd == 2 ? st.sol[p * o.nd + 1] : 0;
        }
    }
    if (n && threadIdx.x == 0) atomicAdd(changes, n);
}

// ---- the sweeps without a cost evaluation on most steps (round 4f) -----------------------------------------------------------------------
// A pixel of a sweep is tested with the running candidate: the solution of its predecessor as the sweep left it.  When the predecessor
// KEPT its own solution -- most pixels, once the first iterations are over -- that is the solution the predecessor had before the sweep
// started, and the cost of the pixel against it depends on nothing the sweep does: pm_pred_cost_kernel evaluates it for every pixel in
// parallel beforehand (the same wave_cost: the same number).  The sweep then carries 64 pixels' worth of (that cost, the pixel's own
// cost, both "has a value" flags, the pixel's own solution) in the lanes of a few registers, loaded 64 pixels at a time, and a step whose
// predecessor kept its solution is a handful of v_readlane and a compare; only a step behind an ACCEPTED candidate (the running
// candidate travels on) evaluates a cost on the spot, as every step used to (1.1 - 1.8 us each: the candidate decides which target
// vector is loaded, so nothing of it can be fetched ahead).  Same tests in the same order on the same values: same result.
// axis 0: row sweep (predecessor (i, j - inc)), 1: column sweep ((i - inc, j))
// Two launches: a thread per pixel settles the pixels whose predecessor holds the pixel's own solution (the cost is the pixel's own: most
// pixels once regions agree) and lists the others; a wave per listed pixel evaluates its cost.  (A wave per pixel for both: 75 us per
// pre-pass at 640x480 whatever the state of the solution -- two 64-bit divisions and three dependent loads per pixel before anything else.)
// LOOK-BACK DEPTHS (round 5).  The candidate a step tests is the solution some pixel m steps back had BEFORE the sweep: that pixel kept its
// own solution (so the candidate became fresh there) and the m - 1 pixels between accepted it.  The pre-pass above is m = 1.  With the costs
// against the solutions 2 .. PM_DEPTH steps back evaluated beforehand as well (same kernels, `m` steps instead of one: only pixels whose own
// solution differs are evaluated, i.e. nothing in a region that agrees), a travelling candidate needs an evaluation on the spot only from
// its PM_DEPTH-th accepted step on -- and in the lines that never settle (image borders the true match leaves: the lines every sweep waits
// for once the rest has converged) runs are short: a step accepts with probability about one half.
constexpr int PM_DEPTH = 4;
__global__ void __launch_bounds__(256) pm_pred_classify_kernel(OdVolume o, int axis, int inc, PmState st, float *__restrict__ pcost, uint8_t *__restrict__ pvalid,
                                                               int32_t *__restrict__ work, int *__restrict__ n_work, int m = 1) {
    const int i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= o.Ws) return;
    const int64_t p = (int64_t)i * o.Ws + j;
    const int pi = axis ? i - m * inc : i, pj = axis ? j : j - m * inc;
    if (pi < 0 || pi >= o.Hs || pj < 0 || pj >= o.Ws) { // (no predecessor: never tested)
        pcost[p] = 0.0f;
        pvalid[p] = 0;
        return;
    }
    const int64_t pp = (int64_t)pi * o.Ws + pj;
    const int c0 = st.sol[pp * o.nd], c1 = o.nd == 2 ? st.sol[pp * o.nd + 1] : 0;
    if (c0 == st.sol[p * o.nd] && (o.nd < 2 || c1 == st.sol[p * o.nd + 1])) { // the pixel's own solution: its own cost (the same evaluation)
        pcost[p] = st.cost[p];
        pvalid[p] = st.valid[p];
    } else {
        work[atomicAdd(n_work, 1)] = (int32_t)p; // (pixels < 2^31: checked by the host)
    }
}

__global__ void __launch_bounds__(64) pm_pred_cost_kernel(OdVolume o, int axis, int inc, PmState st, float *__restrict__ pcost, uint8_t *__restrict__ pvalid,
                                                          const int32_t *__restrict__ work, const int *__restrict__ n_work, int m = 1) {
    extern __shared__ __attribute__((aligned(16))) float pm_buf[];
    const int n = *n_work;
    for (int w = blockIdx.x; w < n; w += gridDim.x) {
        const int p = work[w];
        const int j = p % o.Ws, i = p / o.Ws;
        const int64_t pp = (int64_t)(axis ? i - m * inc : i) * o.Ws + (axis ? j : j - m * inc);
        float c = 0.0f;
        const bool ok = wave_cost(o, i, j, st.sol[pp * o.nd], o.nd == 2 ? st.sol[pp * o.nd + 1] : 0, pm_buf, &c);
        if (threadIdx.x == 0) {
            pcost[p] = c;
            pvalid[p] = ok ? 1 : 0;
        }
    }
}

// The same for all look-back depths in one pair of launches (round 5: eight launches of 10 - 22 us per sweep were 5.2 ms of the 1080p chain):
// a thread per pixel settles, for m = 1 .. depths, the pixels whose m-th predecessor holds the pixel's own solution and lists the other
// (pixel, m) pairs -- the depth in the three bits above the pixel index (fewer than 2^29 pixels: checked by the host) --, a wave per listed
// pair evaluates its cost.  pcost / pvalid: [depth - 1][pixel].
__global__ void __launch_bounds__(256) pm_pred_classify_depths_kernel(OdVolume o, int axis, int inc, PmState st, float *__restrict__ pcost,
                                                                      uint8_t *__restrict__ pvalid, int64_t npx, uint32_t *__restrict__ work,
                                                                      int *__restrict__ n_work, int depths) {
    const int i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= o.Ws) return;
    const int64_t p = (int64_t)i * o.Ws + j;
    const int s0 = st.sol[p * o.nd], s1 = o.nd == 2 ? st.sol[p * o.nd + 1] : 0;
    const float own_cost = st.cost[p];
    const uint8_t own_valid = st.valid[p];
    for (int m = 1; m <= depths; m++) {
        float *pc = pcost + (int64_t)(m - 1) * npx;
        uint8_t *pv = pvalid + (int64_t)(m - 1) * npx;
        const int pi = axis ? i - m * inc : i, pj = axis ? j : j - m * inc;
        if (pi < 0 || pi >= o.Hs || pj < 0 || pj >= o.Ws) { // (no such predecessor: never tested)
            pc[p] = 0.0f;
            pv[p] = 0;
            continue;
        }
        const int64_t pp = (int64_t)pi * o.Ws + pj;
        const int c0 = st.sol[pp * o.nd], c1 = o.nd == 2 ? st.sol[pp * o.nd + 1] : 0;
        if (c0 == s0 && (o.nd < 2 || c1 == s1)) { // the pixel's own solution: its own cost (the same evaluation)
            pc[p] = own_cost;
            pv[p] = own_valid;
        } else {
            work[atomicAdd(n_work, 1)] = (uint32_t)p | (uint32_t)(m - 1) << 29;
        }
    }
}

__global__ void __launch_bounds__(64) pm_pred_cost_depths_kernel(OdVolume o, int axis, int inc, PmState st, float *__restrict__ pcost, uint8_t *__restrict__ pvalid,
                                                                 int64_t npx, const uint32_t *__restrict__ work, const int *__restrict__ n_work) {
    extern __shared__ __attribute__((aligned(16))) float pm_buf[];
    const int n = *n_work;
    for (int w = blockIdx.x; w < n; w += gridDim.x) {
        const int p = (int)(work[w] & 0x1fffffffu), m = (int)(work[w] >> 29) + 1;
        const int j = p % o.Ws, i = p / o.Ws;
        const int64_t pp = (int64_t)(axis ? i - m * inc : i) * o.Ws + (axis ? j : j - m * inc);
        float c = 0.0f;
        const bool ok = wave_cost(o, i, j, st.sol[pp * o.nd], o.nd == 2 ? st.sol[pp * o.nd + 1] : 0, pm_buf, &c);
        if (threadIdx.x == 0) {
            pcost[(int64_t)(m - 1) * npx + p] = c;
            pvalid[(int64_t)(m - 1) * npx + p] = ok ? 1 : 0;
        }
    }
}

// 64 listed pairs per wave through pm_costs64 (the first iteration lists every pixel: 1.3 + 0.7 ms a wave per pair)
__global__ void __launch_bounds__(64) pm_pred_cost64_kernel(OdVolume o, int axis, int inc, PmState st, float *__restrict__ pcost, uint8_t *__restrict__ pvalid,
                                                            int64_t npx, const uint32_t *__restrict__ work, const int *__restrict__ n_work) {
    __shared__ __attribute__((aligned(16))) float tab[64 * PMC_PITCH];
    const int n = *n_work, lane = threadIdx.x;
    for (int w0 = blockIdx.x * 64; w0 < n; w0 += gridDim.x * 64) { // (wave uniform)
        const bool mine = w0 + lane < n;
        const uint32_t e = mine ? work[w0 + lane] : 0u;
        const int p = (int)(e & 0x1fffffffu), m = (int)(e >> 29) + 1;
        const int j = p % o.Ws, i = p / o.Ws;
        int tpx = 0;
        bool ok = false;
        if (mine) {
            const int64_t pp = (int64_t)(axis ? i - m * inc : i) * o.Ws + (axis ? j : j - m * inc);
            ok = pm_target_of(o, i, j, st.sol[pp * o.nd], o.nd == 2 ? st.sol[pp * o.nd + 1] : 0, &tpx);
        }
        const float c = pm_costs64(o, p, tpx, ok, tab);
        if (mine) {
            pcost[(int64_t)(m - 1) * npx + p] = ok ? c : 0.0f;
            pvalid[(int64_t)(m - 1) * npx + p] = ok ? 1 : 0;
        }
    }
}

__device__ __forceinline__ int pm_lane_i(int v, int k) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(k)); }
__device__ __forceinline__ float pm_lane_f(float v, int k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), __builtin_amdgcn_readfirstlane(k))); }

// one sweep line: `len` pixels at p = first + t * step (t = 0 is the line's first pixel, which is not tested); (i, j) of pixel t: (i0 + t di, j0 + t dj)
// RUN BATCHES (round 5).  What is left to evaluate on the spot are the steps behind an accepted candidate -- and while a candidate travels
// it does not change, so the costs of the NEXT pixels of the line against it depend on nothing the sweep decides.  An evaluation is one
// memory round trip plus nF dependent additions whatever it fetches, so the sweep fetches the vectors of the next PM_RUN pixels with it
// (every load of the batch in flight together), parks the terms in LDS rows, and lanes 0 .. PM_RUN - 1 add up one row each in the reference's
// order: up to PM_RUN costs in the time of one.  The following steps read theirs from a lane until a pixel rejects the candidate (which
// drops what is left of the batch) or the batch is used up.  At 1080p RGB the first iteration evaluates 2.7 M of its 4.1 M sweep steps on
// the spot, in runs of 31 pixels on average (the planted flow of tools/bench_patchmatch.py crossing the image); later iterations are as
// slow as their slowest line, a border row whose pixels never agree: half of its steps evaluate, in runs of one or two, and a batch costs
// about twice a single evaluation -- so the host asks for batches in the FIRST iteration only (rows 5.2 -> 3.0 ms, columns 2.6 -> 1.9;
// in the later ones they made rows 1.7 -> 3.2 ms).  (A first form waited for three accepted steps and then evaluated 64 pixels through
// pm_costs64 -- five dependent round trips: runs that ended early made it 2.4 x SLOWER.)  Same costs, same tests, same order.
constexpr int PM_RUN = 8;
constexpr int PM_SCAN_EVALS = 12; // evaluations a scanned chunk makes before the step-by-step walk takes the rest of it
// costs of pixels t .. t + n - 1 of a line (n <= PM_RUN) against the candidate (p0, p1): lane e < n returns the cost of pixel t + e, bit e of
// *has_mask says whether it has a value (wave_cost's rules).  rows: n rows of `pitch` floats in LDS.
__device__ __forceinline__ float pm_run_costs(const OdVolume &o, int64_t first, int64_t step, int i0, int j0, int di, int dj, int t, int n, int p0, int p1,
                                              float *rows, int pitch, unsigned *has_mask) {
    const int lane = threadIdx.x, nF = o.nF, func = o.func;
    auto term_of = [&](float a, float b) {
        if (func == SVH_SSD || func == SVH_ZSSD) {
            const float tmp = a - b;
            return tmp * tmp;
        }
        if (func == SVH_SAD || func == SVH_ZSAD) return fabsf(a - b);
        return a * b;
    };
    bool cand_ok;
    if (o.nd == 2) cand_ok = p0 >= o.lower[0] && p0 <= o.upper[0] && p1 >= o.lower[1] && p1 <= o.upper[1];
    else cand_ok = p0 >= o.lower[0] && p0 <= o.upper[0];
    const float *sv[PM_RUN], *tv[PM_RUN]; // (wave uniform)
    unsigned mask = 0;
#pragma unroll
    for (int e = 0; e < PM_RUN; e++) {
        const int tt = t + e, ii = i0 + tt * di, jj = j0 + tt * dj;
        const int ti = o.nd == 2 ? ii + p0 : ii, tj = o.nd == 2 ? jj + p1 : jj + p0;
        const bool has = e < n && cand_ok && ti >= 0 && ti < o.Ht && tj >= 0 && tj < o.Wt;
        mask |= has ? 1u << e : 0u;
        sv[e] = o.fs + (first + (int64_t)(e < n ? tt : t) * step) * nF;
        tv[e] = o.ft + ((int64_t)(has ? ti : 0) * o.Wt + (has ? tj : 0)) * nF;
    }
    __syncthreads(); // the previous batch's readers are done with the rows
    for (int f0 = 0; f0 < nF; f0 += 192) { // three rounds of 64 features per wait: 3 x 2 x PM_RUN loads in flight (7x7 RGB, 147 features: one wait instead of two)
        float a[3][PM_RUN], b[3][PM_RUN];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int e = 0; e < PM_RUN; e++) {
                const int f = f0 + 64 * r + lane;
                a[r][e] = b[r][e] = 0.0f;
                if (f0 + 64 * r < nF && f < nF && ((mask >> e) & 1u)) {
                    a[r][e] = sv[e][f];
                    b[r][e] = tv[e][f];
                }
            }
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int e = 0; e < PM_RUN; e++) {
                const int f = f0 + 64 * r + lane;
                if (f < nF) rows[e * pitch + f] = term_of(a[r][e], b[r][e]);
            }
    }
    __syncthreads();
    float acc = 0.0f;
    if (lane < PM_RUN) { // the ordered sum of row `lane` (wave_cost: acc = 0; acc += term(f) for f = 0, 1, 2, ...)
        const float *row = rows + lane * pitch;
        int f = 0;
        for (; f + 16 <= nF; f += 16) { // (four reads in flight per wait)
            const float4 q0 = *reinterpret_cast<const float4 *>(row + f), q1 = *reinterpret_cast<const float4 *>(row + f + 4);
            const float4 q2 = *reinterpret_cast<const float4 *>(row + f + 8), q3 = *reinterpret_cast<const float4 *>(row + f + 12);
            acc += q0.x; acc += q0.y; acc += q0.z; acc += q0.w;
            acc += q1.x; acc += q1.y; acc += q1.z; acc += q1.w;
            acc += q2.x; acc += q2.y; acc += q2.z; acc += q2.w;
            acc += q3.x; acc += q3.y; acc += q3.z; acc += q3.w;
        }
        for (; f + 4 <= nF; f += 4) {
            const float4 q = *reinterpret_cast<const float4 *>(row + f);
            acc += q.x;
            acc += q.y;
            acc += q.z;
            acc += q.w;
        }
        for (; f < nF; f++) acc += row[f];
    }
    *has_mask = mask;
    return acc;
}

__device__ __forceinline__ int pm_sweep_line(const OdVolume &o, const PmState &st, const float *__restrict__ pcost, const uint8_t *__restrict__ pvalid,
                                             int64_t first, int64_t step, int i0, int j0, int di, int dj, int t_end, float *buf, int run_pitch, int depths,
                                             int64_t npx, bool scan) {
    const int lane = threadIdx.x;
    int n = 0;
    int age = 1;           // the running candidate is the pre-sweep solution of the pixel `age` steps back (1: fresh)
    float r_pd[PM_DEPTH - 1] = {}; // costs against the solutions 2 .. PM_DEPTH steps back, this chunk's pixels (their "has a value" bits: r_flags bits 2 ..)
    int b_t0 = 0, b_n = 0; // the batch in hand: pixels b_t0 .. b_t0 + b_n - 1 of the line against the travelling candidate
    float b_cost = 0.0f;
    unsigned b_has = 0u;
    int p0 = st.sol[first * o.nd], p1 = o.nd == 2 ? st.sol[first * o.nd + 1] : 0; // solution of the line's first pixel (it has no predecessor)
    bool fresh = true; // the running candidate is the predecessor's solution of before the sweep
    float r_pc = 0.0f, r_oc = 0.0f;
    int r_flags = 0, r_s0 = 0, r_s1 = 0;
    // What the steps of a chunk decide is written once per chunk, by the lanes of the pixels that changed: a step used to have lane 0 store
    // the solution, the cost and the flag of its pixel -- four one-lane stores into the cache line the previous step had just written, 0.46 us
    // per step in lines that walk step by step.  (A pixel's entries are read by the chunk load of ITS chunk and by later kernels only.)
    int w_t0 = 1, w_s0 = 0, w_s1 = 0;
    float w_c = 0.0f;
    bool w_set = false;
    auto flush = [&]() {
        if (w_set) {
            const int64_t p = first + (int64_t)(w_t0 + lane) * step;
            st.sol[p * o.nd] = w_s0;
            if (o.nd == 2) st.sol[p * o.nd + 1] = w_s1;
            st.cost[p] = w_c;
            st.valid[p] = 1;
        }
        w_set = false;
    };
    int prev_s0 = p0, prev_s1 = p1; // the pre-sweep solutions of the chunk before (lane 63: the pixel just before this chunk)
    bool batches_on = !scan; // run batches: the whole first iteration; with scanned chunks only where the walk takes over a chunk that keeps evaluating
    for (int t = 1; t < t_end; t++) {
        const int k = (t - 1) & 63;
        if (k == 0) { // the next 64 pixels of the line, one per lane (their entries are only written at their own steps, which come later)
            flush();
            w_t0 = t;
            if (t > 1) {
                prev_s0 = r_s0;
                prev_s1 = r_s1;
            }
            if (scan) batches_on = false;
            const int tt = t + lane;
            const bool in = tt < t_end;
            const int64_t p = first + (int64_t)(in ? tt : t) * step;
            r_pc = pcost[p];
            r_oc = st.cost[p];
            r_flags = (pvalid[p] ? 1 : 0) | (st.valid[p] ? 2 : 0);
#pragma unroll
            for (int m = 2; m <= PM_DEPTH; m++)
                if (m <= depths) {
                    r_pd[m - 2] = pcost[(int64_t)(m - 1) * npx + p];
                    r_flags |= pvalid[(int64_t)(m - 1) * npx + p] ? 1 << m : 0;
                }
            r_s0 = st.sol[p * o.nd];
            r_s1 = o.nd == 2 ? st.sol[p * o.nd + 1] : 0;
            // A region that agrees already: all 64 pixels hold the running candidate as their own solution, with a value that is no NaN.  Every
            // one of their steps then tests the pixel's own cost against itself (fresh: the pre-pass copied that very cost; travelling: the
            // branch below takes it) -- `>=` / `<=` on equal numbers: kept, nothing changes but the count, and the candidate travels on.  One
            // ballot instead of 64 steps: most chunks of most lines once the first iterations are over (1080p RGB: 0.46 us per step,
            // 14 of the 32 ms the sweeps took, profiles/r05l_pm_trace*.txt).
            if (t + 63 < t_end) {
                const bool same = r_s0 == p0 && (o.nd < 2 || r_s1 == p1) && (r_flags & 3) == 3 && r_oc == r_oc && (!fresh || lane > 0 || r_pc == r_oc);
                if (__ballot(same) == ~0ull) {
                    n += 64;
                    fresh = false;
                    age = PM_DEPTH + 1; // (further back than the pre-pass looked: irrelevant while the pixels hold the candidate themselves)
                    t += 63; // (the loop's increment takes the 64th)
                    continue;
                }
            }
        }
        if (k == 0 && scan) {
            // SCANNED CHUNK (round 5).  What a step decides is a function of the state it receives -- how many steps back the travelling
            // candidate was picked up: 1 .. depths (the pre-pass evaluated those), or further (`B`) -- and of numbers the chunk load already
            // holds: for each received state a lane knows whether its pixel keeps the candidate (-> state + 1) or not (-> 1, its own
            // solution travels on).  Those per-lane tables compose associatively, so a prefix scan over the 64 lanes gives every pixel
            // the state it receives in six shuffles instead of 64 dependent steps (0.27 us each on a wave that has its SIMD to itself; the
            // slowest line of a sweep -- an image border whose pixels never agree -- walked all of its 1 918 steps: 0.52 of its 0.72 ms).
            // A candidate further back than the pre-pass looked is known to be kept when it equals the pixel's own solution; the table
            // assumes that, the scan's result is checked, and the first lane where it does not hold gets its cost evaluated on the spot
            // (wave_cost, as the step-by-step walk would) and a constant table; the lanes before it are final, the scan is repeated.
            const int D = depths, B = depths + 1, chunk = min(64, t_end - t);
            const bool in = lane < chunk;
            auto keep_rule = [&](float cn, bool has) { // patchMatchTestCost, patchmatch.h:162-224
                if (!has) return false;
                const bool has_old = (r_flags & 2) != 0;
                return o.score ? (has_old ? cn >= r_oc : true) : (has_old ? cn <= r_oc : false);
            };
            constexpr uint32_t IDENT = 076543210u; // entry a (three bits at 3 a): a
            auto compose = [](uint32_t first_t, uint32_t then_t) { // (then o first)[a] = then[first[a]]
                uint32_t r = 0;
#pragma unroll
                for (int a = 1; a <= PM_DEPTH + 1; a++) r |= ((then_t >> (3 * ((first_t >> (3 * a)) & 7u))) & 7u) << (3 * a);
                return r;
            };
            uint32_t T = (uint32_t)(keep_rule(r_pc, (r_flags & 1) != 0) ? min(2, B) : 1) << 3;
#pragma unroll
            for (int a = 2; a <= PM_DEPTH; a++) {
                if (a <= D) {
                    int c0 = __shfl_up(r_s0, a), c1 = o.nd == 2 ? __shfl_up(r_s1, a) : 0;
                    const int q0 = __shfl(prev_s0, (64 + lane - a) & 63), q1 = o.nd == 2 ? __shfl(prev_s1, (64 + lane - a) & 63) : 0;
                    if (lane < a) {
                        c0 = q0;
                        c1 = q1;
                    }
                    const bool eq = c0 == r_s0 && (o.nd < 2 || c1 == r_s1);
                    const bool ka = keep_rule(eq ? r_oc : r_pd[a - 2], eq ? (r_flags & 2) != 0 : ((r_flags >> a) & 1) != 0);
                    T |= (uint32_t)(ka ? min(a + 1, B) : 1) << (3 * a);
                }
            }
            T |= (uint32_t)(keep_rule(r_oc, (r_flags & 2) != 0) ? B : 1) << (3 * B);
            if (!in) T = IDENT;
            const int a0 = age <= D ? age : B;
            bool fixed = false, kept_l = false, eqc = false;
            float fix_cost = 0.0f;
            int a_in = 1, cand0 = 0, cand1 = 0, evaluated = 0, stop_at = -1;
            for (;;) {
                uint32_t I = T;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t before = __shfl_up(I, off);
                    if (lane >= off) I = compose(before, I);
                }
                uint32_t E = __shfl_up(I, 1);
                if (lane == 0) E = IDENT;
                a_in = (int)((E >> (3 * a0)) & 7u);
                kept_l = in && ((T >> (3 * a_in)) & 7u) != 1u;
                // the candidate a lane receives: the own solution of the last lane before it that did not keep its candidate, or what the chunk received
                const unsigned long long resets = __ballot(in && !kept_l), below = resets & ((1ull << lane) - 1ull);
                const int s_lane = below ? 63 - __clzll((long long)below) : 0;
                const int g0 = __shfl(r_s0, s_lane), g1 = o.nd == 2 ? __shfl(r_s1, s_lane) : 0;
                cand0 = below ? g0 : p0;
                cand1 = below ? g1 : p1;
                eqc = cand0 == r_s0 && (o.nd < 2 || cand1 == r_s1);
                const unsigned long long need = __ballot(in && !fixed && a_in == B && !eqc);
                if (need == 0) break;
                const int f = __ffsll((long long)need) - 1; // (wave uniform) everything before lane f is final
                float cn = 0.0f;
                const bool has = wave_cost(o, i0 + (t + f) * di, j0 + (t + f) * dj, pm_lane_i(cand0, f), pm_lane_i(cand1, f), buf, &cn);
                const bool kept_f = pm_lane_i(keep_rule(cn, has) ? 1 : 0, f) != 0;
                if (lane == f) {
                    fixed = true;
                    fix_cost = cn;
                    T = (kept_f ? (uint32_t)B : 1u) * 011111110u; // every entry: where this pixel sends the candidate
                }
                if (++evaluated > PM_SCAN_EVALS && f + 1 < chunk) {
                    // a chunk that keeps evaluating (a candidate crossing a region that disagrees with it: every step an evaluation) is
                    // cheaper step by step than a scan per evaluation: what is decided up to lane f is written, the walk takes over behind it
                    if (lane == f) kept_l = kept_f;
                    stop_at = f;
                    batches_on = true; // (such a chunk evaluates in runs: the next pixels against the same candidate with it)
                    break;
                }
            }
            if (stop_at >= 0 && lane > stop_at) kept_l = false;
            if (kept_l) {
                float cn = r_oc; // (a candidate that equals the pixel's own solution: the pixel's own cost)
                if (fixed) cn = fix_cost;
                else if (a_in == 1) cn = r_pc;
                else if (!eqc) {
                    cn = r_pd[0];
#pragma unroll
                    for (int m = 3; m <= PM_DEPTH; m++) cn = a_in == m ? r_pd[m - 2] : cn;
                }
                const int64_t p = first + (int64_t)(t + lane) * step;
                st.sol[p * o.nd] = cand0;
                if (o.nd == 2) st.sol[p * o.nd + 1] = cand1;
                st.cost[p] = cn;
                st.valid[p] = 1;
            }
            n += __popcll(__ballot(kept_l));
            // what the next pixel receives: from the chunk's last pixel (or from the lane the walk takes over behind)
            const int last = stop_at >= 0 ? stop_at : chunk - 1;
            const bool kept_last = pm_lane_i(kept_l ? 1 : 0, last) != 0;
            if (kept_last) {
                p0 = pm_lane_i(cand0, last);
                p1 = pm_lane_i(cand1, last);
                age = min(pm_lane_i(a_in, last) + 1, B);
                fresh = false;
            } else {
                p0 = pm_lane_i(r_s0, last);
                p1 = pm_lane_i(r_s1, last);
                age = 1;
                fresh = true;
            }
            b_n = 0;
            t += last; // (the loop's increment takes the last)
            continue;
        }
        const int i = i0 + t * di, j = j0 + t * dj;
        const int flags = pm_lane_i(r_flags, k);
        float c_new;
        bool has_new;
        if (fresh) {
            c_new = pm_lane_f(r_pc, k);
            has_new = (flags & 1) != 0;
        } else if (p0 == pm_lane_i(r_s0, k) && (o.nd < 2 || p1 == pm_lane_i(r_s1, k))) {
            // a travelling candidate that equals the pixel's own solution (a region that agrees already): the cost is the pixel's own cost,
            // the same evaluation of the same two vectors
            c_new = pm_lane_f(r_oc, k);
            has_new = (flags & 2) != 0;
        } else if (age <= depths) { // evaluated by the pre-pass: the candidate is the pre-sweep solution of the pixel `age` steps back
            float v = r_pd[0];
#pragma unroll
            for (int m = 3; m <= PM_DEPTH; m++) v = age == m ? r_pd[m - 2] : v;
            c_new = pm_lane_f(v, k);
            has_new = ((flags >> age) & 1) != 0;
        } else if (run_pitch > 0 && batches_on) {
            if (t < b_t0 || t >= b_t0 + b_n) { // this pixel and the next ones of the line against the travelling candidate
                b_t0 = t;
                b_n = min(PM_RUN, t_end - t);
                b_cost = pm_run_costs(o, first, step, i0, j0, di, dj, t, b_n, p0, p1, buf, run_pitch, &b_has);
            }
            c_new = pm_lane_f(b_cost, t - b_t0);
            has_new = ((b_has >> (t - b_t0)) & 1u) != 0;
        } else {
            has_new = wave_cost(o, i, j, p0, p1, buf, &c_new);
        }
        int kept = 0;
        if (has_new) { // patchMatchTestCost, patchmatch.h:162-224 (a candidate without a value is dropped)
            const bool has_old = (flags & 2) != 0;
            const float c_old = pm_lane_f(r_oc, k);
            bool keep;
            if (o.score) keep = has_old ? (c_new >= c_old) : true;
            else keep = has_old ? (c_new <= c_old) : false;
            if (keep) {
                if (lane == k) { // (written with the rest of the chunk: flush)
                    w_s0 = p0;
                    w_s1 = p1;
                    w_c = c_new;
                    w_set = true;
                }
                kept = 1;
            }
        }
        n += kept;
        if (!kept) { // the next pixel's candidate is this pixel's own (unchanged) solution
            p0 = pm_lane_i(r_s0, k);
            p1 = pm_lane_i(r_s1, k);
            fresh = true;
            age = 1;
            b_n = 0; // (what is left of a batch was for the candidate that just ended)
        } else {
            fresh = false;
            age++;
        }
    }
    flush();
    return n;
}

__global__ void __launch_bounds__(64) pm_rows_fast_kernel(OdVolume o, int inc, PmState st, const float *__restrict__ pcost, const uint8_t *__restrict__ pvalid,
                                                          int *__restrict__ changes, int run_pitch, int depths, int scan) {
    extern __shared__ __attribute__((aligned(16))) float pm_buf[];
    const int i = blockIdx.x;
    const int jfirst = inc > 0 ? 0 : o.Ws - 1;
    // going right: pixels 1 .. Ws - 1; going left: Ws - 2 .. 1 (`j != final` stops before column 0)
    const int t_end = inc > 0 ? o.Ws : o.Ws - 1;
    const int n = pm_sweep_line(o, st, pcost, pvalid, (int64_t)i * o.Ws + jfirst, inc, i, jfirst, 0, inc, t_end, pm_buf, run_pitch, depths, (int64_t)o.Hs * o.Ws, scan != 0);
    if (n && threadIdx.x == 0) atomicAdd(changes, n);
}

__global__ void __launch_bounds__(64) pm_cols_fast_kernel(OdVolume o, int inc, PmState st, const float *__restrict__ pcost, const uint8_t *__restrict__ pvalid,
                                                          int *__restrict__ changes, int run_pitch, int depths, int scan) {
    extern __shared__ __attribute__((aligned(16))) float pm_buf[];
    const int j = blockIdx.x;
    const int ifirst = inc > 0 ? 0 : o.Hs - 1;
    const int t_end = inc > 0 ? o.Hs : o.Hs - 1;
    const int n = pm_sweep_line(o, st, pcost, pvalid, (int64_t)ifirst * o.Ws + j, (int64_t)inc * o.Ws, ifirst, j, inc, 0, t_end, pm_buf, run_pitch, depths, (int64_t)o.Hs * o.Ws, scan != 0);
    if (n && threadIdx.x == 0) atomicAdd(changes, n);
}

// patchMatchSearch, :226-363: a wave per pixel
__global__ void __launch_bounds__(64) pm_search_kernel(OdVolume o, uint64_t seed, uint32_t iter, int n_random, PmState st, int *__restrict__ changes) {
    extern __shared__ __attribute__((aligned(16))) float pm_buf[];
    const int64_t npx = (int64_t)o.Hs * o.Ws;
    int total = 0;
    for (int64_t p = blockIdx.x; p < npx; p += gridDim.x) {
        const int j = (int)(p % o.Ws), i = (int)(p / o.Ws);
        const int base_i = o.nd == 2 ? st.sol[p * o.nd] : 0, base_j = o.nd == 2 ? st.sol[p * o.nd + 1] : st.sol[p * o.nd];
        int n_chang = 0;
        for (int k = 0; k < n_random; k++) {
            int disp_i = 0, disp_j;
            if (o.nd == 1) {
                disp_j = pm_in_range(pm_random(seed, iter, i, j, k, 0), o.lower[0], o.upper[0]);
            } else {
                disp_i = pm_in_range(pm_random(seed, iter, i, j, k, 0), o.lower[0], o.upper[0]);
                disp_j = pm_in_range(pm_random(seed, iter, i, j, k, 1), o.lower[1], o.upper[1]);
            }
            int delta_i = disp_i - base_i, delta_j = disp_j - base_j; // :320-331: exploration shrunk towards the current solution
            delta_j *= k + 1;
            delta_j /= n_random + 1;
            if (o.nd == 2) {
                delta_i *= k + 1;
                delta_i /= n_random + 1;
            }
            disp_i = base_i + delta_i;
            disp_j = base_j + delta_j;
            if (o.nd == 1) {
                if (disp_j == base_j) disp_j = base_j + 1;
            } else if (disp_i == base_i && disp_j == base_j) {
                disp_i = base_i + 1;
                disp_j = base_j + 1;
            }
            n_chang = pm_test(o, st, i, j, o.nd == 2 ? disp_i : disp_j, disp_j, pm_buf); // `=`, not `+=` (:345)
        }
        total += n_chang;
    }
    if (total && threadIdx.x == 0) atomicAdd(changes, total);
}

// patchMatchSearch for up to 64 candidates per wave at a time: the n_random candidates of a pixel are all derived from the
// solution the pixel had when its loop started (patchmatch.h:287-340), so their costs do not depend on one another -- only the
// accept / reject decisions do.  A wave takes P = E / n_random pixels, lane e owns candidate e % n_random of pixel e / n_random:
//   0. every lane draws its candidate and checks that it has a value;
//   A. candidate by candidate, the 64 lanes fetch the two feature vectors coalesced and write the per-feature terms to row e of
//      an LDS table (row pitch odd: the column walk of phase B is conflict free);
//   B. lane e takes the running sum of row e in the reference's order: 64 ordered sums side by side instead of one sum done 64
//      times over;
//   C. the first lane of every pixel replays the reference's loop over its candidates with those costs (same comparisons, same
//      order, `n_chang =` keeping only the last outcome).
// Bit-identical to the wave-per-cost form, 64x fewer additions.
__global__ void __launch_bounds__(64) pm_search_batched_kernel(OdVolume o, uint64_t seed, uint32_t iter, int n_random, int E, int pitch, PmState st,
                                                              int *__restrict__ changes) {
    extern __shared__ __attribute__((aligned(16))) float pm_tab[];
    __shared__ int64_t pm_src[64], pm_tgt[64]; // per candidate: offsets of the two feature vectors (-1: no value)
    const int64_t npx = (int64_t)o.Hs * o.Ws;
    const int lane = threadIdx.x, P = E / n_random;
    int total = 0;
    for (int64_t g0 = (int64_t)blockIdx.x * P; g0 < npx; g0 += (int64_t)gridDim.x * P) {
        // phase 0
        const int slot = lane / n_random, k = lane - slot * n_random;
        const int64_t p = g0 + slot;
        const bool mine = lane < P * n_random && p < npx;
        int c0 = 0, c1 = 0, ti = 0, tj = 0;
        bool has = false;
        if (mine) {
            const int j = (int)(p % o.Ws), i = (int)(p / o.Ws);
            const int base_i = o.nd == 2 ? st.sol[p * o.nd] : 0, base_j = o.nd == 2 ? st.sol[p * o.nd + 1] : st.sol[p * o.nd];
            int disp_i = 0, disp_j;
            if (o.nd == 1) {
                disp_j = pm_in_range(pm_random(seed, iter, i, j, k, 0), o.lower[0], o.upper[0]);
            } else {
                disp_i = pm_in_range(pm_random(seed, iter, i, j, k, 0), o.lower[0], o.upper[0]);
                disp_j = pm_in_range(pm_random(seed, iter, i, j, k, 1), o.lower[1], o.upper[1]);
            }
            int delta_i = disp_i - base_i, delta_j = disp_j - base_j;
            delta_j *= k + 1;
            delta_j /= n_random + 1;
            if (o.nd == 2) {
                delta_i *= k + 1;
                delta_i /= n_random + 1;
            }
            disp_i = base_i + delta_i;
            disp_j = base_j + delta_j;
            if (o.nd == 1) {
                if (disp_j == base_j) disp_j = base_j + 1;
            } else if (disp_i == base_i && disp_j == base_j) {
                disp_i = base_i + 1;
                disp_j = base_j + 1;
            }
            c0 = o.nd == 2 ? disp_i : disp_j;
            c1 = disp_j;
            ti = i;
            tj = j;
            if (o.nd == 2) {
                has = c0 >= o.lower[0] && c0 <= o.upper[0] && c1 >= o.lower[1] && c1 <= o.upper[1];
                ti += c0;
                tj += c1;
            } else {
                has = c0 >= o.lower[0] && c0 <= o.upper[0];
                tj += c0;
            }
            has = has && ti >= 0 && ti < o.Ht && tj >= 0 && tj < o.Wt;
        }
        // phase A: the E x nF terms as one flat loop (independent loads, several in flight per lane)
        __syncthreads(); // the previous round's readers are done with the table and the descriptors
        if (lane < E) {
            pm_src[lane] = (g0 + slot) * o.nF;
            pm_tgt[lane] = has ? ((int64_t)ti * o.Wt + tj) * o.nF : -1;
        }
        __syncthreads();
        const int n_cand = P * n_random, nq = o.nF >> 2;
        auto term_of = [&](float a, float b) {
            if (o.func == SVH_SSD || o.func == SVH_ZSSD) {
                const float tmp = a - b;
                return tmp * tmp;
            }
            if (o.func == SVH_SAD || o.func == SVH_ZSAD) return fabsf(a - b);
            return a * b;
        };
        if (o.nF < 128) {
            // short vectors: one flat loop over the E x nF terms (independent loads, four in flight per lane)
            const int n_terms = n_cand * o.nF;
#pragma unroll 4
            for (int idx = lane; idx < n_terms; idx += 64) {
                const int e = idx / o.nF, f = idx - e * o.nF;
                const int64_t tb = pm_tgt[e];
                if (tb >= 0) pm_tab[e * pitch + f] = term_of(o.fs[pm_src[e] + f], o.ft[tb + f]);
            }
        } else {
            // long vectors: eight candidates at a time, four features per lane and load (16 bytes, 4-byte aligned), so that 16
            // loads of up to 1 KiB per wave are in flight together -- the targets of the random search are scattered over HBM
            // and the kernel lives on memory-level parallelism
            for (int e0 = 0; e0 < n_cand; e0 += 8) {
                for (int q0 = 0; q0 < nq; q0 += 64) {
                    const int q = q0 + lane;
                    Feat4 sa[8], ta[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int e = e0 + u;
                        const int64_t tb = e < n_cand ? pm_tgt[e] : -1;
                        sa[u] = ta[u] = Feat4{0.f, 0.f, 0.f, 0.f};
                        if (tb >= 0 && q < nq) {
                            sa[u] = *reinterpret_cast<const Feat4 *>(o.fs + pm_src[e] + 4 * q);
                            ta[u] = *reinterpret_cast<const Feat4 *>(o.ft + tb + 4 * q);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int e = e0 + u;
                        if (e < n_cand && q < nq && pm_tgt[e] >= 0) {
                            float *row = pm_tab + e * pitch + 4 * q;
                            row[0] = term_of(sa[u].x, ta[u].x);
                            row[1] = term_of(sa[u].y, ta[u].y);
                            row[2] = term_of(sa[u].z, ta[u].z);
                            row[3] = term_of(sa[u].w, ta[u].w);
                        }
                    }
                }
            }
            for (int idx = lane; idx < n_cand * (o.nF & 3); idx += 64) { // the nF % 4 trailing features of every candidate
                const int e = idx / (o.nF & 3), f = 4 * nq + idx % (o.nF & 3);
                const int64_t tb = pm_tgt[e];
                if (tb >= 0) pm_tab[e * pitch + f] = term_of(o.fs[pm_src[e] + f], o.ft[tb + f]);
            }
        }
        __syncthreads();
        // phase B
        float c_new = 0.0f;
        if (has) {
            const float *row = pm_tab + (int64_t)lane * pitch;
            for (int f = 0; f < o.nF; f++) c_new += row[f];
        }
        // phase C: every lane of a pixel replays the loop (the shuffles need all lanes), its first lane writes the outcome
        {
            const int leader = lane - k;
            bool has_old = mine ? st.valid[p] != 0 : false;
            float c_old = mine ? st.cost[p] : 0.0f;
            int s0 = 0, s1 = 0, n_chang = 0;
            bool changed = false;
            for (int q = 0; q < n_random; q++) {
                const int srcl = min(leader + q, 63);
                const int hq = __shfl((int)has, srcl);
                const float cq = __shfl(c_new, srcl);
                const int q0 = __shfl(c0, srcl), q1 = __shfl(c1, srcl);
                if (!hq) { // no value: patchMatchTestCost returns 0 (:197-199)
                    n_chang = 0;
                    continue;
                }
                bool keep;
                if (o.score) keep = has_old ? (cq >= c_old) : true;
                else keep = has_old ? (cq <= c_old) : false;
                if (keep) {
                    s0 = q0;
                    s1 = q1;
                    c_old = cq;
                    has_old = true;
                    changed = true;
                }
                n_chang = keep ? 1 : 0;
            }
            if (mine && k == 0) {
                if (changed) {
                    st.sol[p * o.nd] = s0;
                    if (o.nd == 2) st.sol[p * o.nd + 1] = s1;
                    st.cost[p] = c_old;
                    st.valid[p] = 1;
                }
                total += n_chang;
            }
        }
    }
    // every pixel's first lane holds a partial count
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off);
    if (total && lane == 0) atomicAdd(changes, total);
}

// patchMatchSearch, 64 candidates per wave, the vectors in CHUNKS of 32 features (round 5; the default).  What the counters said about the
// batched kernel above at 1080p RGB 7x7 (profiles/r05i_pm_search_pmc.jsonl): per round of 24 candidates a wave issues 1,180 vector and
// 2,260 scalar instructions -- flat loops over (candidate, feature) with a division per term or a guard per load -- at four waves per
// SIMD; the vector ALUs are active a tenth of the time, the memory system delivers 1 TB/s.  Same idea (coalesced fetches of the
// candidates' vectors, a table of terms in LDS, lane e adds up row e in the reference's order), restructured so that a round is a few
// hundred instructions and a CU holds many waves:
//   * a round takes 64 candidates (lane e owns candidate e: the 64 / n_random pixels of the round times their n_random candidates);
//   * the features go through LDS 32 at a time: a 64 x 36 float table (9 KB per wave instead of one row of nF floats per candidate);
//   * a load instruction fetches 8 candidates x 128 bytes (lane = (candidate of the group, 16-byte piece)): 8 + 8 loads per chunk, all
//     issued before the first term is formed; the source / target pixel of every candidate sits in registers of every lane (eight
//     groups x one shuffle each, once per round);
//   * lane e then adds its 32 terms to its running sum (eight 16-byte LDS reads), f = 0, 1, 2, ... as the reference does.
// The last nF % 4 features of each vector are fetched by the candidate's own lane.  Same products / differences, same order: same bits.
__global__ void __launch_bounds__(64) pm_search_chunked_kernel(OdVolume o, uint64_t seed, uint32_t iter, int n_random, PmState st, int *__restrict__ changes) {
    __shared__ __attribute__((aligned(16))) float tab[64 * PMC_PITCH];
    const int64_t npx = (int64_t)o.Hs * o.Ws;
    const int lane = threadIdx.x, P = 64 / n_random;
    const int slot = lane / n_random, k = lane - slot * n_random;
    int total = 0;
    for (int64_t g0 = (int64_t)blockIdx.x * P; g0 < npx; g0 += (int64_t)gridDim.x * P) { // (wave uniform)
        // ---- this lane's candidate (patchmatch.h:287-340)
        const int64_t p = g0 + slot;
        const bool mine = lane < P * n_random && p < npx;
        int c0 = 0, c1 = 0, tpx = 0;
        bool has = false;
        if (mine) {
            const int j = (int)(p % o.Ws), i = (int)(p / o.Ws);
            const int base_i = o.nd == 2 ? st.sol[p * o.nd] : 0, base_j = o.nd == 2 ? st.sol[p * o.nd + 1] : st.sol[p * o.nd];
            int disp_i = 0, disp_j;
            if (o.nd == 1) {
                disp_j = pm_in_range(pm_random(seed, iter, i, j, k, 0), o.lower[0], o.upper[0]);
            } else {
                disp_i = pm_in_range(pm_random(seed, iter, i, j, k, 0), o.lower[0], o.upper[0]);
                disp_j = pm_in_range(pm_random(seed, iter, i, j, k, 1), o.lower[1], o.upper[1]);
            }
            int delta_i = disp_i - base_i, delta_j = disp_j - base_j; // :320-331: exploration shrunk towards the current solution
            delta_j *= k + 1;
            delta_j /= n_random + 1;
            if (o.nd == 2) {
                delta_i *= k + 1;
                delta_i /= n_random + 1;
            }
            disp_i = base_i + delta_i;
            disp_j = base_j + delta_j;
            if (o.nd == 1) {
                if (disp_j == base_j) disp_j = base_j + 1;
            } else if (disp_i == base_i && disp_j == base_j) {
                disp_i = base_i + 1;
                disp_j = base_j + 1;
            }
            c0 = o.nd == 2 ? disp_i : disp_j;
            c1 = disp_j;
            int ti = i, tj = j;
            if (o.nd == 2) {
                has = c0 >= o.lower[0] && c0 <= o.upper[0] && c1 >= o.lower[1] && c1 <= o.upper[1];
                ti += c0;
                tj += c1;
            } else {
                has = c0 >= o.lower[0] && c0 <= o.upper[0];
                tj += c0;
            }
            has = has && ti >= 0 && ti < o.Ht && tj >= 0 && tj < o.Wt;
            tpx = has ? ti * o.Wt + tj : 0; // (pixels < 2^31: checked by the host)
        }
        const float c_new = pm_costs64(o, (int)min(p, npx - 1), tpx, has, tab);
        // ---- every lane of a pixel replays the reference's loop over the pixel's candidates, its first lane writes the outcome
        {
            const int leader = lane - k;
            bool has_old = mine ? st.valid[p] != 0 : false;
            float c_old = mine ? st.cost[p] : 0.0f;
            int s0 = 0, s1 = 0, n_chang = 0;
            bool changed = false;
            for (int q = 0; q < n_random; q++) {
                const int srcl = min(leader + q, 63);
                const int hq = __shfl((int)has, srcl);
                const float cq = __shfl(c_new, srcl);
                const int q0 = __shfl(c0, srcl), q1 = __shfl(c1, srcl);
                if (!hq) { // no value: patchMatchTestCost returns 0 (:197-199)
                    n_chang = 0;
                    continue;
                }
                bool keep;
                if (o.score) keep = has_old ? (cq >= c_old) : true;
                else keep = has_old ? (cq <= c_old) : false;
                if (keep) {
                    s0 = q0;
                    s1 = q1;
                    c_old = cq;
                    has_old = true;
                    changed = true;
                }
                n_chang = keep ? 1 : 0;
            }
            if (mine && k == 0) {
                if (changed) {
                    st.sol[p * o.nd] = s0;
                    if (o.nd == 2) st.sol[p * o.nd + 1] = s1;
                    st.cost[p] = c_old;
                    st.valid[p] = 1;
                }
                total += n_chang;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off);
    if (total && lane == 0) atomicAdd(changes, total);
}

// patchMatchSearch with a LANE per candidate and no LDS (round 5).  The batched kernel above fetches a candidate's two vectors with all
// 64 lanes (coalesced), parks the per-feature terms in an LDS table and lets lane e add up row e in the reference's order -- the table (one
// row per candidate) limits a CU to a handful of waves, and a kernel that gathers scattered 588-byte vectors (1080p RGB 7x7: 147 floats)
// lives on the number of loads it keeps in flight: 35 ms of the 75 ms chain.  Here lane e walks ITS candidate's two vectors itself,
// 16 bytes at a time, and adds the terms as they arrive: the same products / differences, the same order (f = 0, 1, 2, ...), so the same
// bits -- and nothing but registers, so a SIMD holds eight waves of gathers.  The four candidates of a pixel share the source vector
// (one fetch serves the four lanes); a lane's consecutive loads walk the same cache lines.
// IMAGE: the target vector is not fetched at all.  A candidate's 588 bytes of target features are, to a large part, bytes no other
// candidate will ask for again soon (the targets of a pixel's candidates are spread over the search range: 82.9 M evaluations x 588 B =
// 49 GB of gathers per 1080p RGB call), while the target IMAGE (25 MB) and the decorator's (mean, norm) pair per target pixel (16 MB)
// stay in the caches.  The lane forms each target feature the way on_demand_features_kernel formed it -- the sample, minus the mean,
// divided by the norm: the same float operations on the same operands, so the same bits -- and feeds it to the same ordered sum.
// Arithmetic for bytes: about a dozen instructions per feature (the IEEE division) against a gather the memory system ran at 1.5 TB/s.
template <bool IMAGE>
__global__ void __launch_bounds__(256) pm_search_lanes_kernel(OdVolume o, uint64_t seed, uint32_t iter, int n_random, PmState st, int *__restrict__ changes) {
    const int64_t npx = (int64_t)o.Hs * o.Ws;
    const int lane = threadIdx.x & 63, P = 64 / n_random;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    const int slot = lane / n_random, k = lane - slot * n_random;
    int total = 0;
    for (int64_t g0 = wave_id * P; g0 < npx; g0 += n_waves * P) { // (wave uniform: the shuffles below see every lane)
        const int64_t p = g0 + slot;
        const bool mine = lane < P * n_random && p < npx;
        int c0 = 0, c1 = 0;
        bool has = false;
        float c_new = 0.0f;
        if (mine) {
            const int j = (int)(p % o.Ws), i = (int)(p / o.Ws);
            const int base_i = o.nd == 2 ? st.sol[p * o.nd] : 0, base_j = o.nd == 2 ? st.sol[p * o.nd + 1] : st.sol[p * o.nd];
            int disp_i = 0, disp_j;
            if (o.nd == 1) {
                disp_j = pm_in_range(pm_random(seed, iter, i, j, k, 0), o.lower[0], o.upper[0]);
            } else {
                disp_i = pm_in_range(pm_random(seed, iter, i, j, k, 0), o.lower[0], o.upper[0]);
                disp_j = pm_in_range(pm_random(seed, iter, i, j, k, 1), o.lower[1], o.upper[1]);
            }
            int delta_i = disp_i - base_i, delta_j = disp_j - base_j; // :320-331: exploration shrunk towards the current solution
            delta_j *= k + 1;
            delta_j /= n_random + 1;
            if (o.nd == 2) {
                delta_i *= k + 1;
                delta_i /= n_random + 1;
            }
            disp_i = base_i + delta_i;
            disp_j = base_j + delta_j;
            if (o.nd == 1) {
                if (disp_j == base_j) disp_j = base_j + 1;
            } else if (disp_i == base_i && disp_j == base_j) {
                disp_i = base_i + 1;
                disp_j = base_j + 1;
            }
            c0 = o.nd == 2 ? disp_i : disp_j;
            c1 = disp_j;
            int ti = i, tj = j;
            if (o.nd == 2) {
                has = c0 >= o.lower[0] && c0 <= o.upper[0] && c1 >= o.lower[1] && c1 <= o.upper[1];
                ti += c0;
                tj += c1;
            } else {
                has = c0 >= o.lower[0] && c0 <= o.upper[0];
                tj += c0;
            }
            has = has && ti >= 0 && ti < o.Ht && tj >= 0 && tj < o.Wt;
            if (has) { // the ordered sum of this lane's candidate (wave_cost: acc = 0; acc += term(f) for f = 0, 1, 2, ...)
                const float *sv = o.fs + p * o.nF, *tv = o.ft + ((int64_t)ti * o.Wt + tj) * o.nF;
                const int func = o.func;
                auto term_of = [&](float a, float b) {
                    if (func == SVH_SSD || func == SVH_ZSSD) {
                        const float tmp = a - b;
                        return tmp * tmp;
                    }
                    if (func == SVH_SAD || func == SVH_ZSAD) return fabsf(a - b);
                    return a * b;
                };
                if constexpr (IMAGE) {
                    const float2 ms = reinterpret_cast<const float2 *>(o.tstat)[(int64_t)ti * o.Wt + tj];
                    const float mean = ms.x, norm = ms.y;
                    const bool zm = o.zm, nrm = o.nrm;
                    const int C = o.C, v = 2 * o.v_r + 1, h = 2 * o.h_r + 1;
                    int f = 0;
                    for (int kk = 0; kk < v; kk++) { // feature index f = (kk h + ll) C + c: the order of on_demand_features_kernel
                        const float *rowp = o.timg + (int64_t)min(o.Ht - 1, max(0, ti + kk - o.v_r)) * o.Wt * C;
                        for (int ll = 0; ll < h; ll++) {
                            const float *px = rowp + min(o.Wt - 1, max(0, tj + ll - o.h_r)) * C;
                            for (int c = 0; c < C; c++, f++) {
                                float x = px[c];
                                if (zm) x -= mean;
                                if (nrm) x /= norm;
                                c_new += term_of(sv[f], x);
                            }
                        }
                    }
                } else {
                    int f = 0;
#pragma unroll 4
                    for (; f + 4 <= o.nF; f += 4) {
                        const Feat4 a = *reinterpret_cast<const Feat4 *>(sv + f), b = *reinterpret_cast<const Feat4 *>(tv + f);
                        c_new += term_of(a.x, b.x);
                        c_new += term_of(a.y, b.y);
                        c_new += term_of(a.z, b.z);
                        c_new += term_of(a.w, b.w);
                    }
                    for (; f < o.nF; f++) c_new += term_of(sv[f], tv[f]);
                }
            }
        }
        // every lane of a pixel replays the reference's loop over the pixel's candidates (the shuffles need all lanes), its first lane
        // writes the outcome: same comparisons, same order, `n_chang =` keeping only the last outcome (:345)
        {
            const int leader = lane - k;
            bool has_old = mine ? st.valid[p] != 0 : false;
            float c_old = mine ? st.cost[p] : 0.0f;
            int s0 = 0, s1 = 0, n_chang = 0;
            bool changed = false;
            for (int q = 0; q < n_random; q++) {
                const int srcl = min(leader + q, 63);
                const int hq = __shfl((int)has, srcl);
                const float cq = __shfl(c_new, srcl);
                const int q0 = __shfl(c0, srcl), q1 = __shfl(c1, srcl);
                if (!hq) { // no value: patchMatchTestCost returns 0 (:197-199)
                    n_chang = 0;
                    continue;
                }
                bool keep;
                if (o.score) keep = has_old ? (cq >= c_old) : true;
                else keep = has_old ? (cq <= c_old) : false;
                if (keep) {
                    s0 = q0;
                    s1 = q1;
                    c_old = cq;
                    has_old = true;
                    changed = true;
                }
                n_chang = keep ? 1 : 0;
            }
            if (mine && k == 0) {
                if (changed) {
                    st.sol[p * o.nd] = s0;
                    if (o.nd == 2) st.sol[p * o.nd + 1] = s1;
                    st.cost[p] = c_old;
                    st.valid[p] = 1;
                }
                total += n_chang;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off);
    if (total && lane == 0) atomicAdd(changes, total);
}

struct OdInputs {
    int func, nd, h_r, v_r, H, Ws, Ht, Wt, C;
};

int check_params(svh_context *ctx, const svh_on_demand_params *p, const svh_array *src, const svh_array *tgt, OdInputs *in) {
    if (!p) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "null parameters");
    SVH_TRY(validate(ctx, src, "img_source", SVH_F32, 2, 3));
    SVH_TRY(validate(ctx, tgt, "img_target", SVH_F32, 2, 3));
    if (!func_supported(p->match_func) || func_census(p->match_func))
        return fail(ctx, SVH_ERR_UNSUPPORTED, "on-demand volumes take the float matching functions (CC ... ZSAD), not %d", p->match_func);
    if (p->search_dims != 1 && p->search_dims != 2) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "search_dims must be 1 (stereo) or 2 (flow)");
    if (p->h_radius < 0 || p->v_radius < 0 || p->h_radius > 64 || p->v_radius > 64) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad window radius");
    if (src->ndim != tgt->ndim) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "image ranks differ");
    const int C = src->ndim == 3 ? (int)src->shape[2] : 1;
    if (src->ndim == 3 && src->shape[2] != tgt->shape[2]) return fail(ctx, SVH_EMPTY_RESULT, "channel counts differ"); // patchmatch.h:583-585
    if (p->search_dims == 1 && src->shape[0] != tgt->shape[0]) return fail(ctx, SVH_EMPTY_RESULT, "row counts differ"); // :587-591
    if (p->upper1 < p->lower1 || (p->search_dims == 2 && p->upper0 < p->lower0)) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "empty search range");
    *in = OdInputs{p->match_func, p->search_dims, p->h_radius, p->v_radius, (int)src->shape[0], (int)src->shape[1], (int)tgt->shape[0], (int)tgt->shape[1], C};
    return SVH_OK;
}

int dev_on_demand_features(svh_context *ctx, int func, const float *img, int H, int W, int C, int h_r, int v_r, float *out, float2 *stats = nullptr) {
    const int64_t npx = (int64_t)H * W;
    if (npx == 0) return SVH_OK;
    const size_t tile_bytes = (size_t)(2 * v_r + 1) * (ODF_PX + 2 * h_r) * C * sizeof(float);
    if (tile_bytes <= 56 * 1024 && H <= 65535) { // the window rows of a block's pixels in LDS
        SVH_LAUNCH(ctx, "on_demand_features", on_demand_features_tiled_kernel, dim3(ceil_div(W, ODF_PX), H), 256, tile_bytes, img, H, W, C, h_r, v_r,
                   func_zero_mean(func), func_normalized(func), out, stats);
    } else {
        SVH_LAUNCH(ctx, "on_demand_features", on_demand_features_kernel, (int)((npx + ODF_PX - 1) / ODF_PX), 256, 0, img, H, W, C, h_r, v_r,
                   func_zero_mean(func), func_normalized(func), out, stats);
    }
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

int make_volume(svh_context *ctx, Scratch &scr, const svh_on_demand_params *p, const OdInputs &in, const float *d_src, const float *d_tgt, OdVolume *o) {
    const int nF = (2 * in.v_r + 1) * (2 * in.h_r + 1) * in.C;
    float *fs = scr.get_n<float>((size_t)in.H * in.Ws * nF), *ft = scr.get_n<float>((size_t)in.Ht * in.Wt * nF);
    if (!fs || !ft) return SVH_ERR_OUT_OF_MEMORY;
    float2 *tstat = scr.get_n<float2>((size_t)in.Ht * in.Wt);
    if (!tstat) return SVH_ERR_OUT_OF_MEMORY;
    SVH_TRY(dev_on_demand_features(ctx, in.func, d_src, in.H, in.Ws, in.C, in.h_r, in.v_r, fs));
    SVH_TRY(dev_on_demand_features(ctx, in.func, d_tgt, in.Ht, in.Wt, in.C, in.h_r, in.v_r, ft, tstat));
    o->fs = fs;
    o->ft = ft;
    o->timg = d_tgt;
    o->tstat = reinterpret_cast<const float *>(tstat);
    o->C = in.C;
    o->h_r = in.h_r;
    o->v_r = in.v_r;
    o->zm = func_zero_mean(in.func);
    o->nrm = func_normalized(in.func);
    o->func = in.func;
    o->nd = in.nd;
    if (in.nd == 2) {
        o->lower[0] = p->lower0; o->upper[0] = p->upper0; o->lower[1] = p->lower1; o->upper[1] = p->upper1;
    } else { // stereo: the only search dimension is the column axis
        o->lower[0] = p->lower1; o->upper[0] = p->upper1; o->lower[1] = 0; o->upper[1] = 0;
    }
    o->Hs = in.H; o->Ws = in.Ws; o->Ht = in.Ht; o->Wt = in.Wt; o->nF = nF;
    o->score = func_strategy(in.func) == SVH_SCORE;
    return SVH_OK;
}

} // namespace

} // namespace svh

using namespace svh;

extern "C" int svh_on_demand_features(svh_context *ctx, int match_func, const svh_array *img, int h_radius, int v_radius, svh_array *out) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, img, "img", SVH_F32, 2, 3));
    SVH_TRY(validate(ctx, out, "out", SVH_F32, 3, 3));
    if (!func_supported(match_func) || func_census(match_func)) return fail(ctx, SVH_ERR_UNSUPPORTED, "matching function %d", match_func);
    if (h_radius < 0 || v_radius < 0 || h_radius > 64 || v_radius > 64) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad window radius");
    const int H = (int)img->shape[0], W = (int)img->shape[1], C = img->ndim == 3 ? (int)img->shape[2] : 1;
    const int nF = (2 * v_radius + 1) * (2 * h_radius + 1) * C;
    if (out->shape[0] != H || out->shape[1] != W || out->shape[2] != nF) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "out must have shape (%d,%d,%d)", H, W, nF);
    Scratch scr(ctx);
    void *di;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *img, &di));
    SVH_TRY(stage_out(ctx, scr, *out, &os));
    SVH_TRY(dev_on_demand_features(ctx, match_func, (const float *)di, H, W, C, h_radius, v_radius, (float *)os.dptr));
    return finish_out(ctx, os);
}

extern "C" int svh_on_demand_truncated_cost_volume(svh_context *ctx, const svh_on_demand_params *params, const svh_array *img_source,
                                                   const svh_array *img_target, const svh_array *disp, int radius, svh_array *tcv) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    OdInputs in;
    SVH_TRY(check_params(ctx, params, img_source, img_target, &in));
    SVH_TRY(validate(ctx, disp, "disp", SVH_I32, 3, 3));
    if (radius < 0 || radius > 64) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad radius");
    if (disp->shape[0] != in.H || disp->shape[1] != in.Ws) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "disp must have shape (%d,%d,%d)", in.H, in.Ws, in.nd);
    if (disp->shape[2] != in.nd) return fail(ctx, SVH_EMPTY_RESULT, "disp holds %lld components, the volume searches %d dimensions", (long long)disp->shape[2], in.nd); // :478-480
    const int T = 2 * radius + 1;
    SVH_TRY(validate(ctx, tcv, "tcv", SVH_F32, 2 + in.nd, 2 + in.nd));
    if (tcv->shape[0] != in.H || tcv->shape[1] != in.Ws || tcv->shape[2] != T || (in.nd == 2 && tcv->shape[3] != T))
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "tcv must have shape (H, W, %d%s)", T, in.nd == 2 ? ", same" : "");
    Scratch scr(ctx);
    void *ds, *dt, *dd;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *img_source, &ds));
    SVH_TRY(stage_in(ctx, scr, *img_target, &dt));
    SVH_TRY(stage_in(ctx, scr, *disp, &dd));
    SVH_TRY(stage_out(ctx, scr, *tcv, &os));
    OdVolume o;
    SVH_TRY(make_volume(ctx, scr, params, in, (const float *)ds, (const float *)dt, &o));
    const int64_t n = (int64_t)in.H * in.Ws * (in.nd == 2 ? T * T : T);
    if (n) {
        // defaultCvValForMatchFunc, matching_costs.h:706-713: max() for costs, min() (the smallest positive normal) for scores
        SVH_LAUNCH(ctx, "on_demand_truncated", od_truncated_kernel, grid_for(n, 256, 16384), 256, 0, o, (const int32_t *)dd, radius, o.score ? FLT_MIN : FLT_MAX,
                   (float *)os.dptr);
        SVH_CHECK_LAUNCH(ctx);
    }
    return finish_out(ctx, os);
}

namespace svh {
namespace {
// patchMatchImpl (patchmatch.h:447-493) on a volume of feature vectors: initial solution (drawn, or the caller's), then n_iter rounds of
// row sweep, column sweep and random search.  `sol`: (H, Ws, nd) int32 on the device; *it_out: iterations run.
int run_patch_match(svh_context *ctx, Scratch &scr, const OdInputs &in, const OdVolume &o, int n_iter, int n_random_search, uint64_t seed,
                    const int32_t *d_init, int32_t *sol, int *it_out) {
    int it = 0;
    const int64_t npx = (int64_t)in.H * in.Ws;
    if (npx > 0) {
        int *changes = scr.get_n<int>(1);
        PmState st{sol, scr.get_n<float>((size_t)npx), scr.get_n<uint8_t>((size_t)npx)};
        if (!changes || !st.cost || !st.valid) return SVH_ERR_OUT_OF_MEMORY;
        const size_t shmem = (size_t)((o.nF + 3) & ~3) * sizeof(float);
        const int px_grid = (int)std::min<int64_t>(npx, 256 * 64); // a wave per pixel, grid-stride
        float *pcost = nullptr;
        uint8_t *pvalid = nullptr;
        int32_t *work = nullptr;
        int *n_work = nullptr;
        if (ctx->patchmatch_pred_costs && npx < (1ll << 31) && in.H <= 65535) {
            pcost = scr.get_n<float>((size_t)npx * PM_DEPTH);   // [depth][pixel]: the cost against the pre-sweep solution `depth + 1` steps back
            pvalid = scr.get_n<uint8_t>((size_t)npx * PM_DEPTH);
            work = scr.get_n<int32_t>((size_t)npx * PM_DEPTH); // (pixel, depth) pairs the pre-pass evaluates
            n_work = scr.get_n<int>(2);
            if (!pcost || !pvalid || !work || !n_work) return SVH_ERR_OUT_OF_MEMORY;
        }
        // run batches of the sweeps: PM_RUN rows of terms in LDS (row pitch an odd number of 16-byte pieces: the eight rows' reads fall into different banks)
        const int run_pitch = ctx->patchmatch_run_batches ? 4 * (((o.nF + 3) / 4) | 1) : 0;
        const bool run_batches = run_pitch > 0 && (size_t)PM_RUN * run_pitch * sizeof(float) <= 48 * 1024;
        const size_t sweep_shmem = run_batches ? (size_t)PM_RUN * run_pitch * sizeof(float) : shmem;
        // 64 costs per wave (pm_costs64: the random search's evaluation) wherever costs are evaluated for many pixels at once: the first draw, the
        // pre-pass of the sweeps; "patchmatch_search_form" = 0 keeps the wave per cost everywhere
        const bool costs64 = ctx->patchmatch_search_form != 0 && (int64_t)in.Ht * in.Wt < (1ll << 31) && npx < (1ll << 29);
        const int grid64 = (int)std::min<int64_t>((npx + 63) / 64, 256 * 32);
        if (costs64) SVH_LAUNCH(ctx, "patchmatch_init", pm_init64_kernel, grid64, 64, 0, o, seed, st, d_init);
        else SVH_LAUNCH(ctx, "patchmatch_init", pm_init_kernel, px_grid, 64, shmem, o, seed, st, d_init);
        for (; it < n_iter; it++) {
            SVH_HIP_CHECK(ctx, hipMemsetAsync(changes, 0, sizeof(int), ctx->stream));
            const int inc0 = (it % 4) < 2 ? 1 : -1, inc1 = (it % 2) == 0 ? 1 : -1; // propagation_direction.h:64-86, patchmatch.h:462-479
            // (prefetching the next pixel's vectors under both outcomes of the current test was tried: no gain -- with one wave per
            // SIMD a step is bound by its ~150 dependent additions and the instructions around them, not by the loads)
            if (pcost) { // the cost of every pixel against its predecessor's solution, in parallel; then the sweep (pm_pred_cost_kernel)
                const dim3 cgrid(ceil_div(in.Ws, 256), in.H);
                SVH_HIP_CHECK(ctx, hipMemsetAsync(n_work, 0, 2 * sizeof(int), ctx->stream));
                // iteration 0 starts from random solutions: what travels there travels far (batches); afterwards the lines that still evaluate
                // are the ones that never settle, in short runs (look-back depths)
                const int depths = (it == 0 || !ctx->patchmatch_lookback) ? 1 : PM_DEPTH;
                const int rp = run_batches && (it == 0 || ctx->patchmatch_scan_chunks) ? run_pitch : 0;
                // the pre-pass of a sweep: one pair of launches for all depths (fewer than 2^29 pixels), or a pair per depth
                auto pre_pass = [&](int axis, int inc, int *counter) -> int {
                    if (npx < (1ll << 29)) {
                        SVH_LAUNCH(ctx, "patchmatch_pred_cost", pm_pred_classify_depths_kernel, cgrid, 256, 0, o, axis, inc, st, pcost, pvalid, npx, (uint32_t *)work, counter, depths);
                        if (costs64) SVH_LAUNCH(ctx, "patchmatch_pred_cost", pm_pred_cost64_kernel, grid64, 64, 0, o, axis, inc, st, pcost, pvalid, npx, (const uint32_t *)work, counter);
                        else SVH_LAUNCH(ctx, "patchmatch_pred_cost", pm_pred_cost_depths_kernel, px_grid, 64, shmem, o, axis, inc, st, pcost, pvalid, npx, (const uint32_t *)work, counter);
                        return SVH_OK;
                    }
                    for (int m = 1; m <= depths; m++) {
                        if (m > 1) SVH_HIP_CHECK(ctx, hipMemsetAsync(counter, 0, sizeof(int), ctx->stream));
                        SVH_LAUNCH(ctx, "patchmatch_pred_cost", pm_pred_classify_kernel, cgrid, 256, 0, o, axis, inc, st, pcost + (size_t)(m - 1) * npx, pvalid + (size_t)(m - 1) * npx, work, counter, m);
                        SVH_LAUNCH(ctx, "patchmatch_pred_cost", pm_pred_cost_kernel, px_grid, 64, shmem, o, axis, inc, st, pcost + (size_t)(m - 1) * npx, pvalid + (size_t)(m - 1) * npx, work, counter, m);
                    }
                    return SVH_OK;
                };
                SVH_TRY(pre_pass(0, inc1, n_work));
                SVH_LAUNCH(ctx, "patchmatch_rows", pm_rows_fast_kernel, in.H, 64, sweep_shmem, o, inc1, st, pcost, pvalid, changes, rp, depths, ctx->patchmatch_scan_chunks && it > 0 ? 1 : 0);
                SVH_TRY(pre_pass(1, inc0, n_work + 1));
                SVH_LAUNCH(ctx, "patchmatch_cols", pm_cols_fast_kernel, in.Ws, 64, sweep_shmem, o, inc0, st, pcost, pvalid, changes, rp, depths, ctx->patchmatch_scan_chunks && it > 0 ? 1 : 0);
            } else {
                SVH_LAUNCH(ctx, "patchmatch_rows", pm_rows_kernel, in.H, 64, shmem, o, inc1, st, changes);
                SVH_LAUNCH(ctx, "patchmatch_cols", pm_cols_kernel, in.Ws, 64, shmem, o, inc0, st, changes);
            }
            if (ctx->patchmatch_search_form && n_random_search > 0 && n_random_search <= 64 && (int64_t)in.Ht * in.Wt < (1ll << 31) && npx < (1ll << 31)) {
                // 64 candidates per wave: P = 64 / n pixels per round
                const int P = 64 / n_random_search;
                const int64_t waves = (npx + P - 1) / P;
                if (ctx->patchmatch_search_form == 1) {
                    SVH_LAUNCH(ctx, "patchmatch_search", pm_search_chunked_kernel, (int)std::min<int64_t>(waves, 256 * 32), 64, 0, o, seed, (uint32_t)it, n_random_search, st, changes);
                } else {
                    const int blocks = (int)std::min<int64_t>((waves + 3) / 4, 256 * 8);
                    if (ctx->patchmatch_search_form == 2 || !o.timg) SVH_LAUNCH(ctx, "patchmatch_search", pm_search_lanes_kernel<false>, blocks, 256, 0, o, seed, (uint32_t)it, n_random_search, st, changes);
                    else SVH_LAUNCH(ctx, "patchmatch_search", pm_search_lanes_kernel<true>, blocks, 256, 0, o, seed, (uint32_t)it, n_random_search, st, changes);
                }
                SVH_CHECK_LAUNCH(ctx);
                int h_changes = 0;
                SVH_HIP_CHECK(ctx, hipMemcpyAsync(&h_changes, changes, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                SVH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));

                if (h_changes == 0) { // patchmatch.h:486-488
                    it++;
                    break;
                }
                continue;
            }
            // batched search when at least one pixel's candidates fit a wave and an LDS table of 60 KB
            const int pitch = o.nF | 1;
            int E = std::min<int>(64, (int)(16 * 1024 / (pitch * sizeof(float)))); // table of at most 16 KB: ten waves per CU keep loads in flight
            if (n_random_search > 0 && E >= n_random_search) {
                E = (E / n_random_search) * n_random_search;
                const int P = E / n_random_search;
                const int groups = (int)std::min<int64_t>((npx + P - 1) / P, 256 * 16);
                SVH_LAUNCH(ctx, "patchmatch_search", pm_search_batched_kernel, groups, 64, (size_t)E * pitch * sizeof(float), o, seed, (uint32_t)it,
                           n_random_search, E, pitch, st, changes);
            } else {
                SVH_LAUNCH(ctx, "patchmatch_search", pm_search_kernel, px_grid, 64, shmem, o, seed, (uint32_t)it, n_random_search, st, changes);
            }
            SVH_CHECK_LAUNCH(ctx);
            int h_changes = 0;
            SVH_HIP_CHECK(ctx, hipMemcpyAsync(&h_changes, changes, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            SVH_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            if (h_changes == 0) { // patchmatch.h:486-488
                it++;
                break;
            }
        }
    }
    *it_out = it;
    return SVH_OK;
}
} // namespace
} // namespace svh

extern "C" int svh_cacheless_patch_match_init(svh_context *ctx, const svh_on_demand_params *params, const svh_array *img_source, const svh_array *img_target,
                                              int n_iter, int n_random_search, uint64_t seed, const svh_array *initial_disp, svh_array *disp,
                                              int32_t *iterations_run) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    OdInputs in;
    SVH_TRY(check_params(ctx, params, img_source, img_target, &in));
    SVH_TRY(validate(ctx, disp, "disp", SVH_I32, 3, 3));
    if (n_iter < 0 || n_random_search < 0) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "iteration counts must be non-negative");
    if (disp->shape[0] != in.H || disp->shape[1] != in.Ws || disp->shape[2] != in.nd)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "disp must have shape (%d,%d,%d)", in.H, in.Ws, in.nd);
    if (initial_disp) {
        SVH_TRY(validate(ctx, initial_disp, "initial_disp", SVH_I32, 3, 3));
        if (initial_disp->shape[0] != in.H || initial_disp->shape[1] != in.Ws || initial_disp->shape[2] != in.nd)
            return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "initial_disp must have shape (%d,%d,%d)", in.H, in.Ws, in.nd);
    }
    Scratch scr(ctx);
    void *ds, *dt, *di = nullptr;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *img_source, &ds));
    SVH_TRY(stage_in(ctx, scr, *img_target, &dt));
    if (initial_disp) SVH_TRY(stage_in(ctx, scr, *initial_disp, &di));
    SVH_TRY(stage_out(ctx, scr, *disp, &os));
    int it = 0;
    if ((int64_t)in.H * in.Ws > 0) {
        OdVolume o;
        SVH_TRY(make_volume(ctx, scr, params, in, (const float *)ds, (const float *)dt, &o));
        SVH_TRY(run_patch_match(ctx, scr, in, o, n_iter, n_random_search, seed, (const int32_t *)di, (int32_t *)os.dptr, &it));
    }
    if (iterations_run) *iterations_run = it;
    return finish_out(ctx, os);
}

extern "C" int svh_cacheless_patch_match(svh_context *ctx, const svh_on_demand_params *params, const svh_array *img_source, const svh_array *img_target,
                                         int n_iter, int n_random_search, uint64_t seed, svh_array *disp, int32_t *iterations_run) {
    return svh_cacheless_patch_match_init(ctx, params, img_source, img_target, n_iter, n_random_search, seed, nullptr, disp, iterations_run);
}

// patchMatch (patchmatch.h:496-558): the same iteration on FEATURE VOLUMES the caller built (benchmarkStereoMatchingModels.cpp:187-199 passes
// unfolded images), through the reference's cached cost volume -- whose values are featureComparison of the processed vectors, cached or
// not.  The vectors are processed TWICE when the function is zero-mean or normalised: the entry point calls getFeatureVolumeForMatchFunc
// (:522-523) and the cost volume's constructor calls it again on the result (on_demand_cost_volume.h:62-67); both passes are run here.
extern "C" int svh_patch_match(svh_context *ctx, const svh_on_demand_params *params, const svh_array *feat_source, const svh_array *feat_target, int n_iter,
                               int n_random_search, uint64_t seed, const svh_array *initial_disp, svh_array *disp, int32_t *iterations_run) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    if (!params) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "null parameters");
    SVH_TRY(validate(ctx, feat_source, "feature_vol_s", SVH_F32, 3, 3));
    SVH_TRY(validate(ctx, feat_target, "feature_vol_t", SVH_F32, 3, 3));
    if (!func_supported(params->match_func) || func_census(params->match_func))
        return fail(ctx, SVH_ERR_UNSUPPORTED, "PatchMatch takes the float matching functions (CC ... ZSAD), not %d", params->match_func);
    if (params->search_dims != 1 && params->search_dims != 2) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "search_dims must be 1 (stereo) or 2 (flow)");
    if (feat_source->shape[2] != feat_target->shape[2]) return fail(ctx, SVH_EMPTY_RESULT, "feature counts differ"); // patchmatch.h:529-531
    if (params->search_dims == 1 && feat_source->shape[0] != feat_target->shape[0]) return fail(ctx, SVH_EMPTY_RESULT, "row counts differ"); // :533-537
    if (params->upper1 < params->lower1 || (params->search_dims == 2 && params->upper0 < params->lower0)) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "empty search range");
    if (n_iter < 0 || n_random_search < 0) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "iteration counts must be non-negative");
    OdInputs in{params->match_func, params->search_dims, 0, 0, (int)feat_source->shape[0], (int)feat_source->shape[1], (int)feat_target->shape[0],
                (int)feat_target->shape[1], 1};
    const int F = (int)feat_source->shape[2];
    SVH_TRY(validate(ctx, disp, "disp", SVH_I32, 3, 3));
    if (disp->shape[0] != in.H || disp->shape[1] != in.Ws || disp->shape[2] != in.nd)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "disp must have shape (%d,%d,%d)", in.H, in.Ws, in.nd);
    if (initial_disp) {
        SVH_TRY(validate(ctx, initial_disp, "initial_disp", SVH_I32, 3, 3));
        if (initial_disp->shape[0] != in.H || initial_disp->shape[1] != in.Ws || initial_disp->shape[2] != in.nd)
            return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "initial_disp must have shape (%d,%d,%d)", in.H, in.Ws, in.nd);
    }
    Scratch scr(ctx);
    void *ds, *dt, *di = nullptr;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *feat_source, &ds));
    SVH_TRY(stage_in(ctx, scr, *feat_target, &dt));
    if (initial_disp) SVH_TRY(stage_in(ctx, scr, *initial_disp, &di));
    SVH_TRY(stage_out(ctx, scr, *disp, &os));
    int it = 0;
    if ((int64_t)in.H * in.Ws > 0 && F > 0) {
        const float *fs = (const float *)ds, *ft = (const float *)dt;
        if (func_zero_mean(in.func) || func_normalized(in.func)) { // processed twice (see above)
            float *ps = scr.get_n<float>((size_t)in.H * in.Ws * F), *pt = scr.get_n<float>((size_t)in.Ht * in.Wt * F);
            float *tmp = scr.get_n<float>((size_t)std::max((int64_t)in.H * in.Ws, (int64_t)in.Ht * in.Wt) * F);
            if (!ps || !pt || !tmp) return SVH_ERR_OUT_OF_MEMORY;
            SVH_TRY(dev_feature_volume_for_match_func(ctx, scr, in.func, fs, in.H, in.Ws, F, tmp));
            SVH_TRY(dev_feature_volume_for_match_func(ctx, scr, in.func, tmp, in.H, in.Ws, F, ps));
            SVH_TRY(dev_feature_volume_for_match_func(ctx, scr, in.func, ft, in.Ht, in.Wt, F, tmp));
            SVH_TRY(dev_feature_volume_for_match_func(ctx, scr, in.func, tmp, in.Ht, in.Wt, F, pt));
            fs = ps;
            ft = pt;
        }
        OdVolume o;
        o.fs = fs;
        o.ft = ft;
        o.func = in.func;
        o.nd = in.nd;
        if (in.nd == 2) {
            o.lower[0] = params->lower0; o.upper[0] = params->upper0; o.lower[1] = params->lower1; o.upper[1] = params->upper1;
        } else {
            o.lower[0] = params->lower1; o.upper[0] = params->upper1; o.lower[1] = 0; o.upper[1] = 0;
        }
        o.Hs = in.H; o.Ws = in.Ws; o.Ht = in.Ht; o.Wt = in.Wt; o.nF = F;
        o.score = func_strategy(in.func) == SVH_SCORE;
        SVH_TRY(run_patch_match(ctx, scr, in, o, n_iter, n_random_search, seed, (const int32_t *)di, (int32_t *)os.dptr, &it));
    }
    if (iterations_run) *iterations_run = it;
    return finish_out(ctx, os);
}

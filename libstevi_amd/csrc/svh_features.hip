// unfold (A1), census features / transform (A2, A3).
//
// The image-based census never materialises the unfolded volume: one thread per output pixel walks the
// window in channel order (c = C(2h_r+1)k + C l + ch, correlation/unfold.h:180) and packs comparisons
// against channel 0 (the window's top-left sample, correlation/census.h:89) into 32-bit words, storing a
// word only when it is full (census.h:103-108).  Images are a few MB and stay in L2; neighbouring threads
// read overlapping windows, so the loads are L1/L2 hits.
#include "svh_internal.h"

namespace svh {

__device__ __forceinline__ float image_or_zero(const float *__restrict__ img, int H, int W, int C, int i, int j, int ch) {
    // valueOrAlt({i,j[,c]}, 0): correlation/unfold.h:284, :335
    return (i >= 0 && i < H && j >= 0 && j < W) ? img[((int64_t)i * W + j) * C + ch] : 0.0f;
}

// rule E2: `float t = word; word' = t;` of cross_correlations.h:235-236.  uint32 -> float rounds to nearest
// even; the way back is undefined in C++ when the rounded value is 2^32 (words >= 0xFFFFFF80), and the reference's builds differ:
//   mode 1 (saturate): 0xFFFFFFFF -- a Release build (-march=native, CMakeLists.txt:41) on a host with AVX-512 (vcvttss2usi), and what
//                      v_cvt_u32_f32 does here;
//   mode 2 (zero):     0 -- x86-64 without AVX-512 code generation (-mavx -mavx2 -mfma, CMakeLists.txt:44-58, and every Debug build):
//                      the conversion goes through vcvttss2si r64 and keeps the low 32 bits of 2^32.
// svh_context_set_option("census_float_overflow", 0 | 1) picks saturate | zero; 0 in a kernel argument means "no round trip".
__device__ __forceinline__ uint32_t round_word_through_float(uint32_t w, int mode) {
    float t = (float)w;
    return t >= 4294967296.0f ? (mode == 2 ? 0u : 0xFFFFFFFFu) : (uint32_t)t;
}

__global__ void unfold_kernel(const float *__restrict__ img, int H, int W, int C, int h_r, int v_r, int pl, int pt, int Ho,
                              int Wo, float *__restrict__ out) {
    const int h = 2 * h_r + 1, v = 2 * v_r + 1;
    const int F = h * v * C;
    const int64_t n = (int64_t)Ho * Wo * F;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(e % F);
        int64_t p = e / F;
        int j = (int)(p % Wo), i = (int)(p / Wo);
        int ch = c % C;
        int l = (c / C) % h;
        int k = c / (C * h);
        out[e] = image_or_zero(img, H, W, C, i - pt + k, j - pl + l, ch);
    }
}

// unfold with a patch orientation: sample (k, l, ch) goes to channelFromCord(k, l, ch, h, v, C, orientation) (unfold.h:171-191)
__global__ void unfold_oriented_kernel(const float *__restrict__ img, int H, int W, int C, int h_r, int v_r, int pl, int pt, int Ho, int Wo,
                                       int orientation, float *__restrict__ out) {
    const int h = 2 * h_r + 1, v = 2 * v_r + 1;
    const int F = h * v * C;
    const int64_t n = (int64_t)Ho * Wo * F;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(e % F); // source-ordered sample index (k, l, ch)
        const int64_t p = e / F;
        const int j = (int)(p % Wo), i = (int)(p / Wo);
        const int ch = s % C, l = (s / C) % h, k = s / (C * h);
        int c;
        if (orientation == 1) c = C * v * (h - l - 1) + C * k + ch;
        else if (orientation == 2) c = C * h * (v - k - 1) + C * (h - l - 1) + ch;
        else if (orientation == 3) c = C * v * l + C * (v - k - 1) + ch;
        else c = s;
        out[p * F + c] = image_or_zero(img, H, W, C, i - pt + k, j - pl + l, ch);
    }
}

__global__ void census_image_kernel(const float *__restrict__ img, int H, int W, int C, int h_r, int v_r, int pl, int pt,
                                    int Ho, int Wo, int n_out, int n_written, int round_target,
                                    uint32_t *__restrict__ words) {
    const int h = 2 * h_r + 1, v = 2 * v_r + 1;
    const int64_t npx = (int64_t)Ho * Wo;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npx; p += (int64_t)gridDim.x * blockDim.x) {
        int j = (int)(p % Wo), i = (int)(p / Wo);
        const float ref = image_or_zero(img, H, W, C, i - pt, j - pl, 0);
        uint32_t *o = words + p * n_out;
        uint32_t d = 0;
        int b = 0, word = 0;
        int c = 0;
        for (int k = 0; k < v && word < n_written; k++) {
            for (int l = 0; l < h && word < n_written; l++) {
                for (int ch = 0; ch < C; ch++, c++) {
                    if (c == 0) continue; // channel 0 is the reference sample itself
                    if (word >= n_written) break;
                    float val = image_or_zero(img, H, W, C, i - pt + k, j - pl + l, ch);
                    d |= (ref > val ? 1u : 0u) << b;
                    if (++b == 32) {
                        o[word++] = round_target ? round_word_through_float(d, round_target) : d;
                        d = 0;
                        b = 0;
                    }
                }
            }
        }
        for (int w = n_written; w < n_out; w++) o[w] = 0; // rule E1: never-written trailing word
    }
}

// LDS-tiled variant: a block produces 256 consecutive pixels of one output row from a staged
// (2v_r+1) x (256 + 2h_r) x C tile (zero outside the image), so each input sample is fetched from memory once
// per block instead of once per window.  Lanes read consecutive LDS dwords (conflict free for C == 1).
constexpr int CENSUS_TJ = 256;

__global__ void __launch_bounds__(CENSUS_TJ) census_image_tiled_kernel(const float *__restrict__ img, int H, int W, int C, int h_r,
                                                                       int v_r, int pl, int pt, int Ho, int Wo, int n_out,
                                                                       int n_written, int round_target,
                                                                       uint32_t *__restrict__ words) {
    extern __shared__ float tile[];
    const int h = 2 * h_r + 1, v = 2 * v_r + 1;
    const int i = blockIdx.y, j0 = blockIdx.x * CENSUS_TJ, tj = threadIdx.x, j = j0 + tj;
    const int tw = (CENSUS_TJ + h - 1) * C; // floats per tile row
    for (int k = 0; k < v; k++) {
        const int ii = i - pt + k;
        const bool row_in = ii >= 0 && ii < H;
        const float *row = img + (int64_t)ii * W * C;
        for (int e = tj; e < tw; e += CENSUS_TJ) {
            const int jj = j0 - pl + e / C;
            tile[k * tw + e] = (row_in && jj >= 0 && jj < W) ? row[(int64_t)(j0 - pl) * C + e] : 0.0f;
        }
    }
    __syncthreads();
    if (j >= Wo) return;
    const float ref = tile[tj * C];
    uint32_t *o = words + ((int64_t)i * Wo + j) * n_out;
    // walk the window in channel order c = (h C) k + (l C + ch), starting after channel 0 (the reference sample)
    const int hc = h * C;
    const float *tp = tile + tj * C; // + k * tw + rem
    int rem = 1, row_off = 0;
    if (rem == hc) {
        rem = 0;
        row_off = tw;
    }
    for (int word = 0; word < n_written; word++) {
        uint32_t d = 0;
        for (int b = 0; b < 32; b++) {
            d |= (ref > tp[row_off + rem] ? 1u : 0u) << b;
            if (++rem == hc) {
                rem = 0;
                row_off += tw;
            }
        }
        o[word] = round_target ? round_word_through_float(d, round_target) : d;
    }
    for (int w = n_written; w < n_out; w++) o[w] = 0; // rule E1
}

// Grey images with a compile-time window (the common 7x7 / 9x9 / 11x11 cases): the window walk is fully unrolled, every
// LDS read has an immediate offset, and the word boundaries are known at compile time.
struct CensusJob { // one image of a launch (blockIdx.z selects): both images of a stereo pair go in one launch
    const float *img;
    int H, W, Ho, Wo;
    int round_target;
    uint32_t *words;
};

// word = 2 * word + (ref > sample): the compare writes VCC, the add-with-carry consumes it
__device__ __forceinline__ void shift_in_greater(uint32_t &word, float ref, float sample) {
    asm("v_cmp_gt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(word) : "v"(ref), "v"(sample) : "vcc");
}

// The same for finite operands with every zero stored as +0 (what the staged tile holds when `all_finite`): the sign of
// sample - ref is set exactly when ref > sample (a difference of two distinct floats is never zero: subnormals are kept), so
// word = 2 word + (ref > sample) is v_sub_f32 + v_alignbit_b32 -- 5.5 issue cycles per pair instead of 8.8 for v_cmp + v_addc
// (tools/ubench_valu.hip).  Not for NaN or infinities: inf - inf is a NEGATIVE NaN on this hardware and a NaN operand keeps or
// flips its sign; tiles that hold any take the compare form.
// Eight bits per asm statement: left to itself the compiler packs pairs of the subtractions into v_pk_add_f32 and pays two v_mov per pair
// to do it, hence asm -- and behind an asm statement whose result the next instruction touches it puts an s_nop (it cannot see that a
// plain VALU result needs none): with one subtraction per statement that was one nop per three bits, 963 of the kernel's 4 000 issue
// slots; with eight bits per statement it is one per sixteen instructions.
__device__ __forceinline__ void shift_in_sign8(uint32_t &word, float ref, float s0, float s1, float s2, float s3, float s4, float s5, float s6, float s7) {
    float d;
    asm("v_sub_f32 %1, %3, %2\n\tv_alignbit_b32 %0, %0, %1, 31\n\t"
        "v_sub_f32 %1, %4, %2\n\tv_alignbit_b32 %0, %0, %1, 31\n\t"
        "v_sub_f32 %1, %5, %2\n\tv_alignbit_b32 %0, %0, %1, 31\n\t"
        "v_sub_f32 %1, %6, %2\n\tv_alignbit_b32 %0, %0, %1, 31\n\t"
        "v_sub_f32 %1, %7, %2\n\tv_alignbit_b32 %0, %0, %1, 31\n\t"
        "v_sub_f32 %1, %8, %2\n\tv_alignbit_b32 %0, %0, %1, 31\n\t"
        "v_sub_f32 %1, %9, %2\n\tv_alignbit_b32 %0, %0, %1, 31\n\t"
        "v_sub_f32 %1, %10, %2\n\tv_alignbit_b32 %0, %0, %1, 31"
        : "+v"(word), "=&v"(d)
        : "v"(ref), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "v"(s6), "v"(s7));
}

// ROWS output rows per block: the (2 VR + ROWS) x (512 + 2 HR) tile is staged once and every lane keeps its
// (2 VR + ROWS) x (2 HR + 2) samples in registers, so an output row costs a staged row and (2 HR + 2) / 2 LDS reads
// instead of a whole window of each
template <int HR, int VR, int ROWS, int TJN>
__global__ void __launch_bounds__(TJN) census_grey_kernel(CensusJob job0, CensusJob job1, int pl, int pt, int n_out) {
    const CensusJob &job = blockIdx.z == 0 ? job0 : job1;
    const float *__restrict__ img = job.img;
    uint32_t *__restrict__ words = job.words;
    const int H = job.H, W = job.W, Ho = job.Ho, Wo = job.Wo;
    const int round_target = job.round_target;
    constexpr int PXB = 2 * TJN; // two neighbouring pixels per lane
    if ((int)blockIdx.y * ROWS >= Ho || (int)blockIdx.x * PXB >= Wo) return; // the grid covers the larger image
    constexpr int h = 2 * HR + 1, v = 2 * VR + 1, TR = v + ROWS - 1, TW = PXB + h + 1; // row pitch even: float2 reads stay 8-byte aligned
    constexpr int NWRITTEN = (h * v - 1) / 32;
    __shared__ __attribute__((aligned(16))) float tile[TR * TW];
    const int i0 = blockIdx.y * ROWS, j0 = blockIdx.x * PXB, tj = threadIdx.x;
    // stage the tile: every load of the thread is issued before the first LDS write, so a block pays one memory latency
    constexpr int PER_ROW = (TW + TJN - 1) / TJN;
    int not_finite = 0;
    {
        float r[TR][PER_ROW];
#pragma unroll
        for (int k = 0; k < TR; k++) {
            const int ii = i0 - pt + k;
            const bool row_in = ii >= 0 && ii < H;
            const float *row = img + (int64_t)ii * W;
#pragma unroll
            for (int q = 0; q < PER_ROW; q++) {
                const int e = tj + q * TJN, jj = j0 - pl + e;
                r[k][q] = (row_in && e < TW && jj >= 0 && jj < W) ? row[jj] : 0.0f;
            }
        }
        // -0 -> +0 (the comparisons cannot tell them apart, the sign test below can); note any NaN / infinity of the tile
#pragma unroll
        for (int k = 0; k < TR; k++) {
#pragma unroll
            for (int q = 0; q < PER_ROW; q++) {
                const int e = tj + q * TJN;
                float x;
                asm("v_add_f32 %0, 0, %1" : "=v"(x) : "v"(r[k][q])); // (x + 0.0f, kept from the optimiser)
                not_finite |= (__float_as_uint(x) & 0x7F800000u) == 0x7F800000u;
                if (e < TW) tile[k * TW + e] = x;
            }
        }
    }
    const bool all_finite = !__syncthreads_or(not_finite);
    // pixels A = j0 + 2 tj and B = A + 1 share the window columns 2 tj .. 2 tj + h: h + 1 samples per row, read as (h+1)/2
    // 8-byte pairs (ds_read_b64); A compares against sample l, B against sample l + 1
    const int jA = j0 + 2 * tj;
    if (jA >= Wo) return;
    const float *tp = tile + 2 * tj;
    float smp[TR][h + 1];
#pragma unroll
    for (int k = 0; k < TR; k++) {
#pragma unroll
        for (int q = 0; q < (h + 1) / 2; q++) {
            const float2 pr = *reinterpret_cast<const float2 *>(tp + k * TW + 2 * q);
            smp[k][2 * q] = pr.x;
            smp[k][2 * q + 1] = pr.y;
        }
    }
    const bool hasB = jA + 1 < Wo;
#pragma unroll
    for (int rr = 0; rr < ROWS; rr++) {
        const int i = i0 + rr;
        if (i >= Ho) break;
        // the bits of each word from the highest channel down: word = 2 word + (ref > sample) is one v_cmp (-> VCC) and one
        // v_addc (carry-in = VCC) per bit; the reference pixel is the window's top-left sample (finding F6)
        const float refA = smp[rr][0], refB = smp[rr][1];
        uint32_t dA[NWRITTEN > 0 ? NWRITTEN : 1] = {}, dB[NWRITTEN > 0 ? NWRITTEN : 1] = {};
        if (all_finite) { // (block uniform) the sign of sample - ref
#pragma unroll
            for (int w = 0; w < NWRITTEN; w++) {
#pragma unroll
                for (int b = 31; b >= 0; b -= 8) {
                    // channel index c = 32 w + b + 1 (unfold.h:180) behind bit b of word w (census.h:98-108): window row c / h, column c % h
#define SVH_SMP(BIT, SHIFT) smp[rr + (32 * w + (BIT) + 1) / h][(32 * w + (BIT) + 1) % h + (SHIFT)]
                    shift_in_sign8(dA[w], refA, SVH_SMP(b, 0), SVH_SMP(b - 1, 0), SVH_SMP(b - 2, 0), SVH_SMP(b - 3, 0), SVH_SMP(b - 4, 0), SVH_SMP(b - 5, 0),
                                   SVH_SMP(b - 6, 0), SVH_SMP(b - 7, 0));
                    shift_in_sign8(dB[w], refB, SVH_SMP(b, 1), SVH_SMP(b - 1, 1), SVH_SMP(b - 2, 1), SVH_SMP(b - 3, 1), SVH_SMP(b - 4, 1), SVH_SMP(b - 5, 1),
                                   SVH_SMP(b - 6, 1), SVH_SMP(b - 7, 1));
#undef SVH_SMP
                }
            }
        } else {
#pragma unroll
            for (int w = 0; w < NWRITTEN; w++) {
#pragma unroll
                for (int b = 31; b >= 0; b--) {
                    const int c = 32 * w + b + 1;
                    const int k = c / h, l = c % h;
                    shift_in_greater(dA[w], refA, smp[rr + k][l]);
                    shift_in_greater(dB[w], refB, smp[rr + k][l + 1]);
                }
            }
        }
        uint32_t *oA = words + ((int64_t)i * Wo + jA) * n_out;
#pragma unroll
        for (int w = 0; w < NWRITTEN; w++) {
            oA[w] = round_target ? round_word_through_float(dA[w], round_target) : dA[w];
            if (hasB) oA[n_out + w] = round_target ? round_word_through_float(dB[w], round_target) : dB[w];
        }
        for (int w = NWRITTEN; w < n_out; w++) { // rule E1
            oA[w] = 0;
            if (hasB) oA[n_out + w] = 0;
        }
    }
}
template <int HR, int VR, int ROWS, int TJN>
static void launch_census_grey_as(svh_context *ctx, const CensusJob &a, const CensusJob *b, int pl, int pt, int n_out) {
    dim3 grid(ceil_div(std::max(a.Wo, b ? b->Wo : 0), 2 * TJN), ceil_div(std::max(a.Ho, b ? b->Ho : 0), ROWS), b ? 2 : 1);
    SVH_LAUNCH(ctx, "census_transform", (census_grey_kernel<HR, VR, ROWS, TJN>), grid, TJN, 0, a, b ? *b : a, pl, pt, n_out);
}
template <int HR, int VR>
static void launch_census_grey(svh_context *ctx, const CensusJob &a, const CensusJob *b, int pl, int pt, int n_out) {
    // measured at 1080p, 9x9 (MI355X): 1 / 2 / 3 / 4 rows x 128 / 256 lanes all land within 31-34 us; 4 rows x 128 lanes was the
    // fastest.  11x11 windows keep two rows (register budget: (2 VR + ROWS) x (2 HR + 2) samples per lane)
    if (HR <= 4) launch_census_grey_as<HR, VR, 4, 128>(ctx, a, b, pl, pt, n_out);
    else launch_census_grey_as<HR, VR, 2, 128>(ctx, a, b, pl, pt, n_out);
}

static bool census_grey_dispatch(svh_context *ctx, int h_r, int v_r, const CensusJob &a, const CensusJob *b, int pl, int pt, int n_out) {
    // square windows 7x7 - 11x11 and (round 5) the rectangles with half-widths 2 .. 5 on either side (5x7 ... 11x9: the general tiled kernel
    // took two to three times as long for them)
#define SVH_CG(HRV, VRV)                                       \
    if (h_r == HRV && v_r == VRV) {                            \
        launch_census_grey<HRV, VRV>(ctx, a, b, pl, pt, n_out); \
        return true;                                           \
    }
    SVH_CG(3, 3) SVH_CG(4, 4) SVH_CG(5, 5)
    SVH_CG(2, 3) SVH_CG(3, 2) SVH_CG(2, 4) SVH_CG(4, 2) SVH_CG(2, 5) SVH_CG(5, 2)
    SVH_CG(3, 4) SVH_CG(4, 3) SVH_CG(3, 5) SVH_CG(5, 3) SVH_CG(4, 5) SVH_CG(5, 4)
#undef SVH_CG
    return false;
}

// Compile-time windows the register-blocked grey kernel does not take -- colour images (the channels of a window row are the next
// floats of the image row: unfold.h:180) and grey windows 13 / 15 wide: ROWS output rows per block from one staged tile of
// (2 VR + ROWS) rows, a pixel per lane (lanes C floats apart: conflict free for C = 1 and 3), every LDS read at an immediate offset, one
// v_cmp + v_addc per bit, a tile row read once for all the output rows it serves.  (The general tiled kernel below stages a whole window of rows per output row and spends six instructions of
// bookkeeping per bit: RGB 5x5 / 7x7 / 9x9 at 1080p, both images: 0.109 / 0.165 / 0.265 ms.)
template <int HR, int VR, int C, int ROWS>
__global__ void __launch_bounds__(256) census_fixed_kernel(CensusJob job0, CensusJob job1, int pl, int pt, int n_out) {
    const CensusJob &job = blockIdx.z == 0 ? job0 : job1;
    const float *__restrict__ img = job.img;
    uint32_t *__restrict__ words = job.words;
    const int H = job.H, W = job.W, Ho = job.Ho, Wo = job.Wo;
    const int round_target = job.round_target;
    constexpr int h = 2 * HR + 1, v = 2 * VR + 1, TR = v + ROWS - 1, TW = (256 + h - 1) * C, HC = h * C;
    constexpr int NWRITTEN = (h * v * C - 1) / 32;
    static_assert(NWRITTEN >= 1 && (size_t)TR * TW * sizeof(float) <= 60 * 1024, "");
    __shared__ float tile[TR * TW];
    const int i0 = blockIdx.y * ROWS, j0 = blockIdx.x * 256, tj = threadIdx.x;
    if (i0 >= Ho || j0 >= Wo) return; // the grid covers the larger image
    // stage the tile: every load of the thread is issued before the first LDS write (a block pays one memory latency)
    constexpr int PER_ROW = (TW + 255) / 256;
    {
        float r[TR][PER_ROW];
#pragma unroll
        for (int k = 0; k < TR; k++) {
            const int ii = i0 - pt + k;
            const bool row_in = ii >= 0 && ii < H;
            const float *row = img + ((int64_t)ii * W + (j0 - pl)) * C;
#pragma unroll
            for (int q = 0; q < PER_ROW; q++) {
                const int e = tj + q * 256, jj = j0 - pl + e / C;
                r[k][q] = (row_in && e < TW && jj >= 0 && jj < W) ? row[e] : 0.0f;
            }
        }
#pragma unroll
        for (int k = 0; k < TR; k++) {
#pragma unroll
            for (int q = 0; q < PER_ROW; q++) {
                const int e = tj + q * 256;
                if (e < TW) tile[k * TW + e] = r[k][q];
            }
        }
    }
    __syncthreads();
    const int j = j0 + tj;
    if (j >= Wo) return;
    const float *tp = tile + tj * C;
    // The ROWS output rows advance together over the tile rows, from the bottom one up: a tile row's h C samples are read once and serve
    // every output row whose window holds it (four times fewer LDS reads than a walk per output row).  Bottom-up and right-to-left is the
    // channel index descending -- the order in which word = 2 word + bit fills a word from bit 31 down (census.h:98-108: channel
    // c = 32 w + b + 1 sits behind bit b of word w; window row c / (h C), then column, then channel; channel 0 is the reference sample).
    float ref[ROWS];
    uint32_t d[ROWS][NWRITTEN];
#pragma unroll
    for (int rr = 0; rr < ROWS; rr++) {
        ref[rr] = tp[rr * TW]; // the window's top-left sample, channel 0 (finding F6)
#pragma unroll
        for (int w = 0; w < NWRITTEN; w++) d[rr][w] = 0;
    }
#pragma unroll
    for (int k = TR - 1; k >= 0; k--) {
        float smp[HC];
#pragma unroll
        for (int q = 0; q < HC; q++) smp[q] = tp[k * TW + q];
#pragma unroll
        for (int rr = 0; rr < ROWS; rr++) {
            const int wr = k - rr; // the window row of output row rr that tile row k is
            if (wr >= 0 && wr < v) {
#pragma unroll
                for (int rem = HC - 1; rem >= 0; rem--) {
                    const int c = wr * HC + rem;
                    if (c >= 1 && c <= 32 * NWRITTEN) shift_in_greater(d[rr][(c - 1) / 32], ref[rr], smp[rem]);
                }
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < ROWS; rr++) {
        const int i = i0 + rr;
        if (i >= Ho) break;
        uint32_t *o = words + ((int64_t)i * Wo + j) * n_out;
#pragma unroll
        for (int w = 0; w < NWRITTEN; w++) o[w] = round_target ? round_word_through_float(d[rr][w], round_target) : d[rr][w];
        for (int w = NWRITTEN; w < n_out; w++) o[w] = 0; // rule E1
    }
}
template <int HR, int VR, int C, int ROWS>
static void launch_census_fixed(svh_context *ctx, const CensusJob &a, const CensusJob *b, int pl, int pt, int n_out) {
    dim3 grid(ceil_div(std::max(a.Wo, b ? b->Wo : 0), 256), ceil_div(std::max(a.Ho, b ? b->Ho : 0), ROWS), b ? 2 : 1);
    SVH_LAUNCH(ctx, "census_transform", (census_fixed_kernel<HR, VR, C, ROWS>), grid, 256, 0, a, b ? *b : a, pl, pt, n_out);
}
static bool census_fixed_dispatch(svh_context *ctx, int h_r, int v_r, int C, const CensusJob &a, const CensusJob *b, int pl, int pt, int n_out) {
    if (h_r != v_r) return false;
    if (C == 3) {
        switch (h_r) {
        case 2: launch_census_fixed<2, 2, 3, 4>(ctx, a, b, pl, pt, n_out); return true;
        case 3: launch_census_fixed<3, 3, 3, 4>(ctx, a, b, pl, pt, n_out); return true;
        case 4: launch_census_fixed<4, 4, 3, 4>(ctx, a, b, pl, pt, n_out); return true;
        default: return false;
        }
    }
    if (C == 1) {
        switch (h_r) {
        case 6: launch_census_fixed<6, 6, 1, 4>(ctx, a, b, pl, pt, n_out); return true;
        case 7: launch_census_fixed<7, 7, 1, 4>(ctx, a, b, pl, pt, n_out); return true;
        default: return false;
        }
    }
    return false;
}

__global__ void census_features_kernel(const float *__restrict__ feat, int64_t npx, int F, int n_out, int n_written,
                                       int round_target, uint32_t *__restrict__ words) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npx; p += (int64_t)gridDim.x * blockDim.x) {
        const float *f = feat + p * F;
        uint32_t *o = words + p * n_out;
        const float ref = f[0];
        for (int w = 0; w < n_written; w++) {
            uint32_t d = 0;
            for (int b = 0; b < 32; b++) d |= (ref > f[1 + 32 * w + b] ? 1u : 0u) << b;
            o[w] = round_target ? round_word_through_float(d, round_target) : d;
        }
        for (int w = n_written; w < n_out; w++) o[w] = 0;
    }
}

int dev_unfold(svh_context *ctx, ImageDesc img, int h_r, int v_r, int pl, int pt, int Ho, int Wo, float *out) {
    int64_t n = (int64_t)Ho * Wo * (2 * h_r + 1) * (2 * v_r + 1) * img.C;
    if (n == 0) return SVH_OK;
    SVH_LAUNCH(ctx, "unfold", unfold_kernel, grid_for(n, 256, 16384), 256, 0, img.data, img.H, img.W, img.C, h_r, v_r, pl, pt, Ho,
               Wo, out);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

// kernel argument for rule E2: 0 none, 1 saturate, 2 zero (round_word_through_float)
static inline int round_mode(const svh_context *ctx, bool round_through_float) { return round_through_float ? (ctx->census_float_overflow ? 2 : 1) : 0; }

int dev_census_from_image(svh_context *ctx, ImageDesc img, int h_r, int v_r, int pl, int pt, int Ho, int Wo, int n_out,
                          bool round_through_float, uint32_t *words) {
    int64_t npx = (int64_t)Ho * Wo;
    if (npx == 0 || n_out == 0) return SVH_OK;
    int F = (2 * h_r + 1) * (2 * v_r + 1) * img.C;
    if (img.C == 1) {
        CensusJob job{img.data, img.H, img.W, Ho, Wo, round_mode(ctx, round_through_float), words};
        if (census_grey_dispatch(ctx, h_r, v_r, job, nullptr, pl, pt, n_out)) {
            SVH_CHECK_LAUNCH(ctx);
            return SVH_OK;
        }
    }
    {
        CensusJob job{img.data, img.H, img.W, Ho, Wo, round_mode(ctx, round_through_float), words};
        if (census_fixed_dispatch(ctx, h_r, v_r, img.C, job, nullptr, pl, pt, n_out)) {
            SVH_CHECK_LAUNCH(ctx);
            return SVH_OK;
        }
    }
    const size_t tile_bytes = (size_t)(2 * v_r + 1) * (CENSUS_TJ + 2 * h_r) * img.C * sizeof(float);
    if (tile_bytes <= 60 * 1024 && census_words_written(F) > 0) {
        dim3 grid(ceil_div(Wo, CENSUS_TJ), Ho);
        SVH_LAUNCH(ctx, "census_transform", census_image_tiled_kernel, grid, CENSUS_TJ, tile_bytes, img.data, img.H, img.W, img.C, h_r, v_r,
                   pl, pt, Ho, Wo, n_out, census_words_written(F), round_mode(ctx, round_through_float), words);
        SVH_CHECK_LAUNCH(ctx);
        return SVH_OK;
    }
    SVH_LAUNCH(ctx, "census_transform", census_image_kernel, grid_for(npx, 256, 16384), 256, 0, img.data, img.H, img.W, img.C, h_r,
               v_r, pl, pt, Ho, Wo, n_out, census_words_written(F), round_mode(ctx, round_through_float), words);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

// compact words of both images of a pair (source exact, target rounded through float: rule E2), auto padding; one launch
// for the common grey windows
int dev_census_pair_compact(svh_context *ctx, ImageDesc src, ImageDesc tgt, int h_r, int v_r, int nWw, uint32_t *sw, uint32_t *tw) {
    if (nWw == 0) return SVH_OK;
    if (src.C == 1 && tgt.C == 1) {
        CensusJob a{src.data, src.H, src.W, src.H, src.W, 0, sw}, b{tgt.data, tgt.H, tgt.W, tgt.H, tgt.W, round_mode(ctx, true), tw};
        if (census_grey_dispatch(ctx, h_r, v_r, a, &b, h_r, v_r, nWw)) {
            SVH_CHECK_LAUNCH(ctx);
            return SVH_OK;
        }
    }
    if (src.C == tgt.C) { // both images in one launch
        CensusJob a{src.data, src.H, src.W, src.H, src.W, 0, sw}, b{tgt.data, tgt.H, tgt.W, tgt.H, tgt.W, round_mode(ctx, true), tw};
        if (census_fixed_dispatch(ctx, h_r, v_r, src.C, a, &b, h_r, v_r, nWw)) {
            SVH_CHECK_LAUNCH(ctx);
            return SVH_OK;
        }
    }
    SVH_TRY(dev_census_from_image(ctx, src, h_r, v_r, h_r, v_r, src.H, src.W, nWw, false, sw));
    return dev_census_from_image(ctx, tgt, h_r, v_r, h_r, v_r, tgt.H, tgt.W, nWw, true, tw);
}

int dev_census_from_features(svh_context *ctx, const float *feat, int H, int W, int F, int n_out, bool round_through_float,
                             uint32_t *words) {
    int64_t npx = (int64_t)H * W;
    if (npx == 0 || n_out == 0) return SVH_OK;
    SVH_LAUNCH(ctx, "census_features", census_features_kernel, grid_for(npx, 256, 16384), 256, 0, feat, npx, F, n_out,
               census_words_written(F), round_mode(ctx, round_through_float), words);
    SVH_CHECK_LAUNCH(ctx);
    return SVH_OK;
}

} // namespace svh

using namespace svh;

static int image_desc(svh_context *ctx, const svh_array *img, const char *what, int *H, int *W, int *C) {
    SVH_TRY(validate_image(ctx, img, what, -1));
    *H = (int)img->shape[0];
    *W = (int)img->shape[1];
    *C = img->ndim == 3 ? (int)img->shape[2] : 1;
    return SVH_OK;
}

static void unfold_geometry(int H, int W, int C, int h_r, int v_r, const int32_t pad[4], int *pl, int *pt, int *Ho, int *Wo,
                            int *F) {
    // correlation/unfold.h:256-270
    int l = pad ? pad[0] : h_r, t = pad ? pad[1] : v_r, r = pad ? pad[2] : h_r, b = pad ? pad[3] : v_r;
    int h = 2 * h_r + 1, v = 2 * v_r + 1;
    *pl = l;
    *pt = t;
    *Ho = H - v + t + b + 1;
    *Wo = W - h + l + r + 1;
    *F = h * v * C;
}

extern "C" {

int svh_unfold_shape(const svh_array *img, int h_radius, int v_radius, const int32_t pad[4], int64_t out_shape[3]) {
    if (!img || !out_shape || img->ndim < 2 || img->ndim > 3 || h_radius < 0 || v_radius < 0) return SVH_ERR_INVALID_ARGUMENT;
    int pl, pt, Ho, Wo, F;
    unfold_geometry((int)img->shape[0], (int)img->shape[1], img->ndim == 3 ? (int)img->shape[2] : 1, h_radius, v_radius, pad, &pl,
                    &pt, &Ho, &Wo, &F);
    out_shape[0] = Ho;
    out_shape[1] = Wo;
    out_shape[2] = F;
    return SVH_OK;
}

int svh_unfold(svh_context *ctx, const svh_array *img, int h_radius, int v_radius, const int32_t pad[4], svh_array *out) {
    return svh_unfold_oriented(ctx, img, h_radius, v_radius, pad, SVH_ROTATE0, out);
}

int svh_unfold_oriented(svh_context *ctx, const svh_array *img, int h_radius, int v_radius, const int32_t pad[4], int orientation, svh_array *out) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    if (orientation < SVH_ROTATE0 || orientation > SVH_ROTATE270) return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "bad patch orientation");
    int H, W, C;
    SVH_TRY(image_desc(ctx, img, "img", &H, &W, &C));
    SVH_TRY(validate(ctx, out, "out", SVH_F32, 3, 3));
    if (h_radius < 0 || v_radius < 0 || h_radius > 255 || v_radius > 255)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "radii must be in [0,255] (uint8_t in the reference)");
    int pl, pt, Ho, Wo, F;
    unfold_geometry(H, W, C, h_radius, v_radius, pad, &pl, &pt, &Ho, &Wo, &F);
    if (Ho <= 0 || Wo <= 0) return fail(ctx, SVH_EMPTY_RESULT, "unfold output is empty");
    if (out->shape[0] != Ho || out->shape[1] != Wo || out->shape[2] != F)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "out must have shape (%d,%d,%d)", Ho, Wo, F);
    Scratch scr(ctx);
    void *dimg;
    OutStage os;
    SVH_TRY(stage_image(ctx, scr, *img, &dimg));
    SVH_TRY(stage_out(ctx, scr, *out, &os));
    if (orientation == SVH_ROTATE0) {
        SVH_TRY(dev_unfold(ctx, {(const float *)dimg, H, W, C}, h_radius, v_radius, pl, pt, Ho, Wo, (float *)os.dptr));
    } else {
        SVH_LAUNCH(ctx, "unfold", unfold_oriented_kernel, grid_for((int64_t)Ho * Wo * F, 256, 16384), 256, 0, (const float *)dimg, H, W, C, h_radius, v_radius,
                   pl, pt, Ho, Wo, orientation, (float *)os.dptr);
        SVH_CHECK_LAUNCH(ctx);
    }
    return finish_out(ctx, os);
}

int svh_census_features(svh_context *ctx, const svh_array *feat, svh_array *words) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    SVH_TRY(validate(ctx, feat, "feat", SVH_F32, 3, 3));
    SVH_TRY(validate(ctx, words, "words", SVH_U32, 3, 3));
    int H = (int)feat->shape[0], W = (int)feat->shape[1], F = (int)feat->shape[2];
    if (F <= 1) return fail(ctx, SVH_EMPTY_RESULT, "census needs at least two feature channels (census.h:76-78)");
    int nW = census_words(F);
    if (words->shape[0] != H || words->shape[1] != W || words->shape[2] != nW)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "words must have shape (%d,%d,%d)", H, W, nW);
    Scratch scr(ctx);
    void *df;
    OutStage os;
    SVH_TRY(stage_in(ctx, scr, *feat, &df));
    SVH_TRY(stage_out(ctx, scr, *words, &os));
    SVH_TRY(dev_census_from_features(ctx, (const float *)df, H, W, F, nW, false, (uint32_t *)os.dptr));
    return finish_out(ctx, os);
}

int svh_census_transform(svh_context *ctx, const svh_array *img, int h_radius, int v_radius, const int32_t pad[4],
                         svh_array *words) {
    if (!ctx) return SVH_ERR_INVALID_ARGUMENT;
    int H, W, C;
    SVH_TRY(image_desc(ctx, img, "img", &H, &W, &C));
    SVH_TRY(validate(ctx, words, "words", SVH_U32, 3, 3));
    if (h_radius < 0 || v_radius < 0 || h_radius > 127 || v_radius > 127)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "radii must be in [0,127] (int8_t in the reference)");
    int pl, pt, Ho, Wo, F;
    unfold_geometry(H, W, C, h_radius, v_radius, pad, &pl, &pt, &Ho, &Wo, &F);
    if (Ho <= 0 || Wo <= 0 || F <= 1) return fail(ctx, SVH_EMPTY_RESULT, "census transform output is empty");
    int nW = census_words(F);
    if (words->shape[0] != Ho || words->shape[1] != Wo || words->shape[2] != nW)
        return fail(ctx, SVH_ERR_INVALID_ARGUMENT, "words must have shape (%d,%d,%d)", Ho, Wo, nW);
    Scratch scr(ctx);
    void *dimg;
    OutStage os;
    SVH_TRY(stage_image(ctx, scr, *img, &dimg));
    SVH_TRY(stage_out(ctx, scr, *words, &os));
    SVH_TRY(dev_census_from_image(ctx, {(const float *)dimg, H, W, C}, h_radius, v_radius, pl, pt, Ho, Wo, nW, false,
                                  (uint32_t *)os.dptr));
    return finish_out(ctx, os);
}

} // extern "C"

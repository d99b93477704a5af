/*
 * stevi_hip.h -- C ABI of libstevi_hip.so: the MI355X (gfx950) implementation of LibStevi's
 * correlation/ hot path (cost-volume construction, SGM aggregation, winner extraction, cost-based
 * sub-pixel refinement).
 *
 * The reference has no FFI layer: the path is a set of header-only function templates in
 * namespace StereoVision::Correlation.  Each entry point below replaces one of those templates and
 * cites it; libstevi_amd/include/ holds same-named C++ headers whose bodies marshal
 * Multidim::Array arguments into these calls.
 *
 * Conventions
 *   - every array argument is an svh_array: base pointer, dtype, memory space, shape and strides
 *     (strides in ELEMENTS, may be any positive layout).  No ownership is ever taken.
 *   - index order follows the reference: images (row, col[, channel]), feature volumes
 *     (row, col, feature), cost volumes (row, col, disparity index), maps (row, col).
 *   - SVH_HOST arrays are copied in/out synchronously; SVH_DEVICE arrays are used in place, the work
 *     is enqueued on the context's stream and the call returns without synchronising.
 *   - the kernels' native layout is "last index fastest, dense"; other layouts go through a
 *     relayout kernel (device) or a strided copy (host).
 *   - every function returns an svh_status.  SVH_EMPTY_RESULT marks the situations in which the
 *     reference returns an empty Multidim::Array (shape mismatch etc.); outputs are untouched.
 *   - there is no CPU fallback: without a usable HIP device every compute call fails with
 *     SVH_ERR_NO_DEVICE / SVH_ERR_HIP.
 */
#ifndef STEVI_HIP_H
#define STEVI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVH_VERSION_MAJOR 0
#define SVH_VERSION_MINOR 1

typedef struct svh_context svh_context;

typedef enum svh_status {
    SVH_OK = 0,
    SVH_EMPTY_RESULT = 1,        /* the reference would return an empty array */
    SVH_ERR_INVALID_ARGUMENT = 2,
    SVH_ERR_UNSUPPORTED = 3,     /* valid in the reference, not implemented on the GPU path */
    SVH_ERR_NO_DEVICE = 4,
    SVH_ERR_HIP = 5,
    SVH_ERR_OUT_OF_MEMORY = 6
} svh_status;

typedef enum svh_memspace { SVH_HOST = 0, SVH_DEVICE = 1 } svh_memspace;

/* Element types.  Images are SVH_F32 or SVH_U8 (the reference's two image types): SVH_U8 is accepted by svh_unfold*,
 * svh_census_transform, svh_unfold_cost_volume(_2d), svh_stereo_match and the shard calls for the functions whose features are
 * the plain samples (CENSUS, HAMMING, CC, SSD, SAD -- matching_costs.h:749-783: every comparison is formed after a cast to
 * float, so the bytes are widened once on the device); with a normalised or zero-mean function it is SVH_ERR_UNSUPPORTED (the
 * reference's int16 path, skipped by its own test, testCorrelationFilters.cpp:1249). */
typedef enum svh_dtype { SVH_F32 = 0, SVH_I32 = 1, SVH_U32 = 2, SVH_U8 = 3, SVH_U64 = 4, SVH_I16 = 5, SVH_U16 = 6 } svh_dtype;

/* values of StereoVision::Correlation::matchingFunctions, correlation/matching_costs.h:38-53 */
typedef enum svh_match_func {
    SVH_CC = 0, SVH_NCC = 1, SVH_SSD = 2, SVH_SAD = 3, SVH_ZCC = 4, SVH_ZNCC = 5, SVH_ZSSD = 6, SVH_ZSAD = 7,
    SVH_HAMMING = 10, SVH_CENSUS = 11
} svh_match_func;

/* dispExtractionStartegy / dispDirection / truncatedCostVolumeDirection, correlation/correlation_base.h:31-45 */
typedef enum svh_strategy { SVH_COST = 0, SVH_SCORE = 1 } svh_strategy;
typedef enum svh_disp_direction { SVH_LEFT_TO_RIGHT = 0, SVH_RIGHT_TO_LEFT = 1 } svh_disp_direction;
typedef enum svh_tcv_direction { SVH_TCV_SAME = 0, SVH_TCV_REVERSED = 1, SVH_TCV_BOTH = 2 } svh_tcv_direction;

/* InterpolationKernel, correlation/cost_based_refinement.h:30-35 */
typedef enum svh_interp_kernel { SVH_EQUIANGULAR = 0, SVH_PARABOLA = 1, SVH_GAUSSIAN = 2 } svh_interp_kernel;

#define SVH_MAX_DIMS 4

typedef struct svh_array {
    void *data;
    int32_t ndim;
    int32_t dtype;    /* svh_dtype */
    int32_t memspace; /* svh_memspace */
    int32_t reserved;
    int64_t shape[SVH_MAX_DIMS];
    int64_t strides[SVH_MAX_DIMS]; /* in elements */
} svh_array;

/* ---- context ------------------------------------------------------------------------------------ */

/* Creates a context bound to HIP device `device` (-1 = current device).  `stream` is the hipStream_t to
 * enqueue on (e.g. the caller's / PyTorch's current stream); NULL is the device's default (null) stream. */
int svh_context_create(svh_context **ctx, int device, void *stream);
int svh_context_destroy(svh_context *ctx);
int svh_context_set_stream(svh_context *ctx, void *stream);
int svh_context_synchronize(svh_context *ctx);
/* frees the cached device workspace (it is otherwise kept between calls to stay out of hipMalloc) */
int svh_context_trim(svh_context *ctx);
const char *svh_status_string(int status);
/* message of the last failing call on this context ("" if none) */
const char *svh_last_error(const svh_context *ctx);
/* Options of a context: the five choices a caller has a reason to make.  (Every other switch of the library exists so that the parity
 * tests can run both sides of an A/B; those live behind svh_test_set_option in include/stevi_hip_test.h and are no part of the product
 * surface.)  Unknown names and values return SVH_ERR_INVALID_ARGUMENT.
 * "census_float_overflow" (default 0): what becomes of a target census word that rounds to 2^32 on its way through `float`
 *   (cross_correlations.h:235-236; words >= 0xFFFFFF80; undefined in C++): 0 = 0xFFFFFFFF, what the reference's Release build gives on
 *   a host with AVX-512 and what the GPU's own conversion does; 1 = 0, what x86-64 code generation without AVX-512 gives (the reference's
 *   -mavx -mavx2 -mfma flags, every Debug build).  Smooth image gradients produce such words; random textures almost never.
 * "census_winner_shortcut" (default 1): in the integer-exact regime of the census + SGM Cost-branch pipeline the winning disparity of a
 *   pixel does not depend on the per-pass minima the reference hands along its lines (they shift every disparity of the pixel alike), so
 *   calls that ask for index / disparity maps only skip the line scans; 0 runs them regardless (same maps; bench.py times that form).
 * "census_sweep" (default 0 = automatic): engine of the voxel sweep of the fused census pipeline, 1 = the vector-ALU kernel
 *   (xor + popcount), 3 = the matrix-core kernels with FP4 operands (Hamming distance as a dot product; at most 8 census words,
 *   up to 992 disparities); automatic = 3 where it applies.  Same keys bit for bit.
 * "sgm_score_fused" (default 1): how the Score branch of svh_sgm_cost_volume runs its four downward passes (8 directions, whole image,
 *   P2 >= P1 >= 0, up to 512 disparities; anything else takes a launch per pass).  2: one sweep of the volume, a launch per band of
 *   rows, each block recomputing the lines that enter its strip of columns (23 instead of 44 bytes per voxel over all passes).
 *   0: one read-modify-write sweep per pass.  1: whichever a model of the two predicts faster (small images at moderate ranges run
 *   faster pass by pass, large volumes in bands).  Same bits in all of them.
 * "literal_cost_volumes" (default 0): 1 makes svh_hierarchical_truncated_cost_volume build its coarsest cost volume with the per-voxel
 *   kernel (the reference's operations in the reference's order) instead of the register-blocked one: estimates bit-identical to the
 *   reference's own arithmetic instead of within its 1e-4. */
int svh_context_set_option(svh_context *ctx, const char *name, int value);
/* 1 when a HIP device is visible, 0 otherwise; never fails */
int svh_device_available(void);

/* Device memory for callers that keep arrays on the GPU between calls without a GPU runtime of their own (the C++ drop-in headers'
 * HipBridge::DeviceArray: unfoldBasedCostVolume -> sgmCostVolume -> extractSelectedIndex written with the reference's names crosses
 * PCIe once per image and once for the disparity map instead of once per volume).  Memory from svh_device_alloc is passed back in
 * svh_array descriptors with memspace = SVH_DEVICE; upload / download are synchronous on the context's stream.  The reference has
 * no counterpart (its arrays live in host memory: Multidim::Array). */
int svh_device_alloc(svh_context *ctx, size_t bytes, void **ptr);
int svh_device_free(svh_context *ctx, void *ptr);
/* The HIP device a context is bound to, and a release that needs no context: memory from svh_device_alloc may outlive the context (and
 * the thread) that allocated it -- a worker thread hands its result array to another thread and exits -- so the owner of such an
 * array frees it by device number.  Waits for the device to go idle first (nothing still running may use the memory). */
int svh_context_get_device(const svh_context *ctx);
int svh_device_free_detached(int device, void *ptr);
/* Blocks released through svh_device_free / svh_device_free_detached are kept per device and handed out again by svh_device_alloc (best
 * fit, at most twice the request): a chain written with the reference's names allocates and releases a volume per call, and hipMalloc /
 * hipFree of 2 GB cost tens of milliseconds each.  The cache holds at most SVH_DEVICE_CACHE_MB (environment; default: a quarter of the device's
 * memory) and is returned to the device by this call, by svh_context_trim and by every allocation of the library that would otherwise
 * fail for lack of memory.  Releasing a block twice returns SVH_ERR_INVALID_ARGUMENT. */
int svh_device_cache_trim(int device);
int svh_device_upload(svh_context *ctx, void *device_dst, const void *host_src, size_t bytes);
int svh_device_download(svh_context *ctx, void *host_dst, const void *device_src, size_t bytes);

/* Page-locked host memory for the arrays that cross PCIe.  Every function of the reference takes host arrays and returns a fresh host
 * array (cross_correlations.h:741-745, sgm.h:361-365; its benchmark chain benchmarkCrossCorrelationAlgorithms.cpp:288-294 hands each
 * 2.1 GB volume to the next function), so a chain written with its names moves each volume over the link twice.  A host array in a
 * block from svh_host_alloc is written and read by the DMA engines directly, at the link's rate; the drop-in headers and the Python
 * mirror allocate the results they return there (>= 1 MB), so that the volumes of such a chain never pass through pageable memory.
 * Host arrays in any other memory still work: large ones are copied in chunks through a page-locked ring by a few host threads
 * (SVH_COPY_THREADS, default 4) while the DMA engines move the previous chunks.  Nothing is cached by content: each call transfers
 * what the caller's array holds.  Released blocks are kept page-locked (best fit, at most twice the request) up to SVH_HOST_CACHE_MB
 * (default 8192); svh_host_cache_trim gives them back.  svh_host_free returns SVH_ERR_INVALID_ARGUMENT for a pointer that did not
 * come from svh_host_alloc or was released already.  The reference has no counterpart (Multidim::Array allocates with new[]). */
int svh_host_alloc(size_t bytes, void **ptr);
int svh_host_free(void *ptr);
int svh_host_cache_trim(void);
/* 1 when [ptr, ptr + bytes) lies in page-locked memory (a block of svh_host_alloc, or any memory the HIP runtime has registered) */
int svh_host_is_pinned(const void *ptr, size_t bytes);
/* device -> device, enqueued on the context's stream (no wait) */
int svh_device_copy(svh_context *ctx, void *device_dst, const void *device_src, size_t bytes);

/* Per-kernel timing with hipEvents on the context's stream.  While enabled, every kernel launch is
 * bracketed by two events; svh_profile_collect() synchronises and folds them into per-kernel totals. */
int svh_profile_enable(svh_context *ctx, int enable);
/* restrict the event bracketing to launches of one kernel (NULL or "" = every kernel): two events per launch cost a few
 * microseconds of stream time each, which matters for sub-millisecond steps */
int svh_profile_filter(svh_context *ctx, const char *kernel_name);
/* bracket only every `every`-th launch that passes the filter (1 = every launch): an event pair costs about 5 us of stream time,
 * a tenth of a 0.1 ms step when every launch of one kernel is bracketed */
int svh_profile_sampling(svh_context *ctx, int every);
int svh_profile_reset(svh_context *ctx);
int svh_profile_collect(svh_context *ctx);
int svh_profile_count(const svh_context *ctx);
int svh_profile_get(const svh_context *ctx, int k, char *name, size_t name_len, double *total_ms, int64_t *launches);

/* ---- A1  unfold<T_I,T_O>(h_radius, v_radius, img, padding)            correlation/unfold.h:247-344
 * img (H,W) or (H,W,C) f32 -> out (Ho,Wo,F) f32, F=(2h_r+1)(2v_r+1)C, channel c = C(2h_r+1)k + C l + ch.
 * pad = {left, top, right, bottom} or NULL for PaddingMargins() "auto" = (h_r, v_r).  svh_unfold is Rotate0; svh_unfold_oriented
 * places sample (k, l, ch) at channelFromCord(k, l, ch, h, v, C, orientation) (unfold.h:139-191: the patch rotated by 0 / 90 /
 * 180 / 270 degrees). */
typedef enum svh_patch_orientation { SVH_ROTATE0 = 0, SVH_ROTATE90 = 1, SVH_ROTATE180 = 2, SVH_ROTATE270 = 3 } svh_patch_orientation;
int svh_unfold(svh_context *ctx, const svh_array *img, int h_radius, int v_radius, const int32_t pad[4], svh_array *out);
int svh_unfold_oriented(svh_context *ctx, const svh_array *img, int h_radius, int v_radius, const int32_t pad[4], int orientation, svh_array *out);
int svh_unfold_shape(const svh_array *img, int h_radius, int v_radius, const int32_t pad[4], int64_t out_shape[3]);

/* ---- UnFoldCompressor features: SURVEY.md section 8(f) rank 4 -----------------------------------------------------------
 * unfold(UnFoldCompressor(mask), img, padding)                             correlation/unfold.h:47-121, :346-471
 * mask: HOST array of mask_h x mask_w int labels (row-major); every positive label is a superpixel of the window, feature f is
 * the mean of the image over the f-th smallest label's pixels.  img (H,W[,C]) f32 -> out (Ho,Wo,C*nFeatures) f32 with the
 * channel-major feature index in_c * nFeatures + f (:455); pad = {left, top, right, bottom} or NULL for the compressor's own
 * margins.  Feed the result to svh_feature_cost_volume[_2d] for the compressor overloads of unfoldBasedCostVolume /
 * unfoldBased2dDisparityCostVolume (cross_correlations.h:767-791, :824-851). */
int svh_unfold_compressed(svh_context *ctx, const svh_array *img, const int32_t *mask, int mask_h, int mask_w, const int32_t pad[4],
                          svh_array *out);
int svh_unfold_compressed_shape(const svh_array *img, const int32_t *mask, int mask_h, int mask_w, const int32_t pad[4], int64_t out_shape[3]);

/* ---- A2  censusFeatures(features)                                      correlation/census.h:69-115
 * feat (H,W,F) f32 -> words (H,W,nW) u32, nW=(F-1)/32+1; trailing partial word left 0.  F<=1 -> SVH_EMPTY_RESULT. */
int svh_census_features(svh_context *ctx, const svh_array *feat, svh_array *words);

/* ---- A3  censusTransform2D(img, h_radius, v_radius, padding)           correlation/census.h:117-131 */
int svh_census_transform(svh_context *ctx, const svh_array *img, int h_radius, int v_radius, const int32_t pad[4],
                         svh_array *words);

/* ---- A6/A8  featureVolume2CostVolume<matchFunc,...,dDir,float>(feat_l, feat_r, range)
 *                                                       correlation/cross_correlations.h:724-738 (+ :194-308, :645-722)
 * feat_l (H,Wl,F), feat_r (H,Wr,F) f32 -> cv (H,Ws,D) f32, Ws = source width (right image for RightToLeft).
 * disp_lower = 0 reproduces the disp_t overload; otherwise searchOffset<1>(disp_lower, disp_lower+D-1). */
int svh_feature_cost_volume(svh_context *ctx, int match_func, int disp_direction, const svh_array *feat_l,
                            const svh_array *feat_r, int32_t disp_lower, int32_t disp_count, svh_array *cv);

/* ---- A8  unfoldBasedCostVolume<matchFunc,...>(img_l, img_r, h_radius, v_radius, disp_width)
 *                                                       correlation/cross_correlations.h:740-765
 * Same result as unfold + featureVolume2CostVolume without materialising the unfolded volumes. */
int svh_unfold_cost_volume(svh_context *ctx, int match_func, int disp_direction, const svh_array *img_l,
                           const svh_array *img_r, int h_radius, int v_radius, int32_t disp_lower, int32_t disp_count,
                           svh_array *cv);

/* The same with a by-product for a later sgmCostVolume<.., Cost> on the volume: minima (H,Ws,2) f32 = per pixel the smallest cost among
 * the disparities d with j + d < Ws and among those with j + d >= Ws (the ones sgm.h:287-289 charges Pout; +inf where there are none).
 * *minima_written says what was produced: 0 nothing (the array is left untouched); 1 census / Hamming -- integer costs, the minima of ALL
 * costs; 2 a Cost-strategy float function on grey images through the column-sum kernel (SAD, SSD, ZSSD; windows up to 11 wide) -- the
 * minima of the FINITE costs, and every finite |c| <= 1e30 (checked on the device; one word comes back to the host, so this form of the
 * call waits for its kernel).  Pass the kind on to svh_sgm_cost_volume_winner; svh_sgm_cost_volume_minima takes kind 1. */
int svh_unfold_cost_volume_minima(svh_context *ctx, int match_func, int disp_direction, const svh_array *img_l, const svh_array *img_r,
                                  int h_radius, int v_radius, int32_t disp_lower, int32_t disp_count, svh_array *cv, svh_array *minima,
                                  int *minima_written);
/* The same with the by-product a later extractSelectedIndex<strategy of match_func> on the volume would scan it for (the reference
 * benchmark's own sequence, benchmarkCrossCorrelationAlgorithms.cpp:92-96): winner (H,Ws) i32 = that index map, picked by the kernel that
 * writes the volume while it holds a pixel's costs.  *winner_written: 1 when it was produced (a float function on grey images through the
 * column-sum kernel: windows up to 11 wide, not ZSAD), 0 when not (the array is left untouched; scan the volume). */
int svh_unfold_cost_volume_winner(svh_context *ctx, int match_func, int disp_direction, const svh_array *img_l, const svh_array *img_r,
                                  int h_radius, int v_radius, int32_t disp_lower, int32_t disp_count, svh_array *cv, svh_array *winner,
                                  int *winner_written);

/* ---- A9  sgmCostVolume<nDirections, strategy>(cv, P1, P2, margins, Pout)      correlation/sgm.h:360-404
 * cv (H,W,D) f32 -> out (H,W,D) f32, D <= 2048.  n_directions 4 or 8 (16 is a data race in the reference and unsupported).
 * margins = {left, top, right, bottom}.  Reproduces the reference as written.
 * T_CV: cv may also be SVH_U8, SVH_I16, SVH_U16, SVH_I32 or SVH_U32 -- the reference casts every cost it reads to float
 * (sgm.h:234, :273, :299; the initial copy :369-377), so the volume is converted once on the device and the float kernels run
 * (same bits; also svh_sgm_cost_volume_textbook).  double volumes are not taken (the reference then subtracts in double). */
int svh_sgm_cost_volume(svh_context *ctx, int n_directions, int strategy, const svh_array *cv, float P1, float P2,
                        const int32_t margins[4], float Pout, svh_array *out);
/* The Cost branch on a volume the caller knows two things about: every entry is an integer with |c| <= max_abs, and `minima` holds its
 * regional minima as svh_unfold_cost_volume_minima writes them.  When a Cost-branch call finds integer costs small enough for every
 * float operation of sgm.h:257-300 to be exact it takes a shortcut (the per-pass minima follow mp' = g - mp from one map g); it
 * normally establishes that -- and g -- with one extra read of the whole volume.  With the statement that read is skipped: same output
 * bit for bit, 4 bytes per voxel less.  The statement is the caller's responsibility: the C++ drop-in keeps it attached to the
 * DeviceArray it came with and drops it at the first non-const access (correlation/stevi_hip_bridge.h), the Python mirror ties it to
 * the tensor's version counter.  Score strategy: minima are ignored. */
int svh_sgm_cost_volume_minima(svh_context *ctx, int n_directions, int strategy, const svh_array *cv, const svh_array *minima, float max_abs,
                               float P1, float P2, const int32_t margins[4], float Pout, svh_array *out);
/* The same (minima may be NULL: then exactly svh_sgm_cost_volume) with a by-product for a later extractSelectedIndex<strategy> on `out`:
 * winner_idx (H,W) i32 = the index that call would return.  The kernel that writes a pixel's final aggregated costs holds them in one
 * wave, so its winner costs a few instructions there against a second read of the whole volume (4 bytes per voxel) later.
 * *winner_written = 1 when the map was produced: always, when winner_idx is given.  (Score strategy: the records of the banded sweep from
 * 256 disparities on -- whole image, 8 directions, P2 >= P1 >= 0, at most 512 disparities, a multiple of 64 --, otherwise a scan of the
 * volume the call has just written, which below 256 disparities is also the faster of the two.)  Like the minima it is a statement about
 * `out` that the caller must not let outlive its contents (DeviceArray keeps it with the storage). */
int svh_sgm_cost_volume_winner(svh_context *ctx, int n_directions, int strategy, const svh_array *cv, const svh_array *minima, int minima_kind,
                               float max_abs, float P1, float P2, const int32_t margins[4], float Pout, svh_array *out, svh_array *winner_idx,
                               int *winner_written);

/* "Textbook" semi-global matching -- NOT the reference's behaviour, an explicit second mode:
 * every one of the 4 / 8 directions traverses every line of the margin box once (the reference skips three directions and half of
 * two more), and the Cost strategy penalises the neighbouring disparities (the reference adds the pixel's own cost there).  The
 * Score strategy uses the reference's Score recurrence.  Defined by oracle/stevi_oracle.c so_sgm_textbook. */
int svh_sgm_cost_volume_textbook(svh_context *ctx, int n_directions, int strategy, const svh_array *cv, float P1, float P2,
                                 const int32_t margins[4], float Pout, svh_array *out);

/* ---- A10 extractSelectedIndex<strategy>(cv)                         correlation/correlation_base.h:427-464
 * cv (H,W,D) f32 -> idx (H,W) i32; ties go to the largest index, NaN never replaces the incumbent. */
int svh_extract_selected_index(svh_context *ctx, int strategy, const svh_array *cv, svh_array *idx);

/* ---- selectedIndexToDisp<disp_t,dDir>(idx, offset)                   correlation/correlation_base.h:511-532 */
int svh_selected_index_to_disp(svh_context *ctx, int disp_direction, const svh_array *idx, int32_t disp_offset,
                               svh_array *disp);

/* ---- selectedCost(cv, idx)                                           correlation/correlation_base.h:557-577 */
int svh_selected_cost(svh_context *ctx, const svh_array *cv, const svh_array *idx, svh_array *cost);

/* ---- A11 truncatedCostVolume<float,dir,sdir>(cv, idx, h_radius, v_radius, cost_vol_radius)
 *                                                                      correlation/correlation_base.h:579-674
 * -> tcv (H,W,2r+1) f32 (4r+1 for SVH_TCV_BOTH), NaN outside the valid domain. */
int svh_truncated_cost_volume(svh_context *ctx, int tcv_direction, int disp_direction, const svh_array *cv,
                              const svh_array *idx, int h_radius, int v_radius, int cost_vol_radius, svh_array *tcv);

/* ---- A12 refineDispCostInterpolation<kernel>(tcv, raw)            correlation/cost_based_refinement.h:128-163
 * tcv (H,W,2r+1) f32, raw (H,W) i32 -> refined (H,W) f32 = raw + refineCostTriplet(...) (:43-69); no clamp.
 * Depth not of the form 2r+1, r>=1 -> SVH_EMPTY_RESULT. */
int svh_refine_disp_cost_interpolation(svh_context *ctx, int interp_kernel, const svh_array *tcv, const svh_array *raw,
                                       svh_array *refined);

/* ---- A7 / A8 per-pixel statistics and feature-volume transforms as stand-alone functions -----------------------------------
 * (the cost-volume entry points evaluate them on the fly; examples/stereo_refine_test/main.cpp:386-398 calls them directly)
 * feat (H,W,F) f32; maps (H,W) f32; transformed volumes (H,W,F) f32.
 * channelsMean                                                                   correlation/correlation_base.h:1100-1136 */
int svh_channels_mean(svh_context *ctx, const svh_array *feat, svh_array *mean);
/* channelsNorm: sqrtf(sum v^2)                                                    correlation/cross_correlations.h:149-191 */
int svh_channels_norm(svh_context *ctx, const svh_array *feat, svh_array *norm);
/* channelsZeroMeanNorm: sqrtf(sum (v - mean)^2); mean == NULL -> channelsMean first (the one-argument overload)   :61-122 */
int svh_channels_zero_mean_norm(svh_context *ctx, const svh_array *feat, const svh_array *mean, svh_array *norm);
/* zeromeanFeatureVolume: v - mean                                                                                 :570-594 */
int svh_zeromean_feature_volume(svh_context *ctx, const svh_array *feat, const svh_array *mean, svh_array *out);
/* normalizedFeatureVolume: v / norm (true division, norm 0 -> NaN / inf)                                          :504-550 */
int svh_normalized_feature_volume(svh_context *ctx, const svh_array *feat, const svh_array *norm, svh_array *out);
/* zeromeanNormalizedFeatureVolume: (v - mean) / norm                                                              :416-462 */
int svh_zeromean_normalized_feature_volume(svh_context *ctx, const svh_array *feat, const svh_array *mean, const svh_array *norm, svh_array *out);
/* getFeatureVolumeForMatchFunc<matchFunc>(feature_vol): the volume aggregateCost / computeGuidedCV consume          :645-722
 * out (H,W,F) f32, or (H,W,nW) u32 census words for CENSUS / HAMMING (single-channel input -> SVH_EMPTY_RESULT). */
int svh_feature_volume_for_match_func(svh_context *ctx, int match_func, const svh_array *feat, svh_array *out);

/* ---- 2-D disparity (optical-flow style) volumes: SURVEY.md section 8(f) rank 2 --------------------------------------
 * unfoldBased2dDisparityCostVolume<matchFunc,...>(img_l, img_r, h_radius, v_radius, searchOffset<2>(lower0, upper0, lower1, upper1))
 *                                                                 correlation/cross_correlations.h:794-822 (+ :310-374)
 * cv (H,W,Dh,Dw) f32, CV(i,j,dh,dw) = cmp(src(i,j,:), tgt(i+dh+lower0, j+dw+lower1,:)); image sizes must agree. */
int svh_unfold_cost_volume_2d(svh_context *ctx, int match_func, int disp_direction, const svh_array *img_l, const svh_array *img_r,
                              int h_radius, int v_radius, int32_t lower0, int32_t upper0, int32_t lower1, int32_t upper1, svh_array *cv);
/* featureVolume2CostVolume<matchFunc, ..., searchOffset<2>, dDir>(feature_vol_l, feature_vol_r, searchRange)
 *                                                              correlation/cross_correlations.h:724-738 (+ :310-374)
 * raw feature volumes (H,W*,F) f32 -> cv (H,Ws,Dh,Dw) f32; only the row counts must agree (:324-326). */
int svh_feature_cost_volume_2d(svh_context *ctx, int match_func, int disp_direction, const svh_array *feat_l, const svh_array *feat_r,
                               int32_t lower0, int32_t upper0, int32_t lower1, int32_t upper1, svh_array *cv);
/* extractSelected2dIndex<strategy>(cv) -> idx (H,W,2) i32                          correlation/correlation_base.h:466-509 */
int svh_extract_selected_2d_index(svh_context *ctx, int strategy, const svh_array *cv, svh_array *idx);
/* selected2dIndexToDisp(idx, searchOffset<2>) -> disp (H,W,2) i32                  correlation/correlation_base.h:534-555 */
int svh_selected_2d_index_to_disp(svh_context *ctx, const svh_array *idx, int32_t lower0, int32_t lower1, svh_array *disp);
/* truncatedBidirectionaCostVolume(cv, idx, radius0, radius1) -> tcv (H,W,2r0+1,2r1+1), NaN outside the volume
 *                                                                                  correlation/correlation_base.h:677-725 */
int svh_truncated_bidirectional_cost_volume(svh_context *ctx, const svh_array *cv, const svh_array *idx, int radius0, int radius1,
                                            svh_array *tcv);

/* ---- 2-D cost-based refinement: SURVEY.md section 8(f) rank 1 (the refinement examples/stereo-match --refine calls,
 * examples/stereo-match/main.cpp:198-210) ------------------------------------------------------------------------------
 * refineDisp2dCostInterpolation<kernel, isotropHypothesis>(tcv, raw)          correlation/cost_based_refinement.h:165-376
 * tcv (H,W,2r0+1,2r1+1) f32, raw (H,W,2) i32 -> refined (H,W,2) f32 = raw + (delta0, delta1); both deltas are zeroed when
 * either is NaN or larger than 1 in magnitude (:362-366).  Isotropic: one refineCostTriplet per axis through the centre;
 * anisotropic: the two fitted extremum lines are intersected (:270-358).  Radii < 1 or even depths -> SVH_EMPTY_RESULT. */
typedef enum svh_isotropy { SVH_ISOTROPIC = 0, SVH_ANISOTROPIC = 1 } svh_isotropy; /* IsotropyHypothesis, :37-41 */
int svh_refine_disp_2d_cost_interpolation(svh_context *ctx, int interp_kernel, int isotropy, const svh_array *tcv, const svh_array *raw,
                                          svh_array *refined);
/* refineDisp2dCostPatchInterpolation<Parabola|Gaussian>(tcv, raw): least-squares quadric through the central 3x3 patch
 * (refineCostPatch, :71-126) and its stationary point                              correlation/cost_based_refinement.h:378-436 */
int svh_refine_disp_2d_cost_patch_interpolation(svh_context *ctx, int interp_kernel, const svh_array *tcv, const svh_array *raw,
                                                svh_array *refined);

/* ---- hierarchical (coarse-to-fine) matching: SURVEY.md section 8(f) rank 3 -------------------------------------------
 * Interpolation::averagePoolingDownsample(img, DownSampleWindows(h, v))                interpolation/downsampling.h:67-178
 * img (H,W[,C]) f32 -> out (ceil(H/v), ceil(W/h)[,C]) f32: mean over the valid samples of each window, as written there. */
int svh_average_pooling_downsample(svh_context *ctx, const svh_array *img, int win_horizontal, int win_vertical, svh_array *out);
/* computeGuidedCV<matchFunc, ..., dDir>(feature_vol_l, feature_vol_r, disp_guide, upscale_disp_radius)
 *                                                                                       correlation/hierarchical.h:74-229
 * feat_l / feat_r: the volumes getFeatureVolumeForMatchFunc returns, (H,W*,F) f32 (already zero-mean / normalised) or
 * (H,W*,nW) u32 census words; guide (Hg,Wg) i32, at least 2x2.  -> tcv (H,Ws,2r+1) f32 centred on the selected disparity,
 * disp (H,Ws) i32 = OffsetedCostVolume::{truncated_cost_volume, disp_estimate}.  Row mismatch -> SVH_EMPTY_RESULT. */
int svh_guided_cost_volume(svh_context *ctx, int match_func, int disp_direction, const svh_array *feat_l, const svh_array *feat_r,
                           const svh_array *guide, int32_t upscale_disp_radius, svh_array *tcv, svh_array *disp);
/* hiearchicalTruncatedCostVolume<matchFunc, depth, ..., dDir>(img_l, img_r, h_radiuses, v_radiuses, disp_width, upscale_disp_radius)
 *                                                                                       correlation/hierarchical.h:232-319
 * h_radii / v_radii: depth + 1 entries, coarsest level first (the single-radius overload repeats one value, :296-315).
 * depth 2x2 average poolings, full search over ceil(disp_width / 2^depth) disparities at the coarsest level
 * (unfoldBasedCostVolume + extractSelectedIndex), one computeGuidedCV per level on the way up; nothing leaves the device
 * in between.  Outputs as for svh_guided_cost_volume at the resolution of the source image.
 * The coarsest volume uses the register-blocked kernels (results within rounding of the reference's; an exact tie broken
 * differently changes that pixel's estimate) unless svh_context_set_option(ctx, "literal_cost_volumes", 1) selects the
 * per-voxel kernel, which follows the reference operation by operation. */
int svh_hierarchical_truncated_cost_volume(svh_context *ctx, int match_func, int disp_direction, int depth, const svh_array *img_l,
                                           const svh_array *img_r, const int32_t *h_radii, const int32_t *v_radii, int32_t disp_width,
                                           int32_t upscale_disp_radius, svh_array *tcv, svh_array *disp);

/* ---- on-demand (cacheless) cost volumes and PatchMatch: SURVEY.md section 8(f) rank 1 -- what examples/stereo-match runs
 * (main.cpp:166-210).  Float matching functions only (CC ... ZSAD).  Images (H,W[,C]) f32.
 * Features: OnDemandDecoratedFeaturesVolume<ZNFeaturesVolumeDecorator<ZeroMean, Normalized>, ...> over the full window
 * (di, dj) in [-v_radius, v_radius] x [-h_radius, h_radius], channels innermost (main.cpp:150-164): samples CLAMPED to the image
 * border, mean = sum / nF, norm = sqrt(sum of squares / nF)              correlation/on_demand_features_volume.h:34-214 */
typedef struct svh_on_demand_params {
    int32_t match_func;   /* svh_match_func, not CENSUS / HAMMING */
    int32_t search_dims;  /* 1: CachelessOnDemandStereoCostVolume (columns only), 2: CachelessOnDemandImageFlowVolume (rows, columns) */
    int32_t h_radius, v_radius;
    int32_t lower0, upper0; /* searched row offsets (search_dims == 2 only) */
    int32_t lower1, upper1; /* searched column offsets */
} svh_on_demand_params;
/* getFeatureVec of every pixel -> out (H,W,nF) f32, nF = (2 v_radius + 1)(2 h_radius + 1) C */
int svh_on_demand_features(svh_context *ctx, int match_func, const svh_array *img, int h_radius, int v_radius, svh_array *out);
/* CachelessOnDemandCostVolume::truncatedCostVolume(disp, radius)                 correlation/on_demand_cost_volume.h:474-596
 * disp (H,W,search_dims) i32 -> tcv (H,W,2r+1[,2r+1]) f32, AS WRITTEN: the window is centred on disparity - lowerOffset (an index is
 * handed to costValue where it expects a disparity, :513-514), entries without a value hold defaultCvValForMatchFunc (FLT_MAX for
 * costs, FLT_MIN for scores, matching_costs.h:706-713).  disp with another component count -> SVH_EMPTY_RESULT (:478-480). */
int svh_on_demand_truncated_cost_volume(svh_context *ctx, const svh_on_demand_params *params, const svh_array *img_source,
                                        const svh_array *img_target, const svh_array *disp, int radius, svh_array *tcv);
/* cachelessPatchMatch<matchFunc, searchSpaceDim>(features_source, features_target, searchOffset, nIter, nRandomSearch)
 *                                                                                 correlation/patchmatch.h:560-621
 * -> disp (H,W,search_dims) i32; *iterations_run (optional) = iterations executed before "no change" stopped the loop.
 * The reference draws from std::default_random_engine seeded by std::random_device per thread (not reproducible); here every
 * draw is a pure function of (seed, iteration, pixel, draw, dimension) mapped into the range like the reference's NumbersCache
 * branch (|v % range| + lower), so equal seeds give equal results, bit-identical to oracle/stevi_oracle.c. */
int svh_cacheless_patch_match(svh_context *ctx, const svh_on_demand_params *params, const svh_array *img_source, const svh_array *img_target,
                              int n_iter, int n_random_search, uint64_t seed, svh_array *disp, int32_t *iterations_run);
/* The same with the caller's initial solution in place of the random draw (the reference's `initializer` callback, patchmatch.h:598-605,
 * evaluated by the caller): initial_disp (H,W,search_dims) i32, NULL = draw. */
int svh_cacheless_patch_match_init(svh_context *ctx, const svh_on_demand_params *params, const svh_array *img_source, const svh_array *img_target,
                                   int n_iter, int n_random_search, uint64_t seed, const svh_array *initial_disp, svh_array *disp,
                                   int32_t *iterations_run);
/* patchMatch<matchFunc, searchSpaceDim>(feature_vol_s, feature_vol_t, searchOffset, nIter, nRandomSearch, initializer, randcache)
 *                                                                                 correlation/patchmatch.h:496-558
 * The iteration of svh_cacheless_patch_match on FEATURE VOLUMES the caller built -- (H,W,F) f32, e.g. unfolded images
 * (benchmarkStereoMatchingModels.cpp:187-199) -- through the reference's cached cost volume, whose values are featureComparison of the
 * processed vectors (the cache changes no value).  Zero-mean / normalised functions process the vectors twice, as the reference does
 * (:522-523 and on_demand_cost_volume.h:62-67).  params: match_func, search_dims and the offsets; the radii are not used.  Feature counts
 * that differ, or row counts with search_dims 1 -> SVH_EMPTY_RESULT (:529-537).  initial_disp: NULL = the random draw.  Random stream
 * as svh_cacheless_patch_match (the reference's `randcache` has no counterpart: its own stream is not reproducible). */
int svh_patch_match(svh_context *ctx, const svh_on_demand_params *params, const svh_array *feat_source, const svh_array *feat_target, int n_iter,
                    int n_random_search, uint64_t seed, const svh_array *initial_disp, svh_array *disp, int32_t *iterations_run);

/* ---- fused pipeline: the benchmark / stereo_refine_test call chain kept on the device ------------------
 * unfoldBasedCostVolume -> [sgmCostVolume] -> extractSelectedIndex -> selectedIndexToDisp
 *   (test/benchmarks/benchmarkCrossCorrelationAlgorithms.cpp:92-96, :288-294)
 * -> [truncatedCostVolume -> refineDispCostInterpolation] (examples/stereo_refine_test/main.cpp:367-384).
 * Produces bit-identical results to calling the entry points above one by one; volumes that are not
 * requested as outputs are never written to HBM when the cost function allows it (census/Hamming). */
typedef struct svh_stereo_params {
    int32_t match_func;      /* svh_match_func */
    int32_t disp_direction;  /* svh_disp_direction */
    int32_t h_radius, v_radius;
    int32_t disp_lower;      /* first searched offset (0 for the disp_t overload) */
    int32_t disp_count;      /* D */
    int32_t sgm_directions;  /* 0 = no SGM, 4 or 8 */
    float P1, P2, Pout;
    int32_t margins[4];      /* left, top, right, bottom */
    int32_t refine_kernel;   /* -1 = no refinement, else svh_interp_kernel */
    int32_t refine_h_radius, refine_v_radius; /* radii passed to truncatedCostVolume */
    /* disparity shard for multi-GPU runs: this call handles indices [shard_begin, shard_begin+shard_count)
     * of the D-wide range; shard_count = 0 means the whole range. */
    int32_t shard_begin, shard_count;
} svh_stereo_params;

/* Optional outputs may be NULL.  disp (H,W) i32; refined (H,W) f32; cv / sgm_cv (H,W,D) f32;
 * keys (H,W) u64 = order-preserving (value, index) keys of the local winner for a cross-GPU min-reduction
 * (see svh_keys_to_index). */
int svh_stereo_match(svh_context *ctx, const svh_stereo_params *params, const svh_array *img_l, const svh_array *img_r,
                     svh_array *disp, svh_array *refined, svh_array *cv, svh_array *sgm_cv, svh_array *keys);

/* ---- disparity-sharded census (+ SGM) across GPUs -------------------------------------------------------------
 * Rank r handles the disparity indices [shard_begin, shard_begin + shard_count) of params->disp_count:
 *   svh_census_shard_keys   -> keys (H,W,2) i32: the shard's regional winner keys (cost, global index), all positive
 *   int32 MIN all-reduce of `keys` over the ranks (RCCL / torch.distributed): the only exchange.  When
 *   svh_census_shard_region1_is_global() returns 1 the second key of every pixel (keys[..., 1], the region that pays Pout) is
 *   already the winner over ALL shards as svh_census_shard_keys writes it -- every disparity of that region looks outside the
 *   target image (zero vector: equal costs, the last index of the whole range wins) -- and only keys[..., 0] needs the reduction
 *   (half the bytes; reducing both is still correct)
 *   svh_census_shard_finish -> SGM line recurrences from the reduced keys + winner: disp (H,W) i32 [, refined (H,W) f32]
 * The result is bit-identical to the single-GPU svh_stereo_match.  Census / Hamming costs in the integer-exact regime
 * only (integer Pout, window <= 15x15 = at most 8 census words, <= 1024 disparities per shard, <= 4096 in total); otherwise SVH_ERR_UNSUPPORTED:
 * the Score branch and non-integer costs couple the disparities along every path step and do not shard. */
int svh_census_shard_keys(svh_context *ctx, const svh_stereo_params *params, const svh_array *img_l, const svh_array *img_r,
                          svh_array *keys);
/* 1: RightToLeft and source width + disp_lower >= target width (only shapes are read, no device work); else 0 */
int svh_census_shard_region1_is_global(const svh_stereo_params *params, const svh_array *img_l, const svh_array *img_r);
int svh_census_shard_finish(svh_context *ctx, const svh_stereo_params *params, const svh_array *img_l, const svh_array *img_r,
                            const svh_array *keys, svh_array *disp, svh_array *refined);
/* The exchange itself for hosts that are not Python (the C++ drop-in: correlation/sharded.h): an int32 MIN all-reduce of `keys`
 * (dense, device memory, shape (H,W,2)) in place over the caller's RCCL communicator `nccl_comm` (an ncclComm_t), enqueued on the
 * context's stream like every kernel of the library; with plane0_only != 0 only keys[..., 0] travels (see
 * svh_census_shard_region1_is_global).  One process per GPU.  The library does not link librccl: ncclAllReduce is taken from the RCCL
 * instance already loaded in the process -- the one the communicator belongs to -- or from librccl.so.1; SVH_ERR_UNSUPPORTED when
 * there is none.  Replaces nothing in the reference (single-node CPU code); it is the north star's "RCCL all-reduce for the final
 * per-pixel argmin". */
int svh_census_exchange_keys(svh_context *ctx, void *nccl_comm, svh_array *keys, int plane0_only);

/* ---- row bands: the same disparity map, rows [row_begin, row_begin + row_count) of it ------------------------------
 * In the integer-exact regime the winning disparity of a pixel depends on the pixel's own costs and on its position in the image
 * only (the per-pass minima of sgm.h:257-296 shift all its disparities alike: see "census_winner_shortcut"), so the rows of the
 * disparity map are independent and GPUs can split them with no exchange at all: rank r calls this with its band and holds rows
 * of the map that are bit-identical to the single call's.  img_l / img_r are the WHOLE images (the band's census windows read
 * v_radius rows beyond it); disp_band (row_count, W) i32.  Census costs in the exact regime, the whole disparity range
 * (a multiple of 32 up to 992; RightToLeft ranges that end at the target image's edge: any count from 33 to 512), no refinement; otherwise SVH_ERR_UNSUPPORTED (use svh_stereo_match / the disparity shards). */
int svh_census_band_match(svh_context *ctx, const svh_stereo_params *params, const svh_array *img_l, const svh_array *img_r,
                          int32_t row_begin, int32_t row_count, svh_array *disp_band);

/* decodes reduced keys back to selected indices / disparities (device or host arrays) */
int svh_keys_to_index(svh_context *ctx, int strategy, const svh_array *keys, int32_t disp_count, svh_array *idx);

#ifdef __cplusplus
}
#endif

#endif /* STEVI_HIP_H */

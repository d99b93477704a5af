// Internal declarations shared by the translation units of libstevi_hip.so.
// Nothing here is part of the ABI (include/stevi_hip.h is).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/stevi_hip.h"

namespace svh {

// ---- device workspace: a small caching pool so steady-state calls never reach hipMalloc ----------
struct PoolBlock {
    void *ptr;
    size_t bytes;
    bool in_use;
};

struct ProfPending {
    std::string name;
    hipEvent_t start, stop;
};

struct ProfStat {
    double total_ms = 0;
    int64_t launches = 0;
};

struct Staging; // page-locked ring, copy streams and events of a context's host <-> device transfers (svh_transfer.hip)

} // namespace svh

struct svh_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string last_error;
    std::vector<svh::PoolBlock> pool;
    svh::Staging *staging = nullptr; // created by the first large transfer from or to pageable host memory
    bool profiling = false;
    std::string prof_filter; // when not empty only launches of this kernel are bracketed by events
    int prof_every = 1;      // svh_profile_sampling: bracket every n-th eligible launch
    int64_t prof_seen = 0;
    bool census_fast_path = true; // svh_test_set_option("census_fast_path")
    bool census_winner_shortcut = true; // svh_context_set_option("census_winner_shortcut"): index / disparity maps without the line scans
    int census_sweep_mode = 0;         // svh_context_set_option("census_sweep"): 0 auto, 1 vector-ALU kernel, 3 FP4 matrix-core kernels
    int census_float_overflow = 0;     // svh_context_set_option("census_float_overflow"): rule E2 when a target word rounds to 2^32: 0 saturate (0xFFFFFFFF), 1 zero
    int census_sweep_rl = 1;       // svh_test_set_option("census_sweep_rl"): the FP4 engine may use its RightToLeft specialisation (svh_census_sweep_rl.hip)
    bool census_tiles = true;          // svh_test_set_option("census_tiles"): census + SGM with the recurrences run keeps only the carries of the line scans and replays them per tile in the per-pixel kernel (0: six min_p maps, round 2's pair of kernels)
    bool cost_volume_colsum = true;    // svh_test_set_option("cost_volume_colsum"): float cost volumes of grey images share column sums between windows (0: every window on its own, round 1's kernel)
    bool fold_2d_offsets = true;       // svh_test_set_option("fold_2d_offsets"): 2-D disparity volumes of grey images take as many vertical offsets per launch of the column-sum kernel as its tile holds (0: a launch per vertical offset)
    bool sgm_score_pad = true;         // svh_test_set_option("sgm_score_pad"): Score-branch SGM on 65 .. 511 disparities that are no multiple of 64 runs on copies padded to the next multiple (pads -inf: inert), so that the vector kernels apply (0: the masked forms)
    int guided_shared = 1;             // svh_test_set_option("guided_shared"): computeGuidedCV on grey images shares the processed target features (each sample subtracted / divided once) through LDS: 1 per wave of 64 pixels on large grids and per block of 256 on small ones, 2 per block always (round 4), 3 per wave always, 0 every pixel and offset on its own
    bool feature_volume_tiled = true;  // svh_test_set_option("feature_volume_tiled"): cost volumes of float feature volumes process the features once and compare from LDS (0: the per-voxel kernel processes both vectors of every voxel)
    bool extract_index_wide = true;    // svh_test_set_option("extract_index_wide"): extractSelectedIndex on rows of up to 1 024 costs that the packed kernel does not take (more than 256 costs, or no multiple of four) combines by two all-reduces per pixel (0: six rounds of value / index exchanges)
    bool feature_volume_records = true; // svh_test_set_option("feature_volume_records"): feature vectors of up to 32 floats are compared with the target record in registers (a lane per record walks the source pixels that look at it); 0: every target feature of every voxel read from LDS (round 4)
    bool patchmatch_pred_costs = true; // svh_test_set_option("patchmatch_pred_costs"): PatchMatch sweeps take the cost of a pixel against its predecessor's unchanged solution from a parallel pre-pass (0: every step evaluates its cost)
    int patchmatch_search_form = 1; // svh_test_set_option("patchmatch_search_form"): PatchMatch's random search: 1 the chunked kernel (64 candidates per wave, 32 features at a time through a 9 KB LDS table), 0 round 4's batched kernel, 2 / 3 a lane per candidate without LDS (fetching the target features / forming them again from the target image)
    bool patchmatch_run_batches = true; // svh_test_set_option("patchmatch_run_batches"): a PatchMatch sweep step that evaluates a cost on the spot evaluates the next eight pixels of its line against the same travelling candidate with it (0: one evaluation per step)
    bool patchmatch_lookback = true; // svh_test_set_option("patchmatch_lookback"): after the first iteration PatchMatch's pre-pass also evaluates every pixel against the pre-sweep solutions two to four steps back, so that a travelling candidate needs no evaluation on the spot before its fourth accepted step (0: one step back only)
    bool patchmatch_scan_chunks = true; // svh_test_set_option("patchmatch_scan_chunks"): from the second iteration on a sweep line decides 64 steps at a time by a prefix scan over per-pixel transition tables (0: step by step)
    bool literal_cost_volumes = false; // svh_context_set_option("literal_cost_volumes"): hierarchical matching uses the per-voxel kernel
    bool cost_reduce_fused = true;     // svh_test_set_option("cost_reduce_fused"): svh_stereo_match lets the float cost kernel reduce over the disparity axis while it holds the costs -- the winner of a call without SGM (no volume written), the regional minima of a Cost-branch SGM (no probing read) -- 0: separate kernels read the volume back
    bool sgm_cost_two_minima = true;   // svh_test_set_option("sgm_cost_two_minima"): the Cost branch on a float volume runs its line recurrences on the two regional minima of every pixel (one read of the volume) instead of sweeping the volume once per pass
    bool sgm_score_finish_fused = true; // svh_test_set_option("sgm_score_finish_fused"): svh_stereo_match lets the Score branch's last writer of each pixel emit its winner / taps (0: extract_index + truncatedCostVolume read S back)
    int sgm_score_fused = 1;           // svh_context_set_option("sgm_score_fused"): the four downward Score-branch passes in one sweep (2 bands, 3 bands with 16-column strips forced; 0: a launch per pass; 1: the faster of the two by a model)
    std::vector<svh::ProfPending> prof_pending;
    std::vector<hipEvent_t> prof_free_events;
    std::map<std::string, svh::ProfStat> prof_stats;
    std::vector<std::string> prof_order;
};

namespace svh {

int fail(svh_context *ctx, int status, const char *fmt, ...);

#define SVH_HIP_CHECK(ctx, expr)                                                                     \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return svh::fail(ctx, _e == hipErrorOutOfMemory ? SVH_ERR_OUT_OF_MEMORY : SVH_ERR_HIP,   \
                             "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define SVH_TRY(expr)                 \
    do {                              \
        int _s = (expr);              \
        if (_s != SVH_OK) return _s;  \
    } while (0)

// Makes the context's device current for the duration of a call and puts the caller's current device back afterwards (a
// single-process multi-GPU caller keeps its own; a NULL stream means "the null stream of the CURRENT device" to the runtime)
class DeviceGuard {
  public:
    explicit DeviceGuard(int device) : target(device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != target) (void)hipSetDevice(target);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != target) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;

  private:
    int target, prev = -1;
};

// RAII over pool blocks taken during one API call
class Scratch {
  public:
    // every API call builds one Scratch first: also the place where the context's device is made current
    explicit Scratch(svh_context *c) : ctx(c), guard(c->device) {}
    ~Scratch();
    // returns nullptr (and sets the context error) on failure
    void *get(size_t bytes);
    template <class T> T *get_n(size_t n) { return static_cast<T *>(get(n * sizeof(T))); }

  private:
    svh_context *ctx;
    DeviceGuard guard;
    std::vector<size_t> taken;
};

// bracket a kernel launch with events when profiling is on
class ProfScope {
  public:
    ProfScope(svh_context *c, const char *name);
    ~ProfScope();

  private:
    svh_context *ctx;
    bool active = false;
    hipEvent_t start = nullptr, stop = nullptr;
    const char *name;
};

#define SVH_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                              \
    do {                                                                                    \
        svh::ProfScope _prof(ctx, name);                                                    \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);         \
    } while (0)

#define SVH_CHECK_LAUNCH(ctx) SVH_HIP_CHECK(ctx, hipGetLastError())

// ---- array marshalling ---------------------------------------------------------------------------
size_t dtype_size(int dtype);
int64_t num_elements(const svh_array &a);
bool is_dense(const svh_array &a); // last index fastest, no gaps (size-1 dims ignored)
int validate(svh_context *ctx, const svh_array *a, const char *what, int dtype, int ndim_min, int ndim_max);

// Device-resident dense view of an input array: the array itself when it already is device+dense,
// otherwise a scratch copy (H2D and/or relayout).
int stage_in(svh_context *ctx, Scratch &scr, const svh_array &a, void **dptr);

// Images: float32, or uint8 for the functions whose features are the plain samples (CENSUS, HAMMING, CC, SSD, SAD; unfold).
// The reference keeps uint8 features there and forms every difference / product / comparison after a cast to float
// (matching_costs.h:59-156, :749-757; census.h:89-101), so widening the samples once on the device gives the same bits.
// Normalised and zero-mean functions take an int16 integer path on uint8 images that the reference's own comparison test
// skips (testCorrelationFilters.cpp:1249): SVH_ERR_UNSUPPORTED.  match_func < 0: no function involved (unfold, census).
int validate_image(svh_context *ctx, const svh_array *img, const char *what, int match_func);
int stage_image(svh_context *ctx, Scratch &scr, const svh_array &img, void **dptr); // dense float32 device view
// a cost volume of any element type sgmCostVolume's T_CV may take here (f32, u8, i16, u16, i32, u32) as a dense float32 device view
int validate_volume(svh_context *ctx, const svh_array *cv, const char *what);
int stage_volume_as_float(svh_context *ctx, Scratch &scr, const svh_array &cv, void **dptr);

// Device-resident dense buffer to compute an output into; finish() moves it to the user's array when needed.
struct OutStage {
    void *dptr = nullptr;
    bool direct = false;
    const svh_array *dst = nullptr;
};
// host <-> device copies (svh_transfer.hip): direct DMA for page-locked host memory, chunks through a page-locked ring on two copy
// streams for large pageable arrays.  copy_h2d returns when `src` may be reused, with the data ordered before everything enqueued on the
// context's stream afterwards; copy_d2h returns when `dst` holds what the context's stream had produced at the call.
int copy_h2d(svh_context *ctx, void *dst, const void *src, size_t bytes);
int copy_d2h(svh_context *ctx, void *dst, const void *src, size_t bytes);
void staging_destroy(svh_context *ctx);
// (svh_context.hip) give the device's cache of released svh_device_alloc blocks back to the device: the out-of-memory retry of every allocator
void device_cache_release_all(int device);
// (svh_feature_transforms.hip) getFeatureVolumeForMatchFunc of a float matching function on dense device arrays (H, W, F)
int dev_feature_volume_for_match_func(svh_context *ctx, Scratch &scr, int match_func, const float *feat, int H, int W, int F, float *out);
int stage_out(svh_context *ctx, Scratch &scr, const svh_array &a, OutStage *st);
int finish_out(svh_context *ctx, const OutStage &st);
// true when any host array took part (the call must synchronise before returning)
bool any_host(std::initializer_list<const svh_array *> arrays);

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
inline int grid_for(int64_t n, int block, int max_blocks = 1 << 20) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

// ---- trait helpers (correlation/matching_costs.h:419-685) -----------------------------------------
inline bool func_supported(int f) { return (f >= SVH_CC && f <= SVH_ZSAD) || f == SVH_HAMMING || f == SVH_CENSUS; }
inline bool func_zero_mean(int f) { return f == SVH_ZCC || f == SVH_ZNCC || f == SVH_ZSSD || f == SVH_ZSAD; }
inline bool func_normalized(int f) { return f == SVH_NCC || f == SVH_ZNCC; }
inline bool func_census(int f) { return f == SVH_HAMMING || f == SVH_CENSUS; }
inline int func_strategy(int f) { return (f == SVH_CC || f == SVH_NCC || f == SVH_ZCC || f == SVH_ZNCC) ? SVH_SCORE : SVH_COST; }
inline int census_words(int F) { return (F - 1) / 32 + 1; }
inline int census_words_written(int F) { return (F - 1) / 32; }

// ---- device-level building blocks (dense device pointers; shapes already validated) ---------------
struct ImageDesc { // dense (H, W, C) float image on the device
    const float *data;
    int H, W, C;
};

int dev_unfold(svh_context *ctx, ImageDesc img, int h_r, int v_r, int pl, int pt, int Ho, int Wo, float *out);
// words layout (H, W, n_out) with n_out = nW (API layout, trailing word zero) or nWw (compact);
// round_through_float applies rule E2 (target side of aggregateCost) to every word.
int dev_census_from_image(svh_context *ctx, ImageDesc img, int h_r, int v_r, int pl, int pt, int Ho, int Wo, int n_out,
                          bool round_through_float, uint32_t *words);
int dev_census_pair_compact(svh_context *ctx, ImageDesc src, ImageDesc tgt, int h_r, int v_r, int nWw, uint32_t *sw, uint32_t *tw);
int dev_census_from_features(svh_context *ctx, const float *feat, int H, int W, int F, int n_out, bool round_through_float,
                             uint32_t *words);

// cost volume (H, Ws, D) from feature volumes or images
// per-pixel window statistics of an image pair, kept across the vertical offsets of a 2-D volume
struct WindowStatsCache {
    Scratch *scr = nullptr; // owner of the maps (must outlive the passes)
    bool ready = false;
    float *ms = nullptr, *mt = nullptr, *ns = nullptr, *nt = nullptr, *zc = nullptr;
};
// Asked of dev_cost_volume_grey_tiled by a caller that will aggregate the volume with the Score branch: are all its costs finite and of
// magnitude about one?  True for a normalised function (NCC, ZNCC) exactly when every window norm of both images is a positive finite
// number; the statistics kernels check that as they write the norms and the answer (one word) is read back BEFORE the cost kernel is
// launched, so the wait is for the statistics only.  asked: the caller wants it; known: an answer was produced; all_finite: the answer.
struct FiniteCostsQuery {
    bool asked = false, known = false, all_finite = false;
};
// Per-pixel reductions over the disparity axis that the float cost kernel computes while it holds the costs (the column-sum kernel of
// svh_cost_volume_tiled.hip: a block's waves hold a pixel's whole range between them), so that nobody reads the volume back for them:
//   mode 1  the winner (extractSelectedIndex, correlation_base.h:427-464: extremum, ties to the larger index, NaN never wins unless at
//           index 0) -> idx and / or disp = disp_sign * idx + disp_offset; with store == false the volume is not written at all
//   mode 2  the two regional minima a later Cost-branch sgmCostVolume needs (svh_sgm.hip: the smallest finite cost among the disparities
//           with j + d < W, and among those with j + d >= W) -> minima (H, W) float2; bit 1 of *flag is raised when a finite |c| > big
// done: set by the launcher when the kernel that ran produced them (other kernels do not: the caller falls back to reading the volume).
struct CostReduce {
    int mode = 0;
    bool score = false;
    int32_t *idx = nullptr, *disp = nullptr;
    int disp_sign = 1, disp_offset = 0;
    float *minima = nullptr;
    int *flag = nullptr;
    float big = 1e30f;
    bool store = true;
    bool done = false;
};
struct CostVolumeArgs {
    int func, ddir;
    int H, Ws, Wt;
    int disp_lower, D;
    // 2-D disparity volumes (aggregateCost(searchOffset<2>), cross_correlations.h:310-374) are built one vertical offset
    // at a time: the target is read at row i + tgt_row_off, the column sign is +1 whatever dDir says, and the D costs of
    // a pixel go to cv[pixel * out_px_stride + out_off + d].  Zero / default values give the 1-D behaviour.
    int tgt_row_off = 0;
    int force_sign = 0;          // 0: +1 for RightToLeft, -1 for LeftToRight
    int64_t out_px_stride = 0;   // 0: D
    int64_t out_off = 0;
    // n_dh > 1: one launch takes the vertical offsets tgt_row_off .. tgt_row_off + n_dh - 1 (offset dh at out_off + dh D); only where
    // cost_volume_colsum_applies says the column-sum kernel runs
    int n_dh = 1;
    int C = 1;                   // interleaved channels of the images (set by dev_cost_volume_grey_tiled)
    bool literal = false;        // skip the register-blocked kernel: the reference's operations in the reference's order
    WindowStatsCache *stats = nullptr; // optional: statistics maps shared by several passes over the same image pair
    // census / Hamming volumes: per pixel the smallest cost among the disparities that do not / do pay Pout in a later sgmCostVolume
    // (device, (H, Ws) float2; svh_unfold_cost_volume_minima); *minima_written says whether the kernel that ran produced them
    float *minima = nullptr;
    int *minima_written = nullptr;
    CostReduce *reduce = nullptr; // float costs of grey images: see CostReduce (ignored by the kernels that cannot do it: check reduce->done)
    FiniteCostsQuery *finite_query = nullptr;
    // image rows [row_begin, row_begin + row_count) only (row_count 0: all of them; dev_cost_volume_grey_tiled)
    int row_begin = 0, row_count = 0;
    int sign() const { return force_sign ? force_sign : (ddir == SVH_RIGHT_TO_LEFT ? 1 : -1); }
    int64_t px_stride() const { return out_px_stride ? out_px_stride : D; }
};
int dev_cost_volume_from_features(svh_context *ctx, Scratch &scr, const CostVolumeArgs &a, const float *feat_src,
                                  const float *feat_tgt, int F, float *cv);
int dev_cost_volume_from_images(svh_context *ctx, Scratch &scr, const CostVolumeArgs &a, ImageDesc src, ImageDesc tgt, int h_r,
                                int v_r, float *cv);
int dev_cost_volume_grey_tiled(svh_context *ctx, Scratch &scr, const CostVolumeArgs &a, ImageDesc src, ImageDesc tgt, int h_r, int v_r, float *cv);
// whether dev_cost_volume_grey_tiled would run the column-sum kernel for these arguments (the one that can serve a CostReduce)
// (whether it can also serve a CostReduce: grey images only -- callers that hand it one check src.C == 1 themselves)
bool cost_volume_colsum_applies(const svh_context *ctx, const CostVolumeArgs &a, ImageDesc src, ImageDesc tgt, int h_r, int v_r);
// Hamming volume from compact census words (src exact, tgt already rounded through float)
int dev_hamming_volume(svh_context *ctx, const CostVolumeArgs &a, const uint32_t *src_words, const uint32_t *tgt_words, int nWw,
                       float *cv);

// SGM
struct SgmArgs {
    int n_dir, strategy;
    int H, W, D;
    float P1, P2, Pout;
    int left, top, right, bottom;
    // row bands (svh_census_band_match): the arrays hold rows [row_origin, row_origin + H) of an image of full_H rows (0: the
    // arrays are the whole image); margins and line geometry refer to the full image; only the local rows
    // [store_row0, store_row0 + store_rows) are written, to an output of store_rows rows
    int row_origin = 0, full_H = 0, store_row0 = 0, store_rows = 0;
    // Score branch: every cost is a finite number of magnitude about one (FiniteCostsQuery) and Pout is finite -- then every line state
    // stays finite and the isfinite filters of sgm.h:224, :241, :251 are no-ops: the kernels that are bound by instruction issue drop them
    bool costs_all_finite = false;
    // Score branch: the cost volume's rows already have this pitch (the next multiple of 64 above D) with -inf behind the D costs
    // (svh_stereo_match lets the cost kernel write that layout: dev_sgm_score_branch then skips its pad-in copy)
    int cv_pitch = 0;
};
// The min_p maps of the six effective passes live in FIVE planes: passes 2 and 3 (the two start loops of UpLeft2DownRight, sgm.h:331-345)
// partition the margin box along its diagonal -- pass 2 visits ip >= jp, pass 3 jp >= ip -- and the pixels both visit, the diagonal itself,
// lie on the one line both loops run (the line from the corner runs twice: finding F5), where both compute the same values from the
// same pixels in the same order.  So they share a plane; a diagonal pixel counts it twice.
constexpr int MIN_P_PLANES = 5;
__host__ __device__ __forceinline__ int min_p_plane(int q) { return q <= 2 ? q : q - 1; }

// cost source for the Cost-branch kernels: either a dense float volume or census words evaluated on the fly
struct CostSource {
    const float *cv = nullptr;       // (H, W, D) dense, or nullptr
    // float volume the library has just written (CostReduce mode 2): its regional minima (H, W) float2 and the flag word the kernel that
    // wrote them raised bit 1 of when a finite |c| exceeded 1e30 (bit 0 is set by the host: no exact-integer route is tried); the Cost
    // branch then needs no probing read of the volume
    const float *float_minima = nullptr;
    int *float_flag = nullptr;
    const uint32_t *src_words = nullptr, *tgt_words = nullptr; // compact (H, W*, nWw)
    int nWw = 0, Wt = 0, sign = 1, disp_lower = 0;
    int d_offset = 0; // disparity shards: global index of local disparity 0 (disp_lower already includes it)
    // disparity shards: >= 0 when every disparity that pays Pout looks at a target column outside the image (zero vector), so that
    // the Pout region's winner over ALL shards is known locally -- cost |s|, this index (the last of the whole range) -- and the
    // sweep writes that instead of its own shard's: the second key plane then needs no exchange.  -1: plain regional keys.
    int region1_global_last = -1;
    // dense volume + the caller's statement about it (svh_sgm_cost_volume_minima): every cost an integer of magnitude <= max_abs, and per pixel
    // the minima over the two regions of sgm.h:287-289 -- what the probe pass over the volume would otherwise establish
    const float *minima = nullptr;
    float max_abs = 0.0f;
};
// per-pixel outputs of the winner stage; every pointer is optional
struct WinnerOut {
    int32_t *idx = nullptr;            // extractSelectedIndex
    int32_t *disp = nullptr;           // selectedIndexToDisp: disp_sign * idx + disp_offset
    int disp_sign = 1, disp_offset = 0;
    float *taps = nullptr;             // (H, W, 3): truncatedCostVolume<Same>(S, idx, taps_h_r, taps_v_r, 1)
    int taps_h_r = 0, taps_v_r = 0;
    // the consumer of the taps only uses their differences (parabola and equiangular refinement, cost_based_refinement.h:43-69): in
    // the exact regime the three taps of a pixel may then carry a common integer offset -- the sum of the min_p maps can be left out
    bool taps_up_to_shift = false;
    unsigned long long *keys = nullptr; // cross-shard reduction keys
    int key_offset = 0, key_total = 0;
    bool any() const { return idx || disp || taps || keys; }
};
// Cost branch on either source; out_sgm (H, W, D) optional
int dev_sgm_cost_branch(svh_context *ctx, Scratch &scr, const SgmArgs &a, const CostSource &src, float *out_sgm, const WinnerOut &win);
// Score branch.  finish (optional): every pixel's winner record -- (the three truncatedCostVolume<Same> taps around the winner, the
// winner's index as bits): wave_emit_record in svh_sgm_lines.h -- written by whichever launch writes the pixel's FINAL aggregated costs:
// DownLeft2UpRight for the pixels it visits (i + j < H), the downward sweep for the rest (sgm.h:379-389 fixes the order of the passes; the
// sweep writes a record for every pixel and the later pass overwrites its own), so that nothing reads S back.  finish->done says whether
// that happened (whole image, 8 directions, P2 >= P1 >= 0, vector form: otherwise the caller runs the winner kernels as before).
// finish->store_all == false: out_sgm is scratch, only the costs a later pass reads are stored (the pixels DownLeft2UpRight visits).
struct ScoreFinish {
    float *records = nullptr; // (H, W, 4) floats
    int taps_h_r = 0, taps_v_r = 0;
    bool store_all = true;
    bool done = false;
    int d_valid = 0; // > 0: the rows are padded to a whole number of lanes and only the first d_valid costs exist (dev_sgm_score_branch)
};
// records -> index / disparity / refined maps (any of them may be null; refine_kernel < 0: no refinement)
int dev_finish_records(svh_context *ctx, const float *records, int64_t npx, int refine_kernel, int disp_sign, int disp_offset, int32_t *idx,
                       int32_t *disp, float *refined);
int dev_sgm_score_branch(svh_context *ctx, Scratch &scr, const SgmArgs &a, const float *cv, float *out_sgm, bool textbook = false, ScoreFinish *finish = nullptr);
// -inf into the pads [D, pitch) of every row of a volume whose rows were written at that pitch
int dev_sgm_fill_pads(svh_context *ctx, float *cv, int64_t n_rows, int D, int pitch);
// Score branch, whole image, 8 directions, P2 >= P1 >= 0: the four downward passes as one sweep (svh_sgm_sweep.hip; form = the
// "sgm_score_fused" option); *ran = false when the geometry is outside what the sweep covers.  The line kernel for one pass.
int dev_sgm_score_sweep(svh_context *ctx, Scratch &scr, const SgmArgs &a, const float *cv, float *sgm, bool vec, int form, bool *ran, ScoreFinish *finish = nullptr);
int dev_sgm_score_line_pass(svh_context *ctx, const SgmArgs &a, const float *cv, float *sgm, int pass, bool delta, const ScoreFinish *finish = nullptr);
// census specialisation of the Cost branch (svh_census_sgm.hip)
bool census_lane_kernels_available(int nWw, int D);
bool census_exact_regime(const SgmArgs &a, int nWw);
// exact regime: one sweep (regional winner keys + g map), then the min_p maps by parallel line scans
bool census_tiles_apply(const svh_context *ctx, const SgmArgs &a);
int dev_census_sweep_tiles(svh_context *ctx, Scratch &scr, const SgmArgs &a, const CostSource &cs, const WinnerOut &win);
int dev_census_tiles_from_keys(svh_context *ctx, Scratch &scr, const SgmArgs &a, const CostSource &cs, const uint2 *keys, float *gmap, bool gmap_ready,
                               const WinnerOut &win);
int dev_census_sweep_and_scans(svh_context *ctx, Scratch &scr, const SgmArgs &a, const CostSource &cs, float *mmap, uint2 **keys_out);
int dev_census_sweep(svh_context *ctx, const SgmArgs &a, const CostSource &cs, uint2 *keys, float *gmap /* may be null */);
// exact regime, index / disparity maps only (win.taps and win.keys null): no g map, no line scans, no min_p maps
int dev_census_winner(svh_context *ctx, Scratch &scr, const SgmArgs &a, const CostSource &cs, const WinnerOut &win);
int dev_census_scans(svh_context *ctx, const SgmArgs &a, const uint2 *keys, float *gmap, bool gmap_ready, float *mmap,
                     const int *skip_if_nonzero = nullptr);
int census_max_total_disparities();
int dev_census_finalize(svh_context *ctx, const SgmArgs &a, const CostSource &cs, const float *mmap, const uint2 *keys, const WinnerOut &win);
// any regime: literal float evaluation per voxel from given min_p maps
int dev_census_apply_select(svh_context *ctx, const SgmArgs &a, const CostSource &cs, const float *mmap, const WinnerOut &win);

int dev_extract_index(svh_context *ctx, int strategy, const float *cv, int64_t n_pixels, int D, int32_t *idx,
                      unsigned long long *keys, int key_index_offset, int key_total_D);
int dev_index_to_disp(svh_context *ctx, int ddir, const int32_t *idx, int64_t n, int32_t offset, int32_t *disp);
int dev_selected_cost(svh_context *ctx, const float *cv, const int32_t *idx, int64_t n, int D, float *out);
int dev_truncated_cv(svh_context *ctx, int sdir, int ddir, const float *cv, const int32_t *idx, int H, int W, int D, int h_r,
                     int v_r, int r, float *tcv);
int dev_refine(svh_context *ctx, int kernel, const float *tcv, const int32_t *raw, int64_t n, int T, float *refined);
int dev_keys_to_index(svh_context *ctx, int strategy, const unsigned long long *keys, int64_t n, int total_D, int32_t *idx);

} // namespace svh

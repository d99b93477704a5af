/* stevi_hip_test.h -- the A/B switches of libstevi_hip.so that exist for its parity tests.
 *
 * NOT part of the product surface (include/stevi_hip.h is): every switch below selects between two implementations of the same
 * function that give the same bits (or, for float cost volumes, the same values within the 1e-4 the north star states); the default is
 * the faster one, the other value is the side the tests cross-check it against.  Nothing a caller of the reference's API needs.
 * svh_test_set_option also accepts the five public options of svh_context_set_option, so a test can drive both through one entry. */
#ifndef STEVI_HIP_TEST_H
#define STEVI_HIP_TEST_H

#include "stevi_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* "census_fast_path" (default 1): 0 forces the general wave-per-line SGM kernels for census costs too (same results).
 * "census_sweep_rl" (default 1): 0 keeps the FP4 engine on its general kernel where the RightToLeft specialisation
 * (a multiple of 32 from 64 to 512 disparities, the search range ending at the target image's right edge) would run.  Same keys.
 * (Development A/Bs of that kernel, same keys again: 2 = column-major tile order everywhere; 3 = neighbouring column tiles per wave also
 * in the items at the right image border, where the default deals them out in serpentine order.)
 * "cost_volume_colsum" (default 1): float cost volumes of grey images (all functions but ZSAD) share the per-column sums of
 * neighbouring windows; 0 evaluates every window on its own (round 1's kernel).  Same results within rounding (1e-4 tolerance).
 * "patchmatch_pred_costs" (default 1): svh_cacheless_patch_match evaluates, before each propagation sweep and for every pixel in parallel,
 * the cost of the pixel against its predecessor's solution; the sweep uses it wherever the predecessor kept that solution (most pixels
 * after the first iterations) and evaluates a cost on the spot only behind an accepted candidate.  0: every step of a sweep evaluates
 * its cost.  Same result.
 * "patchmatch_scan_chunks" (default 1): from the second iteration on a propagation line decides the steps of 64 pixels at once: every pixel
 * knows from the pre-pass, for each state it can receive (the travelling candidate was picked up 1 .. 4 steps back, or further), whether
 * it keeps the candidate; those tables are composed by a prefix scan over the wave, and a cost is evaluated on the spot only where a
 * candidate from further back meets a pixel with a different solution.  0: step by step.  Same result.
 * "feature_volume_tiled" (default 1): svh_feature_cost_volume(_2d) with a float matching function processes the two feature volumes once
 * (mean subtracted, divided by the norm: the values of the reference's normalised volumes) and compares 64 pixels of a row with their
 * target records from LDS; 0: the per-voxel kernel processes both vectors of every voxel.  Same bits.
 * "feature_volume_records" (default 1): behind "feature_volume_tiled", feature vectors of up to 32 floats (the 17 superpixel means of the
 * reference's compressors, 3x3 / 5x5 unfolds) are compared with the TARGET record in registers -- a lane owns a record and walks the source
 * pixels that look at it, two at a time with packed multiplies and adds; 0: every target feature of every voxel is read from LDS (round 4).
 * Same bits.
 * "extract_index_wide" (default 1): svh_extract_selected_index (and every call that picks winners from a float volume) on rows of up to
 * 1 024 costs the packed kernel does not take -- more than 256 costs, or a count that is no multiple of four (2-D volumes: 289, 297) --
 * keeps up to sixteen costs per lane and combines by two all-reduces per pixel; 0: the wave-per-pixel kernel of round 1.  Same result.
 * "guided_shared" (default 1): svh_compute_guided_cv / svh_hierarchical_truncated_cost_volume on grey images with a search radius of at most
 * 3 stage the processed feature vectors (mean subtracted, divided by the norm: the reference's operations) of the target windows a group
 * of pixels looks at in LDS once, instead of every pixel processing every sample of every offset again.  1: a wave per 64 pixels, which
 * divides by a norm through its double reciprocal (the same quotient bits), where the grid fills the chip several times over, and a
 * block of 256 pixels with float divisions (round 4) on smaller grids; 2: the blocks always; 3: the waves always; 0: the per-pixel walk.
 * Same bits (tests/test_gpu_hierarchical.py).
 * "sgm_score_pad" (default 1): svh_sgm_cost_volume / svh_stereo_match, Score strategy, on 65 .. 511 disparities that are no multiple of 64
 * (the reference's own benchmark uses 160) aggregate a copy of the volume whose rows are padded to the next multiple of 64 with -inf
 * -- a pad never enters a maximum, and cost + anything stays -inf along every line -- so that the vector kernels, the banded sweep and
 * the winner records apply; the result is copied back without the pads.  0: the masked kernels on the caller's layout.  Same bits.
 * "fold_2d_offsets" (default 1): svh_unfold_cost_volume_2d on grey images with a function the column-sum kernel takes stages the
 * v + Dh - 1 target rows of all vertical offsets once per block and walks the (dh, dw) blocks in one launch (as many offsets as the
 * tile holds); 0 = one launch per vertical offset.  Same bits (tests/test_gpu_2d.py).
 * "census_tiles" (default 1): census + SGM calls that need the aggregated values (reduction keys, refinement taps that are not only
 * subtracted from one another; or any call with "census_winner_shortcut" 0) on a whole image of up to 1152 rows with 8 directions keep,
 * of the line recurrences' per-pass minima, only the values on the edges of 9-row x 64-column tiles, and the per-pixel kernel replays
 * the recurrences inside its tile; 0 writes the six per-pass maps and reads them back (round 2's pair of kernels).  Same results bit
 * for bit (tests/test_gpu_census_tiles.py); margins, 4 directions, taller images and row bands take the maps form regardless.
 * "cost_reduce_fused" (default 1): svh_stereo_match with a float matching function on grey images (windows up to 11 wide, not ZSAD)
 * lets the cost kernel reduce over the disparity axis while a block's waves hold a pixel's whole range: without SGM, for a call that asks
 * for the disparity map alone, the winner (extractSelectedIndex's rule: extremum, ties to the larger index, NaN never wins unless at index
 * 0) -- the volume is then never written and nothing reads it back; with a Cost-strategy SGM the two regional minima its line recurrences
 * run on (see "sgm_cost_two_minima"), so that the probing read of the volume is gone too.  0: extract_index / the probe read the volume.
 * Same maps bit for bit (tests/test_gpu_parity.py::test_winner_and_minima_reduced_inside_the_cost_kernel).
 * "sgm_cost_two_minima" (default 1): svh_sgm_cost_volume, Cost strategy, on a float volume that is not in the exact-integer regime
 * reads the volume ONCE for its line recurrences: a probe leaves every pixel's two regional minima (the smallest finite cost among the
 * disparities that look inside the image, and among those that look past its right border), and the recurrence of sgm.h:257-296 --
 * whose state is one number per line because of `min_a_cost = c_score` -- runs on those two numbers with the reference's float
 * operations (the per-disparity step is non-decreasing in the cost, so a region's minimum over d is the step of the region's minimum).
 * Needs magnitudes that cannot overflow along a line: the probe checks |c| <= 1e30 on the device and otherwise lets the sweeps of the
 * volume run (a finite |Pout| > 1e30 sends the call there directly).  0: one sweep of the volume per pass (rounds 1-3).  Same bits.
 * "sgm_score_finish_fused" (default 1): svh_stereo_match with a Score-strategy function and 8-direction SGM in the banded form below
 * lets the launch that writes a pixel's FINAL aggregated costs emit its winner -- index, disparity, the three truncatedCostVolume<Same>
 * taps for the refinement: DownLeft2UpRight for the pixels it visits (row + column < rows), the downward sweep for the others (the order
 * of the passes is fixed by sgm.h:379-389) -- instead of reading S back in extract_index / truncated_cost_volume; and when the caller did
 * not ask for sgm_cv, only the costs a later pass reads are stored at all.  0: the separate kernels.  Same maps bit for bit
 * (tests/test_gpu_sgm_score_fused.py).
 * "patchmatch_run_batches" (default 1, with "patchmatch_pred_costs"): a sweep step that has to evaluate a cost on the spot (its candidate is
 *   travelling) fetches the vectors of the next eight pixels of the line with it and evaluates them against the same candidate in the
 *   time of one; the following steps read their costs from that batch until a pixel rejects the candidate; 0: one evaluation per step.
 *   Same result.
 * "patchmatch_lookback" (default 1, with "patchmatch_pred_costs"): from the second iteration on the pre-pass also evaluates every pixel against
 *   the pre-sweep solutions two, three and four steps back along the sweep (only where they differ from the pixel's own), so that a candidate
 *   that travels needs an evaluation on the spot only from its fourth accepted step on; 0: one step back only.  Same result.
 * "patchmatch_search_form" (default 1): the random search of svh_cacheless_patch_match.  1: 64 candidates per wave, their vectors fetched
 *   coalesced 32 features at a time into a 9 KB LDS table of terms, lane e adding up row e in the reference's order.  0: round 4's kernel
 *   (a table row of nF floats per candidate, 24 candidates per wave at 1080p RGB 7x7).  2 / 3: a lane per candidate and no LDS, the lane
 *   fetching its candidate's target features itself (2) or forming them again from the target image and the decorator's (mean, norm) of
 *   the target pixel (3) -- experiments that lost at long vectors (the texture addresser serialises their scattered accesses) and are
 *   kept for the cross-check.  Same result in all four.
 * "sgm_score_fused" 3 (the public option takes 0 .. 2): the banded form with the 16-column strips forced that images narrower than about
 *   3000 columns replace by 8-column ones.
 * "census_sweep_rl" 2 / 3: development A/Bs of the RightToLeft sweep (2 = column-major tile order everywhere; 3 = neighbouring column
 *   tiles per wave also in the items at the right image border).  Same keys. */
int svh_test_set_option(svh_context *ctx, const char *name, int value);

#ifdef __cplusplus
}
#endif

#endif /* STEVI_HIP_TEST_H */

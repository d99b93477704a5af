"""On-demand (cacheless) cost volumes and PatchMatch (what examples/stereo-match runs) on the GPU.
The reference's PatchMatch is not reproducible (random_device seeds per thread); with the counter-based stream both sides
use, the device result is bit-identical to the oracle's statement-by-statement restatement."""
import numpy as np
import pytest

import oracle as so

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import libstevi_amd as sv  # noqa: E402
from libstevi_amd import matchingFunctions as MF  # noqa: E402

DEV = torch.device("cuda:0")


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def host(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else x


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def shifted_pair(rng, H, W, C, dv, dh):
    """source(i, j) == target(i + dv, j + dh): a pure translation, smooth enough for window matching to be informative"""
    base = rng.uniform(0, 1, (H + 24, W + 24, C)).astype(np.float32)
    base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) / np.float32(4)
    src = base[12:12 + H, 12:12 + W].copy()
    tgt = base[12 - dv:12 - dv + H, 12 - dh:12 - dh + W].copy()
    return np.ascontiguousarray(src), np.ascontiguousarray(tgt)


@pytest.mark.parametrize("func", [MF.CC, MF.NCC, MF.SSD, MF.SAD, MF.ZCC, MF.ZNCC, MF.ZSSD, MF.ZSAD])
def test_on_demand_features_bit_exact(rng, func):
    for shape, r in (((9, 13, 3), 2), ((7, 8), (1, 3)), ((5, 5, 1), 0)):
        img = rng.uniform(-1, 1, shape).astype(np.float32)
        hr, vr = (r, r) if np.isscalar(r) else r
        exp = so.on_demand_features(int(func), img, hr, vr)
        for mk in (lambda x: x, dev):
            got = host(sv.onDemandFeatures(func, mk(img), r))
            assert got.shape == exp.shape and np.array_equal(bits(got), bits(exp))
    # not the dense path's features: clamped borders and the RMS norm (on_demand_features_volume.h:128-133, :194-207)
    img = rng.uniform(-1, 1, (6, 6)).astype(np.float32)
    dense = so.feature_volume_for_match_func(int(func), so.unfold(img, 1, 1))
    assert not np.allclose(host(sv.onDemandFeatures(func, img, 1)), dense)


@pytest.mark.parametrize("func", [MF.ZNCC, MF.SAD, MF.NCC])
def test_on_demand_truncated_volume_matches_oracle(rng, func):
    src, tgt = shifted_pair(rng, 21, 27, 2, 1, -2)
    for rng2, nd in ((((0, 5), (0, 6)), 2), (((-3, 3), (-4, 2)), 2), ((-5, 5), 1), ((0, 7), 1)):
        off = sv.searchOffset2(rng2[0][0], rng2[0][1], rng2[1][0], rng2[1][1]) if nd == 2 else sv.searchOffset1(*rng2)
        lo = np.array([rng2[0][0], rng2[1][0]] if nd == 2 else [rng2[0]])
        hi = np.array([rng2[0][1], rng2[1][1]] if nd == 2 else [rng2[1]])
        disp = rng.integers(lo, hi + 1, (21, 27, nd)).astype(np.int32)
        for radius in (1, 2):
            exp = so.on_demand_truncated_cv(int(func), src, tgt, 2, 2, rng2, disp, radius)
            for mk in (lambda x: x, dev):
                got = host(sv.onDemandTruncatedCostVolume(func, mk(src), mk(tgt), 2, off, mk(disp), radius))
                assert got.shape == exp.shape and np.array_equal(bits(got), bits(exp))
    # with a range starting at 0 the centre tap is the cost at the given disparity; entries without a value hold FLT_MAX / FLT_MIN
    disp = np.zeros((21, 27, 1), np.int32)
    tcv = host(sv.onDemandTruncatedCostVolume(func, src, tgt, 2, sv.searchOffset1(0, 7), disp, 1))
    default = np.float32(np.finfo(np.float32).max if so.func_strategy(int(func)) == so.COST else np.finfo(np.float32).tiny)
    assert np.all(tcv[:, :, 0] == default)  # disparity -1 is outside the range
    fs, ft = so.on_demand_features(int(func), src, 2, 2), so.on_demand_features(int(func), tgt, 2, 2)
    i, j = 10, 11
    direct = {MF.ZNCC: float(fs[i, j].astype(np.float64) @ ft[i, j]), MF.NCC: float(fs[i, j].astype(np.float64) @ ft[i, j]),
              MF.SAD: float(np.abs(fs[i, j].astype(np.float64) - ft[i, j]).sum())}[func]
    assert abs(tcv[i, j, 1] - direct) < 1e-4
    assert sv.onDemandTruncatedCostVolume(func, src, tgt, 2, sv.searchOffset1(0, 7), np.zeros((21, 27, 2), np.int32), 1).size == 0  # :478-480


@pytest.mark.parametrize("func", [MF.ZNCC, MF.ZSAD, MF.SSD])
@pytest.mark.parametrize("nd", [1, 2])
def test_patch_match_bit_identical_to_oracle(rng, func, nd):
    src, tgt = shifted_pair(rng, 30, 41, 3, 0 if nd == 1 else 2, -3)
    search = ((-4, 4), (-5, 5)) if nd == 2 else (-5, 5)
    off = sv.searchOffset2(-4, 4, -5, 5) if nd == 2 else sv.searchOffset1(-5, 5)
    for seed, n_iter, n_random in ((1, 6, 4), (987654321, 3, 2), (7, 1, 0)):
        exp, its = so.cacheless_patch_match(int(func), nd, src, tgt, 2, 2, search, n_iter, n_random, seed)
        for mk in (lambda x: x, dev):
            got, gits = sv.cachelessPatchMatch(func, mk(src), mk(tgt), 2, off, n_iter, n_random, seed, return_iterations=True)
            assert np.array_equal(host(got), exp) and gits == its


def test_patch_match_recovers_a_planted_flow(rng):
    """the acceptance the reference's own benchmark applies to PatchMatch (benchmarkStereoMatchingModels.cpp: proportion of correct
    pixels) on a synthetic translation: ZNCC 5x5 windows, 13 x 13 offsets, 10 iterations as examples/stereo-match defaults"""
    src, tgt = shifted_pair(rng, 96, 128, 3, -2, 3)
    disp = host(sv.cachelessPatchMatch(MF.ZNCC, dev(src), dev(tgt), 2, sv.searchOffset2(-6, 6, -6, 6), 10, 4, seed=2024))
    inner = disp[8:-8, 8:-8]
    assert np.mean((inner[..., 0] == -2) & (inner[..., 1] == 3)) > 0.97
    # the --refine chain of examples/stereo-match (main.cpp:198-210): truncated on-demand volume + equiangular 2-D refinement
    tcv = sv.onDemandTruncatedCostVolume(MF.ZNCC, dev(src), dev(tgt), 2, sv.searchOffset2(-6, 6, -6, 6), dev(disp), 1)
    refined = host(sv.refineDisp2dCostInterpolation(sv.InterpolationKernel.Equiangular, tcv, dev(disp)))
    assert refined.shape == disp.shape and np.isfinite(refined).all()


def test_patch_match_argument_rules(rng):
    a = rng.uniform(0, 1, (8, 9, 3)).astype(np.float32)
    assert sv.cachelessPatchMatch(MF.ZNCC, a, a[:, :, :2].copy(), 1, sv.searchOffset2(-1, 1, -1, 1)).size == 0  # channel mismatch, patchmatch.h:583-585
    assert sv.cachelessPatchMatch(MF.ZNCC, a, a[:7].copy(), 1, sv.searchOffset1(-1, 1)).size == 0  # stereo needs equal row counts, :587-591
    assert sv.cachelessPatchMatch(MF.ZNCC, a, a[:7].copy(), 1, sv.searchOffset2(-1, 1, -1, 1)).shape == (8, 9, 2)  # flow does not
    with pytest.raises(Exception):
        sv.cachelessPatchMatch(MF.CENSUS, a, a, 1, sv.searchOffset1(-1, 1))


@pytest.mark.parametrize("func", [MF.ZNCC, MF.SAD])
@pytest.mark.parametrize("nd", [1, 2])
def test_patch_match_sweeps_with_predecessor_costs_same_result(rng, func, nd):
    """Option "patchmatch_pred_costs" (default): the sweeps take a pixel's cost against its predecessor's unchanged solution from a
    parallel pre-pass and against its own solution from its own cost; 0: every step evaluates its cost.  Same solutions, same iteration
    counts: lines longer than the 64 pixels a register chunk holds, both sweep directions (iterations 0 - 3), grey and colour."""
    for (H, W, C, n_iter, n_random) in [(70, 150, 1, 5, 4), (90, 67, 3, 4, 2), (3, 200, 1, 4, 0), (130, 5, 1, 4, 3)]:
        src, tgt = shifted_pair(rng, H, W, C, 0 if nd == 1 else 1, -2)
        off = sv.searchOffset2(-3, 3, -6, 6) if nd == 2 else sv.searchOffset1(-6, 6)
        d_src, d_tgt = dev(src), dev(tgt)
        outs = []
        for opt in (1, 0):
            sv.set_test_option(d_src, "patchmatch_pred_costs", opt)
            try:
                outs.append(sv.cachelessPatchMatch(func, d_src, d_tgt, 2, off, n_iter, n_random, 77, return_iterations=True))
            finally:
                sv.set_test_option(d_src, "patchmatch_pred_costs", 1)
        assert np.array_equal(host(outs[0][0]), host(outs[1][0])) and outs[0][1] == outs[1][1], (H, W, C)


@pytest.mark.parametrize("func", [MF.ZNCC, MF.SAD, MF.SSD])
@pytest.mark.parametrize("nd", [1, 2])
def test_patch_match_search_lane_per_candidate_same_result(rng, func, nd):
    """Option "patchmatch_search_form": 1 (default, round 5) 64 candidates per wave through a 32-feature LDS table; 0 round 4's batched kernel;
    2 / 3 a lane per candidate without LDS (fetching the stored target features / forming them again from the target image).  Same
    solutions and iteration counts -- feature counts that
    are and are not multiples of four (25, 75, 27, 9), candidate counts that do and do not divide 64, images smaller than a wave's pixels."""
    for (H, W, C, r, n_iter, n_random) in [(70, 150, 1, 2, 5, 4), (40, 67, 3, 2, 4, 2), (33, 50, 3, 1, 4, 3), (20, 31, 1, 1, 3, 5), (3, 5, 1, 1, 3, 7), (24, 24, 1, 2, 3, 64)]:
        src, tgt = shifted_pair(rng, H, W, C, 0 if nd == 1 else 1, -2)
        off = sv.searchOffset2(-3, 3, -6, 6) if nd == 2 else sv.searchOffset1(-6, 6)
        d_src, d_tgt = dev(src), dev(tgt)
        outs = []
        for opt in (1, 2, 3, 0):
            sv.set_test_option(d_src, "patchmatch_search_form", opt)
            try:
                outs.append(sv.cachelessPatchMatch(func, d_src, d_tgt, r, off, n_iter, n_random, 91, return_iterations=True))
            finally:
                sv.set_test_option(d_src, "patchmatch_search_form", 1)
        for other in outs[1:]:
            assert np.array_equal(host(outs[0][0]), host(other[0])) and outs[0][1] == other[1], (H, W, C, n_random)


@pytest.mark.parametrize("func", [MF.ZNCC, MF.SAD])
@pytest.mark.parametrize("nd", [1, 2])
def test_patch_match_run_batches_same_result(rng, func, nd):
    """Option "patchmatch_run_batches" (default, round 5): a step that evaluates a cost on the spot evaluates the next eight pixels of its line
    against the same travelling candidate with it (first iteration); "patchmatch_lookback": the pre-pass evaluates against the solutions up to
    four steps back (later iterations), and chunks of 64 pixels that agree are skipped; 0 / 0: one evaluation per step.  Same solutions and iteration counts on images whose
    planted shift makes candidates travel along whole lines (lines longer than one batch, shorter than one, both sweep directions), with
    a border the candidate's targets leave, grey and colour."""
    for (H, W, C, r, n_iter, n_random) in [(40, 300, 1, 2, 4, 2), (150, 70, 3, 1, 5, 4), (5, 700, 1, 1, 4, 0), (260, 6, 1, 1, 4, 1), (64, 64, 1, 2, 6, 3)]:
        src, tgt = shifted_pair(rng, H, W, C, 0 if nd == 1 else 2, -3)
        off = sv.searchOffset2(-3, 3, -6, 6) if nd == 2 else sv.searchOffset1(-6, 6)
        d_src, d_tgt = dev(src), dev(tgt)
        outs = []
        for batches, lookback in ((1, 1), (0, 0), (1, 0), (0, 1)):
            sv.set_test_option(d_src, "patchmatch_run_batches", batches)
            sv.set_test_option(d_src, "patchmatch_lookback", lookback)
            try:
                outs.append(sv.cachelessPatchMatch(func, d_src, d_tgt, r, off, n_iter, n_random, 5, return_iterations=True))
            finally:
                sv.set_test_option(d_src, "patchmatch_run_batches", 1)
                sv.set_test_option(d_src, "patchmatch_lookback", 1)
        for other in outs[1:]:
            assert np.array_equal(host(outs[0][0]), host(other[0])) and outs[0][1] == other[1], (H, W, C, n_random)


@pytest.mark.parametrize("func", [MF.ZNCC, MF.SAD, MF.SSD])
@pytest.mark.parametrize("nd", [1, 2])
def test_patch_match_scanned_chunks_same_result(rng, func, nd):
    """Option "patchmatch_scan_chunks" (default, round 5): from the second iteration on a sweep line decides the steps of 64 pixels at once --
    per-pixel transition tables (which received state keeps the travelling candidate) composed by a prefix scan, on-the-spot evaluations
    only where a candidate from further back than the pre-pass looked meets a different solution -- with and without the look-back depths;
    0: step by step.  Same solutions, costs and iteration counts: lines longer and shorter than a chunk, a last chunk that is not full,
    both sweep directions over the iterations, candidates that travel across whole lines (planted shift) and lines that never settle
    (noise), borders the candidate's targets leave, grey and colour."""
    for (H, W, C, r, n_iter, n_random, noise) in [(40, 300, 1, 2, 6, 2, 0.0), (150, 70, 3, 1, 6, 4, 0.0), (5, 700, 1, 1, 5, 0, 0.0), (260, 6, 1, 1, 5, 1, 0.0),
                                                  (64, 129, 1, 2, 8, 3, 0.5), (33, 200, 1, 1, 9, 0, 1.0)]:
        src, tgt = shifted_pair(rng, H, W, C, 0 if nd == 1 else 2, -3)
        if noise:
            tgt = (tgt + noise * rng.uniform(-1, 1, tgt.shape)).astype(np.float32)
        off = sv.searchOffset2(-3, 3, -6, 6) if nd == 2 else sv.searchOffset1(-6, 6)
        d_src, d_tgt = dev(src), dev(tgt)
        outs = []
        for scan, lookback in ((1, 1), (0, 1), (1, 0), (0, 0)):
            sv.set_test_option(d_src, "patchmatch_scan_chunks", scan)
            sv.set_test_option(d_src, "patchmatch_lookback", lookback)
            try:
                outs.append(sv.cachelessPatchMatch(func, d_src, d_tgt, r, off, n_iter, n_random, 5, return_iterations=True))
            finally:
                sv.set_test_option(d_src, "patchmatch_scan_chunks", 1)
                sv.set_test_option(d_src, "patchmatch_lookback", 1)
        for other in outs[1:]:
            assert np.array_equal(host(outs[0][0]), host(other[0])) and outs[0][1] == other[1], (H, W, C, n_random, noise)


@pytest.mark.parametrize("func, so_func", [(MF.ZNCC, so.ZNCC), (MF.NCC, so.NCC), (MF.SAD, so.SAD), (MF.ZSSD, so.ZSSD), (MF.CC, so.CC)])
@pytest.mark.parametrize("nd", [1, 2])
def test_patch_match_on_feature_volumes_bit_identical_to_oracle(rng, func, so_func, nd):
    """patchMatch (patchmatch.h:496-558): PatchMatch on feature volumes the caller built, the reference's cached cost volume behind it.  The
    oracle restates the entry point (features processed twice for zero-mean / normalised functions) and shares the iteration with
    cachelessPatchMatch; host and device arrays, grey and colour unfolds, with and without an initial solution."""
    for (H, W, C, r, n_iter, n_random) in [(30, 47, 1, 2, 4, 4), (21, 33, 3, 1, 3, 2)]:
        src, tgt = shifted_pair(rng, H, W, C, 0 if nd == 1 else 1, -2)
        fs, ft = so.unfold(src, r, r), so.unfold(tgt, r, r)
        off = sv.searchOffset2(-3, 3, -6, 6) if nd == 2 else sv.searchOffset1(-6, 6)
        rng_t = ((-3, 3), (-6, 6)) if nd == 2 else (-6, 6)
        want, its = so.patch_match(so_func, nd, fs, ft, rng_t, n_iter, n_random, seed=31)
        for a, b in ((fs, ft), (dev(fs), dev(ft))):
            got, gits = sv.patchMatch(func, a, b, off, n_iter, n_random, seed=31, return_iterations=True)
            assert np.array_equal(host(got), want) and gits == its, (H, W, C)
        init = rng.integers(-6, 7, (H, W, nd)).astype(np.int32)
        if nd == 2:
            init[..., 0] = np.clip(init[..., 0], -3, 3)
        want_i, _ = so.patch_match(so_func, nd, fs, ft, rng_t, n_iter, n_random, seed=31, init=init)
        assert np.array_equal(host(sv.patchMatch(func, dev(fs), dev(ft), off, n_iter, n_random, seed=31, initial=dev(init))), want_i)
        assert np.array_equal(sv.patchMatch(func, fs, ft, off, n_iter, n_random, seed=31, initial=init), want_i)
    # argument rules: feature counts that differ, row counts with a stereo search -> empty (patchmatch.h:529-537)
    a = rng.uniform(0, 1, (8, 9, 5)).astype(np.float32)
    assert sv.patchMatch(MF.ZNCC, a, a[:, :, :4].copy(), sv.searchOffset1(-1, 1)).size == 0
    assert sv.patchMatch(MF.ZNCC, a, a[:7].copy(), sv.searchOffset1(-1, 1)).size == 0
    assert sv.patchMatch(MF.ZNCC, a, a[:7].copy(), sv.searchOffset2(-1, 1, -1, 1)).shape == (8, 9, 2)


def test_cacheless_patch_match_with_an_initial_solution(rng):
    """the reference's `initializer` callback (patchmatch.h:598-605): its map replaces the random draw"""
    H, W, C, r = 26, 41, 3, 1
    src, tgt = shifted_pair(rng, H, W, C, 1, -2)
    off = sv.searchOffset2(-3, 3, -6, 6)
    init = np.stack([rng.integers(-3, 4, (H, W)), rng.integers(-6, 7, (H, W))], -1).astype(np.int32)
    want, its = so.cacheless_patch_match(so.ZNCC, 2, src, tgt, r, r, ((-3, 3), (-6, 6)), 4, 3, seed=8, init=init)
    got, gits = sv.cachelessPatchMatch(MF.ZNCC, dev(src), dev(tgt), r, off, 4, 3, seed=8, return_iterations=True, initial=dev(init))
    assert np.array_equal(host(got), want) and gits == its
    assert np.array_equal(sv.cachelessPatchMatch(MF.ZNCC, src, tgt, r, off, 4, 3, seed=8, initial=init), want)
    # an initial solution outside the search range has no value: a score function replaces it at the first candidate that has one
    far = np.full((H, W, 2), 50, np.int32)
    want_f, _ = so.cacheless_patch_match(so.ZNCC, 2, src, tgt, r, r, ((-3, 3), (-6, 6)), 3, 2, seed=8, init=far)
    assert np.array_equal(host(sv.cachelessPatchMatch(MF.ZNCC, dev(src), dev(tgt), r, off, 3, 2, seed=8, initial=dev(far))), want_f)

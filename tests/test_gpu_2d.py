"""2-D disparity volumes on the GPU: parity with the oracle and the reference's own
property tests (testCorrelation2d.cpp:75-191) run directly against the HIP path."""
import numpy as np
import pytest

import oracle as so
from helpers import naive_window_cost

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import libstevi_amd as sv  # noqa: E402
from libstevi_amd import matchingFunctions as MF  # noqa: E402

DEV = torch.device("cuda:0")
NAMES = {"NCC": MF.NCC, "ZNCC": MF.ZNCC, "SSD": MF.SSD, "ZSSD": MF.ZSSD, "SAD": MF.SAD, "ZSAD": MF.ZSAD, "CC": MF.CC, "ZCC": MF.ZCC}


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def host(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else x


@pytest.mark.parametrize("func", list(NAMES.values()) + [MF.CENSUS])
@pytest.mark.parametrize("ddir", [sv.dispDirection.RightToLeft, sv.dispDirection.LeftToRight])
def test_2d_volume_matches_oracle(rng, func, ddir):
    left = rng.uniform(-1, 1, (19, 41)).astype(np.float32)
    right = rng.uniform(-1, 1, (19, 41)).astype(np.float32)
    for (h_r, v_r, r0, r1) in [(1, 1, (-2, 3), (-4, 5)), (4, 4, (0, 2), (0, 6)), (2, 3, (-25, -20), (3, 3))]:
        exp = so.unfold_cost_volume_2d(int(func), left, right, h_r, v_r, r0, r1, int(ddir))
        for mk in (lambda x: x, dev):
            got = host(sv.unfoldBased2dDisparityCostVolume(func, mk(left), mk(right), h_r, v_r, sv.searchOffset2(r0[0], r0[1], r1[0], r1[1]), ddir))
            assert got.shape == exp.shape
            if func == MF.CENSUS:
                assert np.array_equal(got, exp)
            else:
                assert np.array_equal(np.isnan(got), np.isnan(exp))
                ok = ~np.isnan(exp)
                assert np.all(np.abs(got[ok] - exp[ok]) <= 1e-4 * np.maximum(1, np.abs(exp[ok])))
        strat = so.func_strategy(int(func))
        idx = sv.extractSelected2dIndex(strat, got)
        assert np.array_equal(host(idx), so.extract_index_2d(got, strat))
        off = sv.searchOffset2(r0[0], r0[1], r1[0], r1[1])
        assert np.array_equal(host(sv.selected2dIndexToDisp(idx, off)), so.index_2d_to_disp(host(idx), r0[0], r1[0]))
        tcv = host(sv.truncatedBidirectionaCostVolume(dev(got), dev(host(idx)), 1, 2))
        exp_t = so.truncated_bidirectional_cv(got, host(idx), 1, 2)
        assert np.array_equal(np.isnan(tcv), np.isnan(exp_t)) and np.array_equal(tcv[~np.isnan(exp_t)], exp_t[~np.isnan(exp_t)])


def test_2d_volume_shape_errors(rng):
    a = rng.uniform(-1, 1, (8, 9)).astype(np.float32)
    b = rng.uniform(-1, 1, (8, 10)).astype(np.float32)
    assert sv.unfoldBased2dDisparityCostVolume(MF.SAD, a, b, 1, 1, sv.searchOffset2(0, 1, 0, 1)).size == 0  # cross_correlations.h:808-810
    assert sv.unfoldBased2dDisparityCostVolume(MF.SAD, a, a, 1, 1, sv.searchOffset2(2, 1, 0, 1)).size == 0  # :338-340


@pytest.mark.parametrize("h_r,v_r,disp_w,disp_h", [(1, 1, 5, 5), (3, 3, 5, 5), (5, 1, 5, 3), (1, 5, 3, 5), (5, 5, 5, 5)])
@pytest.mark.parametrize("name", ["NCC", "ZNCC", "SSD", "ZSSD", "SAD", "ZSAD"])
def test_reference_test2dMatching_on_gpu(rng, name, h_r, v_r, disp_w, disp_h):
    """testCorrelation2d.cpp:75-127 against the HIP path itself, tolerance 1e-3 as in the reference."""
    h, w = 2 * v_r + disp_h + 1, 2 * h_r + disp_w + 1
    left = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    right = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    cv = host(sv.unfoldBased2dDisparityCostVolume(NAMES[name], left, right, h_r, v_r, sv.searchOffset2(0, disp_h, 0, disp_w)))
    w1 = right[0:2 * v_r + 1, 0:2 * h_r + 1]
    for i in range(disp_h):
        for j in range(disp_w):
            assert abs(naive_window_cost(name, w1, left[i:i + 2 * v_r + 1, j:j + 2 * h_r + 1]) - cv[v_r, h_r, i, j]) < 1e-3


@pytest.mark.parametrize("h_r,v_r", [(1, 1), (3, 3), (5, 1), (1, 5)])
@pytest.mark.parametrize("name", ["NCC", "ZNCC", "SSD", "ZSSD", "SAD", "ZSAD"])
def test_reference_test2dDisparity_on_gpu(rng, name, h_r, v_r):
    """testCorrelation2d.cpp:131-191: planted shifts at both ends of the range are recovered exactly."""
    disp_w = disp_h = 3
    h, w = 2 * v_r + disp_h + 1, 2 * h_r + disp_w + 1
    source = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    target0 = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    targetmax = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    for i in range(2 * v_r + 1):
        for j in range(2 * h_r + 1):
            target0[i, j] = source[i + 1, j + 1]
            targetmax[i + disp_h, j + disp_w] = source[i + 1, j + 1]
    off = sv.searchOffset2(-1, disp_h - 1, -1, disp_w - 1)
    strat = sv.matchFuncStrategy(NAMES[name])
    for target, expected in ((target0, (-1, -1)), (targetmax, (disp_h - 1, disp_w - 1))):
        cv = sv.unfoldBased2dDisparityCostVolume(NAMES[name], dev(target), dev(source), h_r, v_r, off)
        disp = host(sv.selected2dIndexToDisp(sv.extractSelected2dIndex(strat, cv), off))
        assert tuple(disp[v_r + 1, h_r + 1]) == expected


# ---- 2-D cost-based refinement (cost_based_refinement.h:165-436) --------------------------
from helpers import isotropic_patch, rotated_quadric_patch, refined_2d_mismatch  # noqa: E402

IK, ISO = sv.InterpolationKernel, sv.IsotropyHypothesis


def _flow_chain(rng, func, r0, r1, shape=(37, 53)):
    """a translated random image -> 2-D volume -> winner -> truncated volume (NaN where the winner sits on the range border)"""
    H, W = shape
    base = rng.uniform(-1, 1, (H + 8, W + 8)).astype(np.float32)
    src = base[4:4 + H, 4:4 + W].copy()
    tgt = (0.6 * base[5:5 + H, 2:2 + W] + 0.4 * base[6:6 + H, 3:3 + W]).astype(np.float32)  # sub-pixel mixture of two shifts
    cv = so.unfold_cost_volume_2d(int(func), tgt, src, 2, 2, (-3, 3), (-3, 3))
    strat = so.func_strategy(int(func))
    idx = so.extract_index_2d(cv, strat)
    tcv = so.truncated_bidirectional_cv(cv, idx, r0, r1)
    return tcv, so.index_2d_to_disp(idx, -3, -3)


@pytest.mark.parametrize("func", [MF.ZNCC, MF.SSD, MF.SAD])
@pytest.mark.parametrize("kernel", [IK.Equiangular, IK.Parabola, IK.Gaussian])
@pytest.mark.parametrize("iso", [ISO.Isotropic, ISO.Anisotropic])
def test_refine_2d_matches_oracle(rng, func, kernel, iso):
    """float tolerance 1e-4 on the refined disparities (the north-star tolerance for refinement outputs)"""
    for (r0, r1) in [(1, 1), (2, 2), (1, 3)]:
        tcv, raw = _flow_chain(rng, func, r0, r1)
        if kernel == IK.Gaussian:
            tcv = np.abs(tcv) + np.float32(0.05)  # log() of the costs
        exp = so.refine_disp_2d(tcv, raw, int(kernel), int(iso))
        for mk in (lambda x: x, dev):
            got = host(sv.refineDisp2dCostInterpolation(kernel, mk(tcv), mk(raw), iso))
            assert got.shape == exp.shape and not np.isnan(got).any()
            bad, flipped = refined_2d_mismatch(got, exp, raw)
            assert bad == 0.0 and flipped <= 0.002, (bad, flipped)
        assert np.mean(np.any(exp != raw, axis=-1)) > 0.3  # the refinement does something on this data


@pytest.mark.parametrize("func", [MF.ZNCC, MF.SSD])
@pytest.mark.parametrize("kernel", [IK.Parabola, IK.Gaussian])
def test_refine_2d_patch_matches_oracle(rng, func, kernel):
    """closed-form normal equations on the device against the oracle's numerical 9x6 least squares"""
    for (r0, r1) in [(1, 1), (2, 1)]:
        tcv, raw = _flow_chain(rng, func, r0, r1)
        if kernel == IK.Gaussian:
            tcv = np.abs(tcv) + np.float32(0.05)
        exp = so.refine_disp_2d_patch(tcv, raw, int(kernel))
        for mk in (lambda x: x, dev):
            got = host(sv.refineDisp2dCostPatchInterpolation(kernel, mk(tcv), mk(raw)))
            bad, flipped = refined_2d_mismatch(got, exp, raw)
            assert bad == 0.0 and flipped <= 0.002, (bad, flipped)
        assert np.mean(np.any(exp != raw, axis=-1)) > 0.3


def test_refine_2d_shape_and_argument_rules(rng):
    raw = np.zeros((4, 5, 2), np.int32)
    assert sv.refineDisp2dCostInterpolation(IK.Parabola, np.zeros((4, 5, 1, 3), np.float32), raw).size == 0  # :180-182
    assert sv.refineDisp2dCostInterpolation(IK.Parabola, np.zeros((4, 5, 3, 4), np.float32), raw).size == 0
    assert sv.refineDisp2dCostPatchInterpolation(IK.Parabola, np.zeros((4, 5, 2, 3), np.float32), raw).size == 0  # :393-395
    with pytest.raises(Exception):
        sv.refineDisp2dCostPatchInterpolation(IK.Equiangular, np.zeros((4, 5, 3, 3), np.float32), raw)  # static_assert :83
    with pytest.raises(Exception):
        sv.refineDisp2dCostInterpolation(IK.Parabola, np.zeros((4, 5, 3, 3), np.float32), np.zeros((4, 6, 2), np.int32))
    flat = np.ones((4, 5, 3, 3), np.float32)  # 0/0 -> NaN -> both deltas zeroed (:362-366)
    raw = rng.integers(-5, 5, (4, 5, 2)).astype(np.int32)
    assert np.array_equal(host(sv.refineDisp2dCostInterpolation(IK.Parabola, dev(flat), dev(raw))), raw.astype(np.float32))
    assert np.array_equal(host(sv.refineDisp2dCostPatchInterpolation(IK.Parabola, dev(flat), dev(raw))), raw.astype(np.float32))


@pytest.mark.parametrize("trial", range(6))
def test_reference_cost_refinement_tests_on_gpu(rng, trial):
    """testCostRefinement.cpp:60-221 run against the HIP path: planted quadric minima are recovered."""
    zero = np.zeros((1, 1, 2), np.int32)
    s = rng.uniform(-1, 1)
    s = 0.5 if abs(s) < 1e-4 else s
    ex, ey = rng.uniform(-1, 1, 2)
    ref = host(sv.refineDisp2dCostInterpolation(IK.Parabola, dev(isotropic_patch(s, ex, ey)), dev(zero), ISO.Isotropic))
    assert ref[0, 0, 0] == pytest.approx(ex, rel=1e-5, abs=1e-6) and ref[0, 0, 1] == pytest.approx(ey, rel=1e-5, abs=1e-6)
    s = np.copysign(max(abs(rng.uniform(-1, 1)), 0.5), rng.uniform(-1, 1))
    d1, d2 = s * rng.uniform(0.7, 1, 2)
    b = rng.uniform(-0.5, 0.5, 2)
    ref = host(sv.refineDisp2dCostInterpolation(IK.Parabola, dev(rotated_quadric_patch(d1, d2, 0.1 * rng.uniform(-1, 1), b, 5)), dev(zero), ISO.Anisotropic))
    assert abs(ref[0, 0, 0] - b[0]) < 2e-6 and abs(ref[0, 0, 1] - b[1]) < 2e-6
    ref = host(sv.refineDisp2dCostPatchInterpolation(IK.Parabola, dev(rotated_quadric_patch(d1, d2, 3 * rng.uniform(-1, 1), b, 3)), dev(zero)))
    assert ref[0, 0, 0] == pytest.approx(b[0], rel=1e-5, abs=2e-6) and ref[0, 0, 1] == pytest.approx(b[1], rel=1e-5, abs=2e-6)


@pytest.mark.parametrize("func", list(NAMES.values()) + [MF.CENSUS])
def test_feature_volume_2d_matches_oracle(rng, func):
    """featureVolume2CostVolume with a searchOffset<2> on raw feature volumes (cross_correlations.h:310-374): widths may differ"""
    fl = rng.uniform(-10, 10, (11, 17, 7)).astype(np.float32)
    fr = rng.uniform(-10, 10, (11, 14, 7)).astype(np.float32)
    if func == MF.CENSUS:
        fl, fr = np.repeat(fl, 6, axis=2)[:, :, :40], np.repeat(fr, 6, axis=2)[:, :, :40]  # at least one full census word
        fl = (fl + rng.uniform(-3, 3, fl.shape)).astype(np.float32)
        fr = (fr + rng.uniform(-3, 3, fr.shape)).astype(np.float32)
    for ddir in (sv.dispDirection.RightToLeft, sv.dispDirection.LeftToRight):
        for r0, r1 in (((-2, 2), (-3, 1)), ((1, 1), (0, 5))):
            exp = so.feature_cost_volume_2d(int(func), fl, fr, r0, r1, int(ddir))
            for mk in (lambda x: x, dev):
                got = host(sv.featureVolume2CostVolume(func, mk(fl), mk(fr), sv.searchOffset2(r0[0], r0[1], r1[0], r1[1]), ddir))
                assert got.shape == exp.shape
                if func == MF.CENSUS:
                    assert np.array_equal(got, exp)
                else:
                    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))  # the per-voxel kernel follows the reference's operation order
    assert sv.featureVolume2CostVolume(MF.SAD, fl, fr[:10], sv.searchOffset2(0, 1, 0, 1)).size == 0


@pytest.mark.parametrize("func", [MF.ZNCC, MF.NCC, MF.SSD, MF.ZSSD, MF.SAD, MF.CC, MF.ZCC])
def test_2d_volume_folded_offsets_same_bits(rng, func):
    """fold_2d_offsets (one launch of the column-sum kernel for as many vertical offsets as its tile holds, the last block of an offset
    starting at Dw - 16) against a launch per vertical offset: the same bits; and against the oracle within the float tolerance.  Ranges
    that are / are not multiples of 16, shorter than 16, offsets that leave the image on every side, more offsets than one tile holds."""
    cases = [((31, 300), 3, 3, (-4, 4), (-16, 16)),   # 9 x 33: two whole blocks + the shifted one
             ((23, 150), 2, 2, (-8, 8), (-8, 8)),     # 17 x 17
             ((17, 140), 1, 2, (-1, 1), (0, 15)),     # exactly one block per offset
             ((12, 130), 2, 1, (-3, 2), (-4, 4)),     # shorter than a block: the per-cost store path
             ((40, 131), 5, 5, (-30, 29), (-3, 40)),  # 60 offsets of 44: several groups; rows far outside the image
             ((9, 64), 4, 4, (2, 4), (-70, -40)),     # (31) every target column left of the image for most pixels
             ((20, 129), 2, 2, (-2, 2), (-17, 17)),   # 35: an odd range that is not 16 k + 1 (last block seventeen wide, starting early)
             ((14, 140), 1, 1, (0, 2), (-9, 8)),      # 18: even, the last block starts at 2
             ((11, 127), 3, 1, (-1, 0), (-9, 9))]     # 19
    for (shape, h_r, v_r, r0, r1) in cases:
        left = rng.uniform(-1, 1, shape).astype(np.float32)
        right = rng.uniform(-1, 1, shape).astype(np.float32)
        dl, dr = dev(left), dev(right)
        off = sv.searchOffset2(r0[0], r0[1], r1[0], r1[1])
        sv.set_test_option(dl, "fold_2d_offsets", 0)
        try:
            per_offset = host(sv.unfoldBased2dDisparityCostVolume(func, dl, dr, h_r, v_r, off))
        finally:
            sv.set_test_option(dl, "fold_2d_offsets", 1)
        folded = host(sv.unfoldBased2dDisparityCostVolume(func, dl, dr, h_r, v_r, off))
        assert folded.shape == per_offset.shape == (shape[0], shape[1], r0[1] - r0[0] + 1, r1[1] - r1[0] + 1)
        assert np.array_equal(folded.view(np.uint32), per_offset.view(np.uint32)), (shape, h_r, v_r, r0, r1)
        exp = so.unfold_cost_volume_2d(int(func), left, right, h_r, v_r, r0, r1)
        assert np.array_equal(np.isnan(folded), np.isnan(exp))
        ok = ~np.isnan(exp)
        assert np.all(np.abs(folded[ok] - exp[ok]) <= 1e-4 * np.maximum(1, np.abs(exp[ok])))


@pytest.mark.parametrize("func", [MF.ZNCC, MF.NCC, MF.SSD, MF.ZSSD, MF.SAD, MF.CC])
def test_2d_volume_folded_offsets_colour_images_same_bits(rng, func):
    """the same for images of interleaved channels (round 5: the column-sum kernel takes several vertical offsets of a colour image in one
    launch too -- the channels are tile rows of their own, an image row down is C tile rows down): folded against a launch per offset, bit
    for bit, and against the oracle within the float tolerance; two, three and four channels."""
    cases = [((25, 200, 3), 3, 3, (-4, 4), (-16, 16)),  # 9 x 33 on RGB: what examples/stereo-match-style flow volumes use
             ((19, 140, 3), 2, 1, (-2, 3), (-8, 8)),    # 17
             ((14, 131, 2), 1, 2, (-1, 1), (0, 15)),
             ((12, 130, 4), 2, 2, (-6, 5), (-4, 4)),    # several groups; shorter than a block
             ((10, 129, 3), 1, 1, (-1, 0), (-17, 17))]  # 35: the seventeen-wide last block
    for (shape, h_r, v_r, r0, r1) in cases:
        left = rng.uniform(-1, 1, shape).astype(np.float32)
        right = rng.uniform(-1, 1, shape).astype(np.float32)
        dl, dr = dev(left), dev(right)
        off = sv.searchOffset2(r0[0], r0[1], r1[0], r1[1])
        sv.set_test_option(dl, "fold_2d_offsets", 0)
        try:
            per_offset = host(sv.unfoldBased2dDisparityCostVolume(func, dl, dr, h_r, v_r, off))
        finally:
            sv.set_test_option(dl, "fold_2d_offsets", 1)
        folded = host(sv.unfoldBased2dDisparityCostVolume(func, dl, dr, h_r, v_r, off))
        assert folded.shape == per_offset.shape == (shape[0], shape[1], r0[1] - r0[0] + 1, r1[1] - r1[0] + 1)
        assert np.array_equal(folded.view(np.uint32), per_offset.view(np.uint32)), (shape, h_r, v_r, r0, r1)
        exp = so.unfold_cost_volume_2d(int(func), left, right, h_r, v_r, r0, r1)
        assert np.array_equal(np.isnan(folded), np.isnan(exp))
        ok = ~np.isnan(exp)
        assert np.all(np.abs(folded[ok] - exp[ok]) <= 1e-4 * np.maximum(1, np.abs(exp[ok])))

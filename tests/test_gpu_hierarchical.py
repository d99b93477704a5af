"""Hierarchical matching on the GPU: averagePoolingDownsample, computeGuidedCV and
hiearchicalTruncatedCostVolume against the oracle, and the reference's own test (testCorrelationHierarchical.cpp) run
directly against the HIP path."""
import numpy as np
import pytest

import oracle as so
from helpers import hierarchical_acceptance, hierarchical_scene, parallax_pair

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import libstevi_amd as sv  # noqa: E402
from libstevi_amd import matchingFunctions as MF  # noqa: E402

DEV = torch.device("cuda:0")
R2L, L2R = sv.dispDirection.RightToLeft, sv.dispDirection.LeftToRight
FLOAT_FUNCS = [MF.CC, MF.NCC, MF.SSD, MF.SAD, MF.ZCC, MF.ZNCC, MF.ZSSD, MF.ZSAD]


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def host(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else x


@pytest.mark.parametrize("shape", [(7, 10), (8, 8), (33, 17), (6, 5, 3), (1, 9)])
def test_downsample_is_bit_exact(rng, shape):
    img = rng.uniform(-1, 1, shape).astype(np.float32)
    for win in (2, 3, (2, 3), (3, 2)):
        wh, wv = (win, win) if np.isscalar(win) else win
        exp = so.average_pooling_downsample(img, wh, wv)
        for mk in (lambda x: x, dev):
            got = host(sv.averagePoolingDownsample(mk(img), win))
            assert got.shape == exp.shape
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))  # NaN (empty windows of the as-written offsets) included


@pytest.mark.parametrize("func", FLOAT_FUNCS + [MF.CENSUS])
@pytest.mark.parametrize("ddir", [R2L, L2R])
def test_guided_cv_matches_oracle(rng, func, ddir):
    """feature-volume entry point: integer estimates bit-exact, costs bit-exact (same float operations in the same order)"""
    H, Wl, Wr, r = 21, 45, 41, 2
    left = rng.uniform(-1, 1, (H, Wl)).astype(np.float32)
    right = rng.uniform(-1, 1, (H, Wr)).astype(np.float32)
    fl, fr = so.match_features(int(func), left, 2, 1), so.match_features(int(func), right, 2, 1)
    ws = Wr if ddir == R2L else Wl
    guide = rng.integers(-3, 9, (H // 2 + 1, ws // 2 + 1)).astype(np.int32)
    for radius in (0, 1, r, 4):
        etcv, edisp = so.guided_cv(int(func), fl, fr, guide, radius, int(ddir))
        for mk in (lambda x: x, dev):
            res = sv.computeGuidedCV(func, mk(fl), mk(fr), mk(guide), radius, ddir)
            assert np.array_equal(host(res.disp_estimate), edisp)
            got = host(res.truncated_cost_volume)
            assert got.shape == etcv.shape
            assert np.array_equal(got.view(np.uint32), etcv.view(np.uint32))


def test_guided_cv_argument_rules(rng):
    f = rng.uniform(-1, 1, (6, 9, 4)).astype(np.float32)
    g = np.zeros((3, 5), np.int32)
    assert sv.computeGuidedCV(MF.SAD, f, f[:5], g, 2).disp_estimate.size == 0  # row mismatch, hierarchical.h:90-92
    with pytest.raises(Exception):
        sv.computeGuidedCV(MF.SAD, f, f, np.zeros((1, 5), np.int32), 2)  # bilinear taps need a 2x2 guide
    with pytest.raises(Exception):
        sv.computeGuidedCV(MF.CENSUS, dev(f), dev(f), dev(g), 2)  # census features are uint32 words


@pytest.mark.parametrize("func", [MF.ZNCC, MF.NCC, MF.SAD, MF.ZSSD, MF.CENSUS])
@pytest.mark.parametrize("depth", [1, 2, 3])
def test_hierarchical_matches_oracle_literal(rng, func, depth):
    """with the per-voxel coarsest volume the whole chain follows the reference operation by operation: estimates and
    truncated costs must be bit-identical to the oracle, both directions, equal and per-level radii"""
    src, tgt, _ = parallax_pair(52, 75, 24, 14, 22, 3, 9, seed=100 + depth)
    x = dev(src)
    sv.set_option(x, "literal_cost_volumes", 1)
    try:
        for ddir in (R2L, L2R):
            for radii in (3, list(range(1, depth + 2))):
                etcv, edisp = so.hierarchical_truncated_cv(int(func), depth, tgt, src, radii, radii, 20, 2, int(ddir))
                res = sv.hiearchicalTruncatedCostVolume(func, depth, dev(tgt), x, radii, radii, 20, 2, ddir)
                assert np.array_equal(host(res.disp_estimate), edisp)
                assert np.array_equal(host(res.truncated_cost_volume).view(np.uint32), etcv.view(np.uint32))
    finally:
        sv.set_option(x, "literal_cost_volumes", 0)


@pytest.mark.parametrize("func", [MF.ZNCC, MF.SAD])
def test_hierarchical_default_kernels_agree_with_literal(rng, func):
    """default (register-blocked coarsest volume): costs within rounding; an estimate may only differ where the coarsest
    winner was a near tie, so at least 99.5 % of the pixels carry identical estimates and costs within 1e-4"""
    src, tgt, _ = parallax_pair(120, 200, 60, 30, 70, 4, 12, seed=7)
    for mk in (lambda a: a, dev):
        res = sv.hiearchicalTruncatedCostVolume(func, 2, mk(tgt), mk(src), 2, 2, 32, 2)
        etcv, edisp = so.hierarchical_truncated_cv(int(func), 2, tgt, src, 2, 2, 32, 2)
        same = host(res.disp_estimate) == edisp
        assert same.mean() >= 0.995
        assert np.max(np.abs(host(res.truncated_cost_volume)[same] - etcv[same])) <= 1e-4


def test_hierarchical_multichannel_and_shape_rules(rng):
    rgb_l = rng.uniform(-1, 1, (40, 56, 3)).astype(np.float32)
    rgb_r = rng.uniform(-1, 1, (40, 56, 3)).astype(np.float32)
    x = dev(rgb_l)
    sv.set_option(x, "literal_cost_volumes", 1)
    try:
        etcv, edisp = so.hierarchical_truncated_cv(so.ZSAD, 2, rgb_l, rgb_r, 1, 1, 12, 1)
        res = sv.hiearchicalTruncatedCostVolume(MF.ZSAD, 2, x, dev(rgb_r), 1, 1, 12, 1)
        assert np.array_equal(host(res.disp_estimate), edisp)
        assert np.array_equal(host(res.truncated_cost_volume).view(np.uint32), etcv.view(np.uint32))
    finally:
        sv.set_option(x, "literal_cost_volumes", 0)
    assert sv.hiearchicalTruncatedCostVolume(MF.SAD, 1, rgb_l, rgb_r[:39], 1, 1, 12).disp_estimate.size == 0  # row mismatch
    with pytest.raises(Exception):
        sv.hiearchicalTruncatedCostVolume(MF.SAD, 5, rgb_l, rgb_r, 1, 1, 12)  # 40 rows do not leave a 2x2 guide after 5 levels
    with pytest.raises(ValueError):
        sv.hiearchicalTruncatedCostVolume(MF.SAD, 2, rgb_l, rgb_r, [1, 1], [1, 1], 12)  # needs depth + 1 radii


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_reference_testMatchingFilter_on_gpu(depth):
    """testCorrelationHierarchical.cpp:27-170 (rows small_lvl{1,2,3}_zncc) against the HIP path with its default kernels."""
    sc = hierarchical_scene(depth, seed=20 + depth)
    res = sv.hiearchicalTruncatedCostVolume(MF.ZNCC, depth, dev(sc["target"]), dev(sc["source"]), 2, 2, 16, 2)
    rev = sv.hiearchicalTruncatedCostVolume(MF.ZNCC, depth, dev(sc["target"]), dev(sc["source"]), 2, 2, 16, 2, L2R)
    disp = host(res.disp_estimate)
    assert disp.shape == sc["source"].shape and tuple(res.truncated_cost_volume.shape) == sc["source"].shape + (5,)
    hierarchical_acceptance(sc, depth, disp, host(rev.disp_estimate))


@pytest.mark.parametrize("func", [MF.ZNCC, MF.NCC, MF.CC, MF.SSD, MF.ZSSD, MF.SAD, MF.ZSAD])
def test_guided_shared_features_same_bits_as_the_per_pixel_walk(func):
    """Option "guided_shared": 3 a wave of 64 pixels stages the processed target feature vectors its pixels look at in LDS and divides by
    the norms through their double reciprocals; 2: a block of 256 pixels, float divisions (round 4); 1 (default): the waves on large
    grids, the blocks on small ones; 0: every pixel processes every sample of every offset itself.  Same estimates and costs bit for
    bit: images wider than a block, a last block that is not full, search radii 1 - 3, windows 3x3 - 9x9 (the last too wide for the
    staged forms), both directions, a foreground square whose parallax differs from the background's by more than the staged span holds
    (waves that take two passes or walk, blocks that fall back), guides that point outside the image."""
    for (H, W, r, radius, bg, sq) in [(40, 700, 3, 2, 4, 12), (33, 530, 2, 3, 2, 300), (21, 300, 1, 1, 0, 40), (18, 600, 4, 2, 3, 9), (9, 257, 3, 3, 120, 5),
                                      (30, 333, 2, 2, 6, 30)]:
        src, tgt, _ = parallax_pair(H, W, max(H // 2, 2), H // 4, W // 3, bg, sq, seed=H + W)
        d_src, d_tgt = dev(src), dev(tgt)
        for ddir in (R2L, L2R):
            outs = []
            for shared in (0, 3, 2, 1):
                sv.set_test_option(d_src, "guided_shared", shared)
                try:
                    res = sv.hiearchicalTruncatedCostVolume(func, 2, d_tgt, d_src, r, r, 2 * max(bg, sq) + 8, radius, ddir)
                finally:
                    sv.set_test_option(d_src, "guided_shared", 1)
                outs.append((host(res.disp_estimate), host(res.truncated_cost_volume)))
            for o in outs[1:]:
                assert np.array_equal(outs[0][0], o[0]), (H, W, r, radius)
                assert np.array_equal(outs[0][1].view(np.uint32), o[1].view(np.uint32)), (H, W, r, radius)


@pytest.mark.parametrize("scale", [1e-30, 1e-19, 1.0, 1e18])
def test_guided_wave_divisions_by_shared_reciprocals_same_bits(scale):
    """div_by_shared (svh_hierarchical.hip): (float)((double)x * (1.0 / (double)y)) against x / y of the per-pixel walk, bit for bit, with
    images scaled so that the normalised samples and the norms run through large, small and denormal magnitudes."""
    H, W, r, radius = 24, 400, 2, 2
    src, tgt, _ = parallax_pair(H, W, H // 2, H // 4, W // 3, 3, 9, seed=77)
    rng = np.random.default_rng(5)
    src = (src * rng.uniform(0.5, 1.5, src.shape)).astype(np.float32) * np.float32(scale)
    tgt = (tgt * rng.uniform(0.5, 1.5, tgt.shape)).astype(np.float32) * np.float32(scale)
    src[3, 40:60] = 0.0  # windows with zero samples
    tgt[10:16, 100:140] = 0.0  # windows whose norm is 0: 0 / 0
    d_src, d_tgt = dev(src), dev(tgt)
    for func in (MF.ZNCC, MF.NCC):
        outs = []
        for shared in (0, 3):
            sv.set_test_option(d_src, "guided_shared", shared)
            try:
                res = sv.hiearchicalTruncatedCostVolume(func, 1, d_tgt, d_src, r, r, 24, radius)
            finally:
                sv.set_test_option(d_src, "guided_shared", 1)
            outs.append((host(res.disp_estimate), host(res.truncated_cost_volume)))
        assert np.array_equal(outs[0][0], outs[1][0])
        assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32))


@pytest.mark.parametrize("func", [MF.ZNCC, MF.NCC, MF.ZSSD, MF.SAD])
def test_guided_forms_same_bits_on_special_values(func):
    """NaN, +-inf, zeros, denormal-range and near-overflow samples in both images, images barely wider than a wave: the three staged
    forms of computeGuidedCV ("guided_shared" 3 / 2 / 1) return the bits of the per-pixel walk (0), NaN payloads included."""
    for trial, (H, W, r, radius, depth) in enumerate([(40, 500, 3, 2, 2), (16, 65, 2, 3, 1), (9, 200, 1, 1, 1), (50, 64, 2, 2, 2)]):
        src, tgt, _ = parallax_pair(H, W, max(H // 3, 1), H // 4, W // 3, 5, 21, seed=trial)
        src, tgt = src.copy(), tgt.copy()
        if H > 5:
            src[2, 10:30] = np.nan
            tgt[H // 2, 40:60] = np.inf
            src[H - 2, 5:9] = -np.inf
            tgt[1:4, 30:50] = 0.0
            src[0:3, 20:60] = 1e-38
            tgt[3:6, 10:40] = 3e38
        d_src, d_tgt = dev(src), dev(tgt)
        outs = []
        for form in (0, 3, 2, 1):
            sv.set_test_option(d_src, "guided_shared", form)
            try:
                res = sv.hiearchicalTruncatedCostVolume(func, depth, d_tgt, d_src, r, r, 64, radius)
            finally:
                sv.set_test_option(d_src, "guided_shared", 1)
            outs.append((host(res.disp_estimate), host(res.truncated_cost_volume).view(np.uint32)))
        for o in outs[1:]:
            assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]), (H, W, r, radius)


@pytest.mark.parametrize("func", [MF.ZNCC, MF.NCC, MF.CC, MF.SSD, MF.ZSSD, MF.SAD, MF.ZSAD])
def test_guided_wave_form_on_rgb_images_same_bits_as_the_per_pixel_walk(func):
    """RGB images take the wave form of computeGuidedCV (round 5: the channels are three times the samples of a window row, feature index
    = row, column, channel: unfold.h:180) at every image size; "guided_shared" 0: every pixel and offset on its own.  Same estimates and
    costs bit for bit -- windows 3x3 - 7x7 (9x9: too wide, both forms walk), radii 1 - 3, both directions, a parallax edge wider than the
    staged span, a last wave that is not full -- and the oracle's estimates."""
    for (H, W, r, radius, bg, sq) in [(24, 300, 3, 2, 4, 12), (20, 200, 2, 3, 2, 90), (12, 130, 1, 1, 0, 30), (10, 150, 4, 2, 3, 9)]:
        g_src, g_tgt, _ = parallax_pair(H, W, max(H // 2, 2), H // 4, W // 3, bg, sq, seed=H + W)
        rng = np.random.default_rng(H)
        gains = np.array([1.0, 0.7, 1.3], np.float32)
        src = (g_src[..., None] * gains + 0.05 * rng.uniform(-1, 1, g_src.shape + (3,))).astype(np.float32)
        tgt = (g_tgt[..., None] * gains + 0.05 * rng.uniform(-1, 1, g_tgt.shape + (3,))).astype(np.float32)
        d_src, d_tgt = dev(src), dev(tgt)
        for ddir in (R2L, L2R):
            outs = []
            for shared in (0, 1):
                sv.set_test_option(d_src, "guided_shared", shared)
                try:
                    res = sv.hiearchicalTruncatedCostVolume(func, 1, d_tgt, d_src, r, r, 2 * max(bg, sq) + 8, radius, ddir)
                finally:
                    sv.set_test_option(d_src, "guided_shared", 1)
                outs.append((host(res.disp_estimate), host(res.truncated_cost_volume)))
            assert np.array_equal(outs[0][0], outs[1][0]), (H, W, r, radius)
            assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32)), (H, W, r, radius)
    _, edisp = so.hierarchical_truncated_cv(int(func), 1, tgt, src, r, r, 2 * max(bg, sq) + 8, radius)
    sv.set_test_option(d_src, "guided_shared", 1)
    got = host(sv.hiearchicalTruncatedCostVolume(func, 1, d_tgt, d_src, r, r, 2 * max(bg, sq) + 8, radius, R2L).disp_estimate)
    assert (got == edisp).mean() > 0.995


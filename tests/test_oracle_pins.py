"""Pins the CPU oracle against the reference's own property tests (no GPU needed).

The reference has no golden vectors: every unit test draws inputs from std::random_device and
compares against a naive in-test formula or an analytic value.  Each test below restates one of
those tests on seeded inputs, so the oracle rows A1, A5-A8 and A12 are pinned the same way the
reference pins itself.  Census, Hamming, SGM, the argmin tie rule and the truncated volume have
no reference test ("parity unpinned", see oracle/stevi_oracle.c) and are covered by
tests/test_oracle_semantics.py against hand-computed cases only.
"""
import numpy as np
import pytest

import oracle as so
from helpers import hierarchical_acceptance, hierarchical_scene, isotropic_patch, rotated_quadric_patch, naive_window_cost

SHAPES = [(3, 3, 1), (1, 1, 5), (3, 3, 5), (5, 1, 5), (1, 5, 5), (5, 5, 5)]  # testCorrelationFilters.cpp:270-275
FUNCS = {"CC": so.CC, "NCC": so.NCC, "SSD": so.SSD, "SAD": so.SAD, "ZCC": so.ZCC, "ZNCC": so.ZNCC, "ZSSD": so.ZSSD,
         "ZSAD": so.ZSAD}


@pytest.mark.parametrize("h_r,v_r,disp_w", SHAPES)
@pytest.mark.parametrize("name", ["ZCC", "ZNCC", "NCC", "SSD", "ZSSD", "SAD", "ZSAD", "CC"])
def test_cost_volume_at_inside_pixel(rng, name, h_r, v_r, disp_w):
    """testCrossCorrelationFilter / testNCCFilter / testUnfoldNCCFilter (testCorrelationFilters.cpp:277-370,
    :462-500) and test2dMatching (testCorrelation2d.cpp:75-127): CV(v_r, h_r, d) equals the naive
    formula on windows right[0:2v+1, 0:2h+1] and left[0:2v+1, d:d+2h+1], tolerance 1e-3."""
    h, w = 2 * v_r + 1, 2 * h_r + disp_w + 1
    left = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    right = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    cv = so.unfold_cost_volume(FUNCS[name], left, right, h_r, v_r, disp_w)
    assert cv.shape == (h, w, disp_w)
    w1 = right[0:2 * v_r + 1, 0:2 * h_r + 1]
    for d in range(disp_w):
        w2 = left[0:2 * v_r + 1, d:d + 2 * h_r + 1]
        assert abs(naive_window_cost(name, w1, w2) - cv[v_r, h_r, d]) < 1e-3


@pytest.mark.parametrize("h_r,v_r", [(1, 1), (3, 3), (5, 1), (1, 5), (5, 5)])
def test_unfold_operator(rng, h_r, v_r):
    """testUnfoldOperator (testCorrelationFilters.cpp:384-445): unpadded unfold of a (2v+1)x(2h+1) image is
    1x1xF and holds the image's values; the auto-padded one is HxWxF with the same multiset at the centre."""
    h, w = 2 * v_r + 1, 2 * h_r + 1
    img = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    u = so.unfold(img, h_r, v_r, pad=(0, 0, 0, 0))
    assert u.shape == (1, 1, h * w)
    assert np.array_equal(np.sort(u[0, 0]), np.sort(img.ravel()))
    up = so.unfold(img, h_r, v_r)
    assert up.shape == (h, w, h * w)
    assert np.array_equal(np.sort(up[v_r, h_r]), np.sort(img.ravel()))
    # the channel order the hot path relies on (unfold.h:180, :283): c = (2h_r+1)*k + l
    assert np.array_equal(up[v_r, h_r], img.ravel())


@pytest.mark.parametrize("n", [5, 50, 500])
def test_channel_mean_and_sigma(rng, n):
    """testChannelMean / testChannelSigma (testCorrelationFilters.cpp:149-170, :232-262), tolerance 1e-3."""
    v = rng.uniform(-1, 1, (1, 1, n)).astype(np.float32)
    mean = v.astype(np.float64).mean()
    assert abs(mean - so.channels_mean(v)[0, 0]) < 1e-3
    sigma = np.sqrt(((v.astype(np.float64) - mean) ** 2).sum())
    assert abs(sigma - so.channels_zeromean_norm(v)[0, 0]) < 1e-3


@pytest.mark.parametrize("seed", range(8))
def test_1d_cost_parabola(seed):
    """test1dCostParabola (testCostRefinement.cpp:33-57): {a-b, 0, a+b} refines to -b/(2a)."""
    r = np.random.default_rng(seed)
    a = np.float32(r.uniform(-1, 1))
    b = np.float32(r.uniform(-2 * abs(a), 2 * abs(a)))
    tcv = np.array([[[a - b, 0, a + b]]], np.float32)
    raw = np.zeros((1, 1), np.int32)
    ref = so.refine_disp(tcv, raw, so.PARABOLA)
    expected = -b / (2 * a)
    assert np.isclose(ref[0, 0], expected, rtol=1e-5, atol=1e-6)  # QCOMPARE float fuzzy compare


def test_refine_rejects_bad_depth():
    """cost_based_refinement.h:141-143: depth must be 2r+1 with r >= 1."""
    assert so.refine_disp(np.zeros((2, 2, 1), np.float32), np.zeros((2, 2), np.int32)).size == 0
    assert so.refine_disp(np.zeros((2, 2, 4), np.float32), np.zeros((2, 2), np.int32)).size == 0


def test_planted_disparity_is_recovered():
    """Benchmark call chain (benchmarkCrossCorrelationAlgorithms.cpp:92-96) on a parallax pair: the planted
    background / square disparities win wherever the window is inside the image (the reference asserts this
    statistically in testCorrelationHierarchical.cpp:38-133)."""
    from helpers import parallax_pair
    src, tgt, gt = parallax_pair(40, 64, 12, 10, 20, 2, 7, seed=3)
    for f in (so.SAD, so.ZNCC, so.CENSUS):
        # 7x7: a 5x5 census writes no full 32-bit word at all (census.h:103-108), so its volume is all zero
        cv = so.unfold_cost_volume(f, tgt, src, 3, 3, 12)
        idx = so.extract_index(cv, so.func_strategy(f))
        disp = so.index_to_disp(idx)
        inner = np.zeros_like(gt, bool)
        inner[5:-5, 5:-20] = True
        # exclude the occlusion band around the square
        inner[6:26, 12:46] = False
        # census is weaker by construction: 32 of the 48 comparisons survive (census.h:103-108) and the target
        # word loses its low bits in the float round trip (cross_correlations.h:235-236)
        assert (disp[inner] == gt[inner]).mean() > (0.8 if f == so.CENSUS else 0.97)


# ---- 2-D disparity volumes: testCorrelation2d.cpp ------------------------------------------------------------------
SHAPES_2D = [(1, 1, 5, 5), (3, 3, 5, 5), (5, 1, 5, 3), (1, 5, 3, 5), (5, 5, 5, 5)]


@pytest.mark.parametrize("h_r,v_r,disp_w,disp_h", SHAPES_2D)
@pytest.mark.parametrize("name", ["NCC", "ZNCC", "SSD", "ZSSD", "SAD", "ZSAD"])
def test_2d_matching(rng, name, h_r, v_r, disp_w, disp_h):
    """test2dMatching (testCorrelation2d.cpp:75-127): CV(v_r, h_r, i, j) of unfoldBased2dDisparityCostVolume with
    searchOffset<2>(0, disp_h, 0, disp_w) equals the naive formula on right[0:2v+1, 0:2h+1] vs left[i:.., j:..], 1e-3."""
    h, w = 2 * v_r + disp_h + 1, 2 * h_r + disp_w + 1
    left = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    right = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    cv = so.unfold_cost_volume_2d(FUNCS[name], left, right, h_r, v_r, (0, disp_h), (0, disp_w))
    assert cv.shape == (h, w, disp_h + 1, disp_w + 1)
    w1 = right[0:2 * v_r + 1, 0:2 * h_r + 1]
    for i in range(disp_h):
        for j in range(disp_w):
            w2 = left[i:i + 2 * v_r + 1, j:j + 2 * h_r + 1]
            assert abs(naive_window_cost(name, w1, w2) - cv[v_r, h_r, i, j]) < 1e-3


@pytest.mark.parametrize("h_r,v_r", [(1, 1), (3, 3), (5, 1), (1, 5)])
@pytest.mark.parametrize("name", ["NCC", "ZNCC", "SSD", "ZSSD", "SAD", "ZSAD"])
def test_2d_disparity(rng, name, h_r, v_r):
    """test2dDisparity (testCorrelation2d.cpp:131-191): a planted shift is recovered by extractSelected2dIndex +
    selected2dIndexToDisp, for Score and Cost functions, at both ends of the search range."""
    disp_w = disp_h = 3
    h, w = 2 * v_r + disp_h + 1, 2 * h_r + disp_w + 1
    source = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    target0 = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    targetmax = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    for i in range(2 * v_r + 1):
        for j in range(2 * h_r + 1):
            target0[i, j] = source[i + 1, j + 1]
            targetmax[i + disp_h, j + disp_w] = source[i + 1, j + 1]
    rng0, rng1 = (-1, disp_h - 1), (-1, disp_w - 1)
    strat = so.func_strategy(FUNCS[name])
    for target, expected in ((target0, (-1, -1)), (targetmax, (disp_h - 1, disp_w - 1))):
        cv = so.unfold_cost_volume_2d(FUNCS[name], target, source, h_r, v_r, rng0, rng1)
        disp = so.index_2d_to_disp(so.extract_index_2d(cv, strat), rng0[0], rng1[0])
        assert tuple(disp[v_r + 1, h_r + 1]) == expected


# ---- 2-D cost-based refinement: testCostRefinement.cpp -------------------------------------------------------------
@pytest.mark.parametrize("trial", range(8))
def test_2d_cost_isotropic_parabola(rng, trial):
    """test2dCostIsotropicParabola (testCostRefinement.cpp:60-115): an isotropic paraboloid's minimum is recovered by
    refineDisp2dCostInterpolation<Parabola, Isotropic>; QCOMPARE on floats is a 1e-5 relative comparison."""
    s = rng.uniform(-1, 1)
    s = 0.5 if abs(s) < 1e-4 else s
    ex, ey = rng.uniform(-1, 1, 2)
    ref = so.refine_disp_2d(isotropic_patch(s, ex, ey), np.zeros((1, 1, 2), np.int32), so.PARABOLA, so.ISOTROPIC)
    assert ref[0, 0, 0] == pytest.approx(ex, rel=1e-5, abs=1e-6) and ref[0, 0, 1] == pytest.approx(ey, rel=1e-5, abs=1e-6)


@pytest.mark.parametrize("trial", range(8))
def test_2d_cost_anisotropic_parabola(rng, trial):
    """test2dCostAnisotropicParabola (testCostRefinement.cpp:117-173): slightly rotated anisotropic quadric on a 5x5
    truncated volume, recovered to 1e-6 by the Anisotropic hypothesis."""
    s = rng.uniform(-1, 1)
    s = np.copysign(max(abs(s), 0.5), s)
    d1, d2 = s * rng.uniform(0.7, 1, 2)
    b = rng.uniform(-0.5, 0.5, 2)
    tcv = rotated_quadric_patch(d1, d2, 0.1 * rng.uniform(-1, 1), b, 5)
    ref = so.refine_disp_2d(tcv, np.zeros((1, 1, 2), np.int32), so.PARABOLA, so.ANISOTROPIC)
    assert abs(ref[0, 0, 0] - b[0]) < 2e-6 and abs(ref[0, 0, 1] - b[1]) < 2e-6  # float32 patch built here, 1e-6 in the reference


@pytest.mark.parametrize("trial", range(8))
def test_2d_cost_full_matching_parabola(rng, trial):
    """test2dCostFullMatchingParabola (testCostRefinement.cpp:176-221): arbitrarily rotated quadric, recovered by the
    3x3 patch fit refineDisp2dCostPatchInterpolation<Parabola>."""
    s = rng.uniform(-1, 1)
    s = np.copysign(max(abs(s), 0.5), s)
    d1, d2 = s * rng.uniform(0.7, 1, 2)
    b = rng.uniform(-0.5, 0.5, 2)
    tcv = rotated_quadric_patch(d1, d2, 3 * rng.uniform(-1, 1), b, 3)
    ref = so.refine_disp_2d_patch(tcv, np.zeros((1, 1, 2), np.int32), so.PARABOLA)
    assert ref[0, 0, 0] == pytest.approx(b[0], rel=1e-5, abs=2e-6) and ref[0, 0, 1] == pytest.approx(b[1], rel=1e-5, abs=2e-6)


def test_2d_refinement_rules():
    """cost_based_refinement.h:180-182 (empty result for bad depths), :362-366 (deltas zeroed beyond 1 or NaN), :184-203 (the
    score probe at the centre pixel only matters for the anisotropic extremum search)."""
    raw = np.array([[[3, -2]]], np.int32)
    assert so.refine_disp_2d(np.zeros((1, 1, 1, 3), np.float32), raw).size == 0
    assert so.refine_disp_2d(np.zeros((1, 1, 4, 3), np.float32), raw).size == 0
    flat = np.ones((1, 1, 3, 3), np.float32)  # 0/0 -> NaN -> zeroed
    assert np.array_equal(so.refine_disp_2d(flat, raw, so.PARABOLA), raw.astype(np.float32))
    far = isotropic_patch(1.0, 1.5, 0.0)  # minimum 1.5 away along axis 0 -> both deltas dropped
    assert np.array_equal(so.refine_disp_2d(far, raw, so.PARABOLA), raw.astype(np.float32))
    with pytest.raises(ValueError):
        so.refine_disp_2d_patch(flat, raw, so.EQUIANGULAR)
    # a score volume (maximum at the centre) is refined towards the same location as the mirrored cost volume
    cost = isotropic_patch(0.8, 0.2, -0.3, 5)
    a = so.refine_disp_2d(cost, raw, so.PARABOLA, so.ANISOTROPIC)
    b = so.refine_disp_2d(-cost, raw, so.PARABOLA, so.ANISOTROPIC)
    assert np.allclose(a, b, atol=1e-6) and np.allclose(a[0, 0], [3.2, -2.3], atol=1e-5)


# ---- hierarchical matching: testCorrelationHierarchical.cpp ---------------------------------------------------------
@pytest.mark.parametrize("depth", [1, 2, 3])
def test_hierarchical_matching_filter(depth):
    """testMatchingFilter rows small_lvl{1,2,3}_zncc (testCorrelationHierarchical.cpp:158-170): 48x64 pair, 5x5 windows,
    disp_w 16, upscale radius 2, both directions; shapes, range bound and >= 95 % exact matches inside the square."""
    sc = hierarchical_scene(depth, seed=10 + depth)
    tcv, disp = so.hierarchical_truncated_cv(so.ZNCC, depth, sc["target"], sc["source"], 2, 2, 16, 2)
    _, disp_rev = so.hierarchical_truncated_cv(so.ZNCC, depth, sc["target"], sc["source"], 2, 2, 16, 2, so.LEFT_TO_RIGHT)
    assert disp.shape == sc["source"].shape and tcv.shape == sc["source"].shape + (5,)
    hierarchical_acceptance(sc, depth, disp, disp_rev)


def test_average_pooling_downsample_rules(rng):
    """interpolation/downsampling.h:67-114: output size ceil(n / window); even sizes are plain 2x2 means, an odd last row /
    column averages the valid samples only (count, :101-106)."""
    img = rng.uniform(-1, 1, (7, 10)).astype(np.float32)
    out = so.average_pooling_downsample(img, 2)
    assert out.shape == (4, 5)
    assert np.allclose(out[:3], img[:6].reshape(3, 2, 5, 2).mean(axis=(1, 3)), atol=1e-6)
    assert np.allclose(out[3], img[6].reshape(5, 2).mean(axis=1), atol=1e-6)
    rgb = rng.uniform(-1, 1, (6, 5, 3)).astype(np.float32)
    out3 = so.average_pooling_downsample(rgb, 2)
    assert out3.shape == (3, 3, 3) and np.allclose(out3[:, 2], rgb[:, 4].reshape(3, 2, 3).mean(axis=1), atol=1e-6)


def test_guided_cv_recentres_on_the_winner(rng):
    """computeGuidedCV (hierarchical.h:74-229): whatever the guide, the truncated volume ends up centred on the selected
    disparity: tcv(i,j,dd) = cmp(src(i,j), tgt(i, j + d_r + dirSign (dd - r))); checked against a direct evaluation."""
    H, W, r = 9, 23, 2
    left = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    right = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    guide = rng.integers(0, 4, (5, 12)).astype(np.int32)
    for func, ddir in ((so.SAD, so.RIGHT_TO_LEFT), (so.ZNCC, so.LEFT_TO_RIGHT), (so.CENSUS, so.RIGHT_TO_LEFT)):
        fl, fr = so.match_features(func, left, 3, 3), so.match_features(func, right, 3, 3)
        tcv, disp = so.guided_cv(func, fl, fr, guide, r, ddir)
        sign = 1 if ddir == so.RIGHT_TO_LEFT else -1
        src, tgt = (fr, fl) if ddir == so.RIGHT_TO_LEFT else (fl, fr)
        cost = so.func_strategy(func) == so.COST
        for i in range(H):
            for j in range(W):
                d_r = sign * disp[i, j]
                for dd in range(2 * r + 1):
                    col = j + d_r + sign * (dd - r)
                    t = tgt[i, col] if 0 <= col < W else np.zeros_like(tgt[0, 0])
                    if func == so.CENSUS:
                        exp = sum(bin(int(a) ^ int(b)).count("1") for a, b in zip(src[i, j], t))
                    elif func == so.SAD:
                        exp = np.abs(src[i, j].astype(np.float64) - t).sum()
                    else:
                        exp = float(src[i, j].astype(np.float64) @ t)
                    assert abs(tcv[i, j, dd] - exp) < 1e-4
    # with a constant guide g the search window is centred on d0 = dirSign * 2g everywhere: the estimate is the first strict
    # extremum of cmp(src(i,j), tgt(i, j + d0 + delta)), delta = -r..r (:157-190)
    fl, fr = so.match_features(so.SAD, left, 1, 1), so.match_features(so.SAD, right, 1, 1)
    tcv, disp = so.guided_cv(so.SAD, fl, fr, np.full((4, 4), 3, np.int32), r, so.RIGHT_TO_LEFT)
    for i in range(H):
        for j in range(W):
            best, arg = np.inf, 6
            for delta in range(-r, r + 1):
                col = j + 6 + delta
                t = fl[i, col] if 0 <= col < W else np.zeros_like(fl[0, 0])
                c = np.float32(np.abs(fr[i, j] - t).astype(np.float32).sum(dtype=np.float32))
                if c < best - 1e-5:
                    best, arg = c, 6 + delta
            assert disp[i, j] == arg


# ---- on-demand cost volume tests that pin the dense feature-volume paths: testOnDemandCostVolume.cpp ------------------
@pytest.mark.parametrize("name", ["NCC", "ZNCC", "SSD", "SAD"])
def test_feature_volume_2d_equals_per_offset_1d(rng, name):
    """testOnDemandCVOpticalFlow (testOnDemandCostVolume.cpp:117-170) compares every in-range entry of
    featureVolume2CostVolume(searchOffset<2>) with the pairwise feature comparison; restated here against the 1-D volume of
    row-shifted targets, which the stereo variant of the same test (:66-115) pins entry by entry."""
    H, W, F, r = 7, 9, 6, 2
    src = rng.uniform(-10, 10, (H, W, F)).astype(np.float32)
    tgt = rng.uniform(-10, 10, (H, W, F)).astype(np.float32)
    cv = so.feature_cost_volume_2d(FUNCS[name], tgt, src, (-r, r), (-r, r))
    assert cv.shape == (H, W, 2 * r + 1, 2 * r + 1)
    for dh in range(-r, r + 1):
        shifted = np.zeros_like(tgt)
        lo, hi = max(0, -dh), min(H, H - dh)
        shifted[lo:hi] = tgt[lo + dh:hi + dh]
        one = so.feature_cost_volume(FUNCS[name], shifted, src, 2 * r + 1, so.RIGHT_TO_LEFT, -r)
        rows = slice(lo, hi)  # rows whose shifted target exists; elsewhere the target vector is zero, not a processed zero row
        assert np.allclose(cv[rows, :, dh + r, :], one[rows], rtol=1e-5, atol=1e-5)


# ---- UnFoldCompressor: testCorrelationFilters.cpp:502-577 -----------------------------------------------------------
@pytest.mark.parametrize("h_radius,v_radius", [(1, 1), (3, 3), (5, 1), (1, 5), (5, 5)])
def test_unfold_compressor(rng, h_radius, v_radius):
    """testUnfoldCompressor: a random partition of the window into h superpixels; with PaddingMargins(0) the single output
    pixel holds the superpixel means (compared as sorted multisets at 1e-4, like the reference)."""
    h, w = 2 * v_radius + 1, 2 * h_radius + 1
    img = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    labels = (np.arange(h * w) % h) + 1
    rng.shuffle(labels)
    mask = labels.reshape(h, w).astype(np.int32)
    out = so.unfold_compressed(img, mask, (0, 0, 0, 0))
    assert out.shape == (1, 1, h)
    check = np.sort([img[mask == f].astype(np.float64).mean() for f in range(1, h + 1)])
    assert np.max(np.abs(np.sort(out[0, 0]) - check)) < 1e-4


def test_unfold_compressor_layout_rules(rng):
    """unfold.h:47-121, :346-471: features in increasing label order whatever the labels are; labels <= 0 are ignored; auto padding
    keeps the image size; multi-channel images give the channel-major index in_c * nFeatures + f; zero outside the image."""
    img = rng.uniform(-1, 1, (6, 7)).astype(np.float32)
    mask = np.array([[0, 7, 7], [3, 3, -2], [0, 40, 0]], np.int32)  # labels 3, 7, 40 -> features 0, 1, 2
    out = so.unfold_compressed(img, mask)
    assert out.shape == (6, 7, 3)
    pad = np.pad(img, 1)
    i, j = 2, 3
    win = pad[i:i + 3, j:j + 3]
    assert np.allclose(out[i, j], [(win[1, 0] + win[1, 1]) / 2, (win[0, 1] + win[0, 2]) / 2, win[2, 1]], atol=1e-6)
    assert np.allclose(out[0, 0, 1], 0.0)  # the row above the image
    rgb = rng.uniform(-1, 1, (5, 4, 2)).astype(np.float32)
    out3 = so.unfold_compressed(rgb, mask)
    assert out3.shape == (5, 4, 6)
    assert np.array_equal(out3[:, :, 0:3], so.unfold_compressed(rgb[:, :, 0], mask)) and np.array_equal(out3[:, :, 3:6], so.unfold_compressed(rgb[:, :, 1], mask))

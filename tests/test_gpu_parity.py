"""GPU parity tests: every entry point of the C ABI (through libstevi_amd's Python mirror) against the CPU
oracle on the same seeded inputs.  Integer / index / bit-pattern outputs must match exactly; float cost
volumes and refined disparities within 1e-4 (BASELINE.json north_star) with identical NaN masks.

Host arrays (numpy) exercise the copy-in / copy-out path, torch CUDA tensors the in-place device path.
"""
import numpy as np
import pytest

import oracle as so
from helpers import parallax_pair

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import libstevi_amd as sv  # noqa: E402
from libstevi_amd import matchingFunctions as MF  # noqa: E402

TOL = 1e-4  # north_star: float NCC costs and sub-pixel refinement within 1e-4
DEV = torch.device("cuda:0")


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def host(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else x


def assert_close(got, exp, tol=TOL):
    got, exp = host(got), host(exp)
    assert got.shape == exp.shape
    assert np.array_equal(np.isnan(got), np.isnan(exp)), "NaN masks differ"
    both_inf = np.isinf(got) & np.isinf(exp) & (np.sign(got) == np.sign(exp))
    ok = ~np.isnan(exp) & ~both_inf
    err = np.abs(got[ok].astype(np.float64) - exp[ok].astype(np.float64))
    lim = tol * np.maximum(1.0, np.abs(exp[ok].astype(np.float64)))
    assert np.all(err <= lim), f"max err {err.max()} (tol {tol})"


def assert_bits(got, exp):
    got, exp = host(got), host(exp)
    assert got.shape == exp.shape and got.dtype == exp.dtype
    a = got.view(np.uint32) if got.dtype == np.float32 else got
    b = exp.view(np.uint32) if exp.dtype == np.float32 else exp
    nbad = int((a != b).sum())
    assert nbad == 0, f"{nbad} of {a.size} elements differ"


# ------------------------------------------------------------------------------------------------ A1-A3
@pytest.mark.parametrize("h_r,v_r", [(0, 0), (1, 1), (2, 3), (4, 4), (5, 2)])
@pytest.mark.parametrize("channels", [1, 3])
def test_unfold(rng, h_r, v_r, channels):
    img = rng.uniform(-1, 1, (9, 13, channels)).astype(np.float32)
    if channels == 1:
        img = img[:, :, 0]
    assert_bits(sv.unfold(h_r, v_r, img), so.unfold(img, h_r, v_r))
    assert_bits(sv.unfold(h_r, v_r, dev(img), sv.PaddingMargins(1, 0, 2, 1)), so.unfold(img, h_r, v_r, (1, 0, 2, 1)))
    if 2 * v_r + 1 <= 9 and 2 * h_r + 1 <= 13:
        assert_bits(sv.unfold(h_r, v_r, img, sv.PaddingMargins(0)), so.unfold(img, h_r, v_r, (0, 0, 0, 0)))


@pytest.mark.parametrize("h_r,v_r", [(1, 1), (2, 2), (3, 3), (4, 4), (5, 2), (4, 3), (7, 7), (2, 3), (3, 2), (2, 4), (4, 2), (2, 5), (3, 4), (3, 5), (5, 3), (4, 5), (5, 4)])
def test_census_transform(rng, h_r, v_r):
    img = rng.uniform(-1, 1, (17, 23)).astype(np.float32)
    img[3, 5] = np.nan
    exp = so.census_transform(img, h_r, v_r)
    assert_bits(sv.censusTransform2D(img, h_r, v_r), exp)
    assert_bits(sv.censusTransform2D(dev(img), h_r, v_r), exp)
    # several tiles of the register-blocked grey kernel (squares 7x7 - 11x11, rectangles with half-widths 2 .. 5), all samples finite and not
    big = rng.uniform(-1, 1, (37, 700)).astype(np.float32)
    assert_bits(sv.censusTransform2D(dev(big), h_r, v_r), so.census_transform(big, h_r, v_r))
    big[20, 300] = np.inf
    big[5, 650] = np.nan
    assert_bits(sv.censusTransform2D(dev(big), h_r, v_r), so.census_transform(big, h_r, v_r))
    # colour image and explicit padding
    rgb = rng.uniform(-1, 1, (19, 21, 3)).astype(np.float32)
    assert_bits(sv.censusTransform2D(rgb, h_r, v_r, sv.PaddingMargins(2, 1, 0, 3)), so.census_transform(rgb, h_r, v_r, (2, 1, 0, 3)))
    # feature-volume entry point
    feat = so.unfold(img, h_r, v_r)
    assert_bits(sv.censusFeatures(feat), so.census_features(feat))


@pytest.mark.parametrize("h_r", [3, 4, 5])
def test_census_transform_special_values(rng, h_r):
    """The grey-image census kernel takes a bit from the sign of sample - ref on tiles whose samples are all finite and from a
    compare on tiles that hold a NaN or an infinity: signed zeros, subnormals, equal neighbours, infinities of both signs next to
    each other and NaNs, in an image wide and tall enough that both kinds of tile occur."""
    H, W = 37, 700
    img = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    img[:, 50:120] = np.round(img[:, 50:120] * 2) / 2                      # many equal neighbours, and zeros of both signs
    img[:, 60:100:3] *= np.float32(-0.0) + np.float32(1.0)
    img[5:9, 70:90] = np.float32(-0.0)
    img[9:12, 70:90] = np.float32(0.0)
    img[:, 130:170] = (rng.integers(-3, 4, (H, 40)) * np.float32(1e-45)).astype(np.float32)  # subnormals around zero
    img[20:30, 300:330] = np.inf                                           # a block of +inf: inf against inf
    img[22:26, 310:315] = -np.inf
    img[31, 333] = np.nan
    img[2, 690] = -np.nan
    img[33:, 400:] = rng.uniform(-1e-38, 1e-38, img[33:, 400:].shape).astype(np.float32)     # differences that are subnormal
    exp = so.census_transform(img, h_r, h_r)
    assert_bits(sv.censusTransform2D(dev(img), h_r, h_r), exp)
    assert_bits(sv.censusTransform2D(img, h_r, h_r), exp)
    # the fused pipeline goes through the same kernel (both images in one launch)
    other = np.roll(img, 7, axis=1)
    cv = so.unfold_cost_volume(so.CENSUS, img, other, h_r, h_r, 32)
    res = sv.stereoMatch(MF.CENSUS, dev(img), dev(other), h_r, h_r, 32, sgmDirections=0, want_cv=True)
    assert_bits(res["cv"], cv)


@pytest.mark.parametrize("h_r,channels", [(2, 3), (3, 3), (4, 3), (6, 1), (7, 1)])
def test_census_transform_fixed_windows(rng, h_r, channels):
    """Colour images (5x5, 7x7, 9x9 RGB) and grey windows 13 / 15 wide take census_fixed_kernel (four output rows per block from one staged
    tile, a compare per bit): special values, explicit padding, images narrower and shorter than a block, and the fused pipeline with both
    images in one launch -- bit for bit against the oracle."""
    H, W = 23, 530
    shape = (H, W, channels) if channels > 1 else (H, W)
    img = rng.uniform(-1, 1, shape).astype(np.float32)
    img[:, 50:120] = np.round(img[:, 50:120] * 2) / 2
    img[5:9, 70:90] = np.float32(-0.0)
    img[9:12, 70:90] = np.float32(0.0)
    img[12:20, 300:330] = np.inf
    img[14:16, 310:315] = -np.inf
    img[3, 333] = np.nan
    img[2, 520] = -np.nan
    exp = so.census_transform(img, h_r, h_r)
    assert_bits(sv.censusTransform2D(dev(img), h_r, h_r), exp)
    assert_bits(sv.censusTransform2D(img, h_r, h_r), exp)
    assert_bits(sv.censusTransform2D(dev(img), h_r, h_r, sv.PaddingMargins(2, 1, 0, 3)), so.census_transform(img, h_r, h_r, (2, 1, 0, 3)))
    small = np.ascontiguousarray(img[:3, :7])
    assert_bits(sv.censusTransform2D(dev(small), h_r, h_r), so.census_transform(small, h_r, h_r))
    other = np.roll(img, 7, axis=1)[:, :500].copy()  # another width: the launch's grid covers the larger image
    cv = so.unfold_cost_volume(so.CENSUS, img, other, h_r, h_r, 32)
    res = sv.stereoMatch(MF.CENSUS, dev(img), dev(other), h_r, h_r, 32, sgmDirections=0, want_cv=True)
    assert_bits(res["cv"], cv)
    vol = so.sgm(cv, 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0)
    assert np.array_equal(host(sv.stereoMatch(MF.CENSUS, dev(img), dev(other), h_r, h_r, 32, sgmDirections=8)["disp"]),
                          so.index_to_disp(so.extract_index(vol, so.COST)))


def test_census_single_channel_is_empty():
    assert sv.censusFeatures(np.zeros((3, 3, 1), np.float32)).size == 0
    assert sv.censusTransform2D(np.zeros((3, 3), np.float32), 0, 0).size == 0


# ------------------------------------------------------------------------------------------------ A4-A8
ALL_FUNCS = [MF.CC, MF.NCC, MF.SSD, MF.SAD, MF.ZCC, MF.ZNCC, MF.ZSSD, MF.ZSAD, MF.HAMMING, MF.CENSUS]


@pytest.mark.parametrize("func", ALL_FUNCS)
@pytest.mark.parametrize("ddir", [sv.dispDirection.RightToLeft, sv.dispDirection.LeftToRight])
def test_unfold_cost_volume(rng, func, ddir):
    left = rng.uniform(-1, 1, (12, 21)).astype(np.float32)
    right = rng.uniform(-1, 1, (12, 17)).astype(np.float32)  # widths may differ (only rows are checked)
    for (h_r, v_r, D, lower) in [(4, 4, 9, 0), (1, 2, 30, 0), (3, 3, 7, -3)]:
        exp = so.unfold_cost_volume(int(func), left, right, h_r, v_r, D, int(ddir), lower)
        rng_arg = D if lower == 0 else sv.searchOffset1(lower, lower + D - 1)
        got = sv.unfoldBasedCostVolume(func, left, right, h_r, v_r, rng_arg, ddir)
        if int(func) in (MF.HAMMING, MF.CENSUS):
            assert_bits(got, exp)
        else:
            assert_close(got, exp)
        got_d = sv.unfoldBasedCostVolume(func, dev(left), dev(right), h_r, v_r, rng_arg, ddir)
        assert_bits(got_d, host(got))  # host and device paths run the same kernels


def test_cost_volume_edge_cases(rng):
    a = rng.uniform(-1, 1, (6, 8)).astype(np.float32)
    b = rng.uniform(-1, 1, (7, 8)).astype(np.float32)
    assert sv.unfoldBasedCostVolume(MF.SAD, a, b, 1, 1, 4).size == 0  # row mismatch -> empty (cross_correlations.h:751)
    const = np.full((6, 8), 0.5, np.float32)
    # zero norm -> NaN (rule E4), constant census -> all zero
    exp = so.unfold_cost_volume(so.ZNCC, const, const, 1, 1, 3)
    assert_close(sv.unfoldBasedCostVolume(MF.ZNCC, const, const, 1, 1, 3), exp)
    assert_bits(sv.unfoldBasedCostVolume(MF.CENSUS, const, const, 4, 4, 3), so.unfold_cost_volume(so.CENSUS, const, const, 4, 4, 3))
    # all-ones census words (monotone ramp) exercise the 2^32 saturation of rule E2
    ramp = (np.arange(20, 0, -1, dtype=np.float32)[None, :] + np.arange(14, 0, -1, dtype=np.float32)[:, None])
    assert_bits(sv.unfoldBasedCostVolume(MF.CENSUS, ramp, ramp, 4, 4, 6), so.unfold_cost_volume(so.CENSUS, ramp, ramp, 4, 4, 6))
    # colour images
    l3 = rng.uniform(-1, 1, (7, 9, 3)).astype(np.float32)
    r3 = rng.uniform(-1, 1, (7, 9, 3)).astype(np.float32)
    assert_close(sv.unfoldBasedCostVolume(MF.SSD, l3, r3, 1, 1, 4), so.unfold_cost_volume(so.SSD, l3, r3, 1, 1, 4))
    assert_bits(sv.unfoldBasedCostVolume(MF.CENSUS, l3, r3, 2, 2, 4), so.unfold_cost_volume(so.CENSUS, l3, r3, 2, 2, 4))


@pytest.mark.parametrize("func", [MF.NCC, MF.ZSAD, MF.SSD, MF.CENSUS])
def test_feature_cost_volume(rng, func):
    """featureVolume2CostVolume on caller-provided feature volumes (examples/stereo_refine_test/main.cpp:367-377)."""
    fl = rng.uniform(-1, 1, (7, 12, 40)).astype(np.float32)
    fr = rng.uniform(-1, 1, (7, 10, 40)).astype(np.float32)
    exp = so.feature_cost_volume(int(func), fl, fr, 6)
    got = sv.featureVolume2CostVolume(func, fl, fr, 6)
    assert_bits(got, exp) if func == MF.CENSUS else assert_close(got, exp)
    # non-dense device view (feature axis first in memory) goes through the relayout kernel
    t = dev(np.ascontiguousarray(fl.transpose(2, 0, 1))).permute(1, 2, 0)
    got2 = sv.featureVolume2CostVolume(func, t, dev(fr), 6)
    assert_bits(got2, host(got))


# ------------------------------------------------------------------------------------------------ A9
SGM_PARAMS = [(0.001, 0.01, 100.0), (2.0, 7.0, 3.0), (5.0, 1.0, 0.5), (0.0, 0.0, 0.0)]


@pytest.mark.parametrize("strategy", [so.COST, so.SCORE])
@pytest.mark.parametrize("D", [1, 5, 64, 70, 130, 160, 192, 256, 300, 320, 384, 430, 448])
def test_sgm_bit_exact(rng, strategy, D):
    H, W = 9, 12
    for integer in (True, False):
        cv = (rng.integers(0, 65, (H, W, D)) if integer else rng.uniform(-1, 1, (H, W, D))).astype(np.float32)
        for n_dir, margins, (P1, P2, Pout) in [(8, (0, 0, 0, 0), SGM_PARAMS[0]), (8, (1, 2, 3, 1), SGM_PARAMS[1]),
                                               (4, (0, 0, 0, 0), SGM_PARAMS[2]), (8, (2, 0, 0, 1), SGM_PARAMS[3])]:
            exp = so.sgm(cv, n_dir, strategy, P1, P2, margins, Pout)
            got = sv.sgmCostVolume(n_dir, strategy, cv, P1, P2, sv.Margins(*margins), Pout)
            assert_bits(got, exp)
    got_d = sv.sgmCostVolume(8, strategy, dev(cv), 0.5, 1.5, None, 10.0)
    assert_bits(got_d, so.sgm(cv, 8, strategy, 0.5, 1.5, (0, 0, 0, 0), 10.0))


@pytest.mark.parametrize("strategy", [so.COST, so.SCORE])
@pytest.mark.parametrize("shape", [(21, 33, 40), (33, 21, 40), (1, 17, 8), (17, 1, 8), (2, 2, 3)])
def test_sgm_shapes_and_nonfinite(rng, strategy, shape):
    cv = rng.uniform(0, 4, shape).astype(np.float32)
    exp = so.sgm(cv, 8, strategy, 0.3, 0.9, (0, 0, 0, 0), 7.0)
    assert_bits(sv.sgmCostVolume(8, strategy, cv, 0.3, 0.9, None, 7.0), exp)
    if cv.size > 40:
        cv[0, 0, 0] = np.nan
        cv[-1, -1, :] = np.inf
        cv[shape[0] // 2, shape[1] // 2, 1] = -np.inf
        cv[0, shape[1] // 2, :] = np.nan
        exp = so.sgm(cv, 8, strategy, 0.3, 0.9, (0, 0, 0, 0), 7.0)
        got = host(sv.sgmCostVolume(8, strategy, cv, 0.3, 0.9, None, 7.0))
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        ok = ~np.isnan(exp)
        assert np.array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32))


def test_sgm_margins_larger_than_image(rng):
    cv = rng.uniform(0, 1, (5, 6, 4)).astype(np.float32)
    for strategy in (so.COST, so.SCORE):
        assert_bits(sv.sgmCostVolume(8, strategy, cv, 1, 2, sv.Margins(4, 4), 5.0), so.sgm(cv, 8, strategy, 1, 2, (4, 4, 4, 4), 5.0))


@pytest.mark.parametrize("D", [1025, 1500, 2048])
def test_sgm_more_than_1024_disparities(rng, D):
    """32 disparities per lane: up to 2048 per call (the reference has no limit, sgm.h:360-404; beyond 2048: SVH_ERR_UNSUPPORTED)."""
    cv = rng.uniform(-2, 2, (5, 9, D)).astype(np.float32)
    cv[2, 3, 7] = np.inf
    cv[1, 1, D - 1] = np.nan
    for strategy in (so.COST, so.SCORE):
        for n_dir, margins in ((8, (0, 0, 0, 0)), (4, (1, 0, 2, 1))):
            exp = so.sgm(cv, n_dir, strategy, 0.3, 0.9, margins, 5.0)
            got = host(sv.sgmCostVolume(n_dir, strategy, dev(cv), 0.3, 0.9, sv.Margins(*margins), 5.0))
            assert np.array_equal(np.isnan(got), np.isnan(exp))
            ok = ~np.isnan(exp)
            assert np.array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32))
    ci = rng.integers(0, 60, (4, 7, D)).astype(np.float32)  # integer costs: the probe declines D > 1024, the general kernels run
    assert_bits(sv.sgmCostVolume(8, so.COST, ci, 0.3, 0.9, None, 100.0), so.sgm(ci, 8, so.COST, 0.3, 0.9, (0, 0, 0, 0), 100.0))
    from libstevi_amd._capi import SvhError, ERR_UNSUPPORTED
    with pytest.raises(SvhError) as e:
        sv.sgmCostVolume(8, so.SCORE, np.zeros((2, 2, 2049), np.float32), 1, 2)
    assert e.value.status == ERR_UNSUPPORTED


def test_sgm_rejects_16_directions(rng):
    from libstevi_amd._capi import SvhError, ERR_UNSUPPORTED
    with pytest.raises(SvhError) as e:
        sv.sgmCostVolume(16, so.COST, np.zeros((4, 4, 4), np.float32), 1, 2)
    assert e.value.status == ERR_UNSUPPORTED


# ------------------------------------------------------------------------------------------------ A10-A12
@pytest.mark.parametrize("D", [1, 2, 3, 4, 5, 8, 60, 63, 64, 65, 128, 132, 200, 255, 256, 257, 260, 289, 297, 320, 511, 512, 513, 1000, 1023, 1024, 1025, 1500])
def test_extract_index(rng, D):
    """(rows of 4 ... 256 costs, a multiple of four, take the kernel that packs several pixels into a wave: extract_index_packed_kernel; other
    rows of up to 1 024 costs a wave per pixel with up to sixteen costs per lane: extract_index_wide_kernel, option "extract_index_wide"; the
    rest -- and that option at 0 -- the wave-per-pixel kernel of round 1)"""
    cv = rng.integers(0, 6, (11, 14, D)).astype(np.float32)  # few levels -> many ties
    cv[0, 0, 0] = np.nan
    if D > 2:
        cv[1, 1, 1] = np.nan
        cv[2, 2, :] = np.nan
        cv[3, 3, D - 1] = -0.0
        cv[3, 3, 0] = 0.0
        cv[4, 4, :] = np.inf            # every value the same infinity: the last index
        cv[4, 5, :] = -np.inf
        cv[5, 5, 1:] = np.nan           # only index 0 compares
        cv[5, 6, : D - 1] = np.nan      # only the last index compares
        cv[6, 6, D // 2] = -np.inf
        cv[6, 7, D // 2] = np.inf
        cv[7, 7, 0] = np.nan            # a NaN at index 0 beside ...
        cv[7, 7, D - 1] = -np.inf       # ... an unbeatable value: index 0 stays (E5)
        cv[10, 13, :] = rng.uniform(-1, 1, D)  # the last pixel of the volume: no ties
    for strategy in (so.COST, so.SCORE):
        exp = so.extract_index(cv, strategy)
        assert_bits(sv.extractSelectedIndex(strategy, cv), exp)
        assert_bits(sv.extractSelectedIndex(strategy, dev(cv)), exp)
        d_cv = dev(cv)
        sv.set_test_option(d_cv, "extract_index_wide", 0)
        try:
            assert_bits(sv.extractSelectedIndex(strategy, d_cv), exp)
        finally:
            sv.set_test_option(d_cv, "extract_index_wide", 1)
    const = np.ones((3, 4, D), np.float32)
    assert np.all(host(sv.extractSelectedIndex(so.COST, const)) == D - 1)  # all ties -> largest index (F9)


def test_index_to_disp_and_selected_cost(rng):
    cv = rng.uniform(0, 1, (6, 7, 9)).astype(np.float32)
    idx = rng.integers(0, 9, (6, 7)).astype(np.int32)
    assert_bits(sv.selectedIndexToDisp(idx, 3), so.index_to_disp(idx, so.RIGHT_TO_LEFT, 3))
    assert_bits(sv.selectedIndexToDisp(idx, -2, sv.dispDirection.LeftToRight), so.index_to_disp(idx, so.LEFT_TO_RIGHT, -2))
    assert_bits(sv.selectedCost(cv, idx), np.take_along_axis(cv, idx[:, :, None].astype(np.int64), 2)[:, :, 0])


@pytest.mark.parametrize("sdir", [so.TCV_SAME, so.TCV_REVERSED, so.TCV_BOTH])
@pytest.mark.parametrize("r", [1, 2])
def test_truncated_cost_volume(rng, sdir, r):
    cv = rng.uniform(0, 1, (9, 15, 7)).astype(np.float32)
    idx = rng.integers(0, 7, (9, 15)).astype(np.int32)
    for ddir in (so.RIGHT_TO_LEFT, so.LEFT_TO_RIGHT):
        for (h_r, v_r) in [(0, 0), (2, 1)]:
            exp = so.truncated_cost_volume(cv, idx, h_r, v_r, r, sdir, ddir)
            got = sv.truncatedCostVolume(cv, idx, h_r, v_r, r, ddir, sdir)
            assert_close(got, exp, 0.0)


@pytest.mark.parametrize("kernel", [so.EQUIANGULAR, so.PARABOLA, so.GAUSSIAN])
def test_refine(rng, kernel):
    tcv = rng.uniform(0.1, 2, (8, 9, 3)).astype(np.float32)
    tcv[0, 0, 0] = np.nan
    tcv[1, 1] = 1.0  # 0/0
    raw = rng.integers(0, 50, (8, 9)).astype(np.int32)
    assert_close(sv.refineDispCostInterpolation(kernel, tcv, raw), so.refine_disp(tcv, raw, kernel), 1e-5)
    tcv5 = rng.uniform(0.1, 2, (4, 5, 5)).astype(np.float32)
    raw5 = rng.integers(0, 9, (4, 5)).astype(np.int32)
    assert_close(sv.refineDispCostInterpolation(kernel, tcv5, raw5), so.refine_disp(tcv5, raw5, kernel), 1e-5)
    assert sv.refineDispCostInterpolation(kernel, np.zeros((2, 2, 4), np.float32), np.zeros((2, 2), np.int32)).size == 0


# ------------------------------------------------------------------------------------------------ fused pipeline
def oracle_chain(func, img_l, img_r, h_r, v_r, D, n_dir, P, margins, refine, rh, rv):
    strategy = so.func_strategy(func)
    cv = so.unfold_cost_volume(func, img_l, img_r, h_r, v_r, D)
    vol = so.sgm(cv, n_dir, strategy, P[0], P[1], margins, P[2]) if n_dir else cv
    idx = so.extract_index(vol, strategy)
    out = {"cv": cv, "vol": vol, "idx": idx, "disp": so.index_to_disp(idx)}
    if refine is not None:
        out["refined"] = so.refine_disp(so.truncated_cost_volume(vol, idx, rh, rv, 1), idx, refine)
    return out


@pytest.mark.parametrize("func,h_r,v_r,D,n_dir,refine", [
    (MF.SAD, 2, 2, 16, 0, None),            # config 1 shape family: SAD 5x5 + argmin
    (MF.CENSUS, 4, 4, 24, 0, None),         # config 2: census 9x9 volume + argmin
    (MF.CENSUS, 4, 4, 40, 8, None),         # config 3: census + SGM-8
    (MF.CENSUS, 4, 4, 70, 4, so.PARABOLA),
    (MF.NCC, 5, 5, 20, 8, so.PARABOLA),     # config 4: NCC 11x11 + SGM-8 + parabola
    (MF.ZNCC, 2, 2, 12, 8, so.EQUIANGULAR),
    (MF.SSD, 1, 1, 9, 8, so.PARABOLA),      # Cost branch on a float volume
])
def test_stereo_match_equals_chain(func, h_r, v_r, D, n_dir, refine):
    src, tgt, _ = parallax_pair(30, 44, 10, 8, 12, 2, 6, seed=int(func) + D)
    P = (0.001, 0.01, 100.0)
    exp = oracle_chain(int(func), tgt, src, h_r, v_r, D, n_dir, P, (0, 0, 0, 0), refine, h_r, v_r)
    for mk in (lambda x: x, dev):
        res = sv.stereoMatch(func, mk(tgt), mk(src), h_r, v_r, D, sgmDirections=n_dir, P1=P[0], P2=P[1], Pout=P[2],
                             refineKernel=refine, refine_h_radius=h_r, refine_v_radius=v_r, want_cv=True, want_sgm_cv=bool(n_dir))
        census = int(func) in (MF.CENSUS, MF.HAMMING)
        if census:
            assert_bits(res["cv"], exp["cv"])
            assert_bits(res["disp"], exp["disp"])
            if n_dir:
                assert_bits(res["sgm_cv"], exp["vol"])
        else:
            assert_close(res["cv"], exp["cv"])
            # winners are compared on the GPU's own volume (float costs differ in the last bits from the CPU's)
            gvol = host(res["sgm_cv"] if n_dir else res["cv"])
            assert_bits(res["disp"], so.index_to_disp(so.extract_index(gvol, so.func_strategy(int(func)))))
            assert (host(res["disp"]) == exp["disp"]).mean() > 0.99
        if refine is not None:
            gvol = host(res["sgm_cv"] if n_dir else res["cv"])
            gidx = so.extract_index(gvol, so.func_strategy(int(func)))
            assert_close(res["refined"], so.refine_disp(so.truncated_cost_volume(gvol, gidx, h_r, v_r, 1), gidx, refine), 1e-5)
            if census:
                assert_close(res["refined"], exp["refined"])
    # the fused call without volume outputs gives the same map
    lean = sv.stereoMatch(func, dev(tgt), dev(src), h_r, v_r, D, sgmDirections=n_dir, P1=P[0], P2=P[1], Pout=P[2])
    assert_bits(lean["disp"], host(res["disp"]))


def test_stereo_match_margins_and_direction(rng):
    src, tgt, _ = parallax_pair(26, 40, 8, 6, 10, 1, 5, seed=11)
    P = (0.5, 2.0, 9.0)
    cv = so.unfold_cost_volume(so.CENSUS, tgt, src, 4, 4, 33, so.LEFT_TO_RIGHT)
    vol = so.sgm(cv, 8, so.COST, *P[:2], (2, 1, 3, 2), P[2])
    idx = so.extract_index(vol, so.COST)
    res = sv.stereoMatch(MF.CENSUS, dev(tgt), dev(src), 4, 4, 33, dDir=sv.dispDirection.LeftToRight, sgmDirections=8, P1=P[0], P2=P[1],
                         Pout=P[2], margins=sv.Margins(2, 1, 3, 2), want_sgm_cv=True)
    assert_bits(res["sgm_cv"], vol)
    assert_bits(res["disp"], so.index_to_disp(idx, so.LEFT_TO_RIGHT))


def test_disparity_shards_reduce_to_full_argmin(rng):
    """Multi-GPU contract on one device: per-shard keys, element-wise min (what the RCCL all-reduce does), decode."""
    src, tgt, _ = parallax_pair(24, 36, 8, 6, 10, 1, 5, seed=5)
    D = 24
    for func, red in ((MF.CENSUS, np.minimum), (MF.SAD, np.minimum), (MF.NCC, np.maximum)):
        full = sv.stereoMatch(func, dev(tgt), dev(src), 4, 4, D, want_cv=True)
        keys = None
        for (b, n) in ((0, 7), (7, 9), (16, 8)):
            part = sv.stereoMatch(func, dev(tgt), dev(src), 4, 4, D, want_keys=True, want_cv=True, shard=(b, n))
            assert_bits(part["cv"], host(full["cv"])[:, :, b:b + n])
            k = host(part["keys"]).view(np.uint64)
            keys = k if keys is None else red(keys, k)
        idx = sv.keysToIndex(sv.matchFuncStrategy(func), keys, D)
        assert_bits(idx, host(full["disp"]))
    # NaN costs at a shard's local index 0 (a window of zeros in the target has norm 0): the first-NaN rule belongs to global index 0 only
    tgt2 = tgt.copy()
    tgt2[:, 12:23] = 0.0
    full = sv.stereoMatch(MF.NCC, dev(tgt2), dev(src), 4, 4, D, want_cv=True)
    assert np.isnan(host(full["cv"])).any() and not np.isnan(host(full["cv"])).all()
    assert_bits(full["disp"], so.index_to_disp(so.extract_index(host(full["cv"]), so.SCORE)))
    keys = None
    for (b, n) in ((0, 8), (8, 4), (12, 12)):
        part = sv.stereoMatch(MF.NCC, dev(tgt2), dev(src), 4, 4, D, want_keys=True, want_cv=True, shard=(b, n))
        assert np.isnan(host(part["cv"])[:, :, 0]).any()
        k = host(part["keys"]).view(np.uint64)
        keys = k if keys is None else np.maximum(keys, k)
    assert_bits(sv.keysToIndex(so.SCORE, keys, D), host(full["disp"]))


def test_non_dense_device_output(rng):
    cv = rng.uniform(0, 1, (6, 8, 10)).astype(np.float32)
    # cost volume stored (row, disparity, col) like the reference's aggregateCost (cross_correlations.h:220)
    stored = dev(np.ascontiguousarray(cv.transpose(0, 2, 1))).permute(0, 2, 1)
    assert not stored.is_contiguous()
    assert_bits(sv.extractSelectedIndex(so.COST, stored), so.extract_index(cv, so.COST))
    assert_bits(sv.sgmCostVolume(8, so.SCORE, stored, 0.1, 0.4, None, 3.0), so.sgm(cv, 8, so.SCORE, 0.1, 0.4, (0, 0, 0, 0), 3.0))


# ------------------------------------------------------------------------------------------------ census fast path
@pytest.mark.parametrize("D", [1, 7, 70, 256, 300])
@pytest.mark.parametrize("ddir", [sv.dispDirection.RightToLeft, sv.dispDirection.LeftToRight])
def test_census_fast_path_equals_general_and_oracle(D, ddir):
    """The pixel-per-lane census kernels (integer-exact min_p recurrence) against the general wave-per-line kernels
    and the oracle: disparity map, refined map and reduction keys."""
    src, tgt, _ = parallax_pair(37, 300, 12, 8, 40, 3, 17, seed=D)
    cases = [((0, 0, 0, 0), 100.0, 8, 4, 4), ((3, 2, 5, 1), 7.0, 8, 4, 4), ((0, 0, 0, 0), 100.0, 4, 3, 3), ((0, 0, 0, 0), 2.5, 8, 4, 4),
             ((0, 0, 0, 0), 100.0, 0, 2, 2), ((0, 0, 0, 0), 1e6, 8, 4, 4)]
    for margins, Pout, n_dir, h_r, v_r in cases:
        kw = dict(dDir=ddir, sgmDirections=n_dir, P1=0.3, P2=0.9, Pout=Pout, margins=sv.Margins(*margins), refineKernel=so.PARABOLA,
                  refine_h_radius=h_r, refine_v_radius=v_r, want_keys=True)
        d_tgt, d_src = dev(tgt), dev(src)
        try:
            sv.set_test_option(d_tgt, "census_fast_path", 1)
            fast = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, h_r, v_r, D, **kw)
            sv.set_test_option(d_tgt, "census_fast_path", 0)
            gen = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, h_r, v_r, D, **kw)
        finally:
            sv.set_test_option(d_tgt, "census_fast_path", 1)
        for k in ("disp", "keys"):
            assert_bits(fast[k], host(gen[k]))
        assert_close(fast["refined"], host(gen["refined"]), 0.0)
        cv = so.unfold_cost_volume(so.CENSUS, tgt, src, h_r, v_r, D, int(ddir))
        vol = so.sgm(cv, n_dir, so.COST, 0.3, 0.9, margins, Pout) if n_dir else cv
        idx = so.extract_index(vol, so.COST)
        assert_bits(fast["disp"], so.index_to_disp(idx, int(ddir)))
        assert_close(fast["refined"], so.refine_disp(so.truncated_cost_volume(vol, idx, h_r, v_r, 1), idx, so.PARABOLA), 1e-6)


@pytest.mark.parametrize("D", [1, 7, 70, 256, 300])
@pytest.mark.parametrize("ddir", [sv.dispDirection.RightToLeft, sv.dispDirection.LeftToRight])
def test_census_winner_without_line_scans(D, ddir):
    """Disparity map alone: in the integer-exact regime the per-pass minima of sgm.h:257-296 shift every disparity of a pixel alike,
    so the winner needs neither the g map nor the line scans ("census_winner_shortcut", the default).  Against the same call with the
    scans (option 0), against the general wave-per-line kernels and against the oracle's full cost volume + SGM + argmin."""
    src, tgt, _ = parallax_pair(41, 310, 12, 8, 40, 3, 17, seed=100 + D)
    cases = [((0, 0, 0, 0), 100.0, 8, 4, 4), ((3, 2, 5, 1), 7.0, 8, 4, 4), ((0, 0, 0, 0), 100.0, 4, 3, 3), ((0, 0, 0, 0), 2.0, 8, 4, 4),
             ((0, 0, 0, 0), 100.0, 0, 2, 2), ((2, 7, 0, 3), 0.0, 8, 3, 3), ((40, 0, 0, 0), 5.0, 8, 4, 4)]
    for margins, Pout, n_dir, h_r, v_r in cases:
        kw = dict(dDir=ddir, sgmDirections=n_dir, P1=0.3, P2=0.9, Pout=Pout, margins=sv.Margins(*margins))
        d_tgt, d_src = dev(tgt), dev(src)
        try:
            short = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, h_r, v_r, D, **kw)
            sv.set_option(d_tgt, "census_winner_shortcut", 0)
            scans = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, h_r, v_r, D, **kw)
            sv.set_test_option(d_tgt, "census_fast_path", 0)
            gen = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, h_r, v_r, D, **kw)
        finally:
            sv.set_option(d_tgt, "census_winner_shortcut", 1)
            sv.set_test_option(d_tgt, "census_fast_path", 1)
        assert_bits(short["disp"], host(scans["disp"]))
        assert_bits(short["disp"], host(gen["disp"]))
        cv = so.unfold_cost_volume(so.CENSUS, tgt, src, h_r, v_r, D, int(ddir))
        vol = so.sgm(cv, n_dir, so.COST, 0.3, 0.9, margins, Pout) if n_dir else cv
        assert_bits(short["disp"], so.index_to_disp(so.extract_index(vol, so.COST), int(ddir)))


@pytest.mark.parametrize("kernel", [so.PARABOLA, so.EQUIANGULAR, so.GAUSSIAN])
@pytest.mark.parametrize("ddir", [sv.dispDirection.RightToLeft, sv.dispDirection.LeftToRight])
def test_census_refined_without_line_scans(kernel, ddir):
    """Refined map: the parabola and equiangular kernels only subtract the three taps from one another (cost_based_refinement.h:43-69),
    so in the exact regime the per-pixel constant the line scans would add cancels and they are skipped as well; the Gaussian kernel
    takes logarithms of the taps and keeps the scans.  Same refined map bit for bit with the option off, and the oracle's within 1e-6."""
    for D, (margins, Pout, n_dir, r) in [(64, ((0, 0, 0, 0), 100.0, 8, 4)), (100, ((3, 2, 5, 1), 7.0, 8, 4)), (256, ((0, 0, 0, 0), 100.0, 4, 3))]:
        src, tgt, _ = parallax_pair(37, 300, 12, 8, 40, 3, 17, seed=900 + D)
        kw = dict(dDir=ddir, sgmDirections=n_dir, P1=0.3, P2=0.9, Pout=Pout, margins=sv.Margins(*margins), refineKernel=kernel,
                  refine_h_radius=r, refine_v_radius=r)
        d_tgt, d_src = dev(tgt), dev(src)
        try:
            short = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, r, r, D, **kw)
            sv.set_option(d_tgt, "census_winner_shortcut", 0)
            scans = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, r, r, D, **kw)
        finally:
            sv.set_option(d_tgt, "census_winner_shortcut", 1)
        assert_bits(short["disp"], host(scans["disp"]))
        assert_close(short["refined"], host(scans["refined"]), 0.0)
        cv = so.unfold_cost_volume(so.CENSUS, tgt, src, r, r, D, int(ddir))
        vol = so.sgm(cv, n_dir, so.COST, 0.3, 0.9, margins, Pout)
        idx = so.extract_index(vol, so.COST)
        if kernel != so.GAUSSIAN:  # (the oracle's log of a non-positive S is a NaN the GPU reproduces; compared above)
            assert_close(short["refined"], so.refine_disp(so.truncated_cost_volume(vol, idx, r, r, 1), idx, kernel), 1e-6)


# ------------------------------------------------------------------------------------------------ row bands
@pytest.mark.parametrize("ddir", [sv.dispDirection.RightToLeft, sv.dispDirection.LeftToRight])
@pytest.mark.parametrize("D", [32, 96, 256])
def test_census_row_bands_equal_single_call(D, ddir):
    """svh_census_band_match: any split of the rows gives the rows of the single call's disparity map bit for bit (census words of
    the band's window rows only: halo of v_radius rows), with margins, 8 / 4 / 0 directions, both directions."""
    src, tgt, _ = parallax_pair(61, 330, 12, 8, 40, 3, 17, seed=700 + D)
    H = src.shape[0]
    d_tgt, d_src = dev(tgt), dev(src)
    for margins, Pout, n_dir, r in [((0, 0, 0, 0), 100.0, 8, 4), ((3, 2, 5, 1), 7.0, 8, 4), ((0, 0, 0, 0), 100.0, 4, 3), ((0, 0, 0, 0), 3.0, 0, 3),
                                     ((0, 30, 0, 20), 9.0, 8, 5)]:
        kw = dict(dDir=ddir, sgmDirections=n_dir, P1=0.3, P2=0.9, Pout=Pout, margins=sv.Margins(*margins))
        full = host(sv.stereoMatch(MF.CENSUS, d_tgt, d_src, r, r, D, **kw)["disp"])
        for bands in ([(0, H)], [(0, 20), (20, 21), (41, 20)], [(0, 1), (1, 1), (2, 58), (60, 1)], [(7, 9)]):
            for b, c in bands:
                got = sv.censusBandMatch(d_tgt, d_src, r, r, D, (b, c), **kw)
                assert_bits(got, full[b:b + c])
        # host arrays in, host band out
        assert_bits(sv.censusBandMatch(tgt, src, r, r, D, (11, 13), **kw), full[11:24])


def test_census_row_bands_unsupported_cases():
    from libstevi_amd._capi import SvhError, ERR_UNSUPPORTED, ERR_INVALID_ARGUMENT
    src, tgt, _ = parallax_pair(20, 100, 6, 5, 8, 1, 4, seed=3)
    # 40 disparities: no whole number of the matrix-core sweep's row tiles -- taken since round 5 (the RightToLeft sweep masks the cells past
    # the range): the band equals the rows of the whole map
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    whole = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, 4, 4, 40, sgmDirections=8)["disp"]
    assert torch.equal(sv.censusBandMatch(d_tgt, d_src, 4, 4, 40, (3, 10)), whole[3:13])
    with pytest.raises(SvhError) as e:  # 20 disparities: fewer than the sweep's smallest range
        sv.censusBandMatch(tgt, src, 4, 4, 20, (0, 10))
    assert e.value.status == ERR_UNSUPPORTED
    with pytest.raises(SvhError) as e:  # LeftToRight, no multiple of 32: the general kernel's tiles only
        sv.censusBandMatch(src, tgt, 4, 4, 40, (0, 10), dDir=sv.dispDirection.LeftToRight)
    assert e.value.status == ERR_UNSUPPORTED
    with pytest.raises(SvhError) as e:  # non-integer Pout: outside the exact regime
        sv.censusBandMatch(tgt, src, 4, 4, 64, (0, 10), Pout=2.5)
    assert e.value.status == ERR_UNSUPPORTED
    with pytest.raises(SvhError) as e:
        sv.censusBandMatch(tgt, src, 4, 4, 64, (15, 10))
    assert e.value.status == ERR_INVALID_ARGUMENT


# ------------------------------------------------------------------------------------------------ disparity shards
@pytest.mark.parametrize("ddir", [sv.dispDirection.RightToLeft, sv.dispDirection.LeftToRight])
def test_census_shards_equal_single_gpu(ddir):
    """svh_census_shard_keys / _finish: three shards reduced with an element-wise min (what the RCCL all-reduce does)
    give the single-call disparity and refined maps bit for bit; the keys match the numpy restatement."""
    from shard_protocol import shard_keys
    src, tgt, _ = parallax_pair(33, 290, 10, 8, 30, 3, 21, seed=42)
    D = 300
    for n_dir, Pout, margins in ((8, 100.0, (0, 0, 0, 0)), (8, 5.0, (3, 1, 2, 4)), (0, 100.0, (0, 0, 0, 0))):
        kw = dict(dDir=ddir, sgmDirections=n_dir, P1=0.2, P2=0.7, Pout=Pout, margins=sv.Margins(*margins))
        d_tgt, d_src = dev(tgt), dev(src)
        full = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, 4, 4, D, refineKernel=so.PARABOLA, refine_h_radius=4, refine_v_radius=4, **kw)
        cv = so.unfold_cost_volume(so.CENSUS, tgt, src, 4, 4, D, int(ddir))
        keys = None
        # RightToLeft with equal widths and a range that starts at 0: plane 1 is written as the winner over all shards
        plane1_global = sv.censusShardRegion1IsGlobal(d_tgt, d_src, D, ddir)
        assert plane1_global == (ddir == sv.dispDirection.RightToLeft)
        for (b, n) in ((0, 101), (101, 64), (165, 135)):
            k = sv.censusShardKeys(d_tgt, d_src, 4, 4, D, (b, n), **kw)
            assert_bits(k, shard_keys(cv[:, :, b:b + n], b, cv.shape[1], cv if plane1_global else None))
            if plane1_global and keys is not None:  # only plane 0 needs the reduction: plane 1 is the same on every shard
                assert torch.equal(keys[..., 1], k[..., 1])
            keys = k if keys is None else torch.minimum(keys, k)
        res = sv.censusShardFinish(d_tgt, d_src, keys, 4, 4, D, refineKernel=so.PARABOLA, refine_h_radius=4, refine_v_radius=4, **kw)
        assert_bits(res["disp"], host(full["disp"]))
        assert_close(res["refined"], host(full["refined"]), 0.0)
    from libstevi_amd._capi import SvhError, ERR_UNSUPPORTED
    with pytest.raises(SvhError) as e:  # non-integer Pout leaves the exact regime: the disparity axis no longer shards
        sv.censusShardKeys(dev(tgt), dev(src), 4, 4, D, (0, 100), Pout=2.5)
    assert e.value.status == ERR_UNSUPPORTED


@pytest.mark.parametrize("D", [70, 64, 128])
def test_sgm_integer_volume_routes_agree(rng, D):
    """svh_sgm_cost_volume on an integer-valued float volume takes the probe + scan route; it must equal the
    wave-per-line route (option off) and the oracle bit for bit, also when a single voxel breaks integrality."""
    cv = rng.integers(0, 65, (23, 31, D)).astype(np.float32)
    for breaker in (None, (5, 7, 3, 0.5), (0, 0, 0, np.inf), (22, 30, D - 1, 1e6)):
        v = cv.copy()
        if breaker is not None:
            v[breaker[0], breaker[1], breaker[2]] = breaker[3]
        d = dev(v)
        for margins, Pout in (((0, 0, 0, 0), 100.0), ((2, 1, 0, 3), 7.0)):
            exp = so.sgm(v, 8, so.COST, 0.3, 0.9, margins, Pout)
            try:
                sv.set_test_option(d, "census_fast_path", 1)
                a = host(sv.sgmCostVolume(8, so.COST, d, 0.3, 0.9, sv.Margins(*margins), Pout))
                sv.set_test_option(d, "census_fast_path", 0)
                b = host(sv.sgmCostVolume(8, so.COST, d, 0.3, 0.9, sv.Margins(*margins), Pout))
            finally:
                sv.set_test_option(d, "census_fast_path", 1)
            for got in (a, b):
                assert np.array_equal(np.isnan(got), np.isnan(exp))
                ok = ~np.isnan(exp)
                assert np.array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32))


@pytest.mark.parametrize("D", [70, 40, 72, 128])
def test_sgm_cost_branch_two_minima_route(rng, D):
    """The Cost branch on a float volume reads the volume once: a probe leaves every pixel's two regional minima (costs that look inside
    / past the right border) and the six line recurrences run on those two numbers (monotonicity of the per-disparity step,
    svh_sgm.hip).  Must equal the oracle and the wave-per-line sweeps of the volume (option off) bit for bit: magnitudes over fifteen
    decades, negative costs, NaN / +-inf voxels and whole non-finite pixels, an infinite Pout, margins, four directions, D not a
    multiple of 64 (a multiple of four up to 128: the probe and the apply kernel that pack several pixels into a wave); and volumes whose
    magnitudes could overflow along a line (> 1e30) must fall back to the sweeps by themselves."""
    H, W = 19, 37
    base = (rng.uniform(-1, 1, (H, W, D)) * 10.0 ** rng.integers(-6, 9, (H, W, 1))).astype(np.float32)
    holes = base.copy()
    holes[3, 5, 7] = np.nan
    holes[0, 0, :] = np.inf           # a pixel without a finite cost: the line's minimum becomes +inf and the next pixel restarts
    holes[7, 30:, :] = np.nan         # ... several in a row, at the right border
    holes[11, 2, 5] = -np.inf
    holes[H - 1, W - 1, D - 1] = np.inf
    holes[9, W - 2, 1:] = np.nan      # only the in-image disparity is finite
    huge = base.copy()
    huge[4, 4, 4] = 3e38              # outside the regime: the probe's second flag bit sends the call to the volume sweeps
    huge[12, 20, D - 10] = -2.5e38
    for name, cv in (("magnitudes", base), ("non-finite", holes), ("huge", huge)):
        d = dev(cv)
        for n_dir, margins, Pout in ((8, (0, 0, 0, 0), 100.0), (8, (2, 1, 3, 2), 0.37), (4, (0, 0, 0, 0), -3.5), (8, (0, 0, 0, 0), np.inf), (8, (0, 3, 0, 0), 1e31)):
            exp = so.sgm(cv, n_dir, so.COST, 0.3, 0.9, margins, Pout, variant=0)  # the literal O(D^2) loops of sgm.h:257-296
            outs = {}
            try:
                for route in (1, 0):
                    sv.set_test_option(d, "sgm_cost_two_minima", route)
                    sv.profile_reset(d)
                    sv.profile_enable(d, True)
                    outs[route] = host(sv.sgmCostVolume(n_dir, so.COST, d, 0.3, 0.9, sv.Margins(*margins), Pout))
                    sv.profile_enable(d, False)
                    kernels = sv.profile_collect(d)
                    if route == 1 and np.isinf(Pout) or route == 1 and abs(Pout) <= 1e30:
                        assert "sgm_cost_minmap_scalar" in kernels and "sgm_volume_probe" in kernels, (name, kernels.keys())
                    else:  # option off, or a finite Pout beyond the regime: no probe for the minima, the sweeps run unconditionally
                        assert "sgm_cost_minmap_scalar" not in kernels
            finally:
                sv.set_test_option(d, "sgm_cost_two_minima", 1)
            for route, got in outs.items():
                assert np.array_equal(np.isnan(got), np.isnan(exp)), (name, n_dir, margins, Pout, route)
                ok = ~np.isnan(exp)
                assert np.array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32)), (name, n_dir, margins, Pout, route)
    # the winner fused behind it (svh_stereo_match, SAD + SGM: Cost branch on the float volume the cost kernel wrote)
    src, tgt, _ = parallax_pair(30, 90, 10, 8, 30, 2, 9, seed=77)
    exp_cv = so.unfold_cost_volume(so.SAD, tgt, src, 2, 2, 40)
    exp_idx = so.extract_index(so.sgm(exp_cv, 8, so.COST, 0.01, 0.1, (0, 0, 0, 0), 5.0), so.COST)
    got = sv.stereoMatch(MF.SAD, dev(tgt), dev(src), 2, 2, 40, sgmDirections=8, P1=0.01, P2=0.1, Pout=5.0, want_cv=True)
    if np.array_equal(host(got["cv"]), exp_cv):  # (the column-sum cost kernel agrees to 1e-4, not always to the bit: the map is only comparable on equal volumes)
        assert np.array_equal(host(got["disp"]), so.index_to_disp(exp_idx))
    same_cv = so.extract_index(so.sgm(host(got["cv"]), 8, so.COST, 0.01, 0.1, (0, 0, 0, 0), 5.0), so.COST)
    assert np.array_equal(host(got["disp"]), so.index_to_disp(same_cv))


def _with_option(t, name, value, fn):
    sv.set_test_option(t, name, value)
    try:
        return fn()
    finally:
        sv.set_test_option(t, name, 1)


@pytest.mark.parametrize("func", [MF.SAD, MF.SSD, MF.CC, MF.NCC, MF.ZNCC, MF.ZSSD, MF.ZCC])
def test_winner_and_minima_reduced_inside_the_cost_kernel(func):
    """svh_stereo_match with a float matching function lets the cost kernel reduce over the disparity axis while it holds a pixel's costs
    (option "cost_reduce_fused"): without SGM the winner -- the volume is then never written --, with a Cost-branch SGM the two regional
    minima its recurrences run on (no probing read).  Same maps as the separate kernels (option 0) and as the oracle's selection on the
    library's own volume: D a multiple of 16 or not, fewer disparity blocks than waves, both directions, ties, NaN costs."""
    strat = sv.matchFuncStrategy(func)
    for (H, W), D, r, ddir in (((37, 150), 64, 2, sv.dispDirection.RightToLeft), ((20, 130), 40, 1, sv.dispDirection.LeftToRight), ((9, 121), 200, 3, sv.dispDirection.RightToLeft),
                               ((64, 64), 16, 2, sv.dispDirection.RightToLeft), ((5, 300), 7, 5, sv.dispDirection.RightToLeft)):
        src, tgt, _ = parallax_pair(H, W, max(2, min(H, W) // 4), H // 4, W // 4, 2, 9, seed=H + W + D)
        src, tgt = src[:H, :W].copy(), tgt[:H, :W].copy()
        l, r_ = dev(tgt), dev(src)
        run = lambda **kw: sv.stereoMatch(func, l, r_, r, r, D, dDir=ddir, **kw)  # noqa: E731
        sv.profile_reset(l)
        sv.profile_enable(l, True)
        a = run()
        sv.profile_enable(l, False)
        kernels = sv.profile_collect(l)
        b = _with_option(l, "cost_reduce_fused", 0, run)
        if func != MF.ZSAD and r <= 5:
            assert "extract_index" not in kernels and "index_to_disp" not in kernels, kernels.keys()
        assert np.array_equal(host(a["disp"]), host(b["disp"])), (func, H, W, D)
        full = run(want_cv=True)
        assert np.array_equal(host(full["disp"]), host(a["disp"]))
        exp = so.index_to_disp(so.extract_index(host(full["cv"]), strat), int(ddir))
        assert np.array_equal(host(a["disp"]), exp)
        if int(strat) == so.COST:  # Cost-branch SGM on the float volume: minima from the cost kernel
            sgm_on = lambda **kw: sv.stereoMatch(func, l, r_, r, r, D, dDir=ddir, sgmDirections=8, P1=0.01, P2=0.1, Pout=3.5, **kw)  # noqa: E731
            sv.profile_reset(l)
            sv.profile_enable(l, True)
            c = sgm_on(want_sgm_cv=True, want_cv=True)
            sv.profile_enable(l, False)
            kernels = sv.profile_collect(l)
            assert "sgm_volume_probe" not in kernels and "sgm_cost_minmap_scalar" in kernels, kernels.keys()
            d_ = _with_option(l, "cost_reduce_fused", 0, lambda: sgm_on(want_sgm_cv=True))
            assert np.array_equal(host(c["disp"]), host(d_["disp"]))
            assert_bits(c["sgm_cv"], host(d_["sgm_cv"]))
            assert_bits(c["sgm_cv"], so.sgm(host(c["cv"]), 8, so.COST, 0.01, 0.1, (0, 0, 0, 0), 3.5, variant=0))
    # ties everywhere (constant images: the larger index wins) and NaN costs (NCC of an all-zero image: index 0)
    flat = dev(np.full((12, 140), 0.25, np.float32))
    zero = dev(np.zeros((12, 140), np.float32))
    for img in (flat, zero):
        a = sv.stereoMatch(func, img, img, 2, 2, 48)
        b = _with_option(img, "cost_reduce_fused", 0, lambda: sv.stereoMatch(func, img, img, 2, 2, 48))
        assert np.array_equal(host(a["disp"]), host(b["disp"]))
        cvh = host(sv.stereoMatch(func, img, img, 2, 2, 48, want_cv=True)["cv"])
        assert np.array_equal(host(a["disp"]), so.index_to_disp(so.extract_index(cvh, strat)))


@pytest.mark.parametrize("func", [MF.SAD, MF.SSD, MF.NCC, MF.ZNCC, MF.ZSSD])
def test_winner_reduced_inside_the_cost_kernel_on_colour_images(rng, func):
    """The same for RGB images (round 5: the column-sum kernel of interleaved channels has a winner form too): a disparity-only call without
    SGM writes no volume and launches no extract_index; same map as with the option off and as the oracle's selection on the library's
    own volume; two- and three-channel images, both directions, ranges off the multiples of 16."""
    strat = sv.matchFuncStrategy(func)
    for (H, W, C), D, r, ddir in (((30, 150, 3), 64, 2, sv.dispDirection.RightToLeft), ((17, 131, 3), 40, 3, sv.dispDirection.LeftToRight), ((12, 200, 2), 23, 1, sv.dispDirection.RightToLeft)):
        g_src, g_tgt, _ = parallax_pair(H, W, max(2, H // 4), H // 4, W // 4, 2, 9, seed=H + W + D)
        gains = np.linspace(0.7, 1.3, C).astype(np.float32)
        src = (g_src[:H, :W, None] * gains + 0.05 * rng.uniform(-1, 1, (H, W, C))).astype(np.float32)
        tgt = (g_tgt[:H, :W, None] * gains + 0.05 * rng.uniform(-1, 1, (H, W, C))).astype(np.float32)
        l, r_ = dev(tgt), dev(src)
        run = lambda **kw: sv.stereoMatch(func, l, r_, r, r, D, dDir=ddir, **kw)  # noqa: E731
        sv.profile_reset(l)
        sv.profile_enable(l, True)
        a = run()
        sv.profile_enable(l, False)
        kernels = sv.profile_collect(l)
        assert "extract_index" not in kernels and "index_to_disp" not in kernels, kernels.keys()
        b = _with_option(l, "cost_reduce_fused", 0, run)
        assert np.array_equal(host(a["disp"]), host(b["disp"])), (func, H, W, C, D)
        full = run(want_cv=True)
        exp = so.index_to_disp(so.extract_index(host(full["cv"]), strat), int(ddir))
        assert np.array_equal(host(a["disp"]), exp)


@pytest.mark.parametrize("strategy", [so.COST, so.SCORE])
def test_winner_travels_with_the_aggregated_volume(rng, strategy):
    """sgmCostVolume(..., keep_winner=True) on a device tensor keeps, with the volume it returns, the index map extractSelectedIndex would
    scan it for (svh_sgm_cost_volume_winner: the kernel that writes a pixel's final costs picks it); extractSelectedIndex on the untouched
    tensor returns a copy of that map -- no extract_index launch -- equal to the scan and to the oracle's; any in-place change, another
    strategy, dropMinima, or a call without keep_winner and the scan runs."""
    for shape, n_dir, margins in (((37, 90, 64), 8, None), ((20, 70, 128), 8, None), ((33, 41, 40), 8, None), ((25, 60, 64), 4, None), ((30, 50, 64), 8, sv.Margins(2, 1, 0, 3)),
                                  ((21, 40, 256), 8, None), ((12, 30, 320), 8, None)):
        cv = (rng.integers(0, 30, shape) if strategy == so.COST and shape[2] == 128 else rng.uniform(-1, 1, shape)).astype(np.float32)
        cv[1, 2, 3] = np.nan
        d = dev(cv)
        m = margins.as_tuple() if margins else (0, 0, 0, 0)
        exp_vol = so.sgm(cv, n_dir, strategy, 0.01, 0.1, m, 5.0)
        exp = so.extract_index(exp_vol, strategy)
        s = sv.sgmCostVolume(n_dir, strategy, d, 0.01, 0.1, margins, 5.0, keep_winner=True)
        got_vol = host(s)
        assert np.array_equal(np.isnan(got_vol), np.isnan(exp_vol)) and np.array_equal(got_vol[~np.isnan(exp_vol)].view(np.uint32), exp_vol[~np.isnan(exp_vol)].view(np.uint32))
        # (Score: from 256 disparities on the banded sweep of a whole image leaves winner records; below that, with margins or four directions
        # the call scans the volume it has just written -- the map travels with the volume either way)
        assert getattr(s, "_svh_winner", None) is not None
        sv.profile_reset(d)
        sv.profile_enable(d, True)
        idx = sv.extractSelectedIndex(strategy, s)
        sv.profile_enable(d, False)
        assert "extract_index" not in sv.profile_collect(d)
        assert np.array_equal(host(idx), exp)
        idx += 1  # the caller's copy: the statement's map is untouched
        assert np.array_equal(host(sv.extractSelectedIndex(strategy, s)), exp)
        other = so.SCORE if strategy == so.COST else so.COST
        assert np.array_equal(host(sv.extractSelectedIndex(other, s)), so.extract_index(got_vol, other))
        s[0, 0, 0] += 1.0  # in-place change: the statement no longer describes the tensor
        assert np.array_equal(host(sv.extractSelectedIndex(strategy, s)), so.extract_index(host(s), strategy))
        s2 = sv.sgmCostVolume(n_dir, strategy, d, 0.01, 0.1, margins, 5.0, keep_winner=True)
        sv.dropMinima(s2)
        assert getattr(s2, "_svh_winner", None) is None
        assert getattr(sv.sgmCostVolume(n_dir, strategy, d, 0.01, 0.1, margins, 5.0), "_svh_winner", None) is None


def _order_code(v):
    u = np.where(v == 0, np.float32(0), v).astype(np.float32).view(np.uint32)
    return np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint64)


@pytest.mark.parametrize("D", [4, 8, 40, 64, 68, 128, 132, 256, 260, 300, 512, 516, 1028])
def test_cost_apply_pipelined_form(rng, D):
    """Aligned dense volumes with D % 4 == 0 take sgm_cost_apply_piped_kernel (more than 128 disparities: rows through buffer descriptors,
    two batches in flight, per-batch winner stores, uncounted stores) or sgm_cost_apply_packed_kernel (128 or fewer: four disparities per
    lane, two or four pixels per wave, reductions inside a pixel's lanes): the aggregated volume and the winner it leaves, bit for bit against
    the oracle -- rows that are no multiple of 64 R (lanes past D), images narrower than a batch of pixels, pixels whose every disparity
    looks past the border, NaN / +-inf voxels, whole non-finite pixels (a non-finite min_p for the next one), magnitudes outside the lean
    regime, an infinite Pout, margins (also margins that leave no pixel to aggregate), four directions."""
    for H, W in ((7, 11), (3, 2), (4, 1), (1, 5), (2, 3)) if D < 1000 else ((3, 5), (2, 1)):
        base = (rng.uniform(-1, 1, (H, W, D)) * 10.0 ** rng.integers(-3, 4, (H, W, 1))).astype(np.float32)
        ints = rng.integers(0, 65, (H, W, D)).astype(np.float32)
        holes = base.copy()
        holes[0, 0, :] = np.inf
        holes[H // 2, W // 2, 1] = -np.inf
        holes[H - 1, W - 1, D - 1] = np.nan
        holes[H - 1, 0, 0] = np.nan  # "the value at index 0 is NaN": index 0 wins
        holes[0, W - 1, : D - 3] = np.nan
        huge = base.copy()
        huge[H // 2, 0, min(5, D - 1)] = 3e38
        for name, cv in (("ints", ints), ("magnitudes", base), ("non-finite", holes), ("huge", huge)):
            d = dev(cv)
            for n_dir, margins, Pout in ((8, (0, 0, 0, 0), 100.0), (8, (1, 0, 1, 0), 0.37), (4, (0, 0, 0, 0), -3.5), (8, (0, 0, 0, 0), np.inf), (8, (5, 0, 9, 0), 1.0)):
                exp = so.sgm(cv, n_dir, so.COST, 0.3, 0.9, margins, Pout)
                sv.profile_reset(d)
                sv.profile_enable(d, True)
                got_t = sv.sgmCostVolume(n_dir, so.COST, d, 0.3, 0.9, sv.Margins(*margins), Pout, keep_winner=True)
                sv.profile_enable(d, False)
                assert "sgm_cost_apply" in sv.profile_collect(d)
                got = host(got_t)
                what = (name, (H, W, D), n_dir, margins, Pout)
                assert np.array_equal(np.isnan(got), np.isnan(exp)), what
                ok = ~np.isnan(exp)
                assert np.array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32)), what
                assert getattr(got_t, "_svh_winner", None) is not None
                assert np.array_equal(host(sv.extractSelectedIndex(so.COST, got_t)), so.extract_index(exp, so.COST)), what


@pytest.mark.parametrize("func,D,W", [(MF.SAD, 132, 150), (MF.SAD, 256, 300), (MF.SSD, 300, 310), (MF.SAD, 260, 64), (MF.SAD, 64, 100), (MF.SAD, 128, 140),
                                      (MF.SSD, 40, 30), (MF.SAD, 8, 21)])
def test_cost_apply_pipelined_form_winner_outputs(func, D, W):
    """... and every per-pixel output of the fused call (disparity, reduction keys, the three refinement taps through the refined map) from
    that kernel, with and without the aggregated volume written beside them, against the oracle run on the GPU's own cost volume."""
    H = 9
    src, tgt, _ = parallax_pair(H, W, 5, 2, W // 3, 2, min(W // 4, 40), seed=D + W)
    kw = dict(sgmDirections=8, P1=0.01, P2=0.1, Pout=5.0, refineKernel=so.PARABOLA, refine_h_radius=2, refine_v_radius=2, want_keys=True)
    full = sv.stereoMatch(func, dev(tgt), dev(src), 2, 2, D, want_cv=True, want_sgm_cv=True, **kw)
    vol = so.sgm(host(full["cv"]), 8, so.COST, 0.01, 0.1, (0, 0, 0, 0), 5.0)
    assert_bits(full["sgm_cv"], vol)
    idx = so.extract_index(vol, so.COST)
    exp_refined = so.refine_disp(so.truncated_cost_volume(vol, idx, 2, 2, 1), idx, so.PARABOLA)
    best = np.take_along_axis(vol, idx[..., None].astype(np.int64), axis=2)[..., 0]
    for res in (full, sv.stereoMatch(func, dev(tgt), dev(src), 2, 2, D, **kw)):
        assert np.array_equal(host(res["disp"]), so.index_to_disp(idx))
        keys = host(res["keys"]).view(np.uint64)
        assert np.array_equal(keys >> np.uint64(32), _order_code(best))
        assert np.array_equal((keys & np.uint64(0xFFFFFFFF)).astype(np.int64), D - 1 - idx.astype(np.int64))
        assert_close(res["refined"], exp_refined, 1e-6)


# ------------------------------------------------------------------------------------------------ randomised sweep
def test_random_configurations_against_oracle():
    """Seeded random shapes / windows / ranges / margins / penalties through the fused pipeline and the per-function
    entry points: odd sizes, images narrower than a block, D > W, negative first offsets, both directions."""
    r = np.random.default_rng(20260401)
    funcs = [MF.CENSUS, MF.CENSUS, MF.CENSUS, MF.SAD, MF.ZNCC, MF.SSD, MF.NCC]
    for case in range(28):
        H, W = int(r.integers(1, 40)), int(r.integers(2, 300))
        D = int(r.integers(1, 90)) if case % 3 else int(r.integers(90, 330))
        func = funcs[case % len(funcs)]
        h_r, v_r = (int(r.integers(3, 6)), int(r.integers(3, 6))) if func == MF.CENSUS else (int(r.integers(1, 4)), int(r.integers(0, 4)))
        if func == MF.CENSUS and case % 5 == 0:
            h_r, v_r = int(r.integers(0, 3)), int(r.integers(1, 3))  # windows with no fully written word at all
            if (2 * h_r + 1) * (2 * v_r + 1) <= 1:
                v_r = 1
        n_dir = int(r.choice([0, 4, 8, 8]))
        lower = int(r.integers(-5, 6)) if case % 4 == 0 else 0
        ddir = int(r.integers(0, 2))
        margins = tuple(int(x) for x in r.integers(0, 3, 4)) if case % 2 else (0, 0, 0, 0)
        Pout = float(r.choice([100.0, 7.0, 0.0, 2.5, -3.0]))
        P1, P2 = float(r.uniform(0, 2)), float(r.uniform(0, 4))
        left = r.uniform(-1, 1, (H, W)).astype(np.float32)
        right = r.uniform(-1, 1, (H, W)).astype(np.float32)
        strategy = so.func_strategy(int(func))
        cv = so.unfold_cost_volume(int(func), left, right, h_r, v_r, D, ddir, lower)
        rng_arg = D if lower == 0 else sv.searchOffset1(lower, lower + D - 1)
        res = sv.stereoMatch(func, dev(left), dev(right), h_r, v_r, rng_arg, dDir=ddir, sgmDirections=n_dir, P1=P1, P2=P2, Pout=Pout,
                             margins=sv.Margins(*margins), want_cv=True, want_sgm_cv=bool(n_dir))
        tag = f"case {case}: {func.name} {H}x{W}xD{D} win {h_r},{v_r} dirs {n_dir} lower {lower} ddir {ddir} margins {margins} Pout {Pout}"
        gcv = host(res["cv"])
        if func == MF.CENSUS:
            assert np.array_equal(gcv, cv), tag
        else:
            assert_close(gcv, cv)
        vol = so.sgm(gcv, n_dir, strategy, P1, P2, margins, Pout) if n_dir else gcv
        if n_dir:
            gvol = host(res["sgm_cv"])
            assert np.array_equal(np.isnan(gvol), np.isnan(vol)), tag
            ok = ~np.isnan(vol)
            assert np.array_equal(gvol[ok].view(np.uint32), vol[ok].view(np.uint32)), tag
        idx = so.extract_index(vol, strategy)
        assert np.array_equal(host(res["disp"]), so.index_to_disp(idx, ddir, (1 if ddir == so.RIGHT_TO_LEFT else -1) * lower)), tag
        # the lean call (no volumes) takes the pixel-per-lane kernels for census
        lean = sv.stereoMatch(func, dev(left), dev(right), h_r, v_r, rng_arg, dDir=ddir, sgmDirections=n_dir, P1=P1, P2=P2, Pout=Pout,
                              margins=sv.Margins(*margins))
        assert np.array_equal(host(lean["disp"]), host(res["disp"])), tag


def test_random_volumes_through_the_packed_and_pipelined_kernels():
    """Seeded random volumes whose rows are a multiple of four costs -- the forms that put several pixels into a wave (<= 128 costs), one pixel
    on 64 lanes (<= 256) or pipeline their loads (> 128) -- through sgmCostVolume (both branches, with and without the winner kept), the
    probe route (no statement) and extractSelectedIndex: shapes down to one pixel, rows of 4 ... 640 costs, margins, penalties of either
    sign, an infinite Pout, NaN / +-inf voxels and pixels, integer and real costs, magnitudes on either side of the regime."""
    r = np.random.default_rng(20261004)
    for case in range(36):
        H, W = int(r.integers(1, 24)), int(r.integers(1, 70))
        D = 4 * int(r.integers(1, 33)) if case % 4 else 4 * int(r.integers(33, 161))
        kind = case % 5
        if kind == 0:
            cv = r.integers(0, 40, (H, W, D)).astype(np.float32)
        else:
            cv = (r.uniform(-1, 1, (H, W, D)) * 10.0 ** r.integers(-2, 3, (H, W, 1))).astype(np.float32)
        if kind >= 2:  # holes
            for _ in range(1 + H * W // 6):
                i, j = int(r.integers(0, H)), int(r.integers(0, W))
                what = int(r.integers(0, 5))
                if what == 0:
                    cv[i, j, int(r.integers(0, D))] = np.nan
                elif what == 1:
                    cv[i, j, :] = np.inf
                elif what == 2:
                    cv[i, j, 0] = np.nan
                elif what == 3:
                    cv[i, j, int(r.integers(0, D))] = -np.inf
                else:
                    cv[i, j, : int(r.integers(0, D))] = np.nan
        if kind == 4:
            cv[int(r.integers(0, H)), int(r.integers(0, W)), int(r.integers(0, D))] = 2.5e38  # outside the regime: the volume sweeps
        n_dir = int(r.choice([4, 8, 8, 8]))
        margins = tuple(int(x) for x in r.integers(0, 3, 4)) if case % 3 == 0 else (0, 0, 0, 0)
        Pout = float(r.choice([100.0, 7.0, 0.0, 2.5, -3.0, np.inf]))
        P1 = float(r.uniform(0, 2))
        P2 = P1 + float(r.uniform(0, 3))
        d = dev(cv)
        tag = f"case {case}: {H}x{W}xD{D} kind {kind} dirs {n_dir} margins {margins} Pout {Pout}"
        for strategy in (so.COST, so.SCORE):
            exp = so.sgm(cv, n_dir, strategy, P1, P2, margins, Pout)
            exp_idx = so.extract_index(exp, strategy)
            for keep in (False, True):
                got_t = sv.sgmCostVolume(n_dir, strategy, d, P1, P2, sv.Margins(*margins), Pout, keep_winner=keep)
                got = host(got_t)
                assert np.array_equal(np.isnan(got), np.isnan(exp)), (tag, strategy, keep)
                ok = ~np.isnan(exp)
                assert np.array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32)), (tag, strategy, keep)
                assert np.array_equal(host(sv.extractSelectedIndex(strategy, got_t)), exp_idx), (tag, strategy, keep)
            assert np.array_equal(host(sv.extractSelectedIndex(strategy, d)), so.extract_index(cv, strategy)), (tag, strategy)


# ------------------------------------------------------------------------------------------------ textbook SGM (explicit second mode)
@pytest.mark.parametrize("strategy", [so.COST, so.SCORE])
@pytest.mark.parametrize("D", [1, 5, 64, 70, 130, 256, 300])
def test_textbook_sgm_matches_its_definition(rng, strategy, D):
    """svh_sgm_cost_volume_textbook against so_sgm_textbook (oracle/stevi_oracle.c): not the reference's behaviour, an extra mode
    (SURVEY.md section 8f rank 4).  Same float operations in the same order: equal values (the Cost form runs on negated costs,
    so a zero may come out with the other sign)."""
    H, W = (9, 21) if D > 100 else (13, 17)
    cv = rng.uniform(0, 8, (H, W, D)).astype(np.float32)
    if D > 2:
        cv[1, 2, 1] = np.inf
        cv[3, 4, 0] = np.nan
    for n_dir, margins, (P1, P2), Pout in ((8, (0, 0, 0, 0), (0.001, 0.01), 100.0), (4, (0, 0, 0, 0), (1.5, 0.5), 7.0), (8, (2, 1, 0, 3), (0.25, 3.0), -2.5)):
        exp = so.sgm_textbook(cv, n_dir, strategy, P1, P2, margins, Pout)
        for mk in (lambda x: x, dev):
            got = host(sv.sgmCostVolume(n_dir, strategy, mk(cv), P1, P2, sv.Margins(*margins), Pout, semantics="textbook"))
            assert np.array_equal(np.isnan(got), np.isnan(exp))
            ok = ~np.isnan(exp)
            assert np.array_equal(got[ok], exp[ok])


def test_textbook_sgm_differs_from_the_reference_mode(rng):
    cv = rng.uniform(0, 8, (12, 16, 20)).astype(np.float32)
    ref = host(sv.sgmCostVolume(8, so.COST, dev(cv), 0.5, 2.0, sv.Margins(), 100.0))
    tb = host(sv.sgmCostVolume(8, so.COST, dev(cv), 0.5, 2.0, sv.Margins(), 100.0, semantics="textbook"))
    assert np.array_equal(ref, so.sgm(cv, 8, so.COST, 0.5, 2.0, (0, 0, 0, 0), 100.0))  # the default stays the reference as written
    assert not np.array_equal(ref, tb)


def test_cost_branch_on_a_census_volume_skips_the_probe_with_the_same_bits(rng):
    """unfoldBasedCostVolume(CENSUS, keep_minima=True) on device tensors leaves the volume's regional minima with the tensor
    (svh_unfold_cost_volume_minima); a sgmCostVolume<Cost> on the untouched tensor uses them instead of probing the volume
    (svh_sgm_cost_volume_minima): same S bit for bit, no sgm_volume_probe launch.  Any in-place change of the tensor, a copy or a view with
    other strides drops the statement; so does dropMinima; without keep_minima there is none (opt-in: ADVICE r03)."""
    src, tgt, _ = parallax_pair(40, 200, 14, 9, 30, 3, 17, seed=5)
    l, r = dev(tgt), dev(src)
    for D, margins, n_dir in ((64, None, 8), (48, sv.Margins(3, 2, 5, 1), 4), (256, None, 8)):
        assert getattr(sv.unfoldBasedCostVolume(MF.CENSUS, l, r, 4, 4, D), "_svh_minima", None) is None
        cv = sv.unfoldBasedCostVolume(MF.CENSUS, l, r, 4, 4, D, keep_minima=True)
        assert getattr(cv, "_svh_minima", None) is not None
        exp = so.sgm(so.unfold_cost_volume(so.CENSUS, tgt, src, 4, 4, D), n_dir, so.COST, 0.001, 0.01, margins.as_tuple() if margins else (0, 0, 0, 0), 100.0)
        sv.profile_reset(l)
        sv.profile_enable(l, True)
        got = sv.sgmCostVolume(n_dir, so.COST, cv, 0.001, 0.01, margins, 100.0)
        sv.profile_enable(l, False)
        kernels = sv.profile_collect(l)
        assert "sgm_volume_probe" not in kernels and "gmap_from_minima" in kernels
        assert_bits(got, exp)
        # minima themselves: smallest cost of each region
        mins = cv._svh_minima[0].cpu().numpy()
        c = host(cv)
        jj, dd = np.meshgrid(np.arange(c.shape[1]), np.arange(D), indexing="ij")
        pays = (jj + dd >= c.shape[1])[None]
        m0 = np.where(pays, np.inf, c).min(axis=2)
        m1 = np.where(pays, c, np.inf).min(axis=2)
        assert np.array_equal(mins[..., 0], m0) and np.array_equal(mins[..., 1], m1)
        # a non-integer Pout is outside the exact regime: the statement is simply not used, same bits as ever
        got2 = sv.sgmCostVolume(n_dir, so.COST, cv, 0.001, 0.01, margins, 0.5)
        assert_bits(got2, so.sgm(host(cv), n_dir, so.COST, 0.001, 0.01, margins.as_tuple() if margins else (0, 0, 0, 0), 0.5))
        # in-place change: the statement no longer describes the tensor -> probe path, and the result follows the new values
        cv[3, 5, 2] += 0.25
        sv.profile_reset(l)
        sv.profile_enable(l, True)
        got3 = sv.sgmCostVolume(n_dir, so.COST, cv, 0.001, 0.01, margins, 100.0)
        sv.profile_enable(l, False)
        assert "gmap_from_minima" not in sv.profile_collect(l)
        assert_bits(got3, so.sgm(host(cv), n_dir, so.COST, 0.001, 0.01, margins.as_tuple() if margins else (0, 0, 0, 0), 100.0))
    # withdrawn by hand (for writes torch's version counter cannot see): probe path again
    cv = sv.unfoldBasedCostVolume(MF.CENSUS, l, r, 4, 4, 32, keep_minima=True)
    sv.dropMinima(cv)
    sv.profile_reset(l)
    sv.profile_enable(l, True)
    got4 = sv.sgmCostVolume(8, so.COST, cv, 0.001, 0.01, None, 100.0)
    sv.profile_enable(l, False)
    assert "gmap_from_minima" not in sv.profile_collect(l)
    assert_bits(got4, so.sgm(host(cv), 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0))
    # Score strategy: no statement (its SGM branch has no use for minima)
    assert getattr(sv.unfoldBasedCostVolume(MF.NCC, l, r, 2, 2, 16, keep_minima=True), "_svh_minima", None) is None
    # float costs of a Cost-strategy function: the column-sum kernel leaves the minima of the finite costs and vouches for the regime
    # (kind 2); sgmCostVolume<Cost> then runs its line recurrences on them without a probing read -- same S as the oracle's literal loops
    for func, r_ in ((MF.SAD, 2), (MF.SSD, 3), (MF.ZSSD, 2)):
        cvf = sv.unfoldBasedCostVolume(func, l, r, r_, r_, 48, keep_minima=True)
        assert cvf._svh_minima[5] == 2
        sv.profile_reset(l)
        sv.profile_enable(l, True)
        got = sv.sgmCostVolume(8, so.COST, cvf, 0.01, 0.1, None, 2.5)
        sv.profile_enable(l, False)
        kernels = sv.profile_collect(l)
        assert "sgm_volume_probe" not in kernels and "sgm_cost_minmap_scalar" in kernels, kernels.keys()
        assert_bits(got, so.sgm(host(cvf), 8, so.COST, 0.01, 0.1, (0, 0, 0, 0), 2.5, variant=0))
        c = host(cvf)
        jj, dd = np.meshgrid(np.arange(c.shape[1]), np.arange(48), indexing="ij")
        pays = (jj + dd >= c.shape[1])[None]
        mins = cvf._svh_minima[0].cpu().numpy()
        assert np.array_equal(mins[..., 0], np.where(pays, np.inf, c).min(axis=2)) and np.array_equal(mins[..., 1], np.where(pays, c, np.inf).min(axis=2))
        cvf[2, 3, 4] = 7.0  # in-place change: probe again
        sv.profile_reset(l)
        sv.profile_enable(l, True)
        got = sv.sgmCostVolume(8, so.COST, cvf, 0.01, 0.1, None, 2.5)
        sv.profile_enable(l, False)
        assert "sgm_volume_probe" in sv.profile_collect(l)
        assert_bits(got, so.sgm(host(cvf), 8, so.COST, 0.01, 0.1, (0, 0, 0, 0), 2.5, variant=0))
    # ZSAD takes the per-window kernel, which makes no minima: nothing is stated
    assert getattr(sv.unfoldBasedCostVolume(MF.ZSAD, l, r, 2, 2, 16, keep_minima=True), "_svh_minima", None) is None
    # an image with an infinite sample: a cost leaves the regime, the library states nothing and the probe finds out by itself
    bad = tgt.copy()
    bad[10, 50] = np.inf
    cvb = sv.unfoldBasedCostVolume(MF.SAD, dev(bad), r, 2, 2, 48, keep_minima=True)
    assert getattr(cvb, "_svh_minima", None) is None
    gb, eb = host(sv.sgmCostVolume(8, so.COST, cvb, 0.01, 0.1, None, 2.5)), so.sgm(host(cvb), 8, so.COST, 0.01, 0.1, (0, 0, 0, 0), 2.5, variant=0)
    assert np.array_equal(np.isnan(gb), np.isnan(eb)) and np.array_equal(gb[~np.isnan(eb)].view(np.uint32), eb[~np.isnan(eb)].view(np.uint32))


# ------------------------------------------------------------------------------------------------ sgmCostVolume<.., T_CV>
@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.uint16, np.int32, np.uint32])
@pytest.mark.parametrize("strategy", [so.COST, so.SCORE])
def test_sgm_integer_cost_volume_types(rng, dtype, strategy):
    """sgmCostVolume is a template on the volume's element type (sgm.h:360); the reference casts every cost it reads to float
    (:234, :273, :299), so an integer volume gives what its float copy gives -- bit for bit, host and device arrays."""
    hi = {np.uint8: 255, np.int16: 3000, np.uint16: 60000, np.int32: 1 << 26, np.uint32: 1 << 27}[dtype]
    lo = -hi if np.issubdtype(dtype, np.signedinteger) else 0
    cv = rng.integers(lo, hi, (23, 41, 19)).astype(dtype)
    exp = so.sgm(cv.astype(np.float32), 8, strategy, 0.5, 2.0, (0, 0, 0, 0), 7.0)
    got = sv.sgmCostVolume(8, strategy, cv, 0.5, 2.0, None, 7.0)
    assert_bits(got, exp)
    if dtype in (np.uint8, np.int16, np.int32):  # (torch has device tensors of these)
        got_d = sv.sgmCostVolume(8, strategy, dev(cv), 0.5, 2.0, None, 7.0)
        assert_bits(got_d, exp)
    got_t = sv.sgmCostVolume(4, strategy, cv, 0.5, 2.0, sv.Margins(1, 2, 0, 1), 7.0, semantics="textbook")
    ref_t = sv.sgmCostVolume(4, strategy, cv.astype(np.float32), 0.5, 2.0, sv.Margins(1, 2, 0, 1), 7.0, semantics="textbook")
    assert_bits(got_t, host(ref_t))


@pytest.mark.parametrize("func", [MF.NCC, MF.ZNCC, MF.CC, MF.ZCC, MF.SSD, MF.ZSSD, MF.SAD, MF.ZSAD])
def test_feature_cost_volume_tiled_same_bits_as_per_voxel(rng, func):
    """Option "feature_volume_tiled" (default): the feature volumes are processed once and compared from LDS; 0: the per-voxel kernel.
    Same bits, and the oracle within the float tolerance: feature counts odd / even / one, both directions, target narrower and wider
    than the source, rows wider than a block, disparity ranges that leave the image, the reference's compressor chain (17 features)."""
    for (H, Ws, Wt, F, D, ddir) in [(9, 150, 150, 17, 40, sv.dispDirection.RightToLeft), (7, 130, 100, 16, 33, sv.dispDirection.LeftToRight),
                                     (5, 70, 90, 1, 70, sv.dispDirection.RightToLeft), (6, 200, 200, 25, 128, sv.dispDirection.LeftToRight),
                                     (4, 333, 333, 20, 320, sv.dispDirection.RightToLeft), (3, 131, 140, 21, 65, sv.dispDirection.LeftToRight), (3, 65, 65, 32, 64, sv.dispDirection.RightToLeft),
                                     (2, 129, 129, 33, 50, sv.dispDirection.RightToLeft),
                                     (3, 100, 100, 9, 48, sv.dispDirection.LeftToRight), (2, 70, 80, 27, 30, sv.dispDirection.RightToLeft),
                                     # long vectors (9x9, 11x11 unfolded and more): the range in chunks, fewer pixels per processing block
                                     (5, 100, 100, 81, 140, sv.dispDirection.RightToLeft), (4, 90, 80, 121, 70, sv.dispDirection.LeftToRight),
                                     (3, 70, 70, 200, 33, sv.dispDirection.RightToLeft)]:
        fl = rng.uniform(-1, 1, (H, Wt if ddir == sv.dispDirection.RightToLeft else Ws, F)).astype(np.float32)
        fr = rng.uniform(-1, 1, (H, Ws if ddir == sv.dispDirection.RightToLeft else Wt, F)).astype(np.float32)
        dl, dr = dev(fl), dev(fr)
        outs = []
        for tiled, records in ((1, 1), (0, 1), (1, 0)):  # (records: vectors of up to 32 floats with the target record in registers, round 5)
            sv.set_test_option(dl, "feature_volume_tiled", tiled)
            sv.set_test_option(dl, "feature_volume_records", records)
            try:
                outs.append(host(sv.featureVolume2CostVolume(func, dl, dr, D, ddir)))
            finally:
                sv.set_test_option(dl, "feature_volume_tiled", 1)
                sv.set_test_option(dl, "feature_volume_records", 1)
        for other in outs[1:]:
            assert np.array_equal(np.isnan(outs[0]), np.isnan(other))
            ok = ~np.isnan(outs[0])
            assert np.array_equal(outs[0][ok].view(np.uint32), other[ok].view(np.uint32)), (H, Ws, Wt, F, D)
        assert_close(outs[0], so.feature_cost_volume(int(func), fl, fr, D, int(ddir)))
    img_l = rng.uniform(-1, 1, (20, 140)).astype(np.float32)
    img_r = rng.uniform(-1, 1, (20, 140)).astype(np.float32)
    comp = sv.UnFoldCompressor(sv.CompressorGenerators.GrPix17R3Filter())
    a = host(sv.unfoldBasedCostVolumeCompressed(func, dev(img_l), dev(img_r), comp, 24))
    sv.set_test_option(dev(img_l), "feature_volume_tiled", 0)
    try:
        b = host(sv.unfoldBasedCostVolumeCompressed(func, dev(img_l), dev(img_r), comp, 24))
    finally:
        sv.set_test_option(dev(img_l), "feature_volume_tiled", 1)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.uint32), b[~np.isnan(b)].view(np.uint32))


@pytest.mark.parametrize("func", [MF.NCC, MF.ZNCC, MF.CC, MF.ZCC, MF.SSD, MF.ZSSD, MF.SAD, MF.ZSAD, MF.CENSUS])
def test_cost_volume_carries_its_winner(rng, func):
    """unfoldBasedCostVolume(..., keep_winner=True): the kernel that writes the volume picks extractSelectedIndex's index while it holds a
    pixel's costs (svh_unfold_cost_volume_winner), and extractSelectedIndex on the untouched tensor returns that map.  Equal to the scan
    of the same volume (ties, both strategies, both directions, ranges that are no multiple of 16); functions without that kernel (ZSAD,
    census) leave no statement and the scan runs; an in-place write withdraws it."""
    strat = sv.matchFuncStrategy(func)
    for (H, W, r, D, ddir) in [(23, 150, 2, 40, sv.dispDirection.RightToLeft), (17, 131, 3, 33, sv.dispDirection.LeftToRight), (9, 64, 1, 70, sv.dispDirection.RightToLeft)]:
        left = np.round(rng.uniform(-2, 2, (H, W)) * 4).astype(np.float32) / 4  # (a coarse grid: ties between disparities)
        right = np.round(rng.uniform(-2, 2, (H, W)) * 4).astype(np.float32) / 4
        dl, dr = dev(left), dev(right)
        plain = sv.unfoldBasedCostVolume(func, dl, dr, r, r, D, ddir)
        kept = sv.unfoldBasedCostVolume(func, dl, dr, r, r, D, ddir, keep_winner=True)
        assert_bits(kept, host(plain))
        has = getattr(kept, "_svh_winner", None) is not None
        assert has == (func not in (MF.ZSAD, MF.CENSUS))
        want = host(sv.extractSelectedIndex(strat, plain))
        assert np.array_equal(host(sv.extractSelectedIndex(strat, kept)), want)
        assert np.array_equal(want, so.extract_index(host(plain), int(strat)))
        if has:
            kept.add_(1.0)  # an in-place write: the statement no longer describes the tensor
            assert np.array_equal(host(sv.extractSelectedIndex(strat, kept)), so.extract_index(host(kept), int(strat)))


@pytest.mark.parametrize("func", [MF.NCC, MF.ZNCC, MF.CC, MF.ZCC, MF.SSD, MF.ZSSD, MF.SAD])
def test_colour_images_through_the_column_sum_kernel(rng, func):
    """Images of 2 - 4 interleaved channels (the reference's unfold takes nImDim = 3: unfold.h:292-344): the column-sum kernel stages the
    channels as window rows of their own; ranges whose tile is over the LDS budget run in chunks.  Against the oracle (1e-4) and the
    per-voxel kernel (option cost_volume_colsum 0): both directions, images wider than a tile, 3x3 - 7x7 windows, a chunked range,
    disparities that leave the image, and the fused call's winner."""
    for (H, W, C, r, D, ddir) in [(9, 150, 3, 2, 40, sv.dispDirection.RightToLeft), (7, 131, 3, 3, 70, sv.dispDirection.LeftToRight),
                                   (6, 140, 2, 1, 33, sv.dispDirection.RightToLeft), (5, 129, 4, 2, 17, sv.dispDirection.LeftToRight),
                                   (4, 300, 3, 3, 300, sv.dispDirection.RightToLeft)]:
        left = rng.uniform(-1, 1, (H, W, C)).astype(np.float32)
        right = rng.uniform(-1, 1, (H, W, C)).astype(np.float32)
        dl, dr = dev(left), dev(right)
        exp = so.unfold_cost_volume(int(func), left, right, r, r, D, int(ddir))
        got = host(sv.unfoldBasedCostVolume(func, dl, dr, r, r, D, ddir))
        sv.set_test_option(dl, "cost_volume_colsum", 0)
        try:
            per_voxel = host(sv.unfoldBasedCostVolume(func, dl, dr, r, r, D, ddir))
        finally:
            sv.set_test_option(dl, "cost_volume_colsum", 1)
        for other in (exp, per_voxel):
            assert np.array_equal(np.isnan(got), np.isnan(other)), (H, W, C, r, D)
            ok = ~np.isnan(other)
            assert np.all(np.abs(got[ok] - other[ok]) <= 1e-4 * np.maximum(1, np.abs(other[ok]))), (H, W, C, r, D)
        strat = sv.matchFuncStrategy(func)
        fused = sv.stereoMatch(func, dl, dr, r, r, D, dDir=ddir)["disp"]
        assert np.array_equal(host(fused), so.index_to_disp(so.extract_index(got, int(strat)), int(ddir)))


@pytest.mark.parametrize("func", [MF.NCC, MF.ZNCC, MF.SSD, MF.ZSSD, MF.SAD, MF.ZSAD])
def test_wide_windows_and_chunked_ranges_through_the_column_sum_kernel(rng, func):
    """Windows 13 and 15 wide (h_radius 6, 7) and disparity ranges whose tile is over the LDS budget (run in chunks of the range) take
    the column-sum kernel instead of the per-voxel one.  Against the oracle within 1e-4, both directions, non-square windows."""
    for (H, W, h_r, v_r, D, ddir) in [(8, 150, 6, 6, 20, sv.dispDirection.RightToLeft), (9, 140, 7, 7, 33, sv.dispDirection.LeftToRight),
                                       (7, 131, 7, 2, 17, sv.dispDirection.RightToLeft), (3, 260, 3, 3, 1100, sv.dispDirection.RightToLeft),
                                       (4, 300, 5, 5, 700, sv.dispDirection.LeftToRight)]:
        left = rng.uniform(-1, 1, (H, W)).astype(np.float32)
        right = rng.uniform(-1, 1, (H, W)).astype(np.float32)
        exp = so.unfold_cost_volume(int(func), left, right, h_r, v_r, D, int(ddir))
        dl = dev(left)
        sv.profile_reset(dl)
        sv.profile_enable(dl, True)
        got = host(sv.unfoldBasedCostVolume(func, dl, dev(right), h_r, v_r, D, ddir))
        sv.profile_enable(dl, False)
        assert "cost_volume_tiled" in sv.profile_collect(dl), (h_r, v_r, D)
        assert np.array_equal(np.isnan(got), np.isnan(exp)), (H, W, h_r, v_r, D)
        ok = ~np.isnan(exp)
        assert np.all(np.abs(got[ok] - exp[ok]) <= 1e-4 * np.maximum(1, np.abs(exp[ok]))), (H, W, h_r, v_r, D)


@pytest.mark.parametrize("h_r,v_r", [(6, 6), (6, 5), (7, 5), (7, 6), (7, 7), (9, 7), (8, 8)])
def test_census_windows_of_five_words(rng, h_r, v_r):
    """Census windows of five to eight written words (13x13: 168 comparisons, 15x15: 224, 19x15: 284) and one of nine (17x17, the general
    kernels): the Hamming volume through the tiled kernel, and the census + SGM pipeline (the pixel-per-lane kernels with long records;
    float and integer penalties) against the oracle, bit for bit."""
    from helpers import parallax_pair
    H, W, D = 40, 150, 40
    src, tgt, _ = parallax_pair(H, W, 12, 10, 40, 2, 9, seed=h_r * 10 + v_r)
    dl, dr = dev(tgt), dev(src)
    cv = so.unfold_cost_volume(so.CENSUS, tgt, src, h_r, v_r, D)
    assert_bits(sv.unfoldBasedCostVolume(MF.CENSUS, dl, dr, h_r, v_r, D), cv)
    for n_dir in (8, 4):
        vol = so.sgm(cv, n_dir, so.COST, 0.3, 0.9, (0, 0, 0, 0), 100.0)
        want = so.index_to_disp(so.extract_index(vol, so.COST))
        a = sv.stereoMatch(MF.CENSUS, dl, dr, h_r, v_r, D, sgmDirections=n_dir, P1=0.3, P2=0.9, want_sgm_cv=True)
        assert np.array_equal(host(a["disp"]), want)
        assert_bits(a["sgm_cv"], vol)
        vol = so.sgm(cv, n_dir, so.COST, 3.0, 17.0, (0, 0, 0, 0), 100.0)
        want = so.index_to_disp(so.extract_index(vol, so.COST))
        a = sv.stereoMatch(MF.CENSUS, dl, dr, h_r, v_r, D, sgmDirections=n_dir, P1=3.0, P2=17.0, want_sgm_cv=True)
        assert np.array_equal(host(a["disp"]), want)
        assert_bits(a["sgm_cv"], vol)
        assert np.array_equal(host(sv.stereoMatch(MF.CENSUS, dl, dr, h_r, v_r, D, sgmDirections=n_dir, P1=3.0, P2=17.0)["disp"]), want)
    if (2 * h_r + 1) * (2 * v_r + 1) - 1 < 9 * 32:  # disparity shards (uneven, one of them no multiple of 32): keys merged by MIN, then the finish
        kw = dict(sgmDirections=8, P1=3.0, P2=17.0)
        keys = None
        for b, n in ((0, 17), (17, D - 17)):
            k = sv.censusShardKeys(dl, dr, h_r, v_r, D, (b, n), **kw)
            keys = k if keys is None else torch.minimum(keys, k)
        vol = so.sgm(cv, 8, so.COST, 3.0, 17.0, (0, 0, 0, 0), 100.0)
        assert np.array_equal(host(sv.censusShardFinish(dl, dr, keys, h_r, v_r, D, **kw)["disp"]), so.index_to_disp(so.extract_index(vol, so.COST)))


# ------------------------------------------------------------------------------------------------ the float chain, end to end
@pytest.mark.parametrize("func, so_func", [(MF.NCC, so.NCC), (MF.ZNCC, so.ZNCC)])
def test_float_chain_against_the_oracles_own_chain(func, so_func):
    """NCC / ZNCC -> SGM-8 (Score) -> argmax -> truncatedCostVolume -> parabola, the GPU's chain against the ORACLE'S OWN chain from the images
    (the C4 checks feed the oracle the GPU's cost band, so nothing held the chain end to end).  Float costs agree to 1e-4, not to
    the bit, so the aggregated volumes are held to a tolerance (six passes of values of magnitude one), the winner to the oracle's volume --
    the GPU's winner must be within that tolerance of the oracle's maximum at its pixel (a near-tie may flip; an exact tie may not pick a
    smaller index) -- and the refined map to 1e-3 where the two winners agree and the parabola is well conditioned."""
    H, W, D, r = 60, 150, 48, 3
    src, tgt, _ = parallax_pair(H, W, 20, 15, 40, 3, 17, seed=123)
    P1, P2, Pout = 0.001, 0.01, 100.0
    res = sv.stereoMatch(func, dev(tgt), dev(src), r, r, D, sgmDirections=8, P1=P1, P2=P2, Pout=Pout, refineKernel=sv.InterpolationKernel.Parabola,
                         refine_h_radius=r, refine_v_radius=r, want_cv=True, want_sgm_cv=True)
    cv = so.unfold_cost_volume(so_func, tgt, src, r, r, D)
    S = so.sgm(cv, 8, so.SCORE, P1, P2, (0, 0, 0, 0), Pout)
    idx = so.extract_index(S, so.SCORE)
    refined = so.refine_disp(so.truncated_cost_volume(S, idx, r, r, 1), idx, so.PARABOLA)
    g_cv, g_S, g_disp, g_ref = host(res["cv"]), host(res["sgm_cv"]), host(res["disp"]), host(res["refined"])
    assert np.array_equal(np.isnan(g_cv), np.isnan(cv))
    ok = ~np.isnan(cv)
    assert np.max(np.abs(g_cv[ok] - cv[ok])) <= 1e-4
    okS = np.isfinite(S)
    assert np.array_equal(np.isfinite(g_S), okS)
    tol = 1e-3  # six passes, each adding a contribution formed from costs that agree to 1e-4 (in practice the volumes agree to a few 1e-6)
    assert np.max(np.abs(g_S[okS] - S[okS])) <= tol
    want = so.index_to_disp(idx)
    differ = g_disp != want
    assert differ.mean() <= 0.01, differ.mean()
    ii, jj = np.nonzero(differ)
    s_at_gpu = S[ii, jj, g_disp[ii, jj]]       # (RightToLeft, offset 0: the disparity is the index)
    s_at_want = S[ii, jj, want[ii, jj]]
    assert np.all(s_at_want - s_at_gpu <= 2 * tol), "a winner of the GPU's chain is not a near-tie of the oracle's volume"
    same = ~differ & ~np.isnan(refined) & ~np.isnan(g_ref)
    assert np.array_equal(np.isnan(refined[~differ]), np.isnan(g_ref[~differ]))
    # the parabola divides by the second difference of three aggregated values: compare where it is not tiny
    tcv = so.truncated_cost_volume(S, idx, r, r, 1)
    curv = np.abs(tcv[..., 0] - 2 * tcv[..., 1] + tcv[..., 2])
    well = same & (curv > 0.05)
    assert well.sum() > 0.5 * same.sum()
    assert np.max(np.abs(g_ref[well] - refined[well])) <= 1e-3

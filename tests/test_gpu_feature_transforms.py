"""Rows A7 / A8 as stand-alone entry points: channelsMean / channelsNorm / channelsZeroMeanNorm, the three feature-volume
transforms and getFeatureVolumeForMatchFunc.  Same float operations in the same order as the oracle -> bit-exact."""
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


@pytest.mark.parametrize("shape", [(9, 13, 1), (17, 21, 25), (5, 64, 121), (1, 1, 7)])
def test_statistics_and_transforms_bit_exact(rng, shape):
    feat = rng.uniform(-3, 3, shape).astype(np.float32)
    mean, norm, sigma = so.channels_mean(feat), so.channels_norm(feat), so.channels_zeromean_norm(feat)
    other_mean = rng.uniform(-1, 1, shape[:2]).astype(np.float32)
    for mk in (lambda x: x, dev):
        f = mk(feat)
        assert np.array_equal(bits(host(sv.channelsMean(f))), bits(mean))
        assert np.array_equal(bits(host(sv.channelsNorm(f))), bits(norm))
        assert np.array_equal(bits(host(sv.channelsZeroMeanNorm(f))), bits(sigma))
        assert np.array_equal(bits(host(sv.channelsZeroMeanNorm(f, mk(other_mean)))), bits(so.channels_zeromean_norm_given(feat, other_mean)))
        assert np.array_equal(bits(host(sv.zeromeanFeatureVolume(f, mk(mean)))), bits(so.affine_feature_volume(feat, mean)))
        assert np.array_equal(bits(host(sv.normalizedFeatureVolume(f, mk(norm)))), bits(so.affine_feature_volume(feat, None, norm)))
        assert np.array_equal(bits(host(sv.zeromeanNormalizedFeatureVolume(f, mk(mean), mk(sigma)))), bits(so.affine_feature_volume(feat, mean, sigma)))


@pytest.mark.parametrize("func", [MF.CC, MF.NCC, MF.SSD, MF.SAD, MF.ZCC, MF.ZNCC, MF.ZSSD, MF.ZSAD, MF.CENSUS, MF.HAMMING])
def test_feature_volume_for_match_func(rng, func):
    feat = rng.uniform(-2, 2, (11, 19, 49)).astype(np.float32)
    exp = so.feature_volume_for_match_func(int(func), feat)
    for mk in (lambda x: x, dev):
        got = host(sv.getFeatureVolumeForMatchFunc(func, mk(feat)))
        assert got.dtype == exp.dtype and got.shape == exp.shape
        assert np.array_equal(bits(got), bits(exp))
    # the volume feeds computeGuidedCV / aggregateCost: a constant pixel has zero norm -> NaN features for the normalised functions (E4)
    flat = np.ones((2, 3, 9), np.float32)
    got = host(sv.getFeatureVolumeForMatchFunc(func, flat))
    assert np.array_equal(np.isnan(got.astype(np.float64)), np.isnan(so.feature_volume_for_match_func(int(func), flat).astype(np.float64)))


def test_feature_transform_rules(rng):
    one = rng.uniform(-1, 1, (4, 5, 1)).astype(np.float32)
    assert sv.getFeatureVolumeForMatchFunc(MF.CENSUS, one).size == 0  # census.h:76-78
    with pytest.raises(Exception):
        sv.zeromeanFeatureVolume(rng.uniform(-1, 1, (4, 5, 3)).astype(np.float32), np.zeros((4, 6), np.float32))


@pytest.mark.parametrize("n", [1, 9, 25, 49, 81])
def test_reference_channel_mean_and_sigma_on_gpu(rng, n):
    """testCorrelationFilters.cpp:149-170, :232-262 (channelsMean / channelsZeroMeanNorm against a direct evaluation, 1e-3)."""
    v = rng.uniform(-1, 1, (1, 1, n)).astype(np.float32)
    mean = float(v.astype(np.float64).mean())
    assert abs(mean - host(sv.channelsMean(dev(v)))[0, 0]) < 1e-3
    sigma = np.sqrt(((v.astype(np.float64) - mean) ** 2).sum())
    assert abs(sigma - host(sv.channelsZeroMeanNorm(dev(v)))[0, 0]) < 1e-3


# ---- UnFoldCompressor (unfold.h:36-137, :346-471) ------------------------------------------
@pytest.mark.parametrize("shape", [(23, 31), (9, 12, 3), (7, 7)])
def test_unfold_compressed_bit_exact(rng, shape):
    img = rng.uniform(-1, 1, shape).astype(np.float32)
    odd = rng.integers(-1, 6, (5, 4)).astype(np.int32)  # even-sized mask, labels <= 0 ignored, label gaps
    for mask in (sv.CompressorGenerators.GrPix17R3Filter(), sv.CompressorGenerators.GrPix17R4Filter(), odd):
        comp = sv.UnFoldCompressor(mask)
        for pad in (None, sv.PaddingMargins(0), sv.PaddingMargins(2, 1, 0, 3)):
            opad = None if pad is None else pad.as_tuple()
            exp = so.unfold_compressed(img, mask, opad)
            for mk in (lambda x: x, dev):
                got = host(sv.unfoldCompressed(comp, mk(img), pad))
                if exp.size == 0:  # window larger than the padded image: the reference returns an empty array
                    assert got.size == 0
                    continue
                assert got.shape == exp.shape
                assert np.array_equal(bits(got), bits(exp))


def test_compressor_accessors_match_the_reference_rules():
    comp = sv.UnFoldCompressor(sv.CompressorGenerators.GrPix17R3Filter())
    assert (comp.nFeatures(), comp.width(), comp.height()) == (17, 7, 7) and comp.margins().as_tuple() == (3, 3, 3, 3)
    idx = comp.indices()
    assert len(idx) == 49 and [e.featureIndex for e in idx] == sorted(e.featureIndex for e in idx)
    assert (idx[0].verticalShift, idx[0].horizontalShift, idx[0].weight) == (0, 0, 1.0)  # superpixel 1 is the centre pixel alone
    off = sv.UnFoldCompressor(np.array([[0, 0, 0], [0, 0, 5], [0, 0, 5]], np.int32))  # off-centre labels: the box still holds the centre
    assert (off.width(), off.height()) == (2, 2) and off.margins().as_tuple() == (0, 0, 1, 1)


@pytest.mark.parametrize("h_radius,v_radius", [(1, 1), (3, 3), (5, 1), (1, 5), (5, 5)])
def test_reference_testUnfoldCompressor_on_gpu(rng, h_radius, v_radius):
    """testCorrelationFilters.cpp:502-577 against the HIP path."""
    h, w = 2 * v_radius + 1, 2 * h_radius + 1
    img = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    labels = (np.arange(h * w) % h) + 1
    rng.shuffle(labels)
    mask = labels.reshape(h, w).astype(np.int32)
    out = host(sv.unfoldCompressed(sv.UnFoldCompressor(mask), dev(img), sv.PaddingMargins(0)))
    assert out.shape == (1, 1, h)
    check = np.sort([img[mask == f].astype(np.float64).mean() for f in range(1, h + 1)])
    assert np.max(np.abs(np.sort(out[0, 0]) - check)) < 1e-4


def test_compressed_cost_volume_matches_oracle(rng):
    """unfoldBasedCostVolume / unfoldBased2dDisparityCostVolume with a compressor (cross_correlations.h:767-791, :824-851)"""
    left = rng.uniform(-1, 1, (21, 33)).astype(np.float32)
    right = rng.uniform(-1, 1, (21, 33)).astype(np.float32)
    mask = sv.CompressorGenerators.GrPix17R4Filter()
    comp = sv.UnFoldCompressor(mask)
    fl, fr = so.unfold_compressed(left, mask), so.unfold_compressed(right, mask)
    for func in (MF.ZNCC, MF.SAD, MF.NCC):
        got = host(sv.unfoldBasedCostVolumeCompressed(func, dev(left), dev(right), comp, 12))
        assert np.array_equal(bits(got), bits(so.feature_cost_volume(int(func), fl, fr, 12)))
        got2 = host(sv.unfoldBasedCostVolumeCompressed(func, dev(left), dev(right), comp, sv.searchOffset2(-1, 1, -2, 2)))
        assert np.array_equal(bits(got2), bits(so.feature_cost_volume_2d(int(func), fl, fr, (-1, 1), (-2, 2))))


# ---- unfold with a patch orientation (unfold.h:139-191, :247-344) --------------------------------------------------------
@pytest.mark.parametrize("shape,h_r,v_r", [((9, 11), 1, 2), ((8, 8, 3), 2, 1), ((6, 7), 3, 3)])
def test_unfold_orientations(rng, shape, h_r, v_r):
    img = rng.uniform(-1, 1, shape).astype(np.float32)
    h, v, Cc = 2 * h_r + 1, 2 * v_r + 1, (shape[2] if len(shape) == 3 else 1)
    base = so.unfold(img, h_r, v_r)
    for orient in sv.UnfoldPatchOrientation:
        exp = so.unfold(img, h_r, v_r, None, int(orient))
        for mk in (lambda x: x, dev):
            got = host(sv.unfold(h_r, v_r, mk(img), None, orient))
            assert np.array_equal(bits(got), bits(exp))
        # every orientation holds the same samples, permuted: rotating the (v, h) patch by 0 / 90 / 180 / 270 degrees
        patch = base.reshape(base.shape[0], base.shape[1], v, h, Cc)
        rot = np.rot90(patch, k=int(orient), axes=(2, 3))
        assert np.array_equal(exp, np.ascontiguousarray(rot).reshape(exp.shape))
    exp = so.unfold(img, h_r, v_r, (1, 0, 2, 1), so.ROTATE270)
    assert np.array_equal(bits(host(sv.unfold(h_r, v_r, dev(img), sv.PaddingMargins(1, 0, 2, 1), sv.UnfoldPatchOrientation.Rotate270))), bits(exp))

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

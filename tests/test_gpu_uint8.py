"""uint8 images ("Input/feature/cost types"): the reference keeps uint8 features for CENSUS / HAMMING / CC /
SSD / SAD and casts each sample to float inside the comparison (matching_costs.h:59-156, :749-783; census.h:89-101 compares the
uint8 samples themselves), so the results equal those of the same images converted to float32 -- which is what the oracle is fed
here.  The HIP path takes the bytes (svh_array dtype SVH_U8) and widens them on the device.  Normalised / zero-mean functions on
uint8 follow an int16 path the reference's own test skips (testCorrelationFilters.cpp:1249): refused."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import libstevi_amd as sv  # noqa: E402
import oracle as so  # noqa: E402

MF = sv.matchingFunctions
DEV = torch.device("cuda:0")


def u8_pair(H, W, C=None, seed=0, shift=3):
    rng = np.random.default_rng(seed)
    shape = (H, W + shift) if C is None else (H, W + shift, C)
    base = rng.integers(0, 256, shape, dtype=np.uint8)
    src, tgt = base[:, shift:].copy(), base[:, :W].copy()  # tgt(i, j + shift) = src(i, j)
    noise = rng.integers(0, 3, tgt.shape, dtype=np.uint8)
    tgt = np.minimum(tgt.astype(np.int32) + noise, 255).astype(np.uint8)
    return src, tgt


def both_spaces(x):
    return (x, torch.from_numpy(x).to(DEV))


def host(x):
    return x.cpu().numpy() if torch.is_tensor(x) else x


@pytest.mark.parametrize("shape", [(9, 13), (16, 32), (7, 11, 3), (5, 6, 2)])
def test_unfold_and_census_of_bytes(shape):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    f = img.astype(np.float32)
    for x in both_spaces(img):
        assert np.array_equal(host(sv.unfold(2, 1, x)), so.unfold(f, 2, 1))
        assert np.array_equal(host(sv.unfold(1, 1, x, (0, 2, 1, 0))), so.unfold(f, 1, 1, (0, 2, 1, 0)))
        assert np.array_equal(host(sv.censusTransform2D(x, 3, 2)), so.census_transform(f, 3, 2))
    mask = np.array([[1, 1, 2], [3, 0, 2], [3, 4, 4]], np.int32)
    assert np.array_equal(sv.unfoldCompressed(sv.UnFoldCompressor(mask), img), so.unfold_compressed(f, mask))


@pytest.mark.parametrize("func", ["CENSUS", "SAD", "SSD", "CC"])
@pytest.mark.parametrize("colour", [False, True])
def test_cost_volumes_of_bytes(func, colour):
    src, tgt = u8_pair(21, 45, 3 if colour else None, seed=5)
    h_r, v_r, D = (4, 4, 12) if func == "CENSUS" else (2, 1, 9)
    fo, fs = getattr(so, func), getattr(MF, func)
    exp = so.unfold_cost_volume(fo, tgt.astype(np.float32), src.astype(np.float32), h_r, v_r, D)
    for l, r in zip(both_spaces(tgt), both_spaces(src)):
        got = host(sv.unfoldBasedCostVolume(fs, l, r, h_r, v_r, D))
        assert np.array_equal(got, exp)  # integer-valued sums below 2^24: exact
    exp_lr = so.unfold_cost_volume(fo, tgt.astype(np.float32), src.astype(np.float32), h_r, v_r, 7, so.LEFT_TO_RIGHT, -2)
    got_lr = sv.unfoldBasedCostVolume(fs, tgt, src, h_r, v_r, sv.searchOffset1(-2, 4), sv.dispDirection.LeftToRight)
    assert np.array_equal(got_lr, exp_lr)
    exp2 = so.unfold_cost_volume_2d(fo, tgt.astype(np.float32), src.astype(np.float32), h_r, v_r, (-1, 2), (-2, 3))
    got2 = host(sv.unfoldBased2dDisparityCostVolume(fs, torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV), h_r, v_r, sv.searchOffset2(-1, 2, -2, 3)))
    assert np.array_equal(got2, exp2)


def test_mixed_float_and_byte_images():
    """T_L and T_R are separate template parameters in the reference (cross_correlations.h:740): one image of each type."""
    src, tgt = u8_pair(15, 30, seed=8)
    exp = so.unfold_cost_volume(so.SAD, tgt.astype(np.float32), src.astype(np.float32), 2, 2, 8)
    assert np.array_equal(sv.unfoldBasedCostVolume(MF.SAD, tgt, src.astype(np.float32), 2, 2, 8), exp)
    assert np.array_equal(sv.unfoldBasedCostVolume(MF.SAD, tgt.astype(np.float32), src, 2, 2, 8), exp)


def test_strided_and_misaligned_byte_views():
    """Device views that are not dense (one channel of an RGB image, a column window starting at an odd byte)."""
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (19, 37, 3), dtype=np.uint8)
    d = torch.from_numpy(rgb).to(DEV)
    for ch in range(3):
        exp = so.census_transform(rgb[:, :, ch].astype(np.float32), 2, 2)
        assert np.array_equal(host(sv.censusTransform2D(d[:, :, ch], 2, 2)), exp)
        assert np.array_equal(sv.censusTransform2D(rgb[:, :, ch], 2, 2), exp)
    flat = torch.from_numpy(rng.integers(0, 256, 1 + 13 * 17, dtype=np.uint8)).to(DEV)
    view = flat[1:].reshape(13, 17)  # dense, but starts one byte into a dword
    exp = so.unfold(view.cpu().numpy().astype(np.float32), 1, 1)
    assert np.array_equal(host(sv.unfold(1, 1, view)), exp)


@pytest.mark.parametrize("sgm", [0, 4, 8])
def test_fused_pipeline_of_bytes(sgm):
    src, tgt = u8_pair(40, 96, seed=11, shift=5)
    D = 20
    cv = so.unfold_cost_volume(so.CENSUS, tgt.astype(np.float32), src.astype(np.float32), 4, 4, D)
    vol = so.sgm(cv, sgm, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0) if sgm else cv
    idx = so.extract_index(vol, so.COST)
    ref = so.refine_disp(so.truncated_cost_volume(vol, idx, 4, 4, 1, so.TCV_SAME), idx, so.PARABOLA)
    for l, r in zip(both_spaces(tgt), both_spaces(src)):
        res = sv.stereoMatch(MF.CENSUS, l, r, 4, 4, D, sgmDirections=sgm, Pout=100.0, refineKernel=sv.InterpolationKernel.Parabola,
                             refine_h_radius=4, refine_v_radius=4, want_cv=True)
        assert np.array_equal(host(res["cv"]), cv)
        assert np.array_equal(host(res["disp"]), idx)
        got = host(res["refined"])
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        ok = ~np.isnan(ref)
        assert np.max(np.abs(got[ok] - ref[ok])) <= 1e-4
    # the shard calls take the same images
    keys = [sv.censusShardKeys(torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV), 4, 4, D, (b, c), sgmDirections=sgm) for b, c in ((0, 9), (9, 11))]
    merged = torch.minimum(keys[0], keys[1])
    out = sv.censusShardFinish(torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV), merged, 4, 4, D, sgmDirections=sgm)
    assert np.array_equal(host(out["disp"]), idx)
    # SAD through the generic pipeline
    cvs = so.unfold_cost_volume(so.SAD, tgt.astype(np.float32), src.astype(np.float32), 2, 2, D)
    res = sv.stereoMatch(MF.SAD, tgt, src, 2, 2, D, sgmDirections=0)
    assert np.array_equal(res["disp"], so.extract_index(cvs, so.COST))


@pytest.mark.parametrize("func", ["NCC", "ZNCC", "ZSAD", "ZSSD", "ZCC"])
def test_normalised_functions_refuse_bytes(func):
    src, tgt = u8_pair(12, 20, seed=2)
    with pytest.raises(sv._capi.SvhError) as e:
        sv.unfoldBasedCostVolume(getattr(MF, func), tgt, src, 2, 2, 6)
    assert e.value.status == sv._capi.ERR_UNSUPPORTED and "uint8" in str(e.value)
    with pytest.raises(sv._capi.SvhError):
        sv.stereoMatch(getattr(MF, func), torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV), 2, 2, 6)
    # the same images as float32 are fine
    assert sv.unfoldBasedCostVolume(getattr(MF, func), tgt.astype(np.float32), src.astype(np.float32), 2, 2, 6).shape == (12, 20, 6)

"""The engines of census_sweep -- the vector-ALU kernel (xor + popcount) and the two matrix-core kernels with FP4 operands (Hamming
distance as a dot product: svh_census_sweep_pm.hip for any geometry, svh_census_sweep_rl.hip for RightToLeft ranges that end at the
image edge with a multiple of 32 from 64 to 512 disparities) -- must give the same regional winner keys bit for bit, and all the oracle's
disparities.  svh_context_set_option("census_sweep", 1 | 3) selects the engine and "census_sweep_rl" 0 keeps the FP4 engine on its
general kernel; 0 (default) takes the matrix cores wherever they apply (1..8 census words; the leading multiple of 32 disparities up to
992) and the vector ALU elsewhere."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import libstevi_amd as sv  # noqa: E402
import oracle as so  # noqa: E402
from helpers import parallax_pair  # noqa: E402

MF = sv.matchingFunctions
DEV = torch.device("cuda:0")


def both_engines(fn, probe):
    """vector ALU, FP4 matrix cores (general kernel), FP4 with its RightToLeft specialisation allowed
    (svh_census_sweep_rl.hip: taken where the geometry fits, the general kernel elsewhere)"""
    out = []
    try:
        for mode, rl in ((1, 1), (3, 0), (3, 1)):
            sv.set_option(probe, "census_sweep", mode)
            sv.set_test_option(probe, "census_sweep_rl", rl)
            out.append(fn())
    finally:
        sv.set_option(probe, "census_sweep", 0)
        sv.set_test_option(probe, "census_sweep_rl", 1)
    return out


def oracle_disp(tgt, src, h_r, D, n_dir, Pout, margins=(0, 0, 0, 0), lower=0):
    cv = so.unfold_cost_volume(so.CENSUS, tgt, src, h_r, h_r, D, so.RIGHT_TO_LEFT, lower)
    vol = so.sgm(cv, n_dir, so.COST, 0.001, 0.01, margins, Pout) if n_dir else cv
    return so.extract_index(vol, so.COST)


@pytest.mark.parametrize("D", [32, 64, 96, 128, 160, 192, 224, 256, 288, 320, 352, 384, 416, 448, 480, 512, 992])
@pytest.mark.parametrize("W", [97, 384, 700, 1100])
def test_keys_and_disparities_agree(D, W):
    src, tgt, _ = parallax_pair(11, W, 9, 3, min(40, W // 3), 2, 13, seed=D + W)
    l, r = torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV)
    for h_r, n_dir, Pout in ((4, 8, 100.0), (3, 4, 7.0), (5, 8, 100.0), (4, 0, 100.0)):  # 2, 1 and 3 census words
        keys = both_engines(lambda: sv.censusShardKeys(l, r, h_r, h_r, D, (0, D), sgmDirections=n_dir, Pout=Pout).cpu().numpy(), l)
        assert all(np.array_equal(keys[0], k) for k in keys[1:]), (h_r, n_dir)
        disp = both_engines(lambda: sv.stereoMatch(MF.CENSUS, l, r, h_r, h_r, D, sgmDirections=n_dir, Pout=Pout)["disp"].cpu().numpy(), l)
        assert all(np.array_equal(disp[0], d) for d in disp[1:])
        if W <= 384 and (D <= 480 or W < 100):
            assert np.array_equal(disp[-1], oracle_disp(tgt, src, h_r, D, n_dir, Pout))


@pytest.mark.parametrize("D", [32, 96, 256])
@pytest.mark.parametrize("W", [97, 420, 700])
def test_left_to_right(D, W):
    """LeftToRight: target column j - lower - d.  The FP4 kernel mirrors the window and the lanes; the Pout boundary then moves by
    two cells per lane, so up to three row tiles of a column tile are masked per cell."""
    src, tgt, _ = parallax_pair(11, W, 9, 3, min(40, W // 3), 2, 13, seed=3 * D + W)
    l, r = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)  # roles swapped: the left image is the source
    L2R = sv.dispDirection.LeftToRight
    for h_r, n_dir, Pout, rng_ in ((4, 8, 100.0, D), (3, 4, 7.0, D), (4, 8, 100.0, sv.searchOffset1(-17, D - 18))):
        keys = both_engines(lambda: sv.censusShardKeys(l, r, h_r, h_r, rng_, (0, D), dDir=L2R, sgmDirections=n_dir, Pout=Pout).cpu().numpy(), l)
        assert all(np.array_equal(keys[0], k) for k in keys[1:]), (h_r, n_dir)
        disp = both_engines(lambda: sv.stereoMatch(MF.CENSUS, l, r, h_r, h_r, rng_, dDir=L2R, sgmDirections=n_dir, Pout=Pout)["disp"].cpu().numpy(), l)
        assert all(np.array_equal(disp[0], d) for d in disp[1:])
        if W <= 420 and isinstance(rng_, int):
            cv = so.unfold_cost_volume(so.CENSUS, src, tgt, h_r, h_r, D, so.LEFT_TO_RIGHT)
            vol = so.sgm(cv, n_dir, so.COST, 0.001, 0.01, (0, 0, 0, 0), Pout) if n_dir else cv
            assert np.array_equal(disp[-1], so.index_to_disp(so.extract_index(vol, so.COST), so.LEFT_TO_RIGHT))
    # shards in this direction
    for shard in ((0, 32), (32, D - 32)) if D > 32 else ((0, 32),):
        keys = both_engines(lambda: sv.censusShardKeys(l, r, 4, 4, D, shard, dDir=L2R, sgmDirections=8).cpu().numpy(), l)
        assert all(np.array_equal(keys[0], k) for k in keys[1:]), shard


def test_four_census_words():
    """13 x 11 windows: 142 census bits = four written words (two MFMAs per tile with FP4 operands), the most the RightToLeft
    specialisation takes."""
    for W in (200, 700):
        src, tgt, _ = parallax_pair(15, W, 9, 3, 40, 2, 11, seed=31 + W)
        l, r = torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV)
        for D in (64, 160, 256, 512):
            keys = both_engines(lambda: sv.censusShardKeys(l, r, 6, 5, D, (0, D), sgmDirections=8, Pout=100.0).cpu().numpy(), l)
            assert all(np.array_equal(keys[0], k) for k in keys[1:])
            disp = both_engines(lambda: sv.stereoMatch(MF.CENSUS, l, r, 6, 5, D, sgmDirections=8, Pout=100.0)["disp"].cpu().numpy(), l)
            assert all(np.array_equal(disp[0], d) for d in disp[1:])
            if W == 200 and D <= 160:
                cv = so.unfold_cost_volume(so.CENSUS, tgt, src, 6, 5, D)
                assert np.array_equal(disp[-1], so.extract_index(so.sgm(cv, 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0), so.COST))


def test_ties_everywhere():
    """Constant and two-level images: most costs are equal, so the winner is decided by the tie rule alone (the last index,
    correlation_base.h:441-455) -- within a tile, across row tiles, across the two halves of a wave and across the Pout boundary."""
    rng = np.random.default_rng(4)
    H, W, D = 9, 333, 96
    flat = np.full((H, W), 0.25, np.float32)
    steps = (rng.integers(0, 2, (H, W)) * 0.5).astype(np.float32)
    blocks = np.repeat(rng.integers(0, 2, (H, W // 9 + 1)), 9, axis=1)[:, :W].astype(np.float32)
    for tgt, src in ((flat, flat), (steps, np.roll(steps, 5, axis=1)), (blocks, blocks), (flat, steps)):
        l, r = torch.from_numpy(tgt.copy()).to(DEV), torch.from_numpy(src.copy()).to(DEV)
        for n_dir in (0, 8):
            keys = both_engines(lambda: sv.censusShardKeys(l, r, 4, 4, D, (0, D), sgmDirections=n_dir).cpu().numpy(), l)
            assert all(np.array_equal(keys[0], k) for k in keys[1:])
            disp = both_engines(lambda: sv.stereoMatch(MF.CENSUS, l, r, 4, 4, D, sgmDirections=n_dir)["disp"].cpu().numpy(), l)
            assert all(np.array_equal(disp[0], d) for d in disp[1:])
            assert np.array_equal(disp[1], oracle_disp(tgt, src, 4, D, n_dir, 100.0))


def test_shards_offsets_and_margins():
    """Disparity shards (global indices in the keys, the Pout boundary shifted by the shard offset), a search range that starts
    below zero, and margins: the boundary then runs through the middle of tiles."""
    src, tgt, _ = parallax_pair(13, 420, 10, 4, 60, 3, 17, seed=77)
    l, r = torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV)
    D = 192
    for shard in ((0, 64), (64, 96), (160, 32)):
        keys = both_engines(lambda: sv.censusShardKeys(l, r, 4, 4, D, shard, sgmDirections=8, Pout=100.0).cpu().numpy(), l)
        assert all(np.array_equal(keys[0], k) for k in keys[1:]), shard
    for rng_ in (sv.searchOffset1(-40, 87), sv.searchOffset1(5, 68)):
        keys = both_engines(lambda: sv.censusShardKeys(l, r, 4, 4, rng_, (0, 128 if rng_.lower < 0 else 64), sgmDirections=8).cpu().numpy(), l)
        assert all(np.array_equal(keys[0], k) for k in keys[1:])
    for margins in ((3, 2, 5, 1), (0, 0, 17, 0)):
        res = both_engines(lambda: sv.stereoMatch(MF.CENSUS, l, r, 4, 4, D, sgmDirections=8, Pout=9.0, margins=sv.Margins(*margins),
                                                  refineKernel=sv.InterpolationKernel.Parabola, refine_h_radius=4, refine_v_radius=4), l)
        for other in res[1:]:
            assert torch.equal(res[0]["disp"], other["disp"])
            a, b = res[0]["refined"].cpu().numpy(), other["refined"].cpu().numpy()
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
        assert np.array_equal(res[1]["disp"].cpu().numpy(), oracle_disp(tgt, src, 4, D, 8, 9.0, margins))


def test_geometries_outside_the_matrix_core_kernel_still_run():
    """D not a multiple of 32 and D > 992 (the range is split between the engines), 13x13 windows (5 words: the general FP4 kernel);
    LeftToRight stays on the general FP4 kernel."""
    src, tgt, _ = parallax_pair(9, 260, 8, 3, 30, 2, 9, seed=5)
    l, r = torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV)
    for kw in (dict(D=64, dDir=sv.dispDirection.LeftToRight, h=4), dict(D=70, dDir=sv.dispDirection.RightToLeft, h=4),
               dict(D=1024, dDir=sv.dispDirection.RightToLeft, h=4), dict(D=64, dDir=sv.dispDirection.RightToLeft, h=6)):
        disp = both_engines(lambda: sv.stereoMatch(MF.CENSUS, l, r, kw["h"], kw["h"], kw["D"], dDir=kw["dDir"], sgmDirections=8)["disp"].cpu().numpy(), l)
        assert all(np.array_equal(disp[0], d) for d in disp[1:])
    for bad in (2, 4):  # (2 was round 1's int8 matrix-core engine: removed)
        with pytest.raises(sv._capi.SvhError):
            sv.set_option(l, "census_sweep", bad)


@pytest.mark.parametrize("D", [33, 50, 100, 250, 333])
@pytest.mark.parametrize("W", [97, 420])
def test_ranges_that_are_no_multiple_of_32(D, W):
    """RightToLeft ranges that end at the image edge: one launch of the RAGGED form of the specialised kernel (the cells past the range are
    masked by the start patterns of the last two row tiles, round 5).  Other geometries: the matrix cores take the leading multiple of 32
    disparities, the vector ALU the rest as a second disparity shard whose keys are MIN-merged into the first part's (launch_sweep).  The
    same keys as the vector ALU alone over the whole range, both directions, whole ranges, shards and ranges that start below zero; the
    oracle's disparities."""
    src, tgt, _ = parallax_pair(11, W, 9, 3, min(40, W // 3), 2, 13, seed=5 * D + W)
    l, r = torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV)
    for h_r, n_dir, Pout in ((4, 8, 100.0), (3, 4, 7.0), (5, 8, 100.0)):
        for shard in ((0, D), (7, D - 7)):
            keys = both_engines(lambda: sv.censusShardKeys(l, r, h_r, h_r, D, shard, sgmDirections=n_dir, Pout=Pout).cpu().numpy(), l)
            assert all(np.array_equal(keys[0], k) for k in keys[1:]), (h_r, n_dir, shard)
        res = both_engines(lambda: sv.stereoMatch(MF.CENSUS, l, r, h_r, h_r, D, sgmDirections=n_dir, Pout=Pout, want_sgm_cv=D <= 100), l)
        assert all(torch.equal(res[0]["disp"], o["disp"]) for o in res[1:])
        if W <= 97 or D <= 100:
            cv = so.unfold_cost_volume(so.CENSUS, tgt, src, h_r, h_r, D, so.RIGHT_TO_LEFT, 0)
            vol = so.sgm(cv, n_dir, so.COST, 0.001, 0.01, (0, 0, 0, 0), Pout)
            assert np.array_equal(res[-1]["disp"].cpu().numpy(), so.extract_index(vol, so.COST))
            if D <= 100:
                assert np.array_equal(res[-1]["sgm_cv"].cpu().numpy().view(np.uint32), vol.view(np.uint32))
    rng_ = sv.searchOffset1(-17, D - 18)
    keys = both_engines(lambda: sv.censusShardKeys(l, r, 4, 4, rng_, (0, D), sgmDirections=8).cpu().numpy(), l)
    assert all(np.array_equal(keys[0], k) for k in keys[1:])
    # LeftToRight (roles swapped: the left image is the source)
    L2R = sv.dispDirection.LeftToRight
    keys = both_engines(lambda: sv.censusShardKeys(r, l, 4, 4, D, (0, D), dDir=L2R, sgmDirections=8).cpu().numpy(), l)
    assert all(np.array_equal(keys[0], k) for k in keys[1:])
    disp = both_engines(lambda: sv.stereoMatch(MF.CENSUS, r, l, 4, 4, D, dDir=L2R, sgmDirections=8)["disp"].cpu().numpy(), l)
    assert all(np.array_equal(disp[0], d) for d in disp[1:])
    if W <= 97 or D <= 100:
        cv = so.unfold_cost_volume(so.CENSUS, src, tgt, 4, 4, D, so.LEFT_TO_RIGHT)
        vol = so.sgm(cv, 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0)
        assert np.array_equal(disp[-1], so.index_to_disp(so.extract_index(vol, so.COST), so.LEFT_TO_RIGHT))


@pytest.mark.parametrize("h_r,v_r", [(6, 6), (7, 6), (7, 7), (9, 7)])
@pytest.mark.parametrize("W", [97, 420, 700])
def test_five_to_eight_census_words(h_r, v_r, W):
    """13x13 ... 19x15 windows (5, 6, 7, 8 census words: three or four MFMAs per tile, costs up to 256): the matrix-core kernel gives the
    vector-ALU kernel's keys, whole ranges and shards, both directions, ranges that are no multiple of 32; the oracle's disparities."""
    src, tgt, _ = parallax_pair(23, W, 9, 3, min(40, W // 3), 2, 13, seed=h_r * 100 + v_r + W)
    l, r = torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV)
    for D, n_dir, Pout in ((64, 8, 100.0), (96, 4, 7.0), (256, 8, 100.0), (70, 8, 100.0)):
        for shard in ((0, D), (32, D - 32)):
            keys = both_engines(lambda: sv.censusShardKeys(l, r, h_r, v_r, D, shard, sgmDirections=n_dir, Pout=Pout).cpu().numpy(), l)
            assert all(np.array_equal(keys[0], k) for k in keys[1:]), (D, n_dir, shard)
        disp = both_engines(lambda: sv.stereoMatch(MF.CENSUS, l, r, h_r, v_r, D, sgmDirections=n_dir, Pout=Pout)["disp"].cpu().numpy(), l)
        assert all(np.array_equal(disp[0], d) for d in disp[1:])
        if W <= 420 and D <= 96:
            cv = so.unfold_cost_volume(so.CENSUS, tgt, src, h_r, v_r, D, so.RIGHT_TO_LEFT, 0)
            vol = so.sgm(cv, n_dir, so.COST, 0.001, 0.01, (0, 0, 0, 0), Pout)
            assert np.array_equal(disp[-1], so.extract_index(vol, so.COST))
    L2R = sv.dispDirection.LeftToRight
    keys = both_engines(lambda: sv.censusShardKeys(r, l, h_r, v_r, 64, (0, 64), dDir=L2R, sgmDirections=8).cpu().numpy(), l)
    assert all(np.array_equal(keys[0], k) for k in keys[1:])

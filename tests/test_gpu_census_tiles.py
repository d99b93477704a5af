"""census + SGM with the line recurrences run and their values used (reduction keys, Gaussian refinement taps): the form that keeps
only the carries of the line scans and replays the lines per tile inside the per-pixel kernel (option "census_tiles", the default)
against the form that writes the six min_p maps (option 0), the general wave-per-line kernels and the oracle -- the disparity map
does not depend on the sum of the min_p (the identity of sgm.h:257-296), so these tests compare what does: the S value at the
winner (keys) and the refined map.  sgm.h:186-311, :329-354 (pass coverage, the diagonal both start loops run)."""
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

DEV = torch.device("cuda:0")


def run(d_tgt, d_src, r, D, tiles, fast=1, **kw):
    sv.set_test_option(d_tgt, "census_tiles", tiles)
    sv.set_test_option(d_tgt, "census_fast_path", fast)
    sv.set_option(d_tgt, "census_winner_shortcut", 0)
    try:
        return sv.stereoMatch(MF.CENSUS, d_tgt, d_src, r, r, D, sgmDirections=8, P1=0.3, P2=0.9, want_keys=True, refineKernel=so.GAUSSIAN,
                              refine_h_radius=r, refine_v_radius=r, **kw)
    finally:
        sv.set_test_option(d_tgt, "census_tiles", 1)
        sv.set_test_option(d_tgt, "census_fast_path", 1)
        sv.set_option(d_tgt, "census_winner_shortcut", 1)


def same(a, b, what):
    for k in ("disp", "keys"):
        if not torch.equal(a[k], b[k]):
            bad = (a[k] != b[k]).nonzero().cpu().numpy()
            raise AssertionError((what, k, len(bad), bad[:12].tolist()))
    ra, rb = a["refined"].cpu().numpy(), b["refined"].cpu().numpy()
    assert np.array_equal(np.isnan(ra), np.isnan(rb)), what
    ok = ~np.isnan(ra)
    assert np.array_equal(ra[ok].view(np.uint32), rb[ok].view(np.uint32)), what


# taller than wide, wider than tall, one row, one column, fewer columns than a tile, rows that are no multiple of the segment length,
# more rows than columns by more than a tile (diagonals that never reach the right edge), square (the corner diagonal)
SHAPES = [(37, 90), (90, 37), (64, 64), (1, 70), (70, 1), (130, 200), (300, 70), (33, 33), (200, 130), (17, 400), (256, 256), (65, 129)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("D,r", [(32, 4), (64, 3), (40, 4)])
def test_tiles_equal_maps(shape, D, r):
    H, W = shape
    rng = np.random.default_rng(H * 1000 + W + D)
    src = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    tgt = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    for Pout in (100.0, 0.0, 7.0):
        a = run(d_tgt, d_src, r, D, 1, Pout=Pout)
        b = run(d_tgt, d_src, r, D, 0, Pout=Pout)
        same(a, b, (shape, D, r, Pout))


@pytest.mark.parametrize("shape", [(37, 90), (90, 37), (1, 70), (70, 1), (130, 200), (64, 64)])
def test_tiles_equal_general_kernels_and_oracle(shape):
    H, W = shape
    D, r = 64, 4
    src, tgt, _ = parallax_pair(H, W, max(min(H, W) // 3, 1), H // 4, W // 4, 2, 9, seed=H + W)
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    a = run(d_tgt, d_src, r, D, 1, Pout=100.0)
    gen = run(d_tgt, d_src, r, D, 1, fast=0, Pout=100.0)
    same(a, gen, shape)
    cv = so.unfold_cost_volume(so.CENSUS, tgt, src, r, r, D, so.RIGHT_TO_LEFT)
    vol = so.sgm(cv, 8, so.COST, 0.3, 0.9, (0, 0, 0, 0), 100.0)
    idx = so.extract_index(vol, so.COST)
    assert np.array_equal(a["disp"].cpu().numpy(), so.index_to_disp(idx, so.RIGHT_TO_LEFT))
    # the S value at the winner, from the reduction key's upper half (an order-preserving code of the float)
    best = np.take_along_axis(vol, idx[..., None].astype(np.int64), axis=2)[..., 0]
    u = best.astype(np.float32).view(np.uint32)
    code = np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint64)
    keys = a["keys"].cpu().numpy().view(np.uint64)
    assert np.array_equal(keys >> np.uint64(32), code)
    exp = so.refine_disp(so.truncated_cost_volume(vol, idx, r, r, 1), idx, so.GAUSSIAN)
    got = a["refined"].cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    assert np.allclose(got[~np.isnan(exp)], exp[~np.isnan(exp)], atol=1e-6)


@pytest.mark.parametrize("r,ddir", [(2, "RightToLeft"), (4, "LeftToRight"), (5, "LeftToRight")])
def test_tiles_other_windows_and_direction(r, ddir):
    """5 x 5 windows (no census word is written: every cost is 0), 11 x 11 (three words), and the other disparity direction."""
    H, W, D = 75, 140, 64
    rng = np.random.default_rng(r)
    src = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    tgt = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    kw = dict(Pout=100.0, dDir=getattr(sv.dispDirection, ddir))
    same(run(d_tgt, d_src, r, D, 1, **kw), run(d_tgt, d_src, r, D, 0, **kw), (r, ddir))
    same(run(d_tgt, d_src, r, D, 1, **kw), run(d_tgt, d_src, r, D, 1, fast=0, **kw), (r, ddir, "general"))


def oracle_checks(out, vol, r, what):
    """The values the recurrences produce, against the oracle's aggregated volume: the disparity map, the S value at the winner (upper
    half of the reduction key: an order-preserving code of the float) and the Gaussian-refined map (it takes logarithms of the three
    taps, so it sees the sum of the per-pass minima that the disparity map cannot see)."""
    idx = so.extract_index(vol, so.COST)
    assert np.array_equal(out["disp"].cpu().numpy(), so.index_to_disp(idx, so.RIGHT_TO_LEFT)), what
    best = np.take_along_axis(vol, idx[..., None].astype(np.int64), axis=2)[..., 0]
    u = best.astype(np.float32).view(np.uint32)
    code = np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint64)
    keys = out["keys"].cpu().numpy().view(np.uint64)
    assert np.array_equal(keys >> np.uint64(32), code), what
    taps = so.truncated_cost_volume(vol, idx, r, r, 1)
    exp = so.refine_disp(taps, idx, so.GAUSSIAN)
    got = out["refined"].cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(exp)), what
    # Tolerance: the north star's 1e-4 for refined disparities.  (At 1080p the aggregated costs reach a few hundred and neighbouring taps
    # differ by a few units: the differences of logarithms in (ln cm1 - ln c1) / (2 (ln c1 - 2 ln c0 + ln cm1)) cancel most digits, so the
    # device's logf against the host's shows as a few 1e-5 here where the small frames above agree to 1e-6.)
    ok = ~np.isnan(exp)
    assert float(np.max(np.abs(got[ok] - exp[ok]), initial=0.0)) <= 1e-4, (what, float(np.max(np.abs(got[ok] - exp[ok]))))
    assert np.median(np.abs(got[ok] - exp[ok])) <= 1e-6, what


def test_tiles_full_hd_rows():
    """1080 rows: sixteen segments of 68 rows (the last one 60), thirty column tiles -- the benchmark's frame, 256 disparities.  Against the
    maps form AND against the oracle's aggregated volume of the whole frame (S at the winner of every pixel, Gaussian-refined map): the
    headline form is not only compared with another form of itself."""
    src, tgt, _ = parallax_pair(1080, 1920, 320, 320, 380, 8, 64, seed=3)
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    a = run(d_tgt, d_src, 4, 256, 1, Pout=100.0)
    b = run(d_tgt, d_src, 4, 256, 0, Pout=100.0)
    same(a, b, "1080p")
    cv = so.unfold_cost_volume(so.CENSUS, tgt, src, 4, 4, 256, so.RIGHT_TO_LEFT)
    vol = so.sgm(cv, 8, so.COST, 0.3, 0.9, (0, 0, 0, 0), 100.0)
    del cv
    oracle_checks(a, vol, 4, "1080p x 256, tiles form")


# taller than one launch of the scans covers (1152 rows): bands of rows, a line's min_p handed from band to band (ScanBands).  Heights
# just past one band, two bands with a short second one, three and four bands, a last band of a single segment, wider than tall by far
TALL = [(1153, 70), (1300, 100), (2304, 64), (2305, 130), (2400, 257), (3500, 90), (4320, 200), (1160, 1500)]


@pytest.mark.parametrize("shape", TALL)
def test_tiles_tall_images_equal_maps(shape):
    H, W = shape
    D, r = 32, 4
    rng = np.random.default_rng(H * 7 + W)
    src = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    tgt = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    for Pout in (100.0, 3.0):
        a = run(d_tgt, d_src, r, D, 1, Pout=Pout)
        b = run(d_tgt, d_src, r, D, 0, Pout=Pout)
        same(a, b, (shape, Pout))


@pytest.mark.parametrize("shape", [(130, 200), (1300, 100), (2500, 70)])
def test_shard_finish_replays_tiles_like_the_single_call(shape):
    """svh_census_shard_finish with the recurrences' values in use (Gaussian refinement): tile-edge values + replay from reduced keys,
    against the six-map form and the one-GPU call."""
    H, W = shape
    D, r = 64, 4
    src, tgt, _ = parallax_pair(H, W, max(min(H, W) // 3, 1), H // 4, W // 4, 2, 9, seed=H + W)
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    kw = dict(sgmDirections=8, P1=0.3, P2=0.9, Pout=100.0)
    single = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, r, r, D, refineKernel=so.GAUSSIAN, refine_h_radius=r, refine_v_radius=r, **kw)
    keys = torch.minimum(sv.censusShardKeys(d_tgt, d_src, r, r, D, (0, 32), **kw), sv.censusShardKeys(d_tgt, d_src, r, r, D, (32, 32), **kw))
    outs = []
    for tiles in (1, 0):
        sv.set_test_option(d_tgt, "census_tiles", tiles)
        try:
            outs.append(sv.censusShardFinish(d_tgt, d_src, keys, r, r, D, refineKernel=so.GAUSSIAN, refine_h_radius=r, refine_v_radius=r, **kw))
        finally:
            sv.set_test_option(d_tgt, "census_tiles", 1)
    for o in outs:
        assert torch.equal(o["disp"], single["disp"])
        ra, rb = o["refined"].cpu().numpy(), single["refined"].cpu().numpy()
        assert np.array_equal(np.isnan(ra), np.isnan(rb))
        ok = ~np.isnan(ra)
        assert np.array_equal(ra[ok].view(np.uint32), rb[ok].view(np.uint32))

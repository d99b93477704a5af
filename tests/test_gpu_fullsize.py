"""Full-size checks at BASELINE.json's shapes: the oracle itself where its O(D) SGM restatement (bitwise identical to the
literal loops, tests/test_oracle_semantics.py) finishes in seconds -- the whole C3 frame, row bands of C4 and C5 -- and
size-independent properties elsewhere: independent numpy / torch restatements of single stages, the Cost-branch identity of
SURVEY.md section 8(a), planted-disparity recovery, fast path == general kernels, shards == single call."""
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


def np_census_9x9(img):
    """numpy restatement of census 9x9 (top-left reference sample, 2 written words)."""
    H, W = img.shape
    pad = np.zeros((H + 8, W + 8), np.float32)
    pad[4:-4, 4:-4] = img
    ref = pad[0:H, 0:W]
    words = np.zeros((H, W, 3), np.uint32)
    c = 0
    for k in range(9):
        for l in range(9):
            if c >= 1 and c <= 64:
                bit = (ref > pad[k:k + H, l:l + W]).astype(np.uint32)
                words[:, :, (c - 1) // 32] |= bit << np.uint32((c - 1) % 32)
            c += 1
    return words


@pytest.fixture(scope="module")
def c3_pair():
    src, tgt, gt = parallax_pair(1080, 1920, 320, 320, 380, 8, 64, seed=3)
    return src, tgt, gt


def test_c2_census_volume_1080p(c3_pair):
    src, tgt, _ = c3_pair
    D = 128
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    words = sv.censusTransform2D(d_src, 4, 4).cpu().numpy()
    assert np.array_equal(words, np_census_9x9(src))
    cv = sv.unfoldBasedCostVolume(MF.CENSUS, d_tgt, d_src, 4, 4, D)
    assert tuple(cv.shape) == (1080, 1920, D)
    # sampled voxels against the numpy words (target rounded through float: rule E2)
    wt = np_census_9x9(tgt)
    rt = np.vectorize(so.round_word_through_float, otypes=[np.uint32])
    rng = np.random.default_rng(0)
    ii, jj, dd = rng.integers(0, 1080, 4000), rng.integers(0, 1920, 4000), rng.integers(0, D, 4000)
    got = cv[torch.from_numpy(ii).to(DEV), torch.from_numpy(jj).to(DEV), torch.from_numpy(dd).to(DEV)].cpu().numpy()
    jt = jj + dd
    inside = jt < 1920
    t = np.where(inside[:, None], rt(wt[ii, np.minimum(jt, 1919)]), 0).astype(np.uint32)
    x = src_words = np_census_9x9(src)[ii, jj] ^ t
    exp = np.array([bin(int(a)).count("1") + bin(int(b)).count("1") for a, b, _ in x], np.float32)
    assert np.array_equal(got, exp)
    # a band of rows against the oracle, bit for bit
    band = so.unfold_cost_volume(so.CENSUS, tgt[:40], src[:40], 4, 4, D)
    assert np.array_equal(cv[:36].cpu().numpy(), band[:36])


def test_c3_census_sgm_1080p(c3_pair):
    """Config 3 at full size: fused pipeline == general kernels == the Cost-branch identity; planted disparities win."""
    src, tgt, gt = c3_pair
    D, Pout = 256, 100.0
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    fast = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, 4, 4, D, sgmDirections=8, Pout=Pout)["disp"]
    try:
        sv.set_test_option(d_tgt, "census_fast_path", 0)
        gen = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, 4, 4, D, sgmDirections=8, Pout=Pout, want_cv=True, want_sgm_cv=True)
    finally:
        sv.set_test_option(d_tgt, "census_fast_path", 1)
    assert torch.equal(fast, gen["disp"])
    # identity (integer costs): S - [(1+n) C + n Pout oob] is the same for every d; its argmin is the argmin of S
    H, W = 1080, 1920
    ii = torch.arange(H, device=DEV)[:, None]
    jj = torch.arange(W, device=DEV)[None, :]
    n = 2 + 1 + (ii == jj).int() + (ii + jj < W).int() + (ii + jj < H).int()
    oob = (jj[:, :, None] + torch.arange(D, device=DEV)[None, None, :] >= W).float()
    t = (1 + n)[:, :, None] * gen["cv"] + n[:, :, None] * Pout * oob
    resid = gen["sgm_cv"] - t
    assert torch.equal(resid, resid[:, :, :1].expand_as(resid))
    del resid
    idx_t = sv.extractSelectedIndex(so.COST, t)
    assert torch.equal(idx_t, gen["disp"])
    # the whole frame against the oracle: cost volume band, then the reference chain's disparity map, pixel for pixel
    # (benchmarkCrossCorrelationAlgorithms.cpp:288-294; bench.py reports the same comparison as end_px_err)
    band = 24
    cvb = so.unfold_cost_volume(so.CENSUS, tgt[:band + 4], src[:band + 4], 4, 4, D)
    assert np.array_equal(gen["cv"][:band].cpu().numpy(), cvb[:band])
    del gen, t
    cv = so.unfold_cost_volume(so.CENSUS, tgt, src, 4, 4, D)
    vol = so.sgm(cv, 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), Pout)
    want = so.index_to_disp(so.extract_index(vol, so.COST))
    del cv
    assert int(np.count_nonzero(fast.cpu().numpy() != want)) == 0
    # the form bench.py times (recurrences run, tile edges + per-tile replay) asked for what depends on the recurrences' VALUES, against
    # the oracle's volume of the whole frame: S at the winner of every pixel (reduction keys) and the Gaussian-refined map
    from test_gpu_census_tiles import oracle_checks
    try:
        sv.set_option(d_tgt, "census_winner_shortcut", 0)
        tiles = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, 4, 4, D, sgmDirections=8, Pout=Pout, want_keys=True, refineKernel=so.GAUSSIAN, refine_h_radius=4,
                               refine_v_radius=4)
    finally:
        sv.set_option(d_tgt, "census_winner_shortcut", 1)
    assert torch.equal(tiles["disp"], fast)
    oracle_checks(tiles, vol, 4, "C3, tiles form")
    del vol, tiles
    # planted disparities are recovered away from borders and occlusions
    disp = fast.cpu().numpy()
    inner = np.zeros((H, W), bool)
    inner[8:-8, 8:W - 256 - 8] = True
    inner[300:660, 300:800] = False
    # (only 64 of the 80 comparisons survive, against the window's top-left sample, and the target words lose their low
    # bits in the float round trip: the reference's census is a weak matcher by construction -- this is a sanity bound)
    assert (disp[inner] == gt[inner]).mean() > 0.6


@pytest.mark.parametrize("h_r,D", [(4, 250), (4, 1000), (6, 256), (7, 100)])
def test_census_sgm_1080p_ranges_and_windows_off_the_benchmark_shape(c3_pair, h_r, D):
    """1080p with disparity counts that are no multiple of 32 (the range split between the matrix cores and the vector ALU), beyond 992,
    and windows of five / seven census words (pixel-per-lane kernels with long records): the winner-identity form, the form with the
    recurrences run and the vector-ALU sweep alone give one disparity map and the same reduction keys; where the oracle finishes in seconds
    (D <= 256) the map, S at the winner and the Gaussian-refined map agree with its aggregated volume of the whole frame."""
    src, tgt, _ = c3_pair
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    fast = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, h_r, h_r, D, sgmDirections=8)["disp"]
    try:
        sv.set_option(d_tgt, "census_winner_shortcut", 0)
        run = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, h_r, h_r, D, sgmDirections=8, want_keys=True, refineKernel=so.GAUSSIAN, refine_h_radius=h_r,
                             refine_v_radius=h_r)
        sv.set_option(d_tgt, "census_sweep", 1)  # the vector ALU alone over the whole range
        valu = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, h_r, h_r, D, sgmDirections=8, want_keys=True)
    finally:
        sv.set_option(d_tgt, "census_winner_shortcut", 1)
        sv.set_option(d_tgt, "census_sweep", 0)
    assert torch.equal(run["disp"], fast) and torch.equal(valu["disp"], fast)
    assert torch.equal(run["keys"], valu["keys"])
    if D <= 256:
        from test_gpu_census_tiles import oracle_checks
        cv = so.unfold_cost_volume(so.CENSUS, tgt, src, h_r, h_r, D)
        vol = so.sgm(cv, 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0)
        del cv
        assert int(np.count_nonzero(fast.cpu().numpy() != so.index_to_disp(so.extract_index(vol, so.COST)))) == 0
        oracle_checks(run, vol, h_r, f"1080p, {2 * h_r + 1} wide, D = {D}")


def test_c1_sad_480x360_full_vs_oracle():
    """Config 1 (480x360, SAD 5x5, D=64) is small enough to run the oracle in full."""
    src, tgt, gt = parallax_pair(360, 480, 120, 120, 120, 4, 24, seed=1)
    cv = so.unfold_cost_volume(so.SAD, tgt, src, 2, 2, 64)
    res = sv.stereoMatch(MF.SAD, torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV), 2, 2, 64, want_cv=True)
    got = res["cv"].cpu().numpy()
    assert np.max(np.abs(got - cv) / np.maximum(1, np.abs(cv))) <= 1e-4
    assert np.array_equal(res["disp"].cpu().numpy(), so.index_to_disp(so.extract_index(got, so.COST)))
    assert (res["disp"].cpu().numpy() == so.index_to_disp(so.extract_index(cv, so.COST))).mean() > 0.999


# ---- row bands of the big configurations against the oracle ---------------------------------------------------------------
def band_mask(H, W, band):
    """Pixels of the top `band` rows whose SGM value is the same whether the oracle sees the whole frame or only its top rows:
    every pass but DownLeft2UpRight reaches a pixel from above or from the left; that pass runs up the anti-diagonals from the
    left column, so a pixel keeps its value when its anti-diagonal starts inside the band (i + j < band) or is
    not traversed at all (i + j >= H)."""
    ii, jj = np.meshgrid(np.arange(band), np.arange(W), indexing="ij")
    return (ii + jj < band) | (ii + jj >= H)


def test_c5_shards_8k_fullsize():
    """Config 5's geometry on one GPU: 8192x4320, D = 512 as 8 shards of 64 -> keys -> element-wise min (what the RCCL
    all-reduce computes) -> finish.  Equal to the single call, to the general (non-exact-regime) kernels, and to the oracle on
    a row band.  First run of the exact-regime bound 8 (2 cmax + Pout) (L + 2) < 2^24 at L = 8192 under test."""
    H, W, D, band = 4320, 8192, 512, 40
    src, tgt, _ = parallax_pair(H, W, 1280, 1280, 1520, 32, 256, seed=5)
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    kw = dict(sgmDirections=8, P1=0.001, P2=0.01, Pout=100.0)
    single = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, 4, 4, D, **kw)["disp"]
    assert sv.censusShardRegion1IsGlobal(d_tgt, d_src, D)
    keys = None
    for r in range(8):
        k = sv.censusShardKeys(d_tgt, d_src, 4, 4, D, (64 * r, 64), **kw)
        keys = k if keys is None else torch.minimum(keys, k)
        del k
    sharded = sv.censusShardFinish(d_tgt, d_src, keys, 4, 4, D, **kw)["disp"]
    assert torch.equal(sharded, single)
    del keys, sharded
    try:  # the general wave-per-line kernels + literal float evaluation per voxel
        sv.set_test_option(d_tgt, "census_fast_path", 0)
        gen = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, 4, 4, D, **kw)["disp"]
    finally:
        sv.set_test_option(d_tgt, "census_fast_path", 1)
    assert torch.equal(gen, single)
    del gen
    try:  # the vector-ALU sweep
        sv.set_option(d_tgt, "census_sweep", 1)
        valu = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, 4, 4, D, **kw)["disp"]
    finally:
        sv.set_option(d_tgt, "census_sweep", 0)
    assert torch.equal(valu, single)
    del valu
    cv = so.unfold_cost_volume(so.CENSUS, tgt[:band + 4], src[:band + 4], 4, 4, D)
    want = so.index_to_disp(so.extract_index(so.sgm(cv, 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0), so.COST))[:band]
    m = band_mask(H, W, band)
    assert m.mean() > 0.4
    assert np.array_equal(single[:band].cpu().numpy()[m], want[m])


@pytest.mark.parametrize("W", [4096, 8192])
def test_census_sgm_wide_short_images_vs_oracle(W):
    """The fused census + SGM path at C4 / C5 widths (row scans wider than 2048 pixels, the exact-regime bound near its limit),
    on an image short enough for the oracle to run whole; every sweep engine (vector ALU, the general and the RightToLeft FP4 kernels)."""
    H, D = 36, 256
    src, tgt, _ = parallax_pair(H, W, 20, 8, 100, 16, 130, seed=W)
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    cv = so.unfold_cost_volume(so.CENSUS, tgt, src, 4, 4, D)
    want = so.index_to_disp(so.extract_index(so.sgm(cv, 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0), so.COST))
    for engine, rl in ((0, 1), (1, 1), (3, 0)):
        try:
            sv.set_option(d_tgt, "census_sweep", engine)
            sv.set_test_option(d_tgt, "census_sweep_rl", rl)
            got = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, 4, 4, D, sgmDirections=8, Pout=100.0)["disp"].cpu().numpy()
        finally:
            sv.set_option(d_tgt, "census_sweep", 0)
            sv.set_test_option(d_tgt, "census_sweep_rl", 1)
        assert np.array_equal(got, want), (engine, rl)


def test_shards_beyond_1024_disparities_vs_oracle():
    """svh_census_shard_finish on a range wider than one launch may sweep (4 shards of 512 = 2048 disparities; the keys hold up to
    4096): the finish step only sees per-pixel maps and must not apply the per-shard limit."""
    H, W, D = 20, 2200, 2048
    src, tgt, _ = parallax_pair(H, W, 10, 4, 60, 9, 700, seed=77)
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    kw = dict(sgmDirections=8, P1=0.001, P2=0.01, Pout=100.0)
    keys = None
    for b in range(0, D, 512):
        k = sv.censusShardKeys(d_tgt, d_src, 4, 4, D, (b, 512), **kw)
        keys = k if keys is None else torch.minimum(keys, k)
    got = sv.censusShardFinish(d_tgt, d_src, keys, 4, 4, D, **kw)["disp"].cpu().numpy()
    cv = so.unfold_cost_volume(so.CENSUS, tgt, src, 4, 4, D)
    want = so.index_to_disp(so.extract_index(so.sgm(cv, 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0), so.COST))
    assert np.array_equal(got, want)


def test_c4_ncc_sgm_refine_4k_fullsize():
    """Config 4 at full size: 4096x2160, NCC 11x11, D = 256, SGM-8 (Score branch), arg-max, truncatedCostVolume(S, idx, 5, 5, 1) +
    parabola (rule E6).  Cost volume band vs the oracle <= 1e-4 (north-star tolerance for float costs); the Score-branch SGM of
    the GPU's own band bit-exact vs the oracle's SGM of that band; refined map of the band <= 1e-4 with the same NaN mask
    (cost_based_refinement.h:128-163, correlation_base.h:606-609)."""
    H, W, D, band, hr = 2160, 4096, 256, 28, 5
    src, tgt, _ = parallax_pair(H, W, 640, 640, 760, 16, 128, seed=4)
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    res = sv.stereoMatch(MF.NCC, d_tgt, d_src, hr, hr, D, sgmDirections=8, P1=0.001, P2=0.01, Pout=100.0, refineKernel=so.PARABOLA,
                         refine_h_radius=hr, refine_v_radius=hr, want_cv=True, want_sgm_cv=True)
    torch.cuda.synchronize()
    gcv = res["cv"][:band].cpu().numpy()
    cvb = so.unfold_cost_volume(so.NCC, tgt[:band + hr], src[:band + hr], hr, hr, D)[:band]
    assert np.array_equal(np.isnan(gcv), np.isnan(cvb))
    ok = ~np.isnan(cvb)
    assert np.max(np.abs(gcv[ok] - cvb[ok])) <= 1e-4
    # Score branch on the GPU's own cost band
    m = band_mask(H, W, band)
    sb = so.sgm(gcv, 8, so.SCORE, 0.001, 0.01, (0, 0, 0, 0), 100.0)
    gs = res["sgm_cv"][:band].cpu().numpy()
    assert np.array_equal(np.isnan(gs[m]), np.isnan(sb[m]))
    a, b = gs[m], sb[m]
    okv = ~np.isnan(b)
    assert np.array_equal(a[okv].view(np.uint32), b[okv].view(np.uint32))
    # winner + refinement of the band: the truncated volume is NaN for rows i + v_r >= H, so only rows < band - v_r of the
    # cropped oracle run see the same border rule as the full frame
    idx = so.extract_index(gs, so.SCORE)
    gdisp = res["disp"][:band].cpu().numpy()
    assert np.array_equal(gdisp[m], so.index_to_disp(idx)[m])
    ref = so.refine_disp(so.truncated_cost_volume(gs, idx, hr, hr, 1), idx, so.PARABOLA)
    gref = res["refined"][:band].cpu().numpy()
    rows = band - hr
    mm = m[:rows]
    a, b = gref[:rows][mm], ref[:rows][mm]
    assert np.array_equal(np.isnan(a), np.isnan(b))
    okr = ~np.isnan(b)
    assert np.max(np.abs(a[okr] - b[okr])) <= 1e-4
    assert okr.mean() > 0.3


def test_c5_row_bands_8k_fullsize():
    """BASELINE config 5's frame split by rows instead of disparities: eight bands of 540 rows (what eight GPUs would compute,
    with no exchange) equal the single call's map bit for bit."""
    H, W, D = 4320, 8192, 512
    src, tgt, _ = parallax_pair(H, W, 1280, 1280, 1520, 32, 256, seed=5)
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    kw = dict(sgmDirections=8, P1=0.001, P2=0.01, Pout=100.0)
    full = sv.stereoMatch(sv.matchingFunctions.CENSUS, d_tgt, d_src, 4, 4, D, **kw)["disp"]
    from libstevi_amd.sharded import band_range
    for r in range(8):
        b, c = band_range(H, r, 8)
        band = sv.censusBandMatch(d_tgt, d_src, 4, 4, D, (b, c), **kw)
        assert torch.equal(band, full[b:b + c]), f"band {r}"

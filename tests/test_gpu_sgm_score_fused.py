"""The Score branch of sgmCostVolume (sgm.h:218-255, :329-389) with the four downward passes carried by one sweep of the volume
(svh_sgm_sweep.hip; option "sgm_score_fused": 1 = sgm_score_band_kernel, a launch per band of rows with the neighbours' entering lines
recomputed; 3 = the same with 16-column strips forced) against the pass-per-launch kernels and the oracle: same bits."""
import numpy as np
import pytest

import oracle as so

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import libstevi_amd as sv  # noqa: E402

DEV = torch.device("cuda:0")


def bits(x):
    x = x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


FORMS = [2, 3]  # (1, the default, lets a model choose between 0 and 2) 2: a launch per band of 16 rows (halos recomputed; 8-column strips at these widths), 3: the same with the 16-column
# strips of wide images


def both_forms(cv, P1, P2, Pout, form):
    d = torch.from_numpy(cv).to(DEV)
    sv.set_test_option(d, "sgm_score_fused", form)
    try:
        fused = sv.sgmCostVolume(8, so.SCORE, d, P1, P2, None, Pout)
        sv.set_option(d, "sgm_score_fused", 0)
        plain = sv.sgmCostVolume(8, so.SCORE, d, P1, P2, None, Pout)
    finally:
        sv.set_option(d, "sgm_score_fused", 1)  # the default
    return fused, plain


# shapes that put several strips of 16 skewed columns side by side, wider than tall and taller than wide, one row, one column;
# disparity counts on each lane layout (1, 2, 4, 8 per lane), multiples of four or not
@pytest.mark.parametrize("shape", [(37, 90, 40), (90, 37, 40), (64, 64, 17), (50, 70, 100), (33, 47, 256), (20, 35, 300), (1, 50, 8), (50, 1, 8),
                                   (130, 16, 5), (16, 130, 64),
                                   # 3, 5, 6, 7 disparities per lane: whole lanes (192, 320, 384, 448) and the masked forms (160, 330, 400)
                                   (21, 40, 192), (18, 33, 320), (17, 20, 384), (19, 18, 448), (23, 37, 160), (12, 35, 330), (9, 21, 400)])
@pytest.mark.parametrize("form", FORMS)
def test_fused_equals_per_pass_and_oracle(rng, shape, form):
    cv = rng.uniform(-1, 1, shape).astype(np.float32)
    for P1, P2, Pout in [(0.001, 0.01, 100.0), (0.3, 0.3, 0.0), (0.0, 2.0, 0.5)]:
        fused, plain = both_forms(cv, P1, P2, Pout, form)
        exp = so.sgm(cv, 8, so.SCORE, P1, P2, (0, 0, 0, 0), Pout)
        assert np.array_equal(bits(plain), bits(exp))
        assert np.array_equal(bits(fused), bits(exp))


@pytest.mark.parametrize("form", FORMS)
def test_fused_nonfinite_costs(rng, form):
    cv = rng.uniform(0, 4, (40, 75, 48)).astype(np.float32)
    cv[0, 0, 0] = np.nan
    cv[-1, -1, :] = np.inf
    cv[20, 37, 1] = -np.inf
    cv[0, 30, :] = np.nan
    cv[17, :, 5] = np.inf
    cv[:, 44, 7] = np.nan
    fused, plain = both_forms(cv, 0.3, 0.9, 7.0, form)
    exp = so.sgm(cv, 8, so.SCORE, 0.3, 0.9, (0, 0, 0, 0), 7.0)
    f, p = fused.cpu().numpy(), plain.cpu().numpy()
    assert np.array_equal(np.isnan(f), np.isnan(exp)) and np.array_equal(np.isnan(p), np.isnan(exp))
    ok = ~np.isnan(exp)
    assert np.array_equal(f[ok].view(np.uint32), exp[ok].view(np.uint32))
    assert np.array_equal(p[ok].view(np.uint32), exp[ok].view(np.uint32))


@pytest.mark.parametrize("form", FORMS)
def test_fused_many_strips_against_per_pass(rng, form):
    """More strips than the card holds blocks at once is not reachable at test sizes; this one has 150 strips and 600 rows, so the
    hand-off runs a few hundred rows deep, and is compared with the pass-per-launch kernels (the oracle takes minutes here)."""
    cv = rng.uniform(-1, 1, (600, 1800, 64)).astype(np.float32)
    fused, plain = both_forms(cv, 0.02, 0.2, 3.0, form)
    assert torch.equal(fused.view(torch.int32), plain.view(torch.int32))


@pytest.mark.parametrize("seed", [3, 5])
def test_fused_forms_on_random_geometries(seed):
    """Random small shapes (one row, one column, fewer columns than a strip, D from 1 to 300, costs with NaN / inf sprinkled in):
    the three fused forms and the per-pass kernels against the oracle."""
    rng = np.random.default_rng(seed)
    for case in range(25):
        H, W = int(rng.integers(1, 45)), int(rng.integers(1, 70))
        D = int(rng.choice([1, 3, 8, 17, 64, 65, 128, 256, 300]))
        cv = rng.uniform(-2, 2, (H, W, D)).astype(np.float32)
        if rng.random() < 0.3:
            cv[rng.integers(0, H), rng.integers(0, W), rng.integers(0, D)] = np.nan
            cv[rng.integers(0, H), rng.integers(0, W), :] = np.inf
        P1 = float(rng.choice([0.0, 0.001, 0.3]))
        P2 = P1 + float(rng.choice([0.0, 0.01, 1.0]))
        Pout = float(rng.choice([0.0, 0.5, 100.0]))
        exp = so.sgm(cv, 8, so.SCORE, P1, P2, (0, 0, 0, 0), Pout)
        ok = ~np.isnan(exp)
        d = torch.from_numpy(cv).to(DEV)
        try:
            for form in (0, 1, 2, 3):
                sv.set_test_option(d, "sgm_score_fused", form)
                got = sv.sgmCostVolume(8, so.SCORE, d, P1, P2, None, Pout).cpu().numpy()
                what = f"case {case} form {form}: {H}x{W}x{D} P1={P1} P2={P2} Pout={Pout}"
                assert np.array_equal(np.isnan(got), np.isnan(exp)), what
                assert np.array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32)), what
        finally:
            sv.set_option(d, "sgm_score_fused", 1)


# ---- the winner riding on the last writer of each pixel (svh_stereo_match, option "sgm_score_finish_fused") -------------------------------
@pytest.mark.parametrize("shape_d", [((40, 150), 64), ((150, 40), 64), ((33, 200), 128), ((70, 90), 256), ((9, 300), 64), ((64, 64), 64), ((1, 80), 64),
                                     ((45, 97), 40), ((30, 260), 192), ((40, 400), 320), ((25, 470), 448), ((33, 300), 160)])
@pytest.mark.parametrize("func_name", ["NCC", "ZNCC", "CC"])
def test_winner_emitted_by_the_last_writer(rng, shape_d, func_name):
    """svh_stereo_match with a Score-strategy function + SGM-8: the index, the disparity and the three truncatedCostVolume<Same> taps of a
    pixel are emitted by the launch that writes its final aggregated costs (DownLeft2UpRight where i + j < H, the downward sweep
    elsewhere) instead of reading S back; without a request for S only the costs a later pass needs are stored.  Must equal the
    separate extract_index / truncatedCostVolume kernels bit for bit (disparity map, refined map incl. its NaN mask), with and without
    the volume requested, every kernel -- and the oracle run on the library's own cost volume.  (D = 40: not the vector form, the fused
    finish declines and the old path runs.)"""
    from helpers import parallax_pair
    (H, W), D = shape_d
    MF = sv.matchingFunctions
    func = getattr(MF, func_name)
    src, tgt, _ = parallax_pair(H, W, max(2, min(H, W) // 4), H // 4, W // 4, 2, 9, seed=H * 1000 + W + D)
    src, tgt = src[:H, :W].copy(), tgt[:H, :W].copy()
    l, r = torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV)
    hr = 2

    def run(fused, kernel, want_sgm_cv):
        sv.set_test_option(l, "sgm_score_finish_fused", fused)
        try:
            sv.profile_reset(l)
            sv.profile_enable(l, True)
            out = sv.stereoMatch(func, l, r, hr, hr, D, sgmDirections=8, P1=0.001, P2=0.01, Pout=100.0, refineKernel=kernel, refine_h_radius=hr,
                                 refine_v_radius=hr, want_sgm_cv=want_sgm_cv, want_cv=True)
            sv.profile_enable(l, False)
            return out, sv.profile_collect(l)
        finally:
            sv.set_test_option(l, "sgm_score_finish_fused", 1)

    for kernel in (sv.InterpolationKernel.Parabola, sv.InterpolationKernel.Equiangular, None):
        for want_sgm_cv in (False, True):
            a, ka = run(1, kernel, want_sgm_cv)
            b, kb = run(0, kernel, want_sgm_cv)
            assert "extract_index" in kb
            if (D % 64 == 0 or 64 < D < 512) and H * W > 0:  # (rows padded to whole lanes take the records route too)
                assert "extract_index" not in ka and "truncated_cost_volume" not in ka and "index_to_disp" not in ka, ka.keys()
            assert torch.equal(a["disp"], b["disp"])
            if kernel is not None:
                ra, rb = a["refined"].cpu().numpy(), b["refined"].cpu().numpy()
                assert np.array_equal(np.isnan(ra), np.isnan(rb))
                assert np.array_equal(bits(ra[~np.isnan(ra)]), bits(rb[~np.isnan(rb)]))
            if want_sgm_cv:
                assert np.array_equal(bits(a["sgm_cv"]), bits(b["sgm_cv"]))
    # against the oracle, on the cost volume the library made (the column-sum kernel agrees with the oracle's to 1e-4, not to the bit)
    cvh = a["cv"].cpu().numpy()
    svol = so.sgm(cvh, 8, so.SCORE, 0.001, 0.01, (0, 0, 0, 0), 100.0)
    idx = so.extract_index(svol, so.SCORE)
    assert np.array_equal(a["disp"].cpu().numpy(), so.index_to_disp(idx))
    out, _ = run(1, sv.InterpolationKernel.Parabola, False)
    exp = so.refine_disp(so.truncated_cost_volume(svol, idx, hr, hr, 1), idx, so.PARABOLA)
    got = out["refined"].cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    ok = ~np.isnan(exp)
    assert np.max(np.abs(got[ok] - exp[ok]), initial=0.0) <= 1e-4


def test_winner_emitted_by_the_last_writer_ties_and_nan(rng):
    """Few cost levels (ties everywhere: the larger index must win), NaN voxels (never win unless at index 0), whole NaN pixels: the fused
    finish against the separate kernels on the same call, through CC on crafted images is not possible -- so the Score branch is fed a
    crafted VOLUME through sgmCostVolume + extractSelectedIndex and compared with the fused pipeline's selection rule in isolation:
    wave_select_index is also what sgm_cost... no: this test drives the rule through stereoMatch on constant images (all costs equal)."""
    MF = sv.matchingFunctions
    H, W, D = 48, 130, 64
    img = np.full((H, W), 0.5, np.float32)
    l = torch.from_numpy(img).to(DEV)
    for fused in (1, 0):
        sv.set_test_option(l, "sgm_score_finish_fused", fused)
        try:
            out = sv.stereoMatch(MF.CC, l, l, 1, 1, D, sgmDirections=8, P1=0.0, P2=0.0, Pout=0.0)
        finally:
            sv.set_test_option(l, "sgm_score_finish_fused", 1)
        cv = so.unfold_cost_volume(so.CC, img, img, 1, 1, D)
        exp = so.index_to_disp(so.extract_index(so.sgm(cv, 8, so.SCORE, 0.0, 0.0, (0, 0, 0, 0), 0.0), so.SCORE))
        assert np.array_equal(out["disp"].cpu().numpy(), exp), fused
    # NCC of an all-zero image: every cost NaN (0 / 0): index 0 everywhere, refined NaN or 0 as the separate kernels say
    z = torch.zeros((H, W), device=DEV)
    outs = []
    for fused in (1, 0):
        sv.set_test_option(z, "sgm_score_finish_fused", fused)
        try:
            outs.append(sv.stereoMatch(MF.NCC, z, z, 1, 1, D, sgmDirections=8, refineKernel=sv.InterpolationKernel.Parabola, refine_h_radius=1, refine_v_radius=1))
        finally:
            sv.set_test_option(z, "sgm_score_finish_fused", 1)
    assert torch.equal(outs[0]["disp"], outs[1]["disp"]) and int(outs[0]["disp"].abs().sum()) == 0
    ra, rb = outs[0]["refined"].cpu().numpy(), outs[1]["refined"].cpu().numpy()
    assert np.array_equal(np.isnan(ra), np.isnan(rb)) and np.array_equal(ra[~np.isnan(ra)], rb[~np.isnan(rb)])


@pytest.mark.parametrize("flat_patch", [False, True])
@pytest.mark.parametrize("func_name", ["NCC", "ZNCC"])
def test_all_finite_regime_drops_the_filters_and_nothing_else(func_name, flat_patch):
    """svh_stereo_match, normalised function + SGM-8 on a volume of 2^26 voxels or more: the statistics kernels report whether every window
    norm is a positive finite number; if so every cost is finite and the Score-branch kernels run without the isfinite filters of
    sgm.h:224, :241, :251 (SgmArgs::costs_all_finite), else (a flat patch: zero norms, NaN costs) with them.  Either way the maps must equal
    the separate-kernel path's (no query there) bit for bit, and the oracle's on a band of the library's own volume."""
    from helpers import parallax_pair
    MF = sv.matchingFunctions
    func = getattr(MF, func_name)
    H, W, D, hr = 512, 1024, 128, 2
    src, tgt, _ = parallax_pair(H, W, 128, 100, 300, 3, 21, seed=17)
    if flat_patch:
        src[200:230, 400:470] = 0.0 if func_name == "NCC" else 0.375  # zero norm (NCC) / zero sigma (ZNCC): 0 / 0 costs
        tgt[40:60, 100:140] = 0.0 if func_name == "NCC" else -0.5
    l, r = torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV)
    kw = dict(sgmDirections=8, P1=0.001, P2=0.01, Pout=100.0, refineKernel=sv.InterpolationKernel.Parabola, refine_h_radius=hr, refine_v_radius=hr)
    sv.set_option(l, "sgm_score_fused", 2)  # (the bands: the automatic choice runs a volume of this size pass by pass)
    try:
        a = sv.stereoMatch(func, l, r, hr, hr, D, want_cv=True, want_sgm_cv=True, **kw)
        sv.set_test_option(l, "sgm_score_finish_fused", 0)
        b = sv.stereoMatch(func, l, r, hr, hr, D, want_sgm_cv=True, **kw)
    finally:
        sv.set_test_option(l, "sgm_score_finish_fused", 1)
        sv.set_option(l, "sgm_score_fused", 1)
    auto = sv.stereoMatch(func, l, r, hr, hr, D, **kw)  # the default choice: the same maps
    assert torch.equal(a["disp"], auto["disp"])
    assert torch.equal(a["disp"], b["disp"])
    sa, sb = a["sgm_cv"].cpu().numpy(), b["sgm_cv"].cpu().numpy()
    assert np.array_equal(np.isnan(sa), np.isnan(sb)) and np.array_equal(sa[~np.isnan(sa)].view(np.uint32), sb[~np.isnan(sb)].view(np.uint32))
    ra, rb = a["refined"].cpu().numpy(), b["refined"].cpu().numpy()
    assert np.array_equal(np.isnan(ra), np.isnan(rb)) and np.array_equal(bits(ra[~np.isnan(ra)]), bits(rb[~np.isnan(rb)]))
    cvh = a["cv"].cpu().numpy()
    assert bool(np.isnan(cvh).any()) == flat_patch
    del sb, b
    # oracle on the top band of the library's own volume (pixels whose lines lie inside the band)
    band = 24
    ii, jj = np.meshgrid(np.arange(band), np.arange(W), indexing="ij")
    m = (ii + jj < band) | (ii + jj >= H)
    ob = so.sgm(cvh[:band], 8, so.SCORE, 0.001, 0.01, (0, 0, 0, 0), 100.0)
    x, y = sa[:band][m], ob[m]
    assert np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(x[~np.isnan(y)].view(np.uint32), y[~np.isnan(y)].view(np.uint32))
    # ... and a band that contains the flat patch of the source image, against the oracle's aggregation of the rows above it
    if flat_patch:
        rows = 240
        ob = so.sgm(cvh[:rows], 8, so.SCORE, 0.001, 0.01, (0, 0, 0, 0), 100.0)
        ii, jj = np.meshgrid(np.arange(rows), np.arange(W), indexing="ij")
        m = (ii + jj < rows) | (ii + jj >= H)
        x, y = sa[:rows][m], ob[m]
        assert np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(x[~np.isnan(y)].view(np.uint32), y[~np.isnan(y)].view(np.uint32))


@pytest.mark.parametrize("D", [70, 100, 160, 330, 500, 511])
def test_rows_padded_to_whole_lanes_equal_the_masked_forms(rng, D):
    """Option "sgm_score_pad" (default): Score-branch SGM on a disparity count that is no multiple of 64 aggregates a copy whose rows are
    padded with -inf to the next multiple, so that the vector kernels, the banded sweep and the winner records apply.  Same bits as the
    masked kernels on the caller's layout and as the oracle: whole image, margins, four directions, P1 > P2, non-finite costs, pixels
    whose every cost is -inf or NaN (where a pad equals the extremum), and the winner that travels with the volume."""
    H, W = 21, 37
    cv = rng.uniform(-1, 1, (H, W, D)).astype(np.float32)
    cv[3, 4, :] = -np.inf          # every real value equals the pads
    cv[5, 6, :] = np.nan           # nothing equals anything: index 0
    cv[7, 8, 1:] = -np.inf         # NaN at index 0 beside infinities
    cv[7, 8, 0] = np.nan
    cv[9, 10, D - 1] = np.inf      # the last real disparity wins: its right tap does not exist
    cv[11, 12, 5] = np.nan
    cv[2, 30, :D // 2] = -np.inf
    d = torch.from_numpy(cv).to(DEV)
    for n_dir, margins, (P1, P2, Pout) in [(8, None, (0.001, 0.01, 100.0)), (8, sv.Margins(1, 2, 3, 1), (0.3, 0.9, 3.0)), (4, None, (0.2, 0.7, 0.5)),
                                           (8, None, (0.9, 0.3, 1.0))]:
        m = (0, 0, 0, 0) if margins is None else (1, 2, 3, 1)
        exp = so.sgm(cv, n_dir, so.SCORE, P1, P2, m, Pout)
        outs = []
        for pad in (1, 0):
            sv.set_test_option(d, "sgm_score_pad", pad)
            try:
                vol = sv.sgmCostVolume(n_dir, so.SCORE, d, P1, P2, margins, Pout, keep_winner=True)
                idx = sv.extractSelectedIndex(so.SCORE, vol)
            finally:
                sv.set_test_option(d, "sgm_score_pad", 1)
            outs.append((vol.cpu().numpy(), idx.cpu().numpy()))
        for vol, idx in outs:
            assert np.array_equal(np.isnan(vol), np.isnan(exp))
            ok = ~np.isnan(exp)
            assert np.array_equal(vol[ok].view(np.uint32), exp[ok].view(np.uint32)), (D, n_dir, P1)
            assert np.array_equal(idx, so.extract_index(exp, so.SCORE)), (D, n_dir, P1)


@pytest.mark.parametrize("shape_d", [((40, 150), 64), ((33, 200), 128), ((30, 260), 192), ((25, 300), 160), ((1, 80), 64), ((50, 1), 64)])
@pytest.mark.parametrize("func_name", ["NCC", "ZNCC"])
def test_winner_emitted_by_left2right_with_four_directions(rng, shape_d, func_name):
    """Four directions (sgm.h:379-383: Up2Down, Left2Right): the second pass visits every pixel and is the last writer of each, so it emits
    the winner records; the volume is stored only on request.  Equal to the separate extract_index / truncatedCostVolume kernels and to the
    oracle on the library's own cost volume."""
    from helpers import parallax_pair
    (H, W), D = shape_d
    func = getattr(sv.matchingFunctions, func_name)
    src, tgt, _ = parallax_pair(H, W, max(2, min(H, W) // 4), H // 4, W // 4, 2, 9, seed=H * 1000 + W + D)
    l, r = torch.from_numpy(tgt[:H, :W].copy()).to(DEV), torch.from_numpy(src[:H, :W].copy()).to(DEV)
    hr = 2

    def run(fused, kernel, want_sgm_cv):
        sv.set_test_option(l, "sgm_score_finish_fused", fused)
        try:
            sv.profile_reset(l)
            sv.profile_enable(l, True)
            out = sv.stereoMatch(func, l, r, hr, hr, D, sgmDirections=4, P1=0.001, P2=0.01, Pout=100.0, refineKernel=kernel, refine_h_radius=hr,
                                 refine_v_radius=hr, want_sgm_cv=want_sgm_cv, want_cv=True)
            sv.profile_enable(l, False)
            return out, sv.profile_collect(l)
        finally:
            sv.set_test_option(l, "sgm_score_finish_fused", 1)

    for kernel in (sv.InterpolationKernel.Parabola, None):
        for want_sgm_cv in (False, True):
            a, ka = run(1, kernel, want_sgm_cv)
            b, kb = run(0, kernel, want_sgm_cv)
            assert "extract_index" in kb and "extract_index" not in ka and "finish_records" in ka, (ka.keys(), kb.keys())
            assert torch.equal(a["disp"], b["disp"])
            if kernel is not None:
                ra, rb = a["refined"].cpu().numpy(), b["refined"].cpu().numpy()
                assert np.array_equal(np.isnan(ra), np.isnan(rb))
                assert np.array_equal(bits(ra[~np.isnan(ra)]), bits(rb[~np.isnan(rb)]))
            if want_sgm_cv:
                assert np.array_equal(bits(a["sgm_cv"]), bits(b["sgm_cv"]))
    svol = so.sgm(a["cv"].cpu().numpy(), 4, so.SCORE, 0.001, 0.01, (0, 0, 0, 0), 100.0)
    assert np.array_equal(a["disp"].cpu().numpy(), so.index_to_disp(so.extract_index(svol, so.SCORE)))


@pytest.mark.parametrize("D", [100, 160, 330])
@pytest.mark.parametrize("n_dir", [4, 8])
def test_fused_call_writes_padded_rows_directly(rng, D, n_dir):
    """svh_stereo_match, Score strategy, a disparity count that is no multiple of 64, nobody asking for the cost volume: the cost kernel
    writes rows of the padded pitch itself and only the pads are filled (no pad-in copy).  Same disparity and refined maps as the call
    that also returns the volumes (which aggregates a padded copy), and as the masked kernels."""
    from helpers import parallax_pair
    H, W = 37, 400
    src, tgt, _ = parallax_pair(H, W, 12, 9, 100, 2, 9, seed=D + n_dir)
    l, r = torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV)
    kw = dict(sgmDirections=n_dir, P1=0.001, P2=0.01, Pout=100.0, refineKernel=sv.InterpolationKernel.Parabola, refine_h_radius=2, refine_v_radius=2)
    a = sv.stereoMatch(sv.matchingFunctions.ZNCC, l, r, 2, 2, D, **kw)
    b = sv.stereoMatch(sv.matchingFunctions.ZNCC, l, r, 2, 2, D, want_cv=True, want_sgm_cv=True, **kw)
    sv.set_test_option(l, "sgm_score_pad", 0)
    try:
        c = sv.stereoMatch(sv.matchingFunctions.ZNCC, l, r, 2, 2, D, **kw)
    finally:
        sv.set_test_option(l, "sgm_score_pad", 1)
    for other in (b, c):
        assert torch.equal(a["disp"], other["disp"])
        ra, rb = a["refined"].cpu().numpy(), other["refined"].cpu().numpy()
        assert np.array_equal(np.isnan(ra), np.isnan(rb)) and np.array_equal(bits(ra[~np.isnan(ra)]), bits(rb[~np.isnan(rb)]))
    svol = so.sgm(b["cv"].cpu().numpy(), n_dir, so.SCORE, 0.001, 0.01, (0, 0, 0, 0), 100.0)
    assert np.array_equal(bits(b["sgm_cv"]), bits(svol))
    assert np.array_equal(a["disp"].cpu().numpy(), so.index_to_disp(so.extract_index(svol, so.SCORE)))


@pytest.mark.parametrize("shape_d", [((40, 150), 64), ((33, 200), 128), ((30, 260), 192), ((25, 300), 160), ((21, 310), 100), ((1, 80), 64), ((50, 1), 64)])
@pytest.mark.parametrize("func_name", ["NCC", "ZNCC"])
def test_winner_records_from_the_volume_when_the_passes_ran(rng, shape_d, func_name):
    """A launch per pass with eight directions (option "sgm_score_fused" 0; what the default picks for small volumes): no pass is the last
    writer of every pixel, so one kernel reads the finished volume (padded rows included) and emits the winner records.  Equal to the
    separate extract_index / truncatedCostVolume kernels, with and without the volume requested, and to the banded form."""
    from helpers import parallax_pair
    (H, W), D = shape_d
    func = getattr(sv.matchingFunctions, func_name)
    src, tgt, _ = parallax_pair(H, W, max(2, min(H, W) // 4), H // 4, W // 4, 2, 9, seed=H * 1000 + W + D)
    l, r = torch.from_numpy(tgt[:H, :W].copy()).to(DEV), torch.from_numpy(src[:H, :W].copy()).to(DEV)
    hr = 2

    def run(form, fused, kernel, want_sgm_cv):
        sv.set_test_option(l, "sgm_score_fused", form)
        sv.set_test_option(l, "sgm_score_finish_fused", fused)
        try:
            sv.profile_reset(l)
            sv.profile_enable(l, True)
            out = sv.stereoMatch(func, l, r, hr, hr, D, sgmDirections=8, P1=0.001, P2=0.01, Pout=100.0, refineKernel=kernel, refine_h_radius=hr,
                                 refine_v_radius=hr, want_sgm_cv=want_sgm_cv)
            sv.profile_enable(l, False)
            return out, sv.profile_collect(l)
        finally:
            sv.set_option(l, "sgm_score_fused", 1)
            sv.set_test_option(l, "sgm_score_finish_fused", 1)

    for kernel in (sv.InterpolationKernel.Parabola, None):
        for want_sgm_cv in (False, True):
            a, ka = run(0, 1, kernel, want_sgm_cv)
            b, kb = run(0, 0, kernel, want_sgm_cv)
            c, _ = run(2, 1, kernel, want_sgm_cv)
            if D > 64 or D % 64 == 0:
                assert "sgm_score_records" in ka and "extract_index" not in ka, ka.keys()
            assert "extract_index" in kb
            for other in (b, c):
                assert torch.equal(a["disp"], other["disp"])
                if kernel is not None:
                    ra, rb = a["refined"].cpu().numpy(), other["refined"].cpu().numpy()
                    assert np.array_equal(np.isnan(ra), np.isnan(rb))
                    assert np.array_equal(bits(ra[~np.isnan(ra)]), bits(rb[~np.isnan(rb)]))
                if want_sgm_cv:
                    assert np.array_equal(bits(a["sgm_cv"]), bits(other["sgm_cv"]))

"""Randomised geometries of the census + SGM disparity map: the default path (winner written by the matrix-core sweep, no line scans),
the scan path (census_winner_shortcut = 0), the general wave-per-line kernels (census_fast_path = 0), the row-band call and the
oracle's materialised volume + SGM + argmin must all give the same map: tiny and ragged images, more disparities than columns,
margins larger than the image, every direction / pass count / Pout."""
import numpy as np
import pytest

import oracle as so

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import libstevi_amd as sv  # noqa: E402
from libstevi_amd import matchingFunctions as MF  # noqa: E402
from libstevi_amd._capi import SvhError, ERR_UNSUPPORTED  # noqa: E402

DEV = torch.device("cuda:0")


@pytest.mark.parametrize("seed", [7, 11])
def test_census_paths_agree_on_random_geometries(seed):
    rng = np.random.default_rng(seed)
    for case in range(40):
        H, W = int(rng.integers(1, 40)), int(rng.integers(2, 150))
        D = int(rng.choice([32, 64, 96, 128, 33, 7]))
        r = int(rng.choice([3, 4, 5]))
        n_dir = int(rng.choice([0, 4, 8]))
        Pout = float(rng.choice([0.0, 1.0, 7.0, 100.0, 5000.0]))
        margins = tuple(int(x) for x in rng.integers(0, 6, 4)) if rng.random() < 0.5 else (0, 0, 0, 0)
        ddir = sv.dispDirection.RightToLeft if rng.random() < 0.5 else sv.dispDirection.LeftToRight
        src = rng.uniform(-1, 1, (H, W)).astype(np.float32)
        tgt = rng.uniform(-1, 1, (H, W + (int(rng.integers(0, 3)) if rng.random() < 0.2 else 0))).astype(np.float32)
        d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
        kw = dict(dDir=ddir, sgmDirections=n_dir, P1=0.3, P2=0.9, Pout=Pout, margins=sv.Margins(*margins))
        what = f"case {case}: {H}x{W} D={D} r={r} dirs={n_dir} Pout={Pout} margins={margins} dDir={int(ddir)} target {tgt.shape}"
        try:
            default = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, r, r, D, **kw)["disp"].cpu().numpy()
            sv.set_option(d_tgt, "census_winner_shortcut", 0)
            scans = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, r, r, D, **kw)["disp"].cpu().numpy()
            sv.set_test_option(d_tgt, "census_fast_path", 0)
            general = sv.stereoMatch(MF.CENSUS, d_tgt, d_src, r, r, D, **kw)["disp"].cpu().numpy()
        finally:
            sv.set_option(d_tgt, "census_winner_shortcut", 1)
            sv.set_test_option(d_tgt, "census_fast_path", 1)
        cv = so.unfold_cost_volume(so.CENSUS, tgt, src, r, r, D, int(ddir))
        vol = so.sgm(cv, n_dir, so.COST, 0.3, 0.9, margins, Pout) if n_dir else cv
        exp = so.index_to_disp(so.extract_index(vol, so.COST), int(ddir))
        assert np.array_equal(default, exp), what
        assert np.array_equal(scans, exp), what
        assert np.array_equal(general, exp), what
        if H >= 2:
            b0 = H // 3
            try:
                band = sv.censusBandMatch(d_tgt, d_src, r, r, D, (b0, H - b0), **kw).cpu().numpy()
            except SvhError as e:  # geometries the matrix-core sweep does not take (D not a multiple of 32, ...)
                assert e.status == ERR_UNSUPPORTED, what
            else:
                assert np.array_equal(band, exp[b0:]), what

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


FORMS = [1, 3]  # 1: a launch per band of 16 rows (halos recomputed; 8-column strips at these widths), 3: the same with the 16-column
# strips of wide images


def both_forms(cv, P1, P2, Pout, form):
    d = torch.from_numpy(cv).to(DEV)
    sv.set_option(d, "sgm_score_fused", form)
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
                                   (130, 16, 5), (16, 130, 64)])
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
            for form in (0, 1, 3):
                sv.set_option(d, "sgm_score_fused", form)
                got = sv.sgmCostVolume(8, so.SCORE, d, P1, P2, None, Pout).cpu().numpy()
                what = f"case {case} form {form}: {H}x{W}x{D} P1={P1} P2={P2} Pout={Pout}"
                assert np.array_equal(np.isnan(got), np.isnan(exp)), what
                assert np.array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32)), what
        finally:
            sv.set_option(d, "sgm_score_fused", 1)

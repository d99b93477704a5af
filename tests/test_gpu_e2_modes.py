"""Rule E2 (cross_correlations.h:235-236): the target census word goes through `float` and back.  A word >= 0xFFFFFF80
rounds to 2^32, whose conversion back is undefined in C++; the reference's builds give 0xFFFFFFFF (AVX-512 code generation) or 0
(x86-64 without it: tests/test_oracle_semantics.py::test_e2_overflow_matches_this_hosts_conversions runs both on the host).
svh_context_set_option("census_float_overflow", 0 | 1) selects saturate | zero; the oracle has the same switch.  Smooth gradients
produce such words everywhere (the census reference sample is the window's top-left corner, so a surface that falls to the right and
downwards sets every bit); random images almost never do, which is why this needs its own fixture."""
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


def gradient_pair(H, W, seed):
    """A surface falling to the right and downwards (all-ones census words) with sparse bumps that clear a few low or high bits, and a
    shifted copy: words on both sides of 0xFFFFFF80."""
    rng = np.random.default_rng(seed)
    base = (-(np.arange(W, dtype=np.float32))[None, :] - 1.5 * np.arange(H, dtype=np.float32)[:, None]).astype(np.float32)
    bumps = np.where(rng.uniform(size=(H, W)) < 0.04, rng.uniform(2.0, 40.0, (H, W)), 0.0).astype(np.float32)
    tgt = base + bumps
    src = np.roll(tgt, -3, axis=1) + np.where(rng.uniform(size=(H, W)) < 0.02, 25.0, 0.0).astype(np.float32)
    return np.ascontiguousarray(src), np.ascontiguousarray(tgt)


@pytest.fixture
def modes():
    probe = torch.zeros(1, device=DEV)

    def set_mode(zero):
        so.set_float_overflow(zero)
        sv.set_option(probe, "census_float_overflow", int(zero))
    yield set_mode
    set_mode(False)
    sv.set_option(probe, "census_sweep", 0)
    sv.set_test_option(probe, "census_sweep_rl", 1)


def test_fixture_reaches_the_overflowing_words():
    src, tgt = gradient_pair(24, 160, 1)
    words = so.census_transform(tgt, 4, 4)[..., :2]
    big = words >= 0xFFFFFF80
    assert big.mean() > 0.3 and (~big).mean() > 0.05


@pytest.mark.parametrize("h_r", [3, 4, 5])  # one, two and three written words
def test_hamming_volume_both_modes(modes, h_r):
    src, tgt = gradient_pair(24, 160, 2 + h_r)
    vols = []
    for zero in (False, True):
        modes(zero)
        exp = so.unfold_cost_volume(so.CENSUS, tgt, src, h_r, h_r, 32)
        got = sv.unfoldBasedCostVolume(MF.CENSUS, torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV), h_r, h_r, 32).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
        vols.append(got)
    # the switch matters on this fixture: costs differ by up to 32 per word
    assert (vols[0] != vols[1]).mean() > 0.2 and np.abs(vols[0] - vols[1]).max() >= 25


@pytest.mark.parametrize("D,W", [(64, 200), (96, 333), (256, 420)])
def test_census_sgm_every_engine_both_modes(modes, D, W):
    """The fused census + SGM path (keys and disparities) through the vector-ALU sweep, the int8 and the two FP4 matrix-core kernels,
    for both conversions, against the oracle chain volume -> sgm -> extract_index."""
    src, tgt = gradient_pair(14, W, D)
    l, r = torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV)
    maps = []
    for zero in (False, True):
        modes(zero)
        cv = so.unfold_cost_volume(so.CENSUS, tgt, src, 4, 4, D)
        want = so.extract_index(so.sgm(cv, 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0), so.COST)
        keys = []
        for mode, rl in ((1, 1), (3, 0), (3, 1)):
            sv.set_option(l, "census_sweep", mode)
            sv.set_test_option(l, "census_sweep_rl", rl)
            keys.append(sv.censusShardKeys(l, r, 4, 4, D, (0, D), sgmDirections=8, Pout=100.0).cpu().numpy())
            for shortcut in (0, 1):
                sv.set_option(l, "census_winner_shortcut", shortcut)
                got = sv.stereoMatch(MF.CENSUS, l, r, 4, 4, D, sgmDirections=8, Pout=100.0)["disp"].cpu().numpy()
                assert np.array_equal(got, want), (zero, mode, rl, shortcut)
            sv.set_option(l, "census_winner_shortcut", 1)
        assert all(np.array_equal(keys[0], k) for k in keys[1:])
        sv.set_option(l, "census_sweep", 0)
        # the S volume itself (per-function chain on the device)
        vol = sv.sgmCostVolume(8, so.COST, sv.unfoldBasedCostVolume(MF.CENSUS, l, r, 4, 4, D), 0.001, 0.01, None, 100.0).cpu().numpy()
        assert np.array_equal(vol.view(np.uint32), so.sgm(cv, 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0).view(np.uint32))
        maps.append(want)
    assert (maps[0] != maps[1]).any()  # the two conversions give different disparity maps here


def test_option_is_validated():
    probe = torch.zeros(1, device=DEV)
    with pytest.raises(sv._capi.SvhError):
        sv.set_option(probe, "census_float_overflow", 2)

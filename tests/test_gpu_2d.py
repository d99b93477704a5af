"""2-D disparity volumes on the GPU: parity with the oracle and the reference's own
property tests (testCorrelation2d.cpp:75-191) run directly against the HIP path."""
import numpy as np
import pytest

import oracle as so
from helpers import naive_window_cost

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import libstevi_amd as sv  # noqa: E402
from libstevi_amd import matchingFunctions as MF  # noqa: E402

DEV = torch.device("cuda:0")
NAMES = {"NCC": MF.NCC, "ZNCC": MF.ZNCC, "SSD": MF.SSD, "ZSSD": MF.ZSSD, "SAD": MF.SAD, "ZSAD": MF.ZSAD, "CC": MF.CC, "ZCC": MF.ZCC}


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def host(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else x


@pytest.mark.parametrize("func", list(NAMES.values()) + [MF.CENSUS])
@pytest.mark.parametrize("ddir", [sv.dispDirection.RightToLeft, sv.dispDirection.LeftToRight])
def test_2d_volume_matches_oracle(rng, func, ddir):
    left = rng.uniform(-1, 1, (19, 41)).astype(np.float32)
    right = rng.uniform(-1, 1, (19, 41)).astype(np.float32)
    for (h_r, v_r, r0, r1) in [(1, 1, (-2, 3), (-4, 5)), (4, 4, (0, 2), (0, 6)), (2, 3, (-25, -20), (3, 3))]:
        exp = so.unfold_cost_volume_2d(int(func), left, right, h_r, v_r, r0, r1, int(ddir))
        for mk in (lambda x: x, dev):
            got = host(sv.unfoldBased2dDisparityCostVolume(func, mk(left), mk(right), h_r, v_r, sv.searchOffset2(r0[0], r0[1], r1[0], r1[1]), ddir))
            assert got.shape == exp.shape
            if func == MF.CENSUS:
                assert np.array_equal(got, exp)
            else:
                assert np.array_equal(np.isnan(got), np.isnan(exp))
                ok = ~np.isnan(exp)
                assert np.all(np.abs(got[ok] - exp[ok]) <= 1e-4 * np.maximum(1, np.abs(exp[ok])))
        strat = so.func_strategy(int(func))
        idx = sv.extractSelected2dIndex(strat, got)
        assert np.array_equal(host(idx), so.extract_index_2d(got, strat))
        off = sv.searchOffset2(r0[0], r0[1], r1[0], r1[1])
        assert np.array_equal(host(sv.selected2dIndexToDisp(idx, off)), so.index_2d_to_disp(host(idx), r0[0], r1[0]))
        tcv = host(sv.truncatedBidirectionaCostVolume(dev(got), dev(host(idx)), 1, 2))
        exp_t = so.truncated_bidirectional_cv(got, host(idx), 1, 2)
        assert np.array_equal(np.isnan(tcv), np.isnan(exp_t)) and np.array_equal(tcv[~np.isnan(exp_t)], exp_t[~np.isnan(exp_t)])


def test_2d_volume_shape_errors(rng):
    a = rng.uniform(-1, 1, (8, 9)).astype(np.float32)
    b = rng.uniform(-1, 1, (8, 10)).astype(np.float32)
    assert sv.unfoldBased2dDisparityCostVolume(MF.SAD, a, b, 1, 1, sv.searchOffset2(0, 1, 0, 1)).size == 0  # cross_correlations.h:808-810
    assert sv.unfoldBased2dDisparityCostVolume(MF.SAD, a, a, 1, 1, sv.searchOffset2(2, 1, 0, 1)).size == 0  # :338-340


@pytest.mark.parametrize("h_r,v_r,disp_w,disp_h", [(1, 1, 5, 5), (3, 3, 5, 5), (5, 1, 5, 3), (1, 5, 3, 5), (5, 5, 5, 5)])
@pytest.mark.parametrize("name", ["NCC", "ZNCC", "SSD", "ZSSD", "SAD", "ZSAD"])
def test_reference_test2dMatching_on_gpu(rng, name, h_r, v_r, disp_w, disp_h):
    """testCorrelation2d.cpp:75-127 against the HIP path itself, tolerance 1e-3 as in the reference."""
    h, w = 2 * v_r + disp_h + 1, 2 * h_r + disp_w + 1
    left = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    right = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    cv = host(sv.unfoldBased2dDisparityCostVolume(NAMES[name], left, right, h_r, v_r, sv.searchOffset2(0, disp_h, 0, disp_w)))
    w1 = right[0:2 * v_r + 1, 0:2 * h_r + 1]
    for i in range(disp_h):
        for j in range(disp_w):
            assert abs(naive_window_cost(name, w1, left[i:i + 2 * v_r + 1, j:j + 2 * h_r + 1]) - cv[v_r, h_r, i, j]) < 1e-3


@pytest.mark.parametrize("h_r,v_r", [(1, 1), (3, 3), (5, 1), (1, 5)])
@pytest.mark.parametrize("name", ["NCC", "ZNCC", "SSD", "ZSSD", "SAD", "ZSAD"])
def test_reference_test2dDisparity_on_gpu(rng, name, h_r, v_r):
    """testCorrelation2d.cpp:131-191: planted shifts at both ends of the range are recovered exactly."""
    disp_w = disp_h = 3
    h, w = 2 * v_r + disp_h + 1, 2 * h_r + disp_w + 1
    source = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    target0 = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    targetmax = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    for i in range(2 * v_r + 1):
        for j in range(2 * h_r + 1):
            target0[i, j] = source[i + 1, j + 1]
            targetmax[i + disp_h, j + disp_w] = source[i + 1, j + 1]
    off = sv.searchOffset2(-1, disp_h - 1, -1, disp_w - 1)
    strat = sv.matchFuncStrategy(NAMES[name])
    for target, expected in ((target0, (-1, -1)), (targetmax, (disp_h - 1, disp_w - 1))):
        cv = sv.unfoldBased2dDisparityCostVolume(NAMES[name], dev(target), dev(source), h_r, v_r, off)
        disp = host(sv.selected2dIndexToDisp(sv.extractSelected2dIndex(strat, cv), off))
        assert tuple(disp[v_r + 1, h_r + 1]) == expected

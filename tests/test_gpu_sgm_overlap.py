"""svh_stereo_match with a float matching function and the Score branch of sgmCostVolume (sgm.h:218-255, :360-404): the Left2Right pass
launched per band of rows on a second stream under the cost-volume kernel of the later bands (option "sgm_overlap": 0 off, the
default, or the number of bands) against the one-after-the-other order and the oracle: the same bits, whatever the row
count does to the bands (fewer rows than bands, a ragged last band, one row)."""
import numpy as np
import pytest

import oracle as so
from helpers import parallax_pair

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import libstevi_amd as sv  # noqa: E402

DEV = torch.device("cuda:0")


def run(func, tgt, src, h_r, v_r, D, overlap, **kw):
    sv.set_option(src, "sgm_overlap", overlap)
    try:
        return sv.stereoMatch(func, tgt, src, h_r, v_r, D, sgmDirections=8, P1=0.001, P2=0.01, Pout=100.0, want_sgm_cv=True, **kw)
    finally:
        sv.set_option(src, "sgm_overlap", 0)


@pytest.mark.parametrize("shape", [(67, 150, 32), (5, 90, 16), (1, 40, 8), (64, 64, 64), (33, 200, 100)])
@pytest.mark.parametrize("func", ["NCC", "ZNCC", "SSD", "SAD"])
@pytest.mark.parametrize("bands", [2, 8])
def test_overlapped_equals_ordered_and_oracle(shape, func, bands):
    H, W, D = shape
    src, tgt, _ = parallax_pair(H, W, max(H // 3, 1), H // 4, W // 4, 2, 6, 11)
    mf = getattr(sv.matchingFunctions, func)
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    a = run(mf, d_tgt, d_src, 2, 2, D, bands, want_cv=True)
    b = run(mf, d_tgt, d_src, 2, 2, D, 0, want_cv=True)
    assert torch.equal(a["cv"].view(torch.int32), b["cv"].view(torch.int32))
    assert torch.equal(a["sgm_cv"].view(torch.int32), b["sgm_cv"].view(torch.int32))
    assert torch.equal(a["disp"], b["disp"])
    if func in ("NCC", "ZNCC"):  # (the Cost branch does not take this path: the option must simply not matter there)
        # the oracle's recurrences over the cost volume the GPU built (its window sums run in another order than the oracle's, 1e-4
        # apart; the SGM over a given volume is exact): bit for bit
        exp = so.sgm(a["cv"].cpu().numpy(), 8, so.SCORE, 0.001, 0.01, (0, 0, 0, 0), 100.0)
        got = a["sgm_cv"].cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        ok = ~np.isnan(exp)
        assert np.array_equal(got[ok].view(np.uint32), exp[ok].view(np.uint32))


def test_overlap_repeated_calls_and_caller_volume(rng):
    """Several calls in a row reuse the second stream and its events; the caller's own cost / SGM arrays are written in place."""
    H, W, D = 96, 256, 64
    src, tgt, _ = parallax_pair(H, W, 30, 20, 40, 3, 9, 5)
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    ref = run(sv.matchingFunctions.ZNCC, d_tgt, d_src, 3, 3, D, 0, want_cv=True)
    for _ in range(4):
        got = run(sv.matchingFunctions.ZNCC, d_tgt, d_src, 3, 3, D, 3, want_cv=True)
        assert torch.equal(got["sgm_cv"].view(torch.int32), ref["sgm_cv"].view(torch.int32))
        assert torch.equal(got["cv"].view(torch.int32), ref["cv"].view(torch.int32))
        assert torch.equal(got["disp"], ref["disp"])

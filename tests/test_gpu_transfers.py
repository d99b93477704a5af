"""Host <-> device transfers behind host arrays (svh_transfer.hip): page-locked result memory (svh_host_alloc), the staged pipeline for
large pageable arrays, and the caches of released device / host blocks (trim on out-of-memory, double release).

The reference's functions take and return host arrays (cross_correlations.h:741-745, sgm.h:361-365), so every bit that reaches a kernel
or comes back from one goes through this path when the caller passes numpy arrays / Multidim::Array."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import libstevi_amd as sv  # noqa: E402
from libstevi_amd import _capi  # noqa: E402
from libstevi_amd import correlation as _c  # noqa: E402
import oracle as so  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MF = sv.matchingFunctions
DEV = torch.device("cuda:0")


def _ctx():
    return _c.context_for(None)


def _roundtrip(host_src, host_dst):
    """host_src -> device -> host_dst through svh_device_upload / svh_device_download (the copy engine stage_in / finish_out share)."""
    lib = _capi.load()
    ctx = _ctx()
    n = host_src.nbytes
    p = C.c_void_p()
    assert lib.svh_device_alloc(ctx, n, C.byref(p)) == _capi.OK
    try:
        assert lib.svh_device_upload(ctx, p, C.c_void_p(host_src.ctypes.data), n) == _capi.OK
        assert lib.svh_device_download(ctx, C.c_void_p(host_dst.ctypes.data), p, n) == _capi.OK
    finally:
        assert lib.svh_device_free(ctx, p) == _capi.OK


@pytest.mark.parametrize("nbytes", [1, 4095, (8 << 20) - 1, 8 << 20, (8 << 20) + 1, (4 << 20) * 9 + 12345, 200 << 20])
def test_pageable_round_trip_is_bit_exact(nbytes):
    """sizes around the pipeline's threshold (8 MB), a ragged last chunk, more chunks than ring slots"""
    rng = np.random.default_rng(nbytes % 1000)
    src = rng.integers(0, 256, nbytes, dtype=np.uint8)
    dst = np.zeros(nbytes, np.uint8)
    lib = _capi.load()
    assert lib.svh_host_is_pinned(C.c_void_p(src.ctypes.data), nbytes) == 0
    _roundtrip(src, dst)
    assert np.array_equal(src, dst)


def test_pinned_round_trip_and_mixed_directions():
    n = 64 << 20
    rng = np.random.default_rng(5)
    pinned = _c.host_empty((n,), np.uint8)
    lib = _capi.load()
    assert lib.svh_host_is_pinned(C.c_void_p(pinned.ctypes.data), n) == 1
    assert lib.svh_host_is_pinned(C.c_void_p(pinned.ctypes.data + 17), n - 17) == 1  # views into a block count
    pinned[:] = rng.integers(0, 256, n, dtype=np.uint8)
    pageable = np.zeros(n, np.uint8)
    _roundtrip(pinned, pageable)  # pinned up, staged down
    assert np.array_equal(pinned, pageable)
    back = _c.host_empty((n,), np.uint8)
    _roundtrip(pageable, back)    # staged up, pinned down
    assert np.array_equal(back, pinned)


def test_repeated_staged_copies_reuse_the_ring():
    """back-to-back uploads leave DMA reads of ring slots in flight: the next transfer must wait for each slot it refills"""
    rng = np.random.default_rng(9)
    for k in range(6):
        src = rng.integers(0, 256, (24 << 20) + 1000 * k, dtype=np.uint8)
        dst = np.empty_like(src)
        _roundtrip(src, dst)
        assert np.array_equal(src, dst)


def test_results_on_host_arrays_are_pinned_and_equal_the_device_results():
    """numpy in -> numpy out: the same bits as tensors in -> tensors out, and the large results sit in page-locked memory"""
    H, W, D, r = 96, 640, 64, 3
    rng = np.random.default_rng(3)
    src, tgt = rng.uniform(-1, 1, (H, W)).astype(np.float32), rng.uniform(-1, 1, (H, W)).astype(np.float32)
    cv = sv.unfoldBasedCostVolume(MF.CENSUS, tgt, src, r, r, D)
    assert isinstance(cv, np.ndarray) and cv.shape == (H, W, D)
    lib = _capi.load()
    assert cv.nbytes >= _c.PINNED_RESULTS_FROM and lib.svh_host_is_pinned(C.c_void_p(cv.ctypes.data), cv.nbytes) == 1
    strat = sv.matchFuncStrategy(MF.CENSUS)
    s = sv.sgmCostVolume(8, strat, cv, 0.001, 0.01, None, 100.0)
    disp = sv.selectedIndexToDisp(sv.extractSelectedIndex(strat, s), 0)
    d_cv = sv.unfoldBasedCostVolume(MF.CENSUS, torch.from_numpy(tgt).to(DEV), torch.from_numpy(src).to(DEV), r, r, D)
    d_s = sv.sgmCostVolume(8, strat, d_cv, 0.001, 0.01, None, 100.0)
    d_disp = sv.selectedIndexToDisp(sv.extractSelectedIndex(strat, d_s), 0)
    assert np.array_equal(cv, d_cv.cpu().numpy())
    assert np.array_equal(s.view(np.uint32), d_s.cpu().numpy().view(np.uint32))
    assert np.array_equal(disp, d_disp.cpu().numpy())
    # ... and the oracle's
    want = so.index_to_disp(so.extract_index(so.sgm(so.unfold_cost_volume(so.CENSUS, tgt, src, r, r, D), 8, so.COST, 0.001, 0.01, (0, 0, 0, 0), 100.0), so.COST))
    assert np.array_equal(disp, want)
    # a caller's write to the returned (page-locked) volume is seen by the next call: nothing is cached by content
    cv2 = cv.copy()            # pageable copy
    cv[:, :, 0] += 1000.0      # in place, in the pinned block
    cv2[:, :, 0] += 1000.0
    a = sv.extractSelectedIndex(strat, cv)
    b = sv.extractSelectedIndex(strat, cv2)
    assert np.array_equal(a, b) and not np.any(a == 0)


def test_non_dense_host_arrays_still_work():
    rng = np.random.default_rng(4)
    big = rng.uniform(-1, 1, (40, 130)).astype(np.float32)
    img = big[::2, 1::2]  # strided view
    got = sv.unfold(1, 1, img)
    assert np.array_equal(got, so.unfold(np.ascontiguousarray(img), 1, 1))


def test_host_block_cache_semantics():
    lib = _capi.load()
    p, q = C.c_void_p(), C.c_void_p()
    assert lib.svh_host_alloc(3 << 20, C.byref(p)) == _capi.OK and p.value
    assert lib.svh_host_free(p) == _capi.OK
    assert lib.svh_host_free(p) == _capi.ERR_INVALID_ARGUMENT          # released twice
    assert lib.svh_host_alloc(3 << 20, C.byref(q)) == _capi.OK
    assert q.value == p.value                                           # handed out again, still page-locked
    assert lib.svh_host_is_pinned(q, 3 << 20) == 1
    assert lib.svh_host_free(q) == _capi.OK
    assert lib.svh_host_cache_trim() == _capi.OK
    assert lib.svh_host_is_pinned(q, 1) == 0                            # given back to the system
    x = np.zeros(16, np.uint8)
    assert lib.svh_host_free(C.c_void_p(x.ctypes.data)) == _capi.ERR_INVALID_ARGUMENT  # not ours
    assert lib.svh_host_free(None) == _capi.OK


def test_device_block_double_release_is_refused():
    lib = _capi.load()
    ctx = _ctx()
    p, q, r = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.svh_device_alloc(ctx, 5 << 20, C.byref(p)) == _capi.OK
    assert lib.svh_device_free(ctx, p) == _capi.OK
    assert lib.svh_device_free(ctx, p) == _capi.ERR_INVALID_ARGUMENT
    assert lib.svh_device_alloc(ctx, 5 << 20, C.byref(q)) == _capi.OK and q.value == p.value
    assert lib.svh_device_alloc(ctx, 5 << 20, C.byref(r)) == _capi.OK and r.value != q.value  # never the same memory twice
    assert lib.svh_device_free(ctx, q) == _capi.OK and lib.svh_device_free(ctx, r) == _capi.OK


_OOM_SCRIPT = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
import torch
from libstevi_amd import _capi
from libstevi_amd import correlation as _c
lib = _capi.load()
ctx = _c.context_for(None)
free_b, total_b = torch.cuda.mem_get_info(0)
chunk = 8 << 30
# fill the device's cache of released blocks with most of the free memory ...
blocks = []
while (len(blocks) + 2) * chunk < free_b:
    p = C.c_void_p()
    assert lib.svh_device_alloc(ctx, chunk, C.byref(p)) == _capi.OK
    blocks.append(p)
free_now = torch.cuda.mem_get_info(0)[0]
if free_now > (7 << 30):                   # leave about 6 GB outside the cache
    p = C.c_void_p()
    assert lib.svh_device_alloc(ctx, free_now - (6 << 30), C.byref(p)) == _capi.OK
    blocks.append(p)
for p in blocks:
    assert lib.svh_device_free(ctx, p) == _capi.OK
free_b = torch.cuda.mem_get_info(0)[0]     # what is free with the cache full
held = 0
# ... then ask the WORKSPACE allocator (Scratch, through a call with host arrays) and torch for more than what is left outside the cache
import numpy as np
need = free_b - held + (2 << 30)           # cannot succeed unless the cache gives memory back
n = need // 4
H, W = 2048, 2048
D = int(n // (H * W))
cv = np.zeros((H, W, D), np.float32)
idx = _c.extractSelectedIndex(0, cv)       # stage_in: Scratch::get(need) -> out of memory -> cache released -> retry
assert idx.shape == (H, W)
print("ok", len(blocks), D)
"""


def test_workspace_allocation_takes_memory_back_from_the_cache():
    """idle cached svh_device_alloc blocks must not make a workspace hipMalloc fail.  Own process: it fills the
    device on purpose (SVH_DEVICE_CACHE_MB raised so that the cache may hold that much)."""
    env = dict(os.environ, SVH_DEVICE_CACHE_MB=str(400 << 10))
    out = subprocess.run([sys.executable, "-c", _OOM_SCRIPT % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.strip().startswith("ok")


def test_tiny_device_cache_cap_frees_at_once():
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); import torch; from libstevi_amd import _capi; from libstevi_amd import correlation as _c;"
            "lib = _capi.load(); ctx = _c.context_for(None); f0 = torch.cuda.mem_get_info(0)[0]; p = C.c_void_p();"
            "assert lib.svh_device_alloc(ctx, 1 << 30, C.byref(p)) == 0; f1 = torch.cuda.mem_get_info(0)[0]; assert f0 - f1 >= (1 << 30) - (64 << 20);"
            "assert lib.svh_device_free(ctx, p) == 0; f2 = torch.cuda.mem_get_info(0)[0]; assert f2 - f1 >= (1 << 30) - (64 << 20), (f0, f1, f2); print('ok')") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, SVH_DEVICE_CACHE_MB="1"), timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]

#!/usr/bin/env python3
"""Host <-> device transfer rates of the library's copy engine (svh_transfer.hip) for a 2 GiB array -- the size of a 1080p x 256 float
volume: page-locked host memory (svh_host_alloc: what the drop-in headers and the Python mirror put results in) against pageable
memory staged through the pinned ring, for a few SVH_COPY_THREADS settings (one process per setting: the ring is made once per context).
    python3 tools/bench_transfers.py            -> one JSON line per (memory, threads)"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(threads):
    import numpy as np
    import torch  # noqa: F401  (HIP runtime shared with the library)
    from libstevi_amd import _capi
    from libstevi_amd import correlation as _c
    lib = _capi.load()
    ctx = _c.context_for(None)
    n = 2 << 30
    dev = C.c_void_p()
    assert lib.svh_device_alloc(ctx, n, C.byref(dev)) == 0
    rows = []
    for kind in ("page-locked (svh_host_alloc)", "pageable (numpy)"):
        host = _c.host_empty((n,), np.uint8) if kind.startswith("page") and "locked" in kind else np.empty(n, np.uint8)
        host[:] = 7  # touch every page
        for direction, fn in (("host -> device", lambda: lib.svh_device_upload(ctx, dev, C.c_void_p(host.ctypes.data), n)),
                              ("device -> host", lambda: lib.svh_device_download(ctx, C.c_void_p(host.ctypes.data), dev, n))):
            assert fn() == 0
            t = []
            for _ in range(3):
                t0 = time.perf_counter()
                assert fn() == 0
                t.append(time.perf_counter() - t0)
            rows.append({"host_memory": kind, "direction": direction, "bytes": n, "copy_threads": threads if "pageable" in kind else None,
                         "ms": round(min(t) * 1e3, 1), "GBps": round(n / min(t) / 1e9, 1)})
        del host
    lib.svh_device_free(ctx, dev)
    for r in rows:
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(int(sys.argv[1]))
    else:
        for threads in (1, 2, 4, 8):
            subprocess.run([sys.executable, os.path.abspath(__file__), str(threads)], env=dict(os.environ, SVH_COPY_THREADS=str(threads)), check=True)

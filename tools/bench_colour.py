#!/usr/bin/env python3
"""unfoldBasedCostVolume on colour (3-channel) images next to the grey image of the same size, per-kernel breakdown: the column-sum kernel
stages the channels as window rows of their own (round 4f; the per-voxel kernel before: 52.7 ms for NCC 7x7 at 1080p x 128)."""
import json, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import libstevi_amd as sv
MF = sv.matchingFunctions
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
for (H, W, C, r, D) in [(480, 640, 3, 2, 64), (1080, 1920, 3, 3, 128), (1080, 1920, 1, 3, 128)]:
    shape = (H, W, C) if C > 1 else (H, W)
    l = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32)).to(dev)
    r_ = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32)).to(dev)
    for func in (MF.NCC, MF.ZNCC, MF.SAD, MF.CENSUS):
        fn = lambda: sv.unfoldBasedCostVolume(func, l, r_, r, r, D)
        fn(); torch.cuda.synchronize()
        sv.profile_reset(l); sv.profile_enable(l, True)
        t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        sv.profile_enable(l, False)
        prof = sv.profile_collect(l)
        print(json.dumps({"shape": [H, W, C, D], "window": 2 * r + 1, "func": func.name, "ms": round(dt * 1e3, 3), "kernel_ms": {k: round(v[0] / 3, 3) for k, v in prof.items()}}), flush=True)

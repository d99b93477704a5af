#!/usr/bin/env python3
"""hiearchicalTruncatedCostVolume on RGB images (ZNCC, depth 1 - 3) at 1080p x 320 and 480x640 x 160: the colour form of tools/bench_hierarchical.py."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libstevi_amd as sv
from helpers import parallax_pair
MF = sv.matchingFunctions
dev = torch.device("cuda:0")
for name, W, H, r, disp_w in [("HD 7x7 RGB", 1920, 1080, 3, 320), ("SD 5x5 RGB", 640, 480, 2, 160)]:
    src, tgt, _ = parallax_pair(H, W, H // 3, H // 3, W // 3, disp_w // 16, disp_w // 4, 11)
    rng = np.random.default_rng(1)
    gains = np.array([1.0, 0.8, 1.2], np.float32)
    src3 = (src[..., None] * gains + 0.01 * rng.uniform(-1, 1, src.shape + (3,))).astype(np.float32)
    tgt3 = (tgt[..., None] * gains + 0.01 * rng.uniform(-1, 1, tgt.shape + (3,))).astype(np.float32)
    d_src, d_tgt = torch.from_numpy(src3).to(dev), torch.from_numpy(tgt3).to(dev)
    for depth in (1, 2, 3):
        run = lambda: sv.hiearchicalTruncatedCostVolume(MF.ZNCC, depth, d_tgt, d_src, r, r, disp_w, 2)
        run(); torch.cuda.synchronize()
        sv.profile_reset(d_src); sv.profile_enable(d_src, True)
        n = 5
        t0 = time.perf_counter()
        for _ in range(n): res = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        sv.profile_enable(d_src, False)
        prof = sv.profile_collect(d_src)
        print(json.dumps({"row": name, "depth": depth, "ms": round(dt * 1e3, 3), "kernel_ms": {k: round(v[0] / n, 3) for k, v in prof.items()}}), flush=True)

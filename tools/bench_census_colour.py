#!/usr/bin/env python3
"""census + SGM-8 on RGB images at 1080p x 256 (the census of a colour window: channels interleaved, unfold.h:180), per-kernel times."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libstevi_amd as sv
from helpers import parallax_pair
MF = sv.matchingFunctions
dev = torch.device("cuda:0")
src, tgt, _ = parallax_pair(1080, 1920, 320, 320, 380, 8, 64, 3)
rng = np.random.default_rng(0)
mix = np.array([1.0, 0.8, 0.6], np.float32)
src3 = src[..., None] * mix + rng.random((1080, 1920, 3), dtype=np.float32) * 0.01
tgt3 = tgt[..., None] * mix + rng.random((1080, 1920, 3), dtype=np.float32) * 0.01
ds, dt_ = torch.from_numpy(src3.astype(np.float32)).to(dev), torch.from_numpy(tgt3.astype(np.float32)).to(dev)
for r in (2, 3, 4):
    f = lambda: sv.stereoMatch(MF.CENSUS, dt_, ds, r, r, 256, sgmDirections=8)
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): o = f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    sv.profile_reset(ds); sv.profile_enable(ds, True)
    for _ in range(5): f()
    torch.cuda.synchronize()
    sv.profile_enable(ds, False)
    prof = sv.profile_collect(ds)
    print(json.dumps({"rgb_window": 2 * r + 1, "words": ((2 * r + 1) ** 2 * 3 - 1) // 32, "ms": round(dt * 1e3, 3), "kernel_ms": {k: round(v[0] / 5, 3) for k, v in prof.items()}}), flush=True)

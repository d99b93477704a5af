#!/usr/bin/env python3
"""ZSAD (the one float function without a column-sum form) on the reference benchmark's shapes: unfoldBasedCostVolume alone, device arrays."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libstevi_amd as sv
from helpers import parallax_pair
MF = sv.matchingFunctions
dev = torch.device("cuda:0")
for res, H, W, r, D in (("StandardDefinition", 480, 640, 2, 160), ("HighDefinition", 1080, 1920, 3, 320)):
    src, tgt, _ = parallax_pair(H, W, H // 3, H // 3, W // 3, D // 16, D // 4, 11)
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    for func in (MF.ZSAD, MF.SAD):
        run = lambda: sv.unfoldBasedCostVolume(func, d_tgt, d_src, r, r, D)
        cv = run(); torch.cuda.synchronize()
        sv.profile_reset(d_src); sv.profile_enable(d_src, True)
        n = 5
        t0 = time.perf_counter()
        for _ in range(n): cv = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        sv.profile_enable(d_src, False)
        prof = sv.profile_collect(d_src)
        print(json.dumps({"row": f"{res} {H}x{W} - {2*r+1}x{2*r+1} - disp {D} - {func.name}", "ms": round(dt * 1e3, 3), "kernel_ms": {k: round(v[0] / n, 3) for k, v in prof.items()},
                          "checksum": float(np.nansum(cv[::7, ::5].float().cpu().numpy().astype(np.float64)))}), flush=True)

#!/usr/bin/env python3
"""census 9x9 + SGM-8 at 1080p: the RightToLeft specialisation of the FP4 sweep (compile-time disparity count) against the general FP4 kernel."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import libstevi_amd as sv
from helpers import parallax_pair
MF = sv.matchingFunctions
dev = torch.device("cuda:0")
src, tgt, _ = parallax_pair(1080, 1920, 320, 320, 380, 8, 64, 3)
ds, dt_ = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
sv.set_option(ds, "census_winner_shortcut", 0)
for D in [int(a) for a in sys.argv[1:]] or [64, 96, 128, 160, 192, 256, 320, 512]:
    for rl in (1, 0):
        sv.set_test_option(ds, "census_sweep_rl", rl)
        f = lambda: sv.stereoMatch(MF.CENSUS, dt_, ds, 4, 4, D, sgmDirections=8)
        f(); f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): o = f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        sv.profile_reset(ds); sv.profile_enable(ds, True)
        for _ in range(5): f()
        torch.cuda.synchronize()
        sv.profile_enable(ds, False)
        prof = sv.profile_collect(ds)
        print(json.dumps({"D": D, "census_sweep_rl": rl, "ms": round(dt * 1e3, 4), "kernel_ms": {k: round(v[0] / 5, 4) for k, v in prof.items() if "sweep" in k}}), flush=True)
sv.set_test_option(ds, "census_sweep_rl", 1); sv.set_option(ds, "census_winner_shortcut", 1)

#!/usr/bin/env python3
"""census + SGM-8 at 1080p x 256 for window radii 4..6 (2, 4, 5 words per descriptor), winner shortcut on and off."""
import sys, time, json
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, libstevi_amd as sv
from helpers import parallax_pair
MF = sv.matchingFunctions
src, tgt, _ = parallax_pair(1080, 1920, 320, 320, 380, 8, 64, 3)
dev = torch.device("cuda:0")
ds, dt_ = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
for r in (4, 5, 6, 7):
    for short in (1, 0):
        sv.set_option(ds, "census_winner_shortcut", short)
        f = lambda: sv.stereoMatch(MF.CENSUS, dt_, ds, r, r, 256, sgmDirections=8)
        f(); f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): o = f()
        torch.cuda.synchronize()
        print(json.dumps({"r": r, "shortcut": short, "ms": round((time.perf_counter() - t0) * 100, 3), "sum": int(o["disp"].double().sum().item())}), flush=True)
sv.set_option(ds, "census_winner_shortcut", 1)
# disparity counts that are no multiple of 32 (the matrix cores take the leading multiple, the vector ALU the rest)
for D in (256, 250, 200, 128, 120, 100):
    for short in (1, 0):
        sv.set_option(ds, "census_winner_shortcut", short)
        f = lambda: sv.stereoMatch(MF.CENSUS, dt_, ds, 4, 4, D, sgmDirections=8)
        f(); f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): o = f()
        torch.cuda.synchronize()
        print(json.dumps({"D": D, "shortcut": short, "ms": round((time.perf_counter() - t0) * 100, 3), "sum": int(o["disp"].double().sum().item())}), flush=True)
sv.set_option(ds, "census_winner_shortcut", 1)

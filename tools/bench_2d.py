#!/usr/bin/env python3
"""Timing of the 2-D disparity (optical-flow style) volume path: unfoldBased2dDisparityCostVolume -> extractSelected2dIndex
-> selected2dIndexToDisp -> truncatedBidirectionaCostVolume -> refineDisp2dCostInterpolation on device arrays, per-kernel breakdown, plus the oracle on the box's host cores for a row band."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import libstevi_amd as sv
import oracle as so
from helpers import parallax_pair
MF = sv.matchingFunctions
dev = torch.device("cuda:0")
for name, W, H, rng0, rng1, func, r in [("ZNCC 7x7, 1920x1080, 9 x 33 offsets", 1920, 1080, (-4, 4), (-16, 16), MF.ZNCC, 3),
                                        ("census 9x9, 1920x1080, 9 x 33 offsets", 1920, 1080, (-4, 4), (-16, 16), MF.CENSUS, 4),
                                        ("SAD 5x5, 640x480, 17 x 17 offsets", 640, 480, (-8, 8), (-8, 8), MF.SAD, 2)]:
    src, tgt, _ = parallax_pair(H, W, H // 4, H // 4, W // 4, 2, 5, 9)
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    off = sv.searchOffset2(rng0[0], rng0[1], rng1[0], rng1[1])
    strat = sv.matchFuncStrategy(func)
    def chain():
        cv = sv.unfoldBased2dDisparityCostVolume(func, d_tgt, d_src, r, r, off)
        idx = sv.extractSelected2dIndex(strat, cv)
        disp = sv.selected2dIndexToDisp(idx, off)
        # the --refine step of examples/stereo-match (main.cpp:198-210): 3x3 truncated volume + equiangular refinement
        sv.refineDisp2dCostInterpolation(sv.InterpolationKernel.Equiangular, sv.truncatedBidirectionaCostVolume(cv, idx, 1, 1), disp)
        return disp
    chain(); torch.cuda.synchronize()
    sv.profile_reset(d_src); sv.profile_enable(d_src, True)
    n = 5
    t0 = time.perf_counter()
    for _ in range(n): disp = chain()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    sv.profile_enable(d_src, False)
    prof = sv.profile_collect(d_src)
    nvox = W * H * (rng0[1] - rng0[0] + 1) * (rng1[1] - rng1[0] + 1)
    rows = 64
    t0 = time.perf_counter()
    so.extract_index_2d(so.unfold_cost_volume_2d(int(func), tgt[:rows], src[:rows], r, r, rng0, rng1), int(strat))
    cpu = rows * W * nvox / (W * H) / (time.perf_counter() - t0) / 1e6
    print(json.dumps({"chain": name, "ms": round(dt * 1e3, 3), "Mvoxels_per_s": round(nvox / dt / 1e6), "cpu_oracle_Mvoxels_per_s": round(cpu, 1),
                      "cpu_threads": so.num_threads(), "kernel_ms": {k: round(v[0] / n, 3) for k, v in prof.items()}, "checksum": int(disp.sum().item())}), flush=True)

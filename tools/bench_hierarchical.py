#!/usr/bin/env python3
"""Timing of hiearchicalTruncatedCostVolume on device arrays at the rows of the reference's own benchmark
(benchmarkCrossCorrelationAlgorithms.cpp:362-402: ZNCC, depth 1-3, 1920x1080 7x7 disp 320; 640x480 5x5 disp 160), per-kernel
breakdown, with the oracle on the box's host cores next to it.  Mdisparities/s counts W*H*disp_w (the search range covered)."""
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
for name, W, H, r, disp_w, func in [("HD 7x7 disp 320 zncc", 1920, 1080, 3, 320, MF.ZNCC), ("SD 5x5 disp 160 zncc", 640, 480, 2, 160, MF.ZNCC),
                                    ("HD 9x9 disp 320 census", 1920, 1080, 4, 320, MF.CENSUS)]:
    src, tgt, _ = parallax_pair(H, W, H // 3, H // 3, W // 3, disp_w // 16, disp_w // 4, 11)
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    for depth in (1, 2, 3):
        run = lambda: sv.hiearchicalTruncatedCostVolume(func, depth, d_tgt, d_src, r, r, disp_w, 2)
        run(); torch.cuda.synchronize()
        sv.profile_reset(d_src); sv.profile_enable(d_src, True)
        n = 5
        t0 = time.perf_counter()
        for _ in range(n): res = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        sv.profile_enable(d_src, False)
        prof = sv.profile_collect(d_src)
        t0 = time.perf_counter()
        etcv, edisp = so.hierarchical_truncated_cv(int(func), depth, tgt, src, r, r, disp_w, 2)
        cpu = time.perf_counter() - t0
        agree = float((res.disp_estimate.cpu().numpy() == edisp).mean())
        print(json.dumps({"row": name, "depth": depth, "ms": round(dt * 1e3, 3), "Mdisp_per_s": round(W * H * disp_w / dt / 1e6), "cpu_oracle_ms": round(cpu * 1e3, 1),
                          "cpu_threads": so.num_threads(), "estimates_equal_to_oracle": round(agree, 6),
                          "kernel_ms": {k: round(v[0] / n, 3) for k, v in prof.items()}}), flush=True)

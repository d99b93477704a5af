#!/usr/bin/env python3
"""2-D disparity volumes of RGB images (unfoldBased2dDisparityCostVolume, ZNCC 7x7, 9 x 33 offsets at 1080p): several vertical offsets
per launch of the column-sum kernel (fold_2d_offsets, round 5 for colour images) against a launch per vertical offset."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import libstevi_amd as sv
MF = sv.matchingFunctions
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
for name, C, func, r, r0, r1 in [("RGB ZNCC 7x7, 9 x 33", 3, MF.ZNCC, 3, (-4, 4), (-16, 16)), ("RGB SAD 5x5, 17 x 17", 3, MF.SAD, 2, (-8, 8), (-8, 8)),
                                 ("grey ZNCC 7x7, 9 x 33", 1, MF.ZNCC, 3, (-4, 4), (-16, 16))]:
    shape = (1080, 1920, C) if C > 1 else (1080, 1920)
    a, b = torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32)).to(dev), torch.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32)).to(dev)
    off = sv.searchOffset2(r0[0], r0[1], r1[0], r1[1])
    row = {"volume": name + " at 1920x1080"}
    for fold in (1, 0):
        sv.set_test_option(a, "fold_2d_offsets", fold)
        f = lambda: sv.unfoldBased2dDisparityCostVolume(func, a, b, r, r, off)
        out = f(); out = f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): out = f()
        torch.cuda.synchronize()
        row["ms_folded" if fold else "ms_launch_per_offset"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
        row["sum_folded" if fold else "sum_per_offset"] = float(out.double().sum().item())
    sv.set_test_option(a, "fold_2d_offsets", 1)
    print(json.dumps(row), flush=True)

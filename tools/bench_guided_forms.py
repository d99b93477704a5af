#!/usr/bin/env python3
"""computeGuidedCV inside hiearchicalTruncatedCostVolume (ZNCC) per staging form -- svh_test_set_option "guided_shared" 2: a block of 256
pixels (round 4), 3: a wave of 64 pixels (round 5), 1: the default's choice by grid size -- over image sizes and windows; `same_bits`:
the truncated volume equals the block form's bit for bit.   python3 tools/bench_guided_forms.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libstevi_amd as sv
from helpers import parallax_pair
MF = sv.matchingFunctions
dev = torch.device("cuda:0")
for name, W, H, r, disp_w, flat in [("HD 7x7", 1920, 1080, 3, 320, False), ("HD 5x5", 1920, 1080, 2, 320, False), ("HD 3x3", 1920, 1080, 1, 320, False), ("1280x720 7x7", 1280, 720, 3, 200, False), ("1280x720 5x5", 1280, 720, 2, 200, False), ("SD 7x7", 640, 480, 3, 160, False), ("SD 5x5", 640, 480, 2, 160, False), ("SD 3x3", 640, 480, 1, 160, False)]:
    src, tgt, _ = parallax_pair(H, W, H // 3, H // 3, W // 3, disp_w // 16, disp_w // 16 if flat else disp_w // 4, 11)
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    ref = None
    for shared in (2, 3, 1):
        sv.set_test_option(d_src, "guided_shared", shared)
        for depth in (1, 3):
            run = lambda: sv.hiearchicalTruncatedCostVolume(MF.ZNCC, depth, d_tgt, d_src, r, r, disp_w, 2)
            res = run(); torch.cuda.synchronize()
            key = (depth,)
            sv.profile_reset(d_src); sv.profile_enable(d_src, True)
            n = 10
            t0 = time.perf_counter()
            for _ in range(n): res = run()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            sv.profile_enable(d_src, False)
            prof = sv.profile_collect(d_src)
            out = res.truncated_cost_volume.cpu().numpy().view(np.uint32)
            if shared == 2: ref = ref or {}; ref[depth] = out
            print(json.dumps({"row": name, "form": shared, "depth": depth, "ms": round(dt * 1e3, 3), "guided_ms": round(prof["guided_fused"][0] / n, 3),
                              "same_bits": bool(np.array_equal(out, ref[depth]))}), flush=True)
    sv.set_test_option(d_src, "guided_shared", 1)

#!/usr/bin/env python3
"""Score-branch SGM at the reference's own benchmark shape (benchmarkCrossCorrelationAlgorithms.cpp:533-545: 480x640, 160 disparities, ZNCC
5x5, 4 / 8 directions): the fused call and the per-function chain, per-kernel event times, for each form of the aggregation
(sgm_score_fused 0: a launch per pass, 2: bands of rows, 1: the library's choice).  argv: forms to run (default 1 0 2)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import libstevi_amd as sv
from helpers import parallax_pair
MF = sv.matchingFunctions
dev = torch.device("cuda:0")
H, W, r, D = 480, 640, 2, 160
src, tgt, _ = parallax_pair(H, W, H // 3, H // 3, W // 3, D // 16, D // 4, 11)
d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
forms = [int(a) for a in sys.argv[1:]] or [1, 0, 2]
ref = None
for form in forms:
    sv.set_option(d_src, "sgm_score_fused", form)
    for ndir in (4, 8):
        fn = lambda: sv.stereoMatch(MF.ZNCC, d_tgt, d_src, r, r, D, sgmDirections=ndir, P1=0.001, P2=0.01, Pout=100.0)["disp"]  # noqa: E731
        for _ in range(3): out = fn()
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n): out = fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        sv.profile_reset(d_src); sv.profile_enable(d_src, True)
        for _ in range(5): fn()
        torch.cuda.synchronize(); sv.profile_enable(d_src, False)
        prof = sv.profile_collect(d_src)
        key = (ndir,)
        if ref is None: ref = {}
        same = None
        if key in ref: same = bool(torch.equal(ref[key], out))
        else: ref[key] = out.clone()
        print(json.dumps({"shape": [H, W, D], "func": "ZNCC 5x5", "directions": ndir, "sgm_score_fused": form, "ms_fused_call": round(ms, 4),
                          "Mdisparities_per_s": round(H * W * D / ms / 1e3, 1), "same_map_as_first_form": same,
                          "kernel_ms": {k: [round(v[0] / 5, 4), v[1] // 5] for k, v in prof.items()}}), flush=True)
sv.set_option(d_src, "sgm_score_fused", 1)

#!/usr/bin/env python3
"""Kernel breakdown of benchmarkLocalAlgorithmWithCompressor's chain (benchmarkCrossCorrelationAlgorithms.cpp:461-531): ZNCC on the 17
superpixel features of the GrPix17R3 / GrPix17R4 masks, full volume + extractSelectedIndex, 480x640 x 160 and 1080x1920 x 320."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libstevi_amd as sv
from helpers import parallax_pair
MF = sv.matchingFunctions
dev = torch.device("cuda:0")
for res, H, W, D in (("StandardDefinition", 480, 640, 160), ("HighDefinition", 1080, 1920, 320)):
    src, tgt, _ = parallax_pair(H, W, H // 3, H // 3, W // 3, D // 16, D // 4, 11)
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    for gen in ("GrPix17R3Filter", "GrPix17R4Filter"):
        comp = sv.UnFoldCompressor(getattr(sv.CompressorGenerators, gen)())
        def chain():
            cv = sv.unfoldBasedCostVolumeCompressed(MF.ZNCC, d_tgt, d_src, comp, D)
            return sv.selectedIndexToDisp(sv.extractSelectedIndex(sv.matchFuncStrategy(MF.ZNCC), cv), 0)
        chain(); torch.cuda.synchronize()
        sv.profile_reset(d_src); sv.profile_enable(d_src, True)
        n = 5
        t0 = time.perf_counter()
        for _ in range(n): chain()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        sv.profile_enable(d_src, False)
        prof = sv.profile_collect(d_src)
        print(json.dumps({"row": f"{res} {H}x{W} - {gen[:-6]} - disp {D} - zncc", "ms": round(dt * 1e3, 3), "kernel_ms": {k: round(v[0] / n, 3) for k, v in prof.items()}}), flush=True)

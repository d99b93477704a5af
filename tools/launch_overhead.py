#!/usr/bin/env python3
"""How much do the per-kernel hipEvents and the launch gaps cost on the headline step?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import libstevi_amd as sv
from helpers import parallax_pair
src, tgt, _ = parallax_pair(1080, 1920, 320, 320, 380, 8, 64, 3)
dev = torch.device("cuda:0")
d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
def step():
    return sv.stereoMatch(sv.matchingFunctions.CENSUS, d_tgt, d_src, 4, 4, 256, sgmDirections=8)
for prof in (False, True, False, True):
    sv.profile_enable(d_src, prof)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    sv.profile_reset(d_src)
    print(f"profiling events {'on ' if prof else 'off'}: {dt*1e3:.4f} ms/step")
g = torch.cuda.CUDAGraph()
sv.profile_enable(d_src, False)
try:
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): out = step()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize()
    print(f"hipGraph replay: {(time.perf_counter()-t0)/200*1e3:.4f} ms/step, checksum {int(out['disp'].sum())}")
except Exception as e:
    print("graph capture failed:", type(e).__name__, str(e)[:300])

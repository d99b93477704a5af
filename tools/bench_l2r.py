import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import libstevi_amd as sv
from helpers import parallax_pair
dev = torch.device("cuda:0")
src, tgt, _ = parallax_pair(1080, 1920, 320, 320, 380, 8, 64, seed=3)
l, r = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
for mode in (1, 0):
    sv.set_option(l, "census_sweep", mode)
    for _ in range(5):
        sv.stereoMatch(sv.matchingFunctions.CENSUS, l, r, 4, 4, 256, dDir=sv.dispDirection.LeftToRight, sgmDirections=8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        out = sv.stereoMatch(sv.matchingFunctions.CENSUS, l, r, 4, 4, 256, dDir=sv.dispDirection.LeftToRight, sgmDirections=8)
    torch.cuda.synchronize()
    print("LeftToRight engine", mode, "ms/step", (time.perf_counter() - t0) * 10, "checksum", int(out["disp"].sum().item()))

import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import libstevi_amd as sv
from helpers import parallax_pair
MF = sv.matchingFunctions
dev = torch.device("cuda:0")
W, H, r, disp_w = 1920, 1080, 3, 320
src, tgt, _ = parallax_pair(H, W, H // 3, H // 3, W // 3, disp_w // 16, disp_w // 4, 11)
d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
for _ in range(6):
    res = sv.hiearchicalTruncatedCostVolume(MF.ZNCC, 1, d_tgt, d_src, r, r, disp_w, 2)
torch.cuda.synchronize()

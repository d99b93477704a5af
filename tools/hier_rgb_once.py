#!/usr/bin/env python3
"""one hierarchical call per depth on a 1080p RGB pair (to trace: bash tools/kernel_trace_cmd.sh guided_ tools/hier_rgb_once.py)"""
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
rng = np.random.default_rng(1)
gains = np.array([1.0, 0.8, 1.2], np.float32)
src3 = (src[..., None] * gains + 0.01 * rng.uniform(-1, 1, src.shape + (3,))).astype(np.float32)
tgt3 = (tgt[..., None] * gains + 0.01 * rng.uniform(-1, 1, tgt.shape + (3,))).astype(np.float32)
d_src, d_tgt = torch.from_numpy(src3).to(dev), torch.from_numpy(tgt3).to(dev)
for _ in range(4):
    sv.hiearchicalTruncatedCostVolume(MF.ZNCC, 3, d_tgt, d_src, r, r, disp_w, 2)
torch.cuda.synchronize()

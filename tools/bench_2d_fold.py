#!/usr/bin/env python3
"""Time of unfoldBased2dDisparityCostVolume alone (cost kernel, per-kernel events) for a list of search ranges, folded offsets against a
launch per vertical offset: tools/bench_2d_fold.py [FUNC h_r Dh Dw]..."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import libstevi_amd as sv
from helpers import parallax_pair
MF = sv.matchingFunctions
dev = torch.device("cuda:0")
args = sys.argv[1:] or ["ZNCC", "3", "9", "33", "ZNCC", "3", "9", "32", "ZNCC", "3", "9", "48", "NCC", "5", "5", "65", "SAD", "2", "17", "17"]
W, H = 1920, 1080
src, tgt, _ = parallax_pair(H, W, H // 4, H // 4, W // 4, 2, 5, 9)
d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
for k in range(0, len(args), 4):
    func, r, Dh, Dw = getattr(MF, args[k]), int(args[k + 1]), int(args[k + 2]), int(args[k + 3])
    off = sv.searchOffset2(-(Dh // 2), Dh - 1 - Dh // 2, -(Dw // 2), Dw - 1 - Dw // 2)
    out = {"func": args[k], "window": 2 * r + 1, "Dh": Dh, "Dw": Dw, "Mvoxels": W * H * Dh * Dw / 1e6}
    for fold in (0, 1):
        sv.set_test_option(d_src, "fold_2d_offsets", fold)
        cv = sv.unfoldBased2dDisparityCostVolume(func, d_tgt, d_src, r, r, off); torch.cuda.synchronize()
        sv.profile_reset(d_src); sv.profile_enable(d_src, True)
        n = 5
        for _ in range(n): cv = sv.unfoldBased2dDisparityCostVolume(func, d_tgt, d_src, r, r, off)
        torch.cuda.synchronize()
        sv.profile_enable(d_src, False)
        prof = sv.profile_collect(d_src)
        ms = prof["cost_volume_tiled"][0] / n
        out["fold" if fold else "per_offset"] = {"ms": round(ms, 3), "TB_per_s_written": round(W * H * Dh * Dw * 4 / ms / 1e9, 2)}
        del cv
    sv.set_test_option(d_src, "fold_2d_offsets", 1)
    print(json.dumps(out), flush=True)

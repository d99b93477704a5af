#!/usr/bin/env python3
"""The reference benchmark's call chain through the per-function entry points on DEVICE arrays (volumes materialised
in HBM between calls, as the drop-in C++ headers do minus the PCIe copies):
    unfoldBasedCostVolume -> sgmCostVolume -> extractSelectedIndex -> selectedIndexToDisp"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import libstevi_amd as sv
from helpers import parallax_pair
MF = sv.matchingFunctions
cases = [("census 9x9 + SGM-8 (Cost)", 1920, 1080, 256, MF.CENSUS, 4), ("NCC 11x11 + SGM-8 (Score)", 1920, 1080, 256, MF.NCC, 5),
         ("SAD 5x5 + SGM-8 (Cost, float costs)", 1920, 1080, 256, MF.SAD, 2)]
# other disparity ranges: python tools/bench_api_chain.py 128 64
if len(sys.argv) > 1:
    cases = [(f"{n}, D = {d}", w, h, int(d), f, r) for d in sys.argv[1:] for (n, w, h, _, f, r) in cases]
    keep = os.environ.get("SVH_CHAIN_PLAIN") != "1"  # SVH_CHAIN_PLAIN=1: no statements carried between the calls (a caller that does not opt in)
else:
    keep = True
dev = torch.device("cuda:0")
for name, W, H, D, func, r in cases:
    src, tgt, _ = parallax_pair(H, W, 320, 320, 380, 8, min(64, D // 2), 3)
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    strat = sv.matchFuncStrategy(func)
    def chain():
        cv = sv.unfoldBasedCostVolume(func, d_tgt, d_src, r, r, D, keep_minima=keep)
        sgm = sv.sgmCostVolume(8, strat, cv, 0.001, 0.01, None, 100.0, keep_winner=keep)
        return sv.selectedIndexToDisp(sv.extractSelectedIndex(strat, sgm), 0)
    chain(); torch.cuda.synchronize()
    sv.profile_reset(d_src); sv.profile_enable(d_src, True)
    n = 5
    t0 = time.perf_counter()
    for _ in range(n): disp = chain()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    sv.profile_enable(d_src, False)
    prof = sv.profile_collect(d_src)
    print(json.dumps({"chain": name + ("" if keep else " (no statements carried between the calls)"), "ms": round(dt * 1e3, 3), "Mdisp_per_s": round(W * H * D / dt / 1e6), "kernel_ms": {k: round(v[0] / n, 3) for k, v in prof.items()},
                      "checksum": int(disp.sum().item())}), flush=True)

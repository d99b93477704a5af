"""Score-branch SGM alone on a random (H, W, D) volume: both forms (the fused downward sweep / one launch per pass), timed with
events; run under `rocprofv3 --kernel-trace --stats` for the per-launch durations.
  python tools/sgm_score_probe.py [W H D]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

import libstevi_amd as sv
import oracle as so

W, H, D = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (4096, 2160, 256)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
cv = torch.rand((H, W, D), device=dev, generator=g) * 2 - 1
res = {"shape": [W, H, D]}
outs = {}
for fused in (0, 2):
    sv.set_test_option(cv, "sgm_score_fused", fused)
    for it in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = sv.sgmCostVolume(8, so.SCORE, cv, 0.001, 0.01, None, 100.0)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    res[("per_pass_ms", "", "band_launches_ms")[fused]] = round(ms, 3)
    outs[fused] = out.view(torch.int32)[::7, ::5].clone()
    del out
sv.set_option(cv, "sgm_score_fused", 1)
res["sampled_equal"] = bool(torch.equal(outs[0], outs[1]))
print(json.dumps(res))

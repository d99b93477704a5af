import json, os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import libstevi_amd as sv
from helpers import parallax_pair
MF = sv.matchingFunctions
dev = torch.device("cuda:0")
for (H, W, r, D, ndir) in [(480, 640, 2, 160, 8), (480, 640, 2, 160, 4), (480, 640, 2, 128, 8), (480, 640, 2, 256, 8), (1080, 1920, 3, 320, 8), (1080, 1920, 3, 320, 4), (1080, 1920, 3, 256, 8), (1080, 1920, 3, 512, 8)]:
    src, tgt, _ = parallax_pair(H, W, H // 3, H // 3, W // 3, max(D // 16, 1), D // 4, 11)
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    fn = lambda: sv.stereoMatch(MF.ZNCC, d_tgt, d_src, r, r, D, sgmDirections=ndir, P1=0.001, P2=0.01, Pout=100.0)["disp"]
    fn(); torch.cuda.synchronize()
    sv.profile_reset(d_src); sv.profile_enable(d_src, True)
    n = 5
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    sv.profile_enable(d_src, False)
    prof = sv.profile_collect(d_src)
    print(json.dumps({"shape": [H, W, D], "ndir": ndir, "ms": round(dt * 1e3, 3), "kernel_ms": {k: round(v[0] / n, 3) for k, v in prof.items()}}), flush=True)

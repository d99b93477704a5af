import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import libstevi_amd as sv
import oracle as so
from libstevi_amd import matchingFunctions as MF
DEV = torch.device("cuda:0")
SHAPES = [(37, 90), (90, 37), (64, 64), (1, 70), (70, 1), (130, 200), (300, 70), (33, 33), (200, 130), (17, 400), (256, 256), (65, 129)]
D, r = 32, 4
def run(d_tgt, d_src, tiles, Pout):
    sv.set_option(d_tgt, "census_tiles", tiles); sv.set_option(d_tgt, "census_winner_shortcut", 0)
    return sv.stereoMatch(MF.CENSUS, d_tgt, d_src, r, r, D, sgmDirections=8, P1=0.3, P2=0.9, want_keys=True, Pout=Pout, refineKernel=so.GAUSSIAN, refine_h_radius=r, refine_v_radius=r)
for (H, W) in SHAPES:
    rng = np.random.default_rng(H * 1000 + W + D)
    src = rng.uniform(-1, 1, (H, W)).astype(np.float32); tgt = rng.uniform(-1, 1, (H, W)).astype(np.float32)
    d_src, d_tgt = torch.from_numpy(src).to(DEV), torch.from_numpy(tgt).to(DEV)
    for Pout in (100.0, 0.0, 7.0):
        a = run(d_tgt, d_src, 1, Pout); b = run(d_tgt, d_src, 0, Pout)
        ka = a["keys"].cpu().numpy().view(np.uint64); kb = b["keys"].cpu().numpy().view(np.uint64)
        bad = np.argwhere(ka != kb)
        if len(bad):
            print((H, W), "Pout", Pout, "bad", len(bad), "disp bad", int((a["disp"] != b["disp"]).sum().item()))
            print(bad[:30].tolist())
            hi_a = (ka >> np.uint64(32)).astype(np.int64); hi_b = (kb >> np.uint64(32)).astype(np.int64)
            print("diffs", (hi_a - hi_b)[ka != kb][:30].tolist())
            a2 = run(d_tgt, d_src, 1, Pout); ka2 = a2["keys"].cpu().numpy().view(np.uint64)
            print("rerun tiles: same as first tiles run:", bool((ka2 == ka).all()), " same as maps:", bool((ka2 == kb).all()))
            b2 = run(d_tgt, d_src, 0, Pout); kb2 = b2["keys"].cpu().numpy().view(np.uint64)
            print("rerun maps same as first maps:", bool((kb2 == kb).all()))
        else:
            print((H, W), Pout, "ok")

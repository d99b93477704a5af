"""Compare the two census sweeps (vector ALU vs matrix cores) key by key on a few geometries."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import libstevi_amd as sv  # noqa: E402
from helpers import parallax_pair  # noqa: E402

dev = torch.device("cuda:0")
for (H, W, D, hr, Pout, nd) in [(37, 300, 256, 4, 100.0, 8), (37, 300, 256, 3, 100.0, 4), (16, 700, 64, 4, 100.0, 8), (16, 640, 32, 4, 100.0, 8)]:
    src, tgt, _ = parallax_pair(H, W, 12, 8, 40, 3, 17, seed=D)
    l, r = torch.from_numpy(tgt).to(dev), torch.from_numpy(src).to(dev)
    out = {}
    for mode in (1, 3):
        sv.set_option(l, "census_sweep", mode)
        res = sv.stereoMatch(sv.matchingFunctions.CENSUS, l, r, hr, hr, D, sgmDirections=nd, Pout=Pout, shard=(0, D))
        keys = sv.censusShardKeys(l, r, hr, hr, D, (0, D), sgmDirections=nd, Pout=Pout)
        out[mode] = (res["disp"].cpu().numpy(), keys.cpu().numpy())
    sv.set_option(l, "census_sweep", 0)
    print((H, W, D, hr), "int8 vs valu: key diff", int((out[1][1] != out[2][1]).sum()))
    dk = out[1][1] != out[3][1]
    out[2] = out[3]
    print((H, W, D, hr), "fp4 vs valu: disp diff", int((out[1][0] != out[3][0]).sum()), "key diff", int(dk.sum()))
    if dk.any():
        ii, jj, rr = np.nonzero(dk)
        for t in range(min(8, len(ii))):
            a, b = int(out[1][1][ii[t], jj[t], rr[t]]), int(out[2][1][ii[t], jj[t], rr[t]])
            print("  px", ii[t], jj[t], "region", rr[t], "valu", (a >> 12, 4095 - (a & 4095)) if a != 0x7FFFFFFF else None,
                  "mfma", (b >> 12, 4095 - (b & 4095)) if b != 0x7FFFFFFF else None)
        print("  columns with diffs:", sorted(set(jj.tolist()))[:40])

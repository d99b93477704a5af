#!/usr/bin/env python3
"""A/B of C4 (4096 x 2160, NCC 11 x 11, D = 256, SGM-8 Score branch, parabola) with the Left2Right pass under the cost-volume kernel
("sgm_overlap" = number of row bands; arms from the command line, default 0 2 4 8) and after it (0), same process, arms alternated: ms per step, disparity checksum, refined-map checksum."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import libstevi_amd as sv  # noqa: E402
from helpers import parallax_pair  # noqa: E402

MF, IK = sv.matchingFunctions, sv.InterpolationKernel


def main():
    small = "1080p" in sys.argv[1:]
    arms = [int(a) for a in sys.argv[1:] if a.isdigit()] or [0, 2, 4, 8]
    W, H, D = (1920, 1080, 256) if small else (4096, 2160, 256)
    recipe = (320, 320, 380, 8, 64, 4) if small else (640, 640, 760, 16, 128, 4)
    src, tgt, _ = parallax_pair(H, W, *recipe)
    dev = torch.device("cuda:0")
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)

    def step():
        return sv.stereoMatch(MF.NCC, d_tgt, d_src, 5, 5, D, sgmDirections=8, refineKernel=IK.Parabola, refine_h_radius=5, refine_v_radius=5)

    for rnd in range(3):
        for arm in arms:
            sv.set_option(d_src, "sgm_overlap", arm)
            for _ in range(2):
                out = step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                out = step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            ref = out["refined"]
            print(json.dumps({"sgm_overlap": arm, "round": rnd, "ms": round(ms, 3), "disp_checksum": int(out["disp"].sum().item()),
                              "refined_bits_checksum": int(ref.view(torch.int32).to(torch.int64).sum().item())}), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A/B of the census + SGM step's kernels under context options, same box, same process (a development tool, not the bench):

    python tools/sweep_ab.py [--config C3|C5] [--reps 40] name=value[,name=value...] ...

Every positional argument is one arm: the options are set, the step (recurrences run, as bench.py's headline) is repeated, and the
arm prints its ms per step (wall clock around `reps` steps), the per-kernel event brackets (each reads ~5 us high: compare arms, do not
quote) and the disparity checksum.  The arms are run twice in alternation so that clock drift shows."""
import argparse
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
from bench import C3, C5  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--shortcut", type=int, default=0, help="census_winner_shortcut (0: recurrences run, the bench headline)")
    ap.add_argument("arms", nargs="*", default=["census_sweep_rl=0", "census_sweep_rl=1"])
    args = ap.parse_args()
    wl = dict(C3 if args.config == "C3" else C5)
    dev = torch.device("cuda", 0)
    src, tgt, _ = parallax_pair(wl["H"], wl["W"], wl["side"], wl["v"], wl["h"], wl["bg"], wl["sq"], wl["seed"])
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    sv.set_option(d_src, "census_winner_shortcut", args.shortcut)

    def step():
        return sv.stereoMatch(sv.matchingFunctions.CENSUS, d_tgt, d_src, wl["h_r"], wl["v_r"], wl["D"], sgmDirections=wl["sgm"],
                              P1=wl["P1"], P2=wl["P2"], Pout=wl["Pout"])

    touched = set()
    for rnd in range(args.rounds):
        for arm in args.arms:
            opts = [kv.split("=") for kv in arm.split(",") if kv]
            for k, v in opts:
                sv.set_test_option(d_src, k, int(v))
                touched.add(k)
            for _ in range(5):
                out = step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                out = step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.reps * 1e3
            sv.profile_reset(d_src)
            sv.profile_enable(d_src, True)
            for _ in range(args.reps):
                step()
            sv.profile_enable(d_src, False)
            prof = sv.profile_collect(d_src)
            print(json.dumps({"arm": arm, "round": rnd, "ms_per_step": round(ms, 4), "checksum": int(out["disp"].sum().item()),
                              "kernel_us_bracketed": {k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in prof.items()}}), flush=True)


if __name__ == "__main__":
    main()

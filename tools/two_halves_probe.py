#!/usr/bin/env python3
"""VERDICT r03 item 5, measured before anything is built: would running census_transform -> sweep on two ROW HALVES of the frame, on
two HIP streams, joined before the scans, shorten the C3 step?

The two halves are independent up to the sweep (cross_correlations.h:222: a cost depends on one row of each image), and
svh_census_band_match IS "census transform of a band + halo -> sweep writing that band's rows": two such calls on two streams, joined
on the third, are exactly the overlapped part of the proposed step (the scans and the per-pixel kernel, 24 us, come after the join
unchanged).  So this script times, same process, arms alternated:
    whole   the whole frame through transform -> sweep on one stream (stereoMatch, winner identity: no scans, no finalize)
    halves  rows [0, H/2) on stream A and rows [H/2, H) on stream B, the main stream waiting for both, every frame
If `halves` is not at least 5 % of the C3 step (4 us) faster than `whole`, the restructuring cannot reach the bar set for it."""
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

dev = torch.device("cuda:0")
W, H, D = 1920, 1080, 256
src, tgt, _ = parallax_pair(H, W, 320, 320, 380, 8, 64, 3)
d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
MF = sv.matchingFunctions
kw = dict(sgmDirections=8, P1=0.001, P2=0.01, Pout=100.0)
main = torch.cuda.current_stream()
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
out = torch.empty((H, W), dtype=torch.int32, device=dev)
splits = [int(a) for a in sys.argv[1:]] or [H // 2]


def whole():
    return sv.stereoMatch(MF.CENSUS, d_tgt, d_src, 4, 4, D, **kw)["disp"]


def halves(h2):
    sa.wait_stream(main)
    sb.wait_stream(main)
    with torch.cuda.stream(sa):
        sv.censusBandMatch(d_tgt, d_src, 4, 4, D, (0, h2), out=out[:h2], **kw)
    with torch.cuda.stream(sb):
        sv.censusBandMatch(d_tgt, d_src, 4, 4, D, (h2, H - h2), out=out[h2:], **kw)
    main.wait_stream(sa)
    main.wait_stream(sb)
    return out


def timed(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ref = whole().clone()
for h2 in splits:
    assert torch.equal(halves(h2), ref)
for rnd in range(3):
    line = {"round": rnd, "whole_ms": round(timed(whole), 4)}
    for h2 in splits:
        line[f"halves_{h2}_ms"] = round(timed(lambda: halves(h2)), 4)
    # the same two band calls one after the other on ONE stream: what the split itself costs (two half-size launches of each kernel, the halo rows twice)
    def serial(h2=splits[0]):
        sv.censusBandMatch(d_tgt, d_src, 4, 4, D, (0, h2), out=out[:h2], **kw)
        sv.censusBandMatch(d_tgt, d_src, 4, 4, D, (h2, H - h2), out=out[h2:], **kw)
    line["halves_one_stream_ms"] = round(timed(serial), 4)
    print(json.dumps(line), flush=True)

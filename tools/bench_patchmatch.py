#!/usr/bin/env python3
"""Timing of the examples/stereo-match compute chain on device arrays: cachelessPatchMatch (ZNCC, on-demand features) -> truncated
on-demand cost volume -> refineDisp2dCostInterpolation<Equiangular>, with the oracle on the box's host cores next to it."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import libstevi_amd as sv
import oracle as so
MF = sv.matchingFunctions
dev = torch.device("cuda:0")
for name, W, H, C, radius, search, n_iter in [("1920x1080 RGB, 7x7 windows, 21 x 41 offsets, 10 iterations", 1920, 1080, 3, 3, (-10, 10, -20, 20), 10),
                                               ("640x480 grey, 5x5 windows, 11 x 33 offsets, 10 iterations", 640, 480, 1, 2, (-5, 5, -16, 16), 10)]:
    rng = np.random.default_rng(5)
    base = rng.uniform(0, 1, (H + 64, W + 64, C)).astype(np.float32)
    base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) / np.float32(4)
    src = np.ascontiguousarray(base[32:32 + H, 32:32 + W])
    tgt = np.ascontiguousarray(base[32 - 3:32 - 3 + H, 32 + 7:32 + 7 + W])  # source(i, j) == target(i + 3, j - 7)
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    off = sv.searchOffset2(*search)
    def chain():
        disp = sv.cachelessPatchMatch(MF.ZNCC, d_src, d_tgt, radius, off, n_iter, 4, seed=11)
        tcv = sv.onDemandTruncatedCostVolume(MF.ZNCC, d_src, d_tgt, radius, off, disp, 1)
        return disp, sv.refineDisp2dCostInterpolation(sv.InterpolationKernel.Equiangular, tcv, disp)
    chain(); torch.cuda.synchronize()
    sv.profile_reset(d_src); sv.profile_enable(d_src, True)
    n = 3
    t0 = time.perf_counter()
    for _ in range(n): disp, refined = chain()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    sv.profile_enable(d_src, False)
    prof = sv.profile_collect(d_src)
    t0 = time.perf_counter()
    rows = 128  # a row band keeps the oracle within seconds; its PatchMatch parallelises over rows / columns
    exp, _ = so.cacheless_patch_match(so.ZNCC, 2, src[:rows], tgt[:rows], radius, radius, ((search[0], search[1]), (search[2], search[3])), n_iter, 4, 11)
    cpu = (time.perf_counter() - t0) * H / rows
    d = disp.cpu().numpy()
    correct = float(np.mean((d[16:-16, 16:-16, 0] == 3) & (d[16:-16, 16:-16, 1] == -7)))
    print(json.dumps({"chain": name, "ms": round(dt * 1e3, 2), "Mpixels_per_s": round(W * H / dt / 1e6, 1), "planted_flow_recovered": round(correct, 4),
                      "cpu_oracle_ms_extrapolated_from_128_rows": round(cpu * 1e3), "cpu_threads": so.num_threads(),
                      "kernel_ms": {k: round(v[0] / n, 3) for k, v in prof.items()}}), flush=True)

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
FORM = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # patchmatch_search_form: 1 default (chunked), 0 round 4's batched kernel, 2 / 3 lane per candidate (fetching / re-forming the target features)
sv.set_test_option(torch.zeros(1, device=dev), "patchmatch_search_form", FORM)
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
    # A bound for the random search: every iteration evaluates n_random candidates per pixel, each the ordered sum of nF
    # per-feature terms of two float vectors: cost evaluations x nF x 8 bytes of operands (the source vector is shared by a pixel's
    # candidates: + nF x 4 per pixel instead of per candidate in the `shared_source` figure).  Against HBM: the operands are gathered (the
    # targets lie within the search range of the pixel), so neighbouring pixels' windows overlap in L2 -- the achieved figure can exceed what
    # HBM alone would deliver.
    nF = (2 * radius + 1) ** 2 * C
    its = n_iter  # (the planted flow keeps every iteration busy: none ends early)
    evals = its * W * H * 4
    search_ms = prof.get("patchmatch_search", (0.0, 0))[0] / n
    bound = {"cost_evaluations": evals, "nF": nF, "operand_bytes": evals * nF * 8, "operand_bytes_shared_source": evals * nF * 4 + its * W * H * nF * 4,
             "search_ms": round(search_ms, 3), "achieved_GBps_on_operand_bytes": round(evals * nF * 8 / (search_ms * 1e-3) / 1e9, 1) if search_ms else None,
             "frac_of_8TBps": round(evals * nF * 8 / (search_ms * 1e-3) / 1e9 / 8000.0, 3) if search_ms else None,
             "unique_feature_bytes_per_iteration": 2 * W * H * nF * 4}
    print(json.dumps({"chain": name, "patchmatch_search_form": FORM, "ms": round(dt * 1e3, 2), "random_search_bound": bound, "Mpixels_per_s": round(W * H / dt / 1e6, 1), "planted_flow_recovered": round(correct, 4),
                      "cpu_oracle_ms_extrapolated_from_128_rows": round(cpu * 1e3), "cpu_threads": so.num_threads(),
                      "kernel_ms": {k: round(v[0] / n, 3) for k, v in prof.items()}}), flush=True)

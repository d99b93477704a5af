#!/usr/bin/env python3
"""The rows of the reference's own benchmark program (test/benchmarks/benchmarkCrossCorrelationAlgorithms.cpp) on the HIP path:
  benchmarkLocalAlgorithm        :433-459  unfoldBasedCostVolume<f> -> extractSelectedIndex -> selectedIndexToDisp          (:92-96)
  benchmarkSemiGlobalAlgorithm   :533-550  ... -> sgmCostVolume<nDir, strategy>(CV, 0.001, 0.01, Margins(), 100) -> ...     (:288-294)
at its three resolutions (48x64 D16 5x5, 480x640 D160 5x5, 1080x1920 D320 7x7; the high-definition SGM rows are the ones the file
comments out as "too slow"), synthetic pairs from the file's recipe (:342-358).  Timed twice: through the per-function entry points on
device arrays (the benchmark's own call sequence, volumes in HBM between the calls) and through the fused call; the oracle's time on the
host threads beside it where it finishes in seconds.  16 directions are refused (the reference's lines race there)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import libstevi_amd as sv
import oracle as so
from helpers import parallax_pair
MF = sv.matchingFunctions
dev = torch.device("cuda:0")
RES = {"VerySmall": (48, 64, 2, 16), "StandardDefinition": (480, 640, 2, 160), "HighDefinition": (1080, 1920, 3, 320)}
LOCAL = [MF.NCC, MF.ZNCC, MF.SSD, MF.ZSSD, MF.SAD, MF.ZSAD]
only = set(sys.argv[1:])


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    regions = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): out = fn()
        torch.cuda.synchronize(); regions.append((time.perf_counter() - t0) / n)
    return sorted(regions)[1], out


for res, (H, W, r, D) in RES.items():
    if only and res not in only: continue
    src, tgt, _ = parallax_pair(H, W, H // 3, H // 3, W // 3, max(D // 16, 1), D // 4, 11)
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    n = 200 if H < 100 else (20 if H < 1000 else 5)
    rows = [("local", f, 0) for f in LOCAL] + [("sgm", MF.ZNCC, 4), ("sgm", MF.ZNCC, 8)]
    for kind, func, ndir in rows:
        strat = sv.matchFuncStrategy(func)
        def chain():
            cv = sv.unfoldBasedCostVolume(func, d_tgt, d_src, r, r, D)
            if ndir: cv = sv.sgmCostVolume(ndir, strat, cv, 0.001, 0.01, None, 100.0)
            return sv.selectedIndexToDisp(sv.extractSelectedIndex(strat, cv), 0)
        def fused():
            return sv.stereoMatch(func, d_tgt, d_src, r, r, D, sgmDirections=ndir, P1=0.001, P2=0.01, Pout=100.0)["disp"]
        t_chain, disp_c = timed(chain, n)
        t_fused, disp_f = timed(fused, n)
        t_stmt = None
        if not ndir:  # the volume carries its winner to extractSelectedIndex (what the C++ drop-in headers do for Score-strategy functions on DeviceArray; opt-in here)
            def chain_stmt():
                cv = sv.unfoldBasedCostVolume(func, d_tgt, d_src, r, r, D, keep_winner=True)
                return sv.selectedIndexToDisp(sv.extractSelectedIndex(strat, cv), 0)
            t_stmt, disp_s = timed(chain_stmt, n)
            assert torch.equal(disp_s, disp_c)
        line = {"row": f"{res} {H}x{W} - {2 * r + 1}x{2 * r + 1} windows - disp {D}" + (f" - {ndir} directions" if ndir else "") + f" - {func.name.lower()}",
                "benchmark": "benchmarkSemiGlobalAlgorithm" if ndir else "benchmarkLocalAlgorithm",
                "ms_per_function_calls": round(t_chain * 1e3, 4), "ms_per_function_calls_winner_carried": None if t_stmt is None else round(t_stmt * 1e3, 4),
                "ms_fused_call": round(t_fused * 1e3, 4),
                "Mdisparities_per_s_fused": round(W * H * D / t_fused / 1e6, 1), "maps_equal": bool(torch.equal(disp_c, disp_f))}
        if H < 1000:  # the oracle on the host threads (the whole frame)
            t0 = time.perf_counter()
            cv = so.unfold_cost_volume(int(func), tgt, src, r, r, D, so.RIGHT_TO_LEFT)
            if ndir: cv = so.sgm(cv, ndir, int(strat), 0.001, 0.01, (0, 0, 0, 0), 100.0)
            want = so.index_to_disp(so.extract_index(cv, int(strat)), so.RIGHT_TO_LEFT)
            line["cpu_oracle_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
            line["cpu_threads"] = so.num_threads()
            line["pixels_equal_to_oracle"] = round(float((disp_f.cpu().numpy() == want).mean()), 6)
        print(json.dumps(line), flush=True)
    # benchmarkLocalAlgorithmWithCompressor (:461-531): ZNCC on the 17 superpixel features of the GrPix17R3 / GrPix17R4 masks
    for gen in ("GrPix17R3Filter", "GrPix17R4Filter"):
        comp = sv.UnFoldCompressor(getattr(sv.CompressorGenerators, gen)())
        def chain():
            cv = sv.unfoldBasedCostVolumeCompressed(MF.ZNCC, d_tgt, d_src, comp, D)
            return sv.selectedIndexToDisp(sv.extractSelectedIndex(sv.matchFuncStrategy(MF.ZNCC), cv), 0)
        t_chain, disp_c = timed(chain, n)
        line = {"row": f"{res} {H}x{W} - {gen[:-6]} windows - disp {D} - zncc", "benchmark": "benchmarkLocalAlgorithmWithCompressor",
                "ms_per_function_calls": round(t_chain * 1e3, 4), "Mdisparities_per_s": round(W * H * D / t_chain / 1e6, 1)}
        print(json.dumps(line), flush=True)
    # benchmarkHierarchicalAlgorithm (:362-402): ZNCC, depth 1 - 3, truncation radius 2 (tools/bench_hierarchical.py has the kernel breakdown)
    if H >= 100:
        for depth in (1, 2, 3):
            t_h, _ = timed(lambda: sv.hiearchicalTruncatedCostVolume(MF.ZNCC, depth, d_tgt, d_src, r, r, D, 2), n)
            print(json.dumps({"row": f"{res} {H}x{W} - {2 * r + 1}x{2 * r + 1} windows - disp {D} - depth {depth} - zncc", "benchmark": "benchmarkHierarchicalAlgorithm",
                              "ms": round(t_h * 1e3, 4), "Mdisparities_per_s": round(W * H * D / t_h / 1e6, 1)}), flush=True)

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json): census 9x9 + 8-path SGM.

    N = 1   BASELINE.json configs[2]: 1920x1080, D = 256 (the configuration the metric is quoted on).
            One "step" = one pass of the fused device pipeline (svh_stereo_match) over one synthetic stereo pair that is already
            resident in HBM: census transform of both images -> Hamming cost (evaluated on the fly) -> the reference's five
            effective SGM passes -> winner -> disparity map (int32).
    N > 1   BASELINE.json configs[4]: 8192x4320, D = 512, the disparity range split N ways (strong scaling: the problem is
            fixed, every rank sweeps D / N disparities), one RCCL int32 MIN all-reduce of the regional winner keys per frame,
            scans + finalize replicated.  One step = one frame through libstevi_amd.sharded.ShardedStereoPipeline.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Metric: Mdisparities/s = W*H*D*steps / seconds / 1e6 (the whole job).
The headline is measured with the SGM line recurrences run; `winner_identity` in the same line is the library's default
path for disparity-only calls, which needs none of them.
The second half of BASELINE.json's metric ("end-px-err vs ref") is `end_px_err`: the disparity map of the timed pipeline
compared, outside the timed region, with the CPU oracle's map of the same pair (benchmarkCrossCorrelationAlgorithms.cpp:288-294
restated: unfoldBasedCostVolume -> sgmCostVolume<8> -> extractSelectedIndex -> selectedIndexToDisp); any differing pixel makes
the run fail (exit code 3).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

torch = None  # imported by main(): the self-launching parent of a bare `--gpus N` run (launch_ranks) must never load it, let alone touch HIP

# C3 = 1920x1080, D=256, census 9x9, SGM-8 (P1=0.001, P2=0.01, Pout=100), seed 3; C5 = 8192x4320, D=512, seed 5
C3 = dict(name="C3", W=1920, H=1080, D=256, h_r=4, v_r=4, sgm=8, P1=0.001, P2=0.01, Pout=100.0, seed=3, bg=8, sq=64, side=320, v=320, h=380)
C5 = dict(name="C5", W=8192, H=4320, D=512, h_r=4, v_r=4, sgm=8, P1=0.001, P2=0.01, Pout=100.0, seed=5, bg=32, sq=256, side=1280, v=1280, h=1520)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# MI355X_MICROARCH.md, matrix cores (dense): BF16 ~2.5 PFLOP/s; int8 runs at twice the BF16 rate, block-scaled FP4 at four times
FP4_PEAK_TOPS = 10000.0
MFMA_PEAK_TOPS = {"census_sweep_rl": FP4_PEAK_TOPS, "census_sweep_pm": FP4_PEAK_TOPS}  # the two FP4 sweep kernels (library profile labels)
# library profile label -> the kernel symbol rocprofv3 prints for it (profiles/*_kernel_stats.csv, profiles/traffic.json)
KERNEL_SYMBOL = {"census_sweep_rl": "census_sweep_rl_kernel", "census_sweep_pm": "census_sweep_pm_kernel", "census_sweep": "census_sweep_kernel",
                 "census_transform": "census_grey_kernel", "sgm_line_scans": "scan_cols_kernel", "census_finalize": "tile_finalize_kernel"}


def algorithmic_bytes(kernel, wl):
    """Algorithmic bytes per launch of `kernel` under the model of SURVEY.md section 8(d), row C3 (DESIGN.md "Roofline"):
    C as u8, S as i16, each of the five effective SGM passes = one read of C + one read-modify-write of S:
        28 B/voxel = 1 (C write) + 5 x (1 + 4) + 2 (final S read),   60 B/pixel = 8 (images) + 48 (census words) + 4 (disparity).
    The fused pipeline has ONE voxel-sized kernel (the sweep); all 28 B/voxel of the model are work it replaces.  The
    per-pixel kernels carry the per-pixel bytes."""
    H, W, D = wl["H"], wl["W"], wl["D"]
    px, vox = H * W, H * W * D
    table = {
        "census_sweep": 28.0 * vox,
        "census_sweep_pm": 28.0 * vox,
        "census_sweep_rl": 28.0 * vox,
        "census_transform": (4.0 + 24.0) * px,      # one image: 4 B read + nW = 3 words written and read back
        "sgm_line_scans": (4.0 + 1.6) * px,          # g read + the min_p values on tile edges written (with six maps instead: 4 + 24)
        "sgm_line_carries": 4.0 * px,                # g read (the carries are per line and band: negligible)
        "census_finalize": (8.0 + 4.0 + 1.6 + 4.0) * px,  # keys + g + tile-edge values + disparity (with six maps instead: 8 + 24 + 4)
        "census_scan_finalize": (4.0 + 8.0 + 4.0) * px,  # g + keys + disparity
    }
    return table.get(kernel)


def load_measured_traffic(kernel):
    """HBM bytes per launch of `kernel` (library profile label) measured with rocprofv3 --pmc (FETCH_SIZE / WRITE_SIZE passes, gfx950
    corrections of MI355X_MICROARCH.md) on this workload; the summary is committed under profiles/ by the round that measured it, keyed by
    the kernel symbol rocprofv3 prints."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(KERNEL_SYMBOL.get(kernel, kernel), {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def oracle_disparity(wl, src, tgt, rows=None, variant=1):
    """The reference chain on the CPU oracle (test infrastructure; only ever the checker / the baseline)."""
    import oracle as so
    s, t = (src, tgt) if rows is None else (np.ascontiguousarray(src[:rows]), np.ascontiguousarray(tgt[:rows]))
    t0 = time.perf_counter()
    cv = so.unfold_cost_volume(so.CENSUS, t, s, wl["h_r"], wl["v_r"], wl["D"])
    vol = so.sgm(cv, wl["sgm"], so.COST, wl["P1"], wl["P2"], (0, 0, 0, 0), wl["Pout"], variant=variant)
    disp = so.index_to_disp(so.extract_index(vol, so.COST))
    return disp, time.perf_counter() - t0


def cpu_baseline(wl, src, tgt, budget_s=10.0):
    """Oracle (CPU restatement of the reference) timed on a bounded sample of the workload: the literal O(D^2) loops on a row
    band, and the O(D) restatement (bitwise identical) on the whole frame -- whose disparity map is the `end_px_err` check."""
    import oracle as so
    D = wl["D"]
    rows = 4
    _, dt = oracle_disparity(wl, src, tgt, rows, variant=0)
    rows2 = int(min(wl["H"], max(rows, rows * (budget_s * 0.75) / max(dt, 1e-6))))
    if rows2 > rows:
        _, dt = oracle_disparity(wl, src, tgt, rows2, variant=0)
        rows = rows2
    full_disp, dt_lin = oracle_disparity(wl, src, tgt, None, variant=1)
    cores = so.num_threads()
    base = {
        "value": round(rows * wl["W"] * D / dt / 1e6, 3), "unit": "Mdisparities/s", "cores": cores, "kind": "port",
        "sample": f"top {rows} rows of the same {wl['W']}x{wl['H']} pair, D={D}: census 9x9 volume + SGM-8 (literal O(D^2) loops of "
                  f"sgm.h:269-295) + argmin, {dt:.1f} s, OpenMP over rows/lines",
        "linear_sgm_value": round(wl["H"] * wl["W"] * D / dt_lin / 1e6, 3),
        "linear_sgm_sample": f"all {wl['H']} rows, O(D) SGM restatement (bitwise identical), {dt_lin:.1f} s",
    }
    return base, full_disp


def cpu_band_baseline(func, strategy, src, tgt, r, D, sgm_dirs, rows=64, cols=None, refine=False):
    """BASELINE.md section 2 for the configurations whose literal port cannot run whole: the oracle (CPU restatement of the reference) on a
    band of `rows` image rows (plus the window's halo) [and the first `cols` columns] of the same pair: cost volume, then -- when the
    configuration aggregates -- SGM with the literal O(D^2) loops of sgm.h:269-295 and with the O(D) restatement (bitwise identical), winner
    [, truncated volume + parabola].  Measured band throughput, and the linear-in-pixels extrapolation to the whole frame, labelled as such."""
    import oracle as so
    H, W = src.shape
    s, t = src[:rows + 2 * r], tgt[:rows + 2 * r]
    if cols is not None and cols < W:
        s, t = s[:, :cols], t[:, :cols]
    s, t = np.ascontiguousarray(s), np.ascontiguousarray(t)
    bh, bw = s.shape
    vox = bh * bw * D
    t0 = time.perf_counter()
    cv = so.unfold_cost_volume(func, t, s, r, r, D)
    t_cv = time.perf_counter() - t0
    out = {"unit": "Mdisparities/s", "cores": so.num_threads(), "kind": "port",
           "cost_volume_Mdisparities_per_s": round(vox / t_cv / 1e6, 2), "cost_volume_s": round(t_cv, 2)}
    frame_scale = (H * W) / float(bh * bw)
    what = f"rows 0..{bh - 1}" + (f", columns 0..{bw - 1}" if bw < W else "") + f" of the same {W}x{H} pair, D={D}"
    if not sgm_dirs:
        out.update({"value": out["cost_volume_Mdisparities_per_s"], "sample": f"{what}: the cost volume (the configuration's whole output), {t_cv:.2f} s",
                    "extrapolated_full_frame_s": round(t_cv * frame_scale, 2), "extrapolation": "linear in pixels (the volume's voxels are independent)"})
        return out
    legs = {}
    for variant, name in ((0, "literal"), (1, "linear")):
        t0 = time.perf_counter()
        vol = so.sgm(cv, sgm_dirs, strategy, 0.001, 0.01, (0, 0, 0, 0), 100.0, variant=variant)
        idx = so.extract_index(vol, strategy)
        if refine:
            so.refine_disp(so.truncated_cost_volume(vol, idx, r, r, 1), idx, so.PARABOLA)
        else:
            so.index_to_disp(idx)
        legs[name] = time.perf_counter() - t0
    lit, lin = t_cv + legs["literal"], t_cv + legs["linear"]
    out.update({"value": round(vox / lit / 1e6, 3),
                "sample": f"{what}: cost volume {t_cv:.2f} s + SGM-{sgm_dirs} with the literal O(D^2) loops of sgm.h:269-295 + winner" + (" + truncated volume + parabola" if refine else "") +
                          f" {legs['literal']:.2f} s, OpenMP over rows / lines",
                "extrapolated_full_frame_s": round(lit * frame_scale, 1), "extrapolation": "band time x (frame pixels / band pixels): LINEAR-IN-H EXTRAPOLATION, not measured",
                "linear_sgm_value": round(vox / lin / 1e6, 2), "linear_sgm_sample": f"the same band, O(D) SGM restatement (bitwise identical), {legs['linear']:.2f} s after the cost volume",
                "linear_sgm_extrapolated_full_frame_s": round(lin * frame_scale, 1)})
    return out


def band_check_mask(H, W, band):
    """Pixels of the top `band` rows whose SGM result is the same whether the oracle sees the whole frame or only the top
    band + v_r rows: every pass but DownLeft2UpRight reaches a pixel from above or from the left; that pass runs up the
    anti-diagonals from the left column, so a pixel keeps its value when its anti-diagonal starts inside the band
    (i + j < band) or is not traversed at all (i + j >= H)."""
    ii, jj = np.meshgrid(np.arange(band), np.arange(W), indexing="ij")
    return (ii + jj < band) | (ii + jj >= H)


def end_px_err(got, want, against, mask=None):
    got = np.asarray(got)
    if mask is not None:
        got, want = got[:want.shape[0]][mask], want[mask]
    diff = int(np.count_nonzero(got != want))
    return {"pixels_differing": diff, "of": int(want.size), "checked_against": against}


def api_chain(sv, wl, d_tgt, d_src, reps=5):
    """The reference benchmark's chain through the PER-FUNCTION entry points on device arrays, volumes materialised in HBM between
    the calls (what the drop-in C++ headers run, minus their PCIe copies): the form SURVEY.md 8(d)'s byte model describes.
    Compulsory HBM bytes: C written (4 B/voxel, API type float) and read once, S written = 12 B/voxel (until round 4 S was also read back
    by extractSelectedIndex: 16; since then the winner the aggregation's last kernel picks travels with the array it wrote, as the
    volume's regional minima do -- statements that die at the first write to the array)."""
    MF = sv.matchingFunctions
    strat = sv.matchFuncStrategy(MF.CENSUS)

    def chain():
        cv = sv.unfoldBasedCostVolume(MF.CENSUS, d_tgt, d_src, wl["h_r"], wl["v_r"], wl["D"], keep_minima=True)  # as the C++ shims do
        sgm = sv.sgmCostVolume(wl["sgm"], strat, cv, wl["P1"], wl["P2"], None, wl["Pout"], keep_winner=True)  # as the C++ shims do
        return sv.selectedIndexToDisp(sv.extractSelectedIndex(strat, sgm), 0)

    disp = chain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        disp = chain()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    sv.profile_reset(d_src)
    sv.profile_enable(d_src, True)
    for _ in range(reps):
        chain()
    sv.profile_enable(d_src, False)
    prof = sv.profile_collect(d_src)
    vox = wl["W"] * wl["H"] * wl["D"]
    comp = 12.0 * vox + 12.0 * wl["W"] * wl["H"]
    ach = comp / (ms * 1e-3) / 1e9
    return {"ms": round(ms, 4), "Mdisparities_per_s": round(vox / ms / 1e3, 1),
            "calls": "unfoldBasedCostVolume -> sgmCostVolume<8,Cost> -> extractSelectedIndex -> selectedIndexToDisp, device arrays, float32 volumes in HBM",
            "roofline": {"bound": "hbm", "compulsory_bytes": int(comp), "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBPS, 4), "model": "12 B/voxel (C written and read once, S written once, float32) + 12 B/pixel",
                         "frac_on_the_16_B_per_voxel_model_of_rounds_2_and_3": round((16.0 * vox + 12.0 * wl["W"] * wl["H"]) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
            "kernel_ms": {k: round(v[0] / reps, 4) for k, v in prof.items()}}, disp


def host_chain(sv, wl, src, tgt, reps=3):
    """The reference benchmark's chain exactly as its callers write it (benchmarkCrossCorrelationAlgorithms.cpp:92-96, :288-294): HOST arrays in,
    host arrays out of every function, so each float32 volume crosses PCIe down and up again -- SURVEY.md 8(d)'s "host API end-to-end
    (H2D + D2H) reported separately".  numpy arrays here; the C++ drop-in headers on Multidim::Array run the same C entry points
    (tools/bench_cpp_chain.cpp).  The results are allocated by the library's mirror in page-locked memory (svh_host_alloc), the images are
    the caller's pageable arrays."""
    MF = sv.matchingFunctions
    strat = sv.matchFuncStrategy(MF.CENSUS)

    def chain():
        cv = sv.unfoldBasedCostVolume(MF.CENSUS, tgt, src, wl["h_r"], wl["v_r"], wl["D"])
        sgm = sv.sgmCostVolume(wl["sgm"], strat, cv, wl["P1"], wl["P2"], None, wl["Pout"])
        return sv.selectedIndexToDisp(sv.extractSelectedIndex(strat, sgm), 0)

    t0 = time.perf_counter()
    disp = chain()  # first call: page-locks the result blocks (kept in the library's cache afterwards)
    first_ms = (time.perf_counter() - t0) * 1e3
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        disp = chain()
        times.append((time.perf_counter() - t0) * 1e3)
    ms = float(np.median(times))
    px, vox = wl["W"] * wl["H"], wl["W"] * wl["H"] * wl["D"]
    pcie = 4.0 * vox * 4 + 8.0 * px + 3 * 4.0 * px  # C down, C up, S down, S up; two images up; index down, index up, disparity down
    assert isinstance(disp, np.ndarray)
    # ... and the fused entry point on the same host images (svh_stereo_match with host arrays: two images up, the disparity map down -- the
    # PCIe-inclusive figure of the headline's own call; `value` is measured with the images resident in HBM)
    def fused():
        return sv.stereoMatch(MF.CENSUS, tgt, src, wl["h_r"], wl["v_r"], wl["D"], sgmDirections=wl["sgm"], P1=wl["P1"], P2=wl["P2"], Pout=wl["Pout"])["disp"]
    fd = fused()
    tf = []
    for _ in range(5):
        t0 = time.perf_counter()
        fd = fused()
        tf.append((time.perf_counter() - t0) * 1e3)
    fused_ms = float(np.median(tf))
    return {"fused_call_on_host_images": {"ms": round(fused_ms, 3), "Mdisparities_per_s": round(vox / fused_ms / 1e3, 1), "pcie_bytes": int(12.0 * px),
                                          "same_disparity_map": bool(np.array_equal(fd, disp)),
                                          "note": "svh_stereo_match, numpy images in (pageable: staged through the pinned ring), numpy disparity map out; synchronous"},
            "ms": round(ms, 2), "ms_all": [round(t, 2) for t in times], "first_call_ms": round(first_ms, 1), "Mdisparities_per_s": round(vox / ms / 1e3, 1),
            "pcie_bytes": int(pcie), "pcie_GBps": round(pcie / (ms * 1e-3) / 1e9, 1),
            "calls": "unfoldBasedCostVolume -> sgmCostVolume<8,Cost> -> extractSelectedIndex -> selectedIndexToDisp on numpy arrays (host memory in, host memory out of every call)",
            "note": "four 2.1 GB crossings of the link per chain are what the reference's own call sequence asks for with host arrays; the kernels take 1.4 ms of it "
                    "(api_chain).  Results live in page-locked blocks (svh_host_alloc) that the DMA engines address directly; first_call_ms includes page-locking them"}, disp


def _time_steps(fn, steps, repeats=3):
    """median over `repeats` regions of `steps` calls, synchronised on both sides; ms per call"""
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    return float(np.median(out)), r


def _kernel_ms(sv, probe, fn, steps):
    sv.profile_reset(probe)
    sv.profile_enable(probe, True)
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    sv.profile_enable(probe, False)
    return {k: round(v[0] / steps, 4) for k, v in sv.profile_collect(probe).items()}


def other_configs(sv, dev, cpu_legs=True):
    """BASELINE.json's other configurations on ONE GPU, after the timed headline (they are parity-test cases, not the metric): a few steps
    each, per-kernel times, a roofline on compulsory bytes, and the result checked against the oracle (whole frame where the oracle
    finishes in seconds, a band of rows otherwise -- the pixels of the band whose SGM lines lie inside it).  Returns (dict, ok)."""
    import oracle as so
    from helpers import parallax_pair
    MF, IK = sv.matchingFunctions, sv.InterpolationKernel
    out, ok = {}, True

    def hbm(bytes_, ms, model):
        ach = bytes_ / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "compulsory_bytes": int(bytes_), "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "model": model}

    # ---- C1: 480x360, SAD 5x5, D = 64, argmin: the reference's own CPU-runnable case; the oracle runs it in full
    t0 = time.perf_counter()
    W, H, D, r = 480, 360, 64, 2
    src, tgt, _ = parallax_pair(H, W, 120, 120, 120, 4, 24, 1)
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    fn = lambda: sv.stereoMatch(MF.SAD, d_tgt, d_src, r, r, D)  # noqa: E731
    ms, res = _time_steps(fn, 50)
    c = {"workload": "480x360 synthetic pair, SAD 5x5 cost volume, D=64, argmin (BASELINE.json configs[0]; stands in for SimStereo)", "ms": round(ms, 4),
         "Mdisparities_per_s": round(W * H * D / ms / 1e3, 1), "kernel_ms": _kernel_ms(sv, d_src, fn, 20),
         "roofline": hbm(12.0 * W * H, ms, "12 B/pixel (two images in, the disparity map out): the winner comes out of the cost kernel, the volume is never written "
                                           "(SURVEY.md 8(d) has no byte model for C1; a design that materialises C moves 8 B/voxel more)")}
    if cpu_legs:
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            cv = so.unfold_cost_volume(so.SAD, tgt, src, r, r, D)
            want = so.index_to_disp(so.extract_index(cv, so.COST))
        dt = (time.perf_counter() - t1) / reps
        full = sv.stereoMatch(MF.SAD, d_tgt, d_src, r, r, D, want_cv=True)
        gcv, gdisp = full["cv"].cpu().numpy(), full["disp"].cpu().numpy()
        own = so.index_to_disp(so.extract_index(gcv, so.COST))
        c["cpu_baseline"] = {"value": round(W * H * D / dt / 1e6, 1), "unit": "Mdisparities/s", "cores": so.num_threads(), "kind": "port",
                             "sample": f"the whole frame, {reps} runs: unfoldBasedCostVolume<SAD> -> extractSelectedIndex -> selectedIndexToDisp, {dt * 1e3:.1f} ms each"}
        c["end_err"] = {"cost_max_rel_err": float(np.max(np.abs(gcv - cv) / np.maximum(1, np.abs(cv)))), "tolerance": 1e-4,
                        "pixels_differing_from_oracle_map": int(np.count_nonzero(gdisp != want)), "of": int(want.size),
                        "pixels_differing_on_the_library_s_own_volume": int(np.count_nonzero(gdisp != own)),
                        "checked_against": "oracle (port), whole frame; float costs agree to 1e-4, so the map is held to the oracle's argmin of the library's own volume (0) "
                                           "and compared with the oracle's own map for information (near-ties may flip)"}
        ok &= c["end_err"]["cost_max_rel_err"] <= 1e-4 and c["end_err"]["pixels_differing_on_the_library_s_own_volume"] == 0
    c["wall_s"] = round(time.perf_counter() - t0, 2)
    out["C1"] = c

    # ---- C2: 1920x1080, census 9x9 + Hamming cost volume, D = 128 (the volume is the output)
    t0 = time.perf_counter()
    W, H, D, r = 1920, 1080, 128, 4
    src, tgt, _ = parallax_pair(H, W, 320, 320, 380, 8, 64, 2)
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    fn = lambda: sv.unfoldBasedCostVolume(MF.CENSUS, d_tgt, d_src, r, r, D)  # noqa: E731
    ms, vol = _time_steps(fn, 20)
    c = {"workload": "1920x1080 synthetic pair, census 9x9 + Hamming cost volume (float32, the API type), D=128 (BASELINE.json configs[1])", "ms": round(ms, 4),
         "Mdisparities_per_s": round(W * H * D / ms / 1e3, 1), "kernel_ms": _kernel_ms(sv, d_src, fn, 10),
         "roofline": hbm(4.0 * W * H * D + 56.0 * W * H, ms, "SURVEY.md 8(d) C2: the volume written once as float32 (4 B/voxel) + 56 B/pixel (images, census words)")}
    if cpu_legs:
        band = 40
        cvb = so.unfold_cost_volume(so.CENSUS, tgt[:band], src[:band], r, r, D)
        c["end_err"] = {"voxels_differing": int(np.count_nonzero(vol[:band - r].cpu().numpy() != cvb[:band - r])), "of": int((band - r) * W * D),
                        "checked_against": f"oracle (port), top {band - r} rows of the volume, bit for bit"}
        ok &= c["end_err"]["voxels_differing"] == 0
        t1 = time.perf_counter()
        so.unfold_cost_volume(so.CENSUS, tgt, src, r, r, D)
        dt = time.perf_counter() - t1
        c["cpu_baseline"] = {"value": round(W * H * D / dt / 1e6, 1), "unit": "Mdisparities/s", "cores": so.num_threads(), "kind": "port",
                             "sample": f"the whole frame, one run: unfoldBasedCostVolume<CENSUS> 9x9, D={D} (census words of both images + the per-voxel Hamming loop), {dt:.2f} s"}
    del vol
    c["wall_s"] = round(time.perf_counter() - t0, 2)
    out["C2"] = c

    # ---- C4: 4096x2160, NCC 11x11, D = 256, SGM-8 (Score branch), argmax, truncatedCostVolume(S, idx, 5, 5, 1) + parabola (rule E6)
    t0 = time.perf_counter()
    W, H, D, r = 4096, 2160, 256, 5
    src, tgt, _ = parallax_pair(H, W, 640, 640, 760, 16, 128, 4)
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    kw = dict(sgmDirections=8, P1=0.001, P2=0.01, Pout=100.0, refineKernel=IK.Parabola, refine_h_radius=r, refine_v_radius=r)
    fn = lambda: sv.stereoMatch(MF.NCC, d_tgt, d_src, r, r, D, **kw)  # noqa: E731
    ms, res = _time_steps(fn, 5)
    vox = W * H * D
    c = {"workload": "4096x2160 synthetic pair, NCC 11x11 cost volume + 8-path SGM (Score branch) + argmax + parabola sub-pixel refinement, D=256 (BASELINE.json configs[3])",
         "ms": round(ms, 3), "Mdisparities_per_s": round(vox / ms / 1e3, 1), "kernel_ms": _kernel_ms(sv, d_src, fn, 3),
         "roofline": hbm((4.0 + 23.2) * vox + 36.0 * W * H, ms,
                         "C written once (4 B/voxel) + Score-branch SGM in its fused form 23.2 B/voxel (Left2Right: read C, write its contribution = 8; downward sweep: "
                         "read C and that contribution, write S = 12; DownLeft2UpRight on the 26 % of the pixels it visits: 12 x 0.264) + 36 B/pixel; SURVEY.md 8(d) "
                         "prices a pass-per-launch design at 68 B/voxel"),
         "sgm_store": "S is only stored where a later pass reads it (DownLeft2UpRight's 26 %): the winner records ride on the last writer of each pixel"}
    if cpu_legs:
        band = 28
        full = sv.stereoMatch(MF.NCC, d_tgt, d_src, r, r, D, want_cv=True, want_sgm_cv=True, **kw)
        torch.cuda.synchronize()
        same_maps = bool(torch.equal(full["disp"], res["disp"])) and bool(torch.equal(torch.nan_to_num(full["refined"], nan=-7.0), torch.nan_to_num(res["refined"], nan=-7.0)))
        gcv = full["cv"][:band].cpu().numpy()
        cvb = so.unfold_cost_volume(so.NCC, tgt[:band + r], src[:band + r], r, r, D)[:band]
        okc = ~np.isnan(cvb)
        m = band_check_mask(H, W, band)
        sb = so.sgm(gcv, 8, so.SCORE, 0.001, 0.01, (0, 0, 0, 0), 100.0)
        gs = full["sgm_cv"][:band].cpu().numpy()
        a, b = gs[m], sb[m]
        okv = ~np.isnan(b)
        idx = so.extract_index(gs, so.SCORE)
        gdisp = res["disp"][:band].cpu().numpy()
        ref = so.refine_disp(so.truncated_cost_volume(gs, idx, r, r, 1), idx, so.PARABOLA)
        gref = res["refined"][:band].cpu().numpy()
        rows = band - r
        mm = m[:rows]
        ra, rb = gref[:rows][mm], ref[:rows][mm]
        okr = ~np.isnan(rb)
        c["end_err"] = {"band_rows": band, "cost_max_abs_err": float(np.max(np.abs(gcv[okc] - cvb[okc]))), "cost_nan_mask_equal": bool(np.array_equal(np.isnan(gcv), np.isnan(cvb))),
                        "sgm_voxels_differing": int(np.count_nonzero(a[okv].view(np.uint32) != b[okv].view(np.uint32))) + int(np.count_nonzero(np.isnan(a) != np.isnan(b))),
                        "disp_pixels_differing": int(np.count_nonzero(gdisp[m] != so.index_to_disp(idx)[m])), "of_pixels": int(m.sum()),
                        "refined_max_abs_err": float(np.max(np.abs(ra[okr] - rb[okr]))), "refined_nan_mask_equal": bool(np.array_equal(np.isnan(ra), np.isnan(rb))),
                        "timed_call_equals_the_call_that_also_returned_the_volumes": same_maps, "tolerance": 1e-4,
                        "checked_against": f"oracle (port), top {band} rows: NCC costs <= 1e-4; Score-branch SGM of the library's own cost band bit for bit on the pixels whose "
                                           "lines lie inside the band; arg-max map exact; parabola-refined map <= 1e-4 with the same NaN mask"}
        e = c["end_err"]
        ok &= (e["cost_max_abs_err"] <= 1e-4 and e["cost_nan_mask_equal"] and e["sgm_voxels_differing"] == 0 and e["disp_pixels_differing"] == 0 and
               e["refined_max_abs_err"] <= 1e-4 and e["refined_nan_mask_equal"] and same_maps)
        del full
        # the oracle band covers the first rows only: the WHOLE maps against the pass-per-launch form of the Score branch (sgm_score_fused = 0)
        sv.set_option(d_src, "sgm_score_fused", 0)
        try:
            alt = fn()
        finally:
            sv.set_option(d_src, "sgm_score_fused", 1)
        c["whole_map_vs_pass_per_launch_form"] = {
            "disp_pixels_differing": int((alt["disp"] != res["disp"]).sum().item()), "of": int(alt["disp"].numel()),
            "refined_bits_differing": int((torch.nan_to_num(alt["refined"], nan=-7.0) != torch.nan_to_num(res["refined"], nan=-7.0)).sum().item()),
            "checked_against": "the same call with sgm_score_fused = 0 (one read-modify-write sweep per pass), whole frame"}
        ok &= c["whole_map_vs_pass_per_launch_form"]["disp_pixels_differing"] == 0 and c["whole_map_vs_pass_per_launch_form"]["refined_bits_differing"] == 0
        del alt
        c["cpu_baseline"] = cpu_band_baseline(so.NCC, so.SCORE, src, tgt, r, D, 8, rows=64, cols=2048, refine=True)
    del res
    torch.cuda.empty_cache()
    c["wall_s"] = round(time.perf_counter() - t0, 2)
    out["C4"] = c

    # ---- C5 on ONE GPU: 8192x4320, census 9x9 + SGM-8, D = 512, the whole range (the N > 1 bench shards it)
    t0 = time.perf_counter()
    W, H, D, r = 8192, 4320, 512, 4
    src, tgt, _ = parallax_pair(H, W, 1280, 1280, 1520, 32, 256, 5)
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    fn = lambda: sv.stereoMatch(MF.CENSUS, d_tgt, d_src, r, r, D, sgmDirections=8, P1=0.001, P2=0.01, Pout=100.0)  # noqa: E731
    ms, res = _time_steps(fn, 10)
    vox = W * H * D
    ops = 128.0 * vox
    c = {"workload": "8192x4320 synthetic pair, census 9x9 + 8-path SGM, D=512, argmin, the whole disparity range on one GPU (BASELINE.json configs[4] without the sharding)",
         "ms": round(ms, 4), "Mdisparities_per_s": round(vox / ms / 1e3, 1), "kernel_ms": _kernel_ms(sv, d_src, fn, 5),
         "sgm_recurrences_timed": True,
         "roofline": {"bound": "mfma", "algorithmic_ops": int(ops), "achieved": round(ops / (ms * 1e-3) / 1e12, 1), "peak": FP4_PEAK_TOPS, "unit": "TFLOP/s",
                      "frac": round(ops / (ms * 1e-3) / 1e12 / FP4_PEAK_TOPS, 4),
                      "model": "128 operations per voxel (64 census bits as a +-1 dot product) on the FP4 matrix cores, whole step; compulsory HBM traffic is 12 B/pixel: "
                               f"{12.0 * W * H / (ms * 1e-3) / 1e9:.0f} GB/s"}}
    if cpu_legs:
        band = 48
        want, _ = oracle_disparity(dict(h_r=r, v_r=r, D=D, sgm=8, P1=0.001, P2=0.01, Pout=100.0), src, tgt, band + r, variant=1)
        c["end_px_err"] = end_px_err(res["disp"].cpu().numpy(), want[:band], f"oracle (port), top {band} rows, pixels whose SGM lines lie inside the band",
                                     band_check_mask(H, W, band))
        ok &= c["end_px_err"]["pixels_differing"] == 0
        # the oracle band lies in the first band of the line scans (ScanBands: 1152 rows): the WHOLE map against the form that keeps the six min_p maps
        # instead of carries between bands and per-tile replays (census_tiles = 0, itself held to the oracle by tests/test_gpu_census_tiles.py)
        sv.set_test_option(d_src, "census_tiles", 0)
        try:
            maps_form = fn()["disp"]
        finally:
            sv.set_test_option(d_src, "census_tiles", 1)
        c["whole_map_vs_maps_form"] = {"pixels_differing": int((maps_form != res["disp"]).sum().item()), "of": int(maps_form.numel()),
                                       "checked_against": "the same call with census_tiles = 0 (six min_p maps, no band-to-band carries), every row band of the frame"}
        ok &= c["whole_map_vs_maps_form"]["pixels_differing"] == 0
        del maps_form
        c["cpu_baseline"] = cpu_band_baseline(so.CENSUS, so.COST, src, tgt, r, D, 8, rows=64, cols=1024)
    c["wall_s"] = round(time.perf_counter() - t0, 2)
    out["C5_one_gpu"] = c
    return out, ok


def reference_benchmark_rows(sv, dev, cpu_legs=True):
    """The rows of the reference's OWN benchmark program (test/benchmarks/benchmarkCrossCorrelationAlgorithms.cpp: local matching :433-459,
    with a compressor :461-531, SGM :533-550 -- the 1080p rows are the ones it comments out as "too slow" --, hierarchical :362-402), ZNCC as
    there, at its 480x640 / 160-disparity and 1080x1920 / 320-disparity sizes, through the per-function entry points on device arrays (the
    benchmark's own call sequence; volumes in HBM between the calls).  A few steps each; the 480x640 SGM row is checked against the oracle
    on the whole frame.  tools/bench_reference_rows.py has all of them."""
    import oracle as so
    from helpers import parallax_pair
    MF = sv.matchingFunctions
    strat = sv.matchFuncStrategy(MF.ZNCC)
    out, ok = {}, True
    t0 = time.perf_counter()
    for res, (H, W, r, D, steps) in {"480x640_5x5_disp160": (480, 640, 2, 160, 10), "1080x1920_7x7_disp320": (1080, 1920, 3, 320, 3)}.items():
        src, tgt, _ = parallax_pair(H, W, H // 3, H // 3, W // 3, D // 16, D // 4, 11)
        d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)

        def chain(ndir):
            cv = sv.unfoldBasedCostVolume(MF.ZNCC, d_tgt, d_src, r, r, D)
            if ndir:
                cv = sv.sgmCostVolume(ndir, strat, cv, 0.001, 0.01, None, 100.0)
            return sv.selectedIndexToDisp(sv.extractSelectedIndex(strat, cv), 0)

        comp = sv.UnFoldCompressor(sv.CompressorGenerators.GrPix17R3Filter())
        rows = {}
        rows["local_zncc_ms"] = round(_time_steps(lambda: chain(0), steps)[0], 4)
        # (the volume carries its winner to extractSelectedIndex: what the C++ drop-in headers do on DeviceArray for Score-strategy functions)
        rows["local_zncc_winner_carried_ms"] = round(_time_steps(lambda: sv.selectedIndexToDisp(sv.extractSelectedIndex(
            strat, sv.unfoldBasedCostVolume(MF.ZNCC, d_tgt, d_src, r, r, D, keep_winner=True)), 0), steps)[0], 4)
        rows["sgm4_zncc_ms"] = round(_time_steps(lambda: chain(4), steps)[0], 4)
        ms8, disp8 = _time_steps(lambda: chain(8), steps)
        rows["sgm8_zncc_ms"] = round(ms8, 4)
        rows["sgm8_zncc_Mdisparities_per_s"] = round(W * H * D / ms8 / 1e3, 1)
        rows["compressor_GrPix17R3_zncc_ms"] = round(_time_steps(lambda: sv.selectedIndexToDisp(sv.extractSelectedIndex(
            strat, sv.unfoldBasedCostVolumeCompressed(MF.ZNCC, d_tgt, d_src, comp, D)), 0), steps)[0], 4)
        rows["hierarchical_depth2_zncc_ms"] = round(_time_steps(lambda: sv.hiearchicalTruncatedCostVolume(MF.ZNCC, 2, d_tgt, d_src, r, r, D, 2), steps)[0], 4)
        if cpu_legs and H < 1000:
            tc = time.perf_counter()
            cvh = sv.unfoldBasedCostVolume(MF.ZNCC, d_tgt, d_src, r, r, D).cpu().numpy()  # (float costs agree with the oracle's to 1e-4: the maps are compared on the library's own volume)
            want = so.index_to_disp(so.extract_index(so.sgm(cvh, 8, int(strat), 0.001, 0.01, (0, 0, 0, 0), 100.0), int(strat)), so.RIGHT_TO_LEFT)
            rows["sgm8_end_px_err"] = end_px_err(disp8.cpu().numpy(), want, "oracle (port) SGM + argmax on the library's cost volume, whole frame")
            rows["sgm8_oracle_ms"] = round((time.perf_counter() - tc) * 1e3, 1)
            ok &= rows["sgm8_end_px_err"]["pixels_differing"] == 0
        out[res] = rows
        del d_src, d_tgt
        torch.cuda.empty_cache()
    out["wall_s"] = round(time.perf_counter() - t0, 2)
    return out, ok


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(argv, gpus, port):
    """The command line the driver's documented N > 1 form spells out, for `gpus` ranks of this script with the caller's own flags."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def needs_launch(gpus, environ):
    """A bare `python bench.py --gpus N` (N > 1, no rank environment) starts its own ranks; a rank started by torch.distributed.run
    (RANK / WORLD_SIZE set) never does."""
    return gpus > 1 and "WORLD_SIZE" not in environ and "RANK" not in environ


def launch_ranks(argv, gpus):
    """Parent of a bare N > 1 run.  It makes NO torch.cuda / HIP call (torch is not even imported here): the N ranks are fresh child
    processes of `python -m torch.distributed.run`, nothing that has touched the GPU is ever re-exec'd.  The children's stderr passes
    through as it comes; their stdout is relayed line by line, and the exit code is the launcher's (non-zero when any rank failed)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    proc = subprocess.Popen(launch_command(argv, gpus, free_port()), stdout=subprocess.PIPE, env=env, text=True)
    for ln in proc.stdout:
        sys.stdout.write(ln)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=7, help="the K-step timed region is run this many times (each bracketed by barrier + synchronize, "
                    "max over ranks); ms_per_step / value are the MEDIAN region, min and max are in the line (K = 20 is a 1.7 ms window)")
    ap.add_argument("--no-configs", action="store_true", help="N = 1: skip the other BASELINE configurations (C1, C2, C4, one-GPU C5) after the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle legs (cpu_baseline and end_px_err)")
    ap.add_argument("--no-api-chain", action="store_true", help="skip the per-function (volume-materialising) chain")
    ap.add_argument("--no-host-chain", action="store_true", help="skip the per-function chain on HOST arrays (numpy in, numpy out of every call: 8.5 GB over PCIe)")
    ap.add_argument("--materialize", action="store_true", help="also write the cost volume and the SGM volume (float32) to HBM")
    ap.add_argument("--winner-identity", action="store_true", help="time the product's default path instead: in the exact regime the winner of a "
                    "pixel does not depend on the per-pass minima, so disparity-only calls run no line recurrence at all (DESIGN.md 4.1); the "
                    "default run reports that figure as the `winner_identity` sub-object and times the recurrences")
    ap.add_argument("--c-abi-exchange", action="store_true", help="N > 1, after the timed regions: also run frames whose exchange goes through "
                    "svh_census_exchange_keys on a communicator of the process's own (what a C++ host does: tools/bench_sharded.cpp) and report "
                    "them as `c_abi_rccl_exchange`.  Off by default: the driver's run creates no second communicator")
    ap.add_argument("--two-frames", action="store_true", help="after the timed region, also time consecutive frames alternating between two "
                    "HIP streams (information only; kept out of the default run so that a profiler's per-kernel averages of this command "
                    "describe the single-stream kernels)")
    args = ap.parse_args()

    if needs_launch(args.gpus, os.environ):
        sys.exit(launch_ranks(sys.argv[1:], args.gpus))
    global torch
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    # Rehearsal switches for a one-GPU box (never set by the driver): all ranks on device 0 and a gloo process group, so the
    # N > 1 code path (shard ranges, key all-reduce, max-over-ranks timing, rank-0 JSON) can be exercised without N GPUs.
    rehearsal = os.environ.get("SVH_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import libstevi_amd as sv
    from helpers import parallax_pair

    wl = dict(C3 if world == 1 else C5)
    if os.environ.get("SVH_BENCH_SMALL") == "1" and world > 1:  # rehearsal on a box that should not spend minutes on C5
        wl.update(W=2048, H=1080, side=320, v=320, h=380, bg=8, sq=64)
    if wl["D"] > wl["W"]:
        raise SystemExit("refusing a disparity range wider than the image")
    # every rank holds the same pair: with N > 1 the ranks cooperate on ONE problem (disparity shards)
    t_setup = time.perf_counter()
    src, tgt, _ = parallax_pair(wl["H"], wl["W"], wl["side"], wl["v"], wl["h"], wl["bg"], wl["sq"], wl["seed"])
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    wall = {"pair_generation_s": round(time.perf_counter() - t_setup, 2)}  # wall-clock of the untimed legs, for whoever sets a timeout

    pipe = None
    if world > 1:
        # north_star: the cost volume shards over the disparity axis, one RCCL all-reduce for the per-pixel argmin.  Strong
        # scaling: the frame and its range are fixed (BASELINE.json configs[4]), a rank sweeps D / N disparities (whole 32-wide
        # row tiles of the matrix-core sweep where D / 32 divides).  A step submits one frame; its exchange runs on RCCL's stream
        # under the next frame's key kernels and its result comes back with the next submit (the last one with the flush inside
        # the timed region): K steps = K frames done.
        from libstevi_amd import sharded
        pipe = sharded.ShardedStereoPipeline(wl["h_r"], wl["v_r"], wl["D"], sgmDirections=wl["sgm"], P1=wl["P1"], P2=wl["P2"], Pout=wl["Pout"],
                                             align=32)

    # SURVEY.md 8(a) asks that the five passes of the aggregation be implemented and run, and that a reported throughput say whether it
    # was measured with them or with the recurrence-free identity argmin S = argmin[(1 + n) C + n Pout oob].  The headline is measured
    # WITH the recurrences (sweep -> g -> line scans -> six min_p maps -> finalize); the identity path -- the library's default for
    # disparity-only calls -- is timed next to it (`winner_identity`).  --winner-identity swaps the two.
    args.with_line_scans = not args.winner_identity
    if args.with_line_scans:
        sv.set_option(d_src, "census_winner_shortcut", 0)

    def step():
        if world > 1:
            return pipe.submit(d_tgt, d_src)
        # (target, source) passed as (img_l, img_r) like benchmarkCrossCorrelationAlgorithms.cpp:93
        return sv.stereoMatch(sv.matchingFunctions.CENSUS, d_tgt, d_src, wl["h_r"], wl["v_r"], wl["D"], sgmDirections=wl["sgm"],
                              P1=wl["P1"], P2=wl["P2"], Pout=wl["Pout"], want_cv=args.materialize, want_sgm_cv=args.materialize)

    def drain():
        return pipe.flush() if world > 1 else None

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # one cold step (first-touch allocations), then warm-up with hipEvents around EVERY kernel: per-kernel breakdown
    # and the dominant kernel of the step
    step()
    drain()
    sv.profile_reset(d_src)
    sv.profile_enable(d_src, True)
    for _ in range(max(args.warmup, 1)):
        out = step()
    drain()
    prof_all = sv.profile_collect(d_src)
    n_warm = max(args.warmup, 1)
    dom_name = max(prof_all.items(), key=lambda kv: kv[1][0])[0]
    # What an event pair costs by itself: two events with nothing between them, inside a busy stream.  A bracketed launch reads
    # this much longer than the kernel ran (rocprofv3's kernel trace, profiles/, has the kernel alone).
    pair_ms = 0.0
    for _ in range(10):
        step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        step()
        torch.cuda.synchronize()
        pair_ms += e0.elapsed_time(e1) / 10
    drain()
    # timed region: events only around the dominant kernel, and only around every `every`-th launch of it (an event pair is ~5 us
    # of stream time: a bracket around each launch of one kernel would cost a tenth of a 0.1 ms step)
    every = max(1, min(10, args.steps // 5))
    sv.profile_reset(d_src)
    sv.profile_enable(d_src, True, only=dom_name, every=every)
    # EXACTLY K steps per timed region, barrier + synchronize on both sides, max over ranks -- R regions, the median reported (at the
    # driver's K = 20 one region of C3 is a 1.7 ms window; `ms_per_step_min` / `_max` say how far single regions scatter)
    repeats = max(1, args.repeats)
    regions = []
    for _ in range(repeats):
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        if world > 1:
            out = drain()  # the last frame's exchange and finish belong to the timed region
        sync()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        regions.append(float(t.item()))
    sv.profile_enable(d_src, False)
    prof = sv.profile_collect(d_src)
    elapsed = float(np.median(regions))
    disp_dev = out["disp"]
    checksum = int(disp_dev.sum().item())

    # ---- N > 1: the phases of one frame, un-overlapped, outside the timed region (events on the compute stream; the exchange is
    # bracketed with the compute stream waiting for RCCL's stream, so its figure is what a non-pipelined frame would pay)
    phases = None
    if world > 1:
        from libstevi_amd import correlation as _c
        from libstevi_amd import sharded
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        acc = [0.0, 0.0, 0.0]
        plane0_only = pipe._plane0_only(d_tgt, d_src)
        reps = 3
        for _ in range(reps):
            sync()
            ev[0].record()
            keys = pipe._keys(d_tgt, d_src)
            ev[1].record()
            keys = sharded.exchange_keys(keys, plane0_only, None, async_op=False).wait()
            ev[2].record()
            pipe._finish(d_tgt, d_src, keys)
            ev[3].record()
            torch.cuda.synchronize()
            for k in range(3):
                acc[k] += ev[k].elapsed_time(ev[k + 1])
        ph = torch.tensor(acc, dtype=torch.float64, device=dev) / reps
        dist.all_reduce(ph, op=dist.ReduceOp.MAX)
        px = wl["W"] * wl["H"]
        phases = {"keys_ms": round(float(ph[0]), 4), "exchange_ms": round(float(ph[1]), 4), "finish_ms": round(float(ph[2]), 4),
                  "exchange_bytes_per_frame": int(px * (4 if plane0_only else 8)), "plane0_only": bool(plane0_only),
                  "shard_disparities": pipe.shard[1], "note": "max over ranks, phases run back to back without overlap; the timed steps overlap the "
                                                              "exchange of frame k with the key kernels of frame k + 1"}

    # ---- N > 1, for information (the headline above stays BASELINE config 5's protocol): the same frame split by ROWS.  In the exact
    # regime the rows of the disparity map are independent (svh_census_band_match), so every rank computes its band of the whole
    # disparity range and no exchange is needed; every rank checks its band against the replicated map of the protocol above.
    row_bands = None
    if world > 1:
        rb = sharded.RowBandStereoPipeline(wl["h_r"], wl["v_r"], wl["D"], sgmDirections=wl["sgm"], P1=wl["P1"], P2=wl["P2"], Pout=wl["Pout"])
        band = rb.submit(d_tgt, d_src)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            band = rb.submit(d_tgt, d_src)
        sync()
        trb = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(trb, op=dist.ReduceOp.MAX)
        rows = rb.rows_of(d_tgt, d_src)
        same = torch.tensor([int(torch.equal(band, disp_dev[rows[0]:rows[0] + rows[1]]))], device=dev)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        ms_rb = float(trb.item()) / args.steps * 1e3
        row_bands = {"ms_per_step": round(ms_rb, 4), "Mdisparities_per_s": round(wl["W"] * wl["H"] * wl["D"] / ms_rb / 1e3, 1),
                     "sgm_recurrences_timed": False,
                     "rows_per_gpu": rows[1], "collective": None, "result": "rank r holds rows band_range(H, r, N) of the map",
                     "bands_equal_the_replicated_map": bool(same.item()),
                     "note": "SECONDARY, never the headline: not BASELINE's protocol (that one is `disparity_split` = `value`) and not the same work (no line "
                             "recurrence is run): the disparity range is not sharded, the image rows are, which rests on "
                             "the recurrence-free winner identity (the winner of a pixel depends on its own costs and its position only, so no "
                             "line recurrence and no exchange); barrier + synchronize on both sides, max over ranks"}

    # ---- N > 1: the whole frame on ONE GPU (rank 0), outside the timed regions: the reference map of the full-frame end_px_err below
    # ... and, timed the same way (K steps between synchronisations, the other ranks idle at the barrier), the one-GPU time of the SAME
    # workload: the N = 1 bench line is C3, so a scaling point needs its own one-GPU denominator
    one_gpu, one_gpu_ms, rccl = None, None, None
    if world > 1:
        if rank == 0:
            def one():
                return sv.stereoMatch(sv.matchingFunctions.CENSUS, d_tgt, d_src, wl["h_r"], wl["v_r"], wl["D"], sgmDirections=wl["sgm"],
                                      P1=wl["P1"], P2=wl["P2"], Pout=wl["Pout"])["disp"]
            one_gpu = one()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                one_gpu = one()
            torch.cuda.synchronize()
            one_gpu_ms = (time.perf_counter() - t1) / args.steps * 1e3
        sync()
        # the communicator as the backend itself reports it: a SUM all-reduce of ones counts the ranks that took part
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        rccl = {"backend": dist.get_backend(), "ranks_in_all_reduce": int(ones.item()), "world_size": dist.get_world_size()}
        try:  # (informational fields: nothing here may take the headline down)
            idx = torch.tensor([dev_index], dtype=torch.int32, device=dev)
            all_idx = [torch.zeros_like(idx) for _ in range(world)]
            dist.all_gather(all_idx, idx)
            rccl["device_of_rank"] = [int(t.item()) for t in all_idx]
            rccl["device_name"] = torch.cuda.get_device_name(dev_index)
            rccl["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None
        except Exception as e:  # noqa: BLE001
            rccl["note"] = f"{type(e).__name__}: {e}"

    # ---- N > 1, opt-in: the exchange through the C ABI on a communicator of our own (the C++ host's path)
    c_abi = None
    if world > 1 and args.c_abi_exchange:
        try:
            comm = sharded.RcclCommunicator(rank=rank, world=world)
            plane0_only = pipe._plane0_only(d_tgt, d_src)

            def frame_c():
                keys = pipe._keys(d_tgt, d_src)
                sharded.exchange_keys_rccl(keys, plane0_only, comm)
                return pipe._finish(d_tgt, d_src, keys)
            frame_c()
            sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                out_c = frame_c()
            sync()
            tc = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            ms_c = float(tc.item()) / args.steps * 1e3
            c_abi = {"ms_per_step": round(ms_c, 4), "Mdisparities_per_s": round(wl["W"] * wl["H"] * wl["D"] / ms_c / 1e3, 1),
                     "same_disparity_map": bool(torch.equal(out_c["disp"], disp_dev)),
                     "note": "keys -> svh_census_exchange_keys (RCCL all-reduce enqueued by the library on the compute stream, no overlap with the next "
                             "frame) -> finish; communicator created through ctypes with the RCCL instance PyTorch loaded"}
            comm.destroy()
        except Exception as e:  # noqa: BLE001 -- an optional leg must not take the headline down
            c_abi = {"error": f"{type(e).__name__}: {e}"}

    rc = 0
    if rank == 0:
        voxels = wl["W"] * wl["H"] * wl["D"]
        # The headline is ONE protocol for every N: BASELINE's disparity split with the SGM recurrences run, as at N = 1 (the
        # exchange-free row split computes the recurrence-free winner identity, different work, and is only ever a secondary entry).
        headline_split = "disparity"
        disparity_split = None
        if world > 1:
            disparity_split = {"ms_per_step": round(elapsed / args.steps * 1e3, 4), "Mdisparities_per_s": round(voxels * args.steps / elapsed / 1e6, 1),
                               "collective": "one int32 MIN all-reduce of the regional winner keys per frame (RCCL), overlapped with the next frame's key kernels",
                               "result": "every rank holds the whole map", "sgm_recurrences_timed": bool(args.with_line_scans)}
        value = voxels * args.steps / elapsed / 1e6
        wl1 = dict(wl, D=wl["D"] if world == 1 else pipe.shard[1])  # what one launch of a kernel processes on one GPU
        ms_per_step = elapsed / args.steps * 1e3
        # dominant kernel by accumulated event time inside the timed region
        dom_ms, dom_n = prof[dom_name]
        avg_raw_ms = dom_ms / max(dom_n, 1)
        # `avg_ms` is what the events read: the bracket.  An empty event pair reads `pair_ms`; around a kernel part of that cost hides behind
        # the kernel's own launch, so the bracket over-reads the kernel by something between 0 and pair_ms: `frac` is therefore a lower
        # bound, and `avg_ms_minus_half_pair` / `frac_minus_half_pair` carry the mid-point estimate (within 2 % of rocprofv3's kernel trace
        # of the same command on every box so far, profiles/).
        avg_ms = max(avg_raw_ms, 1e-6)
        avg_ms_mid = max(avg_raw_ms - 0.5 * pair_ms, 1e-6)
        alg = algorithmic_bytes(dom_name, wl1)
        traffic = load_measured_traffic(dom_name) if world == 1 else None
        hbm_model = {"bound": "hbm", "kernel": KERNEL_SYMBOL.get(dom_name, dom_name), "library_label": dom_name, "launches": dom_n, "avg_ms": round(avg_ms, 5), "avg_ms_bracket": round(avg_raw_ms, 5),
                     "event_pair_ms": round(pair_ms, 5), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "traffic": traffic, "model": "SURVEY.md 8(d) row C3 prices this launch at 28 B/voxel of volume traffic the fused design never generates"}
        if alg is not None:
            ach = alg / (avg_ms * 1e-3) / 1e9
            hbm_model.update({"achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBPS, 4), "algorithmic_bytes_per_launch": int(alg)})
        else:
            hbm_model.update({"achieved": None, "frac": None})
        roof, issue = hbm_model, None
        vox_launch = wl1["W"] * wl1["H"] * wl1["D"]
        bits = 32 * (((2 * wl1["h_r"] + 1) * (2 * wl1["v_r"] + 1) - 1) // 32)  # census bits that are written: 9x9 -> 64
        if dom_name in MFMA_PEAK_TOPS:
            # The sweep runs on the matrix cores: the Hamming distance of two 64-bit census records is a 64-term dot product of
            # +-1 operands (FP4, exact in the f32 accumulators), i.e. 128 operations per voxel;
            # these are the algorithmic operations of a launch.  The kernel issues (D/32 + 1) / (D/32) of them (the band of D
            # disparities is covered by D/32 + 1 row tiles of 32).
            peak = MFMA_PEAK_TOPS[dom_name]
            fp4 = True
            ops = 2.0 * bits * vox_launch
            ach = ops / (avg_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": KERNEL_SYMBOL.get(dom_name, dom_name), "library_label": dom_name,
                    "kernel_is": "the symbol rocprofv3 prints (profiles/r05_bench_kernel_stats.csv, profiles/traffic.json)", "launches": dom_n, "avg_ms": round(avg_ms, 5), "avg_ms_bracket": round(avg_raw_ms, 5),
                    "event_pair_ms": round(pair_ms, 5), "avg_ms_is": "the event bracket as read (over-reads the kernel by 0 .. one event pair)",
                    "avg_ms_minus_half_pair": round(avg_ms_mid, 5), "frac_minus_half_pair": round(ops / (avg_ms_mid * 1e-3) / 1e12 / peak, 4),
                    "bracketed": f"every {every}. launch inside the timed region", "achieved": round(ach, 1),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                    "traffic_source": "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, committed by the round that measured them (not measured in this run)",
                    "dtype": "fp4 (e2m1) operands +-1, f32 accumulators (exact integers)" if fp4 else "int8 multiply-accumulate, int32 accumulators (exact)",
                    "algorithmic_ops_per_launch": int(ops), "ops_per_voxel": 2 * bits,
                    "note": ("v_mfma_scale_f32_32x32x64_f8f6f4 (FP4 x FP4)" if fp4 else "v_mfma_i32_32x32x32_i8") +
                            "; peak = dense rate of the operand type at the nominal clock (the chip holds well under 2.4 GHz under this load)"}
        elif dom_name == "census_sweep":
            # The vector-ALU engine (svh_context_set_option census_sweep = 1): the roof that binds it is VALU issue.  Per 64 voxels a
            # SIMD issues 2 v_xor + 2 v_bcnt + 1 v_lshl_or + 1/2 v_min3; tools/ubench_valu.hip measures 2.76 / 4.58 / 4.46 / 4.55
            # cycles per wave-instruction and SIMD for them: 21.4 cycles per 64 voxels; 256 CUs x 4 SIMDs.
            cyc = 2 * 2.76 + 2 * 4.58 + 4.46 + 0.5 * 4.55
            peak_vox = 1024 * 2.4e9 / cyc * 64.0
            ach_vox = vox_launch / (avg_ms * 1e-3)
            issue = {"bound": "valu_issue", "kernel": KERNEL_SYMBOL.get(dom_name, dom_name), "cycles_per_64_voxels_per_simd": round(cyc, 2), "peak": round(peak_vox / 1e9, 1),
                     "achieved": round(ach_vox / 1e9, 1), "unit": "Gvoxels/s", "frac": round(ach_vox / peak_vox, 4),
                     "rates": "measured per-instruction issue rates, tools/ubench_valu.hip"}
        pipeline_alg = 28.0 * voxels + 60.0 * wl["W"] * wl["H"]
        # what no implementation of this step can go below on one GPU: the images in, the disparity map out (HBM), and the
        # Hamming dot products of every voxel on the fastest unit that can do them (matrix cores, FP4)
        compulsory = 12.0 * wl["W"] * wl["H"]
        lb_us = max(compulsory / (HBM_PEAK_GBPS * 1e9), 2.0 * bits * (voxels / world) / (FP4_PEAK_TOPS * 1e12)) * 1e6
        kernel_ms = {k: round(v[0] / n_warm, 4) for k, v in prof_all.items()}  # warm-up steps, every kernel bracketed (each + one event pair)
        shape = f"{wl['W']}x{wl['H']}"
        # The step a user sees, next to its dominant kernel: the step's time against the floor no implementation can go
        # below, and the HBM bytes the step's four kernels really move (the PMC passes under profiles/) against the compulsory 12 B/pixel
        step_traffic = None
        if world == 1 and args.with_line_scans:
            per_kernel = {k: load_measured_traffic(k) for k in ("census_transform", dom_name, "sgm_line_scans", "census_finalize")}
            if all(v is not None for v in per_kernel.values()):
                step_traffic = int(sum(per_kernel.values()))
        roof.update({"step_ms": round(ms_per_step, 4), "step_lower_bound_us": round(lb_us, 2), "frac_of_step_floor": round(lb_us / (ms_per_step * 1e3), 4),
                     "step_floor_model": "max(12 B/pixel of compulsory HBM traffic at 8 TB/s, 128 ops/voxel on the FP4 matrix cores at 10 POP/s)",
                     "hbm_bytes_per_step": step_traffic, "compulsory_bytes_per_step": int(compulsory),
                     "hbm_bytes_per_step_source": "profiles/traffic.json, the four kernels of the step (census words, keys, the g map and tile edges are written by one kernel and read by the next)"})
        line = {
            "metric": f"Mdisparities/s (W*H*D) for census+SGM, {'1080p' if wl['name'] == 'C3' else shape} D={wl['D']}; end-px-err vs ref",
            "value": round(value, 1), "unit": "Mdisparities/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "repeats": repeats, "ms_per_step_min": round(min(regions) / args.steps * 1e3, 4), "ms_per_step_max": round(max(regions) / args.steps * 1e3, 4),
            "timing": f"median of {repeats} timed regions of exactly {args.steps} steps, each bracketed by barrier + torch.cuda.synchronize(), max over ranks",
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "fp4 (e2m1) +-1 census operands on the matrix cores, f32 accumulators holding exact integers (no rounding anywhere on the path)", "data": "synthetic",
            "config": {"workload": f"{shape} synthetic parallax pair, census 9x9 + Hamming, 8-path SGM (P1=0.001,P2=0.01,Pout=100), "
                                   f"D={wl['D']}, argmin -> int32 disparity map (BASELINE.json configs[{2 if wl['name'] == 'C3' else 4}])",
                       "pipeline": ("svh_stereo_match fused, inputs and outputs resident in HBM" if world == 1 else
                                    "svh_census_shard_keys -> int32 MIN all-reduce -> svh_census_shard_finish, inputs and outputs resident in HBM") +
                                   ("; SGM line recurrences RUN (sweep -> g -> six line scans keeping the min_p values on tile edges -> per-pixel kernel replaying them per tile)" if args.with_line_scans else
                                    "; recurrence-free winner identity (the library's default for disparity-only calls): no line scans"),
                       "sgm_recurrences_timed": bool(args.with_line_scans),
                       "materialize_volumes": bool(args.materialize),
                       "parallelism": (f"disparity axis sharded over {world} GPUs ({pipe.shard[1]} of {wl['D']} disparities per GPU), one RCCL int32 MIN "
                                       "all-reduce of the regional winner keys per frame, overlapped with the next frame's key kernels "
                                       "(one exchange in flight); the finish (" + ("line scans on the reduced keys + finalize" if args.with_line_scans else
                                                                           "winner from the reduced keys") + ") replicated") if world > 1 else "single GPU"},
            "roofline": roof,
            "hbm_model_roofline": hbm_model if roof is not hbm_model else None,
            "pipeline_roofline": {"algorithmic_bytes_per_step": int(pipeline_alg), "achieved": round(pipeline_alg / (ms_per_step * 1e-3) / 1e9, 1),
                                  "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(pipeline_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                  "model": "SURVEY.md 8(d) C3: 28 B/voxel + 60 B/pixel (volumes in HBM; the fused step never writes them)"},
            "step_lower_bound": {"us": round(lb_us, 2), "frac_of_step": round(lb_us / (ms_per_step * 1e3), 4),
                                 "model": "max(12 B/pixel compulsory HBM traffic at 8 TB/s, 128 ops/voxel on the FP4 matrix cores at 10 POP/s per GPU)"},
            "issue_roofline": issue,
            "kernel_ms_per_step_warmup": kernel_ms,
            "disp_checksum": checksum,
        }
        if world > 1:
            line["rccl_ranks"] = rccl
            line["one_gpu_same_workload_ms"] = round(one_gpu_ms, 4)
            line["speedup"] = round(one_gpu_ms / ms_per_step, 4)
            line["efficiency"] = round(one_gpu_ms / ms_per_step / world, 4)
            line["scaling_note"] = ("speedup = one_gpu_same_workload_ms / ms_per_step: svh_stereo_match on rank 0 alone, whole disparity range of this same frame, same K "
                                    "steps, same run (the N = 1 bench line is BASELINE's C3, a different frame; do not draw a curve through both)")
        if phases is not None:
            line["frame_phases"] = phases
        if disparity_split is not None:
            line["headline_split"] = headline_split
            line["disparity_split"] = disparity_split
        if c_abi is not None:
            line["c_abi_rccl_exchange"] = c_abi
        if row_bands is not None:
            line["row_bands"] = row_bands
            if not row_bands["bands_equal_the_replicated_map"]:
                rc = 3
        got = disp_dev.cpu().numpy()
        if not args.no_cpu_baseline:
            import oracle as so
            so.set_num_threads(so.granted_cpus())  # the CPUs the cgroup grants (torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks)
            if world == 1:
                t_cpu = time.perf_counter()
                line["cpu_baseline"], want = cpu_baseline(wl, src, tgt)
                wall["cpu_baseline_s"] = round(time.perf_counter() - t_cpu, 2)
                line["end_px_err"] = end_px_err(got, want, "oracle (port), full frame")
            else:  # the oracle on the whole 8K frame would take about a minute: a row band, on the pixels a band decides ...
                # ... and the WHOLE frame against the map one GPU computes alone on the whole range in this same run (itself oracle-checked
                # at full size by tests/test_gpu_fullsize.py and, at N = 1, by this script)
                diff_one = int((disp_dev != one_gpu).sum().item())
                line["end_px_err_vs_one_gpu"] = {"pixels_differing": diff_one, "of": int(disp_dev.numel()), "checked_against": "svh_stereo_match on rank 0, whole range, same run"}
                if diff_one != 0:
                    rc = 3
                band = 48
                want, wall["oracle_band_s"] = oracle_disparity(wl, src, tgt, band + wl["v_r"], variant=1)
                wall["oracle_band_s"] = round(wall["oracle_band_s"], 2)
                line["end_px_err"] = end_px_err(got, want[:band], f"oracle (port), top {band} rows, pixels whose SGM lines lie inside the band",
                                                band_check_mask(wl["H"], wl["W"], band))
            if line["end_px_err"]["pixels_differing"] != 0:
                rc = 3
        else:
            line["end_px_err"] = None
        if world == 1 and args.with_line_scans:
            # The library's default path for this call, timed the same way (same K steps, same bracketing): the winner written by the
            # sweep from the identity argmin S = argmin[(1 + n) C + n Pout oob]: no g map, no line scans, no min_p maps, no finalize
            # launch (census_winner_shortcut = 1).  `value` above is measured with the recurrences run.
            sv.set_option(d_src, "census_winner_shortcut", 1)
            try:
                for _ in range(max(args.warmup, 1)):
                    out_id = step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    out_id = step()
                torch.cuda.synchronize()
                dt_id = (time.perf_counter() - t1) / args.steps
            finally:
                sv.set_option(d_src, "census_winner_shortcut", 0)
            line["winner_identity"] = {"ms_per_step": round(dt_id * 1e3, 4), "Mdisparities_per_s": round(voxels / dt_id / 1e6, 1),
                                       "same_disparity_map": bool(torch.equal(out_id["disp"], disp_dev)),
                                       "kernels": "census_grey_kernel, census_sweep_rl_kernel (writes the disparity map)",
                                       "note": "what svh_stereo_match does by default when only the disparity map is asked for"}
            if not line["winner_identity"]["same_disparity_map"]:
                rc = 3
        if world == 1 and args.two_frames:
            # Outside the timed region, for information: the same step with consecutive frames alternating between two HIP streams
            # (a video stream's deployment form).  The four kernels of a frame are each bound by instruction issue or by a
            # dependent-latency chain, not by HBM, so two frames in flight fill each other's idle issue slots.  `value` above stays
            # the plain single-stream figure.
            streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
            for st in streams:
                with torch.cuda.stream(st):  # (a context per stream: the same options as the headline's)
                    sv.set_option(d_src, "census_winner_shortcut", 0 if args.with_line_scans else 1)
                    step()
            torch.cuda.synchronize()
            reps = max(args.steps, 20)
            t1 = time.perf_counter()
            for k in range(reps):
                with torch.cuda.stream(streams[k & 1]):
                    out2 = step()
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t1) / reps
            line["two_frames_in_flight"] = {"streams": 2, "sgm_recurrences_timed": bool(args.with_line_scans), "ms_per_frame": round(dt2 * 1e3, 4), "Mdisparities_per_s": round(voxels / dt2 / 1e6, 1),
                                            "same_disparity_map": bool(torch.equal(out2["disp"], disp_dev))}
        if world == 1 and not args.no_api_chain:
            line["api_chain"], chain_disp = api_chain(sv, wl, d_tgt, d_src)
            line["api_chain"]["pixels_differing_from_fused"] = int((chain_disp != disp_dev).sum().item())
            if line["api_chain"]["pixels_differing_from_fused"] != 0:
                rc = 3
        if world == 1 and not args.no_host_chain:
            t_h = time.perf_counter()
            try:
                line["host_chain"], host_disp = host_chain(sv, wl, src, tgt)
                line["host_chain"]["pixels_differing_from_fused"] = int(np.count_nonzero(host_disp != got))
                if line["host_chain"]["pixels_differing_from_fused"] != 0:
                    rc = 3
                del host_disp
            except Exception as e:  # noqa: BLE001 -- report, never lose the headline to a secondary leg
                line["host_chain"] = {"error": f"{type(e).__name__}: {e}"}
                rc = 3
            wall["host_chain_s"] = round(time.perf_counter() - t_h, 2)
        if world == 1 and not args.no_configs:
            # the other BASELINE configurations, after (and outside) everything the headline times; C3 itself is the headline above
            torch.cuda.empty_cache()
            sv.set_option(disp_dev, "census_winner_shortcut", 0)  # as the headline: the SGM recurrences run
            try:
                line["configs"], cfg_ok = other_configs(sv, dev, cpu_legs=not args.no_cpu_baseline)
                if isinstance(line.get("cpu_baseline"), dict):  # the other configurations' CPU legs, next to the headline's
                    line["cpu_baseline"]["other_configs"] = {k: {kk: v["cpu_baseline"][kk] for kk in ("value", "unit", "cores", "sample", "linear_sgm_value", "extrapolated_full_frame_s")
                                                                 if kk in v["cpu_baseline"]} | {"gpu_Mdisparities_per_s": v.get("Mdisparities_per_s")}
                                                             for k, v in line["configs"].items() if isinstance(v, dict) and "cpu_baseline" in v}
                line["configs"]["C3"] = "the headline of this line (value, ms_per_step, roofline, cpu_baseline, end_px_err)"
                if not cfg_ok:
                    rc = 3
            except Exception as e:  # noqa: BLE001 -- report, never lose the headline to a secondary leg
                line["configs"] = {"error": f"{type(e).__name__}: {e}"}
                rc = 3
            try:
                line["reference_benchmark_rows"], rows_ok = reference_benchmark_rows(sv, dev, cpu_legs=not args.no_cpu_baseline)
                if not rows_ok:
                    rc = 3
            except Exception as e:  # noqa: BLE001
                line["reference_benchmark_rows"] = {"error": f"{type(e).__name__}: {e}"}
                rc = 3
        wall["total_s"] = round(time.perf_counter() - t_setup, 2)
        line["wall_clock_s"] = wall
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rc:
        print("bench.py: the disparity map differs from the reference chain (see end_px_err / api_chain)", file=sys.stderr)
        sys.exit(rc)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json): census 9x9 + 8-path SGM, 1920x1080, D=256.

One "step" = one pass of the fused device pipeline (svh_stereo_match) over one synthetic stereo pair that is
already resident in HBM: census transform of both images -> Hamming cost (evaluated on the fly) -> the
reference's five effective SGM passes -> winner -> disparity map (int32).  Output bit-exact vs the oracle
(tests/test_gpu_parity.py, tests/test_gpu_fullsize.py).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Metric: Mdisparities/s = W*H*D*steps / seconds / 1e6 summed over ranks.
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
import torch  # noqa: E402

# C3 of SURVEY.md section 8(d): 1920x1080, D=256, census 9x9, SGM-8 (P1=0.001, P2=0.01, Pout=100), seed 3
WORKLOAD = dict(W=1920, H=1080, D=256, h_r=4, v_r=4, sgm=8, P1=0.001, P2=0.01, Pout=100.0, seed=3,
                bg=8, sq=64, side=320, v=320, h=380)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# MI355X_MICROARCH.md, matrix cores (dense): BF16 ~2.5 PFLOP/s; int8 runs at twice the BF16 rate, block-scaled FP4 at four times
MFMA_PEAK_TOPS = {"census_sweep_mfma": 5000.0, "census_sweep_mfma4": 10000.0}


def visited_voxels_per_pass(H, W, D, n_pass):
    """voxels touched by each effective SGM pass, zero margins"""
    ii, jj = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    masks = [np.ones((H, W), bool), np.ones((H, W), bool), ii >= jj, jj >= ii, ii + jj < W, ii + jj < H]
    return [int(m.sum()) * D for m in masks[:n_pass]]


def algorithmic_bytes(kernel, wl):
    """Algorithmic bytes per launch of `kernel` under the model of SURVEY.md section 8(d), row C3 (DESIGN.md "Roofline"):
    C as u8, S as i16, each of the five effective SGM passes = one read of C + one read-modify-write of S:
        28 B/voxel = 1 (C write) + 5 x (1 + 4) + 2 (final S read),   60 B/pixel = 8 (images) + 48 (census words) + 4 (disparity).
    The fused pipeline has ONE voxel-sized kernel (census_sweep); all 28 B/voxel of the model are work it replaces.  The
    per-pixel kernels carry the per-pixel bytes."""
    H, W, D = wl["H"], wl["W"], wl["D"]
    px, vox = H * W, H * W * D
    table = {
        "census_sweep": 28.0 * vox,
        "census_sweep_mfma": 28.0 * vox,
        "census_sweep_mfma4": 28.0 * vox,
        "census_transform": (4.0 + 24.0) * px,      # one image: 4 B read + nW = 3 words written and read back
        "sgm_line_scans": 4.0 * px + 6 * 4.0 * px,   # g read + six min_p maps written
        "census_finalize": (8.0 + 24.0 + 4.0) * px,  # keys + six min_p maps + disparity
    }
    return table.get(kernel)


def load_measured_traffic(kernel):
    """HBM bytes per launch of `kernel` measured with rocprofv3 --pmc (FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections
    of MI355X_MICROARCH.md) on this workload; the summary is committed under profiles/ by the round that measured it."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def cpu_baseline(wl, budget_s=18.0):
    """Oracle (CPU restatement of the reference, literal O(D^2) SGM loops) on a bounded row band of the workload."""
    import oracle as so
    from helpers import parallax_pair
    src, tgt, _ = parallax_pair(wl["H"], wl["W"], wl["side"], wl["v"], wl["h"], wl["bg"], wl["sq"], wl["seed"])
    D, hr, vr = wl["D"], wl["h_r"], wl["v_r"]

    def run(rows, variant):
        s, t = np.ascontiguousarray(src[:rows]), np.ascontiguousarray(tgt[:rows])
        t0 = time.perf_counter()
        cv = so.unfold_cost_volume(so.CENSUS, t, s, hr, vr, D)
        vol = so.sgm(cv, wl["sgm"], so.COST, wl["P1"], wl["P2"], (0, 0, 0, 0), wl["Pout"], variant=variant)
        so.index_to_disp(so.extract_index(vol, so.COST))
        return time.perf_counter() - t0

    out = {}
    for variant, key in ((0, "literal"), (1, "linear")):
        rows = 4
        dt = run(rows, variant)
        target = budget_s * (0.75 if variant == 0 else 0.25)
        rows2 = int(min(wl["H"], max(rows, rows * target / max(dt, 1e-6))))
        if rows2 > rows:
            dt = run(rows2, variant)
            rows = rows2
        out[key] = (rows * wl["W"] * D / dt / 1e6, rows, dt)
    cores = so.num_threads()
    lit, lin = out["literal"], out["linear"]
    return {
        "value": round(lit[0], 3), "unit": "Mdisparities/s", "cores": cores, "kind": "port",
        "sample": f"top {lit[1]} rows of the same {wl['W']}x{wl['H']} pair, D={D}: census 9x9 volume + SGM-8 (literal O(D^2) loops of "
                  f"sgm.h:269-295) + argmin, {lit[2]:.1f} s, OpenMP over rows/lines",
        "linear_sgm_value": round(lin[0], 3),
        "linear_sgm_sample": f"top {lin[1]} rows, O(D) SGM restatement (bitwise identical), {lin[2]:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--materialize", action="store_true", help="also write the cost volume and the SGM volume (float32) to HBM")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 "
                             "--master-port P bench.py --gpus N ...")
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

    wl = dict(WORKLOAD)
    # every rank holds the same pair: with N > 1 the ranks cooperate on ONE problem (disparity shards)
    src, tgt, _ = parallax_pair(wl["H"], wl["W"], wl["side"], wl["v"], wl["h"], wl["bg"], wl["sq"], wl["seed"])
    d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)

    if world > 1:
        # north_star: the cost volume shards over the disparity axis, one RCCL all-reduce for the per-pixel argmin.
        # Weak scaling: every GPU keeps the single-GPU slice (D=256), the searched range grows to 256 x N.
        # A step submits one frame; its exchange runs on RCCL's stream under the next frame's key kernels and its result
        # comes back with the next submit (the last one with the flush inside the timed region): K steps = K frames done.
        from libstevi_amd import sharded
        wl["D"] = WORKLOAD["D"] * world
        pipe = sharded.ShardedStereoPipeline(wl["h_r"], wl["v_r"], wl["D"], sgmDirections=wl["sgm"], P1=wl["P1"], P2=wl["P2"], Pout=wl["Pout"])

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
    # timed region: events only around the dominant kernel (bracketing all six launches costs ~20 % of a 0.23 ms step)
    sv.profile_reset(d_src)
    sv.profile_enable(d_src, True, only=dom_name)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if world > 1:
        out = drain()  # the last frame's exchange and finish belong to the timed region
    sync()
    elapsed = time.perf_counter() - t0
    sv.profile_enable(d_src, False)
    prof = sv.profile_collect(d_src)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    checksum = int(out["disp"].sum().item())

    if rank == 0:
        voxels = wl["W"] * wl["H"] * wl["D"]  # D = 256 x N for N > 1: the whole job's voxels
        value = voxels * args.steps / elapsed / 1e6
        wl1 = dict(wl, D=WORKLOAD["D"])        # what one launch of a kernel processes on one GPU
        ms_per_step = elapsed / args.steps * 1e3
        # dominant kernel by accumulated event time inside the timed region
        dom_ms, dom_n = prof[dom_name]
        avg_ms = dom_ms / max(dom_n, 1)
        alg = algorithmic_bytes(dom_name, wl1)
        hbm_model = {"bound": "hbm", "kernel": dom_name, "launches": dom_n, "avg_ms": round(avg_ms, 5), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "traffic": None, "model": "SURVEY.md 8(d) row C3 prices this launch at 28 B/voxel of volume traffic the fused design never generates"}
        if alg is not None:
            ach = alg / (avg_ms * 1e-3) / 1e9
            hbm_model.update({"achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBPS, 4), "algorithmic_bytes_per_launch": int(alg)})
        else:
            hbm_model.update({"achieved": None, "frac": None})
        traffic = load_measured_traffic(dom_name)
        if traffic is not None:
            hbm_model["traffic"] = traffic
        roof, issue = hbm_model, None
        vox_launch = wl1["W"] * wl1["H"] * wl1["D"]
        if dom_name in MFMA_PEAK_TOPS:
            # The sweep runs on the matrix cores: the Hamming distance of two 64-bit census records is a 64-term dot product of
            # 0 / +-1 operands (svh_census_sweep_mfma4.hip: FP4 operands, exact in the f32 accumulators; svh_census_sweep_mfma.hip:
            # int8), i.e. 128 operations per voxel; these are the algorithmic operations of a launch.  The kernel issues 9/8 of them
            # (the band of D disparities is covered by D/32 + 1 row tiles of 32).
            peak = MFMA_PEAK_TOPS[dom_name]
            fp4 = dom_name.endswith("4")
            bits = 32 * (((2 * wl1["h_r"] + 1) * (2 * wl1["v_r"] + 1) - 1) // 32)  # census words that are written: 9x9 -> 64 bits
            ops = 2.0 * bits * vox_launch
            ach = ops / (avg_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": dom_name, "launches": dom_n, "avg_ms": round(avg_ms, 5), "achieved": round(ach, 1),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                    "dtype": "fp4 (e2m1) operands 0 / +-1, f32 accumulators (exact integers)" if fp4 else "int8 multiply-accumulate, int32 accumulators (exact)",
                    "algorithmic_ops_per_launch": int(ops), "ops_per_voxel": 2 * bits,
                    "note": ("v_mfma_scale_f32_32x32x64_f8f6f4 (FP4 x FP4)" if fp4 else "v_mfma_i32_32x32x32_i8") +
                            "; peak = dense rate of the operand type at the nominal clock (the chip holds ~1.7 GHz under this load)"}
        elif dom_name == "census_sweep":
            # The vector-ALU engine (svh_context_set_option census_sweep = 1): the roof that binds it is VALU issue.  Per 64 voxels a
            # SIMD issues 2 v_xor + 2 v_bcnt + 1 v_lshl_or + 1/2 v_min3; tools/ubench_valu.hip measures 2.76 / 4.58 / 4.46 / 4.55
            # cycles per wave-instruction and SIMD for them: 21.4 cycles per 64 voxels; 256 CUs x 4 SIMDs.
            cyc = 2 * 2.76 + 2 * 4.58 + 4.46 + 0.5 * 4.55
            peak_vox = 1024 * 2.4e9 / cyc * 64.0
            ach_vox = vox_launch / (avg_ms * 1e-3)
            issue = {"bound": "valu_issue", "kernel": dom_name, "cycles_per_64_voxels_per_simd": round(cyc, 2), "peak": round(peak_vox / 1e9, 1),
                     "achieved": round(ach_vox / 1e9, 1), "unit": "Gvoxels/s", "frac": round(ach_vox / peak_vox, 4),
                     "rates": "measured per-instruction issue rates, tools/ubench_valu.hip"}
        pipeline_alg = 28.0 * voxels + 60.0 * wl["W"] * wl["H"]
        kernel_ms = {k: round(v[0] / n_warm, 4) for k, v in prof_all.items()}  # from the warm-up steps (all kernels bracketed)
        line = {
            "metric": "Mdisparities/s (W*H*D) for census+SGM, 1080p D=256", "value": round(value, 1), "unit": "Mdisparities/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "1920x1080 synthetic parallax pair, census 9x9 + Hamming, 8-path SGM (P1=0.001,P2=0.01,Pout=100), "
                                   "D=256, argmin -> int32 disparity map (BASELINE.json configs[2])",
                       "pipeline": "svh_stereo_match fused, inputs and outputs resident in HBM",
                       "materialize_volumes": bool(args.materialize),
                       "parallelism": (f"disparity axis sharded over {world} GPUs (D=256 per GPU, {wl['D']} in total), one RCCL int32 MIN "
                                       "all-reduce of the regional winner keys per frame (4 B/pixel: the Pout region's key is global as "
                                       "written), overlapped with the next frame's key kernels (one exchange in flight)") if world > 1 else "single GPU"},
            "roofline": roof,
            "hbm_model_roofline": hbm_model if roof is not hbm_model else None,
            "pipeline_roofline": {"algorithmic_bytes_per_step": int(pipeline_alg), "achieved": round(pipeline_alg / (ms_per_step * 1e-3) / 1e9, 1),
                                  "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(pipeline_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                  "model": "SURVEY.md 8(d) C3: 28 B/voxel + 60 B/pixel"},
            "issue_roofline": issue,
            "kernel_ms_per_step_warmup": kernel_ms,
            "disp_checksum": checksum,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

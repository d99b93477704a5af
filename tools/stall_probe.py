#!/usr/bin/env python3
"""Where do the 5 - 30 ms outliers of hiearchicalTruncatedCostVolume at 480x640 come from (one timed call in five reads
25 - 30 ms for 0.2 ms of kernels)?  Per-call wall times of the same call, (a) in a quiet process, (b) right after an oracle call on every
host thread (what tools/bench_hierarchical.py did before each timed loop), (c) right after an oracle call on as many threads as the
cgroup grants.  Prints the cgroup's CPU quota and its throttling counters around each leg."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import libstevi_amd as sv
import oracle as so
from helpers import parallax_pair


def cgroup():
    out = {}
    for name in ("cpu.max", "cpu.stat"):
        for base in ("/sys/fs/cgroup", "/sys/fs/cgroup/cpu"):
            try:
                with open(os.path.join(base, name)) as f:
                    out[name] = f.read().split()
                break
            except OSError:
                pass
    st = out.get("cpu.stat", [])
    d = {st[k]: int(st[k + 1]) for k in range(0, len(st) - 1, 2) if st[k + 1].isdigit()}
    return {"cpu.max": out.get("cpu.max"), "nr_throttled": d.get("nr_throttled"), "throttled_usec": d.get("throttled_usec"), "affinity": len(os.sched_getaffinity(0)),
            "cpu_count": os.cpu_count()}


MF = sv.matchingFunctions
dev = torch.device("cuda:0")
W, H, r, disp_w, depth = 640, 480, 2, 160, 3
src, tgt, _ = parallax_pair(H, W, H // 3, H // 3, W // 3, disp_w // 16, disp_w // 4, 11)
d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
run = lambda: sv.hiearchicalTruncatedCostVolume(MF.ZNCC, depth, d_tgt, d_src, r, r, disp_w, 2)  # noqa: E731
for _ in range(3):
    run()
torch.cuda.synchronize()


def leg(name, before=None, calls=200):
    c0 = cgroup()
    if before is not None:
        before()
    t = []
    for _ in range(calls):
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) * 1e3)
    c1 = cgroup()
    t = np.array(t)
    print(json.dumps({"leg": name, "calls": calls, "median_ms": round(float(np.median(t)), 3), "p99_ms": round(float(np.percentile(t, 99)), 3), "max_ms": round(float(t.max()), 3),
                      "calls_over_2ms": int((t > 2).sum()), "first_10_ms": [round(float(v), 3) for v in t[:10]], "oracle_threads": so.num_threads(),
                      "throttled_periods_during_leg": None if c0["nr_throttled"] is None else c1["nr_throttled"] - c0["nr_throttled"],
                      "throttled_ms_during_leg": None if c0["throttled_usec"] is None else (c1["throttled_usec"] - c0["throttled_usec"]) / 1e3, "cgroup": c1}), flush=True)


def oracle_call():
    so.hierarchical_truncated_cv(int(MF.ZNCC), depth, tgt, src, r, r, disp_w, 2)


leg("quiet process")
full = 128  # what the oracle took by default until round 5 (omp_get_max_threads on a 256-CPU host)
so.set_num_threads(full)
leg(f"right after an oracle call on {full} threads", oracle_call)
leg("quiet again")
so.set_num_threads(16)
leg("right after an oracle call on 16 threads", oracle_call)
so.set_num_threads(full)
leg(f"right after an oracle call on {full} threads, again", oracle_call)

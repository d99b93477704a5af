#!/usr/bin/env python3
"""Folds two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) into
profiles/traffic.json: HBM bytes per launch and kernel.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/traffic.json

Corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at
64 bytes, so it is doubled; WRITE_SIZE is exact for streaming stores.  The doubling is calibrated for 16-byte-per-lane
streams; the kernels here read 4 to 16 bytes per lane, so the read side is an upper bound.  WRITE_SIZE matches the byte
counts of the known outputs exactly (e.g. census words: 1080 x 1920 x 8 B = 16200 KiB)."""
import collections
import csv
import glob
import json
import os
import re
import sys

# keys of the summary = the kernel symbols rocprofv3 prints (bench.py's KERNEL_SYMBOL maps the library's profile labels onto them)


def collect(directory, counter):
    acc = collections.defaultdict(list)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            if m and "svh" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                name = m.group(1)
                # the sweep that writes the disparity map itself (template argument WINNER = true: the winner_identity leg of bench.py) is
                # another kernel as far as bytes go: it writes 4 B / pixel instead of keys + g
                if name in ("census_sweep_pm_kernel", "census_sweep_rl_kernel") and re.search(r"census_sweep_\w+_kernel<[^>]*\btrue\b", r["Kernel_Name"]):
                    name += ":winner"
                acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out[k] = {"FETCH_SIZE_KiB_raw": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counter passes (SQ instruction mix and wait breakdown).

    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
              --kernel-trace --output-format csv -d gpurun_out/pmc_sq1 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-api-chain
    python tools/pmc_sq_summary.py gpurun_out/pmc_sq1 [gpurun_out/pmc_sq2 ...] > profiles/rNN_sq_counters.json

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md, cycle constants)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
                if m and "svh" in r["Kernel_Name"]:
                    name = m.group(1)
                    if re.search(r"census_sweep_\w+_kernel<[^>]*\btrue\b", r["Kernel_Name"]):
                        name += "_winner"  # (WINNER = true: writes the disparity map itself)
                    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in sorted(acc.items()):
        out[k] = {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}
        w = out[k].get("SQ_WAVES")
        if w:
            for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM"):
                if c in out[k]:
                    out[k][c + "_per_wave"] = round(out[k][c] / w, 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

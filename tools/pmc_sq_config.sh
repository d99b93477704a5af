# SQ instruction mix of the kernels of one tools/bench_configs.py workload (a development tool, on the GPU box, from the repository root):
#   bash tools/pmc_sq_config.sh <config> [kernel name substring ...]
# Three rocprofv3 --pmc passes (counters only, no trace domains beside --kernel-trace), averaged per kernel and launch, with the counts
# per wave.  A CU issues at most one scalar and one vector instruction per cycle for its four SIMDs in turn: a kernel whose scalar count
# per wave approaches its vector count is bound by scalar issue, whatever its bytes say.
set -e
CFG=$1; shift
R=$(pwd)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/sqc1 -- python3 $R/tools/bench_configs.py $CFG > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/sqc2 -- python3 $R/tools/bench_configs.py $CFG > /dev/null 2>&1
python3 - "$@" <<'PY'
import collections, csv, glob, json, re, sys
want = sys.argv[1:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/sqc1", "/tmp/sqc2"):
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            if not m or "svh" not in r["Kernel_Name"]:
                continue
            if want and not any(w in r["Kernel_Name"] for w in want):
                continue
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    w = avg.get("SQ_WAVES", 0) or 1
    out = {"launches": len(cs.get("SQ_WAVES", [])), "waves": int(w)}
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS"):
        if c in avg:
            out[c.replace("SQ_INSTS", "per_wave").lower()] = round(avg[c] / w, 1)
    if "SQ_BUSY_CYCLES" in avg:
        out["busy_cycles"] = int(avg["SQ_BUSY_CYCLES"])
    # wave-cycle accounting (SQ_WAVE_CYCLES counts in units of four cycles per resident wave): what share of a wave's life it waited on anything /
    # on an instruction's operands, and the share it had a vector instruction executing
    if "SQ_WAVE_CYCLES" in avg:
        wc = avg["SQ_WAVE_CYCLES"] or 1
        for c in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c in avg:
                out[c.lower() + "_share"] = round(avg[c] / wc, 3)
    print(json.dumps({k: out}))
PY
rm -rf /tmp/sqc1 /tmp/sqc2

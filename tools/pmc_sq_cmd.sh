# SQ instruction mix of the kernels a python tool launches (development tool, on the GPU box, from the repository root):
#   bash tools/pmc_sq_cmd.sh <kernel name substring> tools/<script>.py [args ...]
# Two rocprofv3 --pmc passes (counters only, beside --kernel-trace), per kernel and template instantiation, averaged per launch.
set -e
WANT=$1; shift
R=$(pwd)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/sqd1 -- python3 $R/"$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/sqd2 -- python3 $R/"$@" > /dev/null 2>&1
python3 - "$WANT" <<'PY'
import collections, csv, glob, json, sys
want = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/sqd1", "/tmp/sqd2"):
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if want in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:150] + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    w = avg.get("SQ_WAVES", 0) or 1
    out = {"launches": len(cs.get("SQ_WAVES", [])), "waves": int(w)}
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS"):
        if c in avg: out[c.replace("SQ_INSTS", "per_wave").lower()] = round(avg[c] / w, 1)
    if "SQ_BUSY_CYCLES" in avg: out["busy_cycles"] = int(avg["SQ_BUSY_CYCLES"])
    wc = avg.get("SQ_WAVE_CYCLES", 0) or 1
    for c in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM"):
        if c in avg: out[c.lower() + "_share"] = round(avg[c] / wc, 3)
    print(json.dumps({k: out}))
PY
rm -rf /tmp/sqd1 /tmp/sqd2

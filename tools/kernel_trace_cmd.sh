# per-kernel, per-grid launch durations of what a python tool launches (development tool, on the GPU box, from the repository root):
#   bash tools/kernel_trace_cmd.sh <kernel name substring> tools/<script>.py [args ...]
set -e
WANT=$1; shift
R=$(pwd)
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ktr
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktr -- python3 $R/"$@" > /dev/null 2>&1
python3 - "$WANT" <<'PY'
import collections, csv, glob, json, re, sys
acc = collections.defaultdict(list)
for path in glob.glob("/tmp/ktr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if sys.argv[1] in n:
            key = (re.findall(r"(\w+_kernel)", n) or [n[:40]])[0] + (("<" + n.split("_kernel<", 1)[1].split(">")[0] + ">") if "_kernel<" in n else "") + " lds=" + r.get("LDS_Block_Size", "?") + " grid=" + "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y")) + " wg=" + r.get("Workgroup_Size_X", "?")
            acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v.sort()
    print(json.dumps({"kernel": k, "launches": len(v), "median_us": round(v[len(v) // 2], 1), "min_us": round(v[0], 1)}))
PY
rm -rf /tmp/ktr

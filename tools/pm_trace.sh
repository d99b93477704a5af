# per-launch durations of the PatchMatch kernels at 1080p RGB (development tool): bash tools/pm_trace.sh <patchmatch_run_batches 0|1>
set -e
R=$(pwd)
cat > /tmp/pm_one.py <<PY
import os, sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import numpy as np, torch, libstevi_amd as sv
MF = sv.matchingFunctions; dev = torch.device("cuda:0")
W, H, C, radius, search, n_iter = 1920, 1080, 3, 3, (-10, 10, -20, 20), 10
rng = np.random.default_rng(5)
base = rng.uniform(0, 1, (H + 64, W + 64, C)).astype(np.float32)
base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) / np.float32(4)
src = np.ascontiguousarray(base[32:32 + H, 32:32 + W]); dy, dx = (int(v) for v in os.environ.get("PM_SHIFT", "3,7").split(",")); tgt = np.ascontiguousarray(base[32 - dy:32 - dy + H, 32 + dx:32 + dx + W])
d_src, d_tgt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
sv.set_test_option(d_src, "patchmatch_run_batches", int(sys.argv[1]))
sv.cachelessPatchMatch(MF.ZNCC, d_src, d_tgt, radius, sv.searchOffset2(*search), n_iter, 4, seed=11)
torch.cuda.synchronize()
PY
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmtr
rocprofv3 --kernel-trace --output-format csv -d /tmp/pmtr -- python3 /tmp/pm_one.py $1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for path in glob.glob("/tmp/pmtr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "pm_" in n:
            rows.append((int(r["Start_Timestamp"]), n.split("(")[0].split("::")[-1][:28], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
print(" ".join(f"{n}:{us:.0f}" for _, n, us in rows))
PY

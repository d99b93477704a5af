# Same-box A/B of two builds of libstevi_hip.so under rocprofv3 (a development tool, on the GPU box, from the repository root):
#   bash tools/ab_two_builds.sh _ab/libA.so _ab/libB.so [sweep_ab.py arms ...]
# The two libraries are put in place of libstevi_amd/libstevi_hip.so in turn (A B A B), tools/sweep_ab.py runs the census + SGM step
# under `rocprofv3 --kernel-trace --stats`, and the average duration of every kernel of the step is printed per run together with the
# disparity checksum.  (Event brackets differ by 2 - 3 us between boxes; two builds on one box, alternated, differ by what changed.)
set -e
A=$1; B=$2; shift 2
ARMS=${@:-census_tiles=1}
R=$(pwd)
# the installed library is put back on EVERY way out (a failing step under set -e, a signal), not only after the last arm
KEEP=$(mktemp /tmp/stevi_keep_XXXXXX.so)
cp "$R/libstevi_amd/libstevi_hip.so" "$KEEP"
trap 'cp "$KEEP" "$R/libstevi_amd/libstevi_hip.so"; rm -f "$KEEP"' EXIT
cd /tmp; export TMPDIR=/tmp
for arm in A B A B; do
  if [ $arm = A ]; then cp $R/$A $R/libstevi_amd/libstevi_hip.so; else cp $R/$B $R/libstevi_amd/libstevi_hip.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$arm -- python3 $R/tools/sweep_ab.py --rounds 1 --reps 100 $ARMS > /tmp/ab_$arm.log 2>&1
  f=$(ls /tmp/abl_$arm/*/*kernel_stats.csv | head -1)
  echo "build $arm: $(python3 - "$f" <<'PY'
import csv, re, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+_kernel)", r["Name"])
    if m and "svh" in r["Name"] and int(r["Calls"]) >= 50:
        out.append("%s %.2f us" % (m.group(1), float(r["AverageNs"]) / 1e3))
print("; ".join(out))
PY
) $(grep -o 'checksum": [0-9]*' /tmp/ab_$arm.log | head -1)"
  rm -rf /tmp/abl_$arm
done

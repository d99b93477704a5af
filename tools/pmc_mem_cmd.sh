# Memory-side counters of the kernels a python tool launches (development tool, on the GPU box, from the repository root):
#   bash tools/pmc_mem_cmd.sh <kernel name substring> tools/<script>.py [args ...]
# Separate rocprofv3 --pmc passes (counters only, beside --kernel-trace), averaged per launch and kernel.
set -e
WANT=$1; shift
R=$(pwd)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/md1 -- python3 $R/"$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/md2 -- python3 $R/"$@" > /dev/null 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum --kernel-trace --output-format csv -d /tmp/md3 -- python3 $R/"$@" > /dev/null 2>&1 || true
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d /tmp/md4 -- python3 $R/"$@" > /dev/null 2>&1 || true
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d /tmp/md5 -- python3 $R/"$@" > /dev/null 2>&1 || true
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU TA_BUSY_avr TA_TA_BUSY_sum --kernel-trace --output-format csv -d /tmp/md6 -- python3 $R/"$@" > /dev/null 2>&1 || true
python3 - "$WANT" <<'PY'
import collections, csv, glob, json, sys
want = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/md1", "/tmp/md2", "/tmp/md3", "/tmp/md4", "/tmp/md5", "/tmp/md6"):
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if want in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(json.dumps({k: {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())} | {"launches": max(len(v) for v in cs.values())}}))
PY
rm -rf /tmp/md1 /tmp/md2 /tmp/md3 /tmp/md4 /tmp/md5 /tmp/md6

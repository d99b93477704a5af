set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2b -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r2b_line_rocprof.json 2> $R/gpurun_out/r2b_rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-api-chain > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-api-chain > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq1 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-api-chain > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq2 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-api-chain > /dev/null 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r2b_traffic.json
python3 tools/pmc_sq_summary.py gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 > gpurun_out/r2b_sq_counters.json
cp $(ls gpurun_out/prof_r2b/*/*kernel_stats.csv | head -1) gpurun_out/r2b_kernel_stats.csv
rm -rf gpurun_out/prof_r2b gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
python3 bench.py --steps 200 --warmup 10 > gpurun_out/r2b_line.json 2> gpurun_out/r2b_line.err
python3 bench.py --steps 200 --warmup 10 --two-frames --no-cpu-baseline --no-api-chain > gpurun_out/r2b_line_two_frames.json 2>/dev/null
head -5 gpurun_out/r2b_kernel_stats.csv
python3 -c "
import json; d=json.load(open('gpurun_out/r2b_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac'], d['end_px_err'], d['cpu_baseline']['value'])"

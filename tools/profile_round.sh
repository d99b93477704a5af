# The profile set of a round (on the MI355X box, from the repository root): bash tools/profile_round.sh r03
#   gpurun_out/<tag>_kernel_stats.csv        rocprofv3 --kernel-trace --stats of the default bench command
#   gpurun_out/<tag>_line_rocprof.json       the bench line printed under the profiler
#   gpurun_out/<tag>_traffic.json            HBM bytes per launch and kernel (two --pmc passes, tools/pmc_summary.py)
#   gpurun_out/<tag>_sq_counters.json        SQ instruction mix and wait breakdown (three --pmc passes, tools/pmc_sq_summary.py)
#   gpurun_out/<tag>_line.json, _line_driver_flags.json, _line_two_frames.json   plain bench lines (200 steps; the driver's --steps 20 --warmup 5)
#   gpurun_out/<tag>_c4_kernel_stats.csv     rocprofv3 --kernel-trace --stats of tools/bench_configs.py c4
#   gpurun_out/<tag>_other_workloads.jsonl   C1 - C5 and the per-function chains
set -e
TAG=${1:-r03}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- $B --steps 100 --warmup 10 --no-cpu-baseline --no-configs --no-host-chain > $R/gpurun_out/${TAG}_line_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_c4 -- python3 $R/tools/bench_configs.py c4 > $R/gpurun_out/${TAG}_c4_line_rocprof.json 2>> $R/gpurun_out/${TAG}_rocprof.err
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- $B --steps 5 --warmup 2 --no-cpu-baseline --no-api-chain --no-configs --no-host-chain > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- $B --steps 5 --warmup 2 --no-cpu-baseline --no-api-chain --no-configs --no-host-chain > /dev/null 2>&1
echo "traffic passes done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq1 -- $B --steps 10 --warmup 2 --no-cpu-baseline --no-api-chain --no-configs --no-host-chain > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq2 -- $B --steps 10 --warmup 2 --no-cpu-baseline --no-api-chain --no-configs --no-host-chain > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq3 -- $B --steps 10 --warmup 2 --no-cpu-baseline --no-api-chain --no-configs --no-host-chain > /dev/null 2>&1 || true
echo "sq passes done"
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/${TAG}_traffic.json > /dev/null
python3 tools/pmc_sq_summary.py gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq3 > gpurun_out/${TAG}_sq_counters.json
cp $(ls gpurun_out/prof_$TAG/*/*kernel_stats.csv | head -1) gpurun_out/${TAG}_kernel_stats.csv
cp $(ls gpurun_out/prof_${TAG}_c4/*/*kernel_stats.csv | head -1) gpurun_out/${TAG}_c4_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG gpurun_out/prof_${TAG}_c4 gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq3
python3 bench.py --steps 200 --warmup 10 > gpurun_out/${TAG}_line.json 2> gpurun_out/${TAG}_line.err
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_line_driver_flags.json 2>> gpurun_out/${TAG}_line.err
python3 bench.py --steps 200 --warmup 10 --two-frames --no-cpu-baseline --no-api-chain --no-configs --no-host-chain > gpurun_out/${TAG}_line_two_frames.json 2>/dev/null
echo "bench lines done"
(python3 tools/bench_configs.py c1 c2 c3 c4 c5slice c5 sad_sgm sad_sgm_d128 sad_sgm_d64 ncc_argmax sad_argmin; python3 tools/bench_api_chain.py; python3 tools/bench_api_chain.py 128 64; SVH_CHAIN_PLAIN=1 python3 tools/bench_api_chain.py 256 128 64; python3 tools/bench_l2r.py; python3 tools/sgm_score_probe.py; python3 tools/bench_2d.py; python3 tools/bench_hierarchical.py; python3 tools/bench_reference_rows.py; python3 tools/bench_census_windows.py; python3 tools/bench_census_colour.py) 2>/dev/null | grep "^{" > gpurun_out/${TAG}_other_workloads.jsonl || true
head -8 gpurun_out/${TAG}_kernel_stats.csv
python3 -c "
import json; d=json.load(open('gpurun_out/${TAG}_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac'], d['roofline'].get('frac_minus_half_pair'), d['end_px_err'], d['cpu_baseline']['value'], d['winner_identity']['ms_per_step'], d['api_chain']['ms'])"

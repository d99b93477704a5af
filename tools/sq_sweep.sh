# SQ counter passes over the census + SGM step (tools/sweep_ab.py, one arm): instruction mix, wait breakdown.  On the GPU box:
#   bash tools/sq_sweep.sh <tag> [arm]      -> gpurun_out/<tag>_sq_counters.json
set -e
TAG=${1:-sq}
ARM=${2:-census_sweep_rl=1}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/sweep_ab.py --reps 3 $ARM"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq1 -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq2 -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq3 -- $P > /dev/null 2>&1 || true
cd $R
python3 tools/pmc_sq_summary.py gpurun_out/${TAG}_sq1 gpurun_out/${TAG}_sq2 gpurun_out/${TAG}_sq3 > gpurun_out/${TAG}_sq_counters.json
rm -rf gpurun_out/${TAG}_sq1 gpurun_out/${TAG}_sq2 gpurun_out/${TAG}_sq3

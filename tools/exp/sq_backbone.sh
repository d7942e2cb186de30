#!/bin/bash
# SQ counter pass over the ResNet-18 inference forward (tools/exp/backbone_fwd.py) -> gpurun_out/prof_$1/sq_summary.txt
set -e
TAG=${1:-bbsq}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/tools/exp/backbone_fwd.py > $OUT/run.log 2>&1
cd $ROOT
SD_SQ_TOP=5 python3 tools/summarize_sq.py $OUT > $OUT/sq_summary.txt 2>&1 || true
find $OUT -name "*_kernel_trace.csv" -size +20M -delete || true
find $OUT -name "*counter_collection.csv" -size +30M -delete || true
cat $OUT/sq_summary.txt

#!/bin/bash
# Runs on the GPU box (via gpurun): one SQ counter pass over a few training steps (bench.py --mode train).
# Usage: tools/profile_sq_train.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/sq_summary.txt
set -e
TAG=${1:-sqtrain}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_sq.log 2>&1
cd $ROOT
python3 tools/summarize_sq.py $OUT 14 > $OUT/sq_summary.txt 2>&1 || true
find $OUT -name "*_kernel_trace.csv" -size +20M -delete || true
find $OUT -name "*counter_collection.csv" -size +30M -delete || true
grep -E "dispatches|MFMA pipe|of a wave|LDS bank" $OUT/sq_summary.txt

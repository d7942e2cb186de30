#!/bin/bash
# Runs on the GPU box (via gpurun): FETCH_SIZE / WRITE_SIZE passes over a few training steps (bench.py --mode train).
# Usage: tools/profile_train_traffic.sh <tag>     -> gpurun_out/prof_<tag>/summary.txt
set -e
TAG=${1:-r03train}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --mode train --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing "$@" > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --mode train --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > $OUT/bench_write.log 2>&1
cd $ROOT
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1 || true
find $OUT -name "*_kernel_trace.csv" -size +20M -delete || true
find $OUT -name "*counter_collection.csv" -size +20M -delete || true
tail -45 $OUT/summary.txt

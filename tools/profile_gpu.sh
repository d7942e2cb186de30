#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + two PMC passes for HBM traffic.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -e
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extras "$@" > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-extras "$@" > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-extras "$@" > $OUT/bench_write.log 2>&1
cd $ROOT
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1 || true
# keep only the small files (gpurun_out merge is capped)
find $OUT -name "*_kernel_trace.csv" -size +20M -delete || true
find $OUT -name "*counter_collection.csv" -size +20M -delete || true
tail -40 $OUT/summary.txt

#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel table of the image-conditioned training step (tools/bench_c5.py, BASELINE
# configs[4]) at a bounded batch, NCHW and channels_last.   Usage: tools/profile_c5.sh <tag> [batch]
set -e
TAG=${1:-r03c5}; BATCH=${2:-16}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nchw -- python3 $ROOT/tools/bench_c5.py --batch $BATCH --steps 2 > $OUT/nchw.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nhwc -- python3 $ROOT/tools/bench_c5.py --batch $BATCH --steps 2 --channels-last > $OUT/nhwc.log 2>&1
cd $ROOT
for v in nchw nhwc; do
  echo "== $v: $(grep -h '"workload"' $OUT/$v.log | tail -1 | cut -c1-600)"
  python3 - "$OUT/$v" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':100s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'pct':>6s}")
for r in rows[:14]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}")
print(f"all kernels: {tot/1e6:.1f} ms")
PY
done
find $OUT -name "*_kernel_trace.csv" -size +5M -delete || true

#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of the training step (tools/bench_train.py).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_train
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/bench_train.py --steps 10 --warmup 2 > $OUT/log.txt 2>&1
cd $ROOT
tail -1 $OUT/log.txt
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/trace/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
with open("$OUT/summary.txt","w") as fh:
    for r in rows[:24]:
        line = r["Name"][:100].ljust(100)+r["Calls"].rjust(7)+("%.2f"%(float(r["TotalDurationNs"])/1e6)).rjust(10)+("%.1f"%(float(r["AverageNs"])/1e3)).rjust(10)+r["Percentage"].rjust(8)
        print(line); fh.write(line+"\n")
PY
find $OUT -name "*_kernel_trace.csv" -size +20M -delete || true

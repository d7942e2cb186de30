#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of the training step (bench.py --mode train).
# Usage: tools/profile_train.sh [tag] [bench args...]   -> gpurun_out/prof_train_<tag>/summary.txt
set -e
TAG=${1:-r02}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_train_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --mode train --steps 10 --warmup 2 "$@" > $OUT/log.txt 2>&1
cd $ROOT
tail -1 $OUT/log.txt | cut -c1-300
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/trace/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
with open("$OUT/summary.txt","w") as fh:
    hdr="# rocprofv3 --kernel-trace --stats, bench.py --mode train --steps 10 --warmup 2 $@ (the 3 warm-up steps incl. the graph capture + 10 timed): name, calls, total ms, avg us, %%; all kernels %.2f ms" % (tot/1e6)
    print(hdr); fh.write(hdr+"\n")
    for r in rows[:28]:
        line = r["Name"][:100].ljust(100)+r["Calls"].rjust(7)+("%.2f"%(float(r["TotalDurationNs"])/1e6)).rjust(10)+("%.1f"%(float(r["AverageNs"])/1e3)).rjust(10)+r["Percentage"].rjust(8)
        print(line); fh.write(line+"\n")
PY
find $OUT -name "*_kernel_trace.csv" -size +20M -delete || true

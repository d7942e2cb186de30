#!/bin/bash
# Runs on the GPU box: one SQ counter pass with the dynamic instruction counts of the bench rollout's kernels.
# Usage: tools/profile_insts.sh <tag>     -> gpurun_out/prof_<tag>/insts_summary.txt
set -e
TAG=${1:-insts}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA \
  --kernel-trace --output-format csv -d $OUT/pmc -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-extras "$@" > $OUT/bench.log 2>&1
cd $ROOT
python3 - <<PY > $OUT/insts_summary.txt
import csv, glob, collections
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); grid = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU":
        n[k] += 1
        grid[k] = int(r["Grid_Size"]) // 64 if "Grid_Size" in r else 0
print("per-dispatch averages; per-wave = / (waves per dispatch)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:6]:
    d = n[k]
    print(k, "dispatches", d, "waves/dispatch", grid.get(k))
    for c, x in sorted(v.items()):
        per = x / d
        print(f"   {c:24s} {per:16.0f}" + (f"   per wave {per / grid[k]:10.1f}" if grid.get(k) else ""))
PY
find $OUT -name "*counter_collection.csv" -size +30M -delete || true
find $OUT -name "*_kernel_trace.csv" -size +20M -delete || true
cat $OUT/insts_summary.txt

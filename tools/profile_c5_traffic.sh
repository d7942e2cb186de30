#!/bin/bash
# Runs on the GPU box (via gpurun): FETCH_SIZE / WRITE_SIZE passes (separate, with --kernel-trace only) over the image-conditioned training step
# (tools/bench_c5.py, BASELINE configs[4] per-GPU share) -> gpurun_out/prof_c5_traffic/summary.txt: HBM bytes per launch of every kernel.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_c5_traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/bench_c5.py --batch 16 --steps 1 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/bench_c5.py --batch 16 --steps 1 > $OUT/write.log 2>&1
cd $ROOT
python3 - $OUT <<'PY' > $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
def per_kernel(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    s, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"].split("(")[0]
        s[k] += float(r["Counter_Value"]); n[k] += 1
    return s, n
fs, fn = per_kernel(out + "/pmc_fetch", "FETCH_SIZE")
ws, wn = per_kernel(out + "/pmc_write", "WRITE_SIZE")
print("# HBM traffic per launch of the image-conditioned training step's kernels (tools/bench_c5.py --batch 16; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in")
print("# separate passes; counter unit KiB; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md).  MB per launch, averaged over the launches.")
print(f"{'kernel':70s} {'launches':>8s} {'fetch MB':>10s} {'write MB':>10s} {'total MB':>10s}")
rows = []
for k in set(fs) | set(ws):
    f = 2 * fs[k] * 1024 / max(fn[k], 1) / 1e6
    w = ws[k] * 1024 / max(wn[k], 1) / 1e6
    rows.append((f * fn[k] + w * wn[k], k, max(fn[k], wn[k]), f, w))
for tot, k, n, f, w in sorted(rows, reverse=True)[:26]:
    print(f"{k[:70]:70s} {n:8d} {f:10.1f} {w:10.1f} {f + w:10.1f}")
PY
find $OUT -name "*_kernel_trace.csv" -delete || true
find $OUT -name "*counter_collection.csv" -size +20M -delete || true
cat $OUT/summary.txt

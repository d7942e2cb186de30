#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over the ResNet-18 inference forward (tools/exp/backbone_fwd.py) -> gpurun_out/prof_$1/traffic.txt
set -e
TAG=${1:-bbtraffic}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $ROOT/tools/exp/backbone_fwd.py > $OUT/run_$c.log 2>&1
done
cd $ROOT
python3 - <<PY > $OUT/traffic.txt
import csv, glob, collections
out = "$OUT"
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = (r["Kernel_Name"].split("(")[0][:40], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))
        res[k][c].append(float(r["Counter_Value"]))
print("per-dispatch averages, counter units as reported (FETCH_SIZE in 32-byte... see MI355X_MICROARCH.md: KiB after rocprofv3's derivation)")
for k, v in sorted(res.items()):
    fs, ws = v.get("FETCH_SIZE", [0]), v.get("WRITE_SIZE", [0])
    print(f"{k[0]:42s} grid {k[1]:>10s}  n={len(fs):3d}  FETCH_SIZE {sum(fs)/len(fs):14.0f}  WRITE_SIZE {sum(ws)/max(len(ws),1):14.0f}")
PY
find $OUT -name "*counter_collection.csv" -size +20M -delete || true
cat $OUT/traffic.txt

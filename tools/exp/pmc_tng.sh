#!/bin/bash
# On the GPU box: SQ counters of the grouped weight-gradient GEMM microbenchmark (tools/exp/tng_time.py).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_tng
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/tools/exp/tng_time.py > $OUT/log1.txt 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS \
  --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $ROOT/tools/exp/tng_time.py > $OUT/log2.txt 2>&1 || true
cd $ROOT
python3 tools/summarize_sq.py $OUT > $OUT/sq_summary.txt 2>&1 || true
python3 - <<PY >> $OUT/sq_summary.txt 2>&1 || true
import csv, glob, collections
f = glob.glob("$OUT/pmc_sq2/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(float); n = set()
for r in csv.DictReader(open(f[0])):
    if "gemm_tn16g" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
print("second pass, gemm_tn16g per dispatch:")
for k, v in sorted(agg.items()): print("  ", k, round(v / len(n)))
PY
find $OUT -name "*_kernel_trace.csv" -size +20M -delete || true
cat $OUT/sq_summary.txt | head -60

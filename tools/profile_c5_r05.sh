#!/bin/bash
# Runs on the GPU box (via gpurun): BASELINE configs[4] per-GPU share (tools/bench_c5.py, 16 trajectories x 10 frames of 480 x 640) - the step time of
# this repository's kernels and of the torch.nn / MIOpen route (SD_CONV=torch) on the SAME box, unprofiled, then the rocprofv3 kernel table of the
# former.  Writes gpurun_out/c5_r05.txt (copied to profiles/r05_c5_image_step_kernel_stats.txt).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_c5_r05
mkdir -p $OUT
cd $ROOT
python3 tools/bench_c5.py --batch 16 --steps 3 > $OUT/hip_plain.log 2>$OUT/hip_plain.err
SD_CONV=torch python3 tools/bench_c5.py --batch 16 --steps 3 > $OUT/torch_plain.log 2>$OUT/torch_plain.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hip -- python3 $ROOT/tools/bench_c5.py --batch 16 --steps 2 > $OUT/hip_prof.log 2>&1
cd $ROOT
{
  echo "# BASELINE configs[4] per-GPU share (tools/bench_c5.py: ResNet-18 + denoiser training step, B = 16 x 10 frames of 480 x 640), round 5, one box."
  echo "# Every convolution (forward, data gradient incl. the transposed stride-2 ones, weight gradient), every BatchNorm and the stem's max-pool run on"
  echo "# this repository's kernels (cv:: / cvt::); NO MIOpen kernel is left.  Unprofiled step times first, then rocprofv3 --kernel-trace --stats of"
  echo "# the hip route (warm-up + 2 timed steps + 3 backbone-only passes = 6 forward + backward passes of the backbone in the table)."
  echo "== hip route, unprofiled: $(grep -h '"workload"' $OUT/hip_plain.log | tail -1)"
  echo "== torch.nn / MIOpen route (SD_CONV=torch), unprofiled: $(grep -h '"workload"' $OUT/torch_plain.log | tail -1)"
  echo "== hip route under rocprofv3: $(grep -h '"workload"' $OUT/hip_prof.log | tail -1)"
  python3 - "$OUT/hip" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':100s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'pct':>6s}")
for r in rows[:40]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}")
print(f"all kernels: {tot/1e6:.1f} ms")
mi = sum(float(r["TotalDurationNs"]) for r in rows if "miopen" in r["Name"].lower() or "MIOpen" in r["Name"] or "igemm" in r["Name"].lower() or "naive_conv" in r["Name"])
print(f"MIOpen kernels: {mi/1e6:.2f} ms = {100*mi/tot:.2f} % of all kernel time")
PY
} > $ROOT/gpurun_out/c5_r05.txt
find $OUT -name "*_kernel_trace.csv" -delete || true
tail -50 $ROOT/gpurun_out/c5_r05.txt

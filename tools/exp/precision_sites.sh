#!/bin/bash
# Precision headroom per GEMM site: builds tools/exp/traj_layer.hip with ONE of the three fp16 products dropped at
# ONE site and prints the one-step error of eps / x against the fp64 host restatement (L = 4, B = 16).
#   build half (no GPU):  tools/exp/precision_sites.sh build      run half (GPU box):  tools/exp/precision_sites.sh run
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SITES="QKV:1 SCORES:2 PV:4 OUT:8 XSC:16 XPV:32 W1:64 W2:128 EMB:256 FC:512"
OUTD=$ROOT/tools/exp/prec_bin
mkdir -p $OUTD
if [ "$1" = build ]; then
  for s in $SITES; do n=${s%%:*}; m=${s##*:}
    for which in ALO BLO; do
      hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -DTJ_DROP_$which=$m -I $ROOT/include -I $ROOT/soccerdiffusion_amd/csrc $ROOT/tools/exp/traj_layer.hip -o $OUTD/p_${n}_$which &
    done; wait
  done
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -DTJ_DROP_ALO=1023 -I $ROOT/include -I $ROOT/soccerdiffusion_amd/csrc $ROOT/tools/exp/traj_layer.hip -o $OUTD/p_ALL_ALO
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -DTJ_DROP_BLO=1023 -I $ROOT/include -I $ROOT/soccerdiffusion_amd/csrc $ROOT/tools/exp/traj_layer.hip -o $OUTD/p_ALL_BLO
  exit 0
fi
echo "site      dropped product                      one-step rel L2 error of eps / of x vs fp64   (three products everywhere: ~8e-7 / 9e-8)"
for f in $OUTD/p_*; do
  b=$(basename $f); n=${b#p_}; which=${n##*_}; n=${n%_*}
  r=$(timeout -k 5 60 $f 16 1 2 4 2>&1 | grep "\[B\] traj_step_kernel (L" | sed 's/.*eps \([0-9.e+-]*\), x \([0-9.e+-]*\).*/\1  \2/')
  t=$(timeout -k 5 60 $f 4096 5 0 4 2>&1 | grep "\[B\] traj_step_kernel: B" | sed 's/.*L=4  \([0-9.]*\) us.*/\1/')
  printf "%-9s %-36s %s   (%s us per step at B = 4096)\n" $n "$([ $which = ALO ] && echo 'A-lo x B-hi (weights / K / V / G lo)' || echo 'A-hi x B-lo (activation / Q / P lo)')" "$r" "$t"
done

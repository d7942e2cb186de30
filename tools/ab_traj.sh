#!/bin/bash
# A/B of the trajectory step kernel in the standalone harness (tools/exp/traj_layer.hip): builds the mode-3 (TL_PRECISE=1) and mode-4 variants with
# extra -D flags and prints their step times.  Usage (on the GPU box): tools/ab_traj.sh "<extra flags A>" "<extra flags B>" ...
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
i=0
for flags in "$@"; do
  for prec in 1 0; do
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -Xclang -target-feature -Xclang -packed-fp32-ops -I include -I soccerdiffusion_amd/csrc \
      -DTL_PRECISE=$prec $flags tools/exp/traj_layer.hip -o /tmp/tl_${i}_$prec
    echo "== variant $i [$flags] precise=$prec: $(/tmp/tl_${i}_$prec 4096 20 1 4 | grep 'traj_step_kernel' | tail -2 | tr '\n' ' ')"
  done
  i=$((i+1))
done

#!/bin/bash
# On the GPU box: bench every variant built by tools/ab_build.sh (2 rollouts each), print value + layer-kernel time.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for lib in soccerdiffusion_amd/lib/variants/lib_*.so; do
  SD_HIP_LIB=$PWD/$lib python bench.py --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib'.split('/')[-1], 'value', d['value'], 'layer_ms', r['avg_launch_ms'], 'share', r['kernel_time_share'])"
done

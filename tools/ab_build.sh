#!/bin/bash
# Builds variants of the HIP library for A/B runs on the GPU box:
#   [AB_TU=sd_train] tools/ab_build.sh name1 "-DFOO=1" name2 "-DFOO=2" ...
# -> soccerdiffusion_amd/lib/variants/lib_<name>.so   (run with SD_HIP_LIB=<path> python bench.py ...)
# Only the translation unit AB_TU (default sd_kernels) is recompiled with the flags; the other objects come from the
# regular build (python -m soccerdiffusion_amd.build), which must be current.
set -e
cd "$(dirname "$0")/.."
TU=${AB_TU:-sd_kernels}
mkdir -p soccerdiffusion_amd/lib/variants
OTHERS=""
for o in sd_kernels sd_train sd_train_chain; do
  [ "$o" != "$TU" ] && OTHERS="$OTHERS soccerdiffusion_amd/lib/$o.o"
done
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-value -I include $flags -c soccerdiffusion_amd/csrc/$TU.hip \
      -o soccerdiffusion_amd/lib/variants/${TU}_$name.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC soccerdiffusion_amd/lib/variants/${TU}_$name.o $OTHERS \
      -o soccerdiffusion_amd/lib/variants/lib_$name.so && rm soccerdiffusion_amd/lib/variants/${TU}_$name.o ) &
done
wait
ls -la soccerdiffusion_amd/lib/variants/

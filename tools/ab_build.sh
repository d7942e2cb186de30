#!/bin/bash
# Builds variants of the HIP library for A/B runs on the GPU box:
#   tools/ab_build.sh name1 "-DFOO=1" name2 "-DFOO=2" ...
# -> soccerdiffusion_amd/lib/variants/lib_<name>.so   (run with SD_HIP_LIB=<path> python bench.py ...)
set -e
cd "$(dirname "$0")/.."
mkdir -p soccerdiffusion_amd/lib/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -Wno-unused-value -I include $flags \
    soccerdiffusion_amd/csrc/sd_kernels.hip soccerdiffusion_amd/csrc/sd_train.hip \
    -o soccerdiffusion_amd/lib/variants/lib_$name.so &
done
wait
ls -la soccerdiffusion_amd/lib/variants/

#!/bin/bash
# Builds A/B variants of libptamd.so (kernel knobs of csrc/pt_params.h) into csrc/variants/.
set -e
cd "$(dirname "$0")/../cudapathtracer_amd/csrc"
mkdir -p variants
build() { name=$1; shift; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function "$@" -shared -o variants/lib_$name.so pt_kernels.hip pt_wavefront.hip pt_api.hip novum_host.cpp & }
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  build $name $flags
done
wait
ls -la variants

#!/bin/bash
# Builds A/B variants of libptamd.so (kernel knobs of csrc/pt_params.h) into csrc/variants/.
# usage: tools/build_variants.sh name[:"-DPT_X=1 -DPT_Y=2"] ...
set -e
cd "$(dirname "$0")/../cudapathtracer_amd/csrc"
mkdir -p variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  make -s -j4 OUT=variants/lib_$name.so OBJDIR=variants/_obj_$name EXTRA="$flags" &
done
wait
ls -la variants/*.so

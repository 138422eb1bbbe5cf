#!/bin/bash
# Interleaved A/B of library variants: tools/ab_lib.sh "default nocache ..." workload spp [rounds]
# ("default" = csrc/libptamd.so, anything else = csrc/variants/lib_<name>.so)
names=$1; wl=$2; spp=$3; rounds=${4:-2}
root=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq $rounds); do for n in $names; do
  if [ "$n" = default ]; then lib=$root/cudapathtracer_amd/csrc/libptamd.so; else lib=$root/cudapathtracer_amd/csrc/variants/lib_$n.so; fi
  out=$(PT_LIB_PATH=$lib python bench.py --workload $wl --spp $spp --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$wl $n $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], "ms/step", d["ms_per_step"])')"
done; done

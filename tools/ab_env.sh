#!/bin/bash
# Interleaved A/B of one environment knob: tools/ab_env.sh VAR "v1 v2" workload spp [rounds] [extra bench args]
var=$1; vals=$2; wl=$3; spp=$4; rounds=${5:-2}; shift 5 || shift $#
for r in $(seq $rounds); do for v in $vals; do
  out=$(env $var=$v python bench.py --workload $wl --spp $spp --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1)
  echo "$wl $var=$v $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], "ms/step", d["ms_per_step"])')"
done; done

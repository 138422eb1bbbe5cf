#!/bin/bash
# Interleaved A/B of one pt_set_option: tools/ab_opt.sh NAME "v1 v2" workload spp [rounds] [extra bench args]
var=$1; vals=$2; wl=$3; spp=$4; rounds=${5:-2}; shift 5 || shift $#
for r in $(seq $rounds); do for v in $vals; do
  out=$(python bench.py --workload $wl --spp $spp --steps 2 --warmup 1 --no-cpu-baseline --opt $var=$v "$@" 2>/dev/null | tail -1)
  echo "$wl $var=$v $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], "ms/step", d["ms_per_step"], "kernel ms", d["roofline"]["kernel_ms"], "sha", d.get("frame_sha", "")[:12])')"
done; done

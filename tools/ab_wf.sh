#!/bin/bash
# A/B of wavefront-variant builds: tools/ab_wf.sh <workload> <spp> name...
wl=$1; spp=$2; shift 2
for n in "$@"; do
  PT_LIB_PATH=$PWD/cudapathtracer_amd/csrc/variants/lib_$n.so timeout -k 10 300 python bench.py --workload $wl --variant wavefront --spp $spp --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$n', round(j['value']), 'Mray/s', round(j['roofline']['kernel_ms'],1), 'ms')"
done

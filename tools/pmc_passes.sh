#!/bin/bash
# PMC passes over one bench launch (counters in their own runs, no tracing besides --kernel-trace):
#   tools/pmc_passes.sh outdir workload spp "CTR_A CTR_B ..." ["CTR_C ..." ...]     (PT_BENCH_EXTRA="--opt k=v ..." is passed on to bench.py)
# Writes outdir/pass<i>/..._counter_collection.csv; summarise with tools/pmc_sum.py.
out=$1; wl=$2; spp=$3; shift 3
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$out"; out=$(cd "$out" && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 "$root/bench.py" --workload "$wl" --spp "$spp" --steps 1 --warmup 0 --no-cpu-baseline --no-secondary $PT_BENCH_EXTRA > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/pass$i.log"; exit 1; }
done

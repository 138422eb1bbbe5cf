#!/bin/bash
# Memory-pipeline counters of the kernel for scenes in HBM (run on the GPU box): tools/mem_pmc.sh OUTDIR [workload] [spp]
# One rocprofv3 --pmc pass per counter group (with --kernel-trace only; a group with a counter this GPU lacks is skipped);
# summary with tools/pmc_sum.py.
out=${1:-gpurun_out/mem_pmc}; wl=${2:-atrium262k_1920x1080_4096spp_depth16_mis}; spp=${3:-8}
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $out; out=$(cd $out && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
while read -r ctrs; do
  [ -z "$ctrs" ] && continue
  i=$((i + 1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 "$root/bench.py" --workload "$wl" --spp "$spp" --steps 1 --warmup 0 --no-cpu-baseline --no-secondary $PT_BENCH_EXTRA > "$out/pass$i.log" 2>&1 \
    || { echo "pass $i ($ctrs) failed:"; tail -3 "$out/pass$i.log"; }
  echo "pass $i done"
done <<'GROUPS'
TA_TA_BUSY_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum
TCC_REQ_sum TCC_EA0_RDREQ_sum
SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY
GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES
GROUPS
cd $root && python tools/pmc_sum.py $out | tail -60

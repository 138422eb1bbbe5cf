#!/bin/bash
# Issue / wait / LDS counters of the headline kernel (run on the GPU box): tools/sq_pmc.sh OUTDIR [workload] [spp]; one --pmc pass per four counters.
out=$1; wl=${2:-cornell_1920x1080_1024spp_depth8_mis}; spp=${3:-128}
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $out; out=$(cd $out && pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 5 60 rocprofv3 --list-avail > $out/avail.txt 2>&1
i=0
while read -r ctrs; do
  [ -z "$ctrs" ] && continue
  i=$((i + 1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 "$root/bench.py" --workload "$wl" --spp "$spp" --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > "$out/pass$i.log" 2>&1 \
    || { echo "pass $i ($ctrs) failed:"; grep -i "error\|exceed\|not" "$out/pass$i.log" | head -3; }
  echo "pass $i done"
done <<'GROUPS'
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC
SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY
SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS
SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES
SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_IFETCH_LEVEL SQ_ACTIVE_INST_FLAT
GROUPS
cd $root && python tools/pmc_sum.py $out megakernel_flat2 | tail -50

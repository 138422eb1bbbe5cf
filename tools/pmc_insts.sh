#!/bin/bash
# instruction-class counters of the timed kernels (one more PMC round: counters only with --kernel-trace)
out=gpurun_out/r03_insts; mkdir -p $out
[ -x tools/issue_rate/issue_rate ] || hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/issue_rate/issue_rate.hip -o tools/issue_rate/issue_rate
timeout -k 10 120 tools/issue_rate/issue_rate 20000 > $out/issue_rate.log 2>&1
for spec in "cornell_1920x1080_1024spp_depth8_mis 256 cornell" "blob82k_1920x1080_1024spp_depth8_mis 64 blob" "atrium262k_1920x1080_4096spp_depth16_mis 16 atrium" "blob82k_glass_1920x1080_1024spp_depth8_mis 32 glass"; do
  set -- $spec
  bash tools/pmc_passes.sh $out/pmc_$3 $1 $2 "SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SENDMSG" "GRBM_GUI_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VSKIPPED" > $out/$3.log 2>&1 || { echo "$3 failed"; tail -5 $out/$3.log; exit 1; }
  echo "$3 done"
done

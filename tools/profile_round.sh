#!/bin/bash
# The measurements one round commits under profiles/ (run on the GPU box): tools/profile_round.sh TAG [fullspp]
#   1. PMC passes (tools/pmc_passes.sh: counters in their own runs, --kernel-trace only) for the headline workload and the
#      two secondary workloads -> entries of profiles/roofline_inputs.json (tools/roofline.py collect)
#   2. rocprofv3 --kernel-trace --stats of bench.py -> per-kernel average durations
#   3. bench.py itself (after the inputs exist, so that its roofline block is complete) -> the bench line
set -e
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
SQ="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU"
run_pmc() {  # workload spp name [more passes: one quoted counter list each]
  wl=$1; sp=$2; nm=$3; shift 3
  bash tools/pmc_passes.sh $out/pmc_$nm $wl $sp "$SQ" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM" "$@"
  set -- $wl $sp $nm
  ms=$(python - <<PY
import csv,glob
rows=[r for f in glob.glob("$out/pmc_$3/pass2/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
import re
t=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in rows if "megakernel" in r["Kernel_Name"] and not re.search(r"megakernel(_hbm)?<\d, ?true", r["Kernel_Name"])]
print("%.3f" % t[-1])
PY
)
  python tools/roofline.py collect $out/pmc_$3 $1 $2 --kernel-ms $ms > $out/entry_$3.json
  python tools/roofline.py add $out/entry_$3.json
  echo "pmc $3 done (kernel under the profiler $ms ms)"
}
run_pmc cornell_1920x1080_1024spp_depth8_mis 1024 cornell
run_pmc atrium262k_1920x1080_4096spp_depth16_mis 32 atrium_spp32 "TA_TA_BUSY_sum" "TCP_GATE_EN1_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"      # scenes in HBM: how busy the texture addresser is, and the scratch share of its instructions
run_pmc blob82k_1920x1080_1024spp_depth8_mis 1024 blob "TA_TA_BUSY_sum" "TCP_GATE_EN1_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run_pmc cornell_mixed_1920x1080_1024spp_depth8_mis 1024 cornell_mixed                                       # the general bounce, LDS-resident (VALU roofline)
run_pmc blob82k_glass_1920x1080_1024spp_depth8_mis 128 blob_glass_spp128 "TA_TA_BUSY_sum" "TCP_GATE_EN1_sum"  # the general bounce, scene in HBM
cp profiles/roofline_inputs.json $out/roofline_inputs.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/stats -- python3 $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$out/stats.log 2>&1 )
# ... and of the headline alone (no secondaries, no share projections: those launch the headline's kernel on 1/2, 1/4, 1/8 of the tiles
# and would pull its AVERAGE away from a frame's duration): this is the summary whose average must agree with bench.py's kernel_ms
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/stats_headline -- python3 $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OLDPWD/$out/stats_headline.log 2>&1 )
python bench.py --steps 5 --warmup 2 > $out/bench.json 2> $out/bench.err
python tools/roofline.py check $out/bench.json
tail -c 400 $out/bench.json
if [ "$2" = "fullspp" ]; then      # BASELINE configs[3] at its own 4096 spp on one GPU (~4 min: a counted and a timed frame)
  python bench.py --workload atrium262k_1920x1080_4096spp_depth16_mis --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_atrium262k_fullspp.json 2> $out/bench_atrium262k_fullspp.err
  tail -c 300 $out/bench_atrium262k_fullspp.json
fi

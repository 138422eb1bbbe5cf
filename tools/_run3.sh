set -e
for v in attr dblc dbls; do
  export PT_LIB_PATH=$PWD/cudapathtracer_amd/csrc/variants/lib_$v.so
  bash tools/pmc_passes.sh gpurun_out/r02_phase_$v cornell_1920x1080_1024spp_depth8_mis 128 "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAIT_INST_ANY"
  python tools/roofline.py collect gpurun_out/r02_phase_$v cornell_1920x1080_1024spp_depth8_mis 128 > gpurun_out/r02_phase_$v.json
  cat gpurun_out/r02_phase_$v.json
done

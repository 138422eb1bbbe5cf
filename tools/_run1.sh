set -e
python bench.py --steps 3 --warmup 1 > gpurun_out/r02_bench0.json 2> gpurun_out/r02_bench0.err
tail -c 600 gpurun_out/r02_bench0.json
bash tools/pmc_passes.sh gpurun_out/r02_pmc_cornell_v8 cornell_1920x1080_1024spp_depth8_mis 1024 "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"
python tools/roofline.py collect gpurun_out/r02_pmc_cornell_v8 cornell_1920x1080_1024spp_depth8_mis 1024 > gpurun_out/r02_entry_cornell_v8.json
cat gpurun_out/r02_entry_cornell_v8.json

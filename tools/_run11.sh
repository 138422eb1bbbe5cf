set -e
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r02_t5.log 2>&1 || { tail -30 gpurun_out/r02_t5.log; exit 1; }
tail -2 gpurun_out/r02_t5.log
bash tools/ab_opt.sh simple "0 1" cornell_1920x1080_1024spp_depth8_mis 256 2
bash tools/ab_opt.sh simple "0 1" atrium262k_1920x1080_4096spp_depth16_mis 32 2
bash tools/ab_opt.sh simple "0 1" blob82k_1920x1080_1024spp_depth8_mis 32 2

set -e
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "windows or loop_exits or kernels_agree or deep or fuzz or golden" > gpurun_out/r02_t4.log 2>&1 || { tail -30 gpurun_out/r02_t4.log; exit 1; }
tail -2 gpurun_out/r02_t4.log
for wl in atrium262k_1920x1080_4096spp_depth16_mis blob82k_1920x1080_1024spp_depth8_mis; do
  bash tools/ab_opt.sh spec "0 1 2" $wl 32 2
done

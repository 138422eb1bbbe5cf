set -e
for wl in atrium262k_1920x1080_4096spp_depth16_mis blob82k_1920x1080_1024spp_depth8_mis; do
  python tools/ab_bench.py --spp 32 --rounds 2 --workload $wl spec0 spec1 2>&1 | tail -3
done

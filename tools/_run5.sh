set -e
for wl in blob82k_1920x1080_1024spp_depth8_mis atrium262k_1920x1080_4096spp_depth16_mis; do
python tools/ab_bench.py --spp 32 --rounds 2 --workload $wl logic noslp rcp noslprcp 2>&1 | tail -5
done

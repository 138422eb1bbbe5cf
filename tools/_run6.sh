set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/ab_bench.py --spp 128 --rounds 2 logic noslprcp cur 2>&1 | tail -4
python tools/ab_bench.py --spp 32 --rounds 1 --workload atrium262k_1920x1080_4096spp_depth16_mis logic cur 2>&1 | tail -3
python tools/ab_bench.py --spp 32 --rounds 1 --workload blob82k_1920x1080_1024spp_depth8_mis logic cur 2>&1 | tail -3

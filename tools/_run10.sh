python tools/ab_bench.py --spp 256 --rounds 2 base o2 nopost 2>&1 | tail -5

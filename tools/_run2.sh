set -e
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or fresh or flat or edge or fuzz" 2>&1 | tail -3
python tools/ab_bench.py --spp 128 --rounds 2 attr noattr dblc dbls 2>&1 | tail -14

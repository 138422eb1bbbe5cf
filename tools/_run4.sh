set -e
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or fresh or flat or edge or fuzz or bsdf or windows" 2>&1 | tail -3
python tools/ab_bench.py --spp 128 --rounds 2 attr logic noslp rcp noslprcp 2>&1 | tail -8
for v in noslp rcp noslprcp; do PT_LIB_PATH=$PWD/cudapathtracer_amd/csrc/variants/lib_$v.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or fresh or flat or fuzz or windows" 2>&1 | tail -1; done

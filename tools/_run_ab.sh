set -o pipefail
timeout -k 5 600 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
A=atrium262k_1920x1080_4096spp_depth16_mis
for k in "0 0" "8 8"; do set -- $k
  echo "wavefront node/tri keep $1 $2: $(PT_NODE_KEEP=$1 PT_TRI_KEEP=$2 timeout -k 10 300 python bench.py --workload $A --variant wavefront --spp 32 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value']), 'Mray/s')")" || exit 1
done
timeout -k 5 500 bash tools/ab_lib.sh "default hbm5 hbm4" $A 32 2 || exit 1

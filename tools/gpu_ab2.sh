#!/bin/bash
TAG=$1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "pool or winograd or vgg19 or config1 or worker_end_to_end" > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/${TAG}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
for U in 0 1 0 1; do
  ST2_WINO_UNPOOL=$U timeout -k 10 200 python bench.py --no-cpu-baseline --no-worker-level --steps 30 --repeats 5 > gpurun_out/${TAG}_unpool$U.json 2> gpurun_out/${TAG}_unpool$U.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_unpool$U.json')); print('UNPOOL=$U: %.2f it/s' % d['value'], {k: v for k, v in d['kernel_ms_per_step'].items() if 'conv' in k or 'pool' in k})"
done

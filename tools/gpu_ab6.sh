#!/bin/bash
TAG=$1
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "bf16 and not one_lbfgs_step" > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/${TAG}_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 200 python bench.py --size 2048 --optimizer lbfgs --precision bf16 --no-cpu-baseline --no-worker-level --steps 20 --repeats 3 > gpurun_out/${TAG}_bf16.json 2> gpurun_out/${TAG}_bf16.err || exit 1
python -c "import json; d=json.load(open('gpurun_out/${TAG}_bf16.json')); print('bf16 2048: %.2f it/s' % d['value'], d['kernel_ms_per_step'])"

set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -m gpu -q -x > gpurun_out/r3d_tests.log 2>&1; rc=$?; tail -6 gpurun_out/r3d_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline --no-worker-level > gpurun_out/r3d_bench.json 2> gpurun_out/r3d_bench.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r3d_bench.json')); print('fp32 1024', round(d['value'],2), d['kernel_ms_per_step'])"
timeout -k 10 300 python bench.py --size 2048 --optimizer lbfgs --precision bf16 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline > gpurun_out/r3d_bench16.json 2> gpurun_out/r3d_bench16.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r3d_bench16.json')); print('bf16 2048', round(d['value'],2), d['kernel_ms_per_step'])"

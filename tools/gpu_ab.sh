#!/bin/bash
# A/B and side measurements of a development round (GPU box).  Usage: bash tools/gpu_ab.sh <tag>
TAG=$1
mkdir -p gpurun_out
B="python bench.py --size 2048 --optimizer lbfgs --precision bf16 --no-cpu-baseline --no-worker-level --steps 20 --repeats 3"
for K in 0 64 128; do
  ST2_CONV16_SB_MAXK=$K timeout -k 10 200 $B > gpurun_out/${TAG}_bf16_sb$K.json 2> gpurun_out/${TAG}_bf16_sb$K.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_bf16_sb$K.json')); print('SB_MAXK=$K: %.2f it/s' % d['value'], {k: v for k, v in d['kernel_ms_per_step'].items() if 'conv' in k})"
done
ST2_CONV16_SB_MAXK=128 timeout -k 10 300 python -m pytest tests/test_gpu_bf16.py -m gpu -q -p no:cacheprovider > gpurun_out/${TAG}_bf16_sb_tests.log 2>&1; tail -2 gpurun_out/${TAG}_bf16_sb_tests.log
for R in 0 1; do for P in fp32 bf16; do
  timeout -k 10 300 python tools/bench_tiled.py --size 8192 --grid 2x4 --solo-rank $R --steps 5 --warmup 2 --precision $P > gpurun_out/${TAG}_tiled_solo_rank${R}_$P.json 2> gpurun_out/${TAG}_tiled_solo.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_tiled_solo_rank${R}_$P.json')); print('solo rank $R $P: window', d['window'], '%.1f ms/step' % d['ms_per_step'])"
done; done
for D in engine phases; do
  timeout -k 10 200 python tools/bench_tiled.py --size 1024 --grid 1x1 --steps 30 --warmup 3 --driver $D > gpurun_out/${TAG}_tiled_1x1_$D.json 2> gpurun_out/${TAG}_tiled_1x1.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_tiled_1x1_$D.json')); print('1x1 1024 driver $D: %.1f it/s' % d['value'])"
done

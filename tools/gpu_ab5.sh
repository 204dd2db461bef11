#!/bin/bash
# conv1_1 forward (the only direct fp32 conv of the bf16 path): tile configuration sweep through the whole step
TAG=$1
mkdir -p gpurun_out
for CFG in 3 2 7 4; do
  ST2_CONV_CFG=$CFG timeout -k 10 200 python bench.py --size 2048 --optimizer lbfgs --precision bf16 --no-cpu-baseline --no-worker-level --steps 20 --repeats 3 > gpurun_out/${TAG}_cfg$CFG.json 2> gpurun_out/${TAG}_cfg$CFG.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_cfg$CFG.json')); print('ST2_CONV_CFG=$CFG: %.2f it/s' % d['value'], {k: v for k, v in d['kernel_ms_per_step'].items() if 'f32' in k})"
done

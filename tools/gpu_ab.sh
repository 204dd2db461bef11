#!/bin/bash
# A/B and side measurements of a development round (GPU box).  Usage: bash tools/gpu_ab.sh <tag>
TAG=$1
mkdir -p gpurun_out
for R in 0 1; do for P in fp32 bf16; do
  timeout -k 10 300 python tools/bench_tiled.py --size 8192 --grid 2x4 --solo-rank $R --steps 5 --warmup 2 --precision $P > gpurun_out/${TAG}_tiled_solo_rank${R}_$P.json 2> gpurun_out/${TAG}_tiled_solo.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_tiled_solo_rank${R}_$P.json')); print('solo rank $R $P: window', d['window'], '%.1f ms/step' % d['ms_per_step'])"
done; done
for D in engine phases; do
  timeout -k 10 200 python tools/bench_tiled.py --size 1024 --grid 1x1 --steps 30 --warmup 3 --driver $D > gpurun_out/${TAG}_tiled_1x1_$D.json 2> gpurun_out/${TAG}_tiled_1x1.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_tiled_1x1_$D.json')); print('1x1 1024 driver $D: %.1f it/s' % d['value'])"
done

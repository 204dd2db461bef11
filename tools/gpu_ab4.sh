#!/bin/bash
TAG=$1
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "tiled" > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/${TAG}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
for P in bf16 bf16-full; do
  timeout -k 10 300 python tools/bench_tiled.py --size 8192 --grid 2x4 --solo-rank 1 --steps 5 --warmup 2 --precision $P > gpurun_out/${TAG}_tiled_solo_rank1_$P.json 2> gpurun_out/${TAG}_tiled_solo.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_tiled_solo_rank1_$P.json')); print('solo rank 1 $P: window', d['window'], '%.1f ms/step' % d['ms_per_step'])"
done

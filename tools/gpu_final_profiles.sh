#!/bin/bash
# Final profile set of a round (GPU box): fp32 headline (stats + PMC passes), bf16 configs[2] (stats + per-layer + SQ counters), bench lines.
TAG=$1
ROOT=$(pwd)
timeout -k 10 500 bash tools/profile_round.sh $TAG || exit 1
TRACE_SIZE=2048 timeout -k 10 500 bash tools/profile_round.sh ${TAG}_bf16 --size 2048 --optimizer lbfgs --precision bf16 || exit 1
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_bf16/sq -o q -- python3 $ROOT/bench.py --size 2048 --optimizer lbfgs --precision bf16 --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-worker-level > $ROOT/gpurun_out/prof_${TAG}_bf16/bench_sq.json 2> $ROOT/gpurun_out/prof_${TAG}_bf16/sq.log || exit 1
cd $ROOT
Q=$(find gpurun_out/prof_${TAG}_bf16/sq -name '*counter_collection.csv' | head -1)
python3 tools/pmc_sq.py $Q > gpurun_out/prof_${TAG}_bf16/sq_counters.txt
rm -rf gpurun_out/prof_${TAG}_bf16/sq
timeout -k 10 400 python bench.py --steps 30 --warmup 5 > gpurun_out/${TAG}_bench_1024_adam.json 2> gpurun_out/${TAG}_bench_1024_adam.err || exit 1
timeout -k 10 300 python bench.py --size 2048 --optimizer lbfgs --precision bf16 --no-cpu-baseline --steps 20 > gpurun_out/${TAG}_bench_bf16_2048_lbfgs.json 2> gpurun_out/${TAG}_bench_bf16.err || exit 1
timeout -k 10 300 python bench.py --examples > gpurun_out/${TAG}_bench_examples.json 2> gpurun_out/${TAG}_bench_examples.err || exit 1
python - <<PY
import json
for n in ('bench_1024_adam', 'bench_bf16_2048_lbfgs', 'bench_examples'):
    d = json.load(open('gpurun_out/${TAG}_%s.json' % n))
    print(n, '%.2f it/s' % d['value'], 'frac %.3f' % d['roofline']['frac'], {k: round(v, 1) for k, v in d.get('worker_level', {}).items() if k.endswith('it_s')}, (d.get('parity') or {}).get('image_mse'))
PY

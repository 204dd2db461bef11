#!/bin/bash
# One GPU-box session of a development round: selected tests, the headline bench line, the profile summaries.
# Usage (from the repo root, through gpurun): bash tools/gpu_round.sh <tag> "<test files>" ["<-k expression>"]
TAG=$1; SEL=$2; KEXPR=$3
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest $SEL ${KEXPR:+-k "$KEXPR"} -m gpu -q --durations=15 -p no:cacheprovider > gpurun_out/${TAG}_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/${TAG}_tests.log
tail -3 gpurun_out/${TAG}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests were killed: no further GPU step"; exit 1; fi
timeout -k 10 400 python bench.py --steps 30 --warmup 5 > gpurun_out/${TAG}_bench_1024_adam.json 2> gpurun_out/${TAG}_bench_1024_adam.err || exit 1
python - <<PY
import json
d = json.load(open('gpurun_out/${TAG}_bench_1024_adam.json'))
print('bench: %.2f it/s, frac %.3f, worker_level %s' % (d['value'], d['roofline']['frac'], {k: round(v, 1) for k, v in d.get('worker_level', {}).items() if k.endswith('it_s')}))
print('parity', {k: v for k, v in (d.get('parity') or {}).items() if k not in ('note', 'against')})
print('kernel ms', d.get('kernel_ms_per_step'))
PY

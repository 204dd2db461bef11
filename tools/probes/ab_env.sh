#!/bin/bash
# A/B of one environment switch on the headline bench (GPU box): `tools/probes/ab_env.sh VAR=a VAR=b ...` alternates the settings
# twice and prints it/s + the kernel classes that matter.  BENCH_ARGS="--size 2048 --precision bf16" selects another workload.
for rep in 1 2; do
  for setting in "$@"; do
    echo "== $setting"
    env $setting python bench.py --steps 20 --repeats 3 --no-cpu-baseline --no-worker-level --no-extra-configs $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('%.2f it/s  %.4f ms/step  ' % (d['value'], d['ms_per_step']) + '  '.join('%s %.4f' % (n, k[n]) for n in sorted(k)))"
  done
done

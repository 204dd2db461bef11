#!/bin/bash
# Same-box A/B of the per-launch-kind epilogues of the bf16 conv kernel (ST2_CONV16_EPI=0: the general epilogue): configs[2], three blocks each.
ARGS="--size 2048 --optimizer lbfgs --precision bf16 --steps 10 --warmup 5 --repeats 3 --no-cpu-baseline --no-worker-level --no-extra-configs"
for e in 1 0 1 0; do
  ST2_CONV16_EPI=$e python3 bench.py $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('ST2_CONV16_EPI=$e: %.2f it/s  %.3f ms/step  conv frac %.4f  fwd %.3f ms dgrad %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['frac'], k.get('conv3x3_fwd_mfma_bf16',0), k.get('conv3x3_dgrad_mfma_bf16',0)))"
done

#!/bin/bash
# Same-box A/B of the specialised fp32 Winograd epilogues (ST2_WINO_EPI=0: the generic epilogue): the headline bench, three blocks each.
ARGS="--steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-worker-level --no-extra-configs"
for e in 1 0 1 0; do
  ST2_WINO_EPI=$e python3 bench.py $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ST2_WINO_EPI=$e: %.2f it/s  %.3f ms/step  conv frac %.4f  conv ms %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_per_step'].get('conv3x3_fwd_wino_f32',0)+d['kernel_ms_per_step'].get('conv3x3_dgrad_wino_f32',0)))"
done

#!/bin/bash
# Same-box A/B of environment switches on configs[2] (2048^2, L-BFGS, bf16 conv operands): ab_env16.sh "VAR=val [VAR=val]" ...  (each argument one variant; "" = default)
ARGS="--size 2048 --optimizer lbfgs --precision bf16 --steps 10 --warmup 5 --repeats 3 --no-cpu-baseline --no-worker-level --no-extra-configs"
for rep in 1 2; do
for v in "$@"; do
  env $v python3 bench.py $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('%-44s %.2f it/s  %.3f ms/step  fwd %.3f dgrad %.3f pool_bwd %.3f style %.3f misc %.3f' % ('${v:-default}', d['value'], d['ms_per_step'], k.get('conv3x3_fwd_mfma_bf16',0), k.get('conv3x3_dgrad_mfma_bf16',0), k.get('maxpool_bwd',0), k.get('style_grad_mfma_bf16',0)+k.get('style_grad_mfma_f32',0), k.get('misc',0)))"
done
done

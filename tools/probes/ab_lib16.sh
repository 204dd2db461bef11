#!/bin/bash
# Same-box A/B of two builds of the library on configs[2] (2048^2, L-BFGS, bf16 conv operands): ab_lib16.sh <other libst2_hip.so> [bench args]
# (e.g. the bf16 conv kernel compiled with -DCONV16_BR=0: every tile on the tap-by-tap loop)
OTHER=$1; shift
ARGS="--size 2048 --optimizer lbfgs --precision bf16 --steps 10 --warmup 5 --repeats 3 --no-cpu-baseline --no-worker-level --no-extra-configs $*"
for lib in "" "$OTHER" "" "$OTHER"; do
  ST2_HIP_LIB=${lib:-$PWD/style_transfer2_amd/lib/libst2_hip.so} python3 bench.py $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('%-28s %.2f it/s  %.3f ms/step  conv frac %.4f  fwd %.3f ms dgrad %.3f ms gram %.3f ms' % ('${lib:+other}' or 'this build', d['value'], d['ms_per_step'], d['roofline']['frac'], k.get('conv3x3_fwd_mfma_bf16',0), k.get('conv3x3_dgrad_mfma_bf16',0), k.get('gram_partial_mfma_bf16',0)))"
done

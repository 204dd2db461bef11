#!/bin/bash
# Same-box A/B of the split-operand step (bf16-dense kernels differ by several per cent between boxes of the pool): bench.py --conv-algo 2
# with the library as built, with conv1_2's data gradient on the fp32 kernel (ST2_WS_DGRAD64=0), and -- if a git revision of
# conv3x3_wino_split.hip is given -- with that revision's kernel rebuilt in place (restored afterwards).  Run from the repo root on the GPU box.
set -e
ARGS="--conv-algo 2 --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-worker-level --no-extra-configs"
run() { python3 bench.py $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: %.2f it/s  %.3f ms/step  conv split %.3f + fp32 %.3f ms' % (d['value'], d['ms_per_step'], d['kernel_ms_per_step'].get('conv3x3_fwd_wino_split_bf16x6',0)+d['kernel_ms_per_step'].get('conv3x3_dgrad_wino_split_bf16x6',0), d['kernel_ms_per_step'].get('conv3x3_dgrad_wino_f32',0)+d['kernel_ms_per_step'].get('conv3x3_fwd_wino_f32',0)))"; }
run "as built"
ST2_WS_DGRAD64=0 run "conv1_2 dgrad on the fp32 kernel (unpooling)"
python3 bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-worker-level --no-extra-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('IEEE fp32 headline path on this box: %.2f it/s' % d['value'])"
if [ -n "$1" ] && [ -f "$1" ]; then
  cp style_transfer2_amd/csrc/conv3x3_wino_split.hip /tmp/ws_current.hip
  cp "$1" style_transfer2_amd/csrc/conv3x3_wino_split.hip
  python3 -c "from style_transfer2_amd import build; build.build_lib()" > /dev/null
  run "kernel of $1"
  ST2_WS_DGRAD64=0 run "kernel of $1, conv1_2 dgrad on the fp32 kernel"
  cp /tmp/ws_current.hip style_transfer2_amd/csrc/conv3x3_wino_split.hip
  python3 -c "from style_transfer2_amd import build; build.build_lib()" > /dev/null
fi

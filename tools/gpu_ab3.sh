#!/bin/bash
# Do power-of-two channel-plane strides (1024^2 floats = 4 MiB) slow the kernels that walk many planes at one pixel position?
TAG=$1
mkdir -p gpurun_out
for S in 1024 992 1056 1088; do
  timeout -k 10 200 python bench.py --size $S --no-cpu-baseline --no-worker-level --steps 20 --repeats 3 > gpurun_out/${TAG}_size$S.json 2> gpurun_out/${TAG}_size$S.err || exit 1
  python - <<PY
import json
d = json.load(open('gpurun_out/${TAG}_size$S.json'))
px = $S * $S / 1048576.0
k = d['kernel_ms_per_step']
print('size $S: %.2f it/s (%.3f ms per Mpx)' % (d['value'], d['ms_per_step'] / px), {n: round(v / px, 4) for n, v in k.items() if n in ('conv3x3_dgrad_mfma_f32', 'conv3x3_fwd_mfma_f32', 'gram_partial_mfma_f32', 'style_grad_mfma_f32', 'maxpool_bwd', 'conv3x3_fwd_wino_f32', 'conv3x3_dgrad_wino_f32')})
PY
done

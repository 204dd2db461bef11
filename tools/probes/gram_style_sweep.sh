#!/bin/bash
# A/B of the Gram split plan and the small style-gradient tile on the headline bench (GPU box): kernel ms per step by class.
# Usage: tools/probes/gram_style_sweep.sh > gpurun_out/gram_style_sweep.txt
run() {
  echo "== $1"
  env $1 python bench.py --steps 20 --repeats 3 --no-cpu-baseline --no-worker-level --no-extra-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('%.2f it/s  gram_partial %.4f  gram_reduce %.4f  style_grad %.4f  (ms per step)' % (d['value'], k['gram_partial_mfma_f32'], k['gram_reduce'], k['style_grad_mfma_f32']))"
}
run "ST2_GRAM_BLOCKS=520 ST2_GRAM_BLOCKS64=512 ST2_STYLE_SMALL=0"
run "ST2_NOP=1"
run "ST2_GRAM_BLOCKS64=512"
run "ST2_GRAM_BLOCKS64=1536"
run "ST2_GRAM_BLOCKS64=2048"
run "ST2_GRAM_BLOCKS=256"
run "ST2_GRAM_BLOCKS=520 ST2_GRAM_BLOCKS64=512 ST2_STYLE_SMALL=0"

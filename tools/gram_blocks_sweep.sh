#!/bin/bash
# Sweep of the Gram split-K block target (ST2_GRAM_BLOCKS) on the headline bench (GPU box).
for b in 256 384 512 768 1024 2048; do
  ST2_GRAM_BLOCKS=$b python bench.py --steps 15 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('blocks target $b: %.1f it/s  gram_partial %.3f ms  gram_reduce %.3f ms' % (d['value'], k['gram_partial_mfma_f32'], k['gram_reduce']))"
done

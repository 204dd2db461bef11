#!/bin/bash
# phase times inside the bf16 forward launches (GPU box): prologue wait, main loop, epilogue per workgroup + the shader clock in the loop;
# ST2_BENCH_NODMA=1: the main loop without its staging loads (what the MFMA / LDS side alone sustains); ST2_BENCH_STAGGER=n:
# co-resident workgroups start n x 3.4 us apart; ST2_BENCH_ZERO-like data effects: the probe's operands are small constants
export ST2_BENCH_STAMPS=1
for env in "X=0" "ST2_BENCH_NODMA=1"; do
  echo "== $env"
  env $env python tools/probes/conv16_one.py 64 64 2048 2048 1,4 10
  env $env python tools/probes/conv16_one.py 128 128 1024 1024 1 10
  env $env python tools/probes/conv16_one.py 256 256 512 512 1 10
  env $env python tools/probes/conv16_one.py 512 512 256 256 1 10
done

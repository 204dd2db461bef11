#!/bin/bash
# the short-K bf16 launches of the 2048^2 job in isolation, per tile configuration (GPU box)
for env in "X=0" "ST2_CONV16_CFG=0" "ST2_CONV16_SB_MAXK=128"; do
  echo "== $env"
  env $env python tools/probes/conv16_one.py 64 64 2048 2048 1,4,9,17,33,81,209 20
  env $env python tools/probes/conv16_one.py 64 128 1024 1024 1,9,17 20
  env $env python tools/probes/conv16_one.py 128 128 1024 1024 1,4,17,81 20
done

#!/bin/bash
# What bounds the split-operand Winograd loop: rebuilds conv3x3_wino_split.hip with WS_VAR = 1 (no U refills), 2 (no B builds), 4 (no row
# transform) and their sums into scratch copies of the library and times the layer probe with each.  Run on the GPU box from the repo root:
#   bash tools/probes/wino_split_var.sh [edge]      (the committed library is restored at the end)
set -e
EDGE=${1:-1024}
for v in 0 1 2 3 4 6 7; do
  touch style_transfer2_amd/csrc/conv3x3_wino_split.hip
  ST2_WS_VAR=$v python3 -c "from style_transfer2_amd import build; build.build_lib(); build.build_probes(force=True)" > /dev/null
  echo "=== WS_VAR=$v"
  python3 tools/probes/wino_split_layers.py $EDGE 2>&1 | grep -E "conv(1_2|2_2|3_2|4_2) " || true
done
touch style_transfer2_amd/csrc/conv3x3_wino_split.hip
python3 -c "from style_transfer2_amd import build; build.build_lib(); build.build_probes(force=True)" > /dev/null

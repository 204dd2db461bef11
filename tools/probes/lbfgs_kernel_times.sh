#!/bin/bash
# per-kernel average durations of the L-BFGS kernels in a bf16 2048^2 run (rocprofv3 --kernel-trace --stats); GPU box only
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/lbfgs_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o lb -- python3 $REPO/bench.py --size 2048 --optimizer lbfgs --precision bf16 --no-cpu-baseline --no-worker-level --steps 10 --repeats 1 > $OUT/bench.json 2> $OUT/err.log
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows:
    if "lbfgs" in r["Name"] or "first" in r["Name"] or "64x128_cc4" in r["Name"]:
        print("%-70s calls %5s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY

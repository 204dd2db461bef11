#!/bin/bash
# average durations of the kernels whose name matches $1 (python regex) in a bench.py run with the remaining arguments
# (rocprofv3 --kernel-trace --stats); GPU box only.  Example: tools/probes/kernel_times.sh 'lbfgs|first' --size 2048 --optimizer lbfgs --precision bf16
set -e
PAT=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/kernel_times
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python3 $REPO/bench.py --no-cpu-baseline --no-worker-level --no-extra-configs --steps 10 --repeats 1 "$@" > $OUT/bench.json 2> $OUT/err.log
python3 - <<PY
import csv, glob, re
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows:
    if re.search(r"$PAT", r["Name"]):
        print("%-70s calls %5s avg %8.1f us min %8.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY

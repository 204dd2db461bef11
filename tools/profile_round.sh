#!/bin/bash
# One-shot profiling of the headline bench on the GPU box: kernel-trace stats + three PMC passes (FETCH_SIZE, WRITE_SIZE,
# MFMA busy), each in its own run with --kernel-trace only.
# Outputs under gpurun_out/prof_$1/.  Usage: tools/profile_round.sh <tag> [bench args...]
set -e
TAG=$1; shift
export ST2_PROFILE_TAG=$TAG
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 $ROOT/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-worker-level --no-extra-configs "$@" > $OUT/bench_stats.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- python3 $ROOT/bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-worker-level --no-extra-configs "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- python3 $ROOT/bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-worker-level --no-extra-configs "$@" > $OUT/bench_write.json 2> $OUT/write.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o m -- python3 $ROOT/bench.py --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-worker-level --no-extra-configs "$@" > $OUT/bench_mfma.json 2> $OUT/mfma.log
cd $ROOT
M=$(find $OUT/mfma -name '*counter_collection.csv' | head -1)
python3 tools/pmc_mfma.py $M $OUT/pmc_mfma.json > $OUT/pmc_mfma.txt
rm -rf $OUT/mfma
F=$(find $OUT/fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/write -name '*counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py $F $W $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt
S=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
cp $S $OUT/kernel_stats.csv
T=$(find $OUT/stats -name '*kernel_trace.csv' | head -1)
python3 tools/trace_layers.py $T ${TRACE_SIZE:-1024} > $OUT/per_layer.txt 2>&1 || true
# the raw traces are large: keep the summaries only
rm -rf $OUT/fetch $OUT/write
find $OUT/stats -name '*kernel_trace.csv' -delete

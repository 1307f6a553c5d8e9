#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 --kernel-trace --stats of ONE bench leg: tools/profile_leg.sh <leg> [bench args...]
#   -> gpurun_out/prof_leg/kernel_stats_<leg>.csv, bench_<leg>.json
set -u
NAME=$1; shift
OUT=gpurun_out/prof_leg
mkdir -p $OUT
export TMPDIR=/tmp
export MPF_BENCH_LIVE_TRAFFIC=0
timeout ${PROF_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$NAME -- python3 bench.py --no-cpu --steps 5 --warmup 1 --legs $NAME "$@" > $OUT/bench_$NAME.json 2> $OUT/$NAME.err
find $OUT/trace_$NAME -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$NAME.csv \;
rm -rf $OUT/trace_$NAME
head -6 $OUT/kernel_stats_$NAME.csv | cut -c1-200

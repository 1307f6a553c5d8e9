#!/bin/bash
# rocprofv3 kernel statistics of one bench invocation: tools/prof_leg.sh <tag> <bench args...>  -> gpurun_out/prof_<tag>/kernel_stats.csv
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export MPF_BENCH_LIVE_TRAFFIC=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py "$@" > $OUT/bench.json 2> $OUT/err.txt
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace
head -25 $OUT/kernel_stats.csv | cut -c1-220

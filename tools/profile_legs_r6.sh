#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 --kernel-trace --stats of the round-6 legs, one summary each:
#   gpurun_out/prof_r6legs/kernel_stats_<leg>.csv
set -u
OUT=gpurun_out/prof_r6legs
mkdir -p $OUT
export TMPDIR=/tmp
export MPF_BENCH_LIVE_TRAFFIC=0
leg () {   # name, bench args
  local NAME=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$NAME -- python3 bench.py --no-cpu --steps 5 --warmup 1 "$@" > $OUT/bench_$NAME.json 2> $OUT/$NAME.err
  find $OUT/trace_$NAME -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$NAME.csv \;
  rm -rf $OUT/trace_$NAME
  head -8 $OUT/kernel_stats_$NAME.csv | cut -c1-160
}
leg bb_reference_run --legs start_trees,bb_reference_run --bb-iterations 60 --bb-rounds 2
leg c5_fitch --legs c5_fitch
leg noisy_bootstrap --legs noisy_bootstrap
leg c5_weighted --legs c5_weighted_sweep

#!/bin/bash
# Runs on the GPU box (via gpurun): the round's judged artefacts in one call.
#   tools/round_measure.sh <tag> -> gpurun_out/round_<tag>/{gputest.txt, bench_default.json, concurrent_climbs.txt, kernel_stats_climb.csv, ...}
set -u
TAG=${1:-r3}
OUT=gpurun_out/round_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/gputest.txt 2>&1
tail -3 $OUT/gputest.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.err
{
  echo "# independent C3 climbs from random trees on ONE GPU, one engine per host thread (tools/concurrent_climbs.py)"
  for Q in 4 16; do
    for M in 0 2; do
      echo "## GPU_MAX_HW_QUEUES=$Q climb_device=$M climb_tile=4"
      GPU_MAX_HW_QUEUES=$Q python tools/concurrent_climbs.py --engines 1,4,8,12,16 --opt climb_device=$M --opt climb_tile=4 2>&1 | tail -8
    done
  done
} > $OUT/concurrent_climbs.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/climb_trace -- python3 tools/climb_check.py --workload C3 --notrace > $OUT/climb_c3.txt 2> $OUT/climb_trace.err
find $OUT/climb_trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_climb.csv \;
rm -rf $OUT/climb_trace
tools/profile_gpu.sh $TAG > $OUT/profile.log 2>&1
tail -5 $OUT/profile.log

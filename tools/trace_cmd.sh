#!/bin/bash
# rocprofv3 kernel + memory-copy trace of one python tool: tools/trace_cmd.sh <tag> <script> <args...> -> gpurun_out/trace_<tag>/
set -u
TAG=$1; shift
OUT=gpurun_out/trace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/raw -- python3 "$@" > $OUT/out.txt 2> $OUT/err.txt
tail -5 $OUT/out.txt
python3 tools/batch_timeline.py $OUT/raw 0.5 70 > $OUT/timeline.txt 2>&1
python3 tools/batch_timeline.py $OUT/raw 0.9 70 > $OUT/timeline_late.txt 2>&1
python3 tools/batch_timeline.py $OUT/raw 0.985 70 > $OUT/timeline_end.txt 2>&1
rm -rf $OUT/raw
head -120 $OUT/timeline.txt

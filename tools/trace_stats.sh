#!/bin/bash
# rocprofv3 kernel statistics of one python tool: tools/trace_stats.sh <tag> <script> <args...> -> gpurun_out/stats_<tag>/kernel_stats.csv
set -u
TAG=$1; shift
OUT=gpurun_out/stats_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/out.txt 2> $OUT/err.txt
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace
tail -6 $OUT/out.txt
python3 - <<PY
import csv
for r in list(csv.reader(open("$OUT/kernel_stats.csv")))[1:12]:
    print(r[0][:64].ljust(64), r[1].rjust(7), "%9.2f ms" % (float(r[2]) / 1e6), "%8.1f us" % (float(r[3]) / 1e3))
PY

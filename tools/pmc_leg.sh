#!/bin/bash
# rocprofv3 counter pass of one bench invocation: tools/pmc_leg.sh <tag> "<counters>" <bench args...> -> gpurun_out/pmc_<tag>/summary.txt
set -u
TAG=$1; CTR=$2; shift 2
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export MPF_BENCH_LIVE_TRAFFIC=0
rocprofv3 --pmc $CTR --output-format csv -d $OUT/raw -- python3 bench.py "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/raw/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        acc[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, v in acc.items():
        fh.write(k + "  " + "  ".join(f"{c}={sum(x)/len(x):.4g}(n={len(x)})" for c, x in sorted(v.items())) + "\n")
print(open(out + "/summary.txt").read())
PY
rm -rf $OUT/raw

#!/bin/bash
# rocprofv3 counter pass of any python tool: tools/pmc_cmd.sh <tag> "<counters>" <script.py> [args...] -> gpurun_out/pmc_<tag>/summary.txt
# (one --pmc pass per call, nothing else traced; the program itself behind "--"; FETCH_SIZE and WRITE_SIZE each need a pass of their own --
#  asked for together the profiler aborts and then hangs in its signal handler: hence the timeout)
set -u
TAG=$1; CTR=$2; shift 2
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout ${PMC_TIMEOUT:-240} rocprofv3 --pmc $CTR --output-format csv -d $OUT/raw -- python3 "$@" > $OUT/stdout.txt 2> $OUT/err.txt
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/raw/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        acc[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, v in acc.items():
        fh.write(k + "  " + "  ".join(f"{c}={sum(x)/len(x):.4g}(n={len(x)},max={max(x):.4g})" for c, x in sorted(v.items())) + "\n")
print(open(out + "/summary.txt").read())
PY
tail -3 $OUT/stdout.txt
rm -rf $OUT/raw

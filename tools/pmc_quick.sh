#!/bin/bash
# quick PMC pass for one kernel family: tools/pmc_quick.sh "<counters>" [bench args]
C="$1"; shift
OUT=/tmp/pmcq
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc $C --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu "$@" > /dev/null 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_scan_walk" in k or "k_newview" in k or "k_scan_prog" in k or "k_walk_plan" in k or "k_cntsum" in k:
            acc[k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: round(sum(v)/len(v)) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
PY

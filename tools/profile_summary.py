#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
stats = list(rows("trace/**/*kernel_stats.csv"))
for r in stats:
    name = r.get("Name", "")[:90]
    print(f'{name:90s} calls={r.get("Calls")} total_ns={r.get("TotalDurationNs")} avg_ns={r.get("AverageNs")} pct={r.get("Percentage")}')

print("\n== per-dispatch durations from the kernel trace (ns) ==")
dur = defaultdict(list)
for r in rows("trace/**/*kernel_trace.csv"):
    try:
        dur[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    except Exception:
        pass
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k:60s} n={len(v):5d} avg={sum(v) / len(v):12.0f} med={v2[len(v2) // 2]:10d} max={v2[-1]:10d}")

print("\n== PMC passes (per kernel: mean counter value per dispatch) ==")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    acc = defaultdict(lambda: defaultdict(list))
    for r in rows(os.path.relpath(d, out) + "/**/*counter_collection.csv"):
        try:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        except Exception:
            pass
    print(f"-- {os.path.basename(d)}")
    for k, cs in acc.items():
        line = " ".join(f"{c}={sum(v) / len(v):.4g}(n={len(v)})" for c, v in cs.items())
        print(f"   {k:60s} {line}")

#!/usr/bin/env python3
"""The dispatch chain of a few consecutive batches of a host-driven loop, from a rocprofv3 trace
(--kernel-trace --memory-copy-trace): start offset, duration and the idle gap in front of every kernel / copy.
Usage: tools/batch_timeline.py <trace dir> [first dispatch fraction 0..1] [number of rows]"""
import csv, glob, os, sys

d = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 60
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"]
        nm = nm.split("(")[0].replace("void ", "").replace("mpf::", "").replace("(anonymous namespace)::", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm[:60]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
i0 = int(len(ev) * frac)
t0 = ev[i0][0]
last = ev[i0 - 1][1] if i0 else t0
print(f"{len(ev)} events; window from event {i0}")
for s, e, nm in ev[i0:i0 + nrows]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - last) / 1e3:7.1f}  {nm}")
    last = max(last, e)
# totals per name over the whole trace: busy time, and the idle time in front of each
from collections import defaultdict
busy, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
last = ev[0][0]
for s, e, nm in ev:
    k = nm.split("<")[0] if not nm.startswith("COPY") else " ".join(nm.split()[:2])
    busy[k] += e - s; gap[k] += max(0, s - last); cnt[k] += 1
    last = max(last, e)
print("\nname, calls, busy ms, idle-in-front ms")
for k in sorted(busy, key=lambda k: -(busy[k] + gap[k])):
    print(f"{k[:50]:50s} {cnt[k]:7d} {busy[k] / 1e6:9.2f} {gap[k] / 1e6:9.2f}")

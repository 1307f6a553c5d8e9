#!/usr/bin/env python3
"""How many kernels of concurrent engines really run side by side: from a rocprofv3 kernel trace (..._kernel_trace.csv) of
tools/concurrent_climbs.py, the time-weighted distribution of the number of simultaneously running kernels (all, and k_climb
alone), and the dispatches per hardware queue.   Usage: tools/concurrency_timeline.py <trace dir> [name filter]"""
import csv, glob, os, sys
from collections import Counter

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "k_climb"
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
qcol = next((c for c in rows[0].keys() if c.lower().startswith("queue")), None)
ev_all, ev_f = [], []
perq = Counter()
for r in rows:
    s, e, nm = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]
    ev_all += [(s, 1), (e, -1)]
    if flt in nm:
        ev_f += [(s, 1), (e, -1)]
        if qcol:
            perq[r[qcol]] += 1


def dist(ev):
    ev.sort()
    busy = Counter()
    cur, last = 0, ev[0][0]
    for t, dlt in ev:
        busy[cur] += t - last
        cur += dlt
        last = t
    tot = sum(v for k, v in busy.items() if k > 0)
    return {k: v / tot for k, v in sorted(busy.items()) if k > 0}, tot


da, ta = dist(ev_all)
df, tf = dist(ev_f)
print(f"{len(rows)} dispatches; any kernel running: {ta/1e6:.1f} ms; {flt} running: {tf/1e6:.1f} ms")
print("share of that time with k kernels of ANY kind running :", " ".join(f"{k}:{v:.2f}" for k, v in da.items()))
print(f"share of that time with k {flt} kernels running     :", " ".join(f"{k}:{v:.2f}" for k, v in df.items()))
print(f"mean number of {flt} kernels running while at least one runs: {sum(k * v for k, v in df.items()):.2f}")
if qcol:
    print(f"{flt} dispatches per hardware queue ({qcol}):", dict(perq))

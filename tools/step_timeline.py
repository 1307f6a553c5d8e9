#!/usr/bin/env python3
"""GPU-side timeline of cold sweep steps from a rocprofv3 kernel trace (…_kernel_trace.csv): per dispatch start / end relative to
the step's first dispatch, and the idle gaps in between.  Usage: tools/step_timeline.py <dir with the trace> [steps to print]"""
import csv, glob, os, sys
d = sys.argv[1]
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void mpf::", "").split("(")[0][:28]))
rows.sort()
# a step starts at every full-tree refresh (k_newview_wgq) that is followed by a k_scan_prog
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_scan_prog")]
for i in idx[-nshow:]:
    j = i
    while j > 0 and not rows[j][2].startswith("k_newview_wgq"):
        j -= 1
    k = i
    while k + 1 < len(rows) and rows[k + 1][0] - rows[k][1] < 30000 and not rows[k + 1][2].startswith("k_newview_wgq"):
        k += 1
    j0 = max(0, j - 2)
    t0 = rows[j0][0]
    print("-- step")
    prev_end = None
    for s, e, nme in rows[j0:k + 1]:
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"   {nme:30s} start {(s - t0)/1e3:8.1f} us  dur {(e - s)/1e3:7.1f} us  gap before {gap:6.1f} us")
        prev_end = e
    if k + 1 < len(rows):
        print(f"   (next dispatch {(rows[k + 1][0] - rows[k][1])/1e3:.1f} us after the last one ended)")

#!/usr/bin/env python3
"""Timeline of one planned-program sweep scan (k_scan_prog): when every wave started / ended, on which XCD / CU, and how
the number of resident waves evolves.  Diagnostic only (engine option scan_trace)."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpboot_amd import engine, synth

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C3")
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
cfg = synth.WORKLOADS[a.workload]
letters, _ = synth.workload(a.workload)
codes = synth.letters_to_codes(letters, cfg["alphabet"])
e = engine.FitchEngine(codes)
e.seed_ties(engine.TIE_RANDOM, 1)
e.make_parsimony_tree(12345, 0)
back = e.get_tree()
for kv in a.opt:
    k, v = kv.split("="); e.set_option(k, int(v))
for _ in range(5):
    e.set_tree(back); e.sweep_scan(1, 6)
e.set_option("scan_trace", 1)
e.set_option("timing", 1)
e.reset_stats()
e.set_tree(back); e.sweep_scan(1, 6)
t = e.scan_trace()
st = e.stats()
t = t[t[:, 1] > 0]
b, en = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
t0 = b.min()
b, en = (b - t0) / 100.0, (en - t0) / 100.0       # microseconds
xcc = (t[:, 2] >> 32).astype(np.int64)
hw = (t[:, 2] & 0xFFFFFFFF).astype(np.int64)
tests = (t[:, 3] & 0xFFFF).astype(np.int64)
dur = en - b
print(f"waves {len(t)}  kernel (events) {st['last_scan_kernel_ms']*1e3:.1f} us  span {en.max():.1f} us  mean life {dur.mean():.2f} us  median {np.median(dur):.2f}  max {dur.max():.1f}")
print("life by insertion tests of the scan part: " + "  ".join(f"[{lo}-{hi}) n={((tests>=lo)&(tests<hi)).sum()} mean={dur[(tests>=lo)&(tests<hi)].mean():.1f}us" for lo, hi in ((0, 8), (8, 32), (32, 64), (64, 128), (128, 300)) if ((tests>=lo)&(tests<hi)).any()))
edges = np.linspace(0, en.max(), 21)
for lo, hi in zip(edges[:-1], edges[1:]):
    mid = (lo + hi) / 2
    res = ((b <= mid) & (en > mid)).sum()
    started = ((b >= lo) & (b < hi)).sum()
    print(f"t={mid:7.1f} us resident {res:6d} ({res/ (256*4):.2f}/SIMD) started {started}")
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"xcc {x}: waves {m.sum()} first start {b[m].min():.1f} last end {en[m].max():.1f} busy-sum {dur[m].sum()/1e3:.2f} ms")
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
key = xcc * 1000 + se * 100 + sh * 20 + cu
u, c = np.unique(key, return_counts=True)
print(f"distinct (xcc,se,sh,cu) {len(u)}; waves per CU min {c.min()} max {c.max()}")

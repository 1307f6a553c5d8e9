#!/usr/bin/env python3
"""One randomized stepwise-addition tree at a workload's size (no SPR behind it): seconds and score, optionally with an engine
option set (tools/ras_timing.py C3 views_waves=0)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpboot_amd import engine, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = synth.WORKLOADS[wl]
letters, names = synth.workload(wl)
codes = synth.letters_to_codes(letters, cfg["alphabet"])
dt = engine.DNA if cfg["alphabet"] == "DNA" else engine.AA
e = engine.FitchEngine(codes, datatype=dt)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    e.set_option(k, int(v))
e.seed_ties(engine.TIE_RANDOM, 7)
e.make_parsimony_tree(4242, 0)              # warm-up: allocations
for rep in range(3):
    e.seed_ties(engine.TIE_RANDOM, 7)
    t0 = time.perf_counter()
    s = e.make_parsimony_tree(1234, 0)
    print(f"{wl} {' '.join(sys.argv[2:])}: {time.perf_counter() - t0:.4f} s, score {s}")
print("k_grow: launches", e.get_option("grow_launches"), "steps", e.get_option("grow_steps"), "us", e.get_option("grow_us"), "last_err", e.get_option("grow_last_err"),
      "| phase us of the launches (plan, skeleton, parts, exchange, decide, insert, path):", [e.get_option("grow_ticks_%d" % k) / 100.0 for k in range(7)])

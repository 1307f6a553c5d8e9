#!/usr/bin/env python3
"""One randomized stepwise-addition tree at a workload's size, with and without the pruned refresh of the addition phase
(engine option ras_prune): seconds, score, and that both build the same tree."""
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
res = []
for prune in (1, 0, 1, 0):
    e = engine.FitchEngine(codes, datatype=dt)
    e.set_option("ras_prune", prune)
    e.seed_ties(engine.TIE_RANDOM, 7)
    e.make_parsimony_tree(4242, 0)          # warm-up: allocations
    e.seed_ties(engine.TIE_RANDOM, 7)
    t0 = time.perf_counter()
    s = e.make_parsimony_tree(1234, 0)
    t1 = time.perf_counter()
    res.append((s, e.get_tree().tolist()))
    print(f"{wl} ras_prune={prune}: {t1 - t0:.4f} s, score {s}")
assert all(r == res[0] for r in res), "trees differ"
print("same tree with and without")

#!/usr/bin/env python3
"""Where a sweep step's wall time goes (C3): host scheduling / planning / launch+wait, with and without the plan cache."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpboot_amd import engine, synth
letters, names = synth.workload("C3")
codes = synth.letters_to_codes(letters, "DNA")
e = engine.FitchEngine(codes, datatype=engine.DNA)
e.seed_ties(engine.TIE_RANDOM, 1)
e.make_parsimony_tree(12345, 0)
back = e.get_tree()
for cache in (1, 0):
    e.set_option("plan_cache", cache)
    for _ in range(5):
        e.set_tree(back); e.sweep_scan(1, 6)
    e.reset_stats()
    t0 = time.perf_counter()
    K = 50
    for _ in range(K):
        e.set_tree(back); e.sweep_scan(1, 6)
    dt = (time.perf_counter() - t0) / K
    st = e.stats()
    print(f"plan_cache {cache}: step {dt*1e3:.3f} ms; host views {st['host_views_ms_total']/K:.3f} plan {st['host_plan_ms_total']/K:.3f} scan(launch+wait) {st['host_scan_ms_total']/K:.3f} "
          f"sweep_call {st['host_sweep_ms_total']/K:.3f}")

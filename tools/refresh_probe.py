#!/usr/bin/env python3
"""Kernel time of a from-scratch refresh of all directional vectors at C3: the level-synchronous kernel (device- or host-planned)
against the chained kernel with a host-made chain schedule (option chain_max_ops).   python tools/refresh_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpboot_amd import engine, synth, trees
letters, _ = synth.workload("C3")
codes = synth.letters_to_codes(letters, "DNA")
e = engine.FitchEngine(codes)
e.seed_ties(engine.TIE_RANDOM, 1)
e.make_parsimony_tree(12345, 0)
back = e.get_tree()
for name, opts in (("level kernel, device schedule", {}), ("level kernel, host schedule", {"dev_sched": 0}), ("chain kernel, host schedule", {"dev_sched": 0, "chain_max_ops": 1000000})):
    for k, v in opts.items():
        e.set_option(k, v)
    e.set_option("timing", 2)
    e.set_option("plan_cache", 0)
    for _ in range(3):
        e.set_tree(back); s = e.score_tree()
    e.reset_stats()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        e.set_tree(back); s = e.score_tree()
    dt = (time.perf_counter() - t0) / K
    st = e.stats()
    print(f"{name}: score {s}, call {dt * 1e3:.3f} ms, view kernels {st['view_kernel_ms_total'] / K:.4f} ms, host views {st['host_views_ms_total'] / K:.3f} ms", flush=True)

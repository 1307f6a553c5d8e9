"""C5 (protein, 20 rows) sweep step under different refresh tiles / kernels.   python tools/c5_views_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpboot_amd import engine, synth
letters, _ = synth.workload("C5")
codes = synth.letters_to_codes(letters, "AA")
e0 = engine.FitchEngine(codes, datatype=engine.AA)
e0.seed_ties(engine.TIE_RANDOM, 1); e0.make_parsimony_tree(12345, 0); back = e0.get_tree()
for opts in ({}, {"views_tile": 8}, {"views_tile": 16}, {"views_tile": 32}, {"views_pipe": 0}, {"scan_prog": 0}):
    e = engine.FitchEngine(codes, datatype=engine.AA)
    for k, v in opts.items():
        e.set_option(k, v)
    e.set_option("plan_cache", 0)
    e.set_option("timing", 2)
    for _ in range(4):
        e.set_tree(back); e.sweep_scan(1, 6)
    e.reset_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        e.set_tree(back); k_, b_ = e.sweep_scan(1, 6)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    st = e.stats()
    print(opts, f"step {dt * 1e3:.3f} ms (timing on), views {st['view_kernel_ms_total'] / 20:.3f} ms, scan {st['scan_kernel_ms_total'] / max(1, st['scan_launches']):.3f} ms, tests {k_}, best {b_}", flush=True)

#!/usr/bin/env python3
"""What one move-less whole sweep with the -bb bookkeeping costs at C3, without a cut-off and (argument: repetitions) under one --
MPF_UFB_PROFILE=1 splits it into scan / device / replay.   python tools/sweep_cut_probe.py [n]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpboot_amd import engine, synth
cfg = synth.WORKLOADS["C3"]
letters, names = synth.workload("C3")
codes = synth.letters_to_codes(letters, "DNA")
e = engine.FitchEngine(codes, datatype=engine.DNA)
e.seed_ties(engine.TIE_RANDOM, 1)
P = codes.shape[1]
samples = np.random.default_rng(1).multinomial(P, np.ones(P) / P, size=1000).astype(np.uint16)
s0 = e.make_parsimony_tree(1001, 0)
back0 = e.get_tree()
e.ufboot_attach(samples)
for rep in range(2):
    e.set_tree(back0); e.reset_node_order()
    t0 = time.perf_counter(); e.optimize_spr(1, 6); t1 = time.perf_counter()
    print("sweep without cut-off", t1 - t0)
if len(sys.argv) > 1:
    cut = e.ufboot_next_cutoff(10); e.ufboot_set_cutoff(cut)
    for rep in range(int(sys.argv[1])):
        e.set_tree(back0); e.reset_node_order()
        t0 = time.perf_counter(); e.optimize_spr(1, 6); t1 = time.perf_counter()
        print("sweep under cut-off", -cut, t1 - t0)
e.ufboot_detach()

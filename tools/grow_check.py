#!/usr/bin/env python3
"""k_grow against the host's own addition loop on one synthetic alignment: python tools/grow_check.py n P [alphabet] [opt=value ...]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpboot_amd import engine, synth
n, P = int(sys.argv[1]), int(sys.argv[2])
alpha = sys.argv[3] if len(sys.argv) > 3 and "=" not in sys.argv[3] else "DNA"
letters, _ = synth.synth_alignment(n, P, alpha, 0.07, seed=n + P)
codes = synth.letters_to_codes(letters, alpha)
dt = engine.DNA if alpha == "DNA" else engine.AA
res = []
for dev in (0, 1):
    e = engine.FitchEngine(codes, datatype=dt)
    for kv in sys.argv[3:]:
        if "=" in kv:
            k, v = kv.split("="); e.set_option(k, int(v))
    e.set_option("grow_device", dev)
    out = []
    for rep, seed in enumerate((4242, 1234, 99)):          # (the later calls find tr->nodep as nodeRectifierPars left it)
        e.seed_ties(engine.TIE_RANDOM, 7)
        t0 = time.perf_counter()
        sc = e.make_parsimony_tree(seed, 3 if rep == 1 else 0)
        dtm = time.perf_counter() - t0
        out.append((sc, e.get_tree().tolist(), e.tie_state()))
    e.seed_ties(engine.TIE_RANDOM, 7)
    r = e.stepwise_addition(1234)
    res.append((r, out, e.get_tree().tolist(), e.tie_state()))
    print(f"n={n} P={P} {alpha} grow_device={dev}: {dtm:.4f}s launches {e.get_option('grow_launches')} err {e.get_option('grow_last_err')}", flush=True)
a, b = res
same = a[1:] == b[1:] and all(np.array_equal(x, y) for x, y in zip(a[0], b[0]))
print("IDENTICAL" if same else "DIFFERENT")

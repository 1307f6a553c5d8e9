#!/usr/bin/env python3
"""How the bootstrap refinement behaves per sample at C3: distinct start topologies, accepted moves (improving / sideways)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpboot_amd import engine, synth, trees, shard

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
nref = int(sys.argv[3]) if len(sys.argv) > 3 else 100
cfg = synth.WORKLOADS[wl]
letters, _ = synth.workload(wl)
codes = synth.letters_to_codes(letters, cfg["alphabet"])
n, P = codes.shape
eng = engine.FitchEngine(codes)
samples = np.random.default_rng(4242).multinomial(P, np.ones(P) / P, size=B).astype(np.uint16)
back = trees.random_topology(n, np.random.default_rng(2024))
eng.ufboot_attach(samples, 0.5)
eng.set_tree(back); eng.reset_node_order(); eng.seed_ties(engine.TIE_RANDOM, 1)
t0 = time.perf_counter()
s = eng.optimize_spr(1, 6)
print("online phase", time.perf_counter() - t0, "s score", s)
logl, cnt, bt = eng.ufboot_state()
cache = {}
tl = []
for b in range(B):
    t = int(bt[b])
    if t not in cache:
        cache[t] = eng.ufboot_tree(t)
    tl.append(cache[t])
print("distinct boot trees", len(cache), "of", B)
eng.ufboot_detach()
mv, imp = [], 0
t0 = time.perf_counter()
for b in range(nref):
    eng.set_weights(samples[b].astype(np.int32))
    eng.seed_ties(1, shard.unit_seed(5, b))
    eng.reset_node_order()
    eng.set_tree(np.asarray(tl[b], dtype=np.int32))
    eng.reset_stats()
    s0 = eng.score_tree()
    s1 = eng.optimize_spr(1, 6)
    mv.append(eng.stats()["moves_applied"])
    imp += s1 < s0
print("refined", nref, "in", time.perf_counter() - t0, "s; moves per sample:", np.bincount(mv).tolist(), "improved", imp)

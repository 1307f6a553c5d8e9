#!/usr/bin/env python3
"""One alignment whose directional-vector store exceeds 2 GiB (the scan kernel's 64-bit addressing path at real size):
RAS tree, full sweep, and the candidate lists of a few prune nodes checked against the CPU oracle."""
import argparse, sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpboot_amd import engine, synth

ap = argparse.ArgumentParser()
ap.add_argument("--taxa", type=int, default=4000)
ap.add_argument("--sites", type=int, default=300000)
ap.add_argument("--check", type=int, default=3)
a = ap.parse_args()
rng = np.random.default_rng(7)
n, m = a.taxa, a.sites
children, leaves = synth.random_tree_parents(n, rng)
leaf_pos = {v: i for i, v in enumerate(leaves)}
letters = np.empty((n, m), dtype=np.uint8)
seqs = {0: rng.integers(0, 4, size=m, dtype=np.uint8)}
stack = [0]
while stack:
    v = stack.pop()
    s = seqs.pop(v)
    if not children[v]:
        letters[leaf_pos[v]] = s
        continue
    for c in children[v]:
        mut = rng.random(m) < 0.04
        seqs[c] = np.where(mut, rng.integers(0, 4, size=m, dtype=np.uint8), s)
        stack.append(c)
codes = (1 << letters).astype(np.uint8)
del letters
t0 = time.perf_counter()
e = engine.FitchEngine(codes)
t1 = time.perf_counter()
store = (n + 3 * (n - 1)) * e.S * e.Wp * 4
print(f"{n} taxa x {m} sites: {e.num_informative} informative patterns, W = {e.W}, vector store {store / 2**30:.2f} GiB, engine created in {t1 - t0:.1f} s")
e.seed_ties(engine.TIE_RANDOM, 1)
t0 = time.perf_counter(); s0 = e.make_parsimony_tree(4242, 0); t1 = time.perf_counter()
back = e.get_tree()
print(f"randomized stepwise addition: length {s0} in {t1 - t0:.1f} s")
e.set_option("timing", 2)
e.set_tree(back); e.sweep_scan(1, 6); e.reset_stats()
t0 = time.perf_counter(); e.set_tree(back); k, best = e.sweep_scan(1, 6); dt = time.perf_counter() - t0
st = e.stats()
print(f"full sweep: {k} insertion tests in {dt * 1e3:.1f} ms = {k / dt:.3e} evals/s (scan kernel {st['scan_kernel_ms_total']:.1f} ms), best candidate {best}")
if a.check:
    from oracle import pyoracle as po
    o = po.Oracle(codes)
    assert o.score_tree(back) == s0
    o.seed_ties(po.TIE_RANDOM, 1)
    nodep = o.nodep()
    for i in rng.integers(1, 2 * n - 2, size=a.check):
        rec = int(nodep[i])
        o.set_best(s0); o.trace(True); o.rearrange(rec, 1, 6)
        q_o, mp_o = o.get_trace()
        keep = q_o >= 0
        q, mp, n_p = e.spr_scan(rec, 1, 6)
        assert q.tolist() == q_o[keep].tolist() and mp.tolist() == mp_o[keep].tolist(), rec
        print(f"prune record {rec}: {len(q)} candidates identical to the oracle")

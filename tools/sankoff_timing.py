#!/usr/bin/env python3
"""Time the weighted (Sankoff, `-cost`) engine: full sweep scan on the GPU vs the C oracle on a bounded sample."""
import argparse, sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpboot_amd import engine, synth, trees

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C5")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--cpu-budget", type=float, default=10.0)
ap.add_argument("--bb", type=int, default=0, help="also time one SPR climb with the saveCurrentTree bookkeeping for this many bootstrap samples")
a = ap.parse_args()
cfg = synth.WORKLOADS[a.workload]
letters, names = synth.workload(a.workload)
codes = synth.letters_to_codes(letters, cfg["alphabet"])
dt = engine.DNA if cfg["alphabet"] == "DNA" else engine.AA
S = 4 if dt == engine.DNA else 20
rng = np.random.default_rng(5)
c = rng.integers(1, 6, size=(S, S)); cost = (np.triu(c, 1) + np.triu(c, 1).T).astype(np.uint32)
n, P = codes.shape
f = engine.FitchEngine(codes, datatype=dt)
f.seed_ties(engine.TIE_RANDOM, 1); f.make_parsimony_tree(12345, 0); back = f.get_tree(); del f
e = engine.FitchEngine(codes, datatype=dt, cost=cost)
e.set_option("timing", 2)
e.set_tree(back); s0 = e.score_tree()
e.sweep_scan(1, 6)
e.reset_stats()
t0 = time.perf_counter()
for _ in range(a.steps):
    e.set_tree(back); k, best = e.sweep_scan(1, 6)
dt_s = (time.perf_counter() - t0) / a.steps
st = e.stats()
print(f"{a.workload} Sankoff ({S} states, random symmetric costs 1..5): tree length {s0}; sweep of {k} insertion tests in {dt_s*1e3:.2f} ms "
      f"= {k/dt_s:.3e} evals/s (scan kernels {st['scan_kernel_ms_total']/a.steps:.2f} ms, view kernels {st['view_kernel_ms_total']/a.steps:.2f} ms per step); best candidate {best}")
if a.bb:
    samples = np.random.default_rng(1).multinomial(P, np.ones(P) / P, size=a.bb).astype(np.uint16)
    e.set_tree(back); e.seed_ties(engine.TIE_RANDOM, 1)
    t0 = time.perf_counter(); s_plain = e.optimize_spr(1, 6); t_plain = time.perf_counter() - t0
    mv_plain = len(e.moves()[0])
    e.set_tree(back); e.seed_ties(engine.TIE_RANDOM, 1); e.reset_node_order()
    e.ufboot_attach(samples)
    e.reset_stats()
    t0 = time.perf_counter(); s_bb = e.optimize_spr(1, 6); t_bb = time.perf_counter() - t0
    st2, cn = e.stats(), e.ufboot_counters()
    print(f"-bb {a.bb}: plain climb -> {s_plain} in {t_plain:.3f} s ({mv_plain} moves); with bookkeeping -> {s_bb} in {t_bb:.3f} s "
          f"({st2['insertion_tests']} tests, {st2['moves_applied']} moves, {len(e.ufboot_tree_logl())} trees booked, product kernels {cn['reps_kernel_ms']:.1f} ms "
          f"over {cn['reps_rows']} plane rows, {cn['events']} events, {cn['tie_draws']} draws)")
    e.ufboot_detach()
from oracle import pyoracle as po
o = po.Oracle(codes, datatype=po.DNA if dt == engine.DNA else po.AA, cost=cost)
assert o.score_tree(back) == s0
o.seed_ties(po.TIE_RANDOM, 1); o.set_best(s0)
nodep = o.nodep()
tc0 = time.perf_counter(); k0 = o.counters()[2]; i = 1
while time.perf_counter() - tc0 < a.cpu_budget and i <= 2 * n - 2:
    o.rearrange(int(nodep[i]), 1, 6); i += 1
tc = time.perf_counter() - tc0; kc = o.counters()[2] - k0
print(f"CPU port (scalar C, exact 32-bit): {kc} insertion tests in {tc:.1f} s = {kc/tc:.3e} evals/s -> GPU/CPU {k/dt_s/(kc/tc):.0f}x")

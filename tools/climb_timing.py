#!/usr/bin/env python3
"""Time a full SPR hill climb (RAS start tree -> SPR-local optimum) on the GPU engine; optional oracle check."""
import argparse, sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpboot_amd import engine, synth

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C2")
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--check", action="store_true")
ap.add_argument("--start", default="ras")
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
cfg = synth.WORKLOADS[a.workload]
letters, names = synth.workload(a.workload)
codes = synth.letters_to_codes(letters, cfg["alphabet"])
dt = engine.DNA if cfg["alphabet"] == "DNA" else engine.AA
e = engine.FitchEngine(codes, datatype=dt)
e.set_option("timing", 0)
for kv in a.opt:
    k, v = kv.split("="); e.set_option(k, int(v))
e.seed_ties(engine.TIE_RANDOM, a.seed)
if a.start == "ras":
    t0 = time.perf_counter(); s0 = e.make_parsimony_tree(1000 + a.seed, 0); t1 = time.perf_counter()
else:
    from mpboot_amd import trees
    t0 = time.perf_counter(); s0 = e.score_tree(trees.random_topology(codes.shape[0], np.random.default_rng(a.seed))); t1 = time.perf_counter()
st0 = e.stats(); e.reset_stats()
back0 = e.get_tree()
s1 = e.optimize_spr(1, 6); t2 = time.perf_counter()
st = e.stats()
print(f"{a.workload}: RAS tree score {s0} in {t1-t0:.3f}s (tests {st0['insertion_tests']}); SPR climb -> {s1} in {t2-t1:.3f}s; moves {st['moves_applied']} "
      f"tests {st['insertion_tests']} scan_launches {st['scan_launches']} view_launches {st['view_launches']} "
      f"scan_kernel_ms {st['scan_kernel_ms_total']:.1f} view_kernel_ms {st['view_kernel_ms_total']:.1f} host_plan {st['host_plan_ms_total']:.1f} host_views {st['host_views_ms_total']:.1f} host_scan {st['host_scan_ms_total']:.1f}")
if a.check:
    from oracle import pyoracle as po
    o = po.Oracle(codes, datatype=dt)
    o.seed_ties(po.TIE_RANDOM, a.seed)
    t3 = time.perf_counter()
    if a.start == "ras":
        so, _ = o.make_tree(1000 + a.seed, 0)
    else:
        so = o.score_tree(back0)
    t4 = time.perf_counter()
    assert so == s0 and (o.get_tree() == back0).all(), "start tree mismatch"
    s2 = o.optimize_spr(1, 6); t5 = time.perf_counter()
    print(f"oracle: RAS {t4-t3:.2f}s, climb {t5-t4:.2f}s score {s2}; identical tree: {(o.get_tree() == e.get_tree()).all()} counters {o.counters()}")
    assert s2 == s1

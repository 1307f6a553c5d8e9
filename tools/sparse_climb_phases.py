#!/usr/bin/env python3
"""k_climb's phases on climbs that start near an optimum (the later iterations of a search: the best tree perturbed by 30 random SPR
moves): moves are sparse, most steps end without one.   python tools/sparse_climb_phases.py [--opt k=v ...]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpboot_amd import engine, synth, trees

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C3")
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--climbs", type=int, default=6)
ap.add_argument("--device", type=int, default=2)
ap.add_argument("--nni", type=int, default=0, help="perturb with this many random NNIs (doRandomNNIs: floor(0.5 (n - 3)) = 498 at C3) instead of 30 random SPR moves")
a = ap.parse_args()
cfg = synth.WORKLOADS[a.workload]
letters, _ = synth.workload(a.workload)
codes = synth.letters_to_codes(letters, cfg["alphabet"])
e = engine.FitchEngine(codes, datatype=engine.DNA if cfg["alphabet"] == "DNA" else engine.AA)
n = codes.shape[0]
e.set_tree(trees.random_topology(n, np.random.default_rng(2024))); e.seed_ties(engine.TIE_RANDOM, 1)
e.optimize_spr(1, 6)
best = e.get_tree()
scratch = engine.FitchEngine(codes, datatype=engine.DNA if cfg["alphabet"] == "DNA" else engine.AA)
e.set_option("climb_device", a.device)
for kv in a.opt:
    k, v = kv.split("="); e.set_option(k, int(v))
rng = np.random.default_rng(5)
names = "setup enum closure refresh scan exchange decide".split()
tot = np.zeros(7); wall = 0.0; steps = nodes = moves = launches = 0
for c in range(a.climbs + 1):
    pert = engine.iq_random_nnis(best, a.nni, 1000 + c)[0] if a.nni else trees.random_spr_moves(scratch, best, rng, 30, 6)
    e.set_tree(pert); e.reset_node_order(); e.reset_stats()
    p0 = np.array([e.get_option(f"climb_phase_us{k}") for k in "0123456"], dtype=float)
    t0 = time.perf_counter()
    e.optimize_spr(1, 6)
    dt = time.perf_counter() - t0
    p1 = np.array([e.get_option(f"climb_phase_us{k}") for k in "0123456"], dtype=float)
    st = e.stats()
    if c == 0:
        continue
    tot += p1 - p0; wall += dt; steps += st["climb_steps"]; nodes += st["climb_nodes"]; moves += st["moves_applied"]; launches += st["climb_launches"]
print(f"{a.workload} {a.opt}: {a.climbs} climbs, {wall / a.climbs * 1e3:.2f} ms each; {moves / a.climbs:.0f} moves, {launches / a.climbs:.1f} launches, {steps / a.climbs:.0f} steps, "
      f"{nodes / a.climbs:.0f} prune nodes in k_climb | us per step: " + " ".join(f"{nm} {t / max(steps, 1):.1f}" for nm, t in zip(names, tot)) + f" | sum {tot.sum() / max(steps, 1):.1f}")

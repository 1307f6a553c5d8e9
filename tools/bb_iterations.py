#!/usr/bin/env python3
"""The -bb search as a run repeats it: one tracked climb from a start tree, then N later iterations -- cut-off from the saved trees
(top 10 %, iqtree.cpp:1662-1676), the best tree perturbed by k random SPR moves (stand-in for doRandomNNIs, iqtree.cpp:1742-1747),
one more tracked climb under the cut-off (iqtree.cpp:2132) -- and the refinement of every sample's tree behind them.

   python tools/bb_iterations.py [--workload C3] [--iters 50] [--samples 1000] [--opt key=value ...]
MPF_UFB_PROFILE=1 prints the tracker's host-side split at detach."""
import argparse, hashlib, os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpboot_amd import engine, synth, trees, bootstrap


run = bootstrap.bb_search


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--perturb", type=int, default=30)
    ap.add_argument("--maxtrav", type=int, default=6)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--ratchet-every", type=int, default=0)
    ap.add_argument("--start", default="random", choices=["random", "ras"])
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args()
    cfg = synth.WORKLOADS[a.workload]
    letters, names = synth.workload(a.workload)
    codes = synth.letters_to_codes(letters, cfg["alphabet"])
    e = engine.FitchEngine(codes, datatype=engine.DNA if cfg["alphabet"] == "DNA" else engine.AA)
    for kv in a.opt:
        k, v = kv.split("=")
        e.set_option(k, int(v))
    n, P = codes.shape
    samples = np.random.default_rng(4242).multinomial(P, np.ones(P) / P, size=a.samples).astype(np.uint16)
    if a.start == "ras":
        e.seed_ties(engine.TIE_RANDOM, a.seed)
        e.make_parsimony_tree(1000 + a.seed, 0)
        start = e.get_tree()
    else:
        start = trees.random_topology(n, np.random.default_rng(2024))
    for rep in range(2):               # the first pass allocates
        r = run(e, samples, start, a.iters if rep else min(3, a.iters), a.perturb, a.maxtrav, a.seed, a.ratchet_every, a.v and rep == 1)
    its = np.array(r["iter_s"])
    print(json.dumps({"workload": a.workload, "samples": a.samples, "first_climb_s": r["first_s"], "iterations": len(its),
                      "iter_ms_mean": float(its.mean() * 1e3) if len(its) else None, "iter_ms_median": float(np.median(its) * 1e3) if len(its) else None,
                      "iter_ms_max": float(its.max() * 1e3) if len(its) else None, "iters_total_s": float(its.sum()),
                      "refine_s": r["refine_s"], "wall_s": r["first_s"] + float(its.sum()) + r["refine_s"],
                      "moves_mean": float(np.mean([s[0] for s in r["iter_stats"]])) if len(its) else None,
                      "tests_mean": float(np.mean([s[1] for s in r["iter_stats"]])) if len(its) else None,
                      "best_score": r["best_score"], "saved_trees": r["saved_trees"], "distinct_boot_trees": r["distinct_boot_trees"],
                      "state_hash": r["state_hash"]}))

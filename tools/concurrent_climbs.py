#!/usr/bin/env python3
"""Independent SPR climbs on ONE GPU, one engine per host thread (the shape of the 100 start trees / the bootstrap refinements of a
run): a single climb is a chain of ~70 us batches, latency all the way down, so concurrent engines fill each other's gaps."""
import argparse, sys, os, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpboot_amd import engine, synth, trees

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C3")
ap.add_argument("--engines", default="1,2,4,8,12,16")
ap.add_argument("--climbs", type=int, default=1, help="climbs per engine")
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
cfg = synth.WORKLOADS[a.workload]
letters, names = synth.workload(a.workload)
codes = synth.letters_to_codes(letters, cfg["alphabet"])
dt = engine.DNA if cfg["alphabet"] == "DNA" else engine.AA
n = codes.shape[0]
emax = max(int(x) for x in a.engines.split(","))
engs = [engine.FitchEngine(codes, datatype=dt) for _ in range(emax)]
for e in engs:
    e.set_option("timing", 0)
    for kv in a.opt:
        k, v = kv.split("="); e.set_option(k, int(v))
    e.score_tree(trees.random_topology(n, np.random.default_rng(999)))       # warm-up: buffers
    e.optimize_spr(1, 6)
for E in (int(x) for x in a.engines.split(",")):
    starts = [[trees.random_topology(n, np.random.default_rng(100 * k + c)) for c in range(a.climbs)] for k in range(E)]
    res = [None] * E
    busy = [0.0] * emax

    def work(k):
        e = engs[k]
        out = []
        for c in range(a.climbs):
            e.set_tree(starts[k][c]); e.seed_ties(engine.TIE_RANDOM, k + 1); e.reset_node_order()
            t_ = time.perf_counter()
            out.append(e.optimize_spr(1, 6))
            busy[k] += time.perf_counter() - t_
        res[k] = out

    th = [threading.Thread(target=work, args=(k,)) for k in range(E)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt_s = time.perf_counter() - t0
    sts = [engs[k].stats() for k in range(E)]
    print(f"   kernel launches {sum(s_['climb_launches'] for s_ in sts)} host scan launches {sum(s_['scan_launches'] for s_ in sts)} "
          f"kernel ms/engine {np.mean([s_['climb_ms_total'] for s_ in sts]):.0f} | host ms/engine: views {np.mean([s_['host_views_ms_total'] for s_ in sts]):.0f} "
          f"plan {np.mean([s_['host_plan_ms_total'] for s_ in sts]):.0f} scan {np.mean([s_['host_scan_ms_total'] for s_ in sts]):.0f} | busy per engine {np.mean(busy[:E]) * 1e3:.0f} ms")
    for k in range(E):
        engs[k].reset_stats()
    print(f"{a.workload}: {E} engines x {a.climbs} climbs from random trees in {dt_s:.3f} s = {E * a.climbs / dt_s:.2f} climbs/s "
          f"({dt_s / a.climbs:.3f} s per round; scores {min(min(r) for r in res)}..{max(max(r) for r in res)})", flush=True)

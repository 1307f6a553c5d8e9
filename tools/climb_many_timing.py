"""Throughput of independent SPR climbs in ONE launch per round (mpf_optimize_spr_many / k_climb_many: one resident workgroup per climb).
   python tools/climb_many_timing.py [--workload C2] [--engines 64,128,256] [--opt climb_tile=4]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpboot_amd import engine, synth, trees
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C2")
ap.add_argument("--engines", default="64,128,256")
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--phases", action="store_true", help="per-phase time of three of the climbs (their workgroup's own clock)")
a = ap.parse_args()
cfg = synth.WORKLOADS[a.workload]
letters, _ = synth.workload(a.workload)
codes = synth.letters_to_codes(letters, cfg["alphabet"])
dt = engine.DNA if cfg["alphabet"] == "DNA" else engine.AA
n = codes.shape[0]
emax = max(int(x) for x in a.engines.split(","))
t0 = time.perf_counter()
engs = []
for k in range(emax):
    e = engine.FitchEngine(codes, datatype=dt)
    e.set_option("timing", 0)
    for kv in a.opt:
        kk, vv = kv.split("="); e.set_option(kk, int(vv))
    engs.append(e)
print(f"{emax} engines created in {time.perf_counter() - t0:.1f} s", flush=True)
for E in (int(x) for x in a.engines.split(",")):
    for rep in range(a.reps):
        starts = [trees.random_topology(n, np.random.default_rng(1000 * rep + k)) for k in range(E)]
        for k in range(E):
            engs[k].set_tree(starts[k]); engs[k].reset_node_order(); engs[k].seed_ties(engine.TIE_RANDOM, k + 1); engs[k].reset_stats()
        t0 = time.perf_counter()
        sc = engine.optimize_spr_many(engs[:E], 1, 6)
        dt_s = time.perf_counter() - t0
    steps = sum(e.stats()["climb_steps"] for e in engs[:E]); launches = sum(e.stats()["climb_launches"] for e in engs[:E])
    print(f"{a.workload} {a.opt}: {E} climbs in {dt_s:.3f} s = {E / dt_s:.1f} climbs/s (scores {int(sc.min())}..{int(sc.max())}; "
          f"{steps / E:.0f} kernel steps and {launches / E:.1f} launches per climb)", flush=True)
    if a.phases:
        # (cumulative since the engine was made: the last repetition dominates only if reps == 1)
        names = ["set-up", "enumerate", "closure", "refresh", "scan", "exchange", "decide"]
        for k in (0, E // 2, E - 1):
            us = [engs[k].get_option(f"climb_phase_us{j}") for j in range(7)]
            print(f"  climb {k}: " + ", ".join(f"{nm} {u / 1e3:.1f} ms" for nm, u in zip(names, us)) + f"; sum {sum(us) / 1e3:.1f} ms; "
                  f"refresh ops {engs[k].get_option('climb_ctr0')}, chains {engs[k].get_option('climb_ctr3')}, steps {engs[k].stats()['climb_steps']}, "
                  f"tests {engs[k].stats()['insertion_tests']}; plain-path steps {engs[k].get_option('climb_phase_use')} took "
                  f"{engs[k].get_option('climb_phase_usd') / 1e3:.1f} ms, sequence passes {engs[k].get_option('climb_phase_usf') / 1e3:.1f} ms", flush=True)

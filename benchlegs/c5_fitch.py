"""bench.py legs `c5_fitch_sweep` / `c5_fitch_climb`: BASELINE config 5 in its Fitch form -- 500 taxa x 20 000 protein patterns on the
20-row kernels (the reference's `case 20`, sprparsimony.cpp:779-823, :1127-1163).

  sweep   the headline's step on this alignment: a tree handed over (plan cache off), all views refreshed, every insertion test of
          every prune node within the radius scored (rearrangeParsimony over the sweep), best length per prune node back on the host
  climb   pllOptimizeSprParsimony from a random tree (k_climb, five states per lane)

CPU side: the reference's own PLL AVX code on the same inputs (oracle/_ref/pll_ref_driver time / spr: kind "reference")."""
import time

import numpy as np


def run(device, maxtrav, steps, warmup, barrier, cpu_baseline=None, climb_cpu_baseline=None, cpu_budget=10.0):
    import torch
    from mpboot_amd import engine, synth, trees
    letters, names = synth.workload("C5")
    codes = synth.letters_to_codes(letters, "AA")
    n, P = codes.shape
    eng = engine.FitchEngine(codes, datatype=engine.AA, device=device)
    eng.seed_ties(engine.TIE_RANDOM, 1)
    eng.make_parsimony_tree(12345, 0)
    back = eng.get_tree()
    eng.set_option("timing", 1)
    eng.set_option("plan_cache", 0)
    for _ in range(max(3, warmup)):
        eng.set_tree(back)
        eng.sweep_scan(1, maxtrav)
    eng.reset_stats()
    barrier()
    t0 = time.perf_counter()
    tests = 0
    for _ in range(steps):
        eng.set_tree(back)
        k, _best = eng.sweep_scan(1, maxtrav)
        tests += k
    barrier()
    dt = time.perf_counter() - t0
    st = eng.stats()
    eng.set_option("timing", 2)
    eng.reset_stats()
    for _ in range(3):
        eng.set_tree(back)
        eng.sweep_scan(1, maxtrav)
    torch.cuda.synchronize()
    view_ms = eng.stats()["view_kernel_ms_total"] / 3
    eng.set_option("timing", 0)
    kernel_ms = st["scan_kernel_ms_total"] / max(1, st["scan_launches"])
    launches_per_step = st["scan_launches"] / steps
    S, Wp = 20, int(eng.Wp) if hasattr(eng, "Wp") else None
    vec_bytes = 20 * (Wp or ((P + 31) // 32 + 31) // 32 * 32) * 4
    per_step = tests / steps
    achieved = per_step * vec_bytes / (kernel_ms * launches_per_step * 1e-3) / 1e9
    sweep = {"workload": "C5: %d taxa x %d protein patterns, Fitch on 20 state rows, SPR radius %d, RAS start tree" % (n, P, maxtrav),
             "ms_per_step": dt / steps * 1e3, "steps": steps, "evals_per_step": per_step, "evals_per_s": tests / dt,
             "site_ops_per_s": n * P * tests / dt, "scan_kernel_ms_per_step": kernel_ms * launches_per_step, "views_ms_per_step": view_ms,
             "roofline": {"bound": "l2", "achieved": achieved, "peak": 34500.0, "unit": "GB/s", "frac": achieved / 34500.0, "traffic": None,
                          "kernel": "k_scan_walk<16 | 20-row split>" , "bytes_per_eval": vec_bytes,
                          "note": "one 20-row vector (%d B) loaded per insertion test, priced against the aggregate L2 -> CU rate as the headline's "
                                  "DNA kernel is" % vec_bytes}}
    if cpu_baseline is not None:
        cb = cpu_baseline(codes, back, names, letters, "AA", maxtrav, cpu_budget, all_cores=False)
        if cb:
            sweep["cpu_baseline"] = cb
            sweep["gpu_over_cpu"] = sweep["site_ops_per_s"] / cb["value"]
    # ---- the climb
    start = trees.random_topology(n, np.random.default_rng(9))
    tt = []
    for _ in range(3):
        eng.set_option("plan_cache", 1)
        eng.set_tree(start)
        eng.reset_node_order()
        eng.seed_ties(engine.TIE_RANDOM, 5)
        eng.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s = eng.optimize_spr(1, maxtrav)
        tt.append(time.perf_counter() - t0)
    st = eng.stats()
    climb = {"workload": "C5 Fitch: full SPR hill climb from a random topology (numpy default_rng(9)), radius %d" % maxtrav,
             "seconds": min(tt[1:]), "seconds_each_pass": tt, "score": int(s), "moves": st["moves_applied"], "insertion_tests": st["insertion_tests"],
             "climb_kernel_launches": st["climb_launches"], "climb_kernel_steps": st["climb_steps"], "climb_kernel_ms": st["climb_ms_total"],
             "us_per_kernel_step": st["climb_ms_total"] * 1e3 / max(1, st["climb_steps"])}
    if climb_cpu_baseline is not None:
        cb = climb_cpu_baseline(names, letters, "AA", start, maxtrav)
        if cb:
            climb["cpu_baseline"] = cb
            climb["gpu_over_cpu"] = cb["seconds"] / climb["seconds"]
    return sweep, climb

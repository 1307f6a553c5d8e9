"""bench.py leg `climbs_in_one_launch`: independent SPR hill climbs from random trees as workgroups of ONE launch per round
(mpf_optimize_spr_many / k_climb_many: one resident workgroup per climb works through every tile of sites itself, sweep after sweep
until the climb is at its optimum).  Every climb makes the moves of its solo mpf_optimize_spr call (tests/test_gpu_climb_many.py).
Beside it in the line: the same climbs on a host thread per engine (concurrent_climbs, c2_climb.concurrent)."""
import time

import numpy as np


def one(workload, n_climbs, tile, device, maxtrav, barrier):
    from mpboot_amd import engine, synth, trees
    cfg = synth.WORKLOADS[workload]
    letters, _ = synth.workload(workload)
    codes = synth.letters_to_codes(letters, cfg["alphabet"])
    n, P = codes.shape
    t0 = time.perf_counter()
    engs = []
    for _ in range(n_climbs):
        e = engine.FitchEngine(codes, datatype=engine.DNA, device=device)
        e.set_option("timing", 0)
        if tile:
            e.set_option("climb_tile", tile)
        engs.append(e)
    t_make = time.perf_counter() - t0
    best = None
    for rep in range(2):                              # (first pass: every engine's buffers)
        starts = [trees.random_topology(n, np.random.default_rng(7000 + 97 * k + rep)) for k in range(n_climbs)]
        for k, e in enumerate(engs):
            e.set_tree(starts[k])
            e.reset_node_order()
            e.seed_ties(engine.TIE_RANDOM, k + 1)
            e.reset_stats()
        barrier()
        t0 = time.perf_counter()
        sc = engine.optimize_spr_many(engs, 1, maxtrav)
        barrier()
        dt = time.perf_counter() - t0
        best = dt
    st = [e.stats() for e in engs]
    tw = engs[0].get_option("climb_tile_many")          # (the width the library picked: Engine::climb_fit_vw)
    out = {"workload": "%s: %d taxa x %d DNA patterns, %d climbs from different random topologies, radius %d" % (workload, n, P, n_climbs, maxtrav),
           "climbs": n_climbs, "seconds": best, "climbs_per_s": n_climbs / best, "tile_words": 16 * tw, "tiles_per_climb": (engs[0].Wp + 16 * tw - 1) // (16 * tw),
           "kernel_steps_per_climb": float(np.mean([s["climb_steps"] for s in st])), "sweeps_per_climb": float(np.mean([s["climb_launches"] for s in st])),
           "moves_per_climb": float(np.mean([s["moves_applied"] for s in st])), "score_min": int(sc.min()), "score_max": int(sc.max()),
           "engines_created_s": t_make}
    del engs
    return out


def run(device, maxtrav, barrier, c2_climbs=512, c3_climbs=256):
    leg = {"what": "mpf_optimize_spr_many: one resident workgroup per climb (it works through all tiles of sites itself; nothing crosses between "
                   "workgroups), every sweep of a climb inside the one launch (nodeRectifierPars on the device), one host thread"}
    leg["c2"] = one("C2", c2_climbs, 0, device, maxtrav, barrier)
    if c3_climbs > 0:
        leg["c3"] = one("C3", c3_climbs, 0, device, maxtrav, barrier)
    return leg

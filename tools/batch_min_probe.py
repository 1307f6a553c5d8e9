import sys, time
sys.path.insert(0, '.')
import numpy as np
from mpboot_amd import engine, synth, trees
for wl, seed in (("C3", 2024), ("C2", 1)):
    letters, _ = synth.workload(wl)
    codes = synth.letters_to_codes(letters, "DNA")
    back = trees.random_topology(codes.shape[0], np.random.default_rng(seed))
    for bm in (2, 3, 4, 6):
        e = engine.FitchEngine(codes)
        e.set_option("climb_batch_min", bm)
        tt = []
        for _ in range(3):
            e.set_tree(back); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 1); e.reset_stats()
            t0 = time.perf_counter(); s = e.optimize_spr(1, 6); tt.append(time.perf_counter() - t0)
        st = e.stats()
        print(wl, "batch_min", bm, "score", s, "ms %.1f" % (min(tt) * 1e3), "steps", st["climb_steps"], "kernel ms %.1f" % st["climb_ms_total"], flush=True)

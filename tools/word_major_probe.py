import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from mpboot_amd import engine, synth, trees
for wl in ("C2", "C3"):
    letters, _ = synth.workload(wl)
    codes = synth.letters_to_codes(letters, "DNA")
    n = codes.shape[0]
    back = trees.random_topology(n, np.random.default_rng(2024))
    ref = None
    for opts in ({"climb_tile": 1}, {"climb_tile": 4}, {"climb_tile": 4, "climb_word_major": 1}):
        e = engine.FitchEngine(codes)
        e.set_option("climb_device", 2)
        for k, v in opts.items(): e.set_option(k, v)
        for rep in range(2):
            e.set_tree(back); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 7)
            t0 = time.perf_counter(); s = e.optimize_spr(1, 6); dt = time.perf_counter() - t0
        sig = (s, [x.tolist() for x in e.moves()], e.get_tree().tolist(), e.tie_state())
        ref = ref or sig
        print(wl, opts, f"{dt*1e3:.1f} ms", "same trajectory" if sig == ref else "DIFFERENT", flush=True)

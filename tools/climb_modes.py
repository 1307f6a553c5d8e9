import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from mpboot_amd import engine, synth, trees
letters, names = synth.workload("C3")
codes = synth.letters_to_codes(letters, "DNA")
n = codes.shape[0]
for mode in (0, 1, 2):
    e = engine.FitchEngine(codes)
    e.set_option("climb_device", mode)
    for seed in (2024, 5, 6):
        back = trees.random_topology(n, np.random.default_rng(seed))
        e.score_tree(back); e.set_tree(back); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 1); e.reset_stats()
        t0 = time.perf_counter(); s = e.optimize_spr(1, 6); dt = time.perf_counter() - t0
        st = e.stats()
        print(f"mode {mode} seed {seed}: {dt*1e3:.1f} ms score {s} moves {st['moves_applied']} dev moves {st['climb_moves']} launches {st['climb_launches']} steps {st['climb_steps']} scan_launches {st['scan_launches']}", flush=True)

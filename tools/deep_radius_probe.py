import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpboot_amd import engine, synth, trees
letters, _ = synth.workload("C3")
codes = synth.letters_to_codes(letters, "DNA")
P = codes.shape[1]
samples = np.random.default_rng(1).multinomial(P, np.ones(P) / P, size=64).astype(np.uint16)
back = trees.random_topology(1000, np.random.default_rng(3))
res = []
for opts in ({"ufb_pipe": 0, "ufb_fast": 0}, {}):
    e = engine.FitchEngine(codes)
    for k, v in opts.items(): e.set_option(k, v)
    e.set_option("max_visits", 300)
    e.set_tree(back); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 5)
    e.ufboot_attach(samples, 0.5)
    t0 = time.perf_counter(); s = e.optimize_spr(1, 11); dt = time.perf_counter() - t0
    logl, cnt, tr = e.ufboot_state()
    res.append((s, [m.tolist() for m in e.moves()], logl.tolist(), cnt.tolist(), len(e.ufboot_tree_logl()), e.tie_state()))
    print(opts, "score", s, "moves", len(e.moves()[0]), "saved", len(e.ufboot_tree_logl()), "%.2f s" % dt, flush=True)
print("equal:", res[0] == res[1])
# plain deep scans at full size: a whole sweep at radius 14 (k_scan_deep) twice, chunked differently
e = engine.FitchEngine(codes)
e.set_tree(back)
t0 = time.perf_counter(); a = e.sweep_scan(1, 14); t1 = time.perf_counter() - t0
e.set_option("deep_scratch_kwords", 4096)
e.set_tree(back)
t0 = time.perf_counter(); b = e.sweep_scan(1, 14); t2 = time.perf_counter() - t0
print("radius-14 sweep:", a, "%.3f s" % t1, "| small scratch:", b, "%.3f s" % t2, "equal:", tuple(a) == tuple(b))

import sys, os, time, threading
import numpy as np
sys.path.insert(0, "/root/repo")
from mpboot_amd import engine, synth
letters, names = synth.workload("C3")
codes = synth.letters_to_codes(letters, "DNA")
K = 8; ntrees = 48
pool = []
for k in range(K):
    e = engine.FitchEngine(codes); e.seed_ties(engine.TIE_RANDOM, 1); e.make_parsimony_tree(1, 6); pool.append(e)
for mode in ("grow+climb", "grow only"):
    for e in pool: e.reset_stats()
    nxt = iter(range(ntrees)); lock = threading.Lock()
    def work(k):
        while True:
            with lock:
                u = next(nxt, None)
            if u is None: return
            e = pool[k]; e.seed_ties(engine.TIE_RANDOM, 100 + u)
            if mode == "grow only": e.stepwise_addition(5000 + u)
            else: e.make_parsimony_tree(5000 + u, 6)
    th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    st = [e.stats() for e in pool]
    print(mode, f"{ntrees} trees in {dt:.3f} s = {ntrees/dt:.1f}/s; climb ms total {sum(s['climb_ms_total'] for s in st):.0f}, climb steps {sum(s['climb_steps'] for s in st)}, moves {sum(s['moves_applied'] for s in st)}, launches {sum(s['climb_launches'] for s in st)}", flush=True)
# the climbs alone, as one launch: trees from stepwise addition on the pool, then 48 engines
engs = [engine.FitchEngine(codes) for _ in range(ntrees)]
for u, e in enumerate(engs):
    p = pool[u % K]; p.seed_ties(engine.TIE_RANDOM, 100 + u); p.stepwise_addition(5000 + u)
    e.set_tree(p.get_tree()); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 100 + u)
for rep in range(2):
    for u, e in enumerate(engs):
        p = pool[u % K]; p.seed_ties(engine.TIE_RANDOM, 100 + u); p.stepwise_addition(5000 + u)
        e.set_tree(p.get_tree()); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 100 + u); e.reset_stats()
    t0 = time.perf_counter()
    sc = engine.optimize_spr_many(engs, 1, 6)
    dt = time.perf_counter() - t0
    print(f"the {ntrees} climbs in one launch: {dt:.3f} s; steps per climb {np.mean([e.stats()['climb_steps'] for e in engs]):.0f}, moves {np.mean([e.stats()['moves_applied'] for e in engs]):.0f}", flush=True)

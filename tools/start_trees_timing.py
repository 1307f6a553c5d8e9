#!/usr/bin/env python3
"""Start trees per second: K engines on K host threads, each building randomized-stepwise-addition trees + SPR climb
(_pllComputeRandomizedStepwiseAdditionParsimonyTree, sprDist 6).   python tools/start_trees_timing.py [workload] [trees] [K ...] [opt=value ...]"""
import sys, os, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpboot_amd import engine, synth
args = [a for a in sys.argv[1:] if "=" not in a]
opts = [a for a in sys.argv[1:] if "=" in a]
wl = args[0] if args else "C3"
ntrees = int(args[1]) if len(args) > 1 else 48
Ks = [int(x) for x in args[2:]] or [1, 4, 8, 12]
cfg = synth.WORKLOADS[wl]
letters, names = synth.workload(wl)
codes = synth.letters_to_codes(letters, cfg["alphabet"])
dt = engine.DNA if cfg["alphabet"] == "DNA" else engine.AA
pool = []
def mk():
    e = engine.FitchEngine(codes, datatype=dt)
    for kv in opts:
        k, v = kv.split("="); e.set_option(k, int(v))
    e.seed_ties(engine.TIE_RANDOM, 1); e.make_parsimony_tree(1, 6)      # allocations
    return e
for K in Ks:
    while len(pool) < K:
        pool.append(mk())
    scores = [None] * ntrees
    nxt = iter(range(ntrees))
    lock = threading.Lock()
    def work(k):
        while True:
            with lock:
                u = next(nxt, None)
            if u is None:
                return
            e = pool[k]
            e.seed_ties(engine.TIE_RANDOM, 100 + u)
            scores[u] = e.make_parsimony_tree(5000 + u, 6)
    th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dtm = time.perf_counter() - t0
    print(f"{wl} {' '.join(opts)}: {K} engines, {ntrees} trees in {dtm:.3f} s = {ntrees / dtm:.1f} trees/s (score sum {sum(scores)})", flush=True)

"""Feasibility of MANY iteration-parallel chains on one GPU: (1) a climb from a 498-NNI-perturbed optimum as ONE workgroup (k_climb_many), W
of them in one launch; (2) device memory per chain (engine + attached tracker, B = 1000).   python tools/many_chains_probe.py [W]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpboot_amd import bootstrap, engine, synth
from mpboot_amd.rng import Lcg64
W = int(sys.argv[1]) if len(sys.argv) > 1 else 128
letters, _ = synth.workload("C3")
codes = synth.letters_to_codes(letters, "DNA")
n, P = codes.shape
free0 = torch.cuda.mem_get_info()[0]
e0 = engine.FitchEngine(codes)
e0.seed_ties(engine.TIE_RANDOM, 1)
e0.make_parsimony_tree(1, 6)
best = e0.get_tree()
free1 = torch.cuda.mem_get_info()[0]
w = np.ones(P, dtype=np.int32)
samples = np.stack([bootstrap.bootstrap_weights(w, Lcg64(100 + b)) for b in range(1000)]).astype(np.uint16)
e0.ufboot_attach(samples, 0.5)
e0.seed_ties(engine.TIE_RANDOM, 3)
e0.optimize_spr(1, 6)
torch.cuda.synchronize()
free2 = torch.cuda.mem_get_info()[0]
print(f"device memory: engine {(free0 - free1) / 2**20:.0f} MiB, + tracker after one tracked climb {(free1 - free2) / 2**20:.0f} MiB", flush=True)
e0.ufboot_detach()
engs = [engine.FitchEngine(codes) for _ in range(W)]
for rep in range(2):
    for k, e in enumerate(engs):
        b, st, _ = engine.iq_random_nnis(best, (n - 3) // 2, 1000 + 17 * k + rep)
        e.set_tree(b); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 5 + k); e.reset_stats()
    t0 = time.perf_counter()
    sc = engine.optimize_spr_many(engs, 1, 6)
    dt = time.perf_counter() - t0
    st = [e.stats() for e in engs]
    print(f"{W} climbs from 498-NNI-perturbed trees in one launch: {dt:.3f} s = {W / dt:.0f} climbs/s; per climb {np.mean([s['climb_steps'] for s in st]):.0f} steps, "
          f"{np.mean([s['moves_applied'] for s in st]):.0f} moves, lengths {int(sc.min())}..{int(sc.max())}", flush=True)
# the same climb alone on a workgroup per tile, for reference
e = engs[0]
b, st_, _ = engine.iq_random_nnis(best, (n - 3) // 2, 1000)
for rep in range(2):
    e.set_tree(b); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 5); e.reset_stats()
    t0 = time.perf_counter(); s = e.optimize_spr(1, 6); dt = time.perf_counter() - t0
print(f"one such climb alone (k_climb, a workgroup per tile): {dt * 1e3:.1f} ms, length {s}", flush=True)

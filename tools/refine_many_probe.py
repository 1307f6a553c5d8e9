"""The refinement of a noisy -bb run (C4N: 951 of 1000 samples really climb) on host threads with an engine each against the same climbs
as workgroups of one launch (many_launch).   python tools/refine_many_probe.py [engines_threads] [engines_many]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpboot_amd import bootstrap, engine, synth
from mpboot_amd.rng import Lcg64
kt = int(sys.argv[1]) if len(sys.argv) > 1 else 6
km = int(sys.argv[2]) if len(sys.argv) > 2 else 128
letters, _ = synth.workload("C4N")
codes = synth.letters_to_codes(letters, "DNA")
n, P = codes.shape
pool = [engine.FitchEngine(codes) for _ in range(max(kt, km))]
eng = pool[0]
w = np.ones(P, dtype=np.int32)
samples = np.stack([bootstrap.bootstrap_weights(w, Lcg64(100 + b)) for b in range(1000)]).astype(np.uint16)
starts = []
for k in range(12):
    eng.seed_ties(engine.TIE_RANDOM, 1 + k)
    s = eng.make_parsimony_tree(1 + (k + 1) * 12345, 6)
    starts.append((eng.get_tree(), int(s)))
ref = None
for name, engs, kw in (("host threads", pool[:kt], {}), ("one launch", pool[:km], {"many_launch": True}), ("host threads", pool[:kt], {}), ("one launch", pool[:km], {"many_launch": True})):
    r = bootstrap.bb_run(eng, samples, starts, 20, 6, 1, engines=engs, refine=True, refine_kw=kw)
    if ref is None:
        ref = r["refined_scores"]
    print(f"{name} ({len(engs)} engines): refinement {r['refine_s']:.3f} s, {r['samples_improved_by_refinement']} samples improved, same scores {bool((r['refined_scores'] == ref).all())}", flush=True)

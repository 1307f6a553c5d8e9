import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from mpboot_amd import bootstrap, engine, synth
from mpboot_amd.rng import Lcg64
for (n, P, r, seed) in [(1000, 1500, 0.1, 17), (1000, 2000, 0.3, 17), (1000, 1000, 0.05, 17), (500, 800, 0.2, 17)]:
    letters, _ = synth.synth_alignment(n, P, "DNA", r, seed)
    codes = synth.letters_to_codes(letters, "DNA")
    eng = engine.FitchEngine(codes)
    w = np.ones(P, dtype=np.int32)
    samples = np.stack([bootstrap.bootstrap_weights(w, Lcg64(100 + b)) for b in range(1000)]).astype(np.uint16)
    starts = []
    for k in range(12):
        eng.seed_ties(engine.TIE_RANDOM, 1 + k)
        s = eng.make_parsimony_tree(1 + (k + 1) * 12345, 6)
        starts.append((eng.get_tree(), int(s[0] if isinstance(s, tuple) else s)))
    t0 = time.perf_counter()
    rr = bootstrap.bb_run(eng, samples, starts, 20, 6, 1, refine=True)
    its = np.array([x["seconds"] for x in rr["log"]])
    print(n, P, r, "starts", sorted(x[1] for x in starts)[:6], "best", rr["best_score"], "distinct boot trees", rr["distinct_boot_trees"],
          "improved by refinement", rr["samples_improved_by_refinement"], "refine_s %.3f" % rr["refine_s"], "it ms %.1f" % (its.mean() * 1e3), flush=True)
    del eng

"""bench.py leg `noisy_bootstrap`: the -bb flow on an alignment whose bootstrap samples DISAGREE (synth C4N: 1000 taxa x 2 000 DNA
patterns, 30 % substitutions per branch).  On C3 every sample keeps the ONE optimal tree and the refinement is a single masked sweep;
here the samples keep a couple of hundred distinct trees and nearly every refinement climbs -- the cost of optimizeBootTrees
(iqtree.cpp:2797-2862) per distinct topology and per real climb."""
import time

import numpy as np


def run(device, maxtrav, n_start, iters, n_engines, barrier, B=1000):
    from mpboot_amd import bootstrap, engine, synth
    from mpboot_amd.rng import Lcg64
    cfg = synth.WORKLOADS["C4N"]
    letters, _names = synth.workload("C4N")
    codes = synth.letters_to_codes(letters, "DNA")
    n, P = codes.shape
    pool = [engine.FitchEngine(codes, datatype=engine.DNA, device=device) for _ in range(max(1, n_engines))]
    eng = pool[0]
    w = np.ones(P, dtype=np.int32)
    samples = np.stack([bootstrap.bootstrap_weights(w, Lcg64(100 + b)) for b in range(B)]).astype(np.uint16)
    eng.seed_ties(engine.TIE_RANDOM, 1)
    eng.make_parsimony_tree(1, maxtrav)                                    # allocations
    barrier()
    t0 = time.perf_counter()
    starts = []
    for k in range(n_start):
        eng.seed_ties(engine.TIE_RANDOM, 1 + k)
        s = eng.make_parsimony_tree(1 + (k + 1) * 12345, maxtrav)
        starts.append((eng.get_tree(), int(s[0] if isinstance(s, tuple) else s)))
    barrier()
    t_start = time.perf_counter() - t0
    bootstrap.bb_run(eng, samples, starts, 2, maxtrav, 1, refine=False)  # allocations
    barrier()
    r = bootstrap.bb_run(eng, samples, starts, iters, maxtrav, 1, engines=pool, refine=False)
    its = np.array([x["seconds"] for x in r["log"]])
    rat = np.array([bool(x["ratchet"]) for x in r["log"]])
    # the samples' trees as the online phase left them (bb_run detached the tracker: take them from a second, identical run's books)
    eng.ufboot_attach(samples, 0.5)
    eng.ufboot_detach()
    rr = bootstrap.bb_run(eng, samples, starts, iters, maxtrav, 1, engines=pool, refine=True)
    assert rr["state_hash"] == r["state_hash"]                            # (deterministic: the same run)
    leg = {"workload": "C4N: %d taxa x %d DNA patterns, r = %.2f per branch (synth seed %d)" % (n, P, cfg["r"], cfg["seed"]),
           "samples": B, "start_trees": n_start, "start_trees_s": t_start, "start_lengths_distinct": len({s for _t, s in starts}),
           "iterations": int(len(its)), "iterations_s": float(its.sum()),
           "iteration_ms_nni": float(its[~rat].mean() * 1e3) if (~rat).any() else None,
           "iteration_ms_ratchet": float(its[rat].mean() * 1e3) if rat.any() else None,
           "trees_booked": int(r["saved_trees"]), "best_length": r["best_score"], "best_start_length": r["start_best_score"],
           "distinct_boot_trees": rr["distinct_boot_trees"], "refinement_s": rr["refine_s"], "refinement_engines": len(pool),
           "samples_improved_by_refinement": rr["samples_improved_by_refinement"],
           "mean_sample_length_online": float(np.mean(rr["online_scores"])), "mean_sample_length_refined": rr["mean_refined"],
           "seconds": t_start + float(its.sum()) + rr["refine_s"],
           "what": "the -bb flow of bb_reference_run (one chain) where the samples disagree: start trees + doTreeSearch iterations + "
                   "refinement.  refinement_s = refine_boot_trees: one masked sweep + one mask x weight product per DISTINCT topology the "
                   "samples kept, then one re-weighting + SPR climb per sample whose first sweep accepts a move (several engines per GPU on host "
                   "threads; as workgroups of one launch -- mpf_optimize_spr_many_round, 128 engines -- the same climbs take 1.0 s: they start "
                   "next to an optimum, where the host path's whole-chip batches are the better tool)"}
    return leg

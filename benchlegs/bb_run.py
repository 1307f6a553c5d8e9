"""bench.py leg `bb_reference_run`: `-bb 1000` as the reference runs it (BASELINE config 4).

  start trees      the start-up phase's trees are the candidate set (phyloanalysis.cpp:1261-1317) -- timed by the start_trees leg
  iterations       IQTree::doTreeSearch (iqtree.cpp:1631-1965; mpboot_amd/search.py): a random one of the 5 best candidate trees,
                   floor(0.5 (n - 3)) random NNIs (498 at C3), every second iteration the ratchet's two climbs instead, logl_cutoff =
                   top 10 % of the saved trees, saveCurrentTree behind every insertion test; the run ends after
                   ((n - 1) / 100 + 1) * 100 iterations without a better tree (iqtree.cpp:129-130: 1000 at C3)
  refinement       optimizeBootTrees: one climb per sample from its tree (batched first sweep, mpboot_amd/bootstrap.py)

`sequential` is ONE chain on one engine -- the reference's flow draw for draw (parity: tests/test_gpu_bb_iterations.py).
`parallel` runs W chains per GPU (and the ranks' chains beside them) with an exchange every `sync_every` iterations
(mpboot_amd/parsearch.py: one all-reduce of best lengths per round; tests/test_parsearch.py, tests/test_gpu_parsearch.py).
Both time a bounded number of iterations; the stop rule's horizon is reported beside them, extrapolated from the measured rate.
"""
import time

import numpy as np


def _summ(log):
    its = np.array([x["seconds"] for x in log])
    rat = np.array([bool(x["ratchet"]) for x in log])
    out = {"iterations": int(len(its)), "iterations_s": float(its.sum()),
           "iteration_ms_nni": float(its[~rat].mean() * 1e3) if (~rat).any() else None,
           "iteration_ms_ratchet": float(its[rat].mean() * 1e3) if rat.any() else None,
           "iteration_ms_mean": float(its.mean() * 1e3)}
    for key in ("moves", "insertion_tests", "climb_steps", "climb_ms"):
        if all(key in x for x in log):
            v = np.array([x[key] for x in log], dtype=np.float64)
            out[key + "_per_nni_iteration"] = float(v[~rat].mean()) if (~rat).any() else None
            out[key + "_per_ratchet_iteration"] = float(v[rat].mean()) if rat.any() else None
    out["perturbation_ms_nni"] = float(np.mean([x["perturb_s"] for x, r in zip(log, rat) if not r]) * 1e3) if (~rat).any() else None
    out["perturbation_ms_ratchet"] = float(np.mean([x["perturb_s"] for x, r in zip(log, rat) if r]) * 1e3) if rat.any() else None
    return out


def run(pool, samples, starts, maxtrav, rank, world, barrier, iters_seq, workers, rounds_par, sync_every, start_trees_s, refine_engines):
    from mpboot_amd import bootstrap, parsearch, search
    n = pool[0].n
    unsuccess = search.unsuccess_iterations(n)
    leg = {"samples": int(samples.shape[0]), "start_trees": len(starts), "start_trees_s": start_trees_s, "stop_rule_unsuccessful_iterations": unsuccess,
           "nnis_per_perturbation": int(0.5 * (n - 3))}
    # ---- one chain: the reference's flow
    if world == 1 and iters_seq > 0:
        for x in pool[:1]:
            x.set_option("timing", 0)
        bootstrap.bb_run(pool[0], samples, starts, 4, maxtrav, 1, refine=False)                  # allocations
        barrier()
        t0 = time.perf_counter()
        r = bootstrap.bb_run(pool[0], samples, starts, iters_seq, maxtrav, 1, refine=False)
        barrier()
        seq = _summ(r["log"])
        seq.update(seconds_wall=time.perf_counter() - t0, trees_booked=int(r["saved_trees"]), best_length=r["best_score"],
                   best_start_length=r["start_best_score"], distinct_boot_trees=r["distinct_boot_trees"], state_sha16=r["state_hash"],
                   insertion_tests=int(sum(x["insertion_tests"] for x in r["log"])),
                   iterations_to_stop_rule_at_least=int(r["iterations_left_by_stop_rule"] + len(r["log"])))
        per_pair = (seq["iteration_ms_nni"] or 0.0) + (seq["iteration_ms_ratchet"] or seq["iteration_ms_nni"] or 0.0)
        seq["iterations_s_extrapolated_to_stop_rule"] = per_pair * 0.5e-3 * seq["iterations_to_stop_rule_at_least"]
        leg["sequential"] = seq
    # ---- W chains per GPU, exchanges every sync_every iterations
    par = None
    if workers > 0 and rounds_par > 0:
        W = min(workers, len(pool))
        # (chains side by side: 64-word tiles -- 25 workgroups per climb instead of 98, 2.7 x less CU-time held per climb; measured on one
        #  box: 8 / 16 / 24 chains 87 / 85 / 90 iterations/s on 16-word tiles, 92 / 108 / 113 on these.  Same trajectories on any width)
        for x in pool[:W]:
            x.set_option("climb_tile", 4)
        run_ = parsearch.ParallelBbRun(pool[:W], samples, starts, maxtrav=maxtrav, seed=1, sync_every=sync_every)
        run_.round(2)                                                                             # allocations
        barrier()
        t0 = time.perf_counter()
        infos = [run_.round() for _ in range(rounds_par)]
        barrier()
        t_par = time.perf_counter() - t0
        log = [x for i in infos for w in i["per_worker"] for x in w]
        n_it = sum(i["iterations"] for i in infos)
        par = _summ(log)
        par.update(workers_per_gpu=W, n_gpus=world, sync_every=sync_every, rounds=rounds_par, iterations=int(n_it), seconds_wall=t_par,
                   iterations_per_s=n_it / t_par, sync_ms_mean=float(np.mean([i["sync_s"] for i in infos]) * 1e3),
                   adopted_sample_trees=int(sum(i["adopted"] for i in infos)), shipped_trees=int(sum(i["shipped_trees"] for i in infos)),
                   best_length=int(-run_.best_score), state_sha16=run_.state_hash(),
                   iterations_to_stop_rule_at_least=int(max(unsuccess, run_.last_improved_at + unsuccess)))
        par["iterations_s_extrapolated_to_stop_rule"] = par["iterations_to_stop_rule_at_least"] / par["iterations_per_s"]
        lens, bts, n_distinct = run_.books()
        par["distinct_boot_trees"] = int(n_distinct)
        run_.detach()
        for x in pool[:W]:
            x.set_option("climb_tile", 1)
        # refinement of every sample's tree (sharded by sample over the ranks)
        barrier()
        t0 = time.perf_counter()
        sc, _ = bootstrap.refine_boot_trees(refine_engines, samples, bts, 11, maxtrav)
        barrier()
        par["refinement_s"] = time.perf_counter() - t0
        par["samples_improved_by_refinement"] = int((np.asarray(sc) < lens).sum())
        leg["parallel"] = par
    refine_s = par["refinement_s"] if par else None
    if "sequential" in leg and refine_s is not None:
        s = leg["sequential"]
        leg["seconds_measured_one_chain"] = start_trees_s + s["iterations_s"] + refine_s
        leg["seconds_extrapolated_to_stop_rule_one_chain"] = start_trees_s + s["iterations_s_extrapolated_to_stop_rule"] + refine_s
    if par:
        leg["seconds_extrapolated_to_stop_rule_parallel"] = start_trees_s + par["iterations_s_extrapolated_to_stop_rule"] + par["refinement_s"]
    leg["what"] = ("-bb %d as the reference runs it: %d start trees (start_trees leg) -> candidate set -> doTreeSearch iterations (random NNIs / "
                   "ratchet alternating, tracked climbs under the top-10 %% cut-off) -> refinement.  sequential = ONE chain, the reference's "
                   "flow draw for draw; parallel = W chains per GPU x n_gpus with an exchange of books and candidates every sync_every "
                   "iterations (one all-reduce of best lengths per round).  A bounded number of iterations is timed; the stop rule needs at "
                   "least stop_rule_unsuccessful_iterations of them: *_extrapolated_to_stop_rule = start trees + that many iterations at the "
                   "measured rate + refinement" % (samples.shape[0], len(starts)))
    return leg

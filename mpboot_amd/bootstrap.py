"""Bootstrap replicates on the engine: resampling weights + per-replicate SPR climbs, sharded over ranks.

Reference flow: the B replicates of `-bb` are refined one after another by
IQTree::optimizeBootTrees (iqtree.cpp:2475-2915): the alignment is re-weighted with the replicate's pattern
frequencies (modifyPatternFreq :2520), the parsimony structures are rebuilt, and ONE SPR hill climb is run from
the replicate's best tree (:2837).  Standard bootstrap (phyloanalysis.cpp:1945-2085) instead searches every
resampled alignment from scratch.  Both are independent per replicate: replicate b goes to rank b % world.
"""
from __future__ import annotations

import numpy as np

from . import shard
from .rng import Lcg64


def bootstrap_weights(weights: np.ndarray, rng: Lcg64) -> np.ndarray:
    """Alignment::createBootstrapAlignment(int *pattern_freq) (alignment.cpp:1981-1990): nsite draws of
    random_int(nsite), each incrementing the frequency of the drawn site's pattern."""
    weights = np.asarray(weights, dtype=np.int64)
    nsite = int(weights.sum())
    site_pattern = np.repeat(np.arange(len(weights)), weights)        # getPatternID(site) for pattern-sorted sites
    draws = rng.ints(nsite, nsite)
    return np.bincount(site_pattern[draws], minlength=len(weights)).astype(np.int32)


def _one_replicate(eng, weights, b, base_seed, radius, start_tree, mode):
    seed = shard.unit_seed(base_seed, b)
    w = bootstrap_weights(weights, Lcg64(seed))
    eng.set_weights(w)
    eng.seed_ties(1, seed)
    eng.reset_node_order()          # every replicate starts from a fresh instance state: results do not depend on sharding
    if mode == "refine":
        eng.set_tree(start_tree)
        s = eng.optimize_spr(1, radius)
    else:
        r = eng.make_parsimony_tree(seed, radius)
        s = r[0] if isinstance(r, tuple) else r
    return s, eng.get_tree()


def run_replicates(eng, weights, n_rep: int, base_seed: int, radius: int = 6, start_tree=None, mode: str = "refine"):
    """mode "refine": SPR climb from start_tree on every re-weighted alignment (optimizeBootTrees);
    mode "search": randomized stepwise addition + SPR on every re-weighted alignment (standard bootstrap).
    `eng` may be a list of engines on the same GPU: replicates are then spread over one host thread per engine
    (each engine has its own HIP stream; small alignments are launch-latency-bound, concurrent climbs fill the GPU).
    Returns (scores[n_rep] after the all-reduce, {replicate: tree} of this rank's units)."""
    rank, ws = shard.world()
    units = shard.units_of_rank(n_rep, rank, ws)
    local, trees = {}, {}
    engines = eng if isinstance(eng, (list, tuple)) else [eng]
    if len(engines) == 1:
        for b in units:
            local[b], trees[b] = _one_replicate(engines[0], weights, b, base_seed, radius, start_tree, mode)
    else:
        from concurrent.futures import ThreadPoolExecutor

        def work(k):
            out = {}
            for b in units[k::len(engines)]:
                out[b] = _one_replicate(engines[k], weights, b, base_seed, radius, start_tree, mode)
            return out

        with ThreadPoolExecutor(len(engines)) as ex:
            for part in ex.map(work, range(len(engines))):
                for b, (s, t) in part.items():
                    local[b], trees[b] = s, t
    scores, _best, _owner = shard.reduce_best(local, n_rep)
    return scores, trees


def refine_boot_trees(eng, samples, boot_trees, base_seed: int, radius: int = 6, batched=None, attached: bool = False, many_launch: bool = False):
    """The refinement step of UFBoot-MP, IQTree::optimizeBootTrees default branch (iqtree.cpp:2797-2862): for every
    bootstrap sample b the alignment is re-weighted with boot_samples_pars[b] (modifyPatternFreq :2520), the sample's
    tree from the online phase (boot_trees[b]) is read back and ONE SPR hill climb is run from it (:2837); the result
    replaces boot_trees[b] / boot_logl[b] (:2860-2861).

    samples: [B][P] weights; boot_trees: [B][nrec] topologies (mpf_ufboot_get_tree per sample).  Sample b is refined on
    rank b % world (and on engine k of that rank's list); every replicate draws its ties from its own stream so the
    result does not depend on the sharding.

    batched (default: whenever the engine offers it): the samples of this rank are grouped by start topology and the FIRST sweep
    of every group's climbs is computed at once on engine 0 (mpf_ufboot_refine_sweep: one masked scan + one mask x weight product
    per topology); a sample whose sweep accepts no move is done -- its climb would return the tree unchanged --, the others run
    their climb alone as before.  Same results either way.  Engine 0 must hold the weights the samples were drawn from.
    attached = True: the tracker of the online phase is still attached to engine 0 with exactly these samples and this sharding
    (no second upload of the weights); it is released before any per-sample climb runs on engine 0.

    many_launch = True (eight or more engines): the samples' own climbs as workgroups of ONE launch, an engine per sample,
    a finished engine taking the next sample at once (mpf_optimize_spr_many_round).  Same results; NOT the default: these climbs start
    next to an optimum -- two or three sweeps with a handful of moves --, and a move-less sweep of a thousand taxa costs one resident
    workgroup milliseconds where the host path's whole-chip batch takes 0.3 ms (C4N, 955 climbs: 1.0 s on 128 engines, 0.76-0.81 s on 256,
    against 0.77-0.81 s on six host threads; tools/refine_many_probe.py).  The one-launch form is for DENSE climbs (random start trees: C2 3 000 climbs/s against 345).

    Returns (scores[B] after the all-reduce, {b: refined tree} of this rank)."""
    samples = np.asarray(samples)
    B = samples.shape[0]
    rank, ws = shard.world()
    units = shard.units_of_rank(B, rank, ws)
    engines = eng if isinstance(eng, (list, tuple)) else [eng]
    if batched is None:
        # (both engines: Fitch state sets and the weighted engine's per-pattern cost vectors alike do not depend on the pattern
        #  weights -- those enter in the product with the samples)
        batched = hasattr(engines[0], "ufboot_refine_sweep")

    def one(e, b):
        e.set_weights(samples[b].astype(np.int32))
        e.seed_ties(1, shard.unit_seed(base_seed, b))
        e.reset_node_order()
        e.set_tree(np.asarray(boot_trees[b], dtype=np.int32))
        return e.optimize_spr(1, radius), e.get_tree()

    local, trees = {}, {}
    todo = units
    # every engine is handed back under the weights it came with: one() leaves an engine on the last sample it climbed for, and
    # the batched block multiplies the samples against the packing in force on engine 0 -- a pattern at weight 0 there would
    # silently drop out of every sample's product
    held = [e.weights() if hasattr(e, "weights") else None for e in engines]
    touched = set()
    try:
        if batched and units:
            e0 = engines[0]
            if held[0] is not None and ((held[0] == 0) & (samples[units].max(axis=0) > 0)).any():
                raise ValueError("refine_boot_trees: engine 0 holds weights that leave a pattern some sample counts without a site -- "
                                 "it must hold the weights the samples were drawn from (set_weights(original) first)")
            if not attached:
                e0.ufboot_attach(samples, 0.5, shard=(rank, ws))
            e0.seed_ties(1, 0)
            seeds = np.array([shard.unit_seed(base_seed, b) for b in range(B)], dtype=np.int64).astype(np.int32)
            groups = {}
            for b in units:
                groups.setdefault(np.asarray(boot_trees[b], dtype=np.int32).tobytes(), []).append(b)
            todo = []
            for key, members in groups.items():
                t = np.frombuffer(key, dtype=np.int32).copy()
                e0.reset_node_order()
                e0.set_tree(t)
                sc, stable, _first = e0.ufboot_refine_sweep(radius, seeds)
                for b in members:
                    if stable[b]:
                        local[b], trees[b] = int(sc[b]), t
                    else:
                        todo.append(b)
            if not attached or todo:
                e0.ufboot_detach()                       # (a climb under an attached tracker would be booked like a search iteration)
            todo.sort()
        many = len(engines) >= 8 and all(hasattr(e, "h") for e in engines) and many_launch
        if many and todo:
            # the samples' own climbs side by side: an engine per sample of a chunk (re-weighted, seeded, its tree set), then ONE call --
            # every climb a resident workgroup of one launch (mpf_optimize_spr_many_round)
            from concurrent.futures import ThreadPoolExecutor

            from . import engine as _engine

            def setup(kb):
                k, b = kb
                e = engines[k]
                e.set_weights(samples[b].astype(np.int32))           # (re-pack on the device + a wait: side by side on a few host threads)
                e.seed_ties(1, shard.unit_seed(base_seed, b))
                e.reset_node_order()
                e.set_tree(np.asarray(boot_trees[b], dtype=np.int32))

            # a finished engine takes the next sample at once: the launches stay full until the samples run out
            batch = _engine.ClimbBatch(engines, 1, radius)
            owner = {}
            queue = list(todo)
            free = list(range(len(engines)))
            with ThreadPoolExecutor(min(16, len(engines))) as ex:
                while queue or batch.active():
                    fill = []
                    while queue and free:
                        k = free.pop()
                        b = queue.pop(0)
                        owner[k] = b
                        fill.append((k, b))
                    touched.update(k for k, _b in fill)
                    list(ex.map(setup, fill))
                    for k, _b in fill:
                        batch.start(k)
                    for k in batch.round():
                        b = owner.pop(k)
                        local[b], trees[b] = int(batch.scores[k]), engines[k].get_tree()
                        free.append(k)
        elif len(engines) == 1:
            for b in todo:
                touched.add(0)
                local[b], trees[b] = one(engines[0], b)
        elif todo:
            from concurrent.futures import ThreadPoolExecutor

            def work(k):
                if todo[k::len(engines)]:
                    touched.add(k)
                return {b: one(engines[k], b) for b in todo[k::len(engines)]}

            with ThreadPoolExecutor(len(engines)) as ex:
                for part in ex.map(work, range(len(engines))):
                    for b, (sc, t) in part.items():
                        local[b], trees[b] = sc, t
    finally:
        # (also when a climb raises: an engine left on some sample's weights is the silent-drop state the guard above looks for)
        for k in sorted(touched):
            if held[k] is not None:
                engines[k].set_weights(held[k])
    scores, _best, _owner = shard.reduce_best(local, B)
    return scores, trees


def bb_run(eng, samples, start_trees, iters, maxtrav=6, seed=1, engines=None, verbose=False, refine=True, search_kw=None, refine_kw=None):
    """`-bb` as the reference runs it: the start trees enter the candidate set (phyloanalysis.cpp:1300-1313; they are
    not booked -- the start-tree phase runs without per-site scores, sprparsimony.cpp:3228), then IQTree::doTreeSearch's iterations
    (mpboot_amd.search.MpSearch: a random one of the 5 best candidate trees perturbed by floor(0.5 (n - 3)) random NNIs, every second
    iteration the ratchet's two climbs instead, cut-off = top 10 % of the saved trees) with saveCurrentTree behind every insertion test,
    then the refinement of every sample's tree (optimizeBootTrees).

    start_trees: [(back, length)].  iters: iterations to run (the reference runs until unsuccess_iterations(n) in a row bring no better
    tree, iqtree.cpp:129-130: the caller extrapolates).  -> dict; per-iteration dicts in ["log"].  Deterministic for a given seed."""
    import hashlib
    import time

    from . import engine, search
    eng.ufboot_attach(samples, 0.5)
    eng.seed_ties(engine.TIE_RANDOM, seed)
    S = search.MpSearch(eng, maxtrav=maxtrav, tracked=True, **(search_kw or {}))
    for t, length in start_trees:
        S.add_candidate(t, length)
    start_best = -S.best_score
    log = []
    t_all = time.perf_counter()
    for _ in range(iters):
        eng.reset_stats()
        c0 = eng.ufboot_counters()
        n0 = len(eng.ufboot_tree_logl()) if verbose else 0
        info = S.iterate()
        st = eng.stats()
        c1 = eng.ufboot_counters()
        info.update(moves=st["moves_applied"], insertion_tests=st["insertion_tests"], events=c1["events"] - c0["events"],
                    tie_draws=c1["tie_draws"] - c0["tie_draws"], climb_launches=st["climb_launches"], climb_steps=st["climb_steps"],
                    climb_nodes=st["climb_nodes"], climb_ms=st["climb_ms_total"], scan_launches=st["scan_launches"])
        log.append(info)
        if verbose:
            print(f"  it {info['iteration']} {'ratchet' if info['ratchet'] else 'nni    '} -> {info['score']} in {info['seconds'] * 1e3:.1f} ms "
                  f"(perturb {info['perturb_s'] * 1e3:.2f}, after {info['after_s'] * 1e3:.2f}; {info['moves']} moves, {info['insertion_tests']} tests, "
                  f"{info['events']} events; k_climb {info['climb_launches']} launches {info['climb_steps']} steps {info['climb_ms']:.1f} ms; "
                  f"scan launches {info['scan_launches']}; booked +{len(eng.ufboot_tree_logl()) - n0})", flush=True)
    t_iters = time.perf_counter() - t_all
    logl, counts, bt = eng.ufboot_state()
    n_saved = len(eng.ufboot_tree_logl())
    cache, bts = {}, []
    for b in range(samples.shape[0]):
        t = int(bt[b])
        if t not in cache:
            cache[t] = eng.ufboot_tree(t) if t >= 0 else S.best_tree
        bts.append(cache[t])
    state_hash = hashlib.sha256(logl.tobytes() + counts.tobytes() + bt.tobytes() + str(eng.tie_state()).encode()).hexdigest()[:16]
    eng.ufboot_detach()
    out = {"iterations_s": t_iters, "log": log, "best_score": int(-S.best_score), "start_best_score": int(start_best), "saved_trees": n_saved,
           "distinct_boot_trees": len(cache), "state_hash": state_hash, "iterations_left_by_stop_rule": S.iterations_left(),
           "unsuccess_iterations": S.unsuccess, "last_improved_iteration": S.last_improved, "online_scores": (-logl).astype(np.int64)}
    if refine:
        t0 = time.perf_counter()
        sc, _ = refine_boot_trees(engines or [eng], samples, bts, 11, maxtrav, **(refine_kw or {}))
        out["refined_scores"] = np.asarray(sc)
        out["refine_s"] = time.perf_counter() - t0
        out["mean_refined"] = float(np.mean(sc))
        out["samples_improved_by_refinement"] = int((np.asarray(sc) < out["online_scores"]).sum())
    return out

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


def refine_boot_trees(eng, samples, boot_trees, base_seed: int, radius: int = 6):
    """The refinement step of UFBoot-MP, IQTree::optimizeBootTrees default branch (iqtree.cpp:2797-2862): for every
    bootstrap sample b the alignment is re-weighted with boot_samples_pars[b] (modifyPatternFreq :2520), the sample's
    tree from the online phase (boot_trees[b]) is read back and ONE SPR hill climb is run from it (:2837); the result
    replaces boot_trees[b] / boot_logl[b] (:2860-2861).

    samples: [B][P] weights; boot_trees: [B][nrec] topologies (mpf_ufboot_get_tree per sample).  Sample b is refined on
    rank b % world (and on engine k of that rank's list); every replicate draws its ties from its own stream so the
    result does not depend on the sharding.  Returns (scores[B] after the all-reduce, {b: refined tree} of this rank)."""
    samples = np.asarray(samples)
    B = samples.shape[0]
    rank, ws = shard.world()
    units = shard.units_of_rank(B, rank, ws)
    engines = eng if isinstance(eng, (list, tuple)) else [eng]

    def one(e, b):
        e.set_weights(samples[b].astype(np.int32))
        e.seed_ties(1, shard.unit_seed(base_seed, b))
        e.reset_node_order()
        e.set_tree(np.asarray(boot_trees[b], dtype=np.int32))
        return e.optimize_spr(1, radius), e.get_tree()

    local, trees = {}, {}
    if len(engines) == 1:
        for b in units:
            local[b], trees[b] = one(engines[0], b)
    else:
        from concurrent.futures import ThreadPoolExecutor

        def work(k):
            return {b: one(engines[k], b) for b in units[k::len(engines)]}

        with ThreadPoolExecutor(len(engines)) as ex:
            for part in ex.map(work, range(len(engines))):
                for b, (sc, t) in part.items():
                    local[b], trees[b] = sc, t
    scores, _best, _owner = shard.reduce_best(local, B)
    return scores, trees

"""Iteration-parallel `-bb`: several chains of IQTree::doTreeSearch side by side, meeting every few iterations.

The search chain of one mpboot process is sequential (perturb -> climb -> accept, iqtree.cpp:1631-1965) and at a thousand taxa it
runs for at least 1000 iterations (iqtree.cpp:129-130).  The reference's own parallel form distributes the ITERATIONS over
processes that exchange their results from time to time (README.md:71-78, branches mpboot-mpi-sync / -async -- not in the tree
under /root/reference; the rule for merging a better tree into a sample's books is saveCurrentTree's, iqtree.cpp:3686-3730).
Here a *worker* is one engine with its own UFBoot tracker (all B samples), its own candidate set and its own random_double()
stream (seed + worker * 12345, as the reference seeds independent units, phyloanalysis.cpp:1273); there are W workers per
process (host threads on one GPU: a climb keeps a quarter of the chip busy) and one process per GPU.  A *round* = `sync_every`
iterations on every worker, then ONE exchange:

  books       per sample the shortest REPS length any worker holds and who holds it: one all-reduce(MIN) of B int64 words
              (length << 16 | worker) -- the "single all-reduce of best scores per round" --, then the topologies that changed
              hands (all-gather of the few trees whose (sample, owner) entry is new); every worker adopts what is strictly
              shorter than its own (mpf_ufboot_adopt: the strict branch of saveCurrentTree's rule; equal lengths keep the holder)
  candidates  every worker's results of the round enter every worker's candidate set (CandidateSet::update) in worker order
  stop rule   iterations count over all workers; a round in which some worker found a better tree than the run's best resets
              the count of unsuccessful iterations (stoprule.cpp:92-93)

Deterministic for a given (seed, world, W, sync_every).  With one worker and one rank it is the sequential run (bootstrap.bb_run).
The trajectories are NOT those of the one-stream sequential run -- like the reference's own MPI form: same rules, other draws.
"""
from __future__ import annotations

import hashlib
import threading
import time

import numpy as np

from . import engine as _engine
from . import search, shard

_UNSET = np.uint32(0xFFFFFFFF)


def _dist():
    import torch
    import torch.distributed as dist
    return torch, dist


class ParallelBbRun:
    def __init__(self, engines, samples, start_trees, maxtrav: int = 6, seed: int = 1, sync_every: int = 8, search_kw=None, tie_mode=None,
                 worker_base=None):
        self.rank, self.world = shard.world()
        self.engines = list(engines)
        self.W = len(self.engines)
        self.B = int(samples.shape[0])
        self.maxtrav, self.sync_every = maxtrav, sync_every
        self.searches = []
        tie_mode = _engine.TIE_RANDOM if tie_mode is None else tie_mode
        self.g0 = self.rank * self.W if worker_base is None else int(worker_base)      # (worker_base: tests run one chain of a larger run alone)
        for i, e in enumerate(self.engines):
            g = self.g0 + i
            e.ufboot_attach(samples, 0.5)
            e.seed_ties(tie_mode, shard.unit_seed(seed, g))
            s = search.MpSearch(e, maxtrav=maxtrav, tracked=True, **(search_kw or {}))
            for t, length in start_trees:
                s.add_candidate(t, length)
            self.searches.append(s)
        self.n_workers = self.world * self.W
        assert self.n_workers < 65536
        self.iterations = 0                       # over all workers
        self.last_improved_at = 0
        self.best_score = self.searches[0].best_score
        self.unsuccess = self.searches[0].unsuccess
        self.prev_key = np.full(self.B, -1, dtype=np.int64)
        self.rounds = []
        self.errors = []

    # ---- one round
    def _run_worker(self, i, k, out):
        try:
            s = self.searches[i]
            e = self.engines[i]
            res = []
            for _ in range(k):
                if hasattr(e, "reset_stats"):
                    e.reset_stats()
                info = s.iterate()
                if hasattr(e, "stats"):
                    st = e.stats()
                    info.update(moves=st["moves_applied"], insertion_tests=st["insertion_tests"], climb_steps=st["climb_steps"], climb_ms=st["climb_ms_total"])
                res.append(info)
            out[i] = res
        except Exception as exc:                  # (a worker thread must not die silently: the round re-raises)
            self.errors.append(exc)
            out[i] = []

    def round(self, k=None):
        k = self.sync_every if k is None else k
        out = [None] * self.W
        t0 = time.perf_counter()
        if self.W == 1:
            self._run_worker(0, k, out)
        else:
            th = [threading.Thread(target=self._run_worker, args=(i, k, out)) for i in range(self.W)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if self.errors:
            raise self.errors[0]
        t_it = time.perf_counter() - t0
        info = self.sync(out)
        info.update(iterations_s=t_it, sync_s=time.perf_counter() - t0 - t_it, per_worker=out)
        self.rounds.append(info)
        return info

    # ---- the exchange
    def sync(self, results):
        W, B = self.W, self.B
        # books of the local workers: lengths (UINT32_MAX = nothing booked), tree index per sample
        L = np.empty((W, B), dtype=np.int64)
        T = np.empty((W, B), dtype=np.int64)
        for i, e in enumerate(self.engines):
            logl, _cnt, bt = e.ufboot_state()
            ln = -np.asarray(logl, dtype=np.float64)
            L[i] = np.where(np.asarray(bt) >= 0, np.minimum(ln, 2.0 ** 40), 2.0 ** 40).astype(np.int64)
            T[i] = bt
        g0 = self.g0
        key_local = (L << 16) | (g0 + np.arange(W, dtype=np.int64))[:, None]
        key = key_local.min(axis=0)
        if self.world > 1:
            torch, dist = _dist()
            t = torch.from_numpy(key.copy()).to(shard._device())
            dist.all_reduce(t, op=dist.ReduceOp.MIN)          # the one all-reduce of best scores per round
            key = t.cpu().numpy()
        owner = (key & 0xFFFF).astype(np.int64)
        best_len = key >> 16
        # the topologies that changed hands: entries whose (length, owner) is new since the last exchange and whose owner is mine
        changed = (key != self.prev_key) & (best_len < 2 ** 40)
        mine = changed & (owner >= g0) & (owner < g0 + W)
        ship = []                                 # (owner, tree index there) -> samples
        groups = {}
        for b in np.nonzero(mine)[0]:
            i = int(owner[b]) - g0
            groups.setdefault((i, int(T[i][b])), []).append(int(b))
        treels = {}
        for (i, ti), members in sorted(groups.items()):
            e = self.engines[i]
            back = np.asarray(e.ufboot_tree(ti), dtype=np.int32)
            if i not in treels:
                treels[i] = e.ufboot_tree_logl()
            length = int(round(-float(treels[i][ti])))
            ship.append((g0 + i, members, back.tobytes(), length))
        # this round's search results, for everybody's candidate set
        cands = []
        for i, res in enumerate(results):
            for info in res or []:
                cands.append((g0 + i, int(info["iteration"]), int(info["score"]), self.searches[i].log_tree(info)))
        improved_local = max((s.best_score for s in self.searches))
        payload = (ship, cands, float(improved_local))
        if self.world > 1:
            torch, dist = _dist()
            box = [None] * self.world
            dist.all_gather_object(box, payload)
        else:
            box = [payload]
        all_ship = [x for p in box for x in p[0]]
        all_cands = sorted((x for p in box for x in p[1]), key=lambda c: (c[0], c[1]))
        # adoption
        trees = [np.frombuffer(x[2], dtype=np.int32) for x in all_ship]
        lengths = [x[3] for x in all_ship]
        tree_of_sample = np.full(B, -1, dtype=np.int64)
        for k, x in enumerate(all_ship):
            tree_of_sample[np.asarray(x[1], dtype=np.int64)] = k
        adopted = 0
        for i, e in enumerate(self.engines):
            want = np.nonzero((best_len < L[i]) & (tree_of_sample >= 0))[0]
            # (a strictly better entry that did not change this round was shipped in an earlier one -- and adopted then)
            if len(want):
                adopted += e.ufboot_adopt(want, best_len[want], tree_of_sample[want], trees, lengths)
        self.prev_key = key
        # candidate sets and the run's best tree
        n_iter = len(all_cands)
        # (a tree's topology digest once, not once per chain that takes it: 16 chains x 128 results were 50 ms of digests per exchange)
        from . import engine as _engine
        cand_trees = [np.frombuffer(tree, dtype=np.int32) for _g, _it, _score, tree in all_cands]
        cand_keys = [_engine.iq_topology_key(t) for t in cand_trees]
        for i, s in enumerate(self.searches):
            for (g, _it, score, _tree), t, key in zip(all_cands, cand_trees, cand_keys):
                if g != g0 + i:
                    s.absorb(t, score, key)
        self.iterations += n_iter
        best_now = max(p[2] for p in box)
        improved = best_now > self.best_score
        if improved:
            self.best_score = best_now
            self.last_improved_at = self.iterations
        for s in self.searches:                   # every chain sees the run's clock
            s.cur_it = 2 + self.iterations
            s.last_improved = 1 + self.last_improved_at
        return {"iterations": n_iter, "adopted": int(adopted), "shipped_trees": len(all_ship), "improved": bool(improved),
                "best_length": int(-self.best_score), "distinct_owners": int(len(np.unique(owner[best_len < 2 ** 40])))}

    def stop(self) -> bool:
        return 2 + self.iterations > 1 + self.last_improved_at + self.unsuccess

    # ---- results
    def books(self):
        """(lengths[B], trees {sample: back}) of worker 0 of this rank -- after an exchange every worker holds the run's best length
        per sample (the topologies of tied samples may differ between workers)."""
        e = self.engines[0]
        logl, _cnt, bt = e.ufboot_state()
        cache, trees = {}, []
        for b in range(self.B):
            t = int(bt[b])
            if t not in cache:
                cache[t] = e.ufboot_tree(t) if t >= 0 else self.searches[0].best_tree
            trees.append(cache[t])
        return (-np.asarray(logl)).astype(np.int64), trees, len(cache)

    def state_hash(self) -> str:
        h = hashlib.sha256()
        for e in self.engines:
            logl, cnt, _bt = e.ufboot_state()
            h.update(np.asarray(logl).tobytes() + np.asarray(cnt).tobytes() + str(e.tie_state()).encode())
        return h.hexdigest()[:16]

    def detach(self):
        for e in self.engines:
            e.ufboot_detach()

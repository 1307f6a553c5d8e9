"""The search loop around the climb: IQTree::doTreeSearch for maximum parsimony with SPR (iqtree.cpp:1583-1990), host side.

What the reference repeats until its stop rule fires (`-stop_cond`: ((n - 1) / 100 + 1) * 100 iterations without a better tree,
iqtree.cpp:129-130, stoprule.cpp:92-93 -- 1000 at a thousand taxa):

  cut-off       logl_cutoff = the top `cutoff_percent` % of the saved trees once there are more than 1000 (:1662-1676)
  ratchet       every (ratchet_iter + 1)-th iteration (tools.cpp:778: every second): a random candidate tree, the alignment re-weighted
                (createPerturbAlignment: half of the informative sites once more, alignment.cpp:1915-1969), a climb there
                (on_ratchet_hclimb1), then a climb on the original alignment from where that one ended (on_ratchet_hclimb2, :1819-1851)
  otherwise     a random one of the `popSize` best candidate trees (candidateset.cpp:35-45; popSize 5, tools.cpp:658) perturbed by
                floor(curPerStrength * (n - 3)) random NNIs (:1739-1747; initPerStrength 0.5, tools.cpp:750), its length
                (computeParsimony, :1772), one climb (doNNISearch -> pllOptimizeSprParsimony, :1802, :2132)
  afterwards    a better tree than the best so far resets the stop rule's count (:1936-1961); candidateTrees.update (:1965)

All random choices -- candidate tree, NNIs, re-weighted sites -- come from the one random_double() stream the climb's tie rules
draw from; the engine takes and returns that stream by state (set_tie_state / tie_state), so the host draws between two climbs.
The perturbation steps themselves are device-free C++ in the library (mpboot_amd/host/iqflow.cpp: mpf_iq_*).

`eng` is anything with the engine's face (mpboot_amd.engine.FitchEngine; the tests drive the CPU oracle through the same
loop and compare the two after every climb).  The 100 start trees (phyloanalysis.cpp:1261-1317) are the caller's: add_candidate().
"""
from __future__ import annotations

import bisect
import math
import time

import numpy as np

from . import engine as _engine
from .rng import Lcg64


def unsuccess_iterations(n_taxa: int) -> int:
    """params.unsuccess_iteration when -stop_cond is not given (iqtree.cpp:129-130)."""
    return ((n_taxa - 1) // 100 + 1) * 100


class CandidateSet:
    """CandidateSet (candidateset.cpp): a multimap score -> tree (score = -length, the best LAST), at most `limit` trees
    (maxCandidates 100, tools.cpp:657), of which the `pop_size` best are the parents of perturbations."""

    def __init__(self, limit: int = 100, pop_size: int = 5):
        assert pop_size <= limit
        self.limit, self.pop_size = limit, pop_size
        self._scores = []          # ascending; equal scores in insertion order (std::multimap::insert)
        self._items = []           # (key, tree) beside _scores
        self.topologies = {}       # key -> score
        self.best_score = -math.inf

    def __len__(self):
        return len(self._scores)

    def _insert(self, score, key, tree):
        i = bisect.bisect_right(self._scores, score)
        self._scores.insert(i, score)
        self._items.insert(i, (key, tree))

    def _erase_key(self, key):
        for i, (k, _t) in enumerate(self._items):
            if k == key:
                del self._scores[i]
                del self._items[i]
                return

    def update(self, tree, score, key=None) -> bool:
        """CandidateSet::update (candidateset.cpp:104-150) -> True when the topology is new to the set and was taken.
        key: the tree's topology digest if the caller has it already (an exchange hands one tree to many sets)."""
        if key is None:
            key = _engine.iq_topology_key(tree)
        if score > self.best_score:
            self.best_score = score
        if key in self.topologies:
            if self.topologies[key] < score:
                self.topologies[key] = score
                self._erase_key(key)
                self._insert(score, key, tree)
            return False
        if len(self) < self.limit:
            self._insert(score, key, tree)
            self.topologies[key] = score
            return True
        if self._scores[0] <= score:
            del self.topologies[self._items[0][0]]
            del self._scores[0]
            del self._items[0]
            self._insert(score, key, tree)
            self.topologies[key] = score
            return True
        return False

    def rand_cand_tree(self, random_int):
        """getRandCandTree (candidateset.cpp:35-45): the random_int(min(popSize, size))-th tree from the best downwards."""
        assert len(self)
        i = random_int(min(self.pop_size, len(self)))
        return self._items[len(self) - 1 - i][1]

    def best(self):
        return self._items[-1][1], self._scores[-1]


class MpSearch:
    """IQTree::doTreeSearch, one iterate() per pass through its loop."""

    def __init__(self, eng, maxtrav: int = 6, mintrav: int = 1, tracked: bool = False, per_strength: float = 0.5, ratchet_iter: int = 1,
                 ratchet_percent: int = 50, ratchet_wgt: int = 1, pop_size: int = 5, max_candidates: int = 100, cutoff_percent: int = 10,
                 unsuccess: int | None = None, rescore: bool = True, weights=None):
        self.eng = eng
        self.n = int(eng.n)
        self.mintrav, self.maxtrav = mintrav, maxtrav
        self.tracked = tracked                  # a UFBoot tracker is attached to eng: cut-off updates, ratchet booking
        self.per_strength = per_strength
        self.ratchet_iter, self.ratchet_percent, self.ratchet_wgt = ratchet_iter, ratchet_percent, ratchet_wgt
        self.cutoff_percent = cutoff_percent
        self.cands = CandidateSet(max_candidates, pop_size)
        self.unsuccess = unsuccess_iterations(self.n) if unsuccess is None else unsuccess
        self.rescore = rescore                  # computeParsimony() of the perturbed tree and of the climb's result (:1772, :2143)
        self.w0 = np.asarray(eng.weights() if weights is None else weights, dtype=np.int32).copy()
        self._reset_order = getattr(eng, "reset_node_order", None) or eng.reset_nodep      # (the oracle's name for it)
        self.informative = np.asarray(eng.informative(), dtype=np.uint8)
        self.cur_it = 2                         # (curIt = 1 is the start-tree phase, iqtree.cpp:58, :1622)
        self.last_improved = 1                  # stop_rule.addImprovedIteration(1) (:1622)
        self.ratchet_count = 0
        self.best_score = -math.inf             # bestScore: -length of the best tree
        self.best_tree = None
        self.best_key = None
        self.log = []

    # -- the shared stream, between two climbs
    def _stream(self):
        g = Lcg64(0)
        g.state = np.uint64(self.eng.tie_state())
        return g

    def _hand_back(self, g):
        self.eng.set_tie_state(int(g.state))

    def add_candidate(self, tree, length: int):
        """a start tree enters the candidate set (phyloanalysis.cpp:1300-1313)"""
        tree = np.asarray(tree, dtype=np.int32).copy()
        self.cands.update(tree, -float(length))
        if -float(length) > self.best_score:
            self.best_score, self.best_tree, self.best_key = -float(length), tree, _engine.iq_topology_key(tree)

    def stop(self) -> bool:
        """StopRule::meetStopCondition, SC_UNSUCCESS_ITERATION (stoprule.cpp:92-93)"""
        return self.cur_it > self.last_improved + self.unsuccess

    def _climb(self, tree):
        e = self.eng
        e.set_tree(tree)
        self._reset_order()                     # (pllTreeInitTopologyNewick re-links the instance for every climb, :2129)
        s = e.optimize_spr(self.mintrav, self.maxtrav)
        return int(s), e.get_tree()

    def iterate(self) -> dict:
        e = self.eng
        t0 = time.perf_counter()
        if self.tracked:
            e.ufboot_set_cutoff(e.ufboot_next_cutoff(self.cutoff_percent))
        is_ratchet = self.ratchet_iter >= 0 and self.ratchet_iter == self.ratchet_count
        if self.ratchet_iter >= 0:
            self.ratchet_count += 1
        g = self._stream()
        tree = self.cands.rand_cand_tree(lambda k: int(g.ints(1, k)[0]))
        info = {"iteration": self.cur_it, "ratchet": bool(is_ratchet)}
        if is_ratchet:
            w, st = _engine.iq_perturb_weights(self.w0, self.informative, self.ratchet_percent, self.ratchet_wgt, int(g.state))
            e.set_tie_state(st)
            e.set_weights(w)
            t1 = time.perf_counter()
            try:
                s1, tree = self._climb(tree)            # on_ratchet_hclimb1
            finally:
                e.set_weights(self.w0)
            t2 = time.perf_counter()
            s, tree = self._climb(tree)                 # on_ratchet_hclimb2
            self.ratchet_count = 0
            info.update(perturb_s=t1 - t0, climb1_s=t2 - t1, climb2_s=time.perf_counter() - t2, score_perturbed_alignment=s1)
        else:
            num = int(math.floor(self.per_strength * (self.n - 3)))
            tree, st, relists = _engine.iq_random_nnis(tree, num, int(g.state))
            e.set_tie_state(st)
            if self.rescore:
                info["perturbed_score"] = int(e.score_tree(tree))
            t1 = time.perf_counter()
            s, tree = self._climb(tree)
            info.update(perturb_s=t1 - t0, climb_s=time.perf_counter() - t1, nnis=num, relists=relists)
        t3 = time.perf_counter()
        if self.rescore:
            assert int(e.score_tree()) == s             # (the reference's own cross-check of the two kernels, :2143, sprparsimony.cpp:3279)
        cur = -float(s)
        if cur > self.best_score:
            key = _engine.iq_topology_key(tree)
            if key != self.best_key:
                self.best_key = key
                self.last_improved = self.cur_it        # stop_rule.addImprovedIteration (:1946)
            self.best_score, self.best_tree = cur, tree.copy()
            info["better"] = True
        self.cands.update(tree.copy(), cur)
        self.cur_it += 1
        self._last_tree = tree.copy()
        info["_tree"] = self._last_tree
        info.update(score=s, seconds=time.perf_counter() - t0, after_s=time.perf_counter() - t3)
        self.log.append(info)
        return info

    def log_tree(self, info) -> bytes:
        """the tree an iteration ended on, as bytes (for the exchange of an iteration-parallel run)"""
        return np.ascontiguousarray(info["_tree"], dtype=np.int32).tobytes()

    def absorb(self, tree, length: int, key=None):
        """the result of ANOTHER chain's iteration: into the candidate set, and the best tree if it is one (iteration-parallel runs)"""
        tree = np.asarray(tree, dtype=np.int32).copy()
        cur = -float(length)
        if key is None:
            key = _engine.iq_topology_key(tree)
        if cur > self.best_score:
            self.best_score, self.best_tree, self.best_key = cur, tree, key
        self.cands.update(tree, cur, key)

    def iterations_left(self) -> int:
        return max(0, self.last_improved + self.unsuccess - self.cur_it + 1)

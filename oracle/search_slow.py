"""Second, independent restatement of mpboot's SPR hill climb and of IQTree::saveCurrentTree -- TEST INFRASTRUCTURE ONLY.

The C++ layer of the reference (sprparsimony.cpp, iqtree.cpp) cannot be compiled here (it includes the CMake-generated
iqtree_config.h), so the parts of the hot path that exist only there -- mpboot's random tie rule, the sweep-level accept rule,
saveCurrentTree's bookkeeping including its ratchet branch -- are unpinned.  This file is the second witness the first
restatement (oracle/fitch_oracle.c) is checked against: written separately from the reference text, in another language and
with another structure.  Nothing here is incremental: every insertion test builds the candidate topology and scores it FROM
SCRATCH with a per-pattern numpy Fitch pass (no traversal descriptors, no orientation flags, no per-site counters), and every
saveCurrentTree call recomputes the candidate's per-pattern lengths the same way.

    pllOptimizeSprParsimony      sprparsimony.cpp:3244-3319   -> SlowSearch.optimize
    nodeRectifierPars            sprparsimony.cpp:2046-2101   -> SlowSearch.rectify
    rearrangeParsimony           sprparsimony.cpp:2259-2376   -> SlowSearch.rearrange   (incl. its own saveCurrentTree, :2285-2289)
    addTraverseParsimony         sprparsimony.cpp:2208-2218   -> SlowSearch.traverse
    testInsertParsimony          sprparsimony.cpp:2106-2188   -> SlowSearch.test_insert
    IQTree::saveCurrentTree      iqtree.cpp:3271-3731         -> SlowSearch.save_current_tree (default options)

Parity status: UNPINNED (like what it witnesses); its scorer is tied to the pinned one in tests/test_search_slow.py.
"""
from __future__ import annotations

import sys

import numpy as np

from mpboot_amd.rng import Lcg64           # SPRNG lcg64 stream, pinned against the vendored generator (tests/test_oracle_golden.py)

LONG_MAX = float(2 ** 63 - 1)


def nx(r: int) -> int:
    v, s = divmod(r, 3)
    return 3 * v + (s + 1) % 3


def tip_sets(codes: np.ndarray, datatype: int) -> np.ndarray:
    """PLL tip codes -> state sets as integers (globalVariables.h:60-78)"""
    c = codes.astype(np.int64)
    if datatype == 0:
        return c
    sets = np.where(c < 20, np.left_shift(1, np.minimum(c, 19)), 0)
    sets = np.where(c == 20, 12, sets)
    sets = np.where(c == 21, 96, sets)
    return np.where(c >= 22, (1 << 20) - 1, sets)


class SlowSearch:
    def __init__(self, codes, weights, datatype, informative, tie_seed, samples=None, eps=0.5):
        self.n, self.P = codes.shape
        self.sets = tip_sets(np.asarray(codes), datatype)
        self.w = np.asarray(weights, dtype=np.int64).copy()
        self.inf = np.asarray(informative, dtype=bool)
        self.rng = Lcg64(tie_seed)
        self.draws = 0
        self.back = None
        self.nodep = [0] + [3 * i for i in range(1, 2 * self.n - 1)]
        self.moves = []
        self.tests = 0
        sys.setrecursionlimit(max(sys.getrecursionlimit(), 8 * self.n + 200))
        # -bb state (IQTree::setParams, iqtree.cpp:213-262)
        self.bb = samples is not None
        self.bb_on = self.bb
        if self.bb:
            self.samples = np.asarray(samples, dtype=np.int64)
            self.orig = self.w.copy()                      # original_sample
            B = self.samples.shape[0]
            self.eps = eps
            self.cutoff = 0.0
            self.boot_logl = [-LONG_MAX] * B
            self.boot_counts = [0] * B
            self.boot_trees = [-1] * B
            self.treels_logl = []
            self.topologies = {}
            self.ufb_draws = 0
            self.pattern_pars = np.zeros(self.P, dtype=np.int64)     # _pattern_pars: persists between calls
            self.ratchet = False
            self.ratchet_booking = True                    # !params->no_hclimb1_bb
            self.mulhits = False                           # params->multiple_hits
            self.cutoff_from_btrees = False                # params->cutoff_from_btrees (tools.cpp:2442)
            self.boot_tree_orig_logl = [0] * B             # iqtree.cpp:254
            self.store_trees = False                       # params->store_candidate_trees (-storetrees)
            self.duplicates = 0                            # duplication_counter
            self.rebooked = 0                              # ... of which booked again with a better length
            self.treels = {}                               # topology -> tree index (only trees that hit, :3503-3513)
            self.boot_sets = [set() for _ in range(B)]     # boot_trees_parsimony
            self.largest_set = 0
            self.topboot = 0                               # params->store_top_boot_trees (with -mulhits)
            self.boot_top = [[] for _ in range(B)]         # boot_trees_parsimony_top: [(tree index, rell)], best first
            self.boot_threshold = [-(2 ** 31 - 1)] * B     # iqtree.cpp:267
            self.distinct = 0                              # params->distinct_iter_top_boot (without -mulhits)
            self.cur_it = 0                                # IQTree::curIt
            self.boot_top_iter = [[] for _ in range(B)]    # boot_trees_parsimony_top_iter

    def draw(self) -> float:
        self.draws += 1
        return float(self.rng.doubles(1)[0])

    # ---------------------------------------------------------------- scoring from scratch
    def pattern_lengths(self, back) -> np.ndarray:
        """per-pattern Fitch length of the (complete) tree, rooted on tip 1; 0 for patterns the PLL engine drops"""
        n, sets, P = self.n, self.sets, self.P

        def down(rec):
            v = rec // 3
            if v <= n:
                return sets[v - 1], np.zeros(P, dtype=np.int64)
            sa, ca = down(int(back[nx(rec)]))
            sb, cb = down(int(back[nx(nx(rec))]))
            inter = sa & sb
            empty = inter == 0
            return np.where(empty, sa | sb, inter), ca + cb + empty

        s, c = down(int(back[3]))
        ptn = c + ((sets[0] & s) == 0)
        return np.where(self.inf, ptn, 0)

    def length(self, back) -> int:
        return int((self.pattern_lengths(back) * self.w).sum())

    # ---------------------------------------------------------------- tree state
    def set_tree(self, back):
        self.back = [int(x) for x in back]

    def set_weights(self, w):
        self.w = np.asarray(w, dtype=np.int64).copy()
        if self.bb:
            other = bool((self.w != self.orig).any())
            lost = bool(((self.orig > 0) & (self.w <= 0)).any())
            self.bb_on = (not other) or (not lost and self.ratchet_booking)
            self.ratchet = other and self.bb_on

    def hookup(self, a, b):
        self.back[a] = b
        self.back[b] = a

    def is_tip(self, r):
        return r // 3 <= self.n

    def rectify(self):
        """nodeRectifierPars: inner entries of nodep = entry records of a preorder walk from nodep[1]->back"""
        n = self.n
        self.start = self.nodep[1]
        count = 0

        def reorder(p):
            nonlocal count
            if self.is_tip(p):
                return
            self.nodep[count + n + 1] = p
            count += 1
            reorder(self.back[nx(p)])
            reorder(self.back[nx(nx(p))])

        reorder(self.back[self.start])

    def splits(self, back) -> frozenset:
        """the topology as its set of bipartitions (each named by the side without tip 1): what the sorted tree string
        of the reference identifies"""
        n, out = self.n, set()

        def down(rec):
            if rec // 3 <= n:
                return frozenset([rec // 3])
            s = down(int(back[nx(rec)])) | down(int(back[nx(nx(rec))]))
            out.add(s)
            return s

        down(int(back[3]))
        return frozenset(out)

    # ---------------------------------------------------------------- saveCurrentTree, default options (+ -mulhits)
    def save_current_tree(self, cur_logl: float):
        if not self.bb_on:
            return
        if self.ratchet:                                    # iqtree.cpp:3283-3295: from _pattern_pars AS IT STANDS
            cur_logl = -float((self.pattern_pars * self.orig * self.inf).sum())
        looked_up = False
        known = None
        if self.store_trees:                                           # :3302-3311 -storetrees: looked up before anything else
            key = self.splits(self.back)
            known = self.treels.get(key)
            looked_up = True
        if known is not None:                                          # :3313-3341
            self.duplicates += 1
            if cur_logl <= self.treels_logl[known] + 1e-4:
                return
            self.treels_logl[known] = cur_logl
            self.rebooked += 1
            tree_index = known
        else:
            if self.cutoff != 0.0 and cur_logl <= self.cutoff - 1e-4:  # :3343
                return
            tree_index = len(self.treels_logl)
            if self.store_trees:
                self.treels[key] = tree_index                          # :3346
            self.treels_logl.append(cur_logl)
        self.pattern_pars = self.pattern_lengths(self.back)            # :3365 pllComputePatternParsimony
        for b in range(self.samples.shape[0]):                         # :3411
            rell = -float((self.pattern_pars * self.samples[b]).sum())
            if self.distinct and not self.mulhits:                     # :3587-3680
                k, top, its = self.distinct, self.boot_top[b], self.boot_top_iter[b]
                thr = self.boot_threshold[b]
                if rell >= thr:
                    self.boot_counts[b] += 1
                take = rell > thr
                if not take and rell == thr:
                    self.ufb_draws += 1
                    take = self.draw() <= k * 1.0 / self.boot_counts[b]
                if take:
                    if rell > self.boot_logl[b]:
                        self.boot_counts[b] = 1
                    if not looked_up:
                        tree_index = self.treels.setdefault(self.splits(self.back), tree_index)
                        looked_up = True
                    self.topologies.setdefault(tree_index, list(self.back))
                    if self.cutoff_from_btrees:                        # :3617-3619
                        self.boot_tree_orig_logl[b] = int(cur_logl)
                    self.boot_trees[b] = tree_index
                    self.boot_logl[b] = max(self.boot_logl[b], rell)
                    t = min(k, len(its))
                    if any(top[c][0] == tree_index for c in range(t)):
                        continue
                    rep = next((c for c in range(t) if its[c] == self.cur_it), None)
                    if rep is not None:
                        if rell > top[rep][1]:
                            top[rep] = (tree_index, int(rell))
                    elif t < k:
                        its.append(self.cur_it)
                        top.append((tree_index, int(rell)))
                    else:
                        worst = min(range(t), key=lambda d: (top[d][1], d))
                        top[worst] = (tree_index, int(rell))
                        its[worst] = self.cur_it
                    self.boot_threshold[b] = min(r for _, r in top)
                continue
            if self.mulhits and self.topboot:                          # :3542-3585
                top = self.boot_top[b]
                if len(top) < self.topboot or rell > self.boot_threshold[b]:
                    if not looked_up:
                        tree_index = self.treels.setdefault(self.splits(self.back), tree_index)
                        looked_up = True
                    if tree_index == len(self.treels_logl) - 1:        # "if newly added"
                        full = len(top) == self.topboot
                        if not full or rell > self.boot_threshold[b]:
                            if full:
                                top.pop()
                            pos = next((i for i, (_, r) in enumerate(top) if r < rell), len(top))
                            top.insert(pos, (tree_index, int(rell)))
                            if full:
                                self.boot_threshold[b] = top[self.topboot - 1][1]
                            elif not self.boot_threshold[b] < rell:
                                self.boot_threshold[b] = int(rell)
                            self.topologies.setdefault(tree_index, list(self.back))
                continue
            if self.mulhits:                                           # :3498-3540
                if rell >= self.boot_logl[b]:
                    if not looked_up:
                        key = self.splits(self.back)
                        tree_index = self.treels.setdefault(key, tree_index)
                        looked_up = True
                    if rell > self.boot_logl[b]:
                        self.boot_sets[b].clear()
                        self.boot_logl[b] = rell
                    if self.cutoff_from_btrees and cur_logl > self.boot_tree_orig_logl[b]:      # :3523-3527
                        self.boot_tree_orig_logl[b] = int(cur_logl)
                    if tree_index not in self.boot_sets[b]:
                        self.boot_sets[b].add(tree_index)
                        self.largest_set = max(self.largest_set, len(self.boot_sets[b]))
                        self.topologies.setdefault(tree_index, list(self.back))
                continue
            accept = rell > self.boot_logl[b] + self.eps
            if not accept and rell > self.boot_logl[b] - self.eps:     # :3687-3688, short-circuit: the draw only on a tie
                self.ufb_draws += 1
                accept = self.draw() <= 1.0 / (self.boot_counts[b] + 1)
            if accept:
                if not looked_up:                                       # :3689-3707
                    tree_index = self.treels.setdefault(self.splits(self.back), tree_index)
                    looked_up = True
                self.topologies.setdefault(tree_index, list(self.back))
                if rell > self.boot_logl[b]:
                    self.boot_counts[b] = 1
                    self.boot_logl[b] = rell
                if self.cutoff_from_btrees:                             # :3716-3718
                    self.boot_tree_orig_logl[b] = int(cur_logl)
                self.boot_trees[b] = tree_index
            if rell == self.boot_logl[b]:
                self.boot_counts[b] += 1

    # ---------------------------------------------------------------- SPR neighbourhood
    def test_insert(self, p, q):
        r = self.back[q]
        self.hookup(nx(p), q)                               # insertParsimony
        self.hookup(nx(nx(p)), r)
        mp = self.length(self.back)
        self.tests += 1
        self.save_current_tree(-float(mp))                  # :2163-2166, before the tie rule
        if mp < self.best:
            self.hits = 1
        elif mp == self.best:
            self.hits += 1
        if mp < self.best or (mp == self.best and self.draw() <= 1.0 / self.hits):
            self.best, self.insert_rec, self.remove_rec = mp, q, p
        self.hookup(q, r)
        self.back[nx(p)] = self.back[nx(nx(p))] = -1

    def traverse(self, p, q, mintrav, maxtrav):
        mintrav -= 1
        if mintrav <= 0:
            self.test_insert(p, q)
        if not self.is_tip(q):
            maxtrav -= 1
            if maxtrav > 0:
                self.traverse(p, self.back[nx(q)], mintrav, maxtrav)
                self.traverse(p, self.back[nx(nx(q))], mintrav, maxtrav)

    def remove_node(self, p):
        q, r = self.back[nx(p)], self.back[nx(nx(p))]
        self.hookup(q, r)
        self.back[nx(p)] = self.back[nx(nx(p))] = -1

    def rearrange(self, p, mintrav, maxtrav):
        maxtrav = min(maxtrav, self.n - 3)
        if maxtrav < mintrav:
            return
        q = self.back[p]
        self.save_current_tree(-float(self.length(self.back)))        # :2285-2289
        if not self.is_tip(p):
            p1, p2 = self.back[nx(p)], self.back[nx(nx(p))]
            if not self.is_tip(p1) or not self.is_tip(p2):
                self.remove_node(p)
                for x in (p1, p2):
                    if not self.is_tip(x):
                        self.traverse(p, self.back[nx(x)], mintrav, maxtrav)
                        self.traverse(p, self.back[nx(nx(x))], mintrav, maxtrav)
                self.hookup(nx(p), p1)
                self.hookup(nx(nx(p)), p2)
        if not self.is_tip(q) and maxtrav > 0:
            q1, q2 = self.back[nx(q)], self.back[nx(nx(q))]

            def deep(x):
                return not self.is_tip(x) and (not self.is_tip(self.back[nx(x)]) or not self.is_tip(self.back[nx(nx(x))]))

            if deep(q1) or deep(q2):
                self.remove_node(q)
                m2 = max(mintrav, 2)
                for x in (q1, q2):
                    if not self.is_tip(x):
                        self.traverse(q, self.back[nx(x)], m2, maxtrav)
                        self.traverse(q, self.back[nx(nx(x))], m2, maxtrav)
                self.hookup(nx(q), q1)
                self.hookup(nx(nx(q)), q2)

    # ---------------------------------------------------------------- pllOptimizeSprParsimony
    def optimize(self, mintrav=1, maxtrav=6) -> int:
        n = self.n
        self.moves = []
        self.rectify()
        self.best = self.length(self.back)
        if self.bb and self.ratchet:
            # what the IQ-TREE kernel left in _pattern_pars for the climb's start tree (iqtree.cpp:1712-1714)
            self.pattern_pars = self.pattern_lengths(self.back)
        random_mp = self.best
        iter_hits = 1
        while True:
            start_mp = random_mp
            self.rectify()
            for i in range(1, 2 * n - 1):
                self.insert_rec = self.remove_rec = None
                self.hits = 1
                self.rearrange(self.nodep[i], mintrav, maxtrav)
                if self.best == random_mp:
                    iter_hits += 1
                if self.best < random_mp:
                    iter_hits = 1
                if (self.best < random_mp or (self.best == random_mp and self.draw() <= 1.0 / iter_hits)) and \
                        self.remove_rec is not None and self.insert_rec is not None:
                    self.remove_node(self.remove_rec)                       # restoreTreeRearrangeParsimony
                    r = self.back[self.insert_rec]
                    self.hookup(nx(self.remove_rec), self.insert_rec)
                    self.hookup(nx(nx(self.remove_rec)), r)
                    random_mp = self.best
                    self.moves.append((self.remove_rec, self.insert_rec, self.best))
            if not random_mp < start_mp:
                break
        return start_mp

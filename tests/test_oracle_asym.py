"""A second witness for the weighted (Sankoff) oracle under a cost matrix that is NOT symmetric.

The reference accepts any matrix (ParsTree::loadCostMatrixFile, parstree.cpp:31-95) and scores a tree rooted at the edge it
evaluates: newviewSankoffParsimonyIterativeFast (sprparsimony.cpp:477-551) gives a node, for each of its states i, the sum over
its two children of min_j(cost[i][j] + child[j]); evaluateSankoffParsimonyIterativeFast (:880-961) returns
sum_patterns w * min_x(left[x] + min_y(cost[x][y] + right[y])) with right = the record handed to evaluateParsimony and left = its
back.  testInsertParsimony evaluates p->next->next after hooking p->next to q and p->next->next to r = q->back (:2158).
oracle/fitch_oracle.c restates that with PLL's lazily oriented vectors and traversal descriptors; here the same numbers come
from a plain recursion over the rearranged tree, one rooted dynamic programme per number (numpy over the patterns).  With an
asymmetric matrix the two only agree if the oracle roots every evaluation where the reference does -- which is what the GPU
engine is then compared with (tests/test_gpu_sankoff.py).  CPU only."""
import sys

import numpy as np
import pytest

from helpers import load_fixture, trace_tokens

sys.setrecursionlimit(10000)


def nxt(r):
    return 3 * (r // 3) + (r % 3 + 1) % 3


def triangle_fix(c):
    c = c.astype(np.int64).copy()
    S = c.shape[0]
    for k in range(S):                                  # the reference's k-i-j loop, parstree.cpp:74-80
        for i in range(S):
            for j in range(S):
                if c[i, j] > c[i, k] + c[k, j]:
                    c[i, j] = c[i, k] + c[k, j]
    return c


def tip_costs(codes_row, S, highest, aa):
    """compressSankoffDNA (sprparsimony.cpp:2636-2825): 0 for the states of the tip's set, highest_cost for the others"""
    from oracle.search_slow import tip_sets
    sets = tip_sets(codes_row, 1 if aa else 0)
    out = np.full((len(codes_row), S), highest, dtype=np.int64)
    for k in range(S):
        out[(sets >> k) & 1 == 1, k] = 0
    return out


class Rooted:
    def __init__(self, codes, weights, cost, aa):
        self.n, self.P = codes.shape
        self.S = cost.shape[0]
        self.cost = triangle_fix(cost)
        self.w = np.asarray(weights, dtype=np.int64)
        highest = int(self.cost.max()) + 1               # highest_cost, sprparsimony.cpp:160
        self.tips = [tip_costs(codes[t], self.S, highest, aa) for t in range(self.n)]

    def vec(self, back, r):
        """cost vector of the subtree behind record r, seen from back[r] (the viewer is the parent)"""
        node = r // 3
        if node <= self.n:
            return self.tips[node - 1]
        out = np.zeros((self.P, self.S), dtype=np.int64)
        for c in (int(back[nxt(r)]), int(back[nxt(nxt(r))])):
            v = self.vec(back, c)
            out += (v[:, None, :] + self.cost[None, :, :]).min(axis=2)      # [ptn][i] = min_j(cost[i][j] + child[j])
        return out

    def length(self, back, p):
        """evaluateParsimony(p): right = the record handed over, left = its back"""
        left, right = self.vec(back, int(back[p])), self.vec(back, p)
        inner = (right[:, None, :] + self.cost[None, :, :]).min(axis=2)      # [ptn][x] = min_y(cost[x][y] + right[y])
        return int(((left + inner).min(axis=1) * self.w).sum())


def apply_spr(back, p, q):
    b = np.array(back, dtype=np.int64).copy()
    a1, a2 = int(b[nxt(p)]), int(b[nxt(nxt(p))])
    b[a1], b[a2] = a2, a1
    r = int(b[q])
    b[nxt(p)], b[q] = q, nxt(p)
    b[nxt(nxt(p))], b[r] = r, nxt(nxt(p))
    return b


@pytest.mark.parametrize("name", ["dna_clean", "dna_ambig", "dna_dups", "aa"])
def test_oracle_roots_an_asymmetric_matrix_like_the_reference(name):
    from oracle import pyoracle as po
    fx = load_fixture(name)
    codes, w = fx["codes_np"], fx["weights_np"]
    S = fx["S"]
    aa = S == 20
    rng = np.random.default_rng(17)
    c = rng.integers(1, 7, size=(S, S)).astype(np.uint32)
    c[np.triu_indices(S, 1)] += 2
    np.fill_diagonal(c, 0)
    o = po.Oracle(codes, w, datatype=fx["datatype"], cost=c)
    # (parsimony-uninformative patterns are not scored: their weight is zero for the recursion too)
    slow = Rooted(codes, np.where(o.informative() != 0, w, 0), c, aa)
    assert (slow.cost != slow.cost.T).any()                     # still asymmetric after the repair
    start = 3                                                   # tr->start = tip 1
    differ = 0
    for t in fx["trees"][:4]:
        b = np.array(t["back"], dtype=np.int32)
        s = o.score_tree(b)
        assert s == slow.length(b, start)
        differ += s != slow.length(b, int(b[start]))            # the same edge seen from the other end: another number
    assert differ > 0
    # every insertion test of a few prune nodes: the rearranged tree, rooted at the new node's edge towards r = q->back
    sc = fx["scan"][0]
    back = np.array(sc["back"], dtype=np.int32)
    o.reset_nodep()
    o.set_tree(back)
    cur = o.score_tree()
    o.seed_ties(po.TIE_RANDOM, 1)
    checked = 0
    for rec in sc["order"][:6]:
        o.set_best(cur)
        o.trace(True)
        o.rearrange(rec, 1, 6)
        side = None
        for tok in trace_tokens(*o.get_trace()):
            if tok in ("P", "Q"):
                side = tok
                continue
            q, mp = (int(x) for x in tok.split(":"))
            p = rec if side == "P" else int(back[rec])
            nb = apply_spr(back, p, q)
            assert mp == slow.length(nb, nxt(nxt(p))), (rec, side, q)
            checked += 1
    assert checked > 20

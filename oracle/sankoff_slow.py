"""Slow Sankoff dynamic programme on PLL tip codes (numpy over patterns) -- TEST INFRASTRUCTURE ONLY.

An independent second implementation (plain post-order DP from scratch for a given topology, the textbook
algorithm behind ParsTree::computePartialParsimony / computeParsimonyBranch, reference parstree.cpp:127-322,
:439-541) used to cross-check oracle/fitch_oracle.c's Sankoff mode.  Parity status: UNPINNED against a
reference run (sprparsimony.cpp / parstree.cpp cannot be built from their sources alone).
"""
import sys

import numpy as np


def state_sets(codes, datatype):
    """PLL tip codes -> bit masks (pllrepo/src/globalVariables.h:60-78)."""
    c = codes.astype(np.int64)
    if datatype == 0:
        return c
    m = np.where(c < 20, np.left_shift(1, np.minimum(c, 19)), 0)
    m = np.where(c == 20, 12, m)
    m = np.where(c == 21, 96, m)
    m = np.where(c >= 22, (1 << 20) - 1, m)
    return m


def close_triangle(cost):
    c = np.array(cost, dtype=np.int64)
    S = c.shape[0]
    for k in range(S):
        for i in range(S):
            for j in range(S):
                if c[i, j] > c[i, k] + c[k, j]:
                    c[i, j] = c[i, k] + c[k, j]
    return c


def tree_cost(codes, weights, back, cost, datatype=0, root_tip=1):
    """-> (weighted tree cost, per-pattern costs) rooted on the branch of tip `root_tip`: min_i(rest[i] + min_j(leaf[j] + cost[i][j])),
    the rest of the tree as the parent side (ParsTree::computeParsimonyBranch, parstree.cpp:439-541, from computeParsimony :101-116)."""
    n, P = codes.shape
    S = 4 if datatype == 0 else 20
    cost = close_triangle(cost)
    big = int(cost.max()) + 1
    sets = state_sets(codes, datatype)
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 4 * n + 100))

    def tipvec(t):
        v = np.empty((S, P), dtype=np.int64)
        for k in range(S):
            v[k] = np.where((sets[t - 1] >> k) & 1, 0, big)
        return v

    def nx(r):
        v, s = divmod(r, 3)
        return 3 * v + (s + 1) % 3

    def mplus(v):                      # m[z] = min_x (v[x] + cost[z][x])
        return np.min(v[None, :, :] + cost[:, :, None], axis=1)

    def down(rec):
        v = rec // 3
        if v <= n:
            return tipvec(v)
        return mplus(down(int(back[nx(rec)]))) + mplus(down(int(back[nx(nx(rec))])))

    a = tipvec(root_tip)
    b = down(int(back[3 * root_tip]))
    ptn = np.min(b + mplus(a), axis=0)
    return int((ptn * np.asarray(weights, dtype=np.int64)).sum()), ptn


def apply_spr(back, p, q):
    """prune node record p (with the subtree behind back[p]) and regraft it on branch (q, back[q])."""
    b = np.array(back, dtype=np.int32).copy()

    def nx(r):
        v, s = divmod(r, 3)
        return 3 * v + (s + 1) % 3

    a1, a2 = int(b[nx(p)]), int(b[nx(nx(p))])
    b[a1], b[a2] = a2, a1
    r = int(b[q])
    b[nx(p)], b[q] = q, nx(p)
    b[nx(nx(p))], b[r] = r, nx(nx(p))
    return b

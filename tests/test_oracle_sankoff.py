"""Sankoff (weighted parsimony) mode of the oracle.

The C++ layer that holds the reference's Sankoff code cannot be built from its sources alone, so there is no
reference run to pin against (parity status of this mode: unpinned).  What is checked:
  * with unit costs it must reproduce the PINNED Fitch oracle exactly (the reference's own claim: `-cost e`
    gives the identical trajectory and score), on every fixture;
  * with other cost matrices it must agree with an independent from-scratch Sankoff DP (oracle/sankoff_slow.py).
"""
import numpy as np
import pytest

from helpers import FIXTURES, load_fixture, trace_tokens
from oracle import pyoracle as po, sankoff_slow


def unit_cost(S):
    return (1 - np.eye(S, dtype=np.uint32)).astype(np.uint32)


def tstv_cost():
    c = np.full((4, 4), 2, dtype=np.uint32)          # A C G T: transitions A<->G, C<->T cost 1
    np.fill_diagonal(c, 0)
    c[0, 2] = c[2, 0] = c[1, 3] = c[3, 1] = 1
    return c


def random_metric(S, seed):
    rng = np.random.default_rng(seed)
    pts = rng.integers(0, 12, size=(S, 3))
    c = np.abs(pts[:, None, :] - pts[None, :, :]).sum(axis=2).astype(np.uint32)   # L1 distances: a metric
    c[c == 0] = 1
    np.fill_diagonal(c, 0)
    return c


@pytest.fixture(scope="module", params=FIXTURES)
def fx(request):
    return load_fixture(request.param)


def test_unit_costs_reproduce_fitch(fx):
    S = fx["S"]
    f = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    s = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=unit_cost(S))
    f.enable_persite(True)
    for t in fx["trees"][:4]:
        b = np.array(t["back"], dtype=np.int32)
        assert s.score_tree(b) == f.score_tree(b) == t["score"]
        assert (s.pattern_scores()[0] == f.pattern_scores()[0]).all()
    sc = fx["scan"][0]
    back = np.array(sc["back"], dtype=np.int32)
    for o in (f, s):
        o.reset_nodep()
        o.set_tree(back)
        o.score_tree()
        o.seed_ties(po.TIE_RANDOM, 3)
    for rec in sc["order"]:
        toks = []
        for o in (f, s):
            o.set_best(sc["score"])
            o.trace(True)
            o.rearrange(rec, 1, 6)
            toks.append(trace_tokens(*o.get_trace()))
        assert toks[0] == toks[1]


def test_unit_costs_same_trajectories(fx):
    S = fx["S"]
    start = np.array(fx["spr"]["start_back"], dtype=np.int32)
    res = []
    for cost in (None, unit_cost(S)):
        o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
        o.set_tree(start)
        o.seed_ties(po.TIE_RANDOM, 11)
        o.trace(True)
        sc = o.optimize_spr(1, 6)
        res.append((sc, [x.tolist() for x in o.get_moves()], o.get_tree().tolist()))
        o2 = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
        o2.seed_ties(po.TIE_RANDOM, 5)
        res[-1] += (o2.make_tree(42, 3)[0], o2.get_tree().tolist())
    assert res[0] == res[1]


@pytest.mark.parametrize("name", ["dna_ambig", "dna_dups", "aa"])
def test_general_costs_against_slow_dp(name):
    fx = load_fixture(name)
    cost = tstv_cost() if fx["datatype"] == 0 else random_metric(20, 4)
    o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    for t in fx["trees"][:3]:
        b = np.array(t["back"], dtype=np.int32)
        ref, ptn = sankoff_slow.tree_cost(fx["codes_np"], fx["weights"], b, cost, fx["datatype"])
        assert o.score_tree(b) == ref
        got, total = o.pattern_scores()
        assert total == ref
        inf = o.informative().astype(bool)
        assert (got[inf] == ptn[inf]).all()
    # insertion tests: every traced mp equals the from-scratch cost of the rearranged tree
    back = np.array(fx["scan"][0]["back"], dtype=np.int32)
    o.reset_nodep()
    o.set_tree(back)
    cur = o.score_tree()
    o.seed_ties(po.TIE_RANDOM, 2)
    order = o.nodep()[1:2 * o.n - 1]
    checked = 0
    for rec in order[::5]:
        o.set_best(cur)
        o.trace(True)
        o.rearrange(int(rec), 1, 4)
        q, mp = o.get_trace()
        prune = int(rec)
        for a, m in zip(q, mp):
            if a == -1:
                prune = int(rec)
            elif a == -2:
                prune = int(back[rec])
            elif checked < 60:
                moved = sankoff_slow.apply_spr(back, prune, int(a))
                assert sankoff_slow.tree_cost(fx["codes_np"], fx["weights"], moved, cost, fx["datatype"])[0] == int(m)
                checked += 1
    assert checked > 20

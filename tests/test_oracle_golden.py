"""The oracle (oracle/fitch_oracle.c) against what the REFERENCE computed.

Fixtures under tests/golden/ were produced by tests/golden/make_golden.py from
oracle/_ref/pll_ref_driver = the reference's PLL parsimony sources compiled
where they lie.  Integer work: every comparison is bit-exact.
"""
import json
import os

import numpy as np
import pytest

from helpers import FIXTURES, GOLDEN, hex_words, load_fixture, trace_tokens
from oracle import pyoracle as po


@pytest.fixture(scope="module", params=FIXTURES)
def fx(request):
    return load_fixture(request.param)


def make(fx, **kw):
    return po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], **kw)


def test_tip_packing_matches_compressDNA(fx):
    o = make(fx)
    assert o.W == fx["W"] and o.S == fx["S"]
    assert o.informative().tolist() == fx["informative"]
    for i, hx in enumerate(fx["tipvec_hex"]):
        assert (o.node_vector(i + 1).ravel() == hex_words(hx)).all(), f"tip {i + 1}"


def test_tree_scores(fx):
    o = make(fx)
    for t in fx["trees"]:
        assert o.score_tree(np.array(t["back"], dtype=np.int32)) == t["score"]


@pytest.mark.parametrize("which", [0, 1])
def test_scan_candidates_and_first_best(fx, which):
    sc = fx["scan"][which]
    back = np.array(sc["back"], dtype=np.int32)
    # PLL-original rule (no pre-evaluate, first best): the reference's own rearrangeParsimony
    o = make(fx)
    o.set_tree(back)
    assert o.score_tree() == sc["score"]
    assert o.nodep()[1:2 * o.n - 1].tolist() == sc["order"]
    o.seed_ties(po.TIE_FIRST)
    for rec, exp in zip(sc["order"], sc["best"]):
        o.set_best(sc["score"])
        o.rearrange(rec, 1, sc["maxtrav"])
        b, rem, ins = o.get_best()
        assert [rec, b, rem, ins] == exp
    assert o.score_tree() == sc["score_after"]
    # every candidate's score (reference testInsertParsimony, mpboot's pre-evaluate)
    o = make(fx)
    o.set_tree(back)
    o.score_tree()
    o.seed_ties(po.TIE_RANDOM, 1)
    for rec, exp in zip(sc["order"], sc["cands"]):
        o.set_best(sc["score"])
        o.trace(True)
        o.rearrange(rec, 1, sc["maxtrav"])
        assert trace_tokens(*o.get_trace()) == exp


def test_spr_hill_climb_trajectory(fx):
    spr = fx["spr"]
    o = make(fx)
    o.set_tree(np.array(spr["start_back"], dtype=np.int32))
    o.seed_ties(po.TIE_FIRST)
    o.trace(True)
    o.optimize_spr(1, spr["maxtrav"])
    rem, ins, sc = o.get_moves()
    assert [list(map(int, m)) for m in zip(rem, ins, sc)] == spr["moves"]
    assert o.get_tree().tolist() == spr["final_back"]
    assert o.score_tree() == spr["final_score"]


def test_randomized_stepwise_addition_trees(fx):
    for r in fx["ras"]:
        o = make(fx)
        o.seed_ties(po.TIE_FIRST)
        s, perm = o.make_tree(r["seed"], r["spr_dist"])
        assert perm[1:].tolist() == r["perm"]
        assert s == r["score"]
        assert o.get_tree().tolist() == r["back"]


def test_stepwise_addition_checkpoints(fx):
    rx = fx["rasx"]
    o = make(fx)
    s, best, ins = o.stepwise(rx["seed"])
    for step, _tip, b, i in rx["adds"]:
        assert (int(best[step]), int(ins[step])) == (b, i)
    assert o.get_tree().tolist() == rx["back"]
    assert o.score_tree() == rx["score"]


def test_sprng_stream():
    with open(os.path.join(GOLDEN, "sprng_lcg64.json")) as f:
        ref = json.load(f)
    for seed, vals in ref.items():
        got = po.lcg64_doubles(int(seed), len(vals))
        assert [float.fromhex(v) for v in vals] == got


def test_pattern_scores_sum_to_tree_score(fx):
    """sprparsimony.cpp:3363-3392 / iqtree.cpp:3366: sum(ptn * weight) == tree length."""
    o = make(fx)
    o.enable_persite(True)
    for t in fx["trees"][:3]:
        s = o.score_tree(np.array(t["back"], dtype=np.int32))
        ptn, total = o.pattern_scores()
        assert total == s == t["score"]

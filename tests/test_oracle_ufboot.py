"""CPU checks of the oracle's restatement of IQTree::saveCurrentTree (online UFBoot-MP bookkeeping).

The C++ layer of the reference cannot be built from its sources alone (oracle/Makefile), so this part of the oracle is
unpinned; what can be checked without the reference is checked here: every number the bookkeeping keeps is re-derived
through an independent path (from-scratch per-pattern lengths of the stored topologies, numpy REPS).
"""
import numpy as np
import pytest

from helpers import load_fixture
from oracle import pyoracle as po


def samples_for(fx, B, seed):
    rng = np.random.default_rng(seed)
    w = np.asarray(fx["weights"], dtype=np.float64)
    return rng.multinomial(int(w.sum()), w / w.sum(), size=B).astype(np.uint16)


def fresh(fx):
    return po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])


@pytest.mark.parametrize("name", ["dna_clean", "dna_ambig", "aa"])
def test_bookkeeping_is_self_consistent(name):
    fx = load_fixture(name)
    B = 16
    samples = samples_for(fx, B, 7)
    o = fresh(fx)
    o.set_tree(np.array(fx["trees"][1]["back"], dtype=np.int32))
    o.seed_ties(po.TIE_RANDOM, 5)
    o.ufboot_attach(samples)
    score = o.optimize_spr(1, 6)
    assert o.ufboot_bad() == 0                      # sum of per-pattern lengths == mp at every insertion test (:3366)
    logl, counts, trees = o.ufboot_state()
    saved = o.ufboot_tree_logl()
    # no cut-off: every insertion test is saved, and so is the current tree once per prune-node visit
    # (rearrangeParsimony's own call, sprparsimony.cpp:2285-2289): a whole number of sweeps over the 2n - 2 prune nodes
    extra = len(saved) - o.counters()[2]
    assert extra > 0 and extra % (2 * fx["n"] - 2) == 0
    assert saved.max() == -score
    chk = fresh(fx)
    chk.enable_persite(True)
    best_possible = np.full(B, -np.inf)
    for b in range(B):
        t = o.ufboot_tree(int(trees[b]))
        assert t is not None
        s = chk.score_tree(t)
        assert -s == saved[trees[b]]                # the stored topology is the tree that was scored
        ptn, tot = chk.pattern_scores()
        assert tot == s
        assert -(ptn.astype(np.int64) * samples[b]).sum() == logl[b]
        assert counts[b] >= 2                       # :3710-3730: reset to 1 then counted once more for the same tree
    assert (logl <= 0).all()


def test_cutoff_filter_and_percentile_rule():
    fx = load_fixture("dna_clean")
    samples = samples_for(fx, 8, 3)
    o = fresh(fx)
    o.set_tree(np.array(fx["trees"][2]["back"], dtype=np.int32))
    o.seed_ties(po.TIE_RANDOM, 9)
    o.ufboot_attach(samples)
    o.optimize_spr(1, 6)
    saved = o.ufboot_tree_logl()
    assert len(saved) > 1000
    cut = o.ufboot_next_cutoff(10)
    assert cut == np.sort(saved)[::-1][len(saved) * 10 // 100]      # iqtree.cpp:1664-1675
    n0 = len(saved)
    o.ufboot_set_cutoff(cut)
    o.set_tree(np.array(fx["trees"][3]["back"], dtype=np.int32))
    o.optimize_spr(1, 6)
    later = o.ufboot_tree_logl()[n0:]
    assert len(later) > 0 and (later > cut - 1e-4).all()             # :3343
    tests_total = o.counters()[2]
    assert len(o.ufboot_tree_logl()) < tests_total                    # some candidates were filtered


def test_draws_come_from_the_shared_stream():
    """the bookkeeping's tie-breaks and the SPR tie-breaks consume ONE random stream (tools.cpp:3363)"""
    fx = load_fixture("dna_ambig")
    samples = samples_for(fx, 12, 1)
    seq = po.lcg64_doubles(4, 100000)
    pos = {"i": 0}

    import ctypes as C

    @C.CFUNCTYPE(C.c_double, C.c_void_p)
    def draw(_):
        v = seq[pos["i"]]
        pos["i"] += 1
        return v

    a, b = fresh(fx), fresh(fx)
    start = np.array(fx["trees"][0]["back"], dtype=np.int32)
    for o in (a, b):
        o.set_tree(start)
        o.ufboot_attach(samples)
    a.seed_ties(po.TIE_RANDOM, 4)
    b.seed_ties(po.TIE_RANDOM, 4)
    po.lib().orc_set_rand_callback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    po.lib().orc_set_rand_callback(b.h, draw, None)
    sa, sb = a.optimize_spr(1, 6), b.optimize_spr(1, 6)
    assert sa == sb
    assert [x.tolist() for x in a.ufboot_state()] == [x.tolist() for x in b.ufboot_state()]
    assert pos["i"] > a.ufboot_draws() > 0          # the callback served both kinds of draws


def test_ratchet_booking_rule_consequences():
    """re-weighted (ratchet) climbs, reference iqtree.cpp:3283-3295 -- consequences of the rule that can be checked without
    the oracle's own bookkeeping: (1) with no cut-off every insertion test of the climb (and the current tree at every
    prune-node visit) is booked, the first entry with the
    ORIGINAL-alignment length of the climb's start tree (scored by a second oracle instance), never with its own perturbed
    length; (2) under a cut-off no tree passes nothing is booked; (3) -no_hclimb1_bb books nothing"""
    fx = load_fixture("dna_48")
    w0 = fx["weights_np"]
    rng = np.random.default_rng(2)
    samples = rng.multinomial(int(w0.sum()), w0 / w0.sum(), size=10).astype(np.uint16)
    pert = (w0 * (1 + (rng.random(len(w0)) < 0.3))).astype(np.int32)
    start = np.array(fx["trees"][3]["back"], dtype=np.int32)
    ref = po.Oracle(fx["codes_np"], w0, datatype=fx["datatype"])
    l_start = ref.score_tree(start)
    for mode in ("none", "all_fail", "no_hclimb1_bb"):
        o = po.Oracle(fx["codes_np"], w0, datatype=fx["datatype"])
        o.seed_ties(po.TIE_RANDOM, 4)
        o.ufboot_attach(samples)
        if mode == "no_hclimb1_bb":
            o.ufboot_set_ratchet_booking(False)
        if mode == "all_fail":
            o.ufboot_set_cutoff(-1.0)                      # every length >= 1 fails  -len <= -1 - 1e-4
        o.set_weights(pert)
        o.set_tree(start)
        t0 = o.counters()[2]
        s_pert = o.optimize_spr(1, 6)
        booked = o.ufboot_tree_logl()
        if mode == "none":
            extra = len(booked) - (o.counters()[2] - t0)           # + the current tree, once per prune-node visit
            assert len(booked) > 100 and extra > 0 and extra % (2 * fx["n"] - 2) == 0
            assert -booked[0] == l_start
            # original-alignment lengths, not perturbed ones: a perturbed length counts the added site copies too, so it
            # exceeds the original-alignment length of the same tree
            assert s_pert > ref.score_tree(o.get_tree())
            assert (-booked).min() < s_pert
        else:
            assert len(booked) == 0

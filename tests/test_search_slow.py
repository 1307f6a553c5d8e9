"""The two restatements of the unpinned part of the hot path against each other (CPU only).

oracle/fitch_oracle.c (incremental: traversal descriptors, lazily oriented vectors, per-site counters -- the structure of the
reference) and oracle/search_slow.py (every insertion test scored from scratch with a numpy per-pattern Fitch pass, every
saveCurrentTree call recomputing per-pattern lengths) were written separately from the reference text
(sprparsimony.cpp:2046-2376, :3244-3319; iqtree.cpp:3271-3731).  They must agree on every accepted move, the final tree, the
number of random draws and -- with -bb -- on treels_logl, boot_logl, boot_counts, boot_trees and the stored topologies, over a
normal climb, a ratchet climb and the climb back."""
import numpy as np
import pytest

from helpers import load_fixture
from oracle import pyoracle as po
from oracle.search_slow import LONG_MAX, SlowSearch


def both(fx, seed, samples=None):
    o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    o.seed_ties(po.TIE_RANDOM, seed)
    s = SlowSearch(fx["codes_np"], fx["weights_np"], fx["datatype"], fx["informative"], seed, samples)
    if samples is not None:
        o.ufboot_attach(samples)
    return o, s


@pytest.mark.parametrize("name", ["dna_clean", "dna_ambig", "dna_dups", "aa"])
def test_from_scratch_scorer_equals_the_reference_scores(name):
    fx = load_fixture(name)
    s = SlowSearch(fx["codes_np"], fx["weights_np"], fx["datatype"], fx["informative"], 1)
    for t in fx["trees"]:
        assert s.length(t["back"]) == t["score"]            # the reference's own evaluateParsimony (tests/golden)


@pytest.mark.parametrize("name,seed,tree", [("dna_dups", 3, 1), ("aa", 11, 4), ("dna_ambig", 7, 2), ("dna_clean", 5, 6)])
def test_hill_climb_with_random_ties(name, seed, tree):
    fx = load_fixture(name)
    start = np.array(fx["trees"][tree]["back"], dtype=np.int32)
    o, s = both(fx, seed)
    o.set_tree(start)
    s.set_tree(start)
    o.trace(True)
    assert o.optimize_spr(1, 6) == s.optimize(1, 6)
    rem, ins, sc = o.get_moves()
    assert [(int(a), int(b), int(c)) for a, b, c in zip(rem, ins, sc)] == s.moves
    assert len(s.moves) > 2
    assert o.get_tree().tolist() == s.back
    assert o.counters()[2] == s.tests


def bb_same(o, s):
    assert o.ufboot_tree_logl().tolist() == s.treels_logl
    logl, counts, trees = o.ufboot_state()
    assert [-LONG_MAX if v <= -LONG_MAX / 2 else v for v in logl.tolist()] == s.boot_logl
    assert counts.tolist() == s.boot_counts and trees.tolist() == s.boot_trees
    assert o.ufboot_draws() == s.ufb_draws
    for t in set(s.boot_trees):
        if t >= 0:
            assert o.ufboot_tree(t).tolist() == s.topologies[t]
    assert o.get_tree().tolist() == s.back


@pytest.mark.parametrize("cut", ["none", "tight"])
@pytest.mark.parametrize("name,seed", [("dna_dups", 2), ("aa", 9), ("dna_ambig", 4)])
def test_save_current_tree_over_normal_ratchet_normal_climbs(name, seed, cut):
    fx = load_fixture(name)
    w0 = fx["weights_np"]
    rng = np.random.default_rng(seed)
    samples = rng.multinomial(int(w0.sum()), w0 / w0.sum(), size=5).astype(np.uint16)
    pert = (w0 * (1 + (rng.random(len(w0)) < 0.3))).astype(np.int32)
    t = [np.array(fx["trees"][k]["back"], dtype=np.int32) for k in (1, 3, 5)]
    o, s = both(fx, seed, samples)
    for x in (o, s):
        x.set_tree(t[0])
    assert o.optimize_spr(1, 6) == s.optimize(1, 6)
    bb_same(o, s)
    if cut == "tight":
        c = float(np.sort(o.ufboot_tree_logl())[int(0.3 * len(s.treels_logl))])
        o.ufboot_set_cutoff(c)
        s.cutoff = c
    for x in (o, s):                                # the ratchet iteration: other weights, bookkeeping goes on (iqtree.cpp:3283-3295)
        x.set_weights(pert)
        x.set_tree(t[1])
    n0 = len(s.treels_logl)
    assert o.optimize_spr(1, 6) == s.optimize(1, 6)
    bb_same(o, s)
    if cut == "none":
        assert len(s.treels_logl) > n0              # (under a tight cut-off the ratchet climb may book nothing at all)
    for x in (o, s):
        x.set_weights(w0)
        x.set_tree(t[2])
    assert o.optimize_spr(1, 6) == s.optimize(1, 6)
    bb_same(o, s)
    assert o.ufboot_bad() == 0


@pytest.mark.parametrize("rule", ["default", "mulhits", "distinct"])
@pytest.mark.parametrize("name,seed", [("dna_dups", 2), ("aa", 9)])
def test_cutoff_from_btrees_over_normal_ratchet_normal_climbs(name, seed, rule):
    """-cutoff_from_btrees (tools.cpp:2442): boot_tree_orig_logl[b] = the logl under which sample b's tree was booked
    (iqtree.cpp:3523-3527 / :3617-3619 / :3716-3718 -- on ratchet climbs the value saveCurrentTree replaced it by), and the next
    iteration's cut-off is the smallest of them (:1657-1660).  With -mulhits the reference only ever RAISES the entries from their
    initial 0, which negative logls never do: the cut-off stays 0 there (restated as it stands)."""
    fx = load_fixture(name)
    w0 = fx["weights_np"]
    rng = np.random.default_rng(seed)
    samples = rng.multinomial(int(w0.sum()), w0 / w0.sum(), size=6).astype(np.uint16)
    pert = (w0 * (1 + (rng.random(len(w0)) < 0.3))).astype(np.int32)
    t = [np.array(fx["trees"][k]["back"], dtype=np.int32) for k in (1, 3, 5)]
    o, s = both(fx, seed, samples)
    o.ufboot_set_cutoff_from_btrees(True)
    s.cutoff_from_btrees = True
    if rule == "mulhits":
        o.ufboot_set_mulhits(True)
        s.mulhits = True
    if rule == "distinct":
        o.ufboot_set_distinct_iter(2)
        o.ufboot_set_iteration(1)
        s.distinct = 2
        s.cur_it = 1
    for k, (wgt, tree) in enumerate(((w0, t[0]), (pert, t[1]), (w0, t[2]))):
        for x in (o, s):
            x.set_weights(wgt)
            x.set_tree(tree)
        assert o.optimize_spr(1, 6) == s.optimize(1, 6)
        assert o.ufboot_orig_logl().tolist() == s.boot_tree_orig_logl
        cut = o.ufboot_next_cutoff(10)
        assert cut == float(min(s.boot_tree_orig_logl))
        if rule == "mulhits":
            assert cut == 0.0
        else:
            assert cut < 0.0
        o.ufboot_set_cutoff(cut)
        s.cutoff = cut
    assert o.ufboot_tree_logl().tolist() == s.treels_logl
    assert o.ufboot_bad() == 0


@pytest.mark.parametrize("name,seed", [("dna_dups", 2), ("aa", 9), ("dna_clean", 6)])
def test_mulhits_rule_over_normal_ratchet_normal_climbs(name, seed):
    """-mulhits (iqtree.cpp:3498-3540): every tree that reaches a sample's best REPS joins its set, trees of one
    topology share the index of the first of them that hit, no random draw is spent on the bookkeeping"""
    fx = load_fixture(name)
    w0 = fx["weights_np"]
    rng = np.random.default_rng(seed)
    samples = rng.multinomial(int(w0.sum()), w0 / w0.sum(), size=6).astype(np.uint16)
    pert = (w0 * (1 + (rng.random(len(w0)) < 0.3))).astype(np.int32)
    t = [np.array(fx["trees"][k]["back"], dtype=np.int32) for k in (1, 3, 5)]
    o, s = both(fx, seed, samples)
    o.ufboot_set_mulhits(True)
    s.mulhits = True
    largest = 0
    for k, w in enumerate((w0, pert, w0)):
        for x in (o, s):
            x.set_weights(w)
            x.set_tree(t[k])
        assert o.optimize_spr(1, 6) == s.optimize(1, 6)
        assert o.get_tree().tolist() == s.back
        assert o.ufboot_tree_logl().tolist() == s.treels_logl
        logl = o.ufboot_state()[0]
        assert [-LONG_MAX if v <= -LONG_MAX / 2 else v for v in logl.tolist()] == s.boot_logl
        for b in range(len(samples)):
            got = o.ufboot_sample_trees(b)
            assert got == sorted(s.boot_sets[b]) and len(got) >= 1
            for ti in got:
                assert o.ufboot_tree(ti).tolist() == s.topologies[ti]
        assert o.ufboot_draws() == s.ufb_draws == 0
        largest = max(largest, max(len(x) for x in s.boot_sets))
    print("largest set at a checkpoint", largest, "ever", s.largest_set)
    assert s.largest_set > 1                                 # some sample really held several equally good trees
    # one topology met again keeps its first index: fewer distinct indices than hits
    assert len(s.treels) < len(s.treels_logl)


@pytest.mark.parametrize("n_top", [1, 3, 10])
@pytest.mark.parametrize("name,seed", [("dna_dups", 2), ("aa", 9)])
def test_mulhits_topboot_rule(name, seed, n_top):
    """-mulhits -topboot N (iqtree.cpp:3542-3585): per sample the N best NEW trees in decreasing REPS order with the reference's
    threshold quirks (it stays at -INT_MAX until the first replacement in a full list)"""
    fx = load_fixture(name)
    w0 = fx["weights_np"]
    rng = np.random.default_rng(seed)
    samples = rng.multinomial(int(w0.sum()), w0 / w0.sum(), size=5).astype(np.uint16)
    pert = (w0 * (1 + (rng.random(len(w0)) < 0.3))).astype(np.int32)
    t = [np.array(fx["trees"][k]["back"], dtype=np.int32) for k in (1, 3, 5)]
    o, s = both(fx, seed, samples)
    o.ufboot_set_mulhits(True)
    o.ufboot_set_topboot(n_top)
    s.mulhits, s.topboot = True, n_top
    for k, w in enumerate((w0, pert, w0)):
        for x in (o, s):
            x.set_weights(w)
            x.set_tree(t[k])
        assert o.optimize_spr(1, 6) == s.optimize(1, 6)
        assert o.get_tree().tolist() == s.back
        assert o.ufboot_tree_logl().tolist() == s.treels_logl
        for b in range(len(samples)):
            top, thr = o.ufboot_sample_top(b)
            assert top == s.boot_top[b] and thr == s.boot_threshold[b]
            assert len(top) == min(n_top, len(top)) and [r for _, r in top] == sorted((r for _, r in top), reverse=True)
            for ti, _ in top:
                assert o.ufboot_tree(ti).tolist() == s.topologies[ti]
        assert o.ufboot_draws() == s.ufb_draws == 0
    assert all(len(x) == n_top for x in s.boot_top)


@pytest.mark.parametrize("k", [1, 2, 5])
@pytest.mark.parametrize("name,seed", [("dna_dups", 2), ("aa", 9), ("dna_ambig", 4)])
def test_distinct_iter_top_boot_rule(name, seed, k):
    """-distinct_iter_top_boot k (iqtree.cpp:3587-3680): per sample at most k trees, one representative per search iteration
    (curIt), accepted against the list's worst score with a k / count tie draw from the shared stream"""
    fx = load_fixture(name)
    w0 = fx["weights_np"]
    rng = np.random.default_rng(seed)
    samples = rng.multinomial(int(w0.sum()), w0 / w0.sum(), size=5).astype(np.uint16)
    pert = (w0 * (1 + (rng.random(len(w0)) < 0.3))).astype(np.int32)
    t = [np.array(fx["trees"][j]["back"], dtype=np.int32) for j in (1, 3, 5, 2, 6)]
    o, s = both(fx, seed, samples)
    o.ufboot_set_distinct_iter(k)
    s.distinct = k
    for it, w in enumerate((w0, pert, w0, w0, w0)):
        o.ufboot_set_iteration(it + 1)
        s.cur_it = it + 1
        for x in (o, s):
            x.set_weights(w)
            x.set_tree(t[it])
        assert o.optimize_spr(1, 6) == s.optimize(1, 6)
        assert o.get_tree().tolist() == s.back
        bb_same(o, s)
        for b in range(len(samples)):
            top, thr = o.ufboot_sample_top(b)
            assert top == s.boot_top[b] and thr == s.boot_threshold[b]
            assert o.ufboot_sample_iters(b) == s.boot_top_iter[b]
            assert 1 <= len(top) <= k
    assert s.ufb_draws > 0


@pytest.mark.parametrize("rule", ["default", "mulhits", "topboot", "distinct"])
@pytest.mark.parametrize("name,seed", [("dna_dups", 2), ("aa", 9), ("dna_ambig", 4)])
def test_storetrees_books_a_topology_once(name, seed, rule):
    """-storetrees (iqtree.cpp:3302-3346): every tree that reaches saveCurrentTree is looked up by topology before the cut-off;
    one met before is skipped unless its length improved on the recorded one (possible on ratchet climbs, whose lengths
    come from _pattern_pars as it stands), and then it is booked again under its old index without the cut-off test"""
    fx = load_fixture(name)
    w0 = fx["weights_np"]
    rng = np.random.default_rng(seed)
    samples = rng.multinomial(int(w0.sum()), w0 / w0.sum(), size=5).astype(np.uint16)
    pert = (w0 * (1 + (rng.random(len(w0)) < 0.3))).astype(np.int32)
    t = [np.array(fx["trees"][k]["back"], dtype=np.int32) for k in (1, 3, 5, 2)]
    o, s = both(fx, seed, samples)
    o.ufboot_set_store_trees(True)
    s.store_trees = True
    if rule in ("mulhits", "topboot"):
        o.ufboot_set_mulhits(True)
        s.mulhits = True
    if rule == "topboot":
        o.ufboot_set_topboot(3)
        s.topboot = 3
    if rule == "distinct":
        o.ufboot_set_distinct_iter(2)
        s.distinct = 2
    for it, w in enumerate((w0, pert, w0, w0)):
        if rule == "distinct":
            o.ufboot_set_iteration(it + 1)
            s.cur_it = it + 1
        if it == 2:
            c = float(np.sort(o.ufboot_tree_logl())[int(0.5 * len(s.treels_logl))])
            o.ufboot_set_cutoff(c)
            s.cutoff = c
        for x in (o, s):
            x.set_weights(w)
            x.set_tree(t[it])
        assert o.optimize_spr(1, 6) == s.optimize(1, 6)
        assert o.get_tree().tolist() == s.back
        assert o.ufboot_tree_logl().tolist() == s.treels_logl
        assert o.ufboot_duplicates() == s.duplicates
        logl, counts, trees = o.ufboot_state()
        assert [-LONG_MAX if v <= -LONG_MAX / 2 else v for v in logl.tolist()] == s.boot_logl
        if rule == "default":
            assert counts.tolist() == s.boot_counts and trees.tolist() == s.boot_trees
        if rule == "mulhits":
            assert [o.ufboot_sample_trees(b) for b in range(len(samples))] == [sorted(x) for x in s.boot_sets]
        if rule in ("topboot", "distinct"):
            assert [o.ufboot_sample_top(b)[0] for b in range(len(samples))] == s.boot_top
        assert o.ufboot_draws() == s.ufb_draws
    print("duplicates", s.duplicates, "booked again", s.rebooked, "of", len(s.treels_logl), "trees")
    assert s.duplicates > 0
    assert len(s.treels) == len(s.treels_logl)              # one index per topology

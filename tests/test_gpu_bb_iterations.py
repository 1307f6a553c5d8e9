"""A -bb run as the search repeats it (iqtree.cpp:1631-1965): one tracked climb, then later iterations under the percentile cut-off
(:1662-1676) from perturbed trees, every other one a ratchet iteration (climb on the re-weighted alignment, then on the original
one) -- engine against the oracle on every observable after every climb.  The later climbs begin far above the cut-off, where
nothing reaches saveCurrentTree's bookkeeping (:3343): the engine runs that stretch as the plain climb (k_climb / cost-only
batches, Engine::spr_sweeps_run) and hands over to the tracked loop in front of the first admissible insertion test."""
import numpy as np
import pytest

from helpers import same_topology

pytestmark = pytest.mark.gpu


def _perturb(trees, eng, back, rng, k, maxtrav):
    return trees.random_spr_moves(eng, back, rng, k, maxtrav)


def _same(e, o):
    em, om = [x.tolist() for x in e.moves()], [x.tolist() for x in o.get_moves()]
    assert em == [x[len(x) - len(em[0]):] for x in om]           # (the oracle's trace runs on over its climbs)
    assert (e.get_tree() == o.get_tree()).all()
    assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
    le, ce, te = e.ufboot_state()
    lo, co, to = o.ufboot_state()
    assert le.tolist() == lo.tolist() and ce.tolist() == co.tolist() and te.tolist() == to.tolist()
    assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws()
    assert e.tie_state() == o.tie_state()
    assert o.ufboot_bad() == 0
    for t in sorted(set(te.tolist())):
        if t >= 0:
            assert same_topology(e.ufboot_tree(t), o.ufboot_tree(t), e.n)


@pytest.mark.parametrize("rule", ["default", "mulhits"])
@pytest.mark.parametrize("n,P,alphabet,maxtrav", [(40, 1500, "DNA", 6), (64, 2500, "DNA", 4), (30, 600, "AA", 6)])
def test_later_iterations_under_the_cutoff_match_the_oracle(n, P, alphabet, maxtrav, rule):
    from mpboot_amd import engine, synth, trees
    from oracle import pyoracle as po
    letters, _ = synth.synth_alignment(n, P, alphabet, 0.07, seed=n + P)
    codes = synth.letters_to_codes(letters, alphabet)
    dt_e, dt_o = (engine.DNA, po.DNA) if alphabet == "DNA" else (engine.AA, po.AA)
    e = engine.FitchEngine(codes, datatype=dt_e)
    o = po.Oracle(codes, datatype=dt_o)
    rng = np.random.default_rng(3)
    samples = rng.multinomial(P, np.ones(P) / P, size=60).astype(np.uint16)
    start = trees.random_topology(n, np.random.default_rng(8))
    w0 = np.ones(P, dtype=np.int32)
    o.trace(True)
    for x, mode in ((e, engine.TIE_RANDOM), (o, po.TIE_RANDOM)):
        x.set_tree(start)
        x.seed_ties(mode, 21)
        x.ufboot_attach(samples)
        if rule == "mulhits":
            x.ufboot_set_mulhits(True)
    best_s = e.optimize_spr(1, maxtrav)
    assert o.optimize_spr(1, maxtrav) == best_s
    _same(e, o)
    best = e.get_tree()
    scratch = engine.FitchEngine(codes, datatype=dt_e)          # (the perturbation's own scans must not touch the tracked engine)
    quiet_before = e.get_option("ufb_quiet_climbs")
    launches_before = e.stats()["climb_launches"]
    for it in range(7):
        cut = e.ufboot_next_cutoff(10)
        assert cut == o.ufboot_next_cutoff(10)
        e.ufboot_set_cutoff(cut)
        o.ufboot_set_cutoff(cut)
        pert = _perturb(trees, scratch, best, rng, 8 if it % 3 else 2, maxtrav)
        if it % 2 == 1:
            # ratchet iteration (createPerturbAlignment, alignment.cpp:1915-1969): a climb on re-weighted patterns first
            w = w0.copy()
            w[rng.random(P) < 0.5] += 1
            for x in (e, o):
                x.set_weights(w)
                x.set_tree(pert)
            s1 = e.optimize_spr(1, maxtrav)
            assert o.optimize_spr(1, maxtrav) == s1
            _same(e, o)
            pert = e.get_tree()
            for x in (e, o):
                x.set_weights(w0)
        for x in (e, o):
            x.set_tree(pert)
        s = e.optimize_spr(1, maxtrav)
        assert o.optimize_spr(1, maxtrav) == s
        _same(e, o)
        if s <= best_s:
            best_s, best = s, e.get_tree()
    # the later climbs began as plain ones, and the dense stretches ran in the persistent kernel
    assert e.get_option("ufb_quiet_climbs") - quiet_before >= 7
    assert e.stats()["climb_launches"] > launches_before
    memo_batches.append(e.get_option("ufb_memo_batches"))


memo_batches = []


def test_the_search_came_back_to_known_optima_without_multiplying():
    """(behind the runs above) batches of a topology whose complete move-less sweep had produced no candidate event before are
    booked without a product (UfbState::quiet_topo) -- and the oracle comparison above held with them"""
    assert memo_batches and max(memo_batches) > 0


@pytest.mark.parametrize("n,P,alphabet,maxtrav,rate", [(40, 1500, "DNA", 6, 0.07), (64, 2500, "DNA", 4, 0.12), (30, 600, "AA", 6, 0.1)])
def test_the_reference_search_flow_matches_the_oracle(n, P, alphabet, maxtrav, rate):
    """IQTree::doTreeSearch as mpboot runs it (mpboot_amd.search.MpSearch; iqtree.cpp:1631-1965): a random one of the best candidate
    trees, floor(0.5 (n - 3)) random NNIs with doRandomNNIs' used-node rule (:1083-1106) -- every second iteration the ratchet's
    re-weighted climb + the climb on the original alignment instead (:1694-1716, :1819-1851) --, cut-off = top 10 % of the saved
    trees, all draws (candidate, NNIs, re-weighted sites, ties, bookkeeping) from ONE stream.  The same loop drives the engine and
    the oracle; after every iteration every observable of the two must agree."""
    from mpboot_amd import engine, search, synth, trees
    from oracle import pyoracle as po
    letters, _ = synth.synth_alignment(n, P, alphabet, rate, seed=n + P + 1)
    codes = synth.letters_to_codes(letters, alphabet)
    dt_e, dt_o = (engine.DNA, po.DNA) if alphabet == "DNA" else (engine.AA, po.AA)
    e = engine.FitchEngine(codes, datatype=dt_e)
    o = po.Oracle(codes, datatype=dt_o)
    samples = np.random.default_rng(3).multinomial(P, np.ones(P) / P, size=60).astype(np.uint16)
    # start trees (phyloanalysis.cpp:1270-1317: not booked), the same ones for both
    starts = []
    for k in range(4):
        e.seed_ties(engine.TIE_RANDOM, 1 + k)
        e.make_parsimony_tree(1 + (k + 1) * 12345, maxtrav)
        t = e.get_tree()
        starts.append((t, e.score_tree()))
        assert o.score_tree(t) == starts[-1][1]
    o.trace(True)
    S = []
    for x, mode in ((e, engine.TIE_RANDOM), (o, po.TIE_RANDOM)):
        x.ufboot_attach(samples)
        x.seed_ties(mode, 21)
        s = search.MpSearch(x, maxtrav=maxtrav, tracked=True, weights=np.ones(P, dtype=np.int32), unsuccess=10)
        for t, length in starts:
            s.add_candidate(t, length)
        S.append(s)
    kinds = []
    for _ in range(10):
        ie, io = S[0].iterate(), S[1].iterate()
        assert (ie["ratchet"], ie["score"], ie.get("better")) == (io["ratchet"], io["score"], io.get("better"))
        assert ie.get("perturbed_score") == io.get("perturbed_score")
        _same(e, o)
        assert S[0].cands._scores == S[1].cands._scores and [k for k, _t in S[0].cands._items] == [k for k, _t in S[1].cands._items]
        assert S[0].last_improved == S[1].last_improved and S[0].stop() == S[1].stop()
        kinds.append(ie["ratchet"])
    assert kinds == [False, True] * 5                            # ratchet_iter = 1: every second iteration (tools.cpp:778)
    assert e.stats()["climb_launches"] > 0                       # the dense stretches of these climbs ran in the persistent kernel


def test_a_larger_radius_on_a_known_optimum_is_not_taken_for_known():
    """UfbState::quiet_topo remembers topologies whose complete move-less sweep produced no candidate event -- for the
    neighbourhoods of THAT sweep.  A later climb at a larger radius on the same topology tests insertions the memo never saw
    (ADVICE r5): the entry is keyed on the radii, and the books must follow the oracle."""
    from mpboot_amd import engine, synth, trees
    from oracle import pyoracle as po
    n, P = 48, 1800
    letters, _ = synth.synth_alignment(n, P, "DNA", 0.09, seed=77)
    codes = synth.letters_to_codes(letters, "DNA")
    e, o = engine.FitchEngine(codes), po.Oracle(codes)
    samples = np.random.default_rng(4).multinomial(P, np.ones(P) / P, size=80).astype(np.uint16)
    start = trees.random_topology(n, np.random.default_rng(2))
    o.trace(True)
    for x, mode in ((e, engine.TIE_RANDOM), (o, po.TIE_RANDOM)):
        x.set_tree(start)
        x.seed_ties(mode, 5)
        x.ufboot_attach(samples)
    for radius, cut in ((2, False), (2, True), (6, True), (3, True), (7, True)):
        if cut:
            c = e.ufboot_next_cutoff(10)
            c = c if c != 0.0 else -float(e.score_tree() + 3)   # (few trees booked yet: a cut-off just above the optimum)
            e.ufboot_set_cutoff(c)
            o.ufboot_set_cutoff(c)
        t = e.get_tree()
        for x in (e, o):
            x.set_tree(t)
        assert e.optimize_spr(1, radius) == o.optimize_spr(1, radius)
        _same(e, o)

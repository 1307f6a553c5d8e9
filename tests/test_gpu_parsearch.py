"""Iteration-parallel -bb on the GPU (mpboot_amd/parsearch.py): two chains of IQTree::doTreeSearch as workers of one run -- engines on
host threads, trackers of their own, exchange every two iterations (mpf_ufboot_adopt) -- against the same run driven on two oracles:
every chain's books, draws and candidate set after every round.  (CPU twin with the merge rule itself and two gloo ranks:
tests/test_parsearch.py.)"""
import numpy as np
import pytest

from helpers import same_topology

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,P,alphabet,rate,maxtrav", [(40, 300, "DNA", 0.4, 2), (48, 400, "DNA", 0.5, 6), (24, 250, "AA", 0.5, 3)])
def test_two_chains_with_exchanges_match_the_oracle(n, P, alphabet, rate, maxtrav):
    from mpboot_amd import engine, parsearch, synth
    from oracle import pyoracle as po
    letters, _ = synth.synth_alignment(n, P, alphabet, rate, seed=n + P)
    codes = synth.letters_to_codes(letters, alphabet)
    dt_e, dt_o = (engine.DNA, po.DNA) if alphabet == "DNA" else (engine.AA, po.AA)
    samples = np.random.default_rng(3).multinomial(P, np.ones(P) / P, size=50).astype(np.uint16)
    e0 = engine.FitchEngine(codes, datatype=dt_e)
    starts = []
    for k in range(3):
        e0.seed_ties(engine.TIE_RANDOM, 1 + k)
        e0.make_parsimony_tree(1 + (k + 1) * 12345, maxtrav)
        starts.append((e0.get_tree(), e0.score_tree()))
    E = [e0, engine.FitchEngine(codes, datatype=dt_e)]
    O = [po.Oracle(codes, datatype=dt_o), po.Oracle(codes, datatype=dt_o)]
    kw = dict(maxtrav=maxtrav, seed=5, sync_every=2, search_kw=dict(unsuccess=50))
    re_ = parsearch.ParallelBbRun(E, samples, starts, tie_mode=engine.TIE_RANDOM, **kw)
    ro = parsearch.ParallelBbRun(O, samples, starts, tie_mode=po.TIE_RANDOM, **kw)
    adopted = 0
    for _ in range(4):
        ie, io = re_.round(), ro.round()
        for key in ("iterations", "adopted", "shipped_trees", "improved", "best_length"):
            assert ie[key] == io[key], key
        adopted += ie["adopted"]
        for e, o in zip(E, O):
            le, ce, te = e.ufboot_state()
            lo, co, to = o.ufboot_state()
            assert le.tolist() == lo.tolist() and ce.tolist() == co.tolist()
            assert e.tie_state() == o.tie_state()
            assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws()
            for b in range(samples.shape[0]):
                assert same_topology(e.ufboot_tree(int(te[b])), o.ufboot_tree(int(to[b])), n)
            assert o.ufboot_bad() == 0
        for se, so in zip(re_.searches, ro.searches):
            assert se.cands._scores == so.cands._scores and [k for k, _ in se.cands._items] == [k for k, _ in so.cands._items]
        assert re_.stop() == ro.stop()
    assert adopted > 0, "the fixture should let the chains hand trees to each other"
    # after an exchange every chain holds the same (shortest) length for every sample
    assert E[0].ufboot_state()[0].tolist() == E[1].ufboot_state()[0].tolist()
    re_.detach()


def test_adopt_refuses_what_it_cannot_keep():
    from mpboot_amd import engine, synth, trees
    letters, _ = synth.synth_alignment(12, 200, "DNA", 0.2, seed=1)
    codes = synth.letters_to_codes(letters, "DNA")
    e = engine.FitchEngine(codes)
    t = trees.random_topology(12, np.random.default_rng(1))
    with pytest.raises(engine.MpfError):
        e.ufboot_adopt([0], [10], [0], [t], [10])                 # no tracker
    samples = np.random.default_rng(3).multinomial(200, np.ones(200) / 200, size=8).astype(np.uint16)
    e.ufboot_attach(samples)
    bad = t.copy(); bad[40] = 3
    with pytest.raises(engine.MpfError):
        e.ufboot_adopt([0], [10], [0], [bad], [10])
    with pytest.raises(engine.MpfError):
        e.ufboot_adopt([9], [10], [0], [t], [10])                 # sample out of range
    assert e.ufboot_adopt([0, 1], [500, 400], [0, 0], [t], [450]) == 2
    assert e.ufboot_adopt([0, 1], [500, 399], [0, 0], [t], [450]) == 1          # only the strictly shorter one
    logl, cnt, bt = e.ufboot_state()
    assert (-logl[:2]).tolist() == [500, 399] and cnt[:2].tolist() == [2, 2] and bt[0] == bt[1] >= 0
    assert (e.ufboot_tree(int(bt[0])) == t).all()
    e2 = engine.FitchEngine(codes)
    e2.ufboot_attach(samples)
    e2.ufboot_set_mulhits(True)
    with pytest.raises(engine.MpfError):
        e2.ufboot_adopt([0], [10], [0], [t], [10])                # another update rule keeps other books

"""GPU parity: libmpfitch.so (through its C-ABI) against the oracle and the golden fixtures.

Integer work -- every comparison is bit-exact.
"""
import numpy as np
import pytest

from helpers import FIXTURES, hex_words, load_fixture, trace_tokens

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from mpboot_amd import engine, synth, trees
    from oracle import pyoracle as po
    return engine, po, synth, trees


@pytest.fixture(scope="module", params=FIXTURES)
def fx(request):
    return load_fixture(request.param)


# moves of the PLL-original hill climb (fixture "spr", reference's own run) that an exact first-best scorer shares with it
FIRST_BEST_PREFIX = {"dna_clean": 4, "dna_ambig": 1, "dna_dups": 1, "aa": 2, "dna_48": 3, "bin": 4, "morph": 10, "morph32": 4, "morph32_40": 2, "aa_40": 2}


def eng_of(engine, fx, **kw):
    return engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], **kw)


def orc_of(po, fx, **kw):
    return po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], **kw)


def test_native_library_is_loaded(mods):
    engine = mods[0]
    lib = engine.load_library()
    assert lib.mpf_abi_version() == 8
    with open("/proc/self/maps") as f:
        assert "libmpfitch.so" in f.read()


def test_tip_packing_matches_reference(mods, fx):
    engine = mods[0]
    e = eng_of(engine, fx)
    assert (e.S, e.W) == (fx["S"], fx["W"])
    assert e.informative().tolist() == fx["informative"]
    for i, hx in enumerate(fx["tipvec_hex"]):
        assert (e.tip_vector(i + 1).ravel() == hex_words(hx)).all()


def test_tree_scores_match_reference(mods, fx):
    engine = mods[0]
    e = eng_of(engine, fx)
    for t in fx["trees"]:
        assert e.score_tree(np.array(t["back"], dtype=np.int32)) == t["score"]
    backs = np.array([t["back"] for t in fx["trees"]], dtype=np.int32)
    assert e.score_trees(backs).tolist() == [t["score"] for t in fx["trees"]]


@pytest.mark.parametrize("which", [0, 1])
@pytest.mark.parametrize("opts", [dict(check_counts=1), dict(reduce=1), dict(words_per_lane=2),
                                  dict(split_below=0, check_counts=1), dict(scan_mode=0),
                                  dict(scan_mode=0, xcd_map=1), dict(scan_mode=0, words_per_lane=2, reduce=1),
                                  # planned programs (k_walk_plan + k_scan_prog) instead of the device walk
                                  dict(scan_prog=2, check_counts=1), dict(scan_prog=2, split_below=0, check_counts=1),
                                  dict(scan_prog=2, words_per_lane=2, xcd_map=0), dict(scan_prog=2, force_big=1, check_counts=1)])
def test_spr_scan_candidates_match_reference(mods, fx, which, opts):
    """every insertion test's tree length, in the reference's order (fixture 'cands' lines)"""
    engine = mods[0]
    sc = fx["scan"][which]
    e = eng_of(engine, fx)
    for k, v in opts.items():
        e.set_option(k, v)
    e.set_tree(np.array(sc["back"], dtype=np.int32))
    assert e.score_tree() == sc["score"]
    for rec, exp in zip(sc["order"], sc["cands"]):
        q, mp, n_p = e.spr_scan(rec, 1, sc["maxtrav"])
        toks = ["P"] + [f"{a}:{b}" for a, b in zip(q[:n_p], mp[:n_p])] + ["Q"] + [f"{a}:{b}" for a, b in zip(q[n_p:], mp[n_p:])]
        assert toks == exp, rec


def test_stepwise_addition_matches_reference(mods, fx):
    engine = mods[0]
    rx = fx["rasx"]
    e = eng_of(engine, fx)
    e.seed_ties(engine.TIE_FIRST)
    s, best, ins = e.stepwise_addition(rx["seed"])
    for step, _tip, b, i in rx["adds"]:
        assert (int(best[step]), int(ins[step])) == (b, i), step
    assert e.get_tree().tolist() == rx["back"]
    assert e.score_tree() == rx["score"]


@pytest.mark.parametrize("seed,mode,extra", [(1, 1, {}), (7, 1, {}), (2024, 1, {}), (7, 0, {}),
                                             (7, 1, dict(split_below=0, check_counts=1)),
                                             (7, 1, dict(views_mode=0, scan_batch=4, check_counts=1)),
                                             (7, 1, dict(scan_prog=2, check_counts=1)),
                                             (1, 1, dict(scan_prog=2, split_below=0, scan_batch=64))])
def test_spr_hill_climb_matches_oracle_trajectory(mods, fx, seed, mode, extra):
    """pllOptimizeSprParsimony with mpboot's random tie-breaks: same moves, same final topology"""
    engine, po = mods[0], mods[1]
    start = np.array(fx["spr"]["start_back"], dtype=np.int32)
    e = eng_of(engine, fx)
    e.set_option("scan_mode", mode)
    for k, v in extra.items():
        e.set_option(k, v)
    o = orc_of(po, fx)
    e.set_tree(start)
    o.set_tree(start)
    e.seed_ties(engine.TIE_RANDOM, seed)
    o.seed_ties(po.TIE_RANDOM, seed)
    o.trace(True)
    se = e.optimize_spr(1, 6)
    so = o.optimize_spr(1, 6)
    assert se == so
    em, om = e.moves(), o.get_moves()
    assert [x.tolist() for x in em] == [x.tolist() for x in om]
    assert (e.get_tree() == o.get_tree()).all()
    assert e.score_tree() == se


@pytest.mark.parametrize("seed,dist", [(42, 6), (7, 1), (99, 0), (5, 3)])
def test_randomized_stepwise_addition_tree_matches_oracle(mods, fx, seed, dist):
    """_pllComputeRandomizedStepwiseAdditionParsimonyTree, mpboot tie rule"""
    engine, po = mods[0], mods[1]
    e = eng_of(engine, fx)
    o = orc_of(po, fx)
    e.seed_ties(engine.TIE_RANDOM, seed)
    o.seed_ties(po.TIE_RANDOM, seed)
    se = e.make_parsimony_tree(seed, dist)
    so, _perm = o.make_tree(seed, dist)
    assert se == so
    assert (e.get_tree() == o.get_tree()).all()


def test_site_scores_match_oracle(mods, fx):
    """pllComputeSiteParsimony: per expanded site, zero padded"""
    engine, po = mods[0], mods[1]
    e = eng_of(engine, fx)
    o = orc_of(po, fx)
    o.enable_persite(True)
    nsite = int(np.sum(fx["weights_np"])) + 7
    for t in fx["trees"][:3]:
        back = np.array(t["back"], dtype=np.int32)
        e.set_tree(back)
        assert o.score_tree(back) == t["score"]
        se, te = e.site_scores(nsite)
        so, to = o.site_scores(nsite)
        assert te == to == t["score"]
        assert se.tolist() == so.tolist()


def test_pll_original_hill_climb_matches_reference_trajectory(mods, fx):
    """the PLL original's SPR hill climb (first-best rule): the reference's own accepted moves, final tree and score"""
    engine = mods[0]
    spr = fx["spr"]
    e = eng_of(engine, fx)
    e.set_tree(np.array(spr["start_back"], dtype=np.int32))
    e.seed_ties(engine.TIE_FIRST, 0)
    s = e.optimize_spr(1, spr["maxtrav"])
    rem, ins, sc = e.moves()
    got = [list(map(int, m)) for m in zip(rem, ins, sc)]
    # The PLL original has no evaluate before the scan of a prune node, so some of its insertions are scored on
    # vectors it has not refreshed yet and its path leaves an exact scorer's at the first such
    # insertion.  Per fixture, stated and asserted: the engine's trajectory equals the reference's own for exactly the
    # first FIRST_BEST_PREFIX moves (up to the original's first stale read), and both end at the same score.
    ref = spr["moves"]
    k = 0
    while k < min(len(got), len(ref)) and got[k] == ref[k]:
        k += 1
    assert k == FIRST_BEST_PREFIX[fx["name"]], (fx["name"], k)
    assert k < len(ref) and got != ref          # no fixture's original path is free of stale reads
    assert s == spr["final_score"]
    # What MPF_TIE_FIRST is, exactly: the original's first-best rule WITH mpboot's evaluate (sprparsimony.cpp:2285) --
    # the oracle in that configuration must give the engine's trajectory move for move (and the oracle WITHOUT it gives
    # the reference's own trajectory: tests/test_oracle_golden.py).
    po = mods[1]
    o = orc_of(po, fx)
    o.set_tree(np.array(spr["start_back"], dtype=np.int32))
    o.seed_ties(po.TIE_FIRST)
    o.set_pre_evaluate(1)
    o.trace(True)
    so = o.optimize_spr(1, spr["maxtrav"])
    assert so == s
    assert [list(map(int, m)) for m in zip(*o.get_moves())] == got
    assert (o.get_tree() == e.get_tree()).all()


def test_pll_original_parsimony_tree_matches_reference(mods, fx):
    """pllMakeParsimonyTreeFast of the PLL original (first-best rule): tree and score the reference itself produced"""
    engine = mods[0]
    for r in fx["ras"]:
        e = eng_of(engine, fx)
        e.seed_ties(engine.TIE_FIRST, 0)
        assert e.make_parsimony_tree(r["seed"], r["spr_dist"]) == r["score"]
        assert e.get_tree().tolist() == r["back"]


def test_reweighting_matches_oracle(mods, fx):
    """ratchet / bootstrap re-weighting: re-pack on the device, same scores"""
    engine, po = mods[0], mods[1]
    rng = np.random.default_rng(5)
    w = rng.integers(0, 4, size=len(fx["weights"])).astype(np.int32)
    e = eng_of(engine, fx)
    o = orc_of(po, fx)
    e.set_weights(w)
    o.set_weights(w)
    assert e.W == o.W
    for t in fx["trees"][:3]:
        b = np.array(t["back"], dtype=np.int32)
        assert e.score_tree(b) == o.score_tree(b)


def test_midsize_random_alignment_against_oracle(mods):
    """a 120-taxon case: scan of every prune node + hill climb, engine vs oracle"""
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(120, 3000, "DNA", 0.06, seed=12)
    codes = synth.letters_to_codes(letters)
    back = trees.random_topology(120, np.random.default_rng(8))
    e = engine.FitchEngine(codes)
    o = po.Oracle(codes)
    assert e.score_tree(back) == o.score_tree(back)
    o.seed_ties(po.TIE_RANDOM, 3)
    cur = o.score_tree()
    order = o.nodep()[1:2 * 120 - 1]
    for rec in order[::7]:
        o.set_best(cur)
        o.trace(True)
        o.rearrange(int(rec), 1, 6)
        toks = trace_tokens(*o.get_trace())
        q, mp, n_p = e.spr_scan(int(rec), 1, 6)
        mine = ["P"] + [f"{a}:{b}" for a, b in zip(q[:n_p], mp[:n_p])] + ["Q"] + [f"{a}:{b}" for a, b in zip(q[n_p:], mp[n_p:])]
        assert mine == toks
    e.seed_ties(engine.TIE_RANDOM, 11)
    o2 = po.Oracle(codes)
    o2.set_tree(back)
    o2.seed_ties(po.TIE_RANDOM, 11)
    assert e.optimize_spr(1, 6) == o2.optimize_spr(1, 6)
    assert (e.get_tree() == o2.get_tree()).all()


def test_pattern_scores_match_oracle(mods, fx):
    """pllComputePatternParsimony: per-pattern Fitch lengths; sum(ptn * weight) == tree length"""
    engine, po = mods[0], mods[1]
    e = eng_of(engine, fx)
    o = orc_of(po, fx)
    o.enable_persite(True)
    for t in fx["trees"][:4]:
        b = np.array(t["back"], dtype=np.int32)
        assert e.score_tree(b) == t["score"]
        ptn, total = e.pattern_scores()
        o.score_tree(b)
        optn, ototal = o.pattern_scores()
        assert total == ototal == t["score"]
        assert (ptn == optn).all()


def test_pattern_scores_after_reweighting_and_moves(mods, fx):
    engine, po = mods[0], mods[1]
    rng = np.random.default_rng(9)
    w = rng.integers(0, 3, size=len(fx["weights"])).astype(np.int32)
    e = eng_of(engine, fx)
    o = orc_of(po, fx)
    o.enable_persite(True)
    e.set_weights(w)
    o.set_weights(w)
    start = np.array(fx["spr"]["start_back"], dtype=np.int32)
    e.set_tree(start)
    e.seed_ties(engine.TIE_RANDOM, 5)
    s = e.optimize_spr(1, 6)
    ptn, total = e.pattern_scores()
    assert total == s
    assert o.score_tree(e.get_tree()) == s
    optn, _ = o.pattern_scores()
    assert (ptn == optn).all()


def test_compute_parsimony_dropin(mods, fx):
    """PhyloTree::computeParsimony(): tree score + _pattern_pars from IQ-TREE-encoded states"""
    engine = mods[0]
    from oracle import iqtree_fitch
    if fx["dedup"]:
        pytest.skip("fixture columns were re-ordered by PLL's duplicate removal")
    alpha = ("DNA", "AA", "BIN", "MOR")[fx["datatype"]]
    states = iqtree_fitch.convert_states(fx["rows"], alpha)
    codes = engine.encode_iqtree_states(states, fx["datatype"])
    assert (codes == fx["codes_np"]).all()                 # the same tip codes PLL's parser produced
    e = engine.FitchEngine(codes, fx["weights_np"], datatype=fx["datatype"])
    for t in fx["trees"][:3]:
        back = np.array(t["back"], dtype=np.int32)
        score, ptn = e.compute_parsimony(back)
        rs, rptn = iqtree_fitch.compute_parsimony(states, fx["weights"], back, (4, 20, 2, 32)[fx["datatype"]])
        assert score == t["score"] == rs
        assert (ptn == rptn).all()


def test_bootstrap_replicates_match_oracle(mods, fx):
    """re-weighted refinement climbs (optimizeBootTrees) and from-scratch searches per replicate"""
    engine, po = mods[0], mods[1]
    from mpboot_amd import bootstrap, shard
    from mpboot_amd.rng import Lcg64
    start = np.array(fx["spr"]["final_back"], dtype=np.int32)
    for mode in ("refine", "search"):
        e = eng_of(engine, fx)
        scores, trs = bootstrap.run_replicates(e, fx["weights_np"], 3, 17, 6, start, mode)
        for b in range(3):
            seed = shard.unit_seed(17, b)
            o = orc_of(po, fx)
            o.set_weights(bootstrap.bootstrap_weights(fx["weights_np"], Lcg64(seed)))
            o.seed_ties(po.TIE_RANDOM, seed)
            if mode == "refine":
                o.set_tree(start)
                s = o.optimize_spr(1, 6)
            else:
                s = o.make_tree(seed, 6)[0]
            assert s == scores[b]
            assert (o.get_tree() == trs[b]).all()


def test_reps_contraction_exact(mods):
    """K10: rell[m][b] = -sum_ptn pattern_pars[m][ptn] * boot[b][ptn] (iqtree.cpp:3411-3449), odd sizes included"""
    engine = mods[0]
    rng = np.random.default_rng(3)
    for (M, B, P) in ((1, 7, 33), (5, 1000, 2001), (37, 130, 4096)):
        pars = rng.integers(0, 400, size=(M, P)).astype(np.uint16)
        boot = rng.integers(0, 9, size=(B, P)).astype(np.uint16)
        boot[0, :5] = 65535
        pars[0, :5] = 65535 if M > 1 else pars[0, :5]
        r = engine.Reps(boot)
        got = r.scores(pars)
        exp = -(pars.astype(np.int64) @ boot.astype(np.int64).T)
        assert (got.astype(np.int64) == (exp & 0xFFFFFFFF).astype(np.uint32).astype(np.int32).astype(np.int64)).all() or (got == exp).all()
        r.close()


def test_reps_of_real_pattern_scores(mods, fx):
    """pattern scores of a tree x bootstrap resamples == the tree's length under each resample"""
    engine, po = mods[0], mods[1]
    from mpboot_amd import bootstrap
    from mpboot_amd.rng import Lcg64
    e = eng_of(engine, fx)
    back = np.array(fx["trees"][0]["back"], dtype=np.int32)
    score, ptn = e.compute_parsimony(back)
    boots = np.stack([bootstrap.bootstrap_weights(fx["weights_np"], Lcg64(s)) for s in range(6)]).astype(np.uint16)
    rell = engine.Reps(boots).scores(ptn)[0]
    for b in range(6):
        o = orc_of(po, fx)
        o.set_weights(boots[b].astype(np.int32))
        assert -rell[b] == o.score_tree(back)


def test_long_climb_from_random_tree_matches_oracle(mods):
    """hundreds of accepted moves, growing speculative batches, incremental view refreshes (buffers regrow mid-climb)"""
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(260, 1800, "DNA", 0.07, seed=21)
    codes = synth.letters_to_codes(letters)
    back = trees.random_topology(260, np.random.default_rng(4))
    for opts in (dict(), dict(scan_batch=4), dict(scan_batch=256, split_below=0), dict(scan_prog=2),
                 dict(scan_prog=2, scan_batch=256, split_below=0, check_counts=1)):
        e = engine.FitchEngine(codes)
        for k, v in opts.items():
            e.set_option(k, v)
        o = po.Oracle(codes)
        e.set_tree(back)
        o.set_tree(back)
        e.seed_ties(engine.TIE_RANDOM, 6)
        o.seed_ties(po.TIE_RANDOM, 6)
        o.trace(True)
        assert e.optimize_spr(1, 6) == o.optimize_spr(1, 6)
        assert len(e.moves()[0]) > 300
        assert [x.tolist() for x in e.moves()] == [x.tolist() for x in o.get_moves()]
        assert (e.get_tree() == o.get_tree()).all()

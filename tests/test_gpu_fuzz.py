"""Randomised engine-vs-oracle comparison: small random alignments (both alphabets, ambiguity codes, gaps, zero and
large weights, uninformative columns), random trees, radii 1..8, both tie rules, with and without the online UFBoot
bookkeeping.  Everything observable must match the oracle exactly."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# extended runs: MPF_FUZZ_OFFSET shifts the random cases (the option sets keep cycling with the test's own index)
FUZZ_OFFSET = int(os.environ.get("MPF_FUZZ_OFFSET", "0"))


def random_case(seed):
    from mpboot_amd import synth, trees

    seed = seed + FUZZ_OFFSET
    rng = np.random.default_rng(1000 + seed)
    aa = bool(rng.integers(0, 3) == 0)
    n = int(rng.integers(5, 41))
    P = int(rng.integers(20, 401))
    alphabet = "AA" if aa else "DNA"
    letters, names = synth.synth_alignment(n, P, alphabet, float(rng.uniform(0.03, 0.3)), seed=seed)
    codes = synth.letters_to_codes(letters, alphabet).copy()
    und = 22 if aa else 15
    # ambiguity / unknown codes as PLL tip codes
    m = rng.random(codes.shape)
    codes[m < 0.04] = und
    if aa:
        codes[(m >= 0.04) & (m < 0.06)] = 20 + rng.integers(0, 2)
    else:
        amb = np.array([3, 5, 6, 7, 9, 10, 11, 12, 13, 14], dtype=np.uint8)
        k = (m >= 0.04) & (m < 0.08)
        codes[k] = amb[rng.integers(0, len(amb), size=int(k.sum()))]
    # some constant / singleton columns, zero and heavy weights
    for j in rng.integers(0, P, size=max(1, P // 15)):
        codes[:, j] = codes[0, j]
    w = rng.integers(0, 4, size=P).astype(np.int32)
    w[rng.integers(0, P, size=3)] = int(rng.integers(5, 40))
    back = trees.random_topology(n, rng)
    return dict(aa=aa, n=n, P=P, codes=codes, w=w, back=back, maxtrav=int(rng.integers(1, 9)),
                tie=int(rng.integers(0, 2)), keep_all=bool(rng.integers(0, 4) == 0), seed=seed)


@pytest.mark.parametrize("seed", range(240))
def test_random_case_matches_oracle(seed):
    from mpboot_amd import engine
    from oracle import pyoracle as po

    c = random_case(seed)
    dt_e, dt_o = (engine.AA, po.AA) if c["aa"] else (engine.DNA, po.DNA)
    e = engine.FitchEngine(c["codes"], c["w"], datatype=dt_e, keep_all=c["keep_all"])
    o = po.Oracle(c["codes"], c["w"], datatype=dt_o, keep_all=c["keep_all"])
    # kernel / batching variants: none of them may change a result
    opts = [{}, {"words_per_lane": 2}, {"reduce": 1, "xcd_map": 0}, {"scan_batch": 3, "split_below": 0},
            {"scan_mode": 0}, {"views_mode": 0, "scan_batch": 128}, {"force_big": 1}, {"split_below": 4096, "check_counts": 1},
            {"views_mode": 1, "scan_batch": 2}, {"views_mode": 2, "scan_batch": 1, "check_counts": 1},
            {"scan_prog": 2, "check_counts": 1}, {"scan_prog": 2, "split_below": 0, "scan_batch": 64}][seed % 12]
    for k, v in opts.items():
        e.set_option(k, v)
    if opts.get("scan_mode") == 0 and seed % 16 == 4 and not c["aa"]:
        c["maxtrav"] = 10                       # host-planned scans reach radius 12
    if seed % 16 == 10:
        c["maxtrav"] = 9 + seed % 7             # ... and the deep kernels any radius, under the tracker too
    assert (e.W, e.num_informative) == (o.W, o.num_informative)
    if o.num_informative == 0:
        return
    assert e.score_tree(c["back"]) == o.score_tree(c["back"])
    o.enable_persite(True)
    o.score_tree(c["back"])
    pe, te = e.pattern_scores()
    po_, to = o.pattern_scores()
    assert te == to and pe.tolist() == po_.tolist()
    # one scan
    nodep = o.nodep()
    rec = int(nodep[1 + seed % (2 * c["n"] - 2)])
    cur = o.score_tree(c["back"])
    o.seed_ties(po.TIE_RANDOM, 1)
    o.set_best(cur)
    o.trace(True)
    o.rearrange(rec, 1, c["maxtrav"])
    tq, tm = o.get_trace()
    keep = tq >= 0
    e.set_tree(c["back"])
    q, mp, _n_p = e.spr_scan(rec, 1, c["maxtrav"])
    assert q.tolist() == tq[keep].tolist() and mp.tolist() == tm[keep].tolist()
    # a climb, optionally with the bookkeeping
    for x, tmode in ((e, c["tie"]), (o, c["tie"])):
        x.set_tree(c["back"])
        x.seed_ties(tmode, seed + 3)
    if c["tie"] == 0:
        o.set_pre_evaluate(1)                  # MPF_TIE_FIRST = first-best rule on exactly scored candidates
    ufb = seed % 2 == 0 and opts.get("scan_mode", 1) == 1
    if ufb:
        samples = np.random.default_rng(seed).multinomial(max(1, int(c["w"].sum())), (c["w"] + 1e-9) / (c["w"] + 1e-9).sum(), size=7).astype(np.uint16)
        e.ufboot_attach(samples)
        o.ufboot_attach(samples)
    o.trace(True)
    se, so = e.optimize_spr(1, c["maxtrav"]), o.optimize_spr(1, c["maxtrav"])
    assert se == so
    assert [a.tolist() for a in e.moves()] == [a.tolist() for a in o.get_moves()]
    assert (e.get_tree() == o.get_tree()).all()
    if ufb:
        assert [a.tolist() for a in e.ufboot_state()] == [a.tolist() for a in o.ufboot_state()]
        assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
        assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws()
    # a stepwise-addition tree
    e.ufboot_detach() if ufb else None
    for x, tmode in ((e, engine.TIE_RANDOM), (o, po.TIE_RANDOM)):
        x.seed_ties(tmode, seed)
    assert e.make_parsimony_tree(77 + seed, min(c["maxtrav"], 6)) == o.make_tree(77 + seed, min(c["maxtrav"], 6))[0]
    assert (e.get_tree() == o.get_tree()).all()


@pytest.mark.parametrize("seed", range(48))
def test_random_weighted_case_matches_oracle(seed):
    """the same for the weighted (Sankoff) engine: random cost matrices, small (packed 16-bit) and large (32-bit), symmetric and
    -- every fourth case -- not (then every length is the one of the edge the reference roots it at)"""
    from mpboot_amd import engine
    from oracle import pyoracle as po

    c = random_case(500 + seed)
    rng = np.random.default_rng(seed)
    S = 20 if c["aa"] else 4
    hi = 6 if seed % 3 else 4000
    m = rng.integers(1, hi, size=(S, S))
    cost = (np.triu(m, 1) + np.triu(m, 1).T).astype(np.uint32)
    if seed % 4 == 1:
        cost = m.astype(np.uint32)
        np.fill_diagonal(cost, 0)
    dt_e, dt_o = (engine.AA, po.AA) if c["aa"] else (engine.DNA, po.DNA)
    w = np.maximum(c["w"], 0)
    e = engine.FitchEngine(c["codes"], w, datatype=dt_e, cost=cost)
    o = po.Oracle(c["codes"], w, datatype=dt_o, cost=cost)
    if o.num_informative == 0:
        return
    assert e.score_tree(c["back"]) == o.score_tree(c["back"])
    pe, te = e.pattern_scores()
    po_, to = o.pattern_scores()
    assert te == to and pe.tolist() == po_.tolist()
    radius = min(c["maxtrav"], 6)
    for x, tmode in ((e, engine.TIE_RANDOM), (o, po.TIE_RANDOM)):
        x.set_tree(c["back"])
        x.seed_ties(tmode, seed + 1)
    o.trace(True)
    assert e.optimize_spr(1, radius) == o.optimize_spr(1, radius)
    assert [a.tolist() for a in e.moves()] == [a.tolist() for a in o.get_moves()]
    assert (e.get_tree() == o.get_tree()).all()
    for x, tmode in ((e, engine.TIE_RANDOM), (o, po.TIE_RANDOM)):
        x.seed_ties(tmode, seed + 2)
    assert e.make_parsimony_tree(31 + seed, min(radius, 3)) == o.make_tree(31 + seed, min(radius, 3))[0]
    assert (e.get_tree() == o.get_tree()).all()


@pytest.mark.parametrize("seed", list(range(48)) + [198])   # 198: a chained refresh too large for the in-kernel count fold
def test_medium_climb_matches_oracle(seed):
    """Hill climbs on 60-260 taxa from a random tree: hundreds of small scan batches, i.e. the incremental machinery of a
    climb (chained refresh over several chain levels, topology deltas, compact uploads, adaptive batch size) at a size where
    paths are long; moves, scores and the final tree against the oracle, then the start tree builder on the same engine."""
    from mpboot_amd import engine, synth, trees
    from oracle import pyoracle as po

    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(60, 261))
    P = int(rng.integers(200, 801))
    letters, _names = synth.synth_alignment(n, P, "DNA", float(rng.uniform(0.05, 0.25)), seed=100 + seed)
    codes = synth.letters_to_codes(letters, "DNA")
    w = rng.integers(1, 3, size=P).astype(np.int32)
    back = trees.random_topology(n, rng)
    tie = int(rng.integers(0, 2))
    e = engine.FitchEngine(codes, w, datatype=engine.DNA)
    o = po.Oracle(codes, w, datatype=po.DNA)
    opts = [{}, {"scan_batch": 4}, {"scan_batch": 1, "check_counts": 1}, {"views_mode": 1, "scan_batch": 8},
            {"words_per_lane": 2, "scan_batch": 2}, {"chain_max_ops": 0}, {"chain_max_ops": 100000, "scan_batch": 16},
            {"split_below": 0, "scan_batch": 64}][seed % 8]
    for k, v in opts.items():
        e.set_option(k, v)
    maxtrav = int(rng.integers(3, 8))
    for x, mode in ((e, engine.TIE_RANDOM if tie else engine.TIE_FIRST), (o, po.TIE_RANDOM if tie else po.TIE_FIRST)):
        x.seed_ties(mode, 3 + seed)
        x.set_tree(back)
    if not tie:
        o.set_pre_evaluate(1)                  # MPF_TIE_FIRST = first-best rule on exactly scored candidates
    o.trace(True)                              # (the oracle records its moves only while tracing)
    ufb = seed % 3 == 0                        # every third case with the online UFBoot bookkeeping on top
    if ufb:
        samples = rng.multinomial(int(w.sum()), w / w.sum(), size=6).astype(np.uint16)
        e.ufboot_attach(samples)
        o.ufboot_attach(samples)
    se, so = e.optimize_spr(1, maxtrav), o.optimize_spr(1, maxtrav)
    assert se == so
    assert (e.get_tree() == o.get_tree()).all()
    if ufb:
        assert o.ufboot_bad() == 0
        assert [a.tolist() for a in e.ufboot_state()] == [a.tolist() for a in o.ufboot_state()]
        assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
        assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws()
        e.ufboot_detach()
    me, mo = e.moves(), o.get_moves()
    assert [x.tolist() for x in me] == [np.asarray(x).tolist() for x in mo]
    assert len(me[0]) > 10
    # and the start-tree builder afterwards, on the engine that has just finished a climb (state carried over)
    e.seed_ties(engine.TIE_RANDOM, seed)
    o.seed_ties(po.TIE_RANDOM, seed)
    se, so = e.make_parsimony_tree(50 + seed, 2), o.make_tree(50 + seed, 2)[0]
    assert se == so
    assert (e.get_tree() == o.get_tree()).all()

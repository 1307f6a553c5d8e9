"""GPU parity of the weighted (Sankoff) mode: libmpfitch.so against the oracle's Sankoff mode."""
import numpy as np
import pytest

from helpers import FIXTURES, load_fixture, trace_tokens

pytestmark = pytest.mark.gpu


def unit_cost(S):
    return (1 - np.eye(S, dtype=np.uint32)).astype(np.uint32)


def tstv_cost():
    c = np.full((4, 4), 2, dtype=np.uint32)
    np.fill_diagonal(c, 0)
    c[0, 2] = c[2, 0] = c[1, 3] = c[3, 1] = 1
    return c


def random_metric(S, seed):
    rng = np.random.default_rng(seed)
    pts = rng.integers(0, 12, size=(S, 3))
    c = np.abs(pts[:, None, :] - pts[None, :, :]).sum(axis=2).astype(np.uint32)
    c[c == 0] = 1
    np.fill_diagonal(c, 0)
    return c


def random_asymmetric(S, seed):
    """step-matrix style costs: i -> j and j -> i differ (gains dearer than losses, plus noise); the loader's triangle repair
    is applied by engine and oracle alike (reference parstree.cpp:74-80) and leaves it asymmetric"""
    rng = np.random.default_rng(seed)
    c = rng.integers(1, 7, size=(S, S)).astype(np.uint32)
    c[np.triu_indices(S, 1)] += 2
    np.fill_diagonal(c, 0)
    return c


def cost_for(fx, kind):
    S = fx["S"]
    if kind == "unit":
        return unit_cost(S)
    if kind == "asym":
        return random_asymmetric(S, 11)
    return tstv_cost() if S == 4 else random_metric(S, 4)      # protein 20 x 20, binary 2 x 2, multistate 32 x 32


@pytest.fixture(scope="module")
def mods():
    from mpboot_amd import engine
    from oracle import pyoracle as po
    return engine, po


# (multistate data under a cost matrix always runs on the 32-state kernels, the reference's `case 32`, sprparsimony.cpp:571-573:
#  a state no tip has can be an inner node's cheapest label, so the matrix's states are not renumbered)
@pytest.fixture(scope="module", params=list(FIXTURES))
def fx(request):
    return load_fixture(request.param)


def test_weighted_multistate_runs_on_32_state_kernels(mods):
    engine, po = mods
    for name in ("morph", "morph32"):
        fx = load_fixture(name)
        e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=random_metric(32, 4))
        assert e.S == 32 and e.get_option("kernel_states") == 32
    fx = load_fixture("morph")
    assert engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"]).get_option("kernel_states") == 20
    fx = load_fixture("morph32")
    assert engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"]).get_option("kernel_states") == 32


@pytest.mark.parametrize("kind", ["unit", "general", "asym"])
def test_scores_patterns_and_scans(mods, fx, kind):
    """kind asym: a matrix that is not symmetric -- the length of a tree depends on the edge it is rooted at, and every number
    must be the one the reference's rooted kernels give there (evaluate: left = far end of the edge, :880-961; insertion test:
    rooted at the new node's edge towards the near side of the branch, :2158)"""
    engine, po = mods
    cost = cost_for(fx, kind)
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    for t in fx["trees"][:4]:
        b = np.array(t["back"], dtype=np.int32)
        s = e.score_tree(b)
        assert s == o.score_tree(b)
        if kind == "unit":
            assert s == t["score"]                      # unit costs == Fitch == the reference's number
        ptn, total = e.pattern_scores()
        optn, ototal = o.pattern_scores()
        assert total == ototal == s and (ptn == optn).all()
    sc = fx["scan"][0]
    back = np.array(sc["back"], dtype=np.int32)
    e.set_tree(back)
    cur = e.score_tree()
    o.reset_nodep()
    o.set_tree(back)
    assert o.score_tree() == cur
    o.seed_ties(po.TIE_RANDOM, 1)
    for rec in sc["order"]:
        o.set_best(cur)
        o.trace(True)
        o.rearrange(rec, 1, 6)
        toks = trace_tokens(*o.get_trace())
        q, mp, n_p = e.spr_scan(rec, 1, 6)
        mine = ["P"] + [f"{a}:{b}" for a, b in zip(q[:n_p], mp[:n_p])] + ["Q"] + [f"{a}:{b}" for a, b in zip(q[n_p:], mp[n_p:])]
        assert mine == toks, rec


@pytest.mark.parametrize("kind", ["unit", "general", "asym"])
def test_hill_climb_and_ras_trajectories(mods, fx, kind):
    engine, po = mods
    cost = cost_for(fx, kind)
    start = np.array(fx["spr"]["start_back"], dtype=np.int32)
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    e.set_tree(start)
    o.set_tree(start)
    e.seed_ties(engine.TIE_RANDOM, 9)
    o.seed_ties(po.TIE_RANDOM, 9)
    o.trace(True)
    assert e.optimize_spr(1, 6) == o.optimize_spr(1, 6)
    assert [x.tolist() for x in e.moves()] == [x.tolist() for x in o.get_moves()]
    assert (e.get_tree() == o.get_tree()).all()
    e2 = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    o2 = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    e2.seed_ties(engine.TIE_RANDOM, 4)
    o2.seed_ties(po.TIE_RANDOM, 4)
    assert e2.make_parsimony_tree(42, 3) == o2.make_tree(42, 3)[0]
    assert (e2.get_tree() == o2.get_tree()).all()
    if kind == "unit":
        f = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
        f.seed_ties(engine.TIE_RANDOM, 4)
        f.make_parsimony_tree(42, 3)
        assert (f.get_tree() == e2.get_tree()).all()     # `-cost e` == Fitch trajectory


@pytest.mark.parametrize("name,kind,maxtrav,kwords", [("dna_ambig", "general", 13, 0), ("dna_clean", "asym", 30, 16), ("aa", "general", 7, 0), ("aa", "asym", 12, 64),
                                                      ("morph32_40", "general", 9, 0), ("bin", "unit", 8, 0)])
def test_weighted_engine_at_any_radius(mods, name, kind, maxtrav, kwords):
    """-spr_rad above the levels the weighted scan keeps in registers (12 for DNA, 6 otherwise): k_snk_scan_deep parks the levels'
    transforms in HBM scratch (a small scratch cuts the batch into several launches) -- insertion tests in the reference's order
    and whole climbs equal the oracle's"""
    engine, po = mods
    fx = load_fixture(name)
    cost = cost_for(fx, kind)
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    if kwords:
        e.set_option("deep_scratch_kwords", kwords)
    sc = fx["scan"][0]
    back = np.array(sc["back"], dtype=np.int32)
    e.set_tree(back)
    cur = e.score_tree()
    o.reset_nodep()
    o.set_tree(back)
    assert o.score_tree() == cur
    o.seed_ties(po.TIE_RANDOM, 1)
    for rec in sc["order"][:12]:
        o.set_best(cur)
        o.trace(True)
        o.rearrange(rec, 1, maxtrav)
        toks = trace_tokens(*o.get_trace())
        q, mp, n_p = e.spr_scan(rec, 1, maxtrav)
        mine = ["P"] + [f"{a}:{b}" for a, b in zip(q[:n_p], mp[:n_p])] + ["Q"] + [f"{a}:{b}" for a, b in zip(q[n_p:], mp[n_p:])]
        assert mine == toks, rec
    start = np.array(fx["spr"]["start_back"], dtype=np.int32)
    e.set_tree(start)
    o.set_tree(start)
    e.seed_ties(engine.TIE_RANDOM, 9)
    o.seed_ties(po.TIE_RANDOM, 9)
    o.trace(True)
    assert e.optimize_spr(1, maxtrav) == o.optimize_spr(1, maxtrav)
    assert [x.tolist() for x in e.moves()] == [x.tolist() for x in o.get_moves()]
    assert (e.get_tree() == o.get_tree()).all()


def test_asymmetric_matrix_roots_like_the_reference(mods):
    """an asymmetric matrix is accepted (ParsTree::loadCostMatrixFile takes any, parstree.cpp:31-95): the same tree has
    different lengths at different root edges, the engine's are the oracle's at each of them -- the start edge (score_tree), the
    candidates of single prune nodes -- and differ from the transposed matrix's"""
    engine, po = mods
    fx = load_fixture("dna_ambig")
    c = random_asymmetric(4, 3)
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], cost=c)
    et = engine.FitchEngine(fx["codes_np"], fx["weights_np"], cost=c.T.copy())
    o = po.Oracle(fx["codes_np"], fx["weights_np"], cost=c)
    differ = 0
    for t in fx["trees"]:
        b = np.array(t["back"], dtype=np.int32)
        s = e.score_tree(b)
        assert s == o.score_tree(b)
        differ += s != et.score_tree(b)
    assert differ > 0
    # (online UFBoot on such an engine: tests/test_gpu_ufboot.py::test_weighted_tracker_under_an_asymmetric_matrix)


@pytest.mark.parametrize("name", ["dna_ambig", "aa"])
def test_packed_16_bit_and_32_bit_costs_agree(mods, name):
    """the two arithmetic widths (reference: default short / -short_off) give the same numbers while nothing overflows;
    large cost entries fall back to 32 bits by themselves"""
    engine, po = mods
    fx = load_fixture(name)
    cost = cost_for(fx, "general")
    start = np.array(fx["spr"]["start_back"], dtype=np.int32)
    res = []
    for short in (1, 0):
        e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
        e.set_option("sankoff_short", short)
        e.set_tree(start)
        e.seed_ties(engine.TIE_RANDOM, 3)
        s = e.optimize_spr(1, 6)
        ptn, tot = e.pattern_scores()
        q, mp, n_p = e.spr_scan(int(fx["scan"][0]["order"][0]), 1, 6)
        res.append((s, [x.tolist() for x in e.moves()], e.get_tree().tolist(), ptn.tolist(), tot, q.tolist(), mp.tolist(), n_p))
    assert res[0] == res[1]
    big = cost.astype(np.uint64) * 1500                       # 3 n (max + 1) >= 2^16: must not be packed
    big = big.astype(np.uint32)
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=big)
    o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=big)
    for t in fx["trees"][:3]:
        b = np.array(t["back"], dtype=np.int32)
        assert e.score_tree(b) == o.score_tree(b)
    e.set_tree(start)
    o.set_tree(start)
    e.seed_ties(engine.TIE_RANDOM, 3)
    o.seed_ties(po.TIE_RANDOM, 3)
    assert e.optimize_spr(1, 6) == o.optimize_spr(1, 6)
    assert (e.get_tree() == o.get_tree()).all()

"""Edge cases of the GPU path: tiny trees, clamped radius, keep-all-sites, large radius (host-planned and
device-walked), zero weights, error behaviour."""
import numpy as np
import pytest

from helpers import load_fixture, trace_tokens

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from mpboot_amd import engine, synth, trees
    from oracle import pyoracle as po
    return engine, po, synth, trees


def scan_tokens(e, rec, maxtrav):
    q, mp, n_p = e.spr_scan(int(rec), 1, maxtrav)
    return ["P"] + [f"{a}:{b}" for a, b in zip(q[:n_p], mp[:n_p])] + ["Q"] + [f"{a}:{b}" for a, b in zip(q[n_p:], mp[n_p:])]


@pytest.mark.parametrize("n", [4, 5, 6, 9])
def test_tiny_trees(mods, n):
    """maxtrav is clamped to ntips-3 (reference sprparsimony.cpp:2277-2278); with 4 taxa nothing can move"""
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(n, 80, "DNA", 0.3, seed=n)
    codes = synth.letters_to_codes(letters)
    back = trees.random_topology(n, np.random.default_rng(n))
    e = engine.FitchEngine(codes)
    o = po.Oracle(codes)
    assert e.score_tree(back) == o.score_tree(back)
    o.seed_ties(po.TIE_RANDOM, 1)
    cur = o.score_tree()
    for rec in o.nodep()[1:2 * n - 1]:
        o.set_best(cur)
        o.trace(True)
        o.rearrange(int(rec), 1, 6)
        assert scan_tokens(e, rec, 6) == trace_tokens(*o.get_trace())
    e.seed_ties(engine.TIE_RANDOM, 3)
    o2 = po.Oracle(codes)
    o2.set_tree(back)
    o2.seed_ties(po.TIE_RANDOM, 3)
    assert e.optimize_spr(1, 6) == o2.optimize_spr(1, 6)
    assert (e.get_tree() == o2.get_tree()).all()
    e3, o3 = engine.FitchEngine(codes), po.Oracle(codes)
    e3.seed_ties(engine.TIE_RANDOM, 8)
    o3.seed_ties(po.TIE_RANDOM, 8)
    assert e3.make_parsimony_tree(5, 6) == o3.make_tree(5, 6)[0]
    assert (e3.get_tree() == o3.get_tree()).all()


@pytest.mark.parametrize("maxtrav,mode", [(1, 1), (2, 1), (7, 1), (8, 1), (9, 1), (12, 0), (10, 0), (13, 1), (20, 0), (37, 1), (99, 1)])
def test_radii(mods, maxtrav, mode):
    """radius 7-8 uses the deep device-walked kernel, 9-12 the host-planned one, anything above (rearrangeParsimony takes whatever
    -spr_rad gives it: 37 = the whole 40-taxon tree, 99 is clipped to it) the host-planned programs with the levels' up-vectors in
    HBM (k_scan_deep)"""
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(40, 500, "DNA", 0.1, seed=31)
    codes = synth.letters_to_codes(letters)
    back = trees.random_topology(40, np.random.default_rng(5))
    e = engine.FitchEngine(codes)
    e.set_option("scan_mode", mode)
    e.set_option("check_counts", 1)
    o = po.Oracle(codes)
    assert e.score_tree(back) == o.score_tree(back)
    o.seed_ties(po.TIE_RANDOM, 1)
    cur = o.score_tree()
    for rec in o.nodep()[1:79:3]:
        o.set_best(cur)
        o.trace(True)
        o.rearrange(int(rec), 1, maxtrav)
        assert scan_tokens(e, rec, maxtrav) == trace_tokens(*o.get_trace())
    e.seed_ties(engine.TIE_RANDOM, 2)
    o2 = po.Oracle(codes)
    o2.set_tree(back)
    o2.seed_ties(po.TIE_RANDOM, 2)
    assert e.optimize_spr(1, maxtrav) == o2.optimize_spr(1, maxtrav)
    assert (e.get_tree() == o2.get_tree()).all()


@pytest.mark.parametrize("alphabet,n,P,maxtrav,kwords", [("AA", 30, 300, 16, 64), ("DNA", 60, 2500, 25, 16), ("DNA", 60, 2500, 14, 1 << 16)])
def test_radii_above_twelve_in_cut_launches(mods, alphabet, n, P, maxtrav, kwords):
    """k_scan_deep with a scratch so small that a batch of scans is cut into many launches (option deep_scratch_kwords), protein and
    DNA on several tiles: whole climbs against the oracle"""
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(n, P, alphabet, 0.09, seed=n + maxtrav)
    codes = synth.letters_to_codes(letters, alphabet)
    dt_e, dt_o = (engine.DNA, po.DNA) if alphabet == "DNA" else (engine.AA, po.AA)
    back = trees.random_topology(n, np.random.default_rng(9))
    e = engine.FitchEngine(codes, datatype=dt_e)
    e.set_option("deep_scratch_kwords", kwords)
    o = po.Oracle(codes, datatype=dt_o)
    for x, mode in ((e, engine.TIE_RANDOM), (o, po.TIE_RANDOM)):
        x.set_tree(back)
        x.seed_ties(mode, 4)
    o.trace(True)
    assert e.optimize_spr(1, maxtrav) == o.optimize_spr(1, maxtrav)
    assert [x.tolist() for x in e.moves()] == [x.tolist() for x in o.get_moves()]
    assert (e.get_tree() == o.get_tree()).all() and e.tie_state() == o.tie_state()


def test_keep_all_sites_and_zero_weights(mods):
    """-keep_aln (sort_alignment off: every site kept, sprparsimony.cpp:2462) and weights with zeros"""
    engine, po = mods[0], mods[1]
    fx = load_fixture("dna_ambig")
    w = fx["weights_np"].copy()
    w[::3] = 0
    e = engine.FitchEngine(fx["codes_np"], w, keep_all=True)
    o = po.Oracle(fx["codes_np"], w, keep_all=True)
    assert e.W == o.W and e.num_informative == o.num_informative == len(w)
    for t in fx["trees"][:3]:
        b = np.array(t["back"], dtype=np.int32)
        assert e.score_tree(b) == o.score_tree(b)
    o.enable_persite(True)
    o.score_tree()
    ptn, total = e.pattern_scores()
    optn, ototal = o.pattern_scores()
    assert total == ototal and (ptn == optn).all()


def test_all_patterns_uninformative(mods):
    engine, po = mods[0], mods[1]
    codes = np.full((6, 10), 1, dtype=np.uint8)
    codes[0, :] = 15
    e = engine.FitchEngine(codes)
    assert e.num_informative == 0
    from mpboot_amd import trees
    back = trees.random_topology(6, np.random.default_rng(1))
    assert e.score_tree(back) == 0
    e.seed_ties(engine.TIE_RANDOM, 1)
    assert e.optimize_spr(1, 6) == 0


def test_a_new_engine_needs_no_call_but_set_tree(mods):
    """mpf_set_tree + mpf_optimize_spr on an engine nothing else was asked of (no mpf_reset_node_order, no start tree made by it):
    nodep[] is the identity order from creation on, as the reference's tree set-up leaves it -- found by a probe that crashed at
    1000 taxa in round 5; score_tree in front of it, and the kernel path as well as the host loop"""
    engine, po = mods[0], mods[1]
    from mpboot_amd import synth, trees
    letters, _ = synth.synth_alignment(300, 3000, "DNA", 0.05, seed=12)
    codes = synth.letters_to_codes(letters, "DNA")
    back = trees.random_topology(300, np.random.default_rng(4))
    o = po.Oracle(codes)
    o.set_tree(back); o.seed_ties(po.TIE_RANDOM, 5)
    want = o.optimize_spr(1, 6)
    for dev in (0, 2):
        for score_first in (False, True):
            e = engine.FitchEngine(codes)
            e.set_option("climb_device", dev)
            if score_first:
                e.score_tree(back)
            e.set_tree(back)
            e.seed_ties(engine.TIE_RANDOM, 5)
            assert e.optimize_spr(1, 6) == want
            assert (e.get_tree() == o.get_tree()).all() and e.tie_state() == o.tie_state()


def test_error_behaviour(mods):
    engine = mods[0]
    fx = load_fixture("dna_clean")
    e = engine.FitchEngine(fx["codes_np"])
    with pytest.raises(engine.MpfError) as ei:
        e.score_tree()                                   # no tree yet
    assert ei.value.code == -5
    bad = np.array(fx["trees"][0]["back"], dtype=np.int32)
    bad[3] = 7                                           # tip 1 now points at an unused record
    with pytest.raises(engine.MpfError) as ei:
        e.set_tree(bad)
    assert ei.value.code == -2
    with pytest.raises(engine.MpfError):
        engine.FitchEngine(np.zeros((5, 8), dtype=np.uint8))          # code 0 is not a DNA state set
    with pytest.raises(engine.MpfError):
        engine.FitchEngine(fx["codes_np"], -np.ones(fx["codes_np"].shape[1], dtype=np.int32))
    e.set_tree(np.array(fx["trees"][0]["back"], dtype=np.int32))
    q13, mp13, _ = e.spr_scan(6, 1, 13)                  # any radius on the Fitch engine (k_scan_deep above 12 levels) ...
    q99, mp99, _ = e.spr_scan(6, 1, 99)                  # ... clipped to the tree like rearrangeParsimony clips it
    assert len(q13) > 0 and len(q99) >= len(q13)
    sk = engine.FitchEngine(fx["codes_np"], fx["weights_np"], cost=(1 - np.eye(4)).astype(np.uint32))
    sk.set_tree(np.array(fx["trees"][0]["back"], dtype=np.int32))
    k13, _m, _ = sk.spr_scan(6, 1, 13)                   # ... and on the weighted one (k_snk_scan_deep; tests/test_gpu_sankoff.py)
    assert k13.tolist() == q13.tolist()
    k300, _m, _ = sk.spr_scan(6, 1, 300)                 # (clipped to ntips - 3 first)
    assert k300.tolist() == q99.tolist()


def test_engine_reuse_across_trees_and_rebuilds(mods):
    """state carried in the engine (nodep order, views) must not leak between calls"""
    engine, po = mods[0], mods[1]
    fx = load_fixture("dna_dups")
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"])
    o = po.Oracle(fx["codes_np"], fx["weights_np"])
    for k, t in enumerate(fx["trees"]):
        b = np.array(t["back"], dtype=np.int32)
        e.set_tree(b)
        o.set_tree(b)
        e.seed_ties(engine.TIE_RANDOM, k)
        o.seed_ties(po.TIE_RANDOM, k)
        assert e.optimize_spr(1, 4) == o.optimize_spr(1, 4)
        assert (e.get_tree() == o.get_tree()).all()
        assert e.make_parsimony_tree(k, 2) == o.make_tree(k, 2)[0]
        assert (e.get_tree() == o.get_tree()).all()


@pytest.mark.parametrize("name", ["dna_ambig", "aa"])
def test_64_bit_addressing_path_of_the_scan_kernel(name):
    """vector stores of 2 GiB and more cannot go through one raw buffer; `force_big` runs that code path on a small input"""
    from helpers import load_fixture
    from mpboot_amd import engine
    from oracle import pyoracle as po

    fx = load_fixture(name)
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    e.set_option("force_big", 1)
    e.set_option("words_per_lane", 2 if name == "dna_ambig" else 1)      # ignored by the scan on this path
    o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    start = np.array(fx["spr"]["start_back"], dtype=np.int32)
    for x, mode in ((e, engine.TIE_RANDOM), (o, po.TIE_RANDOM)):
        x.set_tree(start)
        x.seed_ties(mode, 4)
    o.trace(True)
    assert e.optimize_spr(1, 6) == o.optimize_spr(1, 6)
    assert [x.tolist() for x in e.moves()] == [x.tolist() for x in o.get_moves()]
    assert (e.get_tree() == o.get_tree()).all()
    samples = np.random.default_rng(3).multinomial(len(fx["weights"]), np.ones(len(fx["weights"])) / len(fx["weights"]), size=9).astype(np.uint16)
    for x in (e, o):
        x.set_tree(start)
        x.ufboot_attach(samples)
    assert e.optimize_spr(1, 8) == o.optimize_spr(1, 8)                  # masks + deep walk on the same path
    assert [a.tolist() for a in e.ufboot_state()] == [a.tolist() for a in o.ufboot_state()]


def _ladder(trees, n):
    """unrooted ladder ((((t1,t2),t3),t4) ..., t_{n-1}, t_n): the deepest tree on n taxa"""
    import sys
    sys.setrecursionlimit(max(10000, 10 * n))
    names = [f"t{i+1}" for i in range(n)]
    inner = "(t1,t2)"
    for i in range(3, n - 1):
        inner = f"({inner},t{i})"
    return trees.newick_to_back(f"({inner},t{n-1},t{n});", names)


def _random_columns(synth, n, P, seed):
    g = np.random.default_rng(seed)
    L = np.repeat(g.integers(0, 4, size=P)[None, :], n, axis=0)
    mut = g.random((n, P)) < 0.35
    L[mut] = g.integers(0, 4, size=int(mut.sum()))
    return synth.letters_to_codes(L.astype(np.uint8))


def _climb_equals_oracle(engine, po, codes, back, w=None, maxtrav=6, aa=False):
    dt_e, dt_o = (engine.AA, po.AA) if aa else (engine.DNA, po.DNA)
    e = engine.FitchEngine(codes, w, datatype=dt_e)
    o = po.Oracle(codes, w, datatype=dt_o)
    assert e.score_tree(back) == o.score_tree(back)
    e.seed_ties(engine.TIE_RANDOM, 3)
    o.seed_ties(po.TIE_RANDOM, 3)
    o.trace(True)
    assert e.optimize_spr(1, maxtrav) == o.optimize_spr(1, maxtrav)
    assert [x.tolist() for x in e.moves()] == [x.tolist() for x in o.get_moves()]
    assert (e.get_tree() == o.get_tree()).all()
    pe, te = e.pattern_scores()
    o.enable_persite(True)
    o.score_tree()
    p_o, t_o = o.pattern_scores()
    inf = o.informative().astype(bool)
    assert te == t_o and (np.asarray(pe)[inf] == np.asarray(p_o)[inf]).all()
    return len(e.moves()[0])


@pytest.mark.parametrize("n,P,aa,maxtrav", [(300, 400, False, 6), (300, 400, False, 12), (150, 200, True, 6), (1200, 160, False, 6)])
def test_ladder_trees(mods, n, P, aa, maxtrav):
    """the deepest tree there is (n - 2 levels of views, prune nodes whose far side is one long chain): the whole climb away
    from it == the oracle's, move for move"""
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(n, P, "AA" if aa else "DNA", 0.08, seed=n + P)
    codes = synth.letters_to_codes(letters, "AA" if aa else "DNA")
    assert _climb_equals_oracle(engine, po, codes, _ladder(trees, n), maxtrav=maxtrav, aa=aa) > n


@pytest.mark.parametrize("n,P,maxtrav", [(3000, 96, 6), (6000, 40, 4)])
def test_thousands_of_taxa_on_a_short_alignment(mods, n, P, maxtrav):
    """three words of sites per vector, 12 000 node records: planning, batching and the refresh schedule dominate"""
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(n, P, "DNA", 0.03, seed=n)
    codes = synth.letters_to_codes(letters)
    assert _climb_equals_oracle(engine, po, codes, trees.random_topology(n, np.random.default_rng(5)), maxtrav=maxtrav) > n


@pytest.mark.parametrize("n,P", [(6, 3_000_000), (12, 1_200_000)])
def test_a_few_taxa_on_millions_of_patterns(mods, n, P):
    """vectors of 10 MB, a handful of them"""
    engine, po, synth, trees = mods
    _climb_equals_oracle(engine, po, _random_columns(synth, n, P, P), trees.random_topology(n, np.random.default_rng(5)))


def test_very_heavy_pattern_weights(mods):
    """weights of 65535, 100000, 250000 next to 0, 1 and 7 (the packed form repeats a pattern weight times, sprparsimony.cpp:2922-2943)"""
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(40, 600, "DNA", 0.1, seed=77)
    codes = synth.letters_to_codes(letters)
    rng = np.random.default_rng(5)
    w = rng.integers(0, 3, size=600).astype(np.int32)
    w[rng.integers(0, 600, size=6)] = [65535, 40000, 100000, 1, 250000, 7]
    assert _climb_equals_oracle(engine, po, codes, trees.random_topology(40, rng), w=w) > 10


@pytest.mark.parametrize("pipe,tile", [(0, 0), (1, 32), (1, 16), (1, 8), (1, 4), (1, 0)])
@pytest.mark.parametrize("name", ["dna_48", "aa"])
def test_refresh_kernel_variants(mods, name, pipe, tile):
    """options "views_pipe" / "views_tile": the level-synchronous refresh on 32-word tiles with half a wave per op, or on 32 / 16 /
    8 / 4-word tiles with the operands requested a round ahead (0: tile chosen from the row length) -- with "views_mode" 1 every
    refresh of the climb, partial ones included, runs on that kernel.  Scores, scans and the whole climb == the oracle's."""
    engine, po, synth, trees = mods
    fx = load_fixture(name)
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    e.set_option("views_mode", 1)
    e.set_option("views_pipe", pipe)
    e.set_option("views_tile", tile)
    assert e.get_option("views_pipe") == pipe and e.get_option("views_tile") == tile
    for t in fx["trees"][:4]:
        back = np.array(t["back"], dtype=np.int32)
        assert e.score_tree(back) == o.score_tree(back) == t["score"]
    o.seed_ties(po.TIE_RANDOM, 4)
    cur = o.score_tree()
    for rec in o.nodep()[1:12]:
        o.set_best(cur)
        o.trace(True)
        o.rearrange(int(rec), 1, 6)
        assert scan_tokens(e, rec, 6) == trace_tokens(*o.get_trace())
    start = np.array(fx["trees"][5]["back"], dtype=np.int32)
    for x, mode in ((e, engine.TIE_RANDOM), (o, po.TIE_RANDOM)):
        x.set_tree(start)
        x.seed_ties(mode, 6)
    o.trace(True)
    assert e.optimize_spr(1, 6) == o.optimize_spr(1, 6)
    assert [a.tolist() for a in e.moves()] == [a.tolist() for a in o.get_moves()]
    assert (e.get_tree() == o.get_tree()).all()
    with pytest.raises(engine.MpfError):
        e.set_option("views_tile", 12)

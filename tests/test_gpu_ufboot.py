"""Online UFBoot-MP bookkeeping (IQTree::saveCurrentTree inside pllOptimizeSprParsimony) -- device vs oracle.

The oracle follows the reference literally: per-site counter arrays, pllComputePatternParsimony after every
insertion test, REPS loop, DEFAULT update rule with its random tie-breaks.  The engine never forms a pattern vector
per candidate: scan masks x weights on the matrix cores + an event replay.  Everything observable must be identical:
scores, moves, the saved-tree list, boot_logl / boot_counts / boot_trees (and the topologies they name), and the
number of random draws consumed (the SPR trajectory is coupled to it).
"""
import os

import numpy as np
import pytest

from helpers import load_fixture, same_topology

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from mpboot_amd import engine
    from oracle import pyoracle as po

    return engine, po


def boot_samples(P, B, seed, weights=None, heavy=False):
    rng = np.random.default_rng(seed)
    w = np.ones(P) if weights is None else np.asarray(weights, dtype=np.float64)
    nsite = int(w.sum())
    s = rng.multinomial(nsite, w / w.sum(), size=B)
    if heavy:
        s[:, : max(1, P // 50)] += rng.integers(100, 400, size=(B, max(1, P // 50)))   # > 127: second weight plane
    return s.astype(np.uint16)


def run_both(engine, po, fx, start_back, samples, seed, maxtrav=6, cutoff=0.0, eps=0.5, keep_all=False, opts=None):
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], keep_all=keep_all)
    o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], keep_all=keep_all)
    for k, v in (opts or {}).items():
        e.set_option(k, v)
    e.set_tree(start_back)
    o.set_tree(start_back)
    e.seed_ties(engine.TIE_RANDOM, seed)
    o.seed_ties(po.TIE_RANDOM, seed)
    e.ufboot_attach(samples, eps)
    o.ufboot_attach(samples, eps)
    if cutoff:
        e.ufboot_set_cutoff(cutoff)
        o.ufboot_set_cutoff(cutoff)
    o.trace(True)
    se = e.optimize_spr(1, maxtrav)
    so = o.optimize_spr(1, maxtrav)
    return e, o, se, so


def assert_same(e, o, se, so):
    assert se == so
    assert o.ufboot_bad() == 0
    assert [x.tolist() for x in e.moves()] == [x.tolist() for x in o.get_moves()]
    assert (e.get_tree() == o.get_tree()).all()
    assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
    le, ce, te = e.ufboot_state()
    lo, co, to = o.ufboot_state()
    assert le.tolist() == lo.tolist()
    assert ce.tolist() == co.tolist()
    assert te.tolist() == to.tolist()
    assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws()
    for t in sorted(set(te.tolist())):
        if t >= 0:
            assert same_topology(e.ufboot_tree(t), o.ufboot_tree(t), e.n)


@pytest.mark.parametrize("name", ["dna_clean", "dna_ambig", "dna_dups", "aa"])
@pytest.mark.parametrize("seed", [3, 11])
def test_online_bookkeeping_matches_oracle(mods, name, seed):
    engine, po = mods
    fx = load_fixture(name)
    samples = boot_samples(len(fx["weights"]), 37, seed, fx["weights"])
    start = np.array(fx["trees"][seed % len(fx["trees"])]["back"], dtype=np.int32)
    e, o, se, so = run_both(engine, po, fx, start, samples, seed)
    assert_same(e, o, se, so)
    assert e.ufboot_counters()["events"] > 0


@pytest.mark.parametrize("opts", [{"scan_batch": 1}, {"scan_batch": 7, "split_below": 0}, {"scan_batch": 64, "split_below": 1000},
                                  {"words_per_lane": 2}, {"reduce": 1, "xcd_map": 0}])
def test_batching_and_kernel_options_do_not_change_the_result(mods, opts):
    engine, po = mods
    fx = load_fixture("dna_ambig")
    samples = boot_samples(len(fx["weights"]), 20, 5, fx["weights"])
    start = np.array(fx["trees"][1]["back"], dtype=np.int32)
    e, o, se, so = run_both(engine, po, fx, start, samples, 17, opts=opts)
    assert_same(e, o, se, so)


def _observables(e, rule):
    logl, cnt, tr = e.ufboot_state()
    obs = {"moves": [x.tolist() for x in e.moves()], "tree": e.get_tree().tolist(), "saved": e.ufboot_tree_logl().tolist(),
           "logl": logl.tolist(), "cnt": cnt.tolist(), "tr": tr.tolist(), "draws": e.ufboot_counters()["tie_draws"],
           "trees": {int(t): e.ufboot_tree(int(t)).tolist() for t in sorted(set(tr.tolist())) if t >= 0}}
    if rule in ("topboot", "distinct"):
        obs["tops"] = [list(map(list, e.ufboot_sample_top(b)[0])) + [e.ufboot_sample_top(b)[1]] for b in range(len(logl))]
    return obs


@pytest.mark.parametrize("tie", ["random", "first"])
@pytest.mark.parametrize("rule", ["default", "mulhits", "topboot", "distinct"])
@pytest.mark.parametrize("name", ["dna_48", "aa_40"])
def test_the_three_ways_through_a_tracked_climb_agree(mods, name, rule, tie):
    """ufb_fast = 0: scan, wait, product + extraction (chunked kernels), wait, replay.  ufb_pipe = 0: one dispatch chain and one
    wait per batch (k_ufb_events2, deferred log).  Default: the pipeline that takes the search's decision from the costs where
    they settle it, its log on a second host thread (ufb_thread = 0: on the same one).  Every observable must be the same -- and the default must really have decided batches early.  (The
    fixed-bound extraction of the top-N rules runs through both kernels here with batches of several hundred candidates.)"""
    engine, po = mods
    fx = load_fixture(name)
    w = np.asarray(fx["weights"], dtype=np.float64)
    samples = np.random.default_rng(77).multinomial(int(w.sum()), w / w.sum(), size=150).astype(np.uint16)
    start = np.array(fx["trees"][1]["back"], dtype=np.int32)
    got = []
    for opts in ({"ufb_fast": 0, "ufb_quiet": 0}, {"ufb_pipe": 0}, {"ufb_thread": 0}, {}, {"ufb_quiet": 0, "ufb_moot": 0, "ufb_memo": 0}):
        e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_tree(start)
        e.seed_ties(engine.TIE_RANDOM if tie == "random" else engine.TIE_FIRST, 5)
        e.ufboot_attach(samples)
        if rule in ("mulhits", "topboot"):
            e.ufboot_set_mulhits(True)
        if rule == "topboot":
            e.ufboot_set_topboot(3)
        if rule == "distinct":
            e.ufboot_set_distinct_iter(2)
            e.ufboot_set_iteration(1)
        s = e.optimize_spr(1, 6)
        early1 = e.get_option("ufb_early_batches")
        obs1 = _observables(e, rule)
        # a later iteration: cut-off from the saved trees, another start tree (every way takes the two-wait loop there: the product
        # is compacted on the host)
        cut = e.ufboot_next_cutoff(10)
        e.ufboot_set_cutoff(cut)
        if rule == "distinct":
            e.ufboot_set_iteration(2)
        e.set_tree(np.array(fx["trees"][5]["back"], dtype=np.int32))
        s2 = e.optimize_spr(1, 6)
        got.append(((s, s2, cut), (obs1, _observables(e, rule)), early1, e.get_option("ufb_batches"), e.get_option("ufb_early_batches"),
                    e.get_option("ufb_quiet_climbs"), e.tie_state()))
    assert got[0][0][2] != 0.0                      # (a cut-off was in force in the second climb)
    for g in got[1:]:
        assert got[0][:2] == g[:2]
        assert got[0][6] == g[6]                    # (the tie stream stands where it stood)
    assert got[0][2] == 0 and got[1][2] == 0
    if rule == "default":
        for g in got[2:]:
            assert g[2] > 0 and g[2] <= g[3]
    # the climb under the cut-off began as the plain one (ufb_quiet) -- except where that is switched off
    assert got[0][5] == 0 and got[4][5] == 0 and got[3][5] == 1


@pytest.mark.parametrize("opts", [{}, {"ufb_pipe": 0}, {"ufb_fast": 0}])
def test_a_failing_exchange_in_mid_climb_leaves_a_usable_engine(mods, opts):
    """the pipelined climb returns from the middle of a batch when the event exchange fails -- with the next batch possibly in
    flight on the device.  The call must fail loudly and the engine must afterwards behave like a new one."""
    import ctypes as C

    from mpboot_amd import shard
    engine, po = mods
    fx = load_fixture("dna_48")
    w = np.asarray(fx["weights"], dtype=np.float64)
    samples = np.random.default_rng(5).multinomial(int(w.sum()), w / w.sum(), size=40).astype(np.uint16)
    start = np.array(fx["trees"][0]["back"], dtype=np.int32)
    calls = {"n": 0}
    keep = {}

    def fn(_arg, tag, local_ptr, n_local, all_ptr, n_all_ptr):
        calls["n"] += 1
        if calls["n"] > 6:
            return 1
        buf = np.ctypeslib.as_array(C.cast(local_ptr, C.POINTER(C.c_uint32)), shape=(n_local, 3)).copy() if n_local else np.zeros((0, 3), dtype=np.uint32)
        keep["buf"] = buf
        all_ptr[0] = buf.ctypes.data if n_local else None
        n_all_ptr[0] = n_local
        return 0

    cb = shard.EXCHANGE_FN(fn)
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    for k, v in opts.items():
        e.set_option(k, v)
    e.set_tree(start)
    e.seed_ties(engine.TIE_RANDOM, 3)
    e.ufboot_attach(samples, 0.5, shard=(0, 2), exchange=cb)
    with pytest.raises(Exception):
        e.optimize_spr(1, 6)
    assert calls["n"] == 7
    e.ufboot_detach()
    f = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    for x in (e, f):
        x.set_tree(start)
        x.reset_node_order()
        x.seed_ties(engine.TIE_RANDOM, 3)
    assert e.score_tree(start) == f.score_tree(start)
    se, sf = e.optimize_spr(1, 6), f.optimize_spr(1, 6)
    assert se == sf and (e.get_tree() == f.get_tree()).all()
    assert [x.tolist() for x in e.moves()] == [x.tolist() for x in f.moves()]


def test_tracked_climb_with_the_hosts_own_random_stream(mods):
    """mpf_set_rand_callback (a host whose stream cannot be handed over by state): every draw of the pipelined tracked climb --
    the bookings' and the search's -- comes through the callback, in the reference's order.  A callback that runs the lcg64 of
    the seeded engine must leave every observable as the seeded engine's, and be called exactly once per draw."""
    import ctypes as C

    from mpboot_amd.engine import load_library
    engine, po = mods
    fx = load_fixture("dna_48")
    w = np.asarray(fx["weights"], dtype=np.float64)
    samples = np.random.default_rng(9).multinomial(int(w.sum()), w / w.sum(), size=60).astype(np.uint16)
    start = np.array(fx["trees"][2]["back"], dtype=np.int32)
    a = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    a.set_tree(start)
    a.seed_ties(engine.TIE_RANDOM, 41)
    state0 = a.tie_state()
    a.ufboot_attach(samples)
    sa = a.optimize_spr(1, 6)
    st = {"s": state0, "n": 0}

    def draw(_arg):
        st["s"] = (st["s"] * 0x27BB2EE687B0B0FD + 3037000493) & 0xFFFFFFFFFFFFFFFF
        st["n"] += 1
        return float(st["s"]) * 5.4210108624275222e-20

    cb = C.CFUNCTYPE(C.c_double, C.c_void_p)(draw)
    b = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    b.set_tree(start)
    b.seed_ties(engine.TIE_RANDOM, 1)                # (the rule; its own stream is not used)
    L = load_library()
    L.mpf_set_rand_callback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.mpf_set_rand_callback(b.h, C.cast(cb, C.c_void_p), None) == 0
    b.ufboot_attach(samples)
    sb = b.optimize_spr(1, 6)
    assert sa == sb
    assert _observables(a, "default") == _observables(b, "default")
    assert st["s"] == a.tie_state()                  # the callback's stream stands where the seeded engine's does
    assert b.get_option("ufb_early_batches") > 0


@pytest.mark.parametrize("opts", [{}, {"ufb_pipe": 0}])
def test_event_buffers_that_overflow_are_grown_and_extracted_again(mods, opts):
    """option ufb_event_cap (tests): event buffers that start at 16 entries -- every batch overflows until they have grown, the
    pipelined climb may not launch a successor early behind a batch that could overflow, and the second extraction must find
    the staging block where the first one did.  Same observables as with the default buffers."""
    engine, po = mods
    fx = load_fixture("dna_48")
    w = np.asarray(fx["weights"], dtype=np.float64)
    samples = np.random.default_rng(3).multinomial(int(w.sum()), w / w.sum(), size=90).astype(np.uint16)
    start = np.array(fx["trees"][4]["back"], dtype=np.int32)
    got = []
    for cap in (16, None):
        e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
        for k, v in opts.items():
            e.set_option(k, v)
        if cap:
            e.set_option("ufb_event_cap", cap)
        e.set_tree(start)
        e.seed_ties(engine.TIE_RANDOM, 13)
        e.ufboot_attach(samples)
        s = e.optimize_spr(1, 6)
        got.append((s, _observables(e, "default")))
    assert got[0] == got[1]


@pytest.mark.parametrize("n,radius", [(5, 1), (5, 6), (6, 1), (9, 2)])
def test_tiny_trees_on_a_sharded_tracker_take_the_same_books_every_way(mods, n, radius):
    """trees so small that whole batches have no insertion test at all (five taxa at radius 1): the current tree is still booked
    at every prune-node visit, and on a sharded tracker those bookings travel as events.  One rank of two with the other one
    silent (an exchange that returns what it was given): the pipelined climb, the one-chain loop and the two-wait loop must
    keep the same books."""
    import ctypes as C

    from mpboot_amd import shard, synth, trees
    engine, po = mods
    letters, _ = synth.synth_alignment(n, 80, "DNA", 0.25, seed=100 + n)
    codes = synth.letters_to_codes(letters, "DNA")
    P = codes.shape[1]
    samples = np.random.default_rng(n).multinomial(P, np.ones(P) / P, size=11).astype(np.uint16)
    start = trees.random_topology(n, np.random.default_rng(7))
    got = []
    for opts in ({"ufb_fast": 0}, {"ufb_pipe": 0}, {}):
        keep = {}

        def fn(_arg, tag, local_ptr, n_local, all_ptr, n_all_ptr):
            buf = np.ctypeslib.as_array(C.cast(local_ptr, C.POINTER(C.c_uint32)), shape=(n_local, 3)).copy() if n_local else np.zeros((0, 3), dtype=np.uint32)
            keep["buf"] = buf
            all_ptr[0] = buf.ctypes.data if n_local else None
            n_all_ptr[0] = n_local
            return 0

        cb = shard.EXCHANGE_FN(fn)
        e = engine.FitchEngine(codes, datatype=engine.DNA)
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_tree(start)
        e.seed_ties(engine.TIE_RANDOM, 5)
        e.ufboot_attach(samples, 0.5, shard=(0, 2), exchange=cb)
        s = e.optimize_spr(1, radius)
        got.append((s, _observables(e, "default")))
    assert got[0] == got[1] and got[0] == got[2]


def test_cutoff_filter_and_next_cutoff(mods):
    engine, po = mods
    fx = load_fixture("dna_clean")
    samples = boot_samples(len(fx["weights"]), 25, 9, fx["weights"])
    start = np.array(fx["trees"][2]["back"], dtype=np.int32)
    e, o, se, so = run_both(engine, po, fx, start, samples, 21)
    assert_same(e, o, se, so)
    cut_e, cut_o = e.ufboot_next_cutoff(10), o.ufboot_next_cutoff(10)
    assert cut_e == cut_o and cut_e != 0.0
    # boundary of the public ABI: 100 % indexes one past the end in the reference (iqtree.cpp:1666); the engine clamps to
    # the worst saved tree, 0 % is the best one
    assert e.ufboot_next_cutoff(100) == e.ufboot_tree_logl().min()
    assert e.ufboot_next_cutoff(0) == e.ufboot_tree_logl().max()
    # second climb (as the next search iteration would) under that cut-off, from another tree
    n_before = len(o.ufboot_tree_logl())
    start2 = np.array(fx["trees"][3]["back"], dtype=np.int32)
    for x in (e, o):
        x.ufboot_set_cutoff(cut_e)
        x.set_tree(start2)
    se, so = e.optimize_spr(1, 6), o.optimize_spr(1, 6)
    assert se == so
    assert len(o.ufboot_tree_logl()) > n_before
    le, ce, te = e.ufboot_state()
    lo, co, to = o.ufboot_state()
    assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
    assert (le.tolist(), ce.tolist(), te.tolist()) == (lo.tolist(), co.tolist(), to.tolist())
    assert (e.ufboot_tree_logl()[n_before:] > cut_e - 1e-4).all()


def test_more_than_128_samples_use_the_wide_column_tile(mods):
    engine, po = mods
    fx = load_fixture("dna_dups")
    samples = boot_samples(len(fx["weights"]), 150, 21, fx["weights"])
    start = np.array(fx["trees"][7]["back"], dtype=np.int32)
    e, o, se, so = run_both(engine, po, fx, start, samples, 2)
    assert_same(e, o, se, so)


def test_a_thousand_samples_match_the_oracle(mods):
    """-bb 1000 (the smallest count the reference accepts, tools.cpp:2106-2118): 1000 bootstrap samples on a fixture against the
    oracle's saveCurrentTree restatement -- every array of the bookkeeping, the moves, the tie draws"""
    engine, po = mods
    fx = load_fixture("dna_48")
    samples = boot_samples(len(fx["weights"]), 1000, 77, fx["weights"])
    start = np.array(fx["trees"][1]["back"], dtype=np.int32)
    e, o, se, so = run_both(engine, po, fx, start, samples, 13)
    assert_same(e, o, se, so)
    assert len(e.ufboot_state()[0]) == 1000


def test_heavy_weights_use_a_second_plane(mods):
    engine, po = mods
    fx = load_fixture("dna_clean")
    samples = boot_samples(len(fx["weights"]), 18, 2, fx["weights"], heavy=True)
    assert samples.max() > 127
    start = np.array(fx["trees"][0]["back"], dtype=np.int32)
    e, o, se, so = run_both(engine, po, fx, start, samples, 5)
    assert_same(e, o, se, so)


def test_weights_above_16383_use_a_third_plane(mods):
    """a sample's pattern weights are u16 (boot_samples_pars, iqtree.cpp:230-233): 7-bit digits on the int8 matrix cores, three planes"""
    engine, po = mods
    fx = load_fixture("dna_ambig")
    samples = boot_samples(len(fx["weights"]), 9, 4, fx["weights"])
    rng = np.random.default_rng(3)
    for b in range(len(samples)):
        samples[b, rng.integers(0, samples.shape[1], size=3)] = [16384, 40000, 65535]
    start = np.array(fx["trees"][4]["back"], dtype=np.int32)
    e, o, se, so = run_both(engine, po, fx, start, samples, 6)
    assert_same(e, o, se, so)


@pytest.mark.parametrize("n_samples", [1, 2, 129])
def test_sample_counts_around_the_tile_edges(mods, n_samples):
    engine, po = mods
    fx = load_fixture("aa")
    samples = boot_samples(len(fx["weights"]), n_samples, 6, fx["weights"])
    start = np.array(fx["trees"][3]["back"], dtype=np.int32)
    e, o, se, so = run_both(engine, po, fx, start, samples, 8)
    assert_same(e, o, se, so)


def test_keep_all_sites_and_small_radius(mods):
    engine, po = mods
    fx = load_fixture("dna_ambig")
    samples = boot_samples(len(fx["weights"]), 12, 4, fx["weights"])
    start = np.array(fx["trees"][4]["back"], dtype=np.int32)
    e, o, se, so = run_both(engine, po, fx, start, samples, 8, maxtrav=3, keep_all=True)
    assert_same(e, o, se, so)


@pytest.mark.parametrize("name,maxtrav", [("dna_clean", 8), ("aa", 7)])
def test_radius_above_six_uses_the_deep_walk_kernel(mods, name, maxtrav):
    engine, po = mods
    fx = load_fixture(name)
    samples = boot_samples(len(fx["weights"]), 10, 6, fx["weights"])
    start = np.array(fx["trees"][5]["back"], dtype=np.int32)
    e, o, se, so = run_both(engine, po, fx, start, samples, 13, maxtrav=maxtrav)
    assert_same(e, o, se, so)


@pytest.mark.parametrize("name,maxtrav,opts", [("dna_clean", 9, {}), ("dna_clean", 13, {"deep_scratch_kwords": 64}), ("dna_ambig", 200, {"ufb_pipe": 0}),
                                               ("aa", 9, {}), ("aa", 40, {"deep_scratch_kwords": 256}), ("dna_dups", 11, {"ufb_fast": 0})])
def test_any_radius_under_the_tracker(mods, name, maxtrav, opts):
    """-spr_rad above 8 with -bb (rearrangeParsimony takes any radius, sprparsimony.cpp:2259-2376): the masked scans come from
    k_scan_walk_deep, which parks the up-vectors of its levels in HBM scratch -- also where a small scratch cuts a batch into
    several launches -- and every book equals the oracle's"""
    engine, po = mods
    fx = load_fixture(name)
    samples = boot_samples(len(fx["weights"]), 12, 5, fx["weights"])
    start = np.array(fx["trees"][4]["back"], dtype=np.int32)
    e, o, se, so = run_both(engine, po, fx, start, samples, 17, maxtrav=maxtrav, opts=opts)
    assert_same(e, o, se, so)
    # a later iteration under a cut-off, same radius
    cut = e.ufboot_next_cutoff(10)
    assert cut == o.ufboot_next_cutoff(10)
    back2 = np.array(fx["trees"][6]["back"], dtype=np.int32)
    for x in (e, o):
        x.ufboot_set_cutoff(cut)
        x.set_tree(back2)
    assert e.optimize_spr(1, maxtrav) == o.optimize_spr(1, maxtrav)
    assert (e.get_tree() == o.get_tree()).all() and e.tie_state() == o.tie_state()
    assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
    assert [a.tolist() for a in e.ufboot_state()] == [a.tolist() for a in o.ufboot_state()]


def test_batched_refinement_at_a_long_radius(mods):
    """mpf_ufboot_refine_sweep above 8 levels == set_weights + optimize_spr per sample (stable <=> no move)"""
    engine, po = mods
    fx = load_fixture("dna_ambig")
    B = 7
    samples = boot_samples(len(fx["weights"]), B, 3, fx["weights"])
    start = np.array(fx["trees"][3]["back"], dtype=np.int32)
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    e.seed_ties(engine.TIE_RANDOM, 0)
    e.ufboot_attach(samples, 0.5)
    e.set_tree(start)
    seeds = np.arange(1, B + 1)
    sc, stable, first = e.ufboot_refine_sweep(12, seeds)
    e.ufboot_detach()
    solo = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    for b in range(B):
        for x, mode in ((solo, engine.TIE_RANDOM), (o, po.TIE_RANDOM)):
            x.set_weights(samples[b].astype(np.int32))
            x.seed_ties(mode, int(seeds[b]))
            x.set_tree(start)
        solo.reset_node_order(); o.reset_nodep()
        s0 = solo.score_tree()
        assert s0 == sc[b] == o.score_tree()
        o.trace(True)
        assert solo.optimize_spr(1, 12) == o.optimize_spr(1, 12)
        assert bool(stable[b]) == (len(solo.moves()[0]) == 0)


@pytest.mark.parametrize("name,kind,radius,chunk", [("dna_ambig", "sym", 6, 0), ("dna_48", "sym", 6, 7), ("aa", "sym", 5, 0), ("dna_clean", "asym", 6, 3), ("aa", "asym", 9, 0),
                                                    ("dna_dups", "big", 4, 0)])
def test_batched_refinement_on_the_weighted_engine(mods, name, kind, radius, chunk):
    """mpf_ufboot_refine_sweep on the weighted (-cost) engine: the first sweep of every sample's climb from one topology out of ONE
    scan with per-pattern lengths + bit-plane products == set_weights + optimize_spr per sample (score of the start tree under
    the sample, stable <=> the climb makes no move, the first move's visit), also under an asymmetric matrix and 32-bit costs"""
    engine, po = mods
    fx = load_fixture(name)
    S = 4 if fx["datatype"] == 0 else 20
    rng = np.random.default_rng(8)
    m = rng.integers(1, 4000 if kind == "big" else 6, size=(S, S))
    cost = (np.triu(m, 1) + np.triu(m, 1).T).astype(np.uint32)
    if kind == "asym":
        cost[np.triu_indices(S, 1)] += 2
    B = 9
    samples = boot_samples(len(fx["weights"]), B, 3, fx["weights"])
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    if chunk:
        e.set_option("refine_chunk", chunk)
    e.seed_ties(engine.TIE_RANDOM, 0)
    e.ufboot_attach(samples, 0.5)
    seeds = np.arange(11, 11 + B)
    solo = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    n_stable = 0
    for start in (np.array(fx["trees"][3]["back"], dtype=np.int32), None):
        if start is None:                               # ... and from a local optimum: most samples stable
            solo.set_weights(fx["weights_np"]); solo.set_tree(np.array(fx["trees"][3]["back"], dtype=np.int32)); solo.seed_ties(engine.TIE_RANDOM, 2)
            solo.optimize_spr(1, radius)
            start = solo.get_tree()
        e.reset_node_order()
        e.set_tree(start)
        sc, stable, first = e.ufboot_refine_sweep(radius, seeds)
        for b in range(B):
            for x, mode in ((solo, engine.TIE_RANDOM), (o, po.TIE_RANDOM)):
                x.set_weights(samples[b].astype(np.int32))
                x.seed_ties(mode, int(seeds[b]))
                x.set_tree(start)
            solo.reset_node_order(); o.reset_nodep()
            assert solo.score_tree() == sc[b] == o.score_tree()
            o.trace(True)
            assert solo.optimize_spr(1, radius) == o.optimize_spr(1, radius)
            moved = len(solo.moves()[0]) > 0
            assert bool(stable[b]) == (not moved), (b, int(first[b]))
            n_stable += bool(stable[b])
    assert n_stable > 0 or kind == "asym"          # (under an asymmetric matrix a tree keeps moving: its length depends on the edge)


def test_first_best_tie_rule(mods):
    """PLL-original tie rule for the SPR part (MPF_TIE_FIRST): the bookkeeping still draws its own ties"""
    engine, po = mods
    fx = load_fixture("dna_dups")
    samples = boot_samples(len(fx["weights"]), 14, 8, fx["weights"])
    start = np.array(fx["trees"][6]["back"], dtype=np.int32)
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    for x, mode in ((e, engine.TIE_FIRST), (o, po.TIE_FIRST)):
        x.set_tree(start)
        x.seed_ties(mode, 3)
        x.ufboot_attach(samples)
    # the oracle's first-best mode skips the pre-evaluate of the prune node (PLL original); compare what both define
    se, so = e.optimize_spr(1, 6), o.optimize_spr(1, 6)
    le, ce, te = e.ufboot_state()
    lo, co, to = o.ufboot_state()
    if se == so and (e.get_tree() == o.get_tree()).all():
        assert le.tolist() == lo.tolist()
    chk = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    chk.enable_persite(True)
    for b in range(len(le)):
        s = chk.score_tree(e.ufboot_tree(int(te[b])))
        ptn, _ = chk.pattern_scores()
        assert -(ptn.astype(np.int64) * samples[b]).sum() == le[b]


def test_online_phase_then_refinement_matches_oracle(mods):
    """-bb end of run: boot trees of the online phase refined per sample (IQTree::optimizeBootTrees, default branch)"""
    engine, po = mods
    from mpboot_amd import bootstrap

    fx = load_fixture("dna_ambig")
    B = 9
    samples = boot_samples(len(fx["weights"]), B, 12, fx["weights"])
    start = np.array(fx["trees"][2]["back"], dtype=np.int32)
    e, o, se, so = run_both(engine, po, fx, start, samples, 4)
    assert_same(e, o, se, so)
    _l, _c, te = e.ufboot_state()
    boot_trees = [e.ufboot_tree(int(t)) for t in te]
    # engine side: two engines on the GPU, oracle side: the same call on the CPU stand-in
    e2 = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    e3 = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    sc_e, tr_e = bootstrap.refine_boot_trees([e2, e3], samples, boot_trees, 77, 6)

    class OracleAsEngine:
        def __init__(self):
            self.o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])

        def set_weights(self, w):
            self.o.set_weights(w)

        def seed_ties(self, mode, seed):
            self.o.seed_ties(mode, seed)

        def reset_node_order(self):
            self.o.reset_nodep()

        def set_tree(self, back):
            self.o.set_tree(back)

        def optimize_spr(self, a, b):
            return self.o.optimize_spr(a, b)

        def get_tree(self):
            return self.o.get_tree()

    sc_o, tr_o = bootstrap.refine_boot_trees(OracleAsEngine(), samples, boot_trees, 77, 6)
    assert sc_e.tolist() == sc_o.tolist()
    for b in range(B):
        assert (tr_e[b] == tr_o[b]).all()
        assert sc_e[b] <= -_l[b]                    # refinement never makes a sample's tree worse under its own weights


SHARD_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["MPF_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MPF_ROOT"], "tests"))
import torch.distributed as dist
from helpers import load_fixture, same_topology
from mpboot_amd import engine, shard
dist.init_process_group("gloo")
rank, ws = shard.world()
fx = load_fixture(os.environ["MPF_FX"])
P = len(fx["weights"])
w = np.asarray(fx["weights"], dtype=np.float64)
samples = np.random.default_rng(31).multinomial(int(w.sum()), w / w.sum(), size=23).astype(np.uint16)
rule = os.environ.get("MPF_RULE", "default")
cost = None
if rule.startswith("weighted"):
    S = 20 if os.environ["MPF_FX"] == "aa" else 4
    c = np.random.default_rng(2).integers(1, 4, size=(S, S))
    cost = (np.triu(c, 1) + np.triu(c, 1).T).astype(np.uint32)
    if rule == "weighted_asym":
        cost[np.triu_indices(S, 1)] += 2
e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
if rank == 1:
    # a different past on this engine (its batch-size estimate, node order ...) must not change how it cuts the climb
    e.set_tree(np.array(fx["trees"][0]["back"], dtype=np.int32))
    e.seed_ties(engine.TIE_RANDOM, 5)
    e.optimize_spr(1, 6)
    e.optimize_spr(1, 6)
    e.set_option("scan_batch", 4)      # same option on both ranks below; here it only shapes this rank's history
    e.optimize_spr(1, 6)
    e.reset_node_order()
e.set_option("scan_batch", 4)          # small batches: several exchanges per climb
e.set_tree(np.array(fx["trees"][3]["back"], dtype=np.int32))
e.seed_ties(engine.TIE_RANDOM, 19)
e.ufboot_attach(samples, 0.5, shard=(rank, ws))
if rule == "weighted_mulhits":
    e.ufboot_set_mulhits(True)
if rule == "storetrees":
    e.ufboot_set_store_trees(True)
if rule == "topboot":
    e.ufboot_set_mulhits(True); e.ufboot_set_topboot(3)
if rule == "distinct":
    e.ufboot_set_distinct_iter(2); e.ufboot_set_iteration(1)
s = e.optimize_spr(1, 6)
cut = e.ufboot_next_cutoff(10)
e.ufboot_set_cutoff(cut)
if rule == "distinct":
    e.ufboot_set_iteration(2)
e.set_tree(np.array(fx["trees"][5]["back"], dtype=np.int32))
s2 = e.optimize_spr(1, 6)
logl, cnt, tr = e.ufboot_state()
res = {"s": [s, s2], "moves": [x.tolist() for x in e.moves()], "logl": logl.tolist(), "cnt": cnt.tolist(), "tr": tr.tolist(),
       "saved": e.ufboot_tree_logl().tolist(), "draws": e.ufboot_counters()["tie_draws"],
       "trees": {str(t): e.ufboot_tree(int(t)).tolist() for t in sorted(set(tr.tolist())) if t >= 0},
       "tops": [list(map(list, e.ufboot_sample_top(b)[0])) + [e.ufboot_sample_top(b)[1]] for b in range(len(samples))] if rule in ("topboot", "distinct") else [],
       "dups": e.ufboot_duplicates()}
allr = [None] * ws
dist.all_gather_object(allr, res)
if rank == 0:
    assert all(r == allr[0] for r in allr), "ranks disagree"
    print("RESULT " + json.dumps(res))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("rule", ["default", "topboot", "distinct", "storetrees", "weighted", "weighted_mulhits", "weighted_asym"])
@pytest.mark.parametrize("name", ["dna_ambig", "aa"])
def test_sample_sharded_online_phase_equals_the_unsharded_run(mods, tmp_path, name, rule):
    """two ranks (sharing this GPU) hold half of the samples each and exchange their events per batch: every rank must
    end with the state of the single-engine run -- scores, moves, saved trees, boot arrays, topologies, draw count"""
    import json
    import socket
    import subprocess
    import sys

    from helpers import ROOT

    engine, po = mods
    fx = load_fixture(name)
    w = np.asarray(fx["weights"], dtype=np.float64)
    samples = np.random.default_rng(31).multinomial(int(w.sum()), w / w.sum(), size=23).astype(np.uint16)
    cost = None
    if rule.startswith("weighted"):                  # the weighted (-cost) tracker, sample-sharded as well (round 5)
        S = 20 if name == "aa" else 4
        c = np.random.default_rng(2).integers(1, 4, size=(S, S))
        cost = (np.triu(c, 1) + np.triu(c, 1).T).astype(np.uint32)
        if rule == "weighted_asym":                  # (every visit's own row and length: device events on every rank)
            cost[np.triu_indices(S, 1)] += 2
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    e.set_tree(np.array(fx["trees"][3]["back"], dtype=np.int32))
    e.seed_ties(engine.TIE_RANDOM, 19)
    e.ufboot_attach(samples)
    if rule == "weighted_mulhits":
        e.ufboot_set_mulhits(True)
    if rule == "storetrees":
        e.ufboot_set_store_trees(True)
    if rule == "topboot":
        e.ufboot_set_mulhits(True)
        e.ufboot_set_topboot(3)
    if rule == "distinct":
        e.ufboot_set_distinct_iter(2)
        e.ufboot_set_iteration(1)
    s = e.optimize_spr(1, 6)
    e.ufboot_set_cutoff(e.ufboot_next_cutoff(10))
    if rule == "distinct":
        e.ufboot_set_iteration(2)
    e.set_tree(np.array(fx["trees"][5]["back"], dtype=np.int32))
    s2 = e.optimize_spr(1, 6)
    logl, cnt, tr = e.ufboot_state()
    script = tmp_path / "worker.py"
    script.write_text(SHARD_WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MPF_ROOT=ROOT, MPF_FX=name, MPF_RULE=rule)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    got = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert got["s"] == [s, s2]
    assert got["moves"] == [x.tolist() for x in e.moves()]
    assert got["logl"] == logl.tolist() and got["cnt"] == cnt.tolist() and got["tr"] == tr.tolist()
    assert got["saved"] == e.ufboot_tree_logl().tolist()
    assert got["draws"] == e.ufboot_counters()["tie_draws"]
    for t, back in got["trees"].items():
        assert back == e.ufboot_tree(int(t)).tolist()
    assert got["dups"] == e.ufboot_duplicates()
    if rule in ("topboot", "distinct"):
        want = [list(map(list, e.ufboot_sample_top(b)[0])) + [e.ufboot_sample_top(b)[1]] for b in range(len(samples))]
        assert got["tops"] == want


def test_unsupported_configurations_fail_loudly(mods):
    engine, po = mods
    fx = load_fixture("dna_clean")
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    samples = boot_samples(len(fx["weights"]), 4, 1)
    with pytest.raises(engine.MpfError):
        e.ufboot_attach(samples, 2.0)
    with pytest.raises(engine.MpfError):
        e.ufboot_next_cutoff(10)
    cost = (1 - np.eye(4)).astype(np.uint32)
    s = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    s.ufboot_attach(samples)                       # the weighted engine keeps the bookkeeping too (sample-sharded as well: see
    s.ufboot_detach()                              # test_sample_sharded_online_phase_equals_the_unsharded_run) ...
    asym = cost.copy()
    asym[0, 1] = 2                                 # ... also under an asymmetric matrix (test_weighted_tracker_under_an_asymmetric_matrix)
    a = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=asym)
    a.ufboot_attach(samples)
    a.ufboot_detach()
    # the bookkeeping lives in the device-walked scan: the other scan mode refuses instead of skipping it (any radius is served:
    # test_any_radius_under_the_tracker)
    e.ufboot_attach(samples)
    e.set_tree(np.array(fx["trees"][0]["back"], dtype=np.int32))
    e.set_option("scan_mode", 0)
    with pytest.raises(engine.MpfError):
        e.optimize_spr(1, 6)
    e.set_option("scan_mode", 1)
    assert e.optimize_spr(1, 6) > 0
    # weights that leave an attach-time pattern without a site (mpboot's ratchet never does that) rest the tracker: a climb
    # on them leaves its state alone, and the attach-time weights resume it
    n_saved = len(e.ufboot_tree_logl())
    other = fx["weights_np"].copy()
    other[::2] = 0
    e.set_weights(other)
    e.set_tree(np.array(fx["trees"][1]["back"], dtype=np.int32))
    assert e.optimize_spr(1, 6) > 0
    assert len(e.ufboot_tree_logl()) == n_saved
    e.set_weights(fx["weights_np"])
    e.set_tree(np.array(fx["trees"][1]["back"], dtype=np.int32))
    assert e.optimize_spr(1, 6) > 0
    assert len(e.ufboot_tree_logl()) > n_saved


@pytest.mark.parametrize("cut", ["none", "top10", "top90", "all_fail"])
@pytest.mark.parametrize("name", ["dna_clean", "dna_ambig", "dna_dups", "aa", "dna_48"])
def test_ratchet_climbs_are_booked_like_the_reference(mods, name, cut):
    """a search iteration on the original alignment, a ratchet iteration (pattern frequencies raised as createPerturbAlignment
    raises them) and the climb back: saveCurrentTree's state after every climb == the oracle's literal restatement of
    iqtree.cpp:3283-3295 (cur_logl = REPS of the not-yet-refreshed _pattern_pars against original_sample), under no
    cut-off, a loose one, a tight one (the bookkeeping of the ratchet climb stops at the first booked tree that fails) and
    one nothing passes"""
    engine, po = mods
    fx = load_fixture(name)
    w0 = fx["weights_np"]
    samples = boot_samples(len(w0), 30, 17, fx["weights"])
    rng = np.random.default_rng(5)
    pert = (w0 * (1 + (rng.random(len(w0)) < 0.3) * rng.integers(1, 3, size=len(w0)))).astype(np.int32)
    e = engine.FitchEngine(fx["codes_np"], w0, datatype=fx["datatype"])
    o = po.Oracle(fx["codes_np"], w0, datatype=fx["datatype"])
    t = [np.array(fx["trees"][k]["back"], dtype=np.int32) for k in (1, 4, 6)]

    def same():
        assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
        assert [x.tolist() for x in e.ufboot_state()] == [x.tolist() for x in o.ufboot_state()]
        assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws()
        assert (e.get_tree() == o.get_tree()).all()
        for ti in sorted(set(e.ufboot_state()[2].tolist())):
            if ti >= 0:
                assert same_topology(e.ufboot_tree(ti), o.ufboot_tree(ti), fx["n"])

    for x, mode in ((e, engine.TIE_RANDOM), (o, po.TIE_RANDOM)):
        x.seed_ties(mode, 23)
        x.ufboot_attach(samples)
        x.set_tree(t[0])
    assert e.optimize_spr(1, 6) == o.optimize_spr(1, 6)
    same()
    n0 = len(o.ufboot_tree_logl())
    logl = np.sort(o.ufboot_tree_logl())
    cutoff = {"none": 0.0, "top10": float(logl[int(0.9 * len(logl))]), "top90": float(logl[int(0.1 * len(logl))]),
              "all_fail": float(logl[-1]) + 50.0}[cut]
    for x in (e, o):
        x.ufboot_set_cutoff(cutoff)
        x.set_weights(pert)                      # the ratchet iteration's alignment
        x.set_tree(t[1])
    tests0 = o.counters()[2]
    assert e.optimize_spr(1, 6) == o.optimize_spr(1, 6)
    same()
    booked = len(o.ufboot_tree_logl()) - n0
    if cut == "none":
        extra = booked - (o.counters()[2] - tests0)        # every insertion test of the climb is booked, and the current tree
        assert extra > 0 and extra % (2 * fx["n"] - 2) == 0    # once per prune-node visit (sprparsimony.cpp:2285-2289)
        # ... the first one with the ORIGINAL-alignment length of the tree the climb started from
        ref = po.Oracle(fx["codes_np"], w0, datatype=fx["datatype"])
        assert -o.ufboot_tree_logl()[n0] == ref.score_tree(t[1])
    if cut == "all_fail":
        assert booked == 0
    for x in (e, o):
        x.set_weights(w0)
        x.set_tree(t[2])
    assert e.optimize_spr(1, 6) == o.optimize_spr(1, 6)
    same()
    assert o.ufboot_bad() == 0


def test_no_hclimb1_bb_leaves_ratchet_climbs_unbooked(mods):
    engine, po = mods
    fx = load_fixture("dna_clean")
    w0 = fx["weights_np"]
    samples = boot_samples(len(w0), 12, 3, fx["weights"])
    e = engine.FitchEngine(fx["codes_np"], w0, datatype=fx["datatype"])
    e.ufboot_attach(samples)
    e.ufboot_set_ratchet_booking(False)
    e.set_tree(np.array(fx["trees"][0]["back"], dtype=np.int32))
    e.optimize_spr(1, 6)
    n_saved = len(e.ufboot_tree_logl())
    e.set_weights((w0 * 2).astype(np.int32))
    e.set_tree(np.array(fx["trees"][2]["back"], dtype=np.int32))
    e.optimize_spr(1, 6)
    assert len(e.ufboot_tree_logl()) == n_saved
    e.set_weights(w0)
    e.set_tree(np.array(fx["trees"][2]["back"], dtype=np.int32))
    e.optimize_spr(1, 6)
    assert len(e.ufboot_tree_logl()) > n_saved


@pytest.mark.parametrize("cut", ["none", "top50"])
@pytest.mark.parametrize("opts", [{}, {"scan_batch": 5, "split_below": 0}])
@pytest.mark.parametrize("name", ["dna_clean", "dna_ambig", "dna_dups", "aa", "dna_48"])
def test_mulhits_rule_matches_oracle(mods, name, cut, opts):
    """-mulhits (params->multiple_hits, iqtree.cpp:3498-3540) over a normal climb, a ratchet climb and the climb back: the
    per-sample sets of equally good trees (with the reference's one-index-per-topology rule), boot_logl, the booked list and
    the topologies the sets name == the oracle's; no random draw is spent, so the SPR trajectory differs from the default
    rule's and must still be the oracle's"""
    engine, po = mods
    fx = load_fixture(name)
    w0 = fx["weights_np"]
    samples = boot_samples(len(w0), 24, 29, fx["weights"])
    rng = np.random.default_rng(8)
    pert = (w0 * (1 + (rng.random(len(w0)) < 0.3))).astype(np.int32)
    e = engine.FitchEngine(fx["codes_np"], w0, datatype=fx["datatype"])
    o = po.Oracle(fx["codes_np"], w0, datatype=fx["datatype"])
    for k, v in opts.items():
        e.set_option(k, v)
    t = [np.array(fx["trees"][k]["back"], dtype=np.int32) for k in (2, 5, 7)]
    for x, mode in ((e, engine.TIE_RANDOM), (o, po.TIE_RANDOM)):
        x.seed_ties(mode, 31)
        x.ufboot_attach(samples)
        x.ufboot_set_mulhits(True)
    largest = 0
    for k, w in enumerate((w0, pert, w0)):
        for x in (e, o):
            x.set_weights(w)
            x.set_tree(t[k])
        o.trace(True)
        assert e.optimize_spr(1, 6) == o.optimize_spr(1, 6)
        assert [x.tolist() for x in e.moves()] == [x.tolist() for x in o.get_moves()]
        assert (e.get_tree() == o.get_tree()).all()
        assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
        assert e.ufboot_state()[0].tolist() == o.ufboot_state()[0].tolist()
        assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws() == 0
        for b in range(len(samples)):
            got = e.ufboot_sample_trees(b)
            assert got == o.ufboot_sample_trees(b) and len(got) >= 1
            largest = max(largest, len(got))
            for ti in got:
                assert same_topology(e.ufboot_tree(ti), o.ufboot_tree(ti), fx["n"])
        if k == 0 and cut == "top50":
            logl = np.sort(o.ufboot_tree_logl())
            for x in (e, o):
                x.ufboot_set_cutoff(float(logl[len(logl) // 2]))
    assert o.ufboot_bad() == 0


def test_mulhits_must_be_chosen_before_the_first_booking(mods):
    engine, _ = mods
    fx = load_fixture("dna_clean")
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    with pytest.raises(engine.MpfError):
        e.ufboot_set_mulhits(True)                               # no tracker
    e.ufboot_attach(boot_samples(len(fx["weights"]), 4, 1, fx["weights"]))
    with pytest.raises(engine.MpfError):
        e.ufboot_sample_trees(0)                                 # default rule in force
    e.set_tree(np.array(fx["trees"][0]["back"], dtype=np.int32))
    e.optimize_spr(1, 3)
    with pytest.raises(engine.MpfError):
        e.ufboot_set_mulhits(True)                               # trees already booked under the default rule


@pytest.mark.parametrize("alphabet", ["DNA", "AA"])
def test_tiny_tree_whose_prune_nodes_have_no_insertion_test(mods, alphabet):
    """five taxa at radius 1: no prune node has a candidate branch, yet rearrangeParsimony books the current tree at every
    visit (sprparsimony.cpp:2285-2289) -- a batch without a scan launch (regression: the mask / info buffers of the tracker
    were only provided by that launch)"""
    engine, po = mods
    from mpboot_amd import synth, trees

    letters, _ = synth.synth_alignment(5, 150, alphabet, 0.25, seed=4)
    codes = synth.letters_to_codes(letters, alphabet)
    dt_e, dt_o = (engine.AA, po.AA) if alphabet == "AA" else (engine.DNA, po.DNA)
    e = engine.FitchEngine(codes, datatype=dt_e)
    o = po.Oracle(codes, datatype=dt_o)
    if o.num_informative == 0:
        pytest.skip("no informative pattern")
    back = trees.random_topology(5, np.random.default_rng(2))
    samples = boot_samples(codes.shape[1], 6, 5)
    for x in (e, o):
        x.set_tree(back)
        x.seed_ties(1, 9)
        x.ufboot_attach(samples)
    for radius in (1, 2, 6):
        o.trace(True)
        assert e.optimize_spr(1, radius) == o.optimize_spr(1, radius)
        assert [a.tolist() for a in e.moves()] == [a.tolist() for a in o.get_moves()]
        assert [x.tolist() for x in e.ufboot_state()] == [x.tolist() for x in o.ufboot_state()]
        assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
        assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws()
    assert len(o.ufboot_tree_logl()) > 0


@pytest.mark.parametrize("name,radius", [("dna_clean", 14), ("aa", 8)])
def test_weighted_tracker_at_a_long_radius(mods, name, radius):
    """-cost with -bb above the levels the weighted scan keeps in registers: the per-pattern lengths come from k_snk_scan_deep"""
    engine, po = mods
    fx = load_fixture(name)
    S = 4 if fx["datatype"] == 0 else 20
    m = np.random.default_rng(3).integers(1, 6, size=(S, S))
    cost = (np.triu(m, 1) + np.triu(m, 1).T).astype(np.uint32)
    samples = boot_samples(len(fx["weights"]), 11, 4, fx["weights"])
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"], cost=cost)
    e.set_option("deep_scratch_kwords", 128)
    for x in (e, o):
        x.seed_ties(1, 5)
        x.ufboot_attach(samples)
        x.set_tree(np.array(fx["trees"][2]["back"], dtype=np.int32))
    o.trace(True)
    assert e.optimize_spr(1, radius) == o.optimize_spr(1, radius)
    assert [a.tolist() for a in e.moves()] == [a.tolist() for a in o.get_moves()]
    assert (e.get_tree() == o.get_tree()).all()
    assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
    assert [x.tolist() for x in e.ufboot_state()] == [x.tolist() for x in o.ufboot_state()]
    assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws()


@pytest.mark.parametrize("mode", ["default", "cutoff", "mulhits", "storetrees"])
@pytest.mark.parametrize("name", ["dna_clean", "dna_ambig", "aa", "dna_48"])
def test_weighted_tracker_under_an_asymmetric_matrix(mods, name, mode):
    """-cost with a matrix that is not symmetric, under -bb: the current tree has another length and other per-pattern lengths at
    every prune node's visit (evaluateParsimony(p) roots it at that node's edge, sprparsimony.cpp:2285) -- the scan writes both
    into the visit's slot.  A normal climb, a ratchet climb, the climb back: every observable == the oracle's"""
    engine, po = mods
    fx = load_fixture(name)
    w0 = fx["weights_np"]
    S = 4 if fx["datatype"] == 0 else 20
    rng = np.random.default_rng(21)
    cost = rng.integers(1, 7, size=(S, S)).astype(np.uint32)
    cost[np.triu_indices(S, 1)] += 2
    np.fill_diagonal(cost, 0)
    samples = boot_samples(len(w0), 14, 9, fx["weights"])
    pert = (w0 * (1 + (rng.random(len(w0)) < 0.3))).astype(np.int32)
    e = engine.FitchEngine(fx["codes_np"], w0, datatype=fx["datatype"], cost=cost)
    o = po.Oracle(fx["codes_np"], w0, datatype=fx["datatype"], cost=cost)
    t = [np.array(fx["trees"][k]["back"], dtype=np.int32) for k in (1, 4, 6)]
    for x in (e, o):
        x.seed_ties(1, 17)
        x.ufboot_attach(samples)
        if mode == "mulhits":
            x.ufboot_set_mulhits(True)
        if mode == "storetrees":
            x.ufboot_set_store_trees(True)
    for k, w in enumerate((w0, pert, w0)):
        for x in (e, o):
            x.set_weights(w)
            x.set_tree(t[k])
        o.trace(True)
        assert e.optimize_spr(1, 5) == o.optimize_spr(1, 5)
        assert [a.tolist() for a in e.moves()] == [a.tolist() for a in o.get_moves()]
        assert (e.get_tree() == o.get_tree()).all()
        assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
        assert [x.tolist() for x in e.ufboot_state()] == [x.tolist() for x in o.ufboot_state()]
        assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws()
        if mode == "mulhits":
            for b in range(len(samples)):
                assert e.ufboot_sample_trees(b) == o.ufboot_sample_trees(b)
        if k == 0 and mode == "cutoff":
            logl = np.sort(o.ufboot_tree_logl())
            for x in (e, o):
                x.ufboot_set_cutoff(float(logl[len(logl) // 2]))
    assert len(set(o.ufboot_tree_logl().tolist())) > 10
    assert len(o.ufboot_tree_logl()) > 50


@pytest.mark.parametrize("mode", ["default", "cutoff", "mulhits"])
@pytest.mark.parametrize("name,big", [("dna_clean", False), ("dna_ambig", True), ("aa", False), ("dna_48", False)])
def test_weighted_engine_bookkeeping_matches_oracle(mods, name, big, mode):
    """-cost with -bb: saveCurrentTree on the weighted (Sankoff) engine.  pllComputePatternParsimony dispatches to
    pllComputeSankoffPatternParsimony (sprparsimony.cpp:3341-3355); here the scan writes every tentative tree's per-pattern
    lengths, bit planes of them are multiplied with the sample weights on the matrix cores.  A normal climb, a re-weighted
    (ratchet) climb and the climb back: every observable == the oracle's"""
    engine, po = mods
    fx = load_fixture(name)
    w0 = fx["weights_np"]
    S = 4 if fx["datatype"] == 0 else 20
    rng = np.random.default_rng(12)
    m = rng.integers(1, 4000 if big else 6, size=(S, S))
    cost = (np.triu(m, 1) + np.triu(m, 1).T).astype(np.uint32)
    samples = boot_samples(len(w0), 20, 41, fx["weights"])
    pert = (w0 * (1 + (rng.random(len(w0)) < 0.3))).astype(np.int32)
    e = engine.FitchEngine(fx["codes_np"], w0, datatype=fx["datatype"], cost=cost)
    o = po.Oracle(fx["codes_np"], w0, datatype=fx["datatype"], cost=cost)
    t = [np.array(fx["trees"][k]["back"], dtype=np.int32) for k in (1, 4, 6)]
    for x in (e, o):
        x.seed_ties(1, 17)
        x.ufboot_attach(samples)
        if mode == "mulhits":
            x.ufboot_set_mulhits(True)
    for k, w in enumerate((w0, pert, w0)):
        for x in (e, o):
            x.set_weights(w)
            x.set_tree(t[k])
        o.trace(True)
        assert e.optimize_spr(1, 5) == o.optimize_spr(1, 5)
        assert [a.tolist() for a in e.moves()] == [a.tolist() for a in o.get_moves()]
        assert (e.get_tree() == o.get_tree()).all()
        assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
        assert [x.tolist() for x in e.ufboot_state()] == [x.tolist() for x in o.ufboot_state()]
        assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws()
        if mode == "mulhits":
            for b in range(len(samples)):
                assert e.ufboot_sample_trees(b) == o.ufboot_sample_trees(b)
        else:
            for ti in sorted(set(e.ufboot_state()[2].tolist())):
                if ti >= 0:
                    assert same_topology(e.ufboot_tree(ti), o.ufboot_tree(ti), fx["n"])
        if k == 0 and mode == "cutoff":
            logl = np.sort(o.ufboot_tree_logl())
            for x in (e, o):
                x.ufboot_set_cutoff(float(logl[len(logl) // 2]))
    assert o.ufboot_bad() == 0
    assert len(o.ufboot_tree_logl()) > 50


@pytest.mark.parametrize("n_top", [1, 4, 10])
@pytest.mark.parametrize("engine_kind", ["fitch", "weighted"])
@pytest.mark.parametrize("name", ["dna_clean", "dna_dups", "aa", "dna_48"])
def test_mulhits_topboot_rule_matches_oracle(mods, name, engine_kind, n_top):
    """-mulhits -topboot N (params->store_top_boot_trees, iqtree.cpp:3542-3585): per sample the N best new trees, best first, and
    boot_threshold, over a normal climb, a ratchet climb and the climb back (with a cut-off from the second climb on) -- on both
    engines.  The device's event bound is the list's threshold at the start of the batch (fixed, not a running minimum)."""
    engine, po = mods
    fx = load_fixture(name)
    w0 = fx["weights_np"]
    cost = None
    if engine_kind == "weighted":
        S = 4 if fx["datatype"] == 0 else 20
        m = np.random.default_rng(3).integers(1, 6, size=(S, S))
        cost = (np.triu(m, 1) + np.triu(m, 1).T).astype(np.uint32)
    samples = boot_samples(len(w0), 16, 53, fx["weights"])
    rng = np.random.default_rng(4)
    pert = (w0 * (1 + (rng.random(len(w0)) < 0.3))).astype(np.int32)
    e = engine.FitchEngine(fx["codes_np"], w0, datatype=fx["datatype"], cost=cost)
    o = po.Oracle(fx["codes_np"], w0, datatype=fx["datatype"], cost=cost)
    t = [np.array(fx["trees"][k]["back"], dtype=np.int32) for k in (2, 5, 7)]
    for x in (e, o):
        x.seed_ties(1, 37)
        x.ufboot_attach(samples)
        x.ufboot_set_mulhits(True)
        x.ufboot_set_topboot(n_top)
    for k, w in enumerate((w0, pert, w0)):
        for x in (e, o):
            x.set_weights(w)
            x.set_tree(t[k])
        o.trace(True)
        assert e.optimize_spr(1, 5) == o.optimize_spr(1, 5)
        assert [a.tolist() for a in e.moves()] == [a.tolist() for a in o.get_moves()]
        assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
        assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws() == 0
        for b in range(len(samples)):
            got, want = e.ufboot_sample_top(b), o.ufboot_sample_top(b)
            assert got == want and len(got[0]) == n_top
            for ti, _ in got[0]:
                assert same_topology(e.ufboot_tree(ti), o.ufboot_tree(ti), fx["n"])
        if k == 0:
            logl = np.sort(o.ufboot_tree_logl())
            for x in (e, o):
                x.ufboot_set_cutoff(float(logl[len(logl) // 3]))
    assert o.ufboot_bad() == 0


@pytest.mark.parametrize("k", [1, 3])
@pytest.mark.parametrize("engine_kind", ["fitch", "weighted"])
@pytest.mark.parametrize("name", ["dna_clean", "dna_dups", "aa", "dna_48"])
def test_distinct_iter_top_boot_rule_matches_oracle(mods, name, engine_kind, k):
    """-distinct_iter_top_boot k (iqtree.cpp:3587-3680) over five search iterations (a ratchet climb among them, a cut-off from
    the third on): lists, their iterations, thresholds, boot_logl / boot_counts / boot_trees, the saved-tree list and the
    number of tie draws == the oracle's, on both engines"""
    engine, po = mods
    fx = load_fixture(name)
    w0 = fx["weights_np"]
    cost = None
    if engine_kind == "weighted":
        S = 4 if fx["datatype"] == 0 else 20
        m = np.random.default_rng(3).integers(1, 6, size=(S, S))
        cost = (np.triu(m, 1) + np.triu(m, 1).T).astype(np.uint32)
    samples = boot_samples(len(w0), 16, 59, fx["weights"])
    rng = np.random.default_rng(6)
    pert = (w0 * (1 + (rng.random(len(w0)) < 0.3))).astype(np.int32)
    e = engine.FitchEngine(fx["codes_np"], w0, datatype=fx["datatype"], cost=cost)
    o = po.Oracle(fx["codes_np"], w0, datatype=fx["datatype"], cost=cost)
    t = [np.array(fx["trees"][j]["back"], dtype=np.int32) for j in (2, 5, 7, 1, 3)]
    for x in (e, o):
        x.seed_ties(1, 43)
        x.ufboot_attach(samples)
        x.ufboot_set_distinct_iter(k)
    for it, w in enumerate((w0, pert, w0, w0, w0)):
        for x in (e, o):
            x.ufboot_set_iteration(it + 1)
            x.set_weights(w)
            x.set_tree(t[it])
        o.trace(True)
        assert e.optimize_spr(1, 5) == o.optimize_spr(1, 5)
        assert [a.tolist() for a in e.moves()] == [a.tolist() for a in o.get_moves()]
        assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
        assert [x.tolist() for x in e.ufboot_state()] == [x.tolist() for x in o.ufboot_state()]
        assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws()
        for b in range(len(samples)):
            assert e.ufboot_sample_top(b) == o.ufboot_sample_top(b)
            assert e.ufboot_sample_iters(b) == o.ufboot_sample_iters(b)
            for ti, _ in e.ufboot_sample_top(b)[0]:
                assert same_topology(e.ufboot_tree(ti), o.ufboot_tree(ti), fx["n"])
        if it == 1:
            logl = np.sort(o.ufboot_tree_logl())
            for x in (e, o):
                x.ufboot_set_cutoff(float(logl[len(logl) // 3]))
    assert o.ufboot_bad() == 0 and o.ufboot_draws() > 0


@pytest.mark.parametrize("rule", ["default", "mulhits", "topboot", "distinct"])
@pytest.mark.parametrize("engine_kind,opts", [("fitch", {}), ("fitch", {"scan_batch": 5, "split_below": 0}), ("weighted", {})])
@pytest.mark.parametrize("name", ["dna_clean", "dna_ambig", "dna_dups", "aa", "dna_48"])
def test_storetrees_matches_oracle(mods, name, engine_kind, opts, rule):
    """-storetrees (params->store_candidate_trees, iqtree.cpp:3302-3346) under every update rule, on both engines: a normal
    climb, a ratchet climb, then two climbs under a cut-off.  A topology met again is a duplicate (same count as the
    oracle's duplication_counter); on the ratchet climb some are booked again under their old index with an improved
    length, past the cut-off test.  Every observable == the oracle's."""
    engine, po = mods
    fx = load_fixture(name)
    w0 = fx["weights_np"]
    kw = {}
    if engine_kind == "weighted":
        S = 4 if fx["datatype"] == 0 else 20
        m = np.random.default_rng(3).integers(1, 6, size=(S, S))
        kw["cost"] = (np.triu(m, 1) + np.triu(m, 1).T).astype(np.uint32)
    samples = boot_samples(len(w0), 20, 13, fx["weights"])
    rng = np.random.default_rng(21)
    pert = (w0 * (1 + (rng.random(len(w0)) < 0.3))).astype(np.int32)
    e = engine.FitchEngine(fx["codes_np"], w0, datatype=fx["datatype"], **kw)
    o = po.Oracle(fx["codes_np"], w0, datatype=fx["datatype"], **kw)
    for k, v in opts.items():
        e.set_option(k, v)
    t = [np.array(fx["trees"][k]["back"], dtype=np.int32) for k in (2, 5, 7, 3)]
    for x in (e, o):
        x.seed_ties(1, 23)
        x.ufboot_attach(samples)
        x.ufboot_set_store_trees(True)
        if rule in ("mulhits", "topboot"):
            x.ufboot_set_mulhits(True)
        if rule == "topboot":
            x.ufboot_set_topboot(3)
        if rule == "distinct":
            x.ufboot_set_distinct_iter(2)
    radius = 5 if engine_kind == "weighted" else 6
    for it, w in enumerate((w0, pert, w0, w0)):
        for x in (e, o):
            if rule == "distinct":
                x.ufboot_set_iteration(it + 1)
            x.set_weights(w)
            x.set_tree(t[it])
        o.trace(True)
        assert e.optimize_spr(1, radius) == o.optimize_spr(1, radius)
        assert [a.tolist() for a in e.moves()] == [a.tolist() for a in o.get_moves()]
        assert (e.get_tree() == o.get_tree()).all()
        assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
        assert e.ufboot_duplicates() == o.ufboot_duplicates()
        assert [x.tolist() for x in e.ufboot_state()] == [x.tolist() for x in o.ufboot_state()]
        assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws()
        for b in range(len(samples)):
            if rule == "mulhits":
                assert e.ufboot_sample_trees(b) == o.ufboot_sample_trees(b)
            if rule in ("topboot", "distinct"):
                assert e.ufboot_sample_top(b) == o.ufboot_sample_top(b)
        if rule == "default":
            for ti in sorted(set(e.ufboot_state()[2].tolist())):
                if ti >= 0:
                    assert same_topology(e.ufboot_tree(ti), o.ufboot_tree(ti), fx["n"])
        if it == 1:
            logl = np.sort(o.ufboot_tree_logl())
            for x in (e, o):
                x.ufboot_set_cutoff(float(logl[len(logl) // 2]))
    assert o.ufboot_bad() == 0
    assert o.ufboot_duplicates() > 0


def test_storetrees_must_be_chosen_before_the_first_booking(mods):
    engine, _ = mods
    fx = load_fixture("dna_clean")
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    with pytest.raises(engine.MpfError):
        e.ufboot_set_store_trees(True)                           # no tracker
    e.ufboot_attach(boot_samples(len(fx["weights"]), 4, 1, fx["weights"]))
    e.set_tree(np.array(fx["trees"][0]["back"], dtype=np.int32))
    e.optimize_spr(1, 3)
    with pytest.raises(engine.MpfError):
        e.ufboot_set_store_trees(True)                           # trees already booked without the map


@pytest.mark.parametrize("opts", [{}, {"ufb_pipe": 0}, {"ufb_fast": 0}])
@pytest.mark.parametrize("rule,engine_kind", [("default", "fitch"), ("mulhits", "fitch"), ("distinct", "fitch"), ("storetrees", "fitch"), ("default", "weighted")])
@pytest.mark.parametrize("name", ["dna_dups", "aa", "dna_48"])
def test_cutoff_from_btrees_matches_oracle(mods, name, rule, engine_kind, opts):
    """-cutoff_from_btrees (tools.cpp:2442): boot_tree_orig_logl -- the logl under which every sample's tree was booked
    (iqtree.cpp:3523-3527 / :3617-3619 / :3716-3718; on ratchet climbs the value saveCurrentTree replaced cur_logl by) -- over
    normal / ratchet / normal climbs, and the next iteration's cut-off = its minimum (:1657-1660), fed back like the main loop
    does.  Every update rule, the weighted tracker, all ways through the tracked loop."""
    engine, po = mods
    fx = load_fixture(name)
    w0 = fx["weights_np"]
    rng = np.random.default_rng(11)
    samples = rng.multinomial(int(w0.sum()), w0 / w0.sum(), size=40).astype(np.uint16)
    pert = (w0 * (1 + (rng.random(len(w0)) < 0.3))).astype(np.int32)
    cost = None
    if engine_kind == "weighted":
        S = 20 if fx["datatype"] == engine.AA else 4
        c = np.random.default_rng(2).integers(1, 4, size=(S, S))
        cost = (np.triu(c, 1) + np.triu(c, 1).T).astype(np.uint32)
    e = engine.FitchEngine(fx["codes_np"], w0, datatype=fx["datatype"], cost=cost)
    o = po.Oracle(fx["codes_np"], w0, datatype=fx["datatype"], cost=cost)
    for k, v in opts.items():
        e.set_option(k, v)
    trees_ = [np.array(fx["trees"][k]["back"], dtype=np.int32) for k in (1, 3, 5)]
    for x, mode in ((e, engine.TIE_RANDOM), (o, po.TIE_RANDOM)):
        x.set_tree(trees_[0])
        x.seed_ties(mode, 13)
        x.ufboot_attach(samples)
        x.ufboot_set_cutoff_from_btrees(True)
        if rule == "mulhits":
            x.ufboot_set_mulhits(True)
        if rule == "distinct":
            x.ufboot_set_distinct_iter(2)
            x.ufboot_set_iteration(1)
        if rule == "storetrees":
            x.ufboot_set_store_trees(True)
    for k, (wgt, tree) in enumerate(((w0, trees_[0]), (pert, trees_[1]), (w0, trees_[2]))):
        for x in (e, o):
            x.set_weights(wgt)
            x.set_tree(tree)
            if rule == "distinct":
                x.ufboot_set_iteration(1 + k)
        assert e.optimize_spr(1, 6) == o.optimize_spr(1, 6)
        assert e.ufboot_orig_logl().tolist() == o.ufboot_orig_logl().tolist()
        ce, co = e.ufboot_next_cutoff(10), o.ufboot_next_cutoff(10)
        assert ce == co
        assert (ce == 0.0) == (rule == "mulhits")
        le, cne, te = e.ufboot_state()
        lo, cno, to = o.ufboot_state()
        assert le.tolist() == lo.tolist() and cne.tolist() == cno.tolist() and te.tolist() == to.tolist()
        assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
        assert e.tie_state() == o.tie_state()
        e.ufboot_set_cutoff(ce)
        o.ufboot_set_cutoff(co)
    assert o.ufboot_bad() == 0

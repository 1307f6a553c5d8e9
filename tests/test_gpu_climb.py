"""Device-resident SPR hill climb (k_climb, mpboot_amd/csrc/climb.hip) against the oracle and against the host-driven batches.

pllOptimizeSprParsimony (reference sprparsimony.cpp:3244-3319) is sequential: every accepted move depends on the one before.
The kernel keeps the whole sweep loop on the GPU -- enumeration in the reference's order, tie rules with the lcg64 stream,
topology edits -- so the test is the trajectory: accepted moves (remove record, insert record, length), final tree, final
length and the state of the tie stream afterwards, bit for bit.  Integer work: every comparison is exact.
"""
import numpy as np
import pytest

from helpers import FIXTURES, load_fixture, trace_tokens

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from mpboot_amd import engine, synth, trees
    from oracle import pyoracle as po
    return engine, po, synth, trees


def climb(engine, codes, dt, back, mode, tie, seed, radius, weights=None, **opts):
    e = engine.FitchEngine(codes, weights, datatype=dt)
    e.set_option("climb_device", mode)
    for k, v in opts.items():
        e.set_option(k, v)
    e.set_tree(back)
    e.seed_ties(tie, seed)
    s = e.optimize_spr(1, radius)
    return e, s, [list(map(int, m)) for m in zip(*e.moves())]


def oracle_climb(po, codes, dt, back, tie, seed, radius, weights=None):
    o = po.Oracle(codes, weights, datatype=dt)
    o.set_tree(back)
    o.seed_ties(tie, seed)
    if tie == po.TIE_FIRST:
        o.set_pre_evaluate(1)
    o.trace(True)
    s = o.optimize_spr(1, radius)
    return o, s, [list(map(int, m)) for m in zip(*o.get_moves())]


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("tie", [1, 0])
@pytest.mark.parametrize("radius", [6, 3])
def test_device_climb_matches_oracle_on_fixtures(mods, name, tie, radius):
    """every fixture alignment (ambiguity codes, weighted duplicate columns, protein), both tie rules, two radii: the kernel's
    trajectory == the oracle's, and a second climb on the same engine continues the tie stream where the oracle's does"""
    engine, po, synth, trees = mods
    fx = load_fixture(name)
    n = fx["codes_np"].shape[0]
    for seed in (1, 5):
        back = trees.random_topology(n, np.random.default_rng(seed))
        e, s, mv = climb(engine, fx["codes_np"], fx["datatype"], back, 2, tie, seed, radius, fx["weights_np"])
        o, so, mo = oracle_climb(po, fx["codes_np"], fx["datatype"], back, tie, seed, radius, fx["weights_np"])
        assert s == so
        assert mv == mo
        assert (e.get_tree() == o.get_tree()).all()
        st = e.stats()
        assert st["climb_launches"] >= 1 and st["climb_moves"] == len(mv)      # the moves did come from the kernel
        # the tie stream was advanced by exactly the reference's number of draws: a second climb from another tree agrees too
        back2 = trees.random_topology(n, np.random.default_rng(seed + 100))
        e.set_tree(back2)
        o.set_tree(back2)
        assert e.optimize_spr(1, radius) == o.optimize_spr(1, radius)
        assert (e.get_tree() == o.get_tree()).all()


@pytest.mark.parametrize("alpha,n,P", [("DNA", 60, 1500), ("DNA", 150, 4000), ("AA", 40, 900)])
def test_device_climb_equals_host_batches(mods, alpha, n, P):
    """the same engine code path with the loop on the host (climb_device 0), on the device while moves are dense (1, the
    default) and always on the device (2): identical moves; tiles of 16, 32, 64 and 128 words; every speculative batch size"""
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(n, P, alpha, 0.06, seed=n)
    codes = synth.letters_to_codes(letters, alpha)
    dt = engine.DNA if alpha == "DNA" else engine.AA
    back = trees.random_topology(n, np.random.default_rng(3))
    ref = None
    variants = [dict(mode=0), dict(mode=1), dict(mode=2), dict(mode=2, climb_batch_min=1, climb_batch_max=1),
                dict(mode=2, climb_batch_min=8, climb_batch_max=8), dict(mode=2, climb_batch_min=3, climb_batch_max=16),
                dict(mode=2, climb_batch_min=16, climb_batch_max=16), dict(mode=1, climb_idle=8)]
    if alpha == "DNA":
        variants += [dict(mode=2, climb_tile=2), dict(mode=2, climb_tile=4), dict(mode=2, climb_tile=8)]
    for v in variants:
        v = dict(v)
        mode = v.pop("mode")
        e, s, mv = climb(engine, codes, dt, back, mode, engine.TIE_RANDOM, 7, 6, **v)
        got = (s, mv, e.get_tree().tolist())
        if ref is None:
            ref = got
            assert len(mv) > n // 2                    # a random start tree: a real climb
        assert got == ref, (mode, v)
        if mode == 2:
            assert e.stats()["climb_moves"] == len(mv)
        if mode == 0:
            assert e.stats()["climb_launches"] == 0


def test_device_climb_leaves_the_engine_consistent(mods):
    """after the kernel has edited the tree the host-side calls see the same tree: score, per-pattern lengths, a full-radius
    scan of every prune node and a further climb all agree with the oracle"""
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(48, 1200, "DNA", 0.07, seed=9)
    codes = synth.letters_to_codes(letters, "DNA")
    back = trees.random_topology(48, np.random.default_rng(2))
    e, s, mv = climb(engine, codes, engine.DNA, back, 2, engine.TIE_RANDOM, 3, 6)
    o, so, mo = oracle_climb(po, codes, engine.DNA, back, engine.TIE_RANDOM, 3, 6)
    assert (s, mv) == (so, mo)
    assert e.score_tree() == o.score_tree() == s
    ptn, tot = e.pattern_scores()
    o3 = po.Oracle(codes)
    o3.enable_persite(True)
    o3.score_tree(o.get_tree())
    ptn_o, tot_o = o3.pattern_scores()
    assert tot == tot_o == s and ptn.tolist() == ptn_o.tolist()
    cur = o.score_tree()
    for rec in o.nodep()[1:2 * 48 - 1][::5]:
        o.set_best(cur)
        o.trace(True)
        o.rearrange(int(rec), 1, 6)
        toks = trace_tokens(*o.get_trace())
        q, mp, n_p = e.spr_scan(int(rec), 1, 6)
        mine = ["P"] + [f"{a}:{b}" for a, b in zip(q[:n_p], mp[:n_p])] + ["Q"] + [f"{a}:{b}" for a, b in zip(q[n_p:], mp[n_p:])]
        assert mine == toks
    # another tree climbed on the same engine: the kernel starts from the engine's state as the host calls left it
    back2 = trees.random_topology(48, np.random.default_rng(77))
    e.set_tree(back2)
    o2 = po.Oracle(codes)
    o2.set_tree(back2)
    e.seed_ties(engine.TIE_RANDOM, 11)
    o2.seed_ties(po.TIE_RANDOM, 11)
    assert e.optimize_spr(1, 6) == o2.optimize_spr(1, 6)
    assert (e.get_tree() == o2.get_tree()).all()


def test_paths_the_kernel_does_not_take_stay_on_the_host(mods):
    """radius above 6, a host random_double() callback and the weighted engine are served by the host-driven batches; the
    results are the oracle's either way"""
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(40, 900, "DNA", 0.07, seed=4)
    codes = synth.letters_to_codes(letters, "DNA")
    back = trees.random_topology(40, np.random.default_rng(1))
    e, s, mv = climb(engine, codes, engine.DNA, back, 2, engine.TIE_RANDOM, 3, 8)
    o, so, mo = oracle_climb(po, codes, engine.DNA, back, engine.TIE_RANDOM, 3, 8)
    assert (s, mv) == (so, mo)
    assert e.stats()["climb_launches"] == 0


def test_stepwise_addition_then_device_climb(mods):
    """_pllComputeRandomizedStepwiseAdditionParsimonyTree (reference sprparsimony.cpp:3224-3235): the SPR sweeps behind the
    addition phase run in the kernel too; tree and length == the reference's own (fixtures, first-best rule) and the oracle's"""
    engine, po, synth, trees = mods
    for name in FIXTURES:
        fx = load_fixture(name)
        for r in fx["ras"]:
            e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
            e.set_option("climb_device", 2)
            e.seed_ties(engine.TIE_FIRST, 0)
            assert e.make_parsimony_tree(r["seed"], r["spr_dist"]) == r["score"]
            assert e.get_tree().tolist() == r["back"]


def test_c2_device_climb_full_size(mods):
    """BASELINE config 2 (200 taxa x 10 000 patterns) from a random tree, every sweep in the kernel: moves == host batches,
    final tree == oracle"""
    engine, po, synth, trees = mods
    letters, _ = synth.workload("C2")
    codes = synth.letters_to_codes(letters, "DNA")
    back = trees.random_topology(codes.shape[0], np.random.default_rng(1))
    e2, s2, m2 = climb(engine, codes, engine.DNA, back, 2, engine.TIE_RANDOM, 1, 6)
    e0, s0, m0 = climb(engine, codes, engine.DNA, back, 0, engine.TIE_RANDOM, 1, 6)
    assert (s2, m2) == (s0, m0) and len(m2) > 300
    o = po.Oracle(codes, datatype=engine.DNA)
    o.set_tree(back)
    o.seed_ties(po.TIE_RANDOM, 1)
    assert o.optimize_spr(1, 6) == s2
    assert (o.get_tree() == e2.get_tree()).all()


@pytest.mark.parametrize("fault", [0xFFFFFFFF, 1, 7])
def test_a_lost_launch_falls_back_to_the_host_path_with_the_same_trajectory(mods, fault):
    """the recovery paths of the persistent kernel (engine option climb_fault, tests only): the start barrier decides "abort"
    (a chip on which the workgroups do not all become resident), or one workgroup withholds its sums in the exchange of step k
    (workgroups that lost each other: the others time out after 100 ms).  Either way the host has taken nothing over from that
    launch; the climb goes on as host-driven batches and ends with the moves of an undisturbed run."""
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(150, 4000, "DNA", 0.06, seed=150)
    codes = synth.letters_to_codes(letters, "DNA")
    back = trees.random_topology(150, np.random.default_rng(3))
    ref_e, ref_s, ref_mv = climb(engine, codes, engine.DNA, back, 2, engine.TIE_RANDOM, 7, 6)
    e = engine.FitchEngine(codes)
    e.set_option("climb_device", 2)
    e.set_option("climb_fault", fault)
    e.set_tree(back)
    e.reset_node_order()
    e.seed_ties(engine.TIE_RANDOM, 7)
    s = e.optimize_spr(1, 6)
    assert s == ref_s
    assert [list(map(int, m)) for m in zip(*e.moves())] == ref_mv
    assert (e.get_tree() == ref_e.get_tree()).all()
    st = e.stats()
    assert st["climb_launches"] >= 1 and st["climb_moves"] < len(ref_mv)          # the faulted launch contributed nothing
    # and the engine is whole afterwards: another climb from another tree, in the kernel again
    back2 = trees.random_topology(150, np.random.default_rng(4))
    ref2 = climb(engine, codes, engine.DNA, back2, 2, engine.TIE_RANDOM, 9, 6)
    e.set_tree(back2)
    e.reset_node_order()
    e.seed_ties(engine.TIE_RANDOM, 9)
    e.reset_stats()
    assert e.optimize_spr(1, 6) == ref2[1] and [list(map(int, m)) for m in zip(*e.moves())] == ref2[2]
    assert e.stats()["climb_moves"] == len(ref2[2])


def test_more_than_a_thousand_taxa_climb_in_the_kernel_with_the_plain_batch_bound():
    """The kernel's control state grows with the taxa AND with the prune nodes a step may hold: 1000 taxa fit beside the quiet stretch's
    sixteen, 1300 still fit beside the plain climb's eight (climb_supported takes the launch's bound) -- same climb as host-driven."""
    from mpboot_amd import engine, synth, trees
    n, P = 1300, 640
    letters, _ = synth.synth_alignment(n, P, "DNA", 0.05, seed=31)
    codes = synth.letters_to_codes(letters, "DNA")
    back = trees.random_topology(n, np.random.default_rng(6))
    res = []
    for mode in (0, 1):
        e = engine.FitchEngine(codes)
        e.set_option("climb_device", mode)
        e.set_tree(back); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 4)
        s = e.optimize_spr(1, 6)
        res.append((s, [x.tolist() for x in e.moves()], e.get_tree().tolist(), e.tie_state(), e.stats()["climb_launches"]))
    assert res[0][:4] == res[1][:4]
    assert res[0][4] == 0 and res[1][4] >= 1

"""Refresh schedule made on the device (k_sched, mpboot_amd/csrc/kernels.hip) against the host's schedule and the oracle.

A newly handed-over topology has no valid vector: the order in which `newviewParsimonyIterativeFast` (reference
sprparsimony.cpp:554-878) must recompute them follows from the traversal descriptors (:434-467).  The engine derives that order
-- dependency levels of all 3(n - 2) directional vectors -- on the GPU from the topology array alone; option dev_sched = 0
keeps the host's two-sweep schedule.  Tree lengths, per-pattern lengths and every insertion test of a sweep must not depend on
who made the schedule.  Integer work: every comparison is exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from mpboot_amd import engine, synth, trees
    from oracle import pyoracle as po
    return engine, po, synth, trees


@pytest.mark.parametrize("n,P,alpha", [(4, 200, "DNA"), (5, 300, "DNA"), (37, 900, "DNA"), (300, 2500, "DNA"), (64, 700, "AA")])
def test_device_schedule_equals_host_schedule(mods, n, P, alpha):
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(n, P, alpha, 0.07, seed=n)
    codes = synth.letters_to_codes(letters, alpha)
    dt = engine.DNA if alpha == "DNA" else engine.AA
    rng = np.random.default_rng(n)
    shapes = [trees.random_topology(n, rng) for _ in range(3)]
    # a caterpillar: built by adding every tip next to the one before it
    back = trees.empty_back(n)
    v = n + 1
    for s, t in enumerate((1, 2, 3)):
        back[3 * v + s] = 3 * t
        back[3 * t] = 3 * v + s
    for t in range(4, n + 1):
        v += 1
        e = 3 * (t - 1)
        f = int(back[e])
        back[3 * v] = 3 * t
        back[3 * t] = 3 * v
        back[3 * v + 1] = e
        back[e] = 3 * v + 1
        back[3 * v + 2] = f
        back[f] = 3 * v + 2
    trees.validate(back, n)
    shapes.append(back)
    o = po.Oracle(codes, datatype=dt)
    for back in shapes:
        got = []
        for dev in (1, 0):
            e = engine.FitchEngine(codes, datatype=dt)
            e.set_option("dev_sched", dev)
            e.set_option("plan_cache", 0)
            s = e.score_tree(back)
            ptn, tot = e.pattern_scores()
            rad = min(6, n - 3)
            if rad >= 1:
                k, mp, offs = e.sweep_costs(1, rad)
                got.append((s, tot, ptn.tolist(), mp.tolist(), offs.tolist(), k))
            else:
                got.append((s, tot, ptn.tolist()))
            # the same topology again (the cached schedule is replayed) and a different one after it
            assert e.score_tree(back) == s
            assert e.score_tree(shapes[0]) == o.score_tree(shapes[0])
        assert got[0] == got[1]
        assert got[0][0] == o.score_tree(back)


def test_device_schedule_after_moves_and_reweighting(mods):
    """the device schedule serves every from-scratch refresh: after a climb's moves were replayed (k_climb invalidates everything),
    after re-weighting, and with the plan cache on (the cached schedule's level count stays on the device)"""
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(90, 2000, "DNA", 0.06, seed=5)
    codes = synth.letters_to_codes(letters, "DNA")
    back = trees.random_topology(90, np.random.default_rng(1))
    res = []
    for dev in (1, 0):
        e = engine.FitchEngine(codes)
        e.set_option("dev_sched", dev)
        e.set_tree(back)
        e.seed_ties(engine.TIE_RANDOM, 3)
        s1 = e.optimize_spr(1, 6)
        t1 = e.get_tree()
        w = np.random.default_rng(2).integers(0, 4, size=codes.shape[1]).astype(np.int32)
        e.set_weights(w)
        s2 = e.score_tree(t1)
        s3 = e.score_tree(t1)
        e.set_weights(np.ones(codes.shape[1], dtype=np.int32))
        s4 = e.optimize_spr(1, 6)
        res.append((s1, t1.tolist(), s2, s3, s4, e.get_tree().tolist()))
    assert res[0] == res[1]
    o = po.Oracle(codes)
    assert o.score_tree(np.asarray(res[0][1], dtype=np.int32)) == res[0][0]


def _caterpillar(trees, n):
    back = trees.empty_back(n)
    v = n + 1
    for s, t in enumerate((1, 2, 3)):
        back[3 * v + s] = 3 * t
        back[3 * t] = 3 * v + s
    for t in range(4, n + 1):
        v += 1
        e = 3 * (t - 1)
        f = int(back[e])
        back[3 * v] = 3 * t
        back[3 * t] = 3 * v
        back[3 * v + 1] = e
        back[e] = 3 * v + 1
        back[3 * v + 2] = f
        back[f] = 3 * v + 2
    trees.validate(back, n)
    return back


@pytest.mark.parametrize("n,P", [(8, 300), (9, 300), (33, 800), (150, 3000), (420, 1500)])
@pytest.mark.parametrize("cache", [1, 0])
def test_device_planned_sweep_equals_host_planned_sweep(mods, n, P, cache):
    """mpf_sweep_scan on a tree just handed over: scan descriptors laid out by k_sched's second workgroup (dev_plan 1, the default)
    vs Engine::plan_walk on the host (dev_plan 0) vs every candidate of the sweep (mpf_spr_sweep_costs) -- number of insertion
    tests and the best length, for random trees and a caterpillar, radius 6 and 3, long neighbourhoods cut (split_cands) or
    not, alternating topologies (a cached plan must not outlive its topology) and after re-weighting"""
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(n, P, "DNA", 0.06, seed=n)
    codes = synth.letters_to_codes(letters, "DNA")
    rng = np.random.default_rng(n)
    A, B, C = trees.random_topology(n, rng), trees.random_topology(n, rng), _caterpillar(trees, n)
    e1 = engine.FitchEngine(codes)
    e0 = engine.FitchEngine(codes)
    e0.set_option("dev_plan", 0)
    for e in (e0, e1):
        e.set_option("plan_cache", 7 if cache else 0)
    ref = engine.FitchEngine(codes)
    ref.set_option("dev_plan", 0)
    ref.set_option("dev_sched", 0)

    def truth(back, rad):
        ref.score_tree(back)
        k, mp, _ = ref.sweep_costs(1, rad)
        return int(k), int(mp.min()) if k else None

    for split in (64, 16, 100000):
        for e in (e0, e1):
            e.set_option("split_cands", split)
        for rad in (6, 3):
            for back in (A, B, A, A, C, B, B):
                want = truth(back, rad)
                for e in (e1, e0):
                    e.set_tree(back)
                    k, best = e.sweep_scan(1, rad)
                    assert (int(k), int(best) if k else None) == want
    # re-weighted: the topology's plans stay, the lengths change
    w = np.random.default_rng(3).integers(0, 3, size=codes.shape[1]).astype(np.int32)
    for e in (e0, e1, ref):
        e.set_weights(w)
    want = truth(A, 6)
    for e in (e1, e0):
        e.set_tree(A)
        k, best = e.sweep_scan(1, 6)
        assert (int(k), int(best)) == want
        k, best = e.sweep_scan(1, 6)              # vectors valid now: the host-planned path
        assert (int(k), int(best)) == want
    assert e1.score_tree() == ref.score_tree()


@pytest.mark.parametrize("n,P", [(40, 900), (130, 2500)])
def test_word_major_copy_follows_every_refresh(mods, n, P):
    """the planned scan (k_scan_prog) reads a word-major copy of the vectors (one 16-byte load per lane and vector) that
    k_pack_tips, k_newview_wgq and k_newview_chain write beside the row-major store: a climb on host-driven batches with every
    scan planned (scan_prog 2) -- chained refreshes between the scans, re-weighting in the middle, a device climb in between
    (k_climb writes the row-major store only: the copy must be left alone until the next full refresh) -- gives the oracle's
    moves with the copy in use (scan_shadow 1, the default) and without"""
    engine, po, synth, trees = mods
    letters, _ = synth.synth_alignment(n, P, "DNA", 0.06, seed=n + 1)
    codes = synth.letters_to_codes(letters, "DNA")
    back = trees.random_topology(n, np.random.default_rng(n))
    back2 = trees.random_topology(n, np.random.default_rng(n + 7))
    w = np.random.default_rng(5).integers(0, 3, size=codes.shape[1]).astype(np.int32)
    o = po.Oracle(codes)
    o.set_tree(back)
    o.seed_ties(po.TIE_RANDOM, 3)
    want = [o.optimize_spr(1, 6)]
    want.append(o.get_tree().tolist())
    o.set_weights(w)
    o.set_tree(back2)
    want.append(o.optimize_spr(1, 6))
    want.append(o.get_tree().tolist())
    o.set_tree(back)
    want.append(o.optimize_spr(1, 6))
    want.append(o.get_tree().tolist())
    for shadow in (1, 0):
        for split in (1000, 0):
            e = engine.FitchEngine(codes)
            e.set_option("scan_shadow", shadow)
            e.set_option("scan_prog", 2)
            e.set_option("split_below", split)
            e.set_option("climb_device", 0)
            e.set_tree(back)
            e.seed_ties(engine.TIE_RANDOM, 3)
            got = [e.optimize_spr(1, 6), e.get_tree().tolist()]
            e.set_weights(w)
            e.set_tree(back2)
            e.set_option("climb_device", 2)          # the kernel edits vectors behind the copy's back ...
            got.append(e.optimize_spr(1, 6))
            got.append(e.get_tree().tolist())
            e.set_option("climb_device", 0)          # ... and the host-driven batches that follow must not read it stale
            e.set_tree(back)
            got.append(e.optimize_spr(1, 6))
            got.append(e.get_tree().tolist())
            assert got == want, (shadow, split)
            assert e.score_tree() == want[4]


def test_trees_beyond_the_device_scheduler_take_the_host_path(mods):
    """k_sched keeps the topology of up to 16 384 vectors in the LDS of one workgroup (4 096 taxa); a larger tree is scheduled and
    planned on the host as before -- same results, no error"""
    engine, po, synth, trees = mods
    n = 4200
    letters, _ = synth.synth_alignment(n, 96, "DNA", 0.05, seed=2)
    codes = synth.letters_to_codes(letters, "DNA")
    back = trees.random_topology(n, np.random.default_rng(4))
    e = engine.FitchEngine(codes)
    o = po.Oracle(codes)
    assert e.score_tree(back) == o.score_tree(back)
    e.set_tree(back)
    k, best = e.sweep_scan(1, 6)
    k2, mp, _ = e.sweep_costs(1, 6)
    assert (int(k), int(best)) == (int(k2), int(mp.min()))
    assert e.get_option("sched_levels") == 0           # (no device-made schedule on record)

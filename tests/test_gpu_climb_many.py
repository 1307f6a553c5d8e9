"""mpf_optimize_spr_many (k_climb_many): many independent SPR hill climbs, one resident workgroup each, one launch per round -- every
climb's accepted moves, final tree, length and tie-stream state equal to its own mpf_optimize_spr call; and the form underneath it,
k_climb with fewer workgroups than tiles (option climb_groups), equal to the workgroup-per-tile launch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solo(codes, dt, back, seed, tie, opts=None):
    from mpboot_amd import engine
    e = engine.FitchEngine(codes, datatype=dt)
    for k, v in (opts or {}).items():
        e.set_option(k, v)
    e.set_tree(back)
    e.reset_node_order()
    e.seed_ties(tie, seed)
    s = e.optimize_spr(1, 6)
    return e, (s, [x.tolist() for x in e.moves()], e.get_tree().tolist(), e.tie_state())


@pytest.mark.parametrize("n,P,alphabet,tile", [(40, 1500, "DNA", 1), (120, 3000, "DNA", 4), (200, 10000, "DNA", 4), (60, 900, "AA", 1)])
def test_fewer_workgroups_than_tiles_take_the_same_path(n, P, alphabet, tile):
    from mpboot_amd import engine, synth, trees
    letters, _ = synth.synth_alignment(n, P, alphabet, 0.08, seed=n + P)
    codes = synth.letters_to_codes(letters, alphabet)
    dt = engine.DNA if alphabet == "DNA" else engine.AA
    back = trees.random_topology(n, np.random.default_rng(5))
    ref = None
    for groups in (0, 1, 2, 3):
        e, sig = _solo(codes, dt, back, 9, engine.TIE_RANDOM, {"climb_device": 2, "climb_tile": tile, "climb_groups": groups})
        assert e.stats()["climb_launches"] >= 1
        ref = ref or sig
        assert sig == ref, groups
    if tile == 4 and alphabet == "DNA":
        # 64-word tiles run four-state data a word per lane (quadtile.hpp, kWordMajor) unless told otherwise: the quad shape of the same tiles
        e, sig = _solo(codes, dt, back, 9, engine.TIE_RANDOM, {"climb_device": 2, "climb_tile": 4, "climb_word_major": 0})
        assert sig == ref


@pytest.mark.parametrize("n,P,alphabet,n_eng,tie", [(40, 1500, "DNA", 7, "random"), (120, 3000, "DNA", 12, "random"), (60, 900, "AA", 5, "random"),
                                                     (200, 10000, "DNA", 24, "random"), (48, 2000, "DNA", 6, "first")])
def test_many_climbs_in_one_launch_equal_their_solo_runs(n, P, alphabet, n_eng, tie):
    from mpboot_amd import engine, synth, trees
    letters, _ = synth.synth_alignment(n, P, alphabet, 0.08, seed=n + P + 1)
    codes = synth.letters_to_codes(letters, alphabet)
    dt = engine.DNA if alphabet == "DNA" else engine.AA
    mode = engine.TIE_RANDOM if tie == "random" else engine.TIE_FIRST
    rng = np.random.default_rng(3)
    starts = [trees.random_topology(n, rng) for _ in range(n_eng)]
    # (one start is an optimum already: a climb that is over after its first sweep, beside others that run for many rounds)
    e_opt, _ = _solo(codes, dt, starts[0], 1, mode)
    starts[1] = e_opt.get_tree()
    engs = []
    for k, t in enumerate(starts):
        e = engine.FitchEngine(codes, datatype=dt)
        if k % 3 == 2:                       # other weights on some engines (bootstrap refinements re-weight per sample)
            w = np.random.default_rng(k).integers(0, 3, size=P).astype(np.int32)
            w[:8] = 1
            e.set_weights(w)
        e.set_tree(t)
        e.reset_node_order()
        e.seed_ties(mode, 100 + k)
        engs.append(e)
    scores = engine.optimize_spr_many(engs, 1, 6)
    for k, e in enumerate(engs):
        s = engine.FitchEngine(codes, datatype=dt)
        if k % 3 == 2:
            s.set_weights(e.weights())
        s.set_tree(starts[k])
        s.reset_node_order()
        s.seed_ties(mode, 100 + k)
        want = s.optimize_spr(1, 6)
        assert int(scores[k]) == want, k
        assert [x.tolist() for x in e.moves()] == [x.tolist() for x in s.moves()], k
        assert (e.get_tree() == s.get_tree()).all() and e.tie_state() == s.tie_state(), k
        assert e.score_tree() == want
    # the engines are usable afterwards, alone and together again
    assert engs[2].optimize_spr(1, 6) == int(scores[2])
    again = engine.optimize_spr_many(engs[:3], 1, 6)
    assert again.tolist() == scores[:3].tolist()


@pytest.mark.parametrize("n,P,n_eng", [(40, 1500, 5), (200, 4000, 9)])
def test_sweeps_inside_the_launch_or_one_launch_per_sweep(n, P, n_eng):
    """option many_sweeps_inside: the next sweep's nodeRectifierPars on the device (default) or on the host between launches -- the
    same climbs either way, in fewer launches; a climb that makes more moves than a launch's move list holds
    comes back in the middle of a sweep the DEVICE started and goes on from the order it left (option many_moves_cap forces that)."""
    from mpboot_amd import engine, synth, trees
    letters, _ = synth.synth_alignment(n, P, "DNA", 0.08, seed=n + P + 2)
    codes = synth.letters_to_codes(letters, "DNA")
    rng = np.random.default_rng(8)
    starts = [trees.random_topology(n, rng) for _ in range(n_eng)]
    res = {}
    for inside in (1, 0, 2):
        engs = []
        for k, t in enumerate(starts):
            e = engine.FitchEngine(codes)
            e.set_option("many_sweeps_inside", min(inside, 1))
            if inside == 2:
                e.set_option("many_moves_cap", 3 * n // 4)
            e.set_tree(t); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 40 + k)
            engs.append(e)
        sc = engine.optimize_spr_many(engs, 1, 6)
        res[inside] = [(int(sc[k]), [x.tolist() for x in e.moves()], e.get_tree().tolist(), e.tie_state()) for k, e in enumerate(engs)]
        launches = [e.stats()["climb_launches"] for e in engs]
        res[("launches", inside)] = launches
    assert res[1] == res[0] and res[2] == res[0]
    assert sum(res[("launches", 1)]) < sum(res[("launches", 0)])
    assert max(res[("launches", 1)]) == 1 and min(res[("launches", 2)]) >= 2
    for k, t in enumerate(starts[:3]):
        _, sig = _solo(codes, engine.DNA, t, 40 + k, engine.TIE_RANDOM)
        assert res[1][k] == sig, k


def test_engines_the_batch_cannot_take_run_alone_inside_the_call():
    from mpboot_amd import engine, synth, trees
    letters, _ = synth.synth_alignment(30, 800, "DNA", 0.1, seed=4)
    codes = synth.letters_to_codes(letters, "DNA")
    letters2, _ = synth.synth_alignment(24, 500, "AA", 0.1, seed=5)
    codes2 = synth.letters_to_codes(letters2, "AA")
    rng = np.random.default_rng(1)
    a = engine.FitchEngine(codes)
    b = engine.FitchEngine(codes2, datatype=engine.AA)                     # another alignment shape
    c = engine.FitchEngine(codes)
    samples = rng.multinomial(800, np.ones(800) / 800, size=8).astype(np.uint16)
    c.ufboot_attach(samples)                                              # a tracked climb
    ta, tb = trees.random_topology(30, rng), trees.random_topology(24, rng)
    for e, t in ((a, ta), (b, tb), (c, ta)):
        e.set_tree(t); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 3)
    got = engine.optimize_spr_many([a, b, c], 1, 6)
    for e0, t, dt, cd, k in ((a, ta, engine.DNA, codes, 0), (b, tb, engine.AA, codes2, 1)):
        s = engine.FitchEngine(cd, datatype=dt)
        s.set_tree(t); s.reset_node_order(); s.seed_ties(engine.TIE_RANDOM, 3)
        assert s.optimize_spr(1, 6) == int(got[k]) and (s.get_tree() == e0.get_tree()).all()
    assert len(c.ufboot_tree_logl()) > 0 and int(got[2]) == c.score_tree()
    with pytest.raises(engine.MpfError):
        engine.optimize_spr_many([a, a], 1, 6)


def test_a_failed_launch_leaves_every_engine_of_the_batch_usable():
    """One climb of a batch comes back as an abort (option climb_fault on its engine): the call fails as a whole.  The launch has
    rewritten vectors of every engine; each of them is consistent afterwards -- the climbs taken over before the failure was seen are
    done, the others hold the tree they came with -- and scores and climbs from where it stands like a fresh engine on that tree."""
    from mpboot_amd import engine, synth, trees
    letters, _ = synth.synth_alignment(60, 5000, "DNA", 0.08, seed=77)
    codes = synth.letters_to_codes(letters, "DNA")
    rng = np.random.default_rng(2)
    starts = [trees.random_topology(60, rng) for _ in range(4)]
    engs = []
    for k, t in enumerate(starts):
        e = engine.FitchEngine(codes)
        e.set_tree(t); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 20 + k)
        engs.append(e)
    engs[2].set_option("climb_fault", 0xFFFFFFFF)
    with pytest.raises(engine.MpfError):
        engine.optimize_spr_many(engs, 1, 6)
    assert (engs[2].get_tree() == starts[2]).all() and (engs[3].get_tree() == starts[3]).all()
    for k, e in enumerate(engs):
        t = e.get_tree()
        f = engine.FitchEngine(codes)
        f.set_tree(t)
        assert e.score_tree() == f.score_tree(), k
        f.reset_node_order(); f.seed_ties(engine.TIE_RANDOM, 50 + k)
        e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 50 + k)
        assert e.optimize_spr(1, 6) == f.optimize_spr(1, 6) and (e.get_tree() == f.get_tree()).all() and e.tie_state() == f.tie_state(), k
    again = engine.optimize_spr_many(engs, 1, 6)          # (and together: all at their optima now)
    assert [int(x) for x in again] == [e.score_tree() for e in engs]

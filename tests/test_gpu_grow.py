"""k_grow -- the addition loop of _pllMakeParsimonyTreeFast (reference sprparsimony.cpp:3107-3181, stepwiseAddition :2977-3019) as one
persistent kernel per tree: the same trees, the same per-step lengths and insertion branches, the same state of the tie stream as
the host-driven loop (which test_gpu_parity.py pins against the reference's own RAS trees and stepwise checkpoints) and as the
oracle; both tie rules, DNA / protein / 32-state data, every tile width, trees whose subtrees cost nothing (the descent cut),
repeated calls on an engine whose tr->nodep nodeRectifierPars has re-ordered, concurrent engines, and the way back to the host's
loop when the launch does not start."""
import threading

import numpy as np
import pytest

from helpers import load_fixture

pytestmark = pytest.mark.gpu


def _both(codes, dt, seed, tie, tie_seed, opts=None, weights=None):
    from mpboot_amd import engine
    out = []
    for dev in (0, 1):
        e = engine.FitchEngine(codes, weights, datatype=dt)
        for k, v in (opts or {}).items():
            e.set_option(k, v)
        e.set_option("grow_device", dev)
        e.seed_ties(tie, tie_seed)
        score, best, ins = e.stepwise_addition(seed)
        out.append((best.tolist(), ins.tolist(), score, e.get_tree().tolist(), e.tie_state(), e.get_option("grow_launches"), e.get_option("grow_last_err")))
    return out


@pytest.mark.parametrize("tie", ["random", "first"])
@pytest.mark.parametrize("name", ["dna_clean", "dna_ambig", "dna_dups", "aa", "dna_48", "aa_40", "morph32_40", "bin"])
def test_fixture_trees_equal_the_host_loop_and_the_oracle(name, tie):
    from mpboot_amd import engine
    from oracle import pyoracle as po
    fx = load_fixture(name)
    for seed in (5, 77):
        h, d = _both(fx["codes_np"], fx["datatype"], seed, engine.TIE_RANDOM if tie == "random" else engine.TIE_FIRST, 3, weights=fx["weights_np"])
        assert d[5] == 1 and d[6] == 0 and h[5] == 0          # the kernel built the tree, and came back clean
        assert h[:5] == d[:5]
        o = po.Oracle(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
        o.seed_ties(po.TIE_RANDOM if tie == "random" else po.TIE_FIRST, 3)
        os_, ob, oi = o.stepwise(seed)
        assert os_ == d[2] and (o.get_tree() == np.array(d[3])).all()
        assert ob.tolist() == d[0] and o.tie_state() == d[4]


@pytest.mark.parametrize("n,P,alphabet,opts", [(30, 600, "DNA", {}), (200, 10000, "DNA", {}), (200, 10000, "DNA", {"grow_tile": 4}), (200, 10000, "DNA", {"grow_tile": 0}), (700, 1500, "DNA", {"grow_tile": 0}),
                                               (300, 3000, "DNA", {"grow_tile": 2}), (90, 9000, "DNA", {"grow_tile": 8}), (120, 3000, "AA", {}),
                                               (700, 1500, "DNA", {})])
def test_synthetic_trees_equal_the_host_loop(n, P, alphabet, opts):
    """larger trees (the skeleton + parts split, several workgroups exchanging rows) and every tile shape: quad tiles of 16 x 1 / 2 / 4 / 8 words (default: fitted) and
    the word-major DNA layout (grow_tile 0: a lane = one word with its four states, read from the engine's word-major copy)"""
    from mpboot_amd import engine, synth
    letters, _ = synth.synth_alignment(n, P, alphabet, 0.07, seed=n + P)
    codes = synth.letters_to_codes(letters, alphabet)
    h, d = _both(codes, engine.DNA if alphabet == "DNA" else engine.AA, 1234, engine.TIE_RANDOM, 7, opts)
    assert d[5] == 1 and d[6] == 0
    assert h[:5] == d[:5]


def test_subtrees_without_a_mutation_are_not_descended_into():
    """stepwiseAddition's cut (:3014): below a node whose subtree costs nothing no branch is tested.  Identical sequences make such
    subtrees; the kernel keeps the flags by exchanging one bit per node of the last root path."""
    from mpboot_amd import engine, synth
    letters, _ = synth.synth_alignment(12, 400, "DNA", 0.1, seed=3)
    codes = synth.letters_to_codes(letters, "DNA")
    big = np.concatenate([codes, codes[:6], codes[:6], codes[2:5]], axis=0)          # clusters of identical taxa
    for seed in range(6):
        h, d = _both(big, engine.DNA, 100 + seed, engine.TIE_RANDOM, 1 + seed)
        assert d[5] == 1 and d[6] == 0
        assert h[:5] == d[:5]


def test_repeated_calls_and_the_spr_phase_behind_the_tree():
    """make_parsimony_tree again and again on one engine: nodeRectifierPars of the SPR phase re-orders tr->nodep, so later trees
    take other inner nodes (and other records of them) in another order"""
    from mpboot_amd import engine, synth
    letters, _ = synth.synth_alignment(80, 2500, "DNA", 0.08, seed=9)
    codes = synth.letters_to_codes(letters, "DNA")
    res = []
    for dev in (0, 1):
        e = engine.FitchEngine(codes)
        e.set_option("grow_device", dev)
        out = []
        for k, (seed, rad) in enumerate(((4242, 0), (1234, 3), (99, 6), (1234, 3))):
            e.seed_ties(engine.TIE_RANDOM, 7 + k)
            sc = e.make_parsimony_tree(seed, rad)
            out.append((sc, e.get_tree().tolist(), e.tie_state(), [x.tolist() for x in e.moves()]))
        res.append(out)
        assert e.get_option("grow_launches") == (4 if dev else 0)
    assert res[0] == res[1]


def test_a_launch_that_does_not_start_falls_back_to_the_host_loop():
    from mpboot_amd import engine
    fx = load_fixture("dna_48")
    e = engine.FitchEngine(fx["codes_np"], datatype=fx["datatype"])
    e.seed_ties(engine.TIE_RANDOM, 4)
    ref = e.stepwise_addition(31)
    ref = (ref[0], ref[1].tolist(), ref[2].tolist(), e.get_tree().tolist(), e.tie_state())
    assert e.get_option("grow_launches") == 1 and e.get_option("grow_last_err") == 0
    e.set_option("grow_fault", 0xFFFFFFFF)              # the start barrier decides "abort"
    e.seed_ties(engine.TIE_RANDOM, 4)
    got = e.stepwise_addition(31)
    assert e.get_option("grow_last_err") != 0
    assert (got[0], got[1].tolist(), got[2].tolist(), e.get_tree().tolist(), e.tie_state()) == ref
    e.seed_ties(engine.TIE_RANDOM, 4)
    got = e.stepwise_addition(31)                       # and the kernel is back for the next tree
    assert e.get_option("grow_launches") == 3
    assert (got[0], got[1].tolist(), got[2].tolist(), e.get_tree().tolist(), e.tie_state()) == ref


def test_concurrent_engines_build_their_solo_trees():
    """the start-up phase's shape: several engines on host threads, their launches admitted together, two workgroups per CU"""
    from mpboot_amd import engine, synth
    letters, _ = synth.synth_alignment(150, 20000, "DNA", 0.06, seed=5)
    codes = synth.letters_to_codes(letters, "DNA")
    K, T = 6, 3
    engines = [engine.FitchEngine(codes) for _ in range(K)]
    solo = {}
    for u in range(K * T):
        e = engines[0]
        e.seed_ties(engine.TIE_RANDOM, 50 + u)
        e.reset_node_order()                         # (tr->nodep as a fresh instance has it: the records a tree takes do not depend on the engine's past)
        solo[u] = (e.make_parsimony_tree(900 + u, 6), e.get_tree().tolist())
    got, errs = {}, []

    def work(k):
        try:
            for u in range(k, K * T, K):
                e = engines[k]
                e.seed_ties(engine.TIE_RANDOM, 50 + u)
                e.reset_node_order()
                got[u] = (e.make_parsimony_tree(900 + u, 6), e.get_tree().tolist())
        except Exception as exc:                      # noqa: BLE001
            errs.append(repr(exc))

    th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert got == solo

"""mpboot_amd/host/iqflow.cpp (device-free part of libmpfitch.so) against its plain-Python witness oracle/iqflow_slow.py: the
perturbation steps of IQTree::doTreeSearch between two climbs -- doRandomNNIs, createPerturbAlignment -- and the topology key of
the candidate set.  Runs without a GPU."""
import numpy as np
import pytest


@pytest.mark.parametrize("n,seed", [(4, 1), (5, 2), (9, 3), (40, 4), (200, 5), (1000, 6)])
def test_random_nnis_match_the_witness(n, seed):
    from mpboot_amd import engine, trees
    from oracle import iqflow_slow as slow
    rng = np.random.default_rng(seed)
    back = trees.random_topology(n, rng)
    state = int(rng.integers(1, 2 ** 62))
    num = int(0.5 * (n - 3))                                   # floor(curPerStrength * (nseq - 3)), iqtree.cpp:1739
    for k in (num, 1, 3 * n):
        b1, s1, r1 = engine.iq_random_nnis(back, k, state)
        b2, s2, r2 = slow.random_nnis(back, k, state)
        assert b1.tolist() == b2 and s1 == s2 and r1 == r2
        trees.validate(b1, n)
        # 1 + 2 draws per NNI, whatever happens
        assert s1 == int(engine.load_library().mpf_tie_state_after(state, 3 * k))
        if n > 5 and k >= num > 0:
            assert trees.splits(b1) != trees.splits(back)
        back, state = b1, s1


def test_nni_changes_exactly_one_split():
    from mpboot_amd import engine, trees
    back = trees.random_topology(30, np.random.default_rng(1))
    b1, _s, _r = engine.iq_random_nnis(back, 1, 12345)
    a, b = trees.splits(back), trees.splits(b1)
    assert len(a - b) == 1 and len(b - a) == 1


@pytest.mark.parametrize("P,seed,maxw", [(50, 1, 1), (400, 2, 3), (3000, 3, 1), (700, 4, 9)])
def test_ratchet_weights_match_the_witness(P, seed, maxw):
    from mpboot_amd import engine
    from oracle import iqflow_slow as slow
    rng = np.random.default_rng(seed)
    w = rng.integers(1, maxw + 1, size=P).astype(np.int32)
    inf = (rng.random(P) < 0.8).astype(np.uint8)
    inf[0] = 1
    for percent, add in ((50, 1), (10, 2), (100, 1), (0, 1)):
        state = int(rng.integers(1, 2 ** 62))
        o1, s1 = engine.iq_perturb_weights(w, inf, percent, add, state)
        o2, s2 = slow.perturb_weights(w.tolist(), inf.tolist(), percent, add, state)
        assert o1.tolist() == o2 and s1 == s2
        n_inf = int(w[inf != 0].sum())
        assert int((o1 - w).sum()) == (n_inf * percent // 100) * add
        assert ((o1 - w)[inf == 0] == 0).all()
        assert ((o1 - w) <= w * add).all()                     # a site is drawn at most once


def test_topology_key_is_a_function_of_the_topology_only():
    from mpboot_amd import engine, trees
    rng = np.random.default_rng(9)
    names = [f"t{i}" for i in range(1, 61)]
    seen = {}
    for _ in range(30):
        back = trees.random_topology(60, rng)
        k = engine.iq_topology_key(back)
        # another numbering / ring order of the same tree: through Newick and back
        again = trees.newick_to_back(trees.back_to_newick(back, names), names)
        assert engine.iq_topology_key(again) == k
        b2, _s, _r = engine.iq_random_nnis(back, 1, 77)
        assert engine.iq_topology_key(b2) != k
        seen[k] = trees.splits(back)
    assert len(seen) == 30

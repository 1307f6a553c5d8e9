"""Full-size checks (BASELINE.json's 1000 taxa x 50 000 patterns) through size-independent properties:
linearity in the pattern weights, per-pattern sums, scan predictions vs re-scoring, monotone climbs, idempotence."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    from mpboot_amd import engine, synth, trees
    letters, _ = synth.workload("C3")
    codes = synth.letters_to_codes(letters, "DNA")
    eng = engine.FitchEngine(codes)
    back = trees.random_topology(codes.shape[0], np.random.default_rng(12))
    return engine, eng, codes, back


def nx(r):
    v, s = divmod(r, 3)
    return 3 * v + (s + 1) % 3


def apply_spr(back, p, q):
    b = back.copy()
    a1, a2 = int(b[nx(p)]), int(b[nx(nx(p))])
    b[a1], b[a2] = a2, a1
    r = int(b[q])
    b[nx(p)], b[q] = q, nx(p)
    b[nx(nx(p))], b[r] = r, nx(nx(p))
    return b


def test_pattern_scores_sum_to_length(big):
    engine, eng, codes, back = big
    eng.set_weights(np.ones(codes.shape[1], dtype=np.int32))
    score, ptn = eng.compute_parsimony(back)
    assert int(ptn.astype(np.int64).sum()) == score
    assert score == eng.score_tree(back)


def test_length_is_linear_in_the_weights(big):
    """len(w1 + w2) == len(w1) + len(w2) on a fixed tree, and equals <pattern scores, w>"""
    engine, eng, codes, back = big
    rng = np.random.default_rng(0)
    P = codes.shape[1]
    w1 = rng.integers(0, 3, size=P).astype(np.int32)
    w2 = rng.integers(0, 4, size=P).astype(np.int32)
    eng.set_weights(np.ones(P, dtype=np.int32))
    _, ptn = eng.compute_parsimony(back)
    vals = []
    for w in (w1, w2, w1 + w2):
        eng.set_weights(w)
        vals.append(eng.score_tree(back))
        assert vals[-1] == int((ptn.astype(np.int64) * w).sum())
    assert vals[2] == vals[0] + vals[1]
    eng.set_weights(np.ones(P, dtype=np.int32))


def test_scan_predictions_equal_rescoring(big):
    """the length predicted for a candidate move == the length of the tree after actually making the move"""
    engine, eng, codes, back = big
    eng.set_tree(back)
    cur = eng.score_tree()
    rng = np.random.default_rng(5)
    n = codes.shape[0]
    checked = 0
    for rec in rng.integers(3 * (n + 1), 3 * (2 * n - 1), size=6):
        eng.set_tree(back)
        q, mp, n_p = eng.spr_scan(int(rec), 1, 6)
        if len(q) == 0:
            continue
        for idx in {0, len(q) // 2, len(q) - 1}:
            prune = int(rec) if idx < n_p else int(back[rec])
            moved = apply_spr(back, prune, int(q[idx]))
            assert eng.score_tree(moved) == int(mp[idx])
            checked += 1
    assert checked >= 6
    assert eng.score_tree(back) == cur


def test_climb_is_monotone_and_idempotent(big):
    engine, eng, codes, back = big
    eng.set_tree(back)
    start = eng.score_tree()
    eng.seed_ties(engine.TIE_RANDOM, 3)
    eng.set_option("scan_batch", 64)
    final = eng.optimize_spr(1, 6)
    _, _, sc = eng.moves()
    assert len(sc) > 1000 and final < start
    assert (np.diff(sc.astype(np.int64)) <= 0).all() and int(sc[-1]) == final
    assert eng.score_tree() == final
    tree = eng.get_tree()
    eng.seed_ties(engine.TIE_FIRST, 0)
    assert eng.optimize_spr(1, 6) == final          # a local optimum: strict rule makes no move
    assert (eng.get_tree() == tree).all() and len(eng.moves()[0]) == 0
    ntests, best = eng.sweep_scan(1, 6)
    assert best >= final and ntests > 50_000

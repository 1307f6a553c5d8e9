"""Host-side lower-bound helpers (mpf_min_pars_score_patterns, mpf_mst_scores, mpf_segment_patterns,
mpf_remain_bounds): C-ABI vs the plain-Python restatement, plus what makes them bounds."""
import itertools

import numpy as np
import pytest

from helpers import load_fixture
from mpboot_amd import engine
from oracle import bounds_slow, pyoracle as po


@pytest.mark.parametrize("name", ["dna_clean", "dna_ambig", "dna_dups", "aa"])
def test_min_pars_score_is_restated_and_is_a_lower_bound(name):
    fx = load_fixture(name)
    codes = fx["codes_np"]
    got = engine.min_pars_score_patterns(codes, fx["datatype"])
    want = [bounds_slow.calc_min_pars_score_pattern(codes, fx["datatype"], s) for s in range(codes.shape[1])]
    assert got.tolist() == want
    # every tree's per-pattern length is at least that many changes (informative patterns; ambiguity codes can only help)
    o = po.Oracle(codes, fx["weights_np"], datatype=fx["datatype"])
    o.enable_persite(True)
    inf = o.informative().astype(bool)
    for t in fx["trees"][:4]:
        o.score_tree(np.array(t["back"], dtype=np.int32))
        ptn, _ = o.pattern_scores()
        if fx["datatype"] == engine.DNA:
            assert (ptn[inf] >= np.maximum(got[inf], 0)).all()


def brute_force_mst(present, cost):
    best = None
    nodes = list(present)
    edges = [(a, b) for a, b in itertools.combinations(nodes, 2)]
    for pick in itertools.combinations(edges, len(nodes) - 1):
        parent = {v: v for v in nodes}

        def find(v):
            while parent[v] != v:
                v = parent[v]
            return v

        ok = True
        for a, b in pick:
            ra, rb = find(a), find(b)
            if ra == rb:
                ok = False
                break
            parent[ra] = rb
        if ok:
            w = sum(int(cost[a, b]) for a, b in pick)
            best = w if best is None else min(best, w)
    return best


def test_mst_scores_match_restatement_and_brute_force():
    rng = np.random.default_rng(3)
    S, n, P = 4, 9, 60
    c = rng.integers(1, 9, size=(S, S))
    cost = np.triu(c, 1) + np.triu(c, 1).T                       # symmetric, zero diagonal
    states = rng.integers(0, 6, size=(n, P)).astype(np.int8)     # 4, 5 = ambiguity / unknown: ignored
    states[:, :5] = 2                                            # constant patterns
    got = engine.mst_scores(states, cost)
    want = [bounds_slow.find_mst_score(states, cost, p) for p in range(P)]
    assert got.tolist() == want
    for p in range(P):
        present = sorted(set(int(v) for v in states[:, p] if v < S))
        assert got[p] == (0 if len(present) <= 1 else brute_force_mst(present, cost))
    # 20 states, unit costs: MST = (#states present) - 1
    st20 = rng.integers(0, 23, size=(12, 40)).astype(np.int8)
    unit = (1 - np.eye(20)).astype(np.uint32)
    m = engine.mst_scores(st20, unit)
    for p in range(40):
        k = len(set(int(v) for v in st20[:, p] if v < 20))
        assert m[p] == max(k - 1, 0)


def test_segmenting_and_remain_bounds():
    rng = np.random.default_rng(8)
    P = 5000
    ras = np.sort(rng.integers(1, 12, size=P))[::-1].astype(np.int32)      # sorted alignment: high scores first
    freq = rng.integers(1, 30, size=P).astype(np.int32)
    for vc in (8, 16):
        up = engine.segment_patterns(ras, freq, P, vc)
        assert up.tolist() == bounds_slow.do_segmenting(ras, freq, P, vc)
        assert len(up) > 1 and up[-1] == P
        lo = 0
        for u in up[:-1]:
            assert u % vc == 0
            s = int((ras[lo:u].astype(np.int64) * freq[lo:u]).sum())
            assert s > 4095                                                 # closed because it exceeded USHRT_MAX / 16
            assert int((ras[lo:u - vc].astype(np.int64) * freq[lo:u - vc]).sum()) <= 4095 or u - vc == lo
            lo = u
        minp = np.minimum(ras, rng.integers(0, 5, size=P)).astype(np.int32)
        w = rng.integers(0, 6, size=P).astype(np.uint16)
        rb = engine.remain_bounds(up, minp, w)
        assert rb.tolist() == bounds_slow.remain_bounds(up.tolist(), minp, w)
        full = int((minp.astype(np.int64) * w).sum())
        assert all(0 <= r <= full for r in rb) and list(rb) == sorted(rb, reverse=True)


def test_bad_arguments():
    with pytest.raises(engine.MpfError):
        engine.mst_scores(np.zeros((3, 3), dtype=np.int8), np.zeros((1, 1), dtype=np.uint32))


def _triangle_fix_plain(c):
    """ParsTree::loadCostMatrixFile's repair loop (parstree.cpp:74-80), transcribed for the check"""
    c = [[int(v) for v in row] for row in c]
    S, changed = len(c), False
    for k in range(S):
        for i in range(S):
            for j in range(S):
                if c[i][j] > c[i][k] + c[k][j]:
                    c[i][j] = c[i][k] + c[k][j]
                    changed = True
    return c, changed


def test_cost_matrix_loader_and_triangle_repair(tmp_path):
    """mpf_cost_matrix_load: keywords, a file, and the triangle-inequality repair (symmetric and asymmetric input)"""
    from mpboot_amd import engine
    for kw, S in (("fitch", 4), ("e", 20)):
        c, ch = engine.load_cost_matrix(kw, S)
        assert c.shape == (S, S) and not ch
        assert (c == (1 - np.eye(S, dtype=np.uint32))).all()
    rng = np.random.default_rng(3)
    for S, sym in ((4, True), (4, False), (20, True), (20, False)):
        m = rng.integers(1, 30, size=(S, S))
        if sym:
            m = np.triu(m, 1) + np.triu(m, 1).T
        np.fill_diagonal(m, 0)
        f = tmp_path / f"cost_{S}_{int(sym)}.txt"
        f.write_text(f"{S}\n" + "\n".join(" ".join(map(str, r)) for r in m.tolist()) + "\n")
        c, ch = engine.load_cost_matrix(str(f), S)
        exp, exp_ch = _triangle_fix_plain(m.tolist())
        assert c.tolist() == exp and ch == exp_ch
        if sym:
            assert (c == c.T).all()                   # the repair keeps a symmetric matrix symmetric
        assert all(c[i, j] <= c[i, k] + c[k, j] for i in range(S) for j in range(S) for k in range(S))
    with pytest.raises(engine.MpfError):
        engine.load_cost_matrix(str(tmp_path / "missing.txt"), 4)
    (tmp_path / "short.txt").write_text("4\n0 1 1\n")
    with pytest.raises(engine.MpfError):
        engine.load_cost_matrix(str(tmp_path / "short.txt"), 4)

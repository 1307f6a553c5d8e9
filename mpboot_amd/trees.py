"""Flat unrooted-binary-tree topologies in the engine's record convention.

A topology over n taxa is an int32 array ``back`` of length ``3*(2n-1)``:
record ``rec = 3*v + s`` is slot ``s`` (0..2) of node ``v``; tips are nodes
1..n (slot 0 only), inner nodes n+1..2n-2 (three slots in cyclic order
s -> (s+1)%3, the analogue of PLL's ``next`` ring, pllrepo/src/pll.h:622-701);
``back[rec]`` is the record across the branch (PLL's ``back``), -1 if unused.
"""
from __future__ import annotations

import numpy as np


def n_taxa_of(back: np.ndarray) -> int:
    return (len(back) // 3 + 1) // 2


def empty_back(n: int) -> np.ndarray:
    return np.full(3 * (2 * n - 1), -1, dtype=np.int32)


def nxt(rec: int) -> int:
    v, s = divmod(rec, 3)
    return 3 * v + (s + 1) % 3


def validate(back: np.ndarray, n: int | None = None) -> None:
    n = n or n_taxa_of(back)
    assert len(back) == 3 * (2 * n - 1)
    for v in range(1, 2 * n - 1):
        for s in range(1 if v <= n else 3):
            r = 3 * v + s
            b = int(back[r])
            assert b >= 3 and back[b] == r, f"broken link at rec {r}"
    # connectivity
    seen = set()
    stack = [3]
    while stack:
        r = stack.pop()
        v = r // 3
        if v in seen:
            continue
        seen.add(v)
        for s in range(1 if v <= n else 3):
            stack.append(int(back[3 * v + s]))
    assert len(seen) == 2 * n - 2, "tree is not connected"


def parse_newick(s: str):
    """-> nested lists of leaf names (branch lengths / inner labels dropped)."""
    s = s.strip().rstrip(";")
    pos = 0

    def node():
        nonlocal pos
        if s[pos] == "(":
            pos += 1
            kids = [node()]
            while s[pos] == ",":
                pos += 1
                kids.append(node())
            assert s[pos] == ")"
            pos += 1
            skip_label()
            return kids
        start = pos
        while s[pos] not in ",():;" if pos < len(s) else False:
            pos += 1
        name = s[start:pos]
        skip_len()
        return name

    def skip_len():
        nonlocal pos
        if pos < len(s) and s[pos] == ":":
            pos += 1
            while pos < len(s) and s[pos] not in ",()":
                pos += 1

    def skip_label():
        nonlocal pos
        while pos < len(s) and s[pos] not in ",():":
            pos += 1
        skip_len()

    t = node()
    return t


def newick_to_back(nwk: str, names: list[str]) -> np.ndarray:
    n = len(names)
    tip = {nm: i + 1 for i, nm in enumerate(names)}
    t = parse_newick(nwk)
    if len(t) == 2:       # rooted: dissolve the root
        a, b = t
        if isinstance(a, list):
            t = a + [b]
        else:
            t = b + [a]
    assert len(t) == 3, "need a binary tree"
    back = empty_back(n)
    counter = [n + 1]

    def link(a, b):
        back[a] = b
        back[b] = a

    def build(sub, parent_rec):
        if isinstance(sub, str):
            link(3 * tip[sub], parent_rec)
            return
        assert len(sub) == 2, "need a binary tree"
        v = counter[0]
        counter[0] += 1
        link(3 * v, parent_rec)
        build(sub[0], 3 * v + 1)
        build(sub[1], 3 * v + 2)

    root = counter[0]
    counter[0] += 1
    for s, sub in enumerate(t):
        build(sub, 3 * root + s)
    validate(back, n)
    return back


def back_to_newick(back: np.ndarray, names: list[str], start_tip: int = 1) -> str:
    n = len(names)

    def sub(rec):           # subtree hanging behind record rec (looking away from back[rec])
        v = rec // 3
        if v <= n:
            return names[v - 1]
        a, b = nxt(rec), nxt(nxt(rec))
        return "(" + sub(int(back[a])) + "," + sub(int(back[b])) + ")"

    r = int(back[3 * start_tip])
    v = r // 3
    if v <= n:
        return f"({names[start_tip - 1]},{names[v - 1]});"
    a, b = nxt(r), nxt(nxt(r))
    return f"({names[start_tip - 1]},{sub(int(back[a]))},{sub(int(back[b]))});"


def random_topology(n: int, rng: np.random.Generator) -> np.ndarray:
    """Random stepwise addition (uniform branch each step)."""
    back = empty_back(n)
    order = rng.permutation(n) + 1
    v = n + 1
    a, b, c = (int(x) for x in order[:3])
    for s, t in enumerate((a, b, c)):
        back[3 * v + s] = 3 * t
        back[3 * t] = 3 * v + s
    edges = [3 * a, 3 * b, 3 * c]          # one record per edge
    for t in order[3:]:
        t = int(t)
        v += 1
        e = edges[int(rng.integers(len(edges)))]
        f = int(back[e])
        back[3 * v] = 3 * t
        back[3 * t] = 3 * v
        back[3 * v + 1] = e
        back[e] = 3 * v + 1
        back[3 * v + 2] = f
        back[f] = 3 * v + 2
        edges.append(3 * t)
        edges.append(3 * v + 2)
    validate(back, n)
    return back


def splits(back: np.ndarray) -> frozenset:
    """Non-trivial bipartitions as frozensets of tip ids on the side not containing tip 1."""
    n = n_taxa_of(back)
    out = set()

    def tips_behind(rec):
        v = rec // 3
        if v <= n:
            return frozenset([v])
        a, b = nxt(rec), nxt(nxt(rec))
        s = tips_behind(int(back[a])) | tips_behind(int(back[b]))
        if 1 < len(s) < n - 1:
            out.add(s if 1 not in s else frozenset(range(1, n + 1)) - s)
        return s

    import sys
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 4 * n + 100))
    try:
        tips_behind(int(back[3]))
    finally:
        sys.setrecursionlimit(old)
    return frozenset(out)


def parse_topology_line(tokens: list[str], n: int) -> np.ndarray:
    """'rec:back' tokens as printed by oracle/ref_driver.c -> back array."""
    back = empty_back(n)
    for tok in tokens:
        r, b = tok.split(":")
        back[int(r)] = int(b)
    return back


def apply_spr(back: np.ndarray, p: int, q: int) -> np.ndarray:
    """Prune node record p (it keeps the subtree behind back[p]) and regraft it on branch (q, back[q])
    -- removeNodeParsimony + insertParsimony of the reference (sprparsimony.cpp:2245-2257, :1942-1952)."""
    b = np.array(back, dtype=np.int32).copy()
    a1, a2 = int(b[nxt(p)]), int(b[nxt(nxt(p))])
    b[a1], b[a2] = a2, a1
    r = int(b[q])
    b[nxt(p)], b[q] = q, nxt(p)
    b[nxt(nxt(p))], b[r] = r, nxt(nxt(p))
    return b


def random_spr_moves(engine_obj, back: np.ndarray, rng: np.random.Generator, k: int, maxtrav: int = 6) -> np.ndarray:
    """k random SPR moves within the given radius (a stand-in for the search's perturbation step): the prune node
    and the regraft branch are drawn uniformly from what rearrangeParsimony would test."""
    n = n_taxa_of(back)
    b = np.array(back, dtype=np.int32).copy()
    done = 0
    while done < k:
        v = int(rng.integers(n + 1, 2 * n - 1))
        rec = 3 * v + int(rng.integers(0, 3))
        engine_obj.set_tree(b)
        q, _mp, n_p = engine_obj.spr_scan(rec, 1, maxtrav)
        if n_p == 0:
            continue
        b = apply_spr(b, rec, int(q[int(rng.integers(0, n_p))]))
        done += 1
    return b

"""Seeded synthetic alignments and trees (SURVEY.md §8(d) "Concrete synthetic inputs").

Generator: a random bifurcating tree grown by repeatedly splitting a random
lineage; the root sequence is i.i.d. uniform over the alphabet; every child
copies its parent and replaces each site with probability ``r`` by a uniform
letter; only distinct, variable columns are kept, topped up until exactly
``n_patterns`` columns exist.  Every pattern has weight 1.

The named workloads of BASELINE.json:

    C1  plumbing stand-in for the absent example.phy : 17 taxa x  500 patterns, DNA
    C2  200 taxa x 10 000 patterns, DNA, r=0.05, seed 11
    C3  1000 taxa x 50 000 patterns, DNA, r=0.04, seed 3     (headline)
    C5  500 taxa x 20 000 patterns, protein, r=0.08, seed 9
    C4N 1000 taxa x 2 000 patterns, DNA, r=0.3, seed 17      (noisy: bootstrap samples that disagree)
"""
from __future__ import annotations

import numpy as np

DNA_LETTERS = "ACGT"
AA_LETTERS = "ARNDCQEGHILKMFPSTWYV"   # reference order, alignment.cpp:18

# PLL tip codes (pllrepo/src/utils.c:98-137)
_DNA_CODE = {"A": 1, "C": 2, "G": 4, "T": 8, "U": 8, "M": 3, "R": 5, "S": 6, "V": 7, "W": 9,
             "Y": 10, "H": 11, "K": 12, "D": 13, "B": 14, "-": 15, "?": 15, "N": 15, "O": 15, "X": 15}
_AA_CODE = {c: i for i, c in enumerate(AA_LETTERS)}
_AA_CODE.update({"B": 20, "Z": 21, "-": 22, "?": 22, "*": 22, "X": 22})

WORKLOADS = {
    "C1": dict(n_taxa=17, n_patterns=500, alphabet="DNA", r=0.08, seed=5),
    "C2": dict(n_taxa=200, n_patterns=10_000, alphabet="DNA", r=0.05, seed=11),
    "C3": dict(n_taxa=1000, n_patterns=50_000, alphabet="DNA", r=0.04, seed=3),
    "C5": dict(n_taxa=500, n_patterns=20_000, alphabet="AA", r=0.08, seed=9),
    # a bootstrap workload whose samples DISAGREE: C3's taxa, 2 000 columns, 30 % substitutions per branch -- the
    # samples of a -bb 1000 run keep some 230 distinct trees and nearly every refinement climbs (tools/noisy_probe.py tried
    # 1 500 x 0.1, 1 000 x 0.05, 5 000 x 0.15: one to three trees)
    "C4N": dict(n_taxa=1000, n_patterns=2_000, alphabet="DNA", r=0.3, seed=17),
}


def random_tree_parents(n_taxa: int, rng: np.random.Generator):
    """Rooted binary tree as (children list, leaf ids). Node 0 is the root."""
    children = {0: []}
    leaves = [0]
    nxt = 1
    while len(leaves) < n_taxa:
        k = int(rng.integers(len(leaves)))
        v = leaves[k]
        a, b = nxt, nxt + 1
        nxt += 2
        children[v] = [a, b]
        children[a] = []
        children[b] = []
        leaves[k] = a
        leaves.append(b)
    return children, leaves


def _newick(children, leaf_name, v=0):
    if not children[v]:
        return leaf_name[v]
    return "(" + ",".join(_newick(children, leaf_name, c) for c in children[v]) + ")"


def synth_alignment(n_taxa: int, n_patterns: int, alphabet: str = "DNA", r: float = 0.05,
                    seed: int = 1, return_tree: bool = False):
    """Returns (letters uint8[n_taxa, n_patterns] of indices into the alphabet, names[, newick])."""
    rng = np.random.default_rng(seed)
    A = 4 if alphabet == "DNA" else 20
    children, leaves = random_tree_parents(n_taxa, rng)
    order = []                       # preorder
    stack = [0]
    while stack:
        v = stack.pop()
        order.append(v)
        stack.extend(children[v])
    leaf_pos = {v: i for i, v in enumerate(leaves)}
    kept = np.zeros((n_taxa, 0), dtype=np.uint8)
    seen = set()
    while kept.shape[1] < n_patterns:
        m = max(1024, int((n_patterns - kept.shape[1]) * 1.3))
        seqs = {0: rng.integers(0, A, size=m, dtype=np.uint8)}
        out = np.empty((n_taxa, m), dtype=np.uint8)
        for v in order:
            s = seqs.pop(v)
            if not children[v]:
                out[leaf_pos[v]] = s
                continue
            for c in children[v]:
                mut = rng.random(m) < r
                repl = rng.integers(0, A, size=m, dtype=np.uint8)
                seqs[c] = np.where(mut, repl, s)
        variable = (out != out[0:1]).any(axis=0)
        out = out[:, variable]
        cols = []
        for j in range(out.shape[1]):
            key = out[:, j].tobytes()
            if key not in seen:
                seen.add(key)
                cols.append(j)
                if kept.shape[1] + len(cols) >= n_patterns:
                    break
        kept = np.concatenate([kept, out[:, cols]], axis=1)
    kept = np.ascontiguousarray(kept[:, :n_patterns])
    names = [f"t{i + 1}" for i in range(n_taxa)]
    if return_tree:
        leaf_name = {v: names[i] for v, i in leaf_pos.items()}
        return kept, names, _newick(children, leaf_name) + ";"
    return kept, names


def workload(name: str, return_tree: bool = False):
    cfg = WORKLOADS[name]
    return synth_alignment(cfg["n_taxa"], cfg["n_patterns"], cfg["alphabet"], cfg["r"], cfg["seed"],
                           return_tree=return_tree)


def letters_to_text(letters: np.ndarray, alphabet: str = "DNA") -> list[str]:
    lut = np.frombuffer((DNA_LETTERS if alphabet == "DNA" else AA_LETTERS).encode(), dtype=np.uint8)
    return [lut[row].tobytes().decode() for row in letters]


def text_to_codes(rows: list[str], alphabet: str = "DNA") -> np.ndarray:
    """Characters -> PLL tip codes (DNA: 4-bit masks 1..15; protein: 0..22)."""
    table = _DNA_CODE if alphabet == "DNA" else _AA_CODE
    lut = np.full(256, 255, dtype=np.uint8)
    for ch, code in table.items():
        lut[ord(ch)] = code
        lut[ord(ch.lower())] = code
    arr = np.stack([np.frombuffer(r.encode(), dtype=np.uint8) for r in rows])
    codes = lut[arr]
    if (codes == 255).any():
        raise ValueError("character outside the %s alphabet" % alphabet)
    return codes


def letters_to_codes(letters: np.ndarray, alphabet: str = "DNA") -> np.ndarray:
    if alphabet == "DNA":
        return (np.uint8(1) << letters).astype(np.uint8)
    return letters.astype(np.uint8)


def write_phylip(path: str, rows: list[str], names: list[str]) -> None:
    with open(path, "w") as f:
        f.write(f"{len(rows)} {len(rows[0])}\n")
        for nm, r in zip(names, rows):
            f.write(f"{nm} {r}\n")


def read_phylip(path: str):
    with open(path) as f:
        n, m = map(int, f.readline().split())
        names, rows = [], []
        for _ in range(n):
            nm, seq = f.readline().split()
            names.append(nm)
            rows.append(seq)
    assert all(len(r) == m for r in rows)
    return rows, names

"""Slow per-pattern Fitch on the IQ-TREE state encoding -- TEST INFRASTRUCTURE ONLY.

Restates PhyloTree::computeParsimonyScore(ptn, states, node, dad) (reference phylotree.cpp:1108-1158), the
recursive per-pattern algorithm the reference itself uses as the slow cross-check of its fast kernels
(parsmultistate.cpp:26-36), vectorised over patterns with numpy, and Alignment::convertState
(alignment.cpp:839-916).  Parity status: UNPINNED against a reference run (the IQ-TREE C++ layer needs the
CMake-generated iqtree_config.h and cannot be built from its sources alone); it is an independent second
algorithm whose totals must agree with the pinned PLL-style oracle.
"""
import sys

import numpy as np

DNA_STATE = {"A": 0, "C": 1, "G": 2, "T": 3, "U": 3, "R": 1 + 4 + 3, "Y": 2 + 8 + 3, "W": 1 + 8 + 3, "S": 2 + 4 + 3,
             "M": 1 + 2 + 3, "K": 4 + 8 + 3, "B": 2 + 4 + 8 + 3, "H": 1 + 2 + 8 + 3, "D": 1 + 4 + 8 + 3, "V": 1 + 2 + 4 + 3,
             "N": 18, "?": 18, "-": 18, ".": 18}
PROT_SYMBOLS = "ARNDCQEGHILKMFPSTWYVX"


MORPH_SYMBOLS = "0123456789ABCDEFGHIJKLMNOPQRSTUV"


def convert_states(rows, alphabet="DNA"):
    """characters -> Alignment::convertState codes, int8[n][L].  "BIN": '0' '1', anything else unknown (= 2, num_states);
    "MOR": the symbol's index, '-' unknown (= 32); a '?' is given the reading PLL's character map gives it (symbol 22,
    pllrepo/src/utils.c:142), so that the fixtures' PLL scores apply"""
    out = np.zeros((len(rows), len(rows[0])), dtype=np.int8)
    for i, r in enumerate(rows):
        for j, ch in enumerate(r.upper()):
            if alphabet == "DNA":
                out[i, j] = DNA_STATE[ch]
            elif alphabet == "BIN":
                out[i, j] = "01".index(ch) if ch in "01" else 2
            elif alphabet == "MOR":
                out[i, j] = 22 if ch == "?" else (MORPH_SYMBOLS.index(ch) if ch in MORPH_SYMBOLS else 32)
            elif ch in "?-.":
                out[i, j] = 22
            elif ch == "B":
                out[i, j] = 20
            elif ch == "Z":
                out[i, j] = 21
            else:
                k = PROT_SYMBOLS.index(ch)
                out[i, j] = k if k < 20 else 22
    return out


def _tip_sets(states, num_states):
    unknown = {4: 18, 20: 22}.get(num_states, num_states)       # STATE_UNKNOWN (alignment.cpp:512-518)
    s = states.astype(np.int64)
    sets = np.where(s < num_states, np.left_shift(1, np.minimum(s, num_states - 1)), 0)
    if num_states == 4:
        sets = np.where((s >= num_states) & (s != unknown), s - 3, sets)
    elif num_states == 20:
        sets = np.where(s == 20, 4 + 8, sets)
        sets = np.where(s == 21, 32 + 64, sets)
    sets = np.where(s == unknown, (1 << num_states) - 1, sets)
    return sets


def compute_parsimony(states, freq, back, num_states=4):
    """-> (weighted tree score, per-pattern scores) for the tree `back`, rooted on tip 1 as the reference does."""
    n, P = states.shape
    sets = _tip_sets(states, num_states)
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 4 * n + 100))

    def nx(r):
        v, s = divmod(r, 3)
        return 3 * v + (s + 1) % 3

    def down(rec):                       # subtree behind record rec
        v = rec // 3
        if v <= n:
            return sets[v - 1].copy(), np.zeros(P, dtype=np.int64)
        sa, ca = down(int(back[nx(rec)]))
        sb, cb = down(int(back[nx(nx(rec))]))
        inter = sa & sb
        empty = inter == 0
        return np.where(empty, sa | sb, inter), ca + cb + empty

    s_child, c_child = down(int(back[3]))
    inter = sets[0] & s_child
    ptn = c_child + (inter == 0)
    return int((ptn * np.asarray(freq, dtype=np.int64)).sum()), ptn.astype(np.uint16)

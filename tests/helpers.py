"""Shared helpers for the test-suite (fixture loading)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
# "morph": PLL_GENERIC_32 data with fewer than 20 symbols in use (20-row kernels, symbols renumbered); "morph32": all 32 symbols
# (32-row kernels, the reference's `default:` branches sprparsimony.cpp:824-869, :1164-1203)
# "morph32_40" / "aa_40": all 32 symbols / protein on 40 taxa (neighbourhoods the tree does not clip, long reference climbs)
FIXTURES = ("dna_clean", "dna_ambig", "dna_dups", "aa", "dna_48", "bin", "morph", "morph32", "morph32_40", "aa_40")
PLL_TYPES = {"DNA": 0, "WAG": 1, "BIN": 2, "MOR": 3}


def load_fixture(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        fx = json.load(f)
    fx["codes_np"] = np.array(fx["codes"], dtype=np.uint8)
    fx["weights_np"] = np.array(fx["weights"], dtype=np.int32)
    fx["datatype"] = PLL_TYPES[fx["pll_type"]]
    fx["name"] = name
    return fx


def hex_words(s):
    return np.array([int(s[i:i + 8], 16) for i in range(0, len(s), 8)], dtype=np.uint32)


def trace_tokens(q, mp):
    out = []
    for a, b in zip(q, mp):
        if a == -1:
            out.append("P")
        elif a == -2:
            out.append("Q")
        else:
            out.append(f"{int(a)}:{int(b)}")
    return out


def topology_splits(back, n):
    """the unrooted topology behind a record-link array as its set of bipartitions (each named by the side without tip 1)"""
    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 8 * n + 200))
    nxt = lambda r: 3 * (r // 3) + (r % 3 + 1) % 3
    out = set()

    def down(rec):
        if rec // 3 <= n:
            return frozenset([rec // 3])
        s = down(int(back[nxt(rec)])) | down(int(back[nxt(nxt(rec))]))
        out.add(s)
        return s

    down(int(back[3]))
    return frozenset(out)


def same_topology(a, b, n):
    """stored boot trees: the engine drops a tree no sample points to any more and stores the topology again (from whichever
    candidate reaches it next) when its index is taken up again, so two stores may number the inner nodes differently"""
    return topology_splits(a, n) == topology_splits(b, n)

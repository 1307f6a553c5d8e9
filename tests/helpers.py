"""Shared helpers for the test-suite (fixture loading)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
FIXTURES = ("dna_clean", "dna_ambig", "dna_dups", "aa", "dna_48")


def load_fixture(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        fx = json.load(f)
    fx["codes_np"] = np.array(fx["codes"], dtype=np.uint8)
    fx["weights_np"] = np.array(fx["weights"], dtype=np.int32)
    fx["datatype"] = 0 if fx["pll_type"] == "DNA" else 1
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

#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only where /root/reference exists (this container): it drives
oracle/_ref/pll_ref_driver and oracle/_ref/sprng_ref -- the reference's PLL
parsimony path and SPRNG generator compiled from their sources where they lie
(oracle/Makefile, target `ref`) -- on small seeded alignments and records what
the reference computed.  The fixtures are data only: inputs (alignment text,
PLL tip codes, topologies as record links) and the reference's outputs.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.json
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mpboot_amd import synth, trees  # noqa: E402

DRV = os.path.join(ROOT, "oracle", "_ref", "pll_ref_driver")
SPRNG = os.path.join(ROOT, "oracle", "_ref", "sprng_ref")
OUT = os.path.dirname(os.path.abspath(__file__))


def run(*args):
    r = subprocess.run([DRV, *map(str, args)], capture_output=True, text=True, check=True)
    return [l for l in r.stdout.splitlines() if not l.startswith("[PLL]")]


def topo(tokens, n):
    return trees.parse_topology_line(tokens, n).tolist()


def make_alignment(kind: str):
    """-> (rows of characters, names, pll type string, dedup flag)"""
    if kind == "dna_clean":
        L, names = synth.synth_alignment(17, 400, "DNA", 0.08, seed=5)
        return synth.letters_to_text(L, "DNA"), names, "DNA", 0
    if kind == "dna_ambig":
        rng = np.random.default_rng(21)
        L, names = synth.synth_alignment(14, 260, "DNA", 0.10, seed=6)
        rows = [list(r) for r in synth.letters_to_text(L, "DNA")]
        amb = "MRSVWYHKDBN-?"
        for r in rows:
            for j in range(len(r)):
                u = rng.random()
                if u < 0.05:
                    r[j] = "N" if rng.random() < 0.5 else "-"
                elif u < 0.09:
                    r[j] = amb[int(rng.integers(len(amb)))]
        # constant, singleton-variant and ambiguity-only columns (uninformative by the reference's rule)
        for r_i, r in enumerate(rows):
            r.extend(["A", "C" if r_i == 3 else "G", "N" if r_i % 2 else "T", "-"])
        return ["".join(r) for r in rows], names, "DNA", 0
    if kind == "dna_48":
        # a larger case: radius-6 neighbourhoods are not clipped by the tree size, deeper chains, gaps and N
        rng = np.random.default_rng(57)
        L, names = synth.synth_alignment(48, 500, "DNA", 0.07, seed=12)
        rows = [list(r) for r in synth.letters_to_text(L, "DNA")]
        for r in rows:
            for j in range(len(r)):
                u = rng.random()
                if u < 0.03:
                    r[j] = "N" if rng.random() < 0.5 else "-"
                elif u < 0.04:
                    r[j] = "RYKM"[int(rng.integers(4))]
        return ["".join(r) for r in rows], names, "DNA", 0
    if kind == "dna_dups":
        rng = np.random.default_rng(33)
        L, names = synth.synth_alignment(12, 120, "DNA", 0.12, seed=8)
        cols = rng.integers(0, 120, size=300)          # columns drawn with replacement -> weights > 1
        return synth.letters_to_text(L[:, cols], "DNA"), names, "DNA", 1
    if kind == "aa":
        rng = np.random.default_rng(44)
        L, names = synth.synth_alignment(11, 160, "AA", 0.15, seed=9)
        rows = [list(r) for r in synth.letters_to_text(L, "AA")]
        for r in rows:
            for j in range(len(r)):
                u = rng.random()
                if u < 0.03:
                    r[j] = "X-?"[int(rng.integers(3))]
                elif u < 0.05:
                    r[j] = "BZ"[int(rng.integers(2))]
        return ["".join(r) for r in rows], names, "WAG", 0
    if kind == "bin":
        # binary characters (PLL_BINARY_DATA, partition type "BIN": '0' '1', '-' / '?' undetermined; utils.c:78-97)
        rng = np.random.default_rng(71)
        L, names = synth.synth_alignment(16, 320, "DNA", 0.10, seed=14)
        rows = [["01"[c >> 1] for c in r] for r in L]
        for r in rows:
            for j in range(len(r)):
                if rng.random() < 0.04:
                    r[j] = "-?"[int(rng.integers(2))]
        for r_i, r in enumerate(rows):                     # constant / singleton / undetermined-only columns
            r.extend(["0", "1" if r_i == 2 else "0", "?" if r_i % 2 else "1"])
        return ["".join(r) for r in rows], names, "BIN", 0
    if kind == "aa_40":
        # protein on 40 taxa (the 20-row kernels beyond the 11 taxa of "aa"): B / Z / X and gaps, neighbourhoods the tree does not clip
        rng = np.random.default_rng(45)
        L, names = synth.synth_alignment(40, 380, "AA", 0.09, seed=23)
        rows = [list(r) for r in synth.letters_to_text(L, "AA")]
        for r in rows:
            for j in range(len(r)):
                u = rng.random()
                if u < 0.03:
                    r[j] = "X-?"[int(rng.integers(3))]
                elif u < 0.045:
                    r[j] = "BZ"[int(rng.integers(2))]
        return ["".join(r) for r in rows], names, "WAG", 0
    if kind == "morph32_40":
        # all 32 symbols again, on 40 taxa: radius-6 neighbourhoods that the tree does not clip, deeper refresh chains and longer
        # climbs for the 32-row kernels (the engine's S = 32 instantiations; "morph32" has 13 taxa)
        sym = "0123456789ABCDEFGHIJKLMNOPQRSTUV"
        rng = np.random.default_rng(91)
        L, names = synth.synth_alignment(40, 360, "AA", 0.10, seed=19)
        rows = [[sym[c] for c in r] for r in L]
        for r in rows:
            for j in range(len(r)):
                if j % 2 == 0 and rng.random() < 0.3:
                    r[j] = sym[20 + int(rng.integers(12))]
                if rng.random() < 0.03:
                    r[j] = "-?"[int(rng.integers(2))]
        return ["".join(r) for r in rows], names, "MOR", 0
    if kind in ("morph", "morph32"):
        # multistate characters (PLL_GENERIC_32, partition type "MOR": symbols 0-9 A-V, '-' / '?' undetermined; utils.c:138-157);
        # "morph" uses twelve symbols plus '?', which PLL_MAP_GENERIC_32 reads as symbol 22 ('M'), not as undetermined (utils.c:142:
        # only '-' and '*' map to 32); "morph32" uses all 32
        sym = "0123456789ABCDEFGHIJKLMNOPQRSTUV"
        rng = np.random.default_rng(83 if kind == "morph" else 84)
        L, names = synth.synth_alignment(13, 220, "AA", 0.14, seed=15 if kind == "morph" else 16)
        rows = [[sym[c % 12] if kind == "morph" else sym[c] for c in r] for r in L]
        if kind == "morph32":
            for r in rows:
                for j in range(len(r)):
                    if rng.random() < 0.25:
                        r[j] = sym[20 + int(rng.integers(12))] if j % 3 == 0 else r[j]
        for r in rows:
            for j in range(len(r)):
                if rng.random() < 0.03:
                    r[j] = "-?"[int(rng.integers(2))]
        return ["".join(r) for r in rows], names, "MOR", 0
    raise KeyError(kind)


def fixture(kind: str, tmp: str):
    rows, names, ptype, dedup = make_alignment(kind)
    n = len(rows)
    aln = os.path.join(tmp, kind + ".phy")
    synth.write_phylip(aln, rows, names)
    fx = {"name": kind, "pll_type": ptype, "dedup": dedup, "names": names, "rows": rows}

    # --- dump: encoded tips + packing
    codes, tipvec = {}, {}
    for l in run("dump", aln, ptype, dedup):
        t = l.split()
        if t[0] in ("n", "P", "S", "W"):
            fx[t[0]] = int(t[1])
        elif t[0] == "weights":
            fx["weights"] = list(map(int, t[1:]))
        elif t[0] == "informative":
            fx["informative"] = list(map(int, t[1:]))
        elif t[0] == "codes":
            codes[int(t[1])] = list(map(int, t[2:]))
        elif t[0] == "tipvec":
            tipvec[int(t[1])] = "".join(t[2:])
    fx["codes"] = [codes[i] for i in range(1, n + 1)]
    fx["tipvec_hex"] = [tipvec[i] for i in range(1, n + 1)]

    # --- score: random trees + a stepwise tree
    rng = np.random.default_rng(100 + n)
    tfile = os.path.join(tmp, kind + ".trees")
    with open(tfile, "w") as f:
        for _ in range(8):
            f.write(trees.back_to_newick(trees.random_topology(n, rng), names) + "\n")
    fx["trees"] = []
    sc = None
    for l in run("score", aln, ptype, dedup, tfile):
        t = l.split()
        if t[0] == "tree":
            sc = int(t[3])
        elif t[0] == "topology":
            fx["trees"].append({"back": topo(t[1:], n), "score": sc})

    # --- scan (radius 6 and 3) + spr from the first random tree
    one = os.path.join(tmp, kind + ".one")
    with open(tfile) as f, open(one, "w") as g:
        g.write(f.readline())
    fx["scan"] = []
    for rad in (6, 3):
        sc = {"maxtrav": rad, "best": [], "cands": []}
        for l in run("scan", aln, ptype, dedup, one, rad):
            t = l.split()
            if t[0] == "score":
                sc["score"] = int(t[1])
            elif t[0] == "topology":
                sc["back"] = topo(t[1:], n)
            elif t[0] == "order":
                sc["order"] = list(map(int, t[1:]))
            elif t[0] == "best":
                sc["best"].append(list(map(int, t[1:])))
            elif t[0] == "cands":
                sc["cands"].append(t[2:])
            elif t[0] == "score_after":
                sc["score_after"] = int(t[1])
        fx["scan"].append(sc)
    spr = {"maxtrav": 6, "moves": [], "sweeps": []}
    for l in run("spr", aln, ptype, dedup, one, 6):
        t = l.split()
        if t[0] == "start_score":
            spr["start_score"] = int(t[1])
        elif t[0] == "start_topology":
            spr["start_back"] = topo(t[1:], n)
        elif t[0] == "move":
            spr["moves"].append(list(map(int, t[2:])))
        elif t[0] == "sweep":
            spr["sweeps"].append([int(t[3]), int(t[5])])
        elif t[0] == "final_score":
            spr["final_score"] = int(t[1])
        elif t[0] == "final_topology":
            spr["final_back"] = topo(t[1:], n)
    fx["spr"] = spr

    # --- ras: the reference's own pllMakeParsimonyTreeFast
    fx["ras"] = []
    for seed, dist in ((42, 6), (7, 1), (99, 0), (2024, 3)):
        r = {"seed": seed, "spr_dist": dist}
        for l in run("ras", aln, ptype, dedup, seed, dist):
            t = l.split()
            if t[0] == "perm":
                r["perm"] = list(map(int, t[1:]))
            elif t[0] == "ras_score":
                r["score"] = int(t[1])
            elif t[0] == "ras_topology":
                r["back"] = topo(t[1:], n)
            elif t[0] == "ras_check":
                assert int(t[1]) == r["score"]
        fx["ras"].append(r)
    rx = {"seed": 42, "adds": []}
    for l in run("rasx", aln, ptype, dedup, 42):
        t = l.split()
        if t[0] == "add":
            rx["adds"].append([int(t[1]), int(t[3]), int(t[5]), int(t[7])])
        elif t[0] == "rasx_topology":
            rx["back"] = topo(t[1:], n)
        elif t[0] == "rasx_check":
            rx["score"] = int(t[1])
    fx["rasx"] = rx
    return fx


def main():
    with tempfile.TemporaryDirectory() as tmp:
        for kind in ("dna_clean", "dna_ambig", "dna_dups", "aa", "dna_48", "bin", "morph", "morph32", "morph32_40", "aa_40"):
            fx = fixture(kind, tmp)
            with open(os.path.join(OUT, kind + ".json"), "w") as f:
                json.dump(fx, f, separators=(",", ":"))
            print(kind, "n", fx["n"], "P", fx["P"], "W", fx["W"], "trees", len(fx["trees"]),
                  "spr moves", len(fx["spr"]["moves"]), "final", fx["spr"]["final_score"])
    rng = {}
    for seed in (1, 42, 12345, 2147483647):
        out = subprocess.run([SPRNG, str(seed), "12"], capture_output=True, text=True, check=True).stdout.split()
        rng[str(seed)] = [float(x).hex() for x in out]
    with open(os.path.join(OUT, "sprng_lcg64.json"), "w") as f:
        json.dump(rng, f)


if __name__ == "__main__":
    main()

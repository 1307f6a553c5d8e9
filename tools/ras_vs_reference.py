import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import load_fixture
from mpboot_amd import engine
for name in ["dna_clean", "dna_ambig", "dna_dups", "aa"]:
    fx = load_fixture(name)
    for r in fx["ras"]:
        e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
        e.seed_ties(engine.TIE_FIRST, 0)
        s = e.make_parsimony_tree(r["seed"], r["spr_dist"])
        same = (e.get_tree().tolist() == r["back"])
        print(name, r["seed"], r["spr_dist"], "engine", s, "reference", r["score"], "same tree", same)

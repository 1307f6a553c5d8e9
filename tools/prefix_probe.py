import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import load_fixture
from mpboot_amd import engine
for name in ("bin", "morph"):
    fx = load_fixture(name)
    e = engine.FitchEngine(fx["codes_np"], fx["weights_np"], datatype=fx["datatype"])
    print(name, "S W", e.S, e.W, "fixture", fx["S"], fx["W"])
    spr = fx["spr"]
    e.set_tree(np.array(spr["start_back"], dtype=np.int32)); e.seed_ties(engine.TIE_FIRST, 0)
    s = e.optimize_spr(1, spr["maxtrav"])
    got = [list(map(int, m)) for m in zip(*e.moves())]
    ref = spr["moves"]; k = 0
    while k < min(len(got), len(ref)) and got[k] == ref[k]: k += 1
    print(name, "prefix", k, "len ref", len(ref), "len got", len(got), "score", s, spr["final_score"])

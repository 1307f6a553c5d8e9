"""Shape extremes of the hot path against the CPU oracle (GPU box): ladder trees, thousands of taxa on a short alignment, a few
taxa on millions of patterns, very heavy pattern weights.  Prints one line per case; exits non-zero at the first difference."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpboot_amd import engine, synth, trees
from oracle import pyoracle as po


def ladder(n):
    """unrooted ladder ((((t1,t2),t3),t4) ..., t_{n-1}, t_n): the deepest tree on n taxa"""
    sys.setrecursionlimit(max(10000, 10 * n))
    names = [f"t{i+1}" for i in range(n)]
    inner = "(t1,t2)"
    for i in range(3, n - 1):
        inner = f"({inner},t{i})"
    return trees.newick_to_back(f"({inner},t{n-1},t{n});", names)


def climb_case(tag, codes, back, w=None, maxtrav=6, aa=False, tie=1, seed=3):
    dt_e, dt_o = (engine.AA, po.AA) if aa else (engine.DNA, po.DNA)
    e = engine.FitchEngine(codes, w, datatype=dt_e)
    o = po.Oracle(codes, w, datatype=dt_o)
    t0 = time.perf_counter(); se = e.score_tree(back); so = o.score_tree(back)
    assert se == so, (tag, "score", se, so)
    e.seed_ties(tie, seed); o.seed_ties(tie, seed); o.trace(True)
    t1 = time.perf_counter(); fe = e.optimize_spr(1, maxtrav); t2 = time.perf_counter(); fo = o.optimize_spr(1, maxtrav); t3 = time.perf_counter()
    ok = fe == fo and [x.tolist() for x in e.moves()] == [x.tolist() for x in o.get_moves()] and (e.get_tree() == o.get_tree()).all()
    pe, te = e.pattern_scores(); o.enable_persite(True); o.score_tree(); p_o, t_o = o.pattern_scores()
    inf = o.informative().astype(bool)
    ok = ok and te == t_o and (np.asarray(pe)[inf] == np.asarray(p_o)[inf]).all()
    print(f"{tag}: start {se} -> {fe} ({len(e.moves()[0])} moves) gpu {t2-t1:.2f}s oracle {t3-t2:.2f}s : {'same' if ok else 'DIFFERENT'}", flush=True)
    if not ok:
        sys.exit(1)


rng = np.random.default_rng(5)
FIRST = int(os.environ.get("MPF_EXTREMES_FIRST", "1"))
# 1. ladder trees (deepest possible), DNA and protein, radius 6 and 12
for n, P, aa, mt in ((300, 400, False, 6), (300, 400, False, 12), (150, 200, True, 6), (1200, 160, False, 6)) if FIRST <= 1 else ():
    letters, _ = synth.synth_alignment(n, P, "AA" if aa else "DNA", 0.08, seed=n + P)
    codes = synth.letters_to_codes(letters, "AA" if aa else "DNA")
    climb_case(f"ladder n={n} P={P} {'AA' if aa else 'DNA'} r={mt}", codes, ladder(n), maxtrav=mt, aa=aa)
# 2. thousands of taxa, short alignment, random start
for n, P in ((3000, 96), (6000, 40)) if FIRST <= 2 else ():
    letters, _ = synth.synth_alignment(n, P, "DNA", 0.03, seed=n)
    codes = synth.letters_to_codes(letters)
    climb_case(f"random n={n} P={P}", codes, trees.random_topology(n, rng), maxtrav=4 if n > 4000 else 6)
# 3. few taxa, millions of patterns (random columns: synth_alignment wants distinct patterns, 4^6 is all there are)
def columns(n, P, seed):
    g = np.random.default_rng(seed)
    L = np.repeat(g.integers(0, 4, size=P)[None, :], n, axis=0)
    mut = g.random((n, P)) < 0.35
    L[mut] = g.integers(0, 4, size=int(mut.sum()))
    return synth.letters_to_codes(L.astype(np.uint8))


for n, P in ((6, 3_000_000), (12, 1_200_000)):
    climb_case(f"random n={n} P={P}", columns(n, P, P), trees.random_topology(n, rng))
# 4. heavy weights
letters, _ = synth.synth_alignment(40, 600, "DNA", 0.1, seed=77)
codes = synth.letters_to_codes(letters)
w = rng.integers(0, 3, size=600).astype(np.int32)
w[rng.integers(0, 600, size=6)] = [65535, 40000, 100000, 1, 250000, 7]
climb_case("weights up to 250000, n=40 P=600", codes, trees.random_topology(40, rng), w=w)
print("all same")

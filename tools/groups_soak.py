"""Random cases for the round-6 forms of the climb kernel: k_climb on fewer workgroups than tiles (climb_groups) and many climbs in one
launch (mpf_optimize_spr_many), each against the solo workgroup-per-tile run -- moves, final tree, length, tie-stream state.
   python tools/groups_soak.py --seconds 120"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpboot_amd import engine, synth, trees
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t_end = time.time() + a.seconds
cases = bad = 0
while time.time() < t_end:
    alphabet = ["DNA", "DNA", "AA", "GEN"][int(rng.integers(0, 4))]
    n = int(rng.integers(5, 140))
    P = int(rng.integers(40, 20000 if alphabet == "DNA" else 1500))
    rate = float(rng.choice([0.03, 0.08, 0.2, 0.5]))
    if alphabet == "GEN":
        letters, _ = synth.synth_alignment(n, P, "AA", rate, seed=int(rng.integers(1 << 30)))
        codes = (letters % int(rng.integers(21, 33))).astype(np.uint8)      # up to 32 symbols: the 32-row kernels
        dt = engine.GENERIC
    else:
        letters, _ = synth.synth_alignment(n, P, alphabet, rate, seed=int(rng.integers(1 << 30)))
        codes = synth.letters_to_codes(letters, alphabet)
        dt = engine.DNA if alphabet == "DNA" else engine.AA
    tie = engine.TIE_RANDOM if rng.random() < 0.8 else engine.TIE_FIRST
    radius = int(rng.integers(1, 7))
    tile = int(rng.choice([1, 2, 4, 8])) if alphabet == "DNA" else 1
    k = int(rng.integers(2, 7))
    starts = [trees.random_topology(n, rng) for _ in range(k)]
    seeds = [int(rng.integers(1, 1 << 20)) for _ in range(k)]
    w = rng.integers(0, 3, size=P).astype(np.int32) if rng.random() < 0.3 else None
    if w is not None:
        w[: min(P, 6)] = 1

    auto_tile = bool(rng.random() < 0.6)          # mpf_optimize_spr_many picks the width (four-state data: the word-major shape)
    inside = int(rng.random() < 0.8)
    cap = int(rng.integers(3, 60)) if rng.random() < 0.3 else 0

    def mk(groups, many=False):
        e = engine.FitchEngine(codes, datatype=dt)
        e.set_option("climb_device", 2); e.set_option("climb_groups", groups)
        if not (many and auto_tile):
            e.set_option("climb_tile", tile)
        if many:
            e.set_option("many_sweeps_inside", inside); e.set_option("many_moves_cap", cap)
        if w is not None:
            e.set_weights(w)
        return e

    try:
        solo = []
        for j in range(k):
            e = mk(0)
            e.set_tree(starts[j]); e.reset_node_order(); e.seed_ties(tie, seeds[j])
            s = e.optimize_spr(1, radius)
            solo.append((s, [x.tolist() for x in e.moves()], e.get_tree().tolist(), e.tie_state()))
        g = int(rng.integers(1, 5))
        e = mk(g)
        e.set_tree(starts[0]); e.reset_node_order(); e.seed_ties(tie, seeds[0])
        s = e.optimize_spr(1, radius)
        ok1 = (s, [x.tolist() for x in e.moves()], e.get_tree().tolist(), e.tie_state()) == solo[0]
        engs = [mk(0, True) for _ in range(k)]
        for j, x in enumerate(engs):
            x.set_tree(starts[j]); x.reset_node_order(); x.seed_ties(tie, seeds[j])
        sc = engine.optimize_spr_many(engs, 1, radius)
        ok2 = all((int(sc[j]), [y.tolist() for y in engs[j].moves()], engs[j].get_tree().tolist(), engs[j].tie_state()) == solo[j] for j in range(k))
    except Exception as exc:
        ok1 = ok2 = False
        print("EXCEPTION", repr(exc))
    cases += 1
    if cases % 10 == 0:
        print(f"  ... {cases} cases, {bad} mismatches", flush=True)
    if not (ok1 and ok2):
        bad += 1
        print(f"MISMATCH: {alphabet} n {n} P {P} rate {rate} tie {tie} radius {radius} tile {tile} groups {g} k {k} weights {w is not None} many: auto tile {auto_tile} sweeps inside {inside} cap {cap}: groups ok {ok1}, many ok {ok2}", flush=True)
print(f"groups_soak: {cases} random cases (seed {a.seed}), {bad} mismatches", flush=True)

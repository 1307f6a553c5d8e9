#!/usr/bin/env python3
"""Device-resident climb (k_climb) against the host-driven batches of the same engine: moves, final tree and score must be
identical, move for move.  On a mismatch the first diverging move and the kernel's per-prune-node trace around it are printed.

    python tools/climb_check.py                     a set of small synthetic cases (seconds)
    python tools/climb_check.py --workload C2       a named workload from a random start tree, with timing
"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpboot_amd import engine, synth, trees

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="")
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--tile", type=int, default=1)
ap.add_argument("--mode", type=int, default=2)
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--radius", type=int, default=6)
ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--notrace", action="store_true")
a = ap.parse_args()


def run(codes, dt, back, mode, seed, tie, radius, trace=False, opts=()):
    e = engine.FitchEngine(codes, datatype=dt)
    e.set_option("climb_device", mode)
    e.set_option("climb_tile", a.tile)
    for k, v in opts:
        e.set_option(k, v)
    if trace:
        e.set_option("climb_trace", 1)
    s0 = e.score_tree(back)
    e.seed_ties(tie, seed)
    t0 = time.perf_counter()
    try:
        s = e.optimize_spr(1, radius)
    except engine.MpfError as ex:
        print("ERROR", ex, flush=True)
        os._exit(3)            # (a launch that did not come back: do not wait for it in the destructors)
    dt_s = time.perf_counter() - t0
    mv = e.moves()
    tr = e.scan_trace() if trace else None
    ph = [e.get_option(f"climb_phase_us{k}") for k in "0123456789abcdef"]
    if mode and not trace:
        print("   phase us (setup enum closure refresh scan exchange decide):", ph, "| refresh ops, closure rounds, invalidation rounds, chains:",
              [e.get_option(f"climb_ctr{k}") for k in range(4)], flush=True)
    return dict(s0=s0, s=s, moves=mv, tree=e.get_tree(), secs=dt_s, stats=e.stats(), trace=tr, eng=e)


def compare(name, codes, dt, back, seed, tie, radius):
    h = run(codes, dt, back, 0, seed, tie, radius)
    d = run(codes, dt, back, a.mode, seed, tie, radius, trace=not a.notrace, opts=[(k, int(v)) for k, v in (o.split("=") for o in a.opt)])
    hm = np.stack([np.asarray(x) for x in h["moves"]], axis=1) if len(h["moves"][0]) else np.zeros((0, 3), int)
    dm = np.stack([np.asarray(x) for x in d["moves"]], axis=1) if len(d["moves"][0]) else np.zeros((0, 3), int)
    ok = h["s"] == d["s"] and hm.shape == dm.shape and (hm == dm).all() and (h["tree"] == d["tree"]).all()
    st = d["stats"]
    print(f"{name}: start {h['s0']} -> host {h['s']} ({len(hm)} moves, {h['secs']*1e3:.1f} ms) | device {d['s']} ({len(dm)} moves, {d['secs']*1e3:.1f} ms; "
          f"launches {st['climb_launches']} steps {st['climb_steps']} nodes {st['climb_nodes']} dev moves {st['climb_moves']} {st['climb_ms_total']:.1f} ms) "
          f"{'OK' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        k = 0
        while k < min(len(hm), len(dm)) and (hm[k] == dm[k]).all():
            k += 1
        print(f"  first diverging move: index {k}; host {hm[k].tolist() if k < len(hm) else None} device {dm[k].tolist() if k < len(dm) else None}")
        tr = d["trace"]
        if tr is not None and len(tr):
            w = tr.view(np.uint32).reshape(-1, 8)
            acc = np.nonzero(w[:, 7])[0]
            lo = acc[k - 1] if 0 < k <= len(acc) else 0
            print("  trace (prune idx, prune cid, tests, p-side, min, best, sel, accepted) from the move before:")
            for row in w[lo:lo + 12]:
                print("   ", row.astype(np.int64).tolist())
    return ok


allok = True
if a.workload:
    cfg = synth.WORKLOADS[a.workload]
    letters, _ = synth.workload(a.workload)
    codes = synth.letters_to_codes(letters, cfg["alphabet"])
    dt = engine.DNA if cfg["alphabet"] == "DNA" else engine.AA
    for r in range(a.reps):
        back = trees.random_topology(codes.shape[0], np.random.default_rng(a.seed + r))
        allok &= compare(f"{a.workload} seed {a.seed + r}", codes, dt, back, a.seed + r, engine.TIE_RANDOM, a.radius)
else:
    cases = [(8, 200, "DNA", 0.1), (12, 300, "DNA", 0.1), (24, 600, "DNA", 0.08), (48, 900, "DNA", 0.06), (120, 3000, "DNA", 0.05),
             (17, 500, "AA", 0.1), (40, 700, "AA", 0.08)]
    for (n, P, alpha, r) in cases:
        letters, _ = synth.synth_alignment(n, P, alpha, r, seed=n)
        codes = synth.letters_to_codes(letters, alpha)
        dt = engine.DNA if alpha == "DNA" else engine.AA
        for seed in (1, 2):
            back = trees.random_topology(n, np.random.default_rng(seed))
            for tie in (engine.TIE_RANDOM, engine.TIE_FIRST):
                for radius in (6, 3):
                    allok &= compare(f"{alpha} {n}x{P} seed {seed} tie {tie} radius {radius}", codes, dt, back, seed, tie, radius)
print("ALL OK" if allok else "FAILURES")
sys.exit(0 if allok else 1)

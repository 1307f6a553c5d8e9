"""`-bb` as the reference runs it, on the GPU: N start trees -> candidate set -> doTreeSearch iterations (random NNIs / ratchet,
tracked climbs under the cut-off) -> refinement.   python tools/bb_reference_flow.py [--workload C3] [--trees 20] [--iters 40]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--trees", type=int, default=20)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--maxtrav", type=int, default=6)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--no-refine", action="store_true")
    args = ap.parse_args()
    from mpboot_amd import bootstrap, engine, synth
    from mpboot_amd.rng import Lcg64
    cfg = synth.WORKLOADS[args.workload]
    letters, _names = synth.workload(args.workload)
    codes = synth.letters_to_codes(letters, cfg["alphabet"])
    n, P = codes.shape
    eng = engine.FitchEngine(codes, datatype=engine.DNA if cfg["alphabet"] == "DNA" else engine.AA)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    w = np.ones(P, dtype=np.int32)
    samples = np.stack([bootstrap.bootstrap_weights(w, Lcg64(100 + b)) for b in range(args.samples)]).astype(np.uint16)
    t0 = time.perf_counter()
    starts = []
    for k in range(args.trees):
        eng.seed_ties(engine.TIE_RANDOM, 1 + k)
        s = eng.make_parsimony_tree(1 + (k + 1) * 12345, args.maxtrav)
        s = s[0] if isinstance(s, tuple) else s
        starts.append((eng.get_tree(), int(s)))
    print(f"{args.trees} start trees in {time.perf_counter() - t0:.2f} s: lengths {sorted(x[1] for x in starts)[:5]} ...", flush=True)
    r = bootstrap.bb_run(eng, samples, starts, args.iters, args.maxtrav, 1, verbose=True, refine=not args.no_refine)
    its = np.array([x["seconds"] for x in r["log"]])
    rat = np.array([x["ratchet"] for x in r["log"]])
    print(f"iterations: {len(its)} in {r['iterations_s']:.3f} s; NNI iterations {its[~rat].mean() * 1e3:.1f} ms mean, ratchet iterations "
          f"{its[rat].mean() * 1e3 if rat.any() else 0:.1f} ms mean; best {r['start_best_score']} -> {r['best_score']}; booked {r['saved_trees']}; "
          f"distinct boot trees {r['distinct_boot_trees']}; stop rule: {r['iterations_left_by_stop_rule']} more iterations at least "
          f"(unsuccess {r['unsuccess_iterations']}, last improvement at {r['last_improved_iteration']})")
    if "refine_s" in r:
        print(f"refinement {r['refine_s']:.3f} s, improved {r['samples_improved_by_refinement']} samples")


if __name__ == "__main__":
    main()

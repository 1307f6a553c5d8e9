"""How much does one GPU gain from running doTreeSearch iterations of a -bb run side by side (one engine + tracker per host thread,
different tie streams)?   python tools/bb_parallel_probe.py --workers 1,2,4,8 --iters 16"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--workers", default="1,2,4,8")
    ap.add_argument("--iters", type=int, default=16)
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--trees", type=int, default=6)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--switch", type=float, default=1e-4)
    args = ap.parse_args()
    from mpboot_amd import bootstrap, engine, synth
    from mpboot_amd.rng import Lcg64
    cfg = synth.WORKLOADS[args.workload]
    letters, _names = synth.workload(args.workload)
    codes = synth.letters_to_codes(letters, cfg["alphabet"])
    n, P = codes.shape
    dt = engine.DNA if cfg["alphabet"] == "DNA" else engine.AA
    w = np.ones(P, dtype=np.int32)
    samples = np.stack([bootstrap.bootstrap_weights(w, Lcg64(100 + b)) for b in range(args.samples)]).astype(np.uint16)
    pool = []
    def mk():
        e = engine.FitchEngine(codes, datatype=dt)
        for kv in args.opt:
            k, v = kv.split("=")
            e.set_option(k, int(v))
        return e
    pool.append(mk())
    starts = []
    for k in range(args.trees):
        pool[0].seed_ties(engine.TIE_RANDOM, 1 + k)
        s = pool[0].make_parsimony_tree(1 + (k + 1) * 12345, 6)
        starts.append((pool[0].get_tree(), int(s[0] if isinstance(s, tuple) else s)))
    for W in [int(x) for x in args.workers.split(",")]:
        while len(pool) < W:
            pool.append(mk())
        for rep in range(2):                       # first pass allocates
            out = [None] * W
            def work(i):
                out[i] = bootstrap.bb_run(pool[i], samples, starts, args.iters if rep else 4, 6, 1 + 977 * i, refine=False)
            sys.setswitchinterval(args.switch)
            th = [threading.Thread(target=work, args=(i,)) for i in range(W)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            dt_ = time.perf_counter() - t0
        its = np.concatenate([[x["seconds"] for x in o["log"]] for o in out])
        print(f"workers {W}: {W * args.iters} iterations in {dt_:.3f} s = {W * args.iters / dt_:.1f} iterations/s "
              f"(mean iteration {its.mean() * 1e3:.1f} ms)", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""K concurrent C3 climbs as K PROCESSES (one engine each) instead of K threads of one process: where does the concurrent-climb
throughput saturate -- in the process (HIP runtime) or on the chip?   python tools/conc_procs.py K [tile]"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    sys.path.insert(0, ROOT)
    import numpy as np
    from mpboot_amd import engine, synth, trees
    i, tile, go = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    WL = os.environ.get("MPF_WL", "C3")
    letters, _ = synth.workload(WL)
    codes = synth.letters_to_codes(letters, synth.WORKLOADS[WL]["alphabet"])
    e = engine.FitchEngine(codes)
    e.set_option("climb_device", 2); e.set_option("climb_tile", tile)
    n = codes.shape[0]
    def run(seed):
        e.set_tree(trees.random_topology(n, np.random.default_rng(7000 + seed))); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 1 + seed)
        return e.optimize_spr(1, 6)
    run(i)                                   # buffers, code
    print("READY", flush=True)
    while not os.path.exists(go):
        time.sleep(0.001)
    t0 = time.time()
    R = int(os.environ.get("MPF_ROUNDS", "3"))
    for r in range(R):
        run(i + 100 * r)
    print("DONE", t0, time.time(), flush=True)
    sys.exit(0)
K = int(sys.argv[1]); tile = int(sys.argv[2]) if len(sys.argv) > 2 else 4
go = "/tmp/mpf_go_%d" % os.getpid()
ps = [subprocess.Popen([sys.executable, __file__, "worker", str(i), str(tile), go], stdout=subprocess.PIPE, text=True) for i in range(K)]
import threading
def _reap():                                   # never leave workers behind: persistent kernels of processes that do not share an
    for p in ps:                               # admission gate can starve each other at their start barriers
        if p.poll() is None:
            p.kill()
_timer = threading.Timer(120.0, _reap)
_timer.daemon = True
_timer.start()
for p in ps:
    assert p.stdout.readline().startswith("READY")
open(go, "w").close()
ts = []
for p in ps:
    l = p.stdout.readline().split()
    ts.append((float(l[1]), float(l[2])))
    p.wait()
os.remove(go)
_timer.cancel()
_reap()
R = int(os.environ.get("MPF_ROUNDS", "3"))
wall = max(t[1] for t in ts) - min(t[0] for t in ts)
print(f"{K} processes x {R} climbs (tile {tile}): {wall:.3f} s -> {R * K / wall:.1f} climbs/s; per climb {sum(t[1]-t[0] for t in ts)/(R*K):.3f} s")

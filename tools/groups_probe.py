"""k_climb with fewer workgroups than tiles (option climb_groups): same trajectory, what a climb costs alone.
   python tools/groups_probe.py [--workload C2] [--groups 0,1,2,4] [--tile 1]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpboot_amd import engine, synth, trees
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C2")
ap.add_argument("--groups", default="0,1,2,4")
ap.add_argument("--tile", default="1")
ap.add_argument("--device", type=int, default=2)
a = ap.parse_args()
cfg = synth.WORKLOADS[a.workload]
letters, _ = synth.workload(a.workload)
codes = synth.letters_to_codes(letters, cfg["alphabet"])
dt = engine.DNA if cfg["alphabet"] == "DNA" else engine.AA
back = trees.random_topology(codes.shape[0], np.random.default_rng(1))
ref = None
for tile in [int(x) for x in a.tile.split(",")]:
    for g in [int(x) for x in a.groups.split(",")]:
        e = engine.FitchEngine(codes, datatype=dt)
        e.set_option("climb_device", a.device)
        e.set_option("climb_tile", tile)
        e.set_option("climb_groups", g)
        tt = []
        for _ in range(2):
            e.set_tree(back); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 1); e.reset_stats()
            t0 = time.perf_counter(); s = e.optimize_spr(1, 6); tt.append(time.perf_counter() - t0)
        st = e.stats()
        sig = (s, [x.tolist() for x in e.moves()], e.get_tree().tolist(), e.tie_state())
        if ref is None:
            ref = sig
        print(f"{a.workload} tile {tile} groups {g}: {min(tt) * 1e3:.1f} ms, score {s}, moves {st['moves_applied']}, launches {st['climb_launches']}, "
              f"steps {st['climb_steps']}, kernel {st['climb_ms_total']:.1f} ms, same as the first run: {sig == ref}", flush=True)

#!/usr/bin/env python3
"""Wall-clock of bootstrap replicates on the engine (refinement climbs or from-scratch searches), sharded over
ranks when launched under torch.distributed.run; optional oracle check/timing of the first replicates."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C3")
ap.add_argument("--replicates", type=int, default=100)
ap.add_argument("--mode", default="refine", choices=["refine", "search"])
ap.add_argument("--radius", type=int, default=6)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--cpu-replicates", type=int, default=0)
ap.add_argument("--workers", type=int, default=1, help="engines (host threads, HIP streams) per GPU")
a = ap.parse_args()

import torch
import torch.distributed as dist
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
if world > 1:
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
from mpboot_amd import engine, synth, bootstrap, shard

cfg = synth.WORKLOADS[a.workload]
letters, names = synth.workload(a.workload)
codes = synth.letters_to_codes(letters, cfg["alphabet"])
dt = engine.DNA if cfg["alphabet"] == "DNA" else engine.AA
n, P = codes.shape
w0 = np.ones(P, dtype=np.int32)
eng = engine.FitchEngine(codes, datatype=dt, device=local_rank)
eng.seed_ties(engine.TIE_RANDOM, a.seed)
best = eng.make_parsimony_tree(a.seed, a.radius)         # the tree every replicate is refined from
start = eng.get_tree()
if world > 1:
    dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
engines = [eng] + [engine.FitchEngine(codes, datatype=dt, device=local_rank) for _ in range(a.workers - 1)]
torch.cuda.synchronize()
t0 = time.perf_counter()
scores, trees = bootstrap.run_replicates(engines if a.workers > 1 else eng, w0, a.replicates, a.seed, a.radius, start, a.mode)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
dt_s = time.perf_counter() - t0
if rank == 0:
    res = {"workload": a.workload, "mode": a.mode, "replicates": a.replicates, "n_gpus": world, "workers": a.workers, "seconds": dt_s,
           "replicates_per_s": a.replicates / dt_s, "per_1000_replicates_s": 1000.0 * dt_s / a.replicates,
           "original_tree_score": int(best), "mean_replicate_score": float(np.mean(scores))}
    if a.cpu_replicates:
        from oracle import pyoracle as po
        from mpboot_amd.rng import Lcg64
        o = po.Oracle(codes, datatype=dt)
        t1 = time.perf_counter()
        ok = True
        for b in range(a.cpu_replicates):
            seed = shard.unit_seed(a.seed, b)
            o.set_weights(bootstrap.bootstrap_weights(w0, Lcg64(seed)))
            o.seed_ties(po.TIE_RANDOM, seed)
            if a.mode == "refine":
                o.set_tree(start)
                s = o.optimize_spr(1, a.radius)
            else:
                o.reset_nodep()
                s = o.make_tree(seed, a.radius)[0]
            ok = ok and (s == int(scores[b])) and (world > 1 or (o.get_tree() == trees[b]).all())
        t2 = time.perf_counter()
        res["cpu_port"] = {"replicates": a.cpu_replicates, "seconds": t2 - t1, "per_1000_replicates_s": 1000.0 * (t2 - t1) / a.cpu_replicates,
                           "identical_scores_and_trees": bool(ok), "cores": 1, "kind": "port (scalar C oracle)"}
        res["gpu_over_cpu"] = res["cpu_port"]["per_1000_replicates_s"] / res["per_1000_replicates_s"]
    print(json.dumps(res))
if world > 1:
    dist.destroy_process_group()

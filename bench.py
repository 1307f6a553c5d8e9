#!/usr/bin/env python3
"""bench.py -- Fitch site-ops/s of the SPR neighbourhood scan on MI355X.

A "step" = one full radius-6 SPR sweep scan of one tree (every prune node, both
sides: what pllOptimizeSprParsimony does between two accepted moves,
reference sprparsimony.cpp:3295-3316): refresh of all directional vectors,
scan programs, the k_scan launch and the copy-back of every candidate's score.
Inputs (packed tips, topology) are resident in HBM before the timed region.

Metric (BASELINE.json): Fitch site-ops/s = taxa x patterns x SPR-evals/s.
N > 1: every rank holds the alignment and scans its own start tree
(independent units, no data-path collective) -> weak scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C3]
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# engines on several host threads launch concurrently only as far as their streams get hardware queues of their own (the
# runtime's default is 4 per process; a persistent climb kernel holds its queue for a whole sweep: profiles/r3/concurrent_climbs.txt)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

I8_PEAK_TOPS = 5000.0      # dense int8 MFMA, 2 x bf16 (MI355X_MICROARCH.md, matrix cores)
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)
# 32-bit integer lane-ops/s: 256 CUs x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md "Wave scheduling": a wave64 VALU instruction issues
# over 2 cycles; tools/ubench/valu_rate.hip measures v_bitop3_b32 / v_and_b32 at that rate and v_bcnt / DPP / v_and_or at half of it)
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12
L2_PEAK_GBS = 34500.0     # aggregate L2 -> CU bandwidth (MI355X_MICROARCH.md, "L2 (per XCD)": ~34.5 TB/s; its measured gather-from-L2
                          # rates are 16.8-18.8 TB/s chip-wide)


def source_hash() -> str:
    """Hash of the device code: an offline PMC figure is only quoted while it still describes this build."""
    import hashlib
    h = hashlib.sha256()
    for rel in ("mpboot_amd/csrc/kernels.hip", "mpboot_amd/csrc/kernels.hpp", "mpboot_amd/csrc/engine.cpp", "mpboot_amd/csrc/ufboot.hip"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def offline_traffic(workload: str, kernel_prefix: str):
    """Bytes that left the L2s per launch of the named kernel family (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes,
    tools/profile_gpu.sh -> profiles/r3/traffic.json); None unless that file was made from exactly these sources."""
    for k, v in LIVE_TRAFFIC.items():               # measured at the start of this run
        if k.startswith(kernel_prefix):
            return v
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):      # newest round first
        try:
            with open(path) as f:
                tj = json.load(f)
        except (OSError, ValueError):
            continue
        if tj.get("workload") != workload or tj.get("source_hash") != source_hash():
            continue
        for k, v in tj.get("kernels", {}).items():
            if k.startswith(kernel_prefix):
                return v.get("bytes_per_launch")
    return None


LIVE_TRAFFIC = {}        # kernel family -> bytes per launch, measured by live_traffic() at the start of this very run


def live_traffic(args) -> dict:
    """HBM traffic of the step's kernels measured NOW: two short copies of this bench under `rocprofv3 --pmc` (FETCH_SIZE, then
    WRITE_SIZE: separate passes, counters only, the program itself behind `--`), before this process touches the GPU.  Bytes per
    launch = 2 x FETCH_SIZE (gfx950: the counter tallies 128-B requests at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE, both in KB.
    Returns {} when the profiler is not there or a pass fails -- the figures of profiles/r3/traffic.json (same sources) are used then."""
    import csv
    import glob
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}
    tmp = tempfile.mkdtemp(prefix="mpf_traffic_")
    env = dict(os.environ)
    env["MPF_BENCH_TRAFFIC_CHILD"] = "1"
    env.setdefault("TMPDIR", "/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu", "--bootstrap-replicates", "0",
             "--ufboot-samples", "0", "--random-start-leg", "0", "--weighted-leg", "0", "--start-trees", "0", "--climb-engines", "0",
             "--workload", args.workload, "--maxtrav", str(args.maxtrav), "--tree-cache", os.path.join(tmp, "tree")]
    for kv in args.opt:
        child += ["--opt", kv]
    acc = {}
    try:
        # the start tree first, without the profiler (its ~1500 small dispatches crawl under counter collection); the passes read it
        r0 = subprocess.run(child + ["--steps", "1", "--warmup", "0"], env=env, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        if r0.returncode != 0:
            return {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child, env=env, cwd=tmp,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
            if r.returncode != 0:
                return {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") != counter:
                            continue
                        name = row["Kernel_Name"].replace("void ", "").replace("mpf::", "")
                        key = name.split("<")[0].split("(")[0]
                        acc.setdefault(key, {}).setdefault(counter, []).append((int(row.get("Grid_Size") or 0), float(row["Counter_Value"])))
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    def sweep_mean(rows):
        # only the dispatches of the timed sweep: the child runs nothing but sweep steps (every secondary leg is off), and of those
        # only the launches with the kernel's most frequent grid count (a start-tree build that was not cached would add small ones)
        grids = [g for g, _ in rows]
        mode = max(set(grids), key=grids.count)
        vals = [x for g, x in rows if g == mode]
        return sum(vals) / len(vals)

    out = {}
    for key, v in acc.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            out[key] = int((2.0 * sweep_mean(v["FETCH_SIZE"]) + sweep_mean(v["WRITE_SIZE"])) * 1024)
    return out


def st_kernel_name(eng) -> str:
    """Name of the scan kernel the engine's sweeps launch (mpf_get_option "scan_prog": 1 = planned program, 0 = device walk)."""
    try:
        # (the planned program exists for the 4-row kernels: DNA and binary data; protein sweeps run the device walk)
        return "k_scan_prog" if eng.get_option("scan_prog") and eng.S == 4 else "k_scan_walk"
    except Exception:
        return "k_scan_walk"


from benchlegs.cpu import (bb_run_cpu_baseline, climb_cpu_baseline, cpu_baseline, cpu_quota, physical_cores,  # noqa: E402
                           refine_cpu_baseline, shim_climb_leg, start_trees_cpu_baseline)


def launch_ranks(n: int) -> int:
    """Parent of a multi-GPU run: N child ranks through torch.distributed.run (one process per GPU, RCCL), stdout relayed.
    The parent itself never touches the GPU (no torch import, no HIP call) and never re-executes itself."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        # stdout carries the ONE result line; whatever else the ranks' libraries print there (gloo's connection notes) goes to
        # stderr
        if out.startswith("{"):
            line = out
            sys.stdout.write(out)
            sys.stdout.flush()
        else:
            sys.stderr.write(out)
            sys.stderr.flush()
    rc = proc.wait()
    if rc == 0 and line is None:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 4
    return rc


def dry_run(args, rank, world, dist, torch):
    """MPF_BENCH_DRYRUN=1 (CPU container tests of the launcher only): the ranks rendezvous, exchange their timing
    tensors exactly as the real run does and rank 0 prints a line marked dry_run -- no engine, no measurement."""
    backend = os.environ.get("MPF_BENCH_BACKEND", "gloo")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)
    tt = torch.tensor([1.0 + rank, 100.0], dtype=torch.float64)
    tmax, tsum = tt.clone(), tt.clone()
    if world > 1:
        dist.barrier()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "backend": backend if world > 1 else None,
                          "dist_world_size": dist.get_world_size() if world > 1 else 1,
                          "max_time": float(tmax[0]), "sum_tests": float(tsum[1]), "steps": args.steps, "warmup": args.warmup}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--maxtrav", type=int, default=6)
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--start-tree", default="ras", choices=["ras", "random"])
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value")
    ap.add_argument("--tree-cache", default=None,
                    help="file to keep rank 0's start tree in (written if absent, read if present): the counter passes of "
                         "tools/profile_gpu.sh re-use the RAS tree of the first pass instead of rebuilding it under the profiler")
    ap.add_argument("--bootstrap-replicates", type=int, default=1000,
                    help="samples of the online phase whose trees are refined afterwards (IQTree::optimizeBootTrees), "
                         "sharded over the GPUs; capped by --ufboot-samples (0 = skip)")
    ap.add_argument("--random-start-leg", type=int, default=1,
                    help="1 = also run the climb / -bb flow from a random topology (a start tree that is not already SPR-optimal)")
    ap.add_argument("--legs-timeout", type=int, default=180,
                    help="multi-GPU runs: seconds after which the headline line is printed without the secondary (-bb) legs")
    ap.add_argument("--engines-per-gpu", type=int, default=6, help="concurrent engines (host threads) per GPU in the refinement leg")
    ap.add_argument("--climb-engines", type=int, default=8,
                    help="independent SPR climbs side by side on one GPU (one engine per host thread): the concurrent_climbs leg (0 = skip)")
    ap.add_argument("--climb-tile", type=int, default=4, help="words per lane group of k_climb in the concurrent-climbs leg (4: 25, 8: 13 workgroups per C3 climb)")
    ap.add_argument("--c2-engines", type=int, default=16, help="concurrent engines (host threads) in the C2 concurrent-climbs leg")
    ap.add_argument("--start-engines", type=int, default=12,
                    help="concurrent engines (host threads) per GPU in the start-trees leg (k_grow: 25 workgroups per C3 tree, two trees per CU)")
    ap.add_argument("--shard-online", default="auto", choices=["auto", "0", "1"],
                    help="online UFBoot phase on several GPUs: 1 = samples sharded, events all-gathered per scan batch; 0 = every rank keeps all samples "
                         "(replicas); auto = sharded from shard.ONLINE_SHARD_MIN_SAMPLES samples on (below, the exchange costs the pipelined climb more than the share saves)")
    ap.add_argument("--weighted-leg", type=int, default=1, help="1 = also time the weighted (Sankoff, -cost) sweep of config 5")
    ap.add_argument("--start-trees", type=int, default=100,
                    help="randomized-stepwise-addition + SPR start trees of the start-up phase (phyloanalysis.cpp:1270-1317), sharded over the GPUs (0 = skip)")
    ap.add_argument("--bb-iterations", type=int, default=200,
                    help="bb_reference_run leg, ONE chain: doTreeSearch iterations timed (random NNIs / ratchet alternating, tracked climbs "
                         "under the cut-off); the stop rule's horizon is extrapolated from them (0 = skip)")
    ap.add_argument("--bb-workers", type=int, default=16, help="bb_reference_run leg, iteration-parallel form: chains (engines on host threads) per GPU")
    ap.add_argument("--bb-rounds", type=int, default=4, help="... rounds timed (a round = --bb-sync iterations on every chain, then one exchange; 0 = skip)")
    ap.add_argument("--bb-sync", type=int, default=8, help="... iterations between two exchanges")
    ap.add_argument("--many-c2", type=int, default=512, help="climbs_in_one_launch leg: C2 climbs per call")
    ap.add_argument("--many-c3", type=int, default=256, help="climbs_in_one_launch leg: C3 climbs per call (an engine each: 0.25 GB; 0 = skip)")
    ap.add_argument("--legs", default="all",
                    help="comma-separated secondary legs to run (default all): ufboot_online (with the refinement of its trees), random_start, concurrent_climbs, start_trees, bb_reference_run, c2_climb, "
                         "c5_weighted_sweep, c5_fitch, noisy_bootstrap, climbs_in_one_launch; the headline step, its roofline and cpu_baseline always run")
    ap.add_argument("--ufboot-samples", type=int, default=1000,
                    help="bootstrap samples of the online UFBoot-MP leg (-bb): one pllOptimizeSprParsimony call with "
                         "saveCurrentTree after every insertion test, timed after the main metric (0 = skip)")
    args = ap.parse_args()
    legs_wanted = set(x.strip() for x in args.legs.split(","))
    leg_on = lambda name: "all" in legs_wanted or name in legs_wanted
    # wall clock of the run, leg by leg (what the driver's own clock around this command is made of)
    _laps, _lap_at = [], [time.perf_counter(), "set-up + headline"]

    def _lap(name):
        now = time.perf_counter()
        _laps.append((_lap_at[1], now - _lap_at[0]))
        _lap_at[0], _lap_at[1] = now, name

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this process (which has made NO GPU call and never imports torch)
        # starts the N ranks as children and relays rank 0's JSON line
        sys.exit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")

    traffic_source = "profiles/r*/traffic.json (rocprofv3 --pmc passes of tools/profile_gpu.sh on the same sources)"
    # (under a profiler this process may hold the GPU already -- the counter tool initialises it before main() -- and must not
    #  start programs any more: the committed figures are quoted then)
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if (world == 1 and os.environ.get("MPF_BENCH_DRYRUN") != "1" and os.environ.get("MPF_BENCH_TRAFFIC_CHILD") != "1"
            and os.environ.get("MPF_BENCH_LIVE_TRAFFIC", "1") != "0" and not profiled):
        # before this process makes its first GPU call: the profiler runs are child processes
        LIVE_TRAFFIC.update(live_traffic(args))
        if LIVE_TRAFFIC:
            traffic_source = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench command, run as child processes at the start of this run"

    import torch
    import torch.distributed as dist

    if os.environ.get("MPF_BENCH_DRYRUN") == "1":
        return dry_run(args, rank, world, dist, torch)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libmpfitch has no CPU fallback)")
    # one process per GPU; MPF_BENCH_SHARE_GPU=1 (testing only) lets several ranks share the visible GPUs
    ndev = torch.cuda.device_count()
    device = local_rank % ndev if os.environ.get("MPF_BENCH_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(device)
    backend = os.environ.get("MPF_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    from mpboot_amd import engine, shard, synth, trees

    cfg = synth.WORKLOADS[args.workload]
    alphabet = cfg["alphabet"]
    letters, names = synth.workload(args.workload)
    codes = synth.letters_to_codes(letters, alphabet)
    n, P = codes.shape
    eng = engine.FitchEngine(codes, datatype=engine.DNA if alphabet == "DNA" else engine.AA, device=device)
    eng.set_option("timing", 1)                # HIP events around the scan kernel (roofline.achieved needs its duration)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    # each rank scans its own start tree (independent SPR start trees shard across GPUs): the randomized
    # stepwise-addition tree the reference would hand to pllOptimizeSprParsimony (built on the GPU, untimed)
    cache = f"{args.tree_cache}.{args.workload}.{args.start_tree}.{rank}.npy" if args.tree_cache else None
    if cache and os.path.exists(cache):
        back = np.load(cache)
    elif args.start_tree == "ras":
        eng.seed_ties(engine.TIE_RANDOM, 1 + rank)
        eng.make_parsimony_tree(12345 + 7919 * rank, 0)
        back = eng.get_tree()
        if cache:
            np.save(cache, back)
    else:
        back = trees.random_topology(n, np.random.default_rng(1000 + rank))
    eng.set_tree(back)
    start_score = eng.score_tree()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # initialisation, not steps: the first sweeps allocate the pinned/device scratch buffers and load the kernels' code
    for _ in range(8):
        eng.set_tree(back)
        eng.sweep_scan(1, args.maxtrav)
    # Side numbers first (the same step on a planned topology; the refresh kernels' own time), the headline after them.
    eng.set_option("plan_cache", 1)
    wsteps = max(1, min(20, args.steps))
    for _ in range(3):
        eng.set_tree(back)
        eng.sweep_scan(1, args.maxtrav)
    torch.cuda.synchronize()
    tc0 = time.perf_counter()
    for _ in range(wsteps):
        eng.set_tree(back)
        eng.sweep_scan(1, args.maxtrav)
    torch.cuda.synchronize()
    warm_ms = (time.perf_counter() - tc0) / wsteps * 1e3
    # the refresh kernels' own time comes from a short untimed pass: their event pair would cost the timed steps ~10 us each
    eng.set_option("timing", 2)
    eng.reset_stats()
    vsteps = max(1, min(5, args.steps))
    for _ in range(vsteps):
        eng.set_tree(back)
        eng.sweep_scan(1, args.maxtrav)
    torch.cuda.synchronize()
    view_ms = eng.stats()["view_kernel_ms_total"] / vsteps
    eng.set_option("timing", 1)

    # The timed steps hand one tree over again and again.  What the engine plans from the topology alone (refresh schedule, scan
    # descriptors, device program) would be planned once and reused -- as it is between bootstrap replicates that share a tree, or
    # after a re-weighting, in a real run -- but a plain search sees a NEW topology after every accepted move, so the headline is
    # timed with that reuse switched off (engine option plan_cache = 0: every step plans from scratch, everything else identical).
    # The short loop above priced the step with the reuse on; both figures go into the line.
    eng.set_option("plan_cache", 0)
    for _ in range(args.warmup):                 # W untimed warm-up steps, the same kind of step as the timed ones
        eng.set_tree(back)
        eng.sweep_scan(1, args.maxtrav)
    eng.reset_stats()
    barrier()
    t0 = time.perf_counter()
    tests = 0
    for _ in range(args.steps):
        eng.set_tree(back)                       # invalidates the views: the step recomputes them
        k, _best = eng.sweep_scan(1, args.maxtrav)
        tests += k
    barrier()
    dt = time.perf_counter() - t0
    sched_us = (eng.get_option("sched_ticks") / 100.0, eng.get_option("sched_desc_ticks") / 100.0, eng.get_option("sched_levels"))
    st = eng.stats()
    eng.set_option("plan_cache", 1)

    tt = torch.tensor([dt, float(tests)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_all, tests_all = float(tmax[0]), float(tsum[1])
    else:
        dt_all, tests_all = dt, float(tests)

    core_res = None
    if rank == 0:
        evals_per_s = tests_all / dt_all
        W = eng.W                                          # the reference's parsimonyLength (32-site words per state row)
        launches = max(1, st["scan_launches"])
        scan_ms = st["scan_kernel_ms_total"] / launches
        evals_per_launch = st["insertion_tests"] / launches
        scan_s = scan_ms * 1e-3
        # ---- what bounds the scan kernel: VALU issue.  The directional-vector formulation loads ONE
        # vector per insertion test and >90 % of that is served by the XCD-local L2, so HBM is nowhere near a limit; the
        # arithmetic is.  Algorithmic 32-bit lane-operations per insertion test and 32-site word (v_bitop3 counted as one):
        #   chain step  U' = fitch(U, sibling)          2 per state
        #   join cost   popcount(~OR_k(fitch(U', own)_k & s_k))   3 per state + 1 popcount
        #   wave reduction of the count                 6 DPP adds per PAIR of tests (two 16-bit counts per dword) = 3
        ops_per_eval_word = 5 * eng.S + 1 + 3
        lane_ops = evals_per_launch * W * ops_per_eval_word
        achieved_valu = lane_ops / scan_s / 1e12 if scan_s > 0 else 0.0
        # SURVEY 8(d)'s byte figure (6 vectors per test) stays as a labelled side number: the kernel does not move those bytes
        survey_bytes_per_eval = 6 * eng.S * W * 4
        survey_gbps = evals_per_launch * survey_bytes_per_eval / scan_s / 1e9 if scan_s > 0 else 0.0
        loaded_gbps = evals_per_launch * eng.S * eng.Wp * 4 / scan_s / 1e9 if scan_s > 0 else 0.0   # L2-side loads: 1 vector per test
        # bytes that MUST come from HBM per launch: every directional vector of the tree once + the candidates' costs
        n_vec = n + 3 * (n - 2)
        compulsory = n_vec * eng.S * eng.Wp * 4 + evals_per_launch * 4
        traffic = offline_traffic(args.workload, st_kernel_name(eng)) if (LIVE_TRAFFIC or not args.opt) else None
        res = {
            "metric": "Fitch site-ops/sec (taxa x patterns x SPR-evals/s)",
            "value": n * P * evals_per_s,
            "unit": "site-ops/s",
            "n_gpus": world,
            "dist": {"world_size": dist.get_world_size() if world > 1 else 1, "backend": backend if world > 1 else None},
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_all / args.steps * 1e3,
            "ms_per_step_same_topology": warm_ms,      # rank 0, untimed side loop: plans of the topology reused (engine option plan_cache = 1)
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {n} taxa x {P} {alphabet} patterns, SPR radius {args.maxtrav}, "
                                   "one full sweep scan per step (all prune nodes, both sides)",
                       "evals_per_step": tests_all / args.steps / world, "evals_per_s": evals_per_s,
                       "planning": "every step plans from scratch (engine option plan_cache = 0): the host uploads the topology array, the "
                                   "device makes the refresh schedule, the scan descriptors and the DFS programs (k_sched; the walk plan rides on "
                                   "the refresh launch), recomputes every vector and re-scores every insertion test -- the step of a search that "
                                   "has just accepted a move.  ms_per_step_same_topology = the same step on a topology the engine has planned "
                                   "before (a re-weighted or re-evaluated tree)",
                       "start_tree_score": start_score, "parallelism": f"independent start trees x{world}",
                       # the three rates side by side.  value = effective (n x P x evals/s, as defined);
                       # touched = node-vector operations actually performed x P: per eval one chain step (fitch of the
                       # running up-vector with a sibling) + one join, plus the directional-vector refresh of the step
                       "touched_site_ops_per_s": (2.0 * tests_all + st["newview_ops"] * world) * P / dt_all},
            # One directional vector per insertion test has to reach a CU's registers, >90 % of them from the XCD's L2.  Round 2's
            # kernel (four 256-byte row loads per vector) sat on the CU's load path: 75 GB/s per CU whatever the cache level.  Round 3
            # reads a word-major copy (one buffer_load_dwordx4 per vector and lane: 124-146 GB/s per CU, tools/ubench/l1_rate) and
            # is now close to its arithmetic + control side (profiles/r3/scan_bounds.txt).  achieved = bytes the kernel loads / its time.
            "roofline": {"bound": "l2", "achieved": loaded_gbps, "peak": L2_PEAK_GBS, "unit": "GB/s",
                         "frac": loaded_gbps / L2_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source if traffic is not None else None,
                         "kernel": st_kernel_name(eng), "kernel_ms_per_launch": scan_ms, "evals_per_launch": evals_per_launch,
                         "plan_kernel_ms_per_launch": st["plan_kernel_ms_total"] / max(1, st["plan_launches"]),
                         "loaded_bytes_per_launch": evals_per_launch * eng.S * eng.Wp * 4,
                         "valu": {"achieved": achieved_valu, "peak": VALU_PEAK_TOPS, "unit": "T lane-op/s", "frac": achieved_valu / VALU_PEAK_TOPS,
                                  "lane_ops_per_eval_word": ops_per_eval_word,
                                  # the issue side priced with MEASURED issue times (tools/ubench/valu_rate, profiles/r3/valu_rate.txt): per
                                  # insertion test and 64-word tile about 20 wave-instructions of the 1.1 ns class (v_bitop3_b32, v_and) and 6
                                  # of the 1.76 ns class (v_bcnt, the DPP adds of the reduction, readlane / writelane) per SIMD
                                  "issue_ms_per_launch_measured_rates": evals_per_launch * ((eng.Wp + 63) // 64) * (20 * 1.1e-9 + 6 * 1.76e-9) / 1024 * 1e3,
                                  "frac_of_kernel_time_at_measured_rates": (evals_per_launch * ((eng.Wp + 63) // 64) * (20 * 1.1e-9 + 6 * 1.76e-9) / 1024) / scan_s if scan_s > 0 else None,
                                  "note": "peak = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz assumes a 2-cycle issue nobody measured; at the measured "
                                          "issue times the kernel's arithmetic alone takes issue_ms_per_launch_measured_rates of its time"},
                         "hbm": {"compulsory_bytes_per_launch": compulsory, "compulsory_GBps": compulsory / scan_s / 1e9 if scan_s > 0 else 0.0,
                                 "frac_of_hbm_peak": compulsory / scan_s / 1e9 / HBM_PEAK_GBS if scan_s > 0 else 0.0,
                                 "peak_GBps": HBM_PEAK_GBS,
                                 "survey_6vector_GBps": survey_gbps, "survey_bytes_per_eval": survey_bytes_per_eval},
                         "note": "bound = L2 -> CU bandwidth: achieved = insertion tests x one vector (S x Wp x 4 B, the bytes the kernel "
                                 "actually loads per test: chain in registers, sibling pairs share loads; read from the word-major copy, 16 B "
                                 "per lane and vector) / HIP-event kernel time; peak = the guide's aggregate L2 figure (its measured gathers "
                                 "from L2 reach 16.8-18.8 TB/s; this box: 124 GB/s per CU = 31.7 TB/s for 1-KB dwordx4 gathers from L2, "
                                 "tools/ubench/l1_rate).  valu.* = algorithmic "
                                 "lane-ops (2 S chain + 3 S join + 1 popcount + 3 reduction per test and 32-site word) against 256 CUs x 128 "
                                 "lanes x 2.4 GHz.  hbm.*: what must come from HBM per launch (every vector once) -- far from a limit; "
                                 "traffic = rocprofv3 PMC bytes that left the L2s per launch (2 x FETCH_SIZE + WRITE_SIZE, KB; see "
                                 "traffic_source; a committed figure is quoted only while the sources match the profiled build).  survey_6vector_GBps is SURVEY 8(d)'s 6-vectors-per-test figure / kernel time: a "
                                 "labelled side number, not a fraction of anything the kernel moves"},
            "views": {"newview_ops": st["newview_ops"] / args.steps, "kernel_ms_per_step": view_ms,
                      "schedule_kernel": {"kernel": "k_sched", "schedule_workgroup_us": sched_us[0], "descriptor_workgroup_us": sched_us[1],
                                          "dependency_levels": sched_us[2],
                                          "what": "one launch in front of the refresh of a NEW topology: refresh schedule (levels, ops in level "
                                                  "order) and the sweep's scan descriptors from the topology array, one workgroup each; durations "
                                                  "measured in the kernel (s_memrealtime)"},
                      "launches_per_step": st["view_launches"] / args.steps,
                      # refresh of every directional vector: 2 vector reads + 1 write per op, HBM-bound by nature
                      "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                   "achieved": (st["newview_ops"] / args.steps) * 3 * eng.S * eng.Wp * 4 / (view_ms * 1e-3) / 1e9 if view_ms > 0 else 0.0,
                                   "frac": (st["newview_ops"] / args.steps) * 3 * eng.S * eng.Wp * 4 / (view_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if view_ms > 0 else 0.0,
                                   "traffic": offline_traffic(args.workload, "k_newview") if (LIVE_TRAFFIC or not args.opt) else None}},
            "host_ms_per_step": {"plan": st["host_plan_ms_total"] / args.steps, "views": st["host_views_ms_total"] / args.steps,
                                 "scan": st["host_scan_ms_total"] / args.steps,
                                 "sweep_call": st["host_sweep_ms_total"] / args.steps},
        }
        core_res = res

    # The secondary legs below exchange data between the ranks.  Should one of them stall (a rank that failed skips a
    # collective the others wait in), the headline line still goes out: after --legs-timeout seconds rank 0 prints it
    # without the legs and every rank leaves.
    import threading

    def give_up():
        if rank == 0:
            out = dict(core_res)
            out["bootstrap_legs_error"] = "secondary legs did not finish within %d s" % args.legs_timeout
            print(json.dumps(out), flush=True)
        os._exit(3)                                # a stalled multi-rank leg is a failure, not a success

    watchdog = threading.Timer(args.legs_timeout, give_up)
    watchdog.daemon = True
    if args.ufboot_samples > 0 and world > 1:
        watchdog.start()

    # ---- second half of BASELINE.json's metric ("bootstrap wall-clock"): the -bb flow on this alignment.
    # (1) online phase: ONE search chain, sequential by nature: every rank makes the same pllOptimizeSprParsimony call from
    #     the same start tree with IQTree::saveCurrentTree's bookkeeping after every insertion test (candidate masks ->
    #     binary x int8 MFMA product -> event replay), but holds only every n_gpus-th bootstrap sample; the ranks
    #     all-gather their per-batch events (RCCL) and replay the merged list, so the chain is identical everywhere;
    # (2) refinement: IQTree::optimizeBootTrees -- every sample's tree from (1) re-weighted + one SPR climb, sample b on
    #     rank b % n_gpus (strong scaling: the total number of samples is fixed).
    ufb = None
    boot = None
    nondeg = None
    back_r = None
    legs_error = None
    samples = None
    n_eng = 1
    tb_nocache = None
    tb_persample = None
    n_distinct_trees = None
    if args.ufboot_samples > 0:
        samples = np.random.default_rng(4242).multinomial(P, np.ones(P) / P, size=args.ufboot_samples).astype(np.uint16)
    _lap("ufboot_online")
    if args.ufboot_samples > 0 and (leg_on("ufboot_online") or leg_on("random_start")):
        try:
            B = args.ufboot_samples
            from mpboot_amd import shard
            back_u = shard.broadcast_tree(back, 0, len(back))      # the chain starts from rank 0's tree on every rank
            # a sample-sharded online phase exchanges its events through the library's own RCCL communicator (mpf_rccl_exchange:
            # ncclAllGather from inside libmpfitch.so) where that is to be had; torch.distributed otherwise (gloo in the CPU tests)
            native_x = None
            if world > 1 and backend == "nccl" and shard.online_shard(B, rank, world, args.shard_online) is not None:
                try:
                    native_x = shard.native_comm(device)
                except Exception:
                    native_x = None
            tus = []
            for timed in (False, True, True):          # first pass: allocations, code load; then two timed passes, the faster one counts
                eng.ufboot_attach(samples, 0.5, shard=shard.online_shard(B, rank, world, args.shard_online), exchange=native_x)   # sharded: samples rank, rank + world, ... ; events all-gathered per batch
                eng.set_tree(back_u)
                eng.reset_node_order()
                eng.seed_ties(engine.TIE_RANDOM, 1)
                eng.reset_stats()
                barrier()
                tu0 = time.perf_counter()
                us = eng.optimize_spr(1, args.maxtrav)
                barrier()
                if timed:
                    tus.append(time.perf_counter() - tu0)
            tu = min(tus)
            ust, ucn = eng.stats(), eng.ufboot_counters()
            ufb = {"samples": B, "samples_local": len(range(rank, B, world)), "seconds": tu, "seconds_each_pass": tus, "score": us, "insertion_tests": ust["insertion_tests"], "moves": ust["moves_applied"],
                   "saved_trees": len(eng.ufboot_tree_logl()), **ucn}
            n_rep = min(B, args.bootstrap_replicates)
            if n_rep > 0:
                from mpboot_amd import bootstrap
                _logl, _cnt, bt = eng.ufboot_state()
                cache = {}
                boot_trees = []
                for b in range(n_rep):
                    t = int(bt[b])
                    if t not in cache:
                        cache[t] = eng.ufboot_tree(t)
                    boot_trees.append(cache[t])
                online_best = -_logl[:n_rep]
                n_distinct_trees = len(cache)
            eng.ufboot_detach()
            if n_rep > 0:
                # every engine is driven by its own host thread: not more of them than this rank's share of the usable cores
                try:
                    cores_here = len(os.sched_getaffinity(0))
                except AttributeError:
                    cores_here = os.cpu_count() or 1
                n_eng = max(1, min(args.engines_per_gpu, max(2, cores_here // max(1, world))))
                engines = [eng] + [engine.FitchEngine(codes, datatype=engine.DNA if alphabet == "DNA" else engine.AA, device=device)
                                   for _ in range(n_eng - 1)]
                for extra in engines[1:]:
                    for kv in args.opt:
                        k, v = kv.split("=")
                        extra.set_option(k, int(v))
                bootstrap.refine_boot_trees(engines, samples[:min(n_rep, 8 * world)], boot_trees[:min(n_rep, 8 * world)], 999, args.maxtrav)  # warm-up
                barrier()
                tb0 = time.perf_counter()
                bscores, _ = bootstrap.refine_boot_trees(engines, samples[:n_rep], boot_trees, 7, args.maxtrav)
                barrier()
                tb = time.perf_counter() - tb0
                boot = (n_rep, tb, float(np.mean(bscores)), float(np.mean(online_best)), bool((bscores <= online_best).all()))
                # the samples of this workload keep ONE tree (the start tree is SPR-optimal), so every refinement after an
                # engine's first re-uses that topology's plans; a run whose samples keep different trees plans each of them:
                # the same leg once more with the plan cache off
                for x in engines:
                    x.set_option("plan_cache", 0)
                barrier()
                tb0 = time.perf_counter()
                bscores_nc, _ = bootstrap.refine_boot_trees(engines, samples[:n_rep], boot_trees, 7, args.maxtrav)
                barrier()
                tb_nocache = time.perf_counter() - tb0
                for x in engines:
                    x.set_option("plan_cache", 1)
                assert (bscores_nc == bscores).all()
                # ... and as the reference's loop has it: one re-weighting + one climb per sample (what the batched first sweep replaces)
                barrier()
                tb0 = time.perf_counter()
                bscores_ps, _ = bootstrap.refine_boot_trees(engines, samples[:n_rep], boot_trees, 7, args.maxtrav, batched=False)
                barrier()
                tb_persample = time.perf_counter() - tb0
                assert (bscores_ps == bscores).all()
                eng.set_weights(np.ones(P, dtype=np.int32))
            # ---- the same flow from a start tree that is NOT a local optimum (the RAS tree of this alignment already is one:
            # zero moves above): a random topology, thousands of accepted moves, refinements that really climb
            _lap("random_start")
            if args.random_start_leg and leg_on("random_start"):
                back_r = shard.broadcast_tree(trees.random_topology(n, np.random.default_rng(2024)), 0, len(back))
                eng.set_tree(back_r)
                eng.reset_node_order()
                eng.seed_ties(engine.TIE_RANDOM, 1)
                eng.reset_stats()
                barrier()
                t0r = time.perf_counter()
                s_plain = eng.optimize_spr(1, args.maxtrav)
                barrier()
                t_plain = time.perf_counter() - t0r
                st_plain = eng.stats()
                # (the same climb with the sweep loop on the host: one launch chain + synchronisation per accepted move)
                eng.set_option("climb_device", 0)
                eng.set_tree(back_r)
                eng.reset_node_order()
                eng.seed_ties(engine.TIE_RANDOM, 1)
                barrier()
                t0r = time.perf_counter()
                s_plain_h = eng.optimize_spr(1, args.maxtrav)
                barrier()
                t_plain_h = time.perf_counter() - t0r
                eng.set_option("climb_device", 1)
                assert s_plain_h == s_plain
                # ... and with every sweep in the persistent kernel (climb_device 2): k_climb's own data rate and where a step's time goes
                eng.set_option("climb_device", 2)
                eng.set_tree(back_r)
                eng.reset_node_order()
                eng.seed_ties(engine.TIE_RANDOM, 1)
                eng.reset_stats()
                ph0 = np.array([eng.get_option("climb_phase_us%d" % k_) for k_ in range(7)], dtype=np.float64)
                barrier()
                t0r = time.perf_counter()
                s_plain_k = eng.optimize_spr(1, args.maxtrav)
                barrier()
                t_plain_k = time.perf_counter() - t0r
                ph1 = np.array([eng.get_option("climb_phase_us%d" % k_) for k_ in range(7)], dtype=np.float64)
                st_k = eng.stats()
                eng.set_option("climb_device", 1)
                assert s_plain_k == s_plain
                vec_b = int(eng.S) * int(eng.Wp) * 4
                k_ms = st_k["climb_ms_total"]
                k_ach = st_k["insertion_tests"] * vec_b / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
                climb_roof = {"bound": "l2", "achieved": k_ach, "peak": 34500.0, "unit": "GB/s", "frac": k_ach / 34500.0, "traffic": None,
                              "kernel": "k_climb", "launches": st_k["climb_launches"], "steps": st_k["climb_steps"], "kernel_ms": k_ms,
                              "seconds_all_in_kernel": t_plain_k, "insertion_tests": st_k["insertion_tests"], "bytes_per_test": vec_b,
                              "us_per_step": k_ms * 1e3 / max(1, st_k["climb_steps"]),
                              "phase_us_per_step": dict(zip(["filter", "enumerate_and_walk", "closure", "refresh", "scan", "exchange", "decide"],
                                                            ((ph1 - ph0) / max(1, st_k["climb_steps"])).round(2).tolist())),
                              "note": "the whole climb in the persistent kernel: insertion tests x one vector (the bytes a test needs, as the headline "
                                      "prices k_scan_prog) / the kernel's own time, against the aggregate L2 -> CU rate.  A step is a chain of "
                                      "dependent phases on ~100 candidates per workgroup: the kernel is bound by their latencies, not by this rate"}
                nondeg = {"start": "random topology (numpy default_rng(2024))", "start_score": eng.score_tree(back_r),
                          "plain_climb": {"seconds": t_plain, "score": s_plain, "moves": st_plain["moves_applied"],
                                          "insertion_tests": st_plain["insertion_tests"], "scan_launches": st_plain["scan_launches"],
                                          "tests_per_s": st_plain["insertion_tests"] / t_plain,
                                          "climb_kernel": {"launches": st_plain["climb_launches"], "steps": st_plain["climb_steps"],
                                                           "prune_nodes": st_plain["climb_nodes"], "moves": st_plain["climb_moves"],
                                                           "ms": st_plain["climb_ms_total"]},
                                          "seconds_host_driven_batches": t_plain_h, "roofline": climb_roof,
                                          "what": "pllOptimizeSprParsimony from a random tree: sweeps with dense moves run in the persistent kernel "
                                                  "k_climb (device-resident loop), sparse ones as whole-chip host-driven batches"}}
                eng.ufboot_attach(samples, 0.5, shard=shard.online_shard(B, rank, world, args.shard_online), exchange=native_x)
                eng.set_tree(back_r)
                eng.reset_node_order()
                eng.seed_ties(engine.TIE_RANDOM, 1)
                eng.set_option("timing", 0)            # (HIP events around the product would cost every batch a synchronisation)
                eng.reset_stats()
                barrier()
                t0r = time.perf_counter()
                s_bb = eng.optimize_spr(1, args.maxtrav)
                barrier()
                t_bb = time.perf_counter() - t0r
                eng.set_option("timing", 1)
                st_bb, cn_bb = eng.stats(), eng.ufboot_counters()
                _l2, _c2, bt2 = eng.ufboot_state()
                cache2, trees2 = {}, []
                for b in range(B):
                    t = int(bt2[b])
                    if t not in cache2:
                        cache2[t] = eng.ufboot_tree(t)
                    trees2.append(cache2[t])
                online2 = -_l2
                # a LATER search iteration, as a run repeats it hundreds of times: cut-off from the saved trees (top 10 %, iqtree.cpp:1662-1676),
                # the best tree perturbed by 30 random SPR moves, one more climb with the bookkeeping (measured behind the captured state:
                # the refinement below does not see it).  Every rank perturbs with the same generator.
                it_leg = None
                try:
                    final_bb = eng.get_tree()
                    cut_bb = eng.ufboot_next_cutoff(10)
                    eng.ufboot_set_cutoff(cut_bb)
                    pert = trees.random_spr_moves(eng, final_bb, np.random.default_rng(77), 30)
                    eng.set_tree(pert)
                    eng.reset_node_order()
                    s_pert = eng.score_tree()
                    eng.set_option("timing", 0)
                    eng.reset_stats()
                    barrier()
                    t0i = time.perf_counter()
                    s_it = eng.optimize_spr(1, args.maxtrav)
                    barrier()
                    t_it = time.perf_counter() - t0i
                    eng.set_option("timing", 1)
                    st_it = eng.stats()
                    it_leg = {"seconds": t_it, "cutoff_length": -cut_bb, "perturbed_score": s_pert, "score": s_it, "moves": st_it["moves_applied"],
                              "insertion_tests": st_it["insertion_tests"],
                              "what": "one later search iteration of the same run: logl_cutoff = top 10 % of the saved trees, best tree perturbed by 30 "
                                      "random SPR moves, SPR climb with saveCurrentTree bookkeeping under the cut-off: the stretch in which nothing "
                                      "reaches the bookkeeping runs as the plain climb (k_climb with a hand-back condition, cost-only batches), the "
                                      "rest on the tracker's two-wait loop (product compacted on the host); the bb_reference_run leg times the reference's own iteration flow"}
                except Exception as exc:
                    it_leg = {"error": repr(exc)}
                eng.ufboot_detach()
                barrier()
                t0r = time.perf_counter()
                if n_rep > 0:
                    bs2, _ = bootstrap.refine_boot_trees(engines, samples[:n_rep], trees2[:n_rep], 11, args.maxtrav)
                barrier()
                t_ref2 = time.perf_counter() - t0r
                eng.set_weights(np.ones(P, dtype=np.int32))
                nondeg["bb_flow"] = {"online_phase_s": t_bb, "score": s_bb, "moves": st_bb["moves_applied"], "insertion_tests": st_bb["insertion_tests"],
                                     "tests_per_s": st_bb["insertion_tests"] / t_bb, "events": cn_bb["events"], "tie_draws": cn_bb["tie_draws"],
                                     "reps_kernel_ms": cn_bb["reps_kernel_ms"] or None, "refined_samples": n_rep, "refinement_s": t_ref2,
                                     "seconds": t_bb + t_ref2, "later_iteration": it_leg,
                                     "mean_sample_score_online": float(np.mean(online2[:n_rep])) if n_rep else None,
                                     "mean_sample_score_refined": float(np.mean(bs2)) if n_rep else None,
                                     "samples_improved_by_refinement": int((bs2 < online2[:n_rep]).sum()) if n_rep else None}
            eng.set_tree(back)
        except Exception as exc:        # the headline metric above must survive a failing secondary leg
            legs_error = repr(exc)
            ufb = boot = nondeg = None

    # ---- further legs: whole climbs, not sweeps
    import threading as _th
    mk = lambda: engine.FitchEngine(codes, datatype=engine.DNA if alphabet == "DNA" else engine.AA, device=device)
    conc = None
    bbref = None
    noisy_leg = None
    many_leg = None
    c5f_sweep = c5f_climb = None
    c2leg = None
    c5leg = None
    startup = None
    try:
        pool = [eng]
        def grow(k):
            while len(pool) < k:
                x = mk()
                for kv in args.opt:
                    kk, vv = kv.split("=")
                    x.set_option(kk, int(vv))
                pool.append(x)
        def run_threads(k, fn):
            out = [None] * k
            def work(i):
                out[i] = fn(i, pool[i])
            th = [_th.Thread(target=work, args=(i,)) for i in range(k)]
            barrier()
            t0_ = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            torch.cuda.synchronize()
            return time.perf_counter() - t0_, out
        _lap("concurrent_climbs")
        if args.climb_engines > 0 and args.random_start_leg and leg_on("concurrent_climbs"):
            # independent climbs from random trees, one engine per host thread (the shape of the start trees / the refinements of
            # a run): every climb is a chain of dependent steps, concurrent engines fill the chip
            E = args.climb_engines
            grow(E)
            def climb_fn(mode, tile):
                def f(i, x):
                    x.set_option("climb_device", mode)
                    x.set_option("climb_tile", tile)
                    x.set_tree(trees.random_topology(n, np.random.default_rng(7000 + 100 * rank + i)))
                    x.reset_node_order()
                    x.seed_ties(engine.TIE_RANDOM, 1 + i)
                    return x.optimize_spr(1, args.maxtrav)
                return f
            run_threads(E, climb_fn(2, args.climb_tile))                       # buffers, code
            t_dev, sc_dev = run_threads(E, climb_fn(2, args.climb_tile))
            t_host, sc_host = run_threads(E, climb_fn(0, 1))
            for x in pool:
                x.set_option("climb_device", 1)
                x.set_option("climb_tile", 1)
            assert sc_dev == sc_host
            conc = {"engines": E, "climbs_per_s": E / t_dev, "seconds_per_round": t_dev, "scores": [int(min(sc_dev)), int(max(sc_dev))],
                    "host_driven_batches": {"climbs_per_s": E / t_host, "seconds_per_round": t_host},
                    "what": "%d independent SPR climbs (radius %d) from random topologies side by side on one GPU, one engine per host "
                            "thread; each climb runs in the persistent kernel k_climb (%d-word tiles: %d workgroups per climb); "
                            "host_driven_batches = the same climbs with the loop on the host (engine option climb_device = 0).  "
                            "GPU_MAX_HW_QUEUES=%s" % (E, args.maxtrav, 16 * args.climb_tile, (eng.Wp + 16 * args.climb_tile - 1) // (16 * args.climb_tile), os.environ.get("GPU_MAX_HW_QUEUES"))}
        _lap("start_trees")
        if args.start_trees > 0 and (leg_on("start_trees") or leg_on("bb_reference_run")):
            # the start-up phase of a run: numpars randomized-stepwise-addition trees, each SPR-optimised (phyloanalysis.cpp:1270-1317,
            # tools.cpp:767); unit u on rank u % n_gpus, several engines per GPU
            k_e = max(1, args.start_engines)
            grow(k_e)
            units = [u for u in range(args.start_trees) if u % world == rank]
            unit_q, unit_lock = iter(units), _th.Lock()          # (the engines take the next tree when they are done with theirs)
            start_trees_built = []
            def ras_fn(i, x):
                best = None
                while True:
                    with unit_lock:
                        u = next(unit_q, None)
                    if u is None:
                        break
                    sd = shard.unit_seed(31337, u)
                    x.seed_ties(engine.TIE_RANDOM, sd)
                    x.reset_node_order()
                    sc = x.make_parsimony_tree(sd, args.maxtrav)
                    with unit_lock:
                        start_trees_built.append((u, x.get_tree(), int(sc)))
                    best = sc if best is None else min(best, sc)
                return best
            def ras_warm(i, x):                     # (an engine's first tree allocates the kernel's scratch: not a step of the phase)
                x.seed_ties(engine.TIE_RANDOM, 1)
                x.make_parsimony_tree(1, args.maxtrav)
                return None
            run_threads(k_e, ras_warm)
            g0 = sum(x.get_option("grow_launches") for x in pool[:k_e])
            barrier()
            t_ras, bests = run_threads(k_e, ras_fn)
            barrier()
            g1 = sum(x.get_option("grow_launches") for x in pool[:k_e])
            bb = [b for b in bests if b is not None]
            startup = {"trees": args.start_trees, "seconds": t_ras, "engines_per_gpu": k_e, "best_score_rank0": int(min(bb)) if bb else None,
                       "seconds_per_tree_per_engine": t_ras * k_e / max(1, len(units)),
                       "trees_grown_in_k_grow": int(g1 - g0),
                       "what": "%d randomized stepwise-addition trees + SPR climb (radius %d) each, as the reference's start-up builds them "
                               "(phyloanalysis.cpp:1270-1317); tree u on rank u %% n_gpus.  The addition loop of a tree is ONE persistent kernel "
                               "launch (k_grow: the rooted tree in LDS, one vector load per branch and added taxon, the insertions replayed on "
                               "the host's mirror); trees_grown_in_k_grow counts the launches that came back clean" % (args.start_trees, args.maxtrav)}
        _lap("bb_reference_run")
        if startup is not None and samples is not None and (args.bb_iterations > 0 or args.bb_rounds > 0) and leg_on("bb_reference_run"):
            # ---- BASELINE config 4: -bb 1000 as the reference runs it (benchlegs/bb_run.py).  The start trees of all ranks are the
            # candidate set of every chain.
            from benchlegs import bb_run as _bbleg
            mine = sorted(start_trees_built)
            if world > 1:
                box = [None] * world
                dist.all_gather_object(box, [(u, t.tobytes(), sc) for u, t, sc in mine])
                mine = sorted((u, np.frombuffer(t, dtype=np.int32), sc) for part in box for u, t, sc in part)
            starts_bb = [(t, sc) for _u, t, sc in mine]
            # (chains are host threads: on a node that grants this job few cores -- the single-GPU boxes give 16 -- the ranks share them)
            bb_workers = max(1, min(args.bb_workers, max(4, cpu_quota() // max(1, world))))
            grow(bb_workers)
            try:
                bbref = _bbleg.run(pool, samples, starts_bb, args.maxtrav, rank, world, barrier, args.bb_iterations, bb_workers, args.bb_rounds,
                                   args.bb_sync, startup["seconds"], pool[:max(1, min(len(pool), args.engines_per_gpu))])
            except Exception as exc:
                bbref = {"error": repr(exc)}
            eng.set_option("timing", 1)
            eng.set_weights(np.ones(P, dtype=np.int32))
        _lap("climbs_in_one_launch")
        if world == 1 and args.workload == "C3" and leg_on("climbs_in_one_launch"):
            try:
                from benchlegs import climbs_many as _cm
                many_leg = _cm.run(device, args.maxtrav, barrier, args.many_c2, args.many_c3)
            except Exception as exc:
                many_leg = {"error": repr(exc)}
        _lap("c5_fitch")
        if world == 1 and args.workload == "C3" and leg_on("c5_fitch"):
            try:
                from benchlegs import c5_fitch as _c5f
                c5f_sweep, c5f_climb = _c5f.run(device, args.maxtrav, 20, 5, barrier, None if args.no_cpu else cpu_baseline,
                                                None if args.no_cpu else climb_cpu_baseline, min(args.cpu_budget, 10.0))
            except Exception as exc:
                c5f_sweep, c5f_climb = {"error": repr(exc)}, None
        _lap("noisy_bootstrap")
        if world == 1 and args.workload == "C3" and leg_on("noisy_bootstrap"):
            try:
                from benchlegs import noisy as _noisy
                noisy_leg = _noisy.run(device, args.maxtrav, 12, 40, args.engines_per_gpu, barrier, B=max(1, args.ufboot_samples))
            except Exception as exc:
                noisy_leg = {"error": repr(exc)}
        _lap("c2_climb")
        if world == 1 and args.workload == "C3" and args.random_start_leg and leg_on("c2_climb"):
            # BASELINE config 2 (200 taxa x 10 000 patterns): a full SPR hill climb from a random tree
            letters2, names2 = synth.workload("C2")
            codes2 = synth.letters_to_codes(letters2, "DNA")
            e2 = engine.FitchEngine(codes2, datatype=engine.DNA, device=device)
            back2 = trees.random_topology(codes2.shape[0], np.random.default_rng(1))
            tt = []
            for _ in range(3):
                e2.set_tree(back2)
                e2.reset_node_order()
                e2.seed_ties(engine.TIE_RANDOM, 1)
                e2.reset_stats()
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                s2 = e2.optimize_spr(1, args.maxtrav)
                tt.append(time.perf_counter() - t0_)
            st2 = e2.stats()
            c2leg = {"workload": "C2: 200 taxa x 10000 DNA patterns, full SPR hill climb from a random topology (numpy default_rng(1)), radius %d" % args.maxtrav,
                     "seconds": min(tt[1:]), "seconds_each_pass": tt, "score": s2, "moves": st2["moves_applied"], "insertion_tests": st2["insertion_tests"],
                     "climb_kernel_launches": st2["climb_launches"], "climb_kernel_steps": st2["climb_steps"]}
            if not args.no_cpu:
                c2leg["cpu_baseline"] = climb_cpu_baseline(names2, letters2, "DNA", back2, args.maxtrav)
                if c2leg["cpu_baseline"]:
                    c2leg["gpu_over_cpu"] = c2leg["cpu_baseline"]["seconds"] / c2leg["seconds"]
            del e2
            # ... and many such climbs side by side (independent start trees: what the start-up phase and the bootstrap
            # refinement of a small alignment look like): one engine per host thread, 64-word tiles = 5 workgroups per climb
            try:
                ncore = len(os.sched_getaffinity(0))
            except AttributeError:
                ncore = os.cpu_count() or 1
            k2 = max(1, min(args.c2_engines, ncore))
            es = []
            for _ in range(k2):
                x = engine.FitchEngine(codes2, datatype=engine.DNA, device=device)
                x.set_option("climb_tile", 4)
                es.append(x)
            per = 6
            starts2 = [[trees.random_topology(codes2.shape[0], np.random.default_rng(7000 + 97 * i + j)) for j in range(per)] for i in range(k2)]
            sc2 = [None] * k2

            def c2_work(i):
                x, out = es[i], []
                for j in range(per):
                    x.set_tree(starts2[i][j])
                    x.reset_node_order()
                    x.seed_ties(engine.TIE_RANDOM, 100 + i * per + j)
                    out.append(x.optimize_spr(1, args.maxtrav))
                sc2[i] = out

            for rep in range(2):                          # (the first round allocates the climb buffers)
                th2 = [_th.Thread(target=c2_work, args=(i,)) for i in range(k2)]
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for t in th2:
                    t.start()
                for t in th2:
                    t.join()
                t_c2 = time.perf_counter() - t0_
            cb2 = c2leg.get("cpu_baseline")
            c2leg["concurrent"] = {"engines": k2, "climbs": k2 * per, "seconds": t_c2, "climbs_per_s": k2 * per / t_c2,
                                   "climb_launches_engine0": es[0].stats()["climb_launches"],
                                   "reference_climbs_per_s_one_core": (1.0 / cb2["seconds"]) if cb2 else None,
                                   "what": "%d engines on host threads, %d climbs each from different random topologies, k_climb with 64-word tiles "
                                           "(5 workgroups per climb)" % (k2, per)}
            del es
        _lap("c5_weighted_sweep")
        if world == 1 and args.workload == "C3" and args.weighted_leg and leg_on("c5_weighted_sweep"):
            # BASELINE config 5 in its `-cost` form: the weighted (Sankoff) engine on 500 taxa x 20 000 protein patterns, 20 states
            letters5, names5 = synth.workload("C5")
            codes5 = synth.letters_to_codes(letters5, "AA")
            n5, P5 = codes5.shape
            cm = np.random.default_rng(5).integers(1, 6, size=(20, 20))
            cost5 = (np.triu(cm, 1) + np.triu(cm, 1).T).astype(np.uint32)
            f5 = engine.FitchEngine(codes5, datatype=engine.AA, device=device)
            f5.seed_ties(engine.TIE_RANDOM, 1)
            f5.make_parsimony_tree(12345, 0)
            back5 = f5.get_tree()
            del f5
            e5 = engine.FitchEngine(codes5, datatype=engine.AA, device=device, cost=cost5)
            e5.set_option("timing", 2)
            e5.set_tree(back5)
            s5 = e5.score_tree()
            e5.sweep_scan(1, args.maxtrav)
            e5.reset_stats()
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            k5 = 0
            for _ in range(5):
                e5.set_tree(back5)
                k5, _b5 = e5.sweep_scan(1, args.maxtrav)
            torch.cuda.synchronize()
            t5 = (time.perf_counter() - t0_) / 5
            st5 = e5.stats()
            scan5 = st5["scan_kernel_ms_total"] / max(1, st5["scan_launches"])
            # algorithmic packed-u16 operations per insertion test and pattern: one min-plus transform of the running up-vector
            # (2 S^2 add/min; every expansion makes two tests from two transforms) + the 3 S-operation test
            ops5 = (2 * 20 * 20 + 3 * 20) * float(e5.num_informative) * k5
            ach5 = ops5 / (scan5 * 1e-3) / 1e12 if scan5 > 0 else 0.0
            c5leg = {"workload": "C5 weighted: %d taxa x %d protein patterns, 20 states, symmetric random costs 1..5 (numpy default_rng(5)), one full "
                                 "sweep scan per step from the RAS tree, radius %d" % (n5, P5, args.maxtrav),
                     "ms_per_step": t5 * 1e3, "evals_per_step": k5, "evals_per_s": k5 / t5, "tree_length": s5,
                     "scan_kernel_ms": scan5, "view_kernel_ms_per_step": st5["view_kernel_ms_total"] / 5,
                     "roofline": {"bound": "valu", "achieved": ach5, "peak": 2.0 * VALU_PEAK_TOPS, "unit": "T packed-u16 op/s",
                                  "frac": ach5 / (2.0 * VALU_PEAK_TOPS), "kernel": "k_snk_scan",
                                  # v_pk_add_u16 / v_pk_min_u16 issue once per 1.76 ns per SIMD (tools/ubench/valu_rate, profiles/r3/valu_rate.txt)
                                  "measured_issue_ceiling": 1024 * 128 / 1.76e-9 / 1e12,
                                  "frac_of_measured_issue_ceiling": ach5 / (1024 * 128 / 1.76e-9 / 1e12),
                                  # the kernel's other bound (profiles/r6/snk_scan_pmc.txt: FETCH_SIZE x 2 per launch of this sweep,
                                  # SQ_INSTS_VALU per launch; counters of the committed build, not of this run)
                                  "hbm": {"traffic_per_launch_bytes": 73.9e9, "achieved": 73.9 / scan5 if scan5 > 0 else None, "peak": 8000.0,
                                          "unit": "GB/s", "frac": 73.9 / scan5 / 8000.0 if scan5 > 0 else None,
                                          "source": "profiles/r6/snk_scan_pmc.txt (rocprofv3 --pmc FETCH_SIZE, gfx950 correction x 2)"},
                                  "valu_issue": {"wave_instructions_per_launch": 9.37e9, "ns_per_instruction_and_simd_two_waves": 1.95,
                                                 "issue_ms": 9.37e9 * 1.95e-9 / 1024 * 1e3,
                                                 "frac_of_kernel_time": 9.37e9 * 1.95e-9 / 1024 * 1e3 / scan5 if scan5 > 0 else None},
                                  "note": "min-plus arithmetic on packed 16-bit costs (v_pk_add_u16 / v_pk_min_u16: two patterns per lane); "
                                          "achieved = insertion tests x patterns x (2 S^2 + 3 S) / HIP-event time of the scan kernel; peak = "
                                          "256 CUs x 4 SIMD-32 x 2.4 GHz x 2 values per lane.  The kernel also moves 74 GB per launch from HBM "
                                          "(two 800 KB vectors per test, nothing re-used within an L2's reach): hbm.*; valu_issue = what its "
                                          "9.4e9 wave-instructions cost at the rate two resident waves per SIMD reach"}}
            if not args.no_cpu:
                from oracle import pyoracle as po5
                o5 = po5.Oracle(codes5, datatype=po5.AA, cost=cost5)
                assert o5.score_tree(back5) == s5
                o5.seed_ties(po5.TIE_RANDOM, 1)
                o5.set_best(s5)
                nodep5 = o5.nodep()
                tc0 = time.perf_counter()
                kk0, i5 = o5.counters()[2], 1
                while time.perf_counter() - tc0 < min(args.cpu_budget, 8.0) and i5 <= 2 * n5 - 2:
                    o5.rearrange(int(nodep5[i5]), 1, args.maxtrav)
                    i5 += 1
                tc = time.perf_counter() - tc0
                kc5 = o5.counters()[2] - kk0
                c5leg["cpu_baseline"] = {"value": kc5 / tc, "unit": "insertion tests/s", "cores": 1, "kind": "port",
                                         "sample": "%d prune nodes (%d insertion tests) of the same sweep on the scalar C oracle (exact 32-bit; the "
                                                   "reference's Sankoff kernels live in the unbuildable C++ layer)" % (i5 - 1, kc5)}
                c5leg["gpu_over_cpu"] = (k5 / t5) / (kc5 / tc)
            del e5
    except Exception as exc:
        legs_error = (legs_error or "") + " | climb legs: " + repr(exc)

    watchdog.cancel()
    if rank == 0:
        res = core_res
        if conc is not None:
            res["concurrent_climbs"] = conc
        if startup is not None:
            res["start_trees"] = startup
            if not args.no_cpu and world == 1 and not profiled:
                startup["cpu_baseline"] = start_trees_cpu_baseline(names, letters, alphabet, args.maxtrav, startup["trees"])
                cb = startup["cpu_baseline"]
                if cb:
                    startup["gpu_over_cpu_one_core"] = cb["seconds_for_%d_trees_one_core" % startup["trees"]] / startup["seconds"]
                    startup["gpu_over_cpu_all_cores"] = cb["all_cores"]["seconds_for_%d_trees" % startup["trees"]] / startup["seconds"]
        if bbref is not None:
            res["bb_reference_run"] = bbref
        if noisy_leg is not None:
            res["noisy_bootstrap"] = noisy_leg
        if many_leg is not None:
            res["climbs_in_one_launch"] = many_leg
            try:
                cb2 = (res.get("c2_climb") or {}).get("cpu_baseline")
                if cb2:
                    many_leg["c2"]["reference_climbs_per_s_one_core"] = 1.0 / cb2["seconds"]
                cb3 = ((res.get("random_start") or {}).get("plain_climb") or {}).get("cpu_baseline")
                if cb3 and "c3" in many_leg:
                    many_leg["c3"]["reference_climbs_per_s_one_core"] = 1.0 / cb3["seconds"]
            except Exception:
                pass
        if c5f_sweep is not None:
            res["c5_fitch_sweep"] = c5f_sweep
        if c5f_climb is not None:
            res["c5_fitch_climb"] = c5f_climb
        if c2leg is not None:
            res["c2_climb"] = c2leg
        if c5leg is not None:
            res["c5_weighted_sweep"] = c5leg
        if legs_error is not None:
            res["bootstrap_legs_error"] = legs_error
        if boot is not None:
            res["bootstrap_wall_clock"] = {
                "samples": ufb["samples"], "online_phase_s": ufb["seconds"], "refined_samples": boot[0], "refinement_s": boot[1],
                "seconds_from_ras_tree": ufb["seconds"] + boot[1] * ufb["samples"] / boot[0],
                # the flow that really searches: from a random topology (thousands of accepted moves online, refinements that climb);
                # the RAS tree of this alignment is SPR-optimal already (zero moves: seconds_from_ras_tree is 1000 move-less sweeps)
                # the run the reference performs (bb_reference_run leg: start trees + doTreeSearch iterations + refinement); where that leg
                # did not run, the older one-climb flow from a random tree
                "seconds": (bbref["seconds_measured_one_chain"] if bbref and "seconds_measured_one_chain" in bbref
                            else nondeg["bb_flow"]["seconds"] if nondeg is not None and "bb_flow" in nondeg
                            else ufb["seconds"] + boot[1] * ufb["samples"] / boot[0]),
                "seconds_is": ("bb_reference_run: %d start trees + %d doTreeSearch iterations of ONE chain (random NNIs / ratchet alternating) + refinement, "
                               "as measured; the stop rule needs >= %d iterations: see seconds_extrapolated_to_stop_rule"
                               % (bbref["start_trees"], bbref["sequential"]["iterations"], bbref["stop_rule_unsuccessful_iterations"])
                               if bbref and "seconds_measured_one_chain" in bbref
                               else "random-start flow, ONE climb + refinement (random_start.bb_flow)" if nondeg is not None and "bb_flow" in nondeg
                               else "flow from the RAS tree"),
                "seconds_extrapolated_to_stop_rule": bbref.get("seconds_extrapolated_to_stop_rule_one_chain") if bbref else None,
                "seconds_extrapolated_to_stop_rule_iteration_parallel": bbref.get("seconds_extrapolated_to_stop_rule_parallel") if bbref else None,
                "refinement_s_plan_cache_off": tb_nocache, "refinement_s_per_sample_climbs": tb_persample, "distinct_boot_trees": n_distinct_trees,
                "scaling": "strong", "engines_per_gpu": n_eng,
                "online_phase_sharded": shard.online_shard(ufb["samples"], rank, world, args.shard_online) is not None,
                "mean_sample_score_online": boot[3], "mean_sample_score_refined": boot[2], "refinement_never_worse": boot[4],
                "what": "-bb %d on this alignment from one start tree: online phase (one SPR climb with saveCurrentTree bookkeeping, "
                        "samples sharded over the ranks from shard.ONLINE_SHARD_MIN_SAMPLES samples on, every rank with all samples below) + refinement of every sample's tree (one SPR climb under the sample's weights, radius %d; "
                        "sample b on rank b %% n_gpus).  Refinement: the first sweep of all climbs that start from one topology is computed at "
                        "once (mpf_ufboot_refine_sweep: masked scan + mask x weight product on the matrix cores + per-sample replay of the tie "
                        "rules; the samples' weights are uploaded and laid out inside the timed region); samples whose sweep accepts a move "
                        "climb alone (re-weight, re-pack, SPR climb; several engines per GPU).  refinement_s_per_sample_climbs = the "
                        "reference's loop, one re-weighting + one climb per sample; seconds = online + refinement scaled to all samples.  "
                        "distinct_boot_trees = topologies the samples kept; refinement_s_plan_cache_off = the same leg planning every tree from scratch"
                        % (ufb["samples"], args.maxtrav)}
        if boot is not None and not args.no_cpu and world == 1:
            res["bootstrap_wall_clock"]["cpu_baseline"] = refine_cpu_baseline(codes, names, letters, alphabet, samples, boot_trees, bscores,
                                                                             args.maxtrav)
        if nondeg is not None:
            res["random_start"] = nondeg
            if not args.no_cpu and world == 1:
                nondeg["plain_climb"]["cpu_baseline"] = climb_cpu_baseline(names, letters, alphabet, back_r, args.maxtrav)
                # (spr_shim_driver is a GPU program: not started from a process a profiler has handed the GPU to)
                if not profiled:
                    nondeg["plain_climb"]["through_reference_binding"] = shim_climb_leg(eng, names, letters, alphabet, back_r, args.maxtrav)
        if ufb is not None:
            algo_ops = 2.0 * ufb["insertion_tests"] * (eng.W * 32) * ufb["samples_local"]
            kms = ufb["reps_kernel_ms"]
            top = algo_ops / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
            res["ufboot_online"] = {
                "what": "one pllOptimizeSprParsimony call (radius %d) from rank 0's start tree with online UFBoot-MP bookkeeping "
                        "for %d bootstrap samples: cut-off filter, REPS of every insertion test, per-sample update rule with the "
                        "reference's tie draws.  The search chain is sequential: every rank runs this same call on its share of the "
                        "samples (%d on rank 0) and the events are all-gathered per scan batch" % (args.maxtrav, ufb["samples"], ufb["samples_local"]),
                "seconds": ufb["seconds"], "seconds_each_pass": ufb["seconds_each_pass"], "insertion_tests": ufb["insertion_tests"], "moves": ufb["moves"],
                "tests_per_s": ufb["insertion_tests"] / ufb["seconds"], "saved_trees": ufb["saved_trees"],
                "events": ufb["events"], "tie_draws": ufb["tie_draws"], "score": ufb["score"],
                "roofline": {"bound": "mfma", "achieved": top, "peak": I8_PEAK_TOPS, "unit": "TOP/s", "frac": top / I8_PEAK_TOPS,
                             "kernel": "k_bitgemm", "kernel_ms_total": kms, "rows": ufb["reps_rows"],
                             "note": "achieved = 2 x insertion tests x sites x samples (the algorithmic REPS work; padding rows, "
                                     "home-edge rows and padding samples not counted) / HIP-event time of the product kernels; "
                                     "peak = dense int8 MFMA (2 x the 2.5 PFLOP/s bf16 figure)"}}
            if not args.no_cpu and world == 1:
                from oracle import pyoracle as po
                o = po.Oracle(codes, datatype=po.DNA if alphabet == "DNA" else po.AA)
                o.seed_ties(po.TIE_RANDOM, 1)
                o.ufboot_attach(np.random.default_rng(4242).multinomial(P, np.ones(P) / P, size=ufb["samples"]).astype(np.uint16))
                o.set_best(o.score_tree(back))
                nodep = o.nodep()
                tc0 = time.perf_counter()
                k0, i = o.counters()[2], 1
                while time.perf_counter() - tc0 < min(args.cpu_budget, 10.0) and i <= 2 * n - 2:
                    o.rearrange(int(nodep[i]), 1, args.maxtrav)
                    i += 1
                tc = time.perf_counter() - tc0
                kc = o.counters()[2] - k0
                res["ufboot_online"]["cpu_baseline"] = {
                    "value": kc / tc, "unit": "insertion tests/s (with REPS for all samples)", "cores": 1, "kind": "port",
                    "sample": "%d prune nodes (%d insertion tests) of the same sweep on the C oracle, AVX2 REPS loop" % (i - 1, kc)}
                res["ufboot_online"]["gpu_over_cpu"] = res["ufboot_online"]["tests_per_s"] / (kc / tc)
        if not args.no_cpu and world == 1:
            res["cpu_baseline"] = cpu_baseline(codes, back, names, letters, alphabet, args.maxtrav, args.cpu_budget)
            res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
            if "all_cores" in res["cpu_baseline"]:
                res["gpu_over_cpu_all_cores"] = res["value"] / res["cpu_baseline"]["all_cores"]["value"]
            if bbref and "sequential" in bbref:
                try:
                    sq = bbref["sequential"]
                    run_counts = {"trees_booked": sq["trees_booked"], "insertion_tests": sq["insertion_tests"]}
                    cbr = bb_run_cpu_baseline(codes, alphabet, samples, back, args.maxtrav, min(args.cpu_budget, 10.0),
                                              res["cpu_baseline"].get("evals_per_s"), run_counts, res.get("bootstrap_wall_clock", {}).get("cpu_baseline"))
                    if cbr:
                        # (the timed iterations only: start trees have a reference baseline of their own in the start_trees leg)
                        cbr["sample"] += "; covers the %d timed iterations + the refinement, not the start trees" % sq["iterations"]
                        bbref["cpu_baseline"] = cbr
                        bbref["gpu_over_cpu_one_core"] = cbr["value"] / (sq["iterations_s"] + (bbref.get("parallel") or {}).get("refinement_s", 0.0))
                except Exception as exc:
                    bbref["cpu_baseline"] = {"error": repr(exc)}
        _lap("end")
        agg = {}
        for k, v in _laps:
            agg[k] = agg.get(k, 0.0) + v
        res["bench_wall_s"] = {k: round(v, 2) for k, v in agg.items()}
        res["bench_wall_s"]["total"] = round(sum(agg.values()), 2)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if legs_error is not None:
        sys.exit(3)                                # headline printed, but a secondary leg failed


if __name__ == "__main__":
    sys.exit(main())

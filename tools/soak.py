#!/usr/bin/env python3
"""Randomised soak of this round's device paths against the host-driven paths of the same engine (both pinned against the oracle
by the test-suite on fixed cases):
  * k_climb (every tile width, both tie rules, radii 1..6, batch sizes) == host-driven batches: moves, tree, tie-stream state
  * mpf_ufboot_refine_sweep == mpf_set_weights + mpf_optimize_spr per sample: stable <=> no move, scores
  * k_grow (one launch per start tree, every tile shape) == the host-driven addition loop: per-step lengths and insertion branches, tree, tie-stream state
  * later search iterations of a -bb run (percentile or -cutoff_from_btrees cut-off, perturbed trees, every other one a ratchet iteration):
    the quiet stretch in k_climb / moot bookings / known optima without a product == every batch through the tracker's two-wait loop
  * the tracked climb (-bb bookkeeping) as a pipeline (ufb_pipe, decisions taken from the costs; its log on a second host thread or,
    ufb_thread 0, on the same one) == one chain per batch (ufb_pipe 0) == scan / wait / product / wait / replay (ufb_fast 0): moves, tree, saved trees, boot arrays, kept topologies, draws, tie state
  * the same tracked climb with the current tree's bookings extracted on the device (the exchange path of a sample-sharded run, one
    rank of one) == the host's walk over R_T; every fourth of these on the weighted engine (symmetric / asymmetric costs)
     python tools/soak.py [seconds] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from mpboot_amd import engine, synth, trees
import soak_lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
BIG = os.environ.get("SOAK_BIG") == "1"       # 120-319 taxa, 100-399 samples: whole-sweep batches of tens of thousands of indices (the chunked extraction, the overflow rule)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
n_climb = n_ref = n_samples = n_trk = n_early = n_grow = n_iter = n_quiet = n_memo = n_fail = n_echo = n_echo_w = 0
while time.time() < t_end:
    alpha = "AA" if rng.random() < 0.25 else "DNA"
    n = int(rng.integers(5, 90)) if not BIG else int(rng.integers(120, 320))
    P = int(rng.integers(40, 2500 if n >= 10 else 120))      # (the generator tops up to P DISTINCT variable columns: few taxa, few patterns)
    letters, _ = synth.synth_alignment(n, P, alpha, float(rng.uniform(0.02, 0.3)), seed=int(rng.integers(1 << 30)))
    codes = synth.letters_to_codes(letters, alpha)
    dt = engine.DNA if alpha == "DNA" else engine.AA
    w = rng.integers(1, 4, size=codes.shape[1]).astype(np.int32) if rng.random() < 0.3 else None
    back = trees.random_topology(n, np.random.default_rng(int(rng.integers(1 << 30))))
    tie = engine.TIE_RANDOM if rng.random() < 0.8 else engine.TIE_FIRST
    radius = int(rng.integers(1, 7)) if rng.random() < 0.85 else int(rng.integers(7, 16))      # (above 6: host loops; above 8: the deep kernels)
    seed = int(rng.integers(1, 1 << 20))
    if os.environ.get("SOAK_VERBOSE"):
        print(f"case {n_climb}: {alpha} n={n} P={P} tie={tie} radius={radius} seed={seed} weighted={w is not None} t={time.time() - (t_end - budget):.1f}", flush=True)
    # ---- climb: host loop vs kernel
    res = []
    for mode, opts in ((0, {}), (2, {"climb_tile": int(rng.choice([1, 2, 4, 8])) if alpha == "DNA" else 1,
                                     "climb_batch_min": int(rng.integers(1, 9)), "climb_batch_max": 8})):
        e = engine.FitchEngine(codes, w, datatype=dt)
        e.set_option("climb_device", mode)
        for k, v in opts.items():
            e.set_option(k, v)
        e.set_tree(back); e.reset_node_order(); e.seed_ties(tie, seed)
        if os.environ.get("SOAK_VERBOSE") == "2":
            print("  climb mode", mode, opts, flush=True)
        s = e.optimize_spr(1, radius)
        res.append((s, [x.tolist() for x in e.moves()], e.get_tree().tolist(), e.tie_state()))
        opt_tree = e.get_tree()
    assert res[0] == res[1], ("climb mismatch", alpha, n, P, tie, radius, seed, opts)
    n_climb += 1
    # ---- start tree: k_grow vs the host-driven addition loop
    if n >= 6:
        gres = []
        gseed = int(rng.integers(1, 1 << 20))
        for dev, tile in ((0, -1), (1, int(rng.choice([-1, 0, 1, 2, 4, 8])) if alpha == "DNA" else -1)):
            e = engine.FitchEngine(codes, w, datatype=dt)
            e.set_option("grow_device", dev)
            e.set_option("grow_tile", tile)
            e.seed_ties(tie, seed)
            sc_, best_, ins_ = e.stepwise_addition(gseed)
            gres.append((sc_, best_.tolist(), ins_.tolist(), e.get_tree().tolist(), e.tie_state()))
            if dev:
                assert e.get_option("grow_launches") == 1 and e.get_option("grow_last_err") == 0, ("k_grow did not build the tree", alpha, n, P, tile)
        assert gres[0] == gres[1], ("k_grow mismatch", alpha, n, P, tie, seed, gseed, tile)
        n_grow += 1
    # ---- refine sweep vs per-sample climbs (random tie rule only)
    if n >= 6:
        B = int(rng.integers(3, 20))
        w0 = w if w is not None else np.ones(codes.shape[1], dtype=np.int32)
        nsite = int(w0.sum())
        sp = np.repeat(np.arange(len(w0)), w0)
        samples = np.stack([np.bincount(sp[rng.integers(0, nsite, size=nsite)], minlength=len(w0)) for _ in range(B)]).astype(np.uint16)
        seeds = rng.integers(1, 1 << 20, size=B)
        start = opt_tree if rng.random() < 0.7 else back
        r2 = int(rng.integers(1, 7))
        e = engine.FitchEngine(codes, w, datatype=dt)
        if rng.random() < 0.5:
            e.set_option("refine_chunk", int(rng.integers(1, 40)))
        e.seed_ties(engine.TIE_RANDOM, 0)
        e.ufboot_attach(samples, 0.5)
        e.reset_node_order(); e.set_tree(start)
        if os.environ.get("SOAK_VERBOSE") == "2":
            print("  refine B", B, "r2", r2, "chunk", e.get_option("refine_chunk"), "start is opt", start is opt_tree, flush=True)
        sc, stable, first = e.ufboot_refine_sweep(r2, seeds)
        if os.environ.get("SOAK_VERBOSE") == "2":
            print("  refine done", flush=True)
        e.ufboot_detach()
        solo = engine.FitchEngine(codes, w, datatype=dt)
        for b in range(B):
            if os.environ.get("SOAK_DUMP"):
                np.savez(os.environ["SOAK_DUMP"], codes=codes, samples=samples, w0=w0, start=start, seeds=seeds, r2=r2, dt=dt, b=b, weighted=w is not None)
            solo.set_weights(samples[b].astype(np.int32))
            solo.seed_ties(engine.TIE_RANDOM, int(seeds[b])); solo.reset_node_order(); solo.set_tree(start)
            s0 = solo.score_tree()
            solo.optimize_spr(1, r2)
            moved = len(solo.moves()[0]) > 0
            assert s0 == sc[b] and bool(stable[b]) == (not moved), ("refine mismatch", alpha, n, P, b, r2, int(seeds[b]), s0, int(sc[b]), bool(stable[b]), moved)
            n_samples += 1
        n_ref += 1
    # ---- tracked climb: the three ways through it
    if n >= 6:
        B = int(rng.integers(2, 70)) if not BIG else int(rng.integers(100, 400))
        w0 = w if w is not None else np.ones(codes.shape[1], dtype=np.int32)
        nsite = int(w0.sum())
        sp = np.repeat(np.arange(len(w0)), w0)
        samples = np.stack([np.bincount(sp[rng.integers(0, nsite, size=nsite)], minlength=len(w0)) for _ in range(B)]).astype(np.uint16)
        sb = int(rng.choice([1, 2, 4, 16, 64]))
        got = []
        for opts in ({"ufb_fast": 0}, {"ufb_pipe": 0}, {"ufb_thread": 0}, {}):
            e = engine.FitchEngine(codes, w, datatype=dt)
            e.set_option("scan_batch", sb)
            for k, v in opts.items():
                e.set_option(k, v)
            e.set_tree(back); e.reset_node_order(); e.seed_ties(tie, seed)
            e.ufboot_attach(samples, 0.5)
            s1 = e.optimize_spr(1, radius)
            logl, cnt, tr = e.ufboot_state()
            got.append((s1, [x.tolist() for x in e.moves()], e.get_tree().tolist(), e.ufboot_tree_logl().tolist(), logl.tolist(), cnt.tolist(),
                        tr.tolist(), e.ufboot_counters()["tie_draws"], e.tie_state(),
                        [e.ufboot_tree(int(t)).tolist() for t in sorted(set(tr.tolist())) if t >= 0]))
            if not opts:
                n_early += e.get_option("ufb_early_batches")
        assert all(g == got[0] for g in got[1:]), ("tracked climb mismatch", alpha, n, P, tie, radius, seed, B, sb)
        n_trk += 1
        # ---- the same climb with the current tree's bookings as DEVICE events (what a sample-sharded run does: one rank of one,
        # an exchange that hands back what it was given), on the Fitch engine and -- every fourth case -- the weighted engine
        # (symmetric or asymmetric random costs): == the host's own walk over R_T
        if not BIG and rng.random() < 0.5:
            import ctypes as C
            from mpboot_amd import shard
            S = 4 if alpha == "DNA" else 20
            cost = None
            if rng.random() < 0.25 and n <= 40 and P <= 900:
                m = rng.integers(1, 6, size=(S, S))
                cost = (np.triu(m, 1) + np.triu(m, 1).T).astype(np.uint32)
                if rng.random() < 0.5:
                    cost[np.triu_indices(S, 1)] += 2
            two = []
            for echo in (False, True):
                keep = {}

                def fn(_arg, tag, local_ptr, n_local, all_ptr, n_all_ptr):
                    buf = np.ctypeslib.as_array(C.cast(local_ptr, C.POINTER(C.c_uint32)), shape=(n_local, 3)).copy() if n_local else np.zeros((0, 3), dtype=np.uint32)
                    keep["buf"] = buf
                    all_ptr[0] = buf.ctypes.data if n_local else None
                    n_all_ptr[0] = n_local
                    return 0

                cb = shard.EXCHANGE_FN(fn)
                e = engine.FitchEngine(codes, w, datatype=dt, cost=cost)
                e.set_option("scan_batch", sb)
                e.set_tree(back); e.reset_node_order(); e.seed_ties(tie, seed)
                if echo:
                    e.ufboot_attach(samples, 0.5, shard=(0, 1), exchange=cb)
                else:
                    e.ufboot_attach(samples, 0.5)
                s1 = e.optimize_spr(1, min(radius, 6))
                logl, cnt, tr = e.ufboot_state()
                two.append((s1, [x.tolist() for x in e.moves()], e.get_tree().tolist(), e.ufboot_tree_logl().tolist(), logl.tolist(), cnt.tolist(), tr.tolist(),
                            e.ufboot_counters()["tie_draws"], e.tie_state()))
            assert two[0] == two[1], ("exchange-path mismatch", alpha, n, P, tie, radius, seed, B, sb, cost is not None)
            n_echo += 1
            n_echo_w += cost is not None
    # ---- later iterations of a -bb run: the round-5 shortcuts against the plain two-wait loop
    if n >= 8 and rng.random() < 0.6:
        B = int(rng.integers(2, 50))
        w0 = w if w is not None else np.ones(codes.shape[1], dtype=np.int32)
        nsite = int(w0.sum())
        sp = np.repeat(np.arange(len(w0)), w0)
        samples = np.stack([np.bincount(sp[rng.integers(0, nsite, size=nsite)], minlength=len(w0)) for _ in range(B)]).astype(np.uint16)
        btrees = rng.random() < 0.3
        iters = int(rng.integers(2, 6))
        it_seed = int(rng.integers(1 << 30))
        case = {"codes": codes, "weighted": w is not None, "alpha": alpha, "back": back, "samples": samples, "w0": w0.astype(np.int32), "tie": tie, "seed": seed, "radius": radius,
                "btrees": btrees, "iters": iters, "it_seed": it_seed}
        got = []
        variants = ({"ufb_quiet": 0, "ufb_moot": 0, "ufb_memo": 0, "ufb_pipe": 0}, {})
        for opts in variants:
            e = engine.FitchEngine(codes, w, datatype=dt)
            for k, v in opts.items():
                e.set_option(k, v)
            got.append(soak_lib.later_iterations(e, engine.FitchEngine(codes, w, datatype=dt), trees, case, tie))
            if not opts:
                n_quiet += e.get_option("ufb_quiet_climbs")
                n_memo += e.get_option("ufb_memo_batches")
        diff = soak_lib.first_difference(got[0], got[1])
        if diff is not None:
            # which shortcut? each one alone against the plain loop
            blame = []
            for alone in ({"ufb_moot": 0, "ufb_memo": 0}, {"ufb_quiet": 0, "ufb_memo": 0}, {"ufb_quiet": 0, "ufb_moot": 0}, {"ufb_quiet": 0, "ufb_moot": 0, "ufb_memo": 0}):
                e = engine.FitchEngine(codes, w, datatype=dt)
                for k, v in alone.items():
                    e.set_option(k, v)
                d = soak_lib.first_difference(got[0], soak_lib.later_iterations(e, engine.FitchEngine(codes, w, datatype=dt), trees, case, tie))
                blame.append((alone, d))
            n_fail += 1
            out = os.path.join(ROOT, "gpurun_out", f"soak_fail_{int(sys.argv[2]) if len(sys.argv) > 2 else 1}_{n_fail}.npz")
            os.makedirs(os.path.dirname(out), exist_ok=True)
            np.savez_compressed(out, **case)
            print("LATER ITERATIONS MISMATCH", alpha, n, P, "tie", tie, "radius", radius, "B", B, "btrees", btrees, "iters", iters, "->", diff, "| alone:", blame, "| case in", out, flush=True)
            if n_fail >= 4:
                break
        n_iter += iters
if n_fail:
    sys.exit(f"soak FAILED: {n_fail} later-iteration cases differ")
print(f"soak ok: {n_grow} start trees (k_grow == host loop), {n_iter} later -bb iterations two ways ({n_quiet} climbs began as plain ones, {n_memo} batches booked without a product); {n_climb} climbs (kernel == host loop), {n_ref} refine sweeps / {n_samples} samples (== per-sample climbs), "
      f"{n_trk} tracked climbs four ways ({n_early} batches decided from the costs), {n_echo} of them also with the current tree's bookings as device events ({n_echo_w} on the weighted engine)")

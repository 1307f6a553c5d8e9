"""bench.py's CPU side: how many cores this job may use, the reference (oracle/_ref, where it travelled with the snapshot) or the scalar
port (oracle/) timed on a bounded sample of each leg's workload, and the climb through the reference-side binding.  Only bench.py's
cpu_baseline legs call into oracle/ -- here."""
from __future__ import annotations

import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def physical_cores() -> int:
    """Distinct (socket, core) pairs of /proc/cpuinfo; half the logical CPUs if that cannot be read."""
    try:
        seen, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        if seen:
            return len(seen)
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_quota() -> int:
    """CPUs this process may use: the cgroup quota (cpu.max) and the affinity mask, whichever is smaller."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(codes, back, names, letters, alphabet, maxtrav, budget_s, all_cores=True):
    """Reference AVX code if oracle/_ref travelled with the snapshot, else the scalar port (oracle)."""
    from mpboot_amd import synth, trees
    n, P = codes.shape
    drv = os.path.join(ROOT, "oracle", "_ref", "pll_ref_driver")
    if os.path.exists(drv) and os.access(drv, os.X_OK):
        try:
            with tempfile.TemporaryDirectory() as tmp:
                aln = os.path.join(tmp, "a.phy")
                synth.write_phylip(aln, synth.letters_to_text(letters, alphabet), names)
                tf = os.path.join(tmp, "t.nwk")
                with open(tf, "w") as f:
                    f.write(trees.back_to_newick(back, names) + "\n")
                cmd = [drv, "time", aln, "DNA" if alphabet == "DNA" else "WAG", "0", tf, str(maxtrav), str(budget_s)]
                out = subprocess.run(cmd, capture_output=True, text=True, check=True, timeout=600).stdout
                # the reference is single-threaded; what the box's cores deliver together is N independent copies of it
                # (as N independent searches would run): N processes at once, N = the physical cores this job may use
                # (the container's cgroup quota counts)
                ncopy = min(physical_cores(), cpu_quota()) if all_cores else 0
                many = []
                if ncopy > 1:
                    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(ncopy)]
                    for pr in procs:
                        try:
                            many.append(pr.communicate(timeout=900)[0])
                        except Exception:
                            pr.kill()

            def rate(text):
                for l in text.splitlines():
                    t = l.split()
                    if t and t[0] == "timed":
                        return int(t[6]), float(t[8]), int(t[2]), int(t[4])
                return None

            r = rate(out)
            if r:
                tests, secs, done, tot = r
                res = {"value": n * P * tests / secs, "unit": "site-ops/s", "cores": 1, "kind": "reference",
                       "evals_per_s": tests / secs,
                       "sample": f"reference PLL AVX testInsertParsimony (oracle/_ref/pll_ref_driver time), {done} prune-node "
                                 f"scans cycling over the {tot} prune nodes of the same tree, radius {maxtrav} "
                                 f"({tests} insertion tests in {secs:.1f} s, 1 thread)"}
                rates = [x for x in map(rate, many) if x]
                if rates:
                    tot_rate = sum(t_ / s_ for t_, s_, _d, _t in rates)
                    res["all_cores"] = {"processes": len(rates), "physical_cores": physical_cores(), "logical_cpus": os.cpu_count(),
                                        "usable_cpus": cpu_quota(),
                                        "evals_per_s": tot_rate,
                                        "value": n * P * tot_rate,
                                        "sample": f"{len(rates)} concurrent copies of the same single-threaded run",
                                        "note": f"this job may use {cpu_quota()} of the host's {physical_cores()} physical cores: on the whole "
                                                f"host the CPU side would be about {physical_cores() / max(1, len(rates)):.0f}x this"}
                return res
        except Exception as exc:  # fall through to the port
            print(f"[bench] reference driver failed ({exc}); using the scalar port", file=sys.stderr)
    from oracle import pyoracle as po
    o = po.Oracle(codes, datatype=po.DNA if alphabet == "DNA" else po.AA)
    o.set_tree(back)
    cur = o.score_tree()
    o.seed_ties(po.TIE_RANDOM, 1)
    order = o.nodep()[1:2 * n - 1]
    t0 = time.perf_counter()
    done = 0
    for rec in order:
        o.set_best(cur)
        o.rearrange(int(rec), 1, maxtrav)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    secs = time.perf_counter() - t0
    tests = o.counters()[2]
    return {"value": n * P * tests / secs, "unit": "site-ops/s", "cores": 1, "kind": "port",
            "evals_per_s": tests / secs,
            "sample": f"scalar C oracle over the first {done} of {len(order)} prune nodes of the same tree, radius {maxtrav} "
                      f"({tests} insertion tests, {secs:.1f} s, 1 thread)"}


def refine_cpu_baseline(codes, names, letters, alphabet, samples, boot_trees, gpu_scores, maxtrav):
    """CPU side of the bootstrap metric: the refinement of bootstrap samples (re-weight, rebuild the parsimony structures,
    SPR hill climb from the sample's tree) on the REFERENCE's PLL code (oracle/_ref/pll_ref_driver refine), one thread for
    sample 0 and then one sample per usable core at once; the scalar port only when the reference build did not travel."""
    from mpboot_amd import shard, synth, trees
    drv = os.path.join(ROOT, "oracle", "_ref", "pll_ref_driver")
    n, P = codes.shape
    if os.path.exists(drv) and os.access(drv, os.X_OK):
        try:
            with tempfile.TemporaryDirectory() as tmp:
                aln = os.path.join(tmp, "a.phy")
                synth.write_phylip(aln, synth.letters_to_text(letters, alphabet), names)
                ncopy = max(1, min(physical_cores(), cpu_quota(), len(boot_trees)))

                def cmd(b):
                    tf, wf = os.path.join(tmp, f"t{b}.nwk"), os.path.join(tmp, f"w{b}.txt")
                    with open(tf, "w") as f:
                        f.write(trees.back_to_newick(boot_trees[b], names) + "\n")
                    with open(wf, "w") as f:
                        f.write(" ".join(map(str, samples[b].astype(np.int64).tolist())) + "\n")
                    return [drv, "refine", aln, "DNA" if alphabet == "DNA" else "WAG", "0", tf, str(maxtrav), wf]

                def parse(text):
                    for l in text.splitlines():
                        t = l.split()
                        if t and t[0] == "refined":
                            return int(t[2]), int(t[4]), int(t[6]), float(t[10])
                    return None

                t0 = time.perf_counter()
                one = parse(subprocess.run(cmd(0), capture_output=True, text=True, check=True, timeout=900).stdout)
                wall1 = time.perf_counter() - t0
                res = {"refinement_per_1000_samples_s": 1000.0 * one[3], "cores": 1, "kind": "reference",
                       "sample": f"sample 0 on the reference's PLL AVX code (pll_ref_driver refine: re-weight + compressDNA + SPR hill "
                                 f"climb, first-best rule): {one[2]} moves, {one[3]:.2f} s (process wall {wall1:.2f} s incl. alignment parsing)",
                       "final_score_reference": one[1], "final_score_gpu": int(gpu_scores[0]),
                       "note": "scores may differ by the tie rule (reference first-best vs mpboot's random ties): both are SPR-local optima"}
                if ncopy > 1:
                    t0 = time.perf_counter()
                    procs = [subprocess.Popen(cmd(b), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for b in range(ncopy)]
                    outs = []
                    for pr in procs:
                        try:
                            outs.append(parse(pr.communicate(timeout=1800)[0]))
                        except Exception:
                            pr.kill()
                    wall = time.perf_counter() - t0
                    ok = [o for o in outs if o]
                    if ok:
                        per_sample = max(o[3] for o in ok) / len(ok)     # N samples finished within the slowest one's time
                        res["all_cores"] = {"processes": len(ok), "physical_cores": physical_cores(), "usable_cpus": cpu_quota(),
                                            "refinement_per_1000_samples_s": 1000.0 * per_sample, "wall_s": wall,
                                            "sample": f"samples 0..{len(ok) - 1}, one single-threaded process each, at once"}
                return res
        except Exception as exc:
            print(f"[bench] reference refine failed ({exc}); using the scalar port", file=sys.stderr)
    from oracle import pyoracle as po
    o = po.Oracle(codes, datatype=po.DNA if alphabet == "DNA" else po.AA)
    tc0 = time.perf_counter()
    o.set_weights(samples[0].astype(np.int32))
    o.seed_ties(po.TIE_RANDOM, shard.unit_seed(7, 0))
    o.set_tree(boot_trees[0])
    s_cpu = o.optimize_spr(1, maxtrav)
    tc = time.perf_counter() - tc0
    return {"refinement_per_1000_samples_s": 1000.0 * tc, "cores": 1, "kind": "port", "sample": "sample 0 on the scalar C oracle",
            "same_score_as_gpu": bool(int(gpu_scores[0]) == int(s_cpu))}


def bb_run_cpu_baseline(codes, alphabet, samples, back, maxtrav, budget_s, plain_evals_per_s, run, refine_cb):
    """CPU time of the same -bb run, extrapolated: the trees that reach saveCurrentTree's bookkeeping (run["trees_booked"]) at the rate
    of the C port with its AVX2 REPS loop, timed here on a prefix of the first climb's first sweep (kind "port": IQTree::saveCurrentTree
    lives in the unbuildable C++ layer); every other insertion test at the reference's own plain rate (PLL AVX testInsertParsimony,
    the headline's cpu_baseline); the refinement at the reference's per-sample time where that leg measured it."""
    import time
    from oracle import pyoracle as po
    if not plain_evals_per_s:
        return None
    o = po.Oracle(codes, datatype=po.DNA if alphabet == "DNA" else po.AA)
    o.seed_ties(po.TIE_RANDOM, 1)
    o.ufboot_attach(samples)
    o.set_best(o.score_tree(back))
    nodep = o.nodep()
    n = codes.shape[0]
    t0 = time.perf_counter()
    k0, i = o.counters()[2], 1
    while time.perf_counter() - t0 < budget_s and i <= 2 * n - 2:
        o.rearrange(int(nodep[i]), 1, maxtrav)
        i += 1
    dt = time.perf_counter() - t0
    booked_rate = (o.counters()[2] - k0) / dt
    booked = run["trees_booked"]
    plain = max(0, run["insertion_tests"] - booked)
    online_s = booked / booked_rate + plain / plain_evals_per_s
    refine_s = None
    if isinstance(refine_cb, dict):
        per1000 = refine_cb.get("refinement_per_1000_samples_s")
        if per1000:
            refine_s = per1000 * samples.shape[0] / 1000.0
    total = online_s + (refine_s or 0.0)
    return {"value": total, "unit": "s (one core, extrapolated)", "cores": 1, "kind": "port",
            "booked_trees_per_s": booked_rate, "plain_tests_per_s": plain_evals_per_s, "online_s": online_s, "refinement_s": refine_s,
            "sample": "%d prune nodes (%d insertion tests with REPS for %d samples) of the first sweep on the C port; extrapolated to %d booked "
                      "trees + %d unbooked insertion tests at the reference's plain rate%s"
                      % (i - 1, o.counters()[2] - k0, samples.shape[0], booked, plain,
                         " + the refinement at the reference's per-sample time" if refine_s else " (refinement not included)")}


def climb_cpu_baseline(names, letters, alphabet, back, maxtrav):
    """A whole SPR hill climb from `back` on the reference's PLL AVX code (pll_ref_driver spr: the loop of
    fastDNAparsimony.c:1919-1938, first-best rule), one thread."""
    from mpboot_amd import synth, trees
    drv = os.path.join(ROOT, "oracle", "_ref", "pll_ref_driver")
    if not (os.path.exists(drv) and os.access(drv, os.X_OK)):
        return None
    try:
        with tempfile.TemporaryDirectory() as tmp:
            aln, tf = os.path.join(tmp, "a.phy"), os.path.join(tmp, "t.nwk")
            synth.write_phylip(aln, synth.letters_to_text(letters, alphabet), names)
            with open(tf, "w") as f:
                f.write(trees.back_to_newick(back, names) + "\n")
            out = subprocess.run([drv, "spr", aln, "DNA" if alphabet == "DNA" else "WAG", "0", tf, str(maxtrav)], capture_output=True, text=True,
                                 check=True, timeout=1800, env=dict(os.environ, REF_DRIVER_QUIET="1")).stdout
        for l in out.splitlines():
            t = l.split()
            if t and t[0] == "climb":
                fin = [x.split()[1] for x in out.splitlines() if x.startswith("final_score")]
                return {"seconds": float(t[6]), "moves": int(t[2]), "sweeps": int(t[4]), "final_score": int(fin[0]) if fin else None,
                        "cores": 1, "kind": "reference",
                        "sample": "the whole climb from the same start tree, reference PLL AVX code, first-best rule (the GPU run draws "
                                  "mpboot's random ties: another path to another local optimum of similar length)"}
    except Exception as exc:
        print(f"[bench] reference climb failed ({exc})", file=sys.stderr)
    return None


def start_trees_cpu_baseline(names, letters, alphabet, maxtrav, n_trees):
    """The reference's start-up tree (randomized stepwise addition + the SPR sweeps behind it, pllMakeParsimonyTreeFast,
    fastDNAparsimony.c:1857 -- the PLL twin of _pllComputeRandomizedStepwiseAdditionParsimonyTree) on the box's host cores:
    one tree on one thread, then one tree per usable core side by side (oracle/_ref/pll_ref_driver ras)."""
    from mpboot_amd import synth
    drv = os.path.join(ROOT, "oracle", "_ref", "pll_ref_driver")
    if not (os.path.exists(drv) and os.access(drv, os.X_OK)):
        return None
    try:
        with tempfile.TemporaryDirectory() as tmp:
            aln = os.path.join(tmp, "a.phy")
            synth.write_phylip(aln, synth.letters_to_text(letters, alphabet), names)
            env = dict(os.environ, REF_DRIVER_QUIET="1")

            def cmd(u):
                return [drv, "ras", aln, "DNA" if alphabet == "DNA" else "WAG", "0", str(31337 + 12345 * u), str(maxtrav)]

            def secs(text):
                for l in text.splitlines():
                    if l.startswith("ras_seconds"):
                        return float(l.split()[1])
                return None

            one = secs(subprocess.run(cmd(0), capture_output=True, text=True, check=True, timeout=900, env=env).stdout)
            k = max(1, min(cpu_quota(), 16))
            t0 = time.perf_counter()
            procs = [subprocess.Popen(cmd(u), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for u in range(k)]
            each = [secs(pr.communicate(timeout=1800)[0]) for pr in procs]
            wall = time.perf_counter() - t0
        each = [x for x in each if x is not None]
        if one is None or not each:
            return None
        per_tree_all = max(each) / len(each)             # k trees finish in max(each) seconds of build time
        return {"seconds_per_tree_one_core": one, "cores": 1, "kind": "reference",
                "all_cores": {"cores": len(each), "seconds_per_tree_each": each, "wall_s_incl_parsing": wall,
                              "seconds_for_%d_trees" % n_trees: per_tree_all * n_trees},
                "seconds_for_%d_trees_one_core" % n_trees: one * n_trees,
                "sample": "one start tree (compressDNA + randomized stepwise addition + SPR sweeps, radius %d) by the reference's PLL AVX "
                          "code on one thread, then %d such trees side by side (one process per usable core); scaled to %d trees"
                          % (maxtrav, len(each), n_trees)}
    except Exception as exc:
        print(f"[bench] reference start trees failed ({exc})", file=sys.stderr)
    return None


def shim_climb_leg(eng, names, letters, alphabet, back, maxtrav, reps=3):
    """pllOptimizeSprParsimony THROUGH the reference-side binding: oracle/_ref/spr_shim_driver is the reference's own PLL program
    (alignment parser, pllInstance, Newick reader, SPRNG generator -- compiled from the reference's sources) linked with
    integration/sprparsimony_shim.cpp instead of sprparsimony.cpp; its `time` mode puts the start tree into the pllInstance and times
    the call itself (topology marshalled in and out, the SPRNG state handed over and taken back).  The same call is then replayed on
    this process's engine from the topology the driver printed: score, number of moves and the generator's final state must agree."""
    from mpboot_amd import engine, synth, trees
    drv = os.path.join(ROOT, "oracle", "_ref", "spr_shim_driver")
    if not (os.path.exists(drv) and os.access(drv, os.X_OK)):
        return None
    try:
        with tempfile.TemporaryDirectory() as tmp:
            aln, tf = os.path.join(tmp, "a.phy"), os.path.join(tmp, "t.nwk")
            synth.write_phylip(aln, synth.letters_to_text(letters, alphabet), names)
            with open(tf, "w") as f:
                f.write(trees.back_to_newick(back, names) + "\n")
            out = subprocess.run([drv, "time", aln, "DNA" if alphabet == "DNA" else "WAG", "1", str(maxtrav), tf, str(reps), "1"],
                                 capture_output=True, text=True, check=True, timeout=900).stdout
        runs, start = [], None
        for l in out.splitlines():
            t = l.split()
            if t and t[0] == "time_start_topology":
                start = np.full(len(back), -1, dtype=np.int32)
                for tok in t[1:]:
                    a, b = tok.split(":")
                    start[int(a)] = int(b)
            elif t and t[0] == "time_climb":
                runs.append({t[i]: (float(t[i + 1]) if t[i] == "seconds" else int(t[i + 1])) for i in range(1, len(t) - 1, 2)})
        if not runs or start is None:
            return None
        eng.set_tree(start)
        eng.reset_node_order()
        eng.seed_ties(engine.TIE_RANDOM, 1)
        eng.reset_stats()
        sc = eng.optimize_spr(1, maxtrav)
        same = all(r["score"] == sc and r["moves"] == eng.stats()["moves_applied"] and r["rng_state"] == eng.tie_state() for r in runs)
        later = [r["seconds"] for r in runs[1:]] or [runs[0]["seconds"]]
        return {"seconds": min(later), "seconds_each_call": [r["seconds"] for r in runs], "score": runs[-1]["score"], "moves": runs[-1]["moves"],
                "climb_launches": runs[-1]["climb_launches"], "insertion_tests": runs[-1]["insertion_tests"],
                "same_as_in_process_engine": bool(same),
                "what": "the same climb through integration/sprparsimony_shim.cpp on the reference's pllInstance (oracle/_ref/spr_shim_driver time): "
                        "wall clock around pllOptimizeSprParsimony itself, SPRNG stream handed over by state (mpf_set_tie_state) so the sweep "
                        "loop runs in k_climb; the first call also creates the engine (tips packed, buffers allocated)"}
    except Exception as exc:
        print(f"[bench] shim climb leg failed ({exc})", file=sys.stderr)
    return None

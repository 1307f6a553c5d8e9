#!/usr/bin/env python3
"""Time one pllOptimizeSprParsimony call with the online UFBoot-MP bookkeeping attached (-bb mode) on the GPU engine,
and verify the result through an independent device path (per-pattern lengths + REPS of the trees the samples kept)."""
import argparse, sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpboot_amd import engine, synth, bootstrap

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C2")
ap.add_argument("--samples", type=int, default=1000)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--start", default="ras")
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--storetrees", action="store_true", help="-storetrees: every tree looked up by topology before it is booked")
ap.add_argument("--check", action="store_true", help="replay the first climb on the CPU oracle and compare every observable")
ap.add_argument("--timing", type=int, default=0, help="engine option timing during the tracked climb (events around kernels cost the batches a synchronisation each)")
ap.add_argument("--verify", type=int, default=8, help="number of samples whose kept tree is re-scored independently")
a = ap.parse_args()
cfg = synth.WORKLOADS[a.workload]
letters, names = synth.workload(a.workload)
codes = synth.letters_to_codes(letters, cfg["alphabet"])
dt = engine.DNA if cfg["alphabet"] == "DNA" else engine.AA
e = engine.FitchEngine(codes, datatype=dt)
for kv in a.opt:
    k, v = kv.split("="); e.set_option(k, int(v))
e.seed_ties(engine.TIE_RANDOM, a.seed)
P = codes.shape[1]
rng = np.random.default_rng(a.seed)
samples = rng.multinomial(P, np.ones(P) / P, size=a.samples).astype(np.uint16)
if a.start == "ras":
    s0 = e.make_parsimony_tree(1000 + a.seed, 0)
else:
    from mpboot_amd import trees
    s0 = e.score_tree(trees.random_topology(codes.shape[0], np.random.default_rng(a.seed)))
back0 = e.get_tree()
# plain climb for reference
e.set_option("timing", 2)
e.reset_stats()
t0 = time.perf_counter(); s_plain = e.optimize_spr(1, 6); t1 = time.perf_counter()
st_plain = e.stats()
moves_plain = [x.tolist() for x in e.moves()]
# same climb with the tracker
e.set_tree(back0); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, a.seed)
ta = time.perf_counter(); e.ufboot_attach(samples); tb = time.perf_counter()
if a.storetrees:
    e.ufboot_set_store_trees(True)
e.set_option("timing", a.timing)
e.reset_stats()
t2 = time.perf_counter(); s_ufb = e.optimize_spr(1, 6); t3 = time.perf_counter()
st = e.stats(); cn = e.ufboot_counters()
print(f"{a.workload} B={a.samples}: start {s0}; plain climb -> {s_plain} in {t1-t0:.3f}s ({st_plain['insertion_tests']} tests, {st_plain['moves_applied']} moves); "
      f"attach {tb-ta:.2f}s; climb with online UFBoot -> {s_ufb} in {t3-t2:.3f}s ({st['insertion_tests']} tests, {st['moves_applied']} moves, "
      f"scan kernels {st['scan_kernel_ms_total']:.1f} ms, REPS product {cn['reps_kernel_ms']:.1f} ms over {cn['reps_rows']} rows, "
      f"{cn['events']} events, {cn['tie_draws']} draws, {len(e.ufboot_tree_logl())} saved trees"
      + (f", {e.ufboot_duplicates()} duplicates" if a.storetrees else "") + ")"
      + f"; batches of the tracked climb {e.get_option('ufb_batches')}, decided from the costs {e.get_option('ufb_early_batches')}")
rows, W = cn["reps_rows"], e.Wp
if cn["reps_kernel_ms"] > 0:
    ops = 2.0 * rows * (W * 32) * (-(-a.samples // 128) * 128)
    print(f"REPS product: {ops / cn['reps_kernel_ms'] / 1e9:.1f} TOP/s (i8 MFMA, dense count of the padded product)")
logl, counts, tr = e.ufboot_state()
final = e.get_tree()
# independent check: per-pattern lengths of the kept trees (k_site_planes) x weights (numpy)
for b in range(min(a.verify, a.samples)):
    t = e.ufboot_tree(int(tr[b]))
    e.set_tree(t)
    ptn, tot = e.pattern_scores()
    rell = -int((ptn.astype(np.int64) * samples[b]).sum())
    assert rell == int(logl[b]), (b, rell, logl[b])
print(f"verified {min(a.verify, a.samples)} samples: boot_logl equals the REPS of the kept tree recomputed from per-pattern lengths")

if a.check:
    from oracle import pyoracle as po
    o = po.Oracle(codes, datatype=po.DNA if dt == engine.DNA else po.AA)
    o.set_tree(back0); o.seed_ties(po.TIE_RANDOM, a.seed); o.ufboot_attach(samples); o.trace(True)
    if a.storetrees:
        o.ufboot_set_store_trees(True)
    tc0 = time.perf_counter(); so = o.optimize_spr(1, 6); tc1 = time.perf_counter()
    lo, co, to = o.ufboot_state()
    ok = (so == s_ufb and (o.get_tree() == final).all() and lo.tolist() == logl.tolist() and co.tolist() == counts.tolist()
          and to.tolist() == tr.tolist() and o.ufboot_tree_logl().tolist() == e.ufboot_tree_logl().tolist()
          and o.ufboot_draws() == cn["tie_draws"] and o.ufboot_bad() == 0)
    for t in sorted(set(tr.tolist())):
        ok = ok and (o.ufboot_tree(t) == e.ufboot_tree(t)).all()
    print(f"oracle (CPU, {tc1-tc0:.1f}s): identical score, final tree, saved-tree list, boot_logl/boot_counts/boot_trees (+ topologies) and draw count: {ok}")
    assert ok

# a later search iteration: cut-off from the saved trees (top 10 %), perturbed tree, climb again
from mpboot_amd import trees as _trees
cut = e.ufboot_next_cutoff(10)
e.ufboot_set_cutoff(cut)
pert = _trees.random_spr_moves(e, final, np.random.default_rng(a.seed + 5), 30)
e.set_tree(pert); e.reset_node_order()
s_pert = e.score_tree()
n_before = len(e.ufboot_tree_logl()); c0 = e.ufboot_counters(); e.reset_stats()
t4 = time.perf_counter(); s2 = e.optimize_spr(1, 6); t5 = time.perf_counter()
st2 = e.stats(); c1 = e.ufboot_counters()
print(f"next iteration: cut-off {-cut:.0f}; 30 random SPR moves -> {s_pert}; climb -> {s2} in {t5-t4:.3f}s ({st2['insertion_tests']} tests, {st2['moves_applied']} moves, "
      f"{len(e.ufboot_tree_logl()) - n_before} saved, REPS rows {c1['reps_rows'] - c0['reps_rows']}, product {c1['reps_kernel_ms'] - c0['reps_kernel_ms']:.1f} ms, events {c1['events'] - c0['events']})")
e.ufboot_detach()

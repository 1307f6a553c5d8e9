"""Shared by tools/soak.py and tests/test_gpu_soak_cases.py: the later iterations of a -bb run on one engine-like object
(mpboot_amd.engine.FitchEngine, or the oracle's wrapper in the tests -- same method names), snapshot after every climb."""
import numpy as np

from mpboot_amd.trees import splits


def snapshot(x):
    logl, cnt, tr = x.ufboot_state()
    draws = x.ufboot_counters()["tie_draws"] if hasattr(x, "ufboot_counters") else x.ufboot_draws()
    snap = {"moves": [m.tolist() for m in x.moves()]} if hasattr(x, "moves") else {}
    return snap | {"tree": x.get_tree().tolist(), "tie_state": x.tie_state(), "tree_logl": x.ufboot_tree_logl().tolist(),
            "boot_logl": logl.tolist(), "boot_counts": cnt.tolist(), "boot_trees": tr.tolist(), "orig_logl": x.ufboot_orig_logl().tolist(),
            "draws": draws, "kept": [sorted(sorted(sp) for sp in splits(x.ufboot_tree(int(t)))) for t in sorted(set(tr.tolist())) if t >= 0]}       # (inner nodes are numbered per store)


def later_iterations(x, scratch, trees, case, tie_mode):
    """case: dict with back, samples, w0, tie, seed, radius, btrees, iters, it_seed.  Returns one snapshot per climb."""
    w0 = case["w0"]
    radius = int(case["radius"])
    x.set_tree(case["back"])
    if hasattr(x, "reset_node_order"):
        x.reset_node_order()
    x.seed_ties(tie_mode, int(case["seed"]))
    x.ufboot_attach(case["samples"], 0.5)
    if case["btrees"]:
        x.ufboot_set_cutoff_from_btrees(True)
    snaps = []
    best_s = x.optimize_spr(1, radius)
    best_t = x.get_tree()
    snaps.append(("first", best_s, snapshot(x)))
    r2 = np.random.default_rng(int(case["it_seed"]))
    for it in range(int(case["iters"])):
        tl = x.ufboot_tree_logl()
        x.ufboot_set_cutoff(x.ufboot_next_cutoff(10) if case["btrees"] or it % 3 == 0 else float(np.sort(tl)[::-1][len(tl) // 5]))
        pert = trees.random_spr_moves(scratch, best_t, r2, int(r2.integers(1, 9)), max(radius, 1))
        if it % 2 == 1:
            wr = w0.copy()
            wr[r2.random(len(w0)) < 0.5] += 1
            x.set_weights(wr); x.set_tree(pert)
            s_r = x.optimize_spr(1, radius)
            snaps.append((f"ratchet {it}", s_r, snapshot(x)))
            pert = x.get_tree()
            x.set_weights(w0)
        x.set_tree(pert)
        s_it = x.optimize_spr(1, radius)
        snaps.append((f"iter {it}", s_it, snapshot(x)))
        if s_it <= best_s:
            best_s, best_t = s_it, x.get_tree()
    return snaps


def first_difference(a, b):
    for (la, sa, da), (lb, sb, db) in zip(a, b):
        if la != lb or sa != sb:
            return f"{la}: score {sa} vs {sb}"
        for k in da:
            if k in db and da[k] != db[k]:
                return f"{la}: {k}"
    return None if len(a) == len(b) else "length"

"""Random SEQUENCES of calls on one engine against the oracle replaying the same calls.

The engine keeps state between calls that the oracle does not have -- validity of the directional vectors, the plans cached per
topology (refresh schedule, sweep descriptors, device program), speculative batch sizes, pooled scratch buffers -- so what
matters here is the ORDER of operations: the same tree handed over again, re-weighting between scans of one topology, sweeps
after climbs, a tree builder in between, options that change the kernels, a tracker attached half-way.  Every observable of
every call must equal the oracle's."""
import os

import numpy as np
import pytest

from test_gpu_fuzz import random_case

pytestmark = pytest.mark.gpu

STATE_OFFSET = int(os.environ.get("MPF_FUZZ_OFFSET", "0"))
OPTION_SETS = [{}, {"scan_batch": 3, "split_below": 0}, {"scan_prog": 2, "split_below": 0, "scan_batch": 64}, {"views_mode": 1, "scan_batch": 2},
               {"scan_prog": 2, "check_counts": 1}, {"views_mode": 2, "scan_batch": 1}, {"plan_cache": 0}, {"words_per_lane": 2},
               {"scan_batch": 64, "prog_min_descs": 0}]


@pytest.mark.parametrize("seed", range(60))
def test_random_call_sequences_match_oracle(seed):
    c = random_case(7000 + seed + STATE_OFFSET)
    if c["aa"] and seed % 3:
        c = random_case(9000 + seed + STATE_OFFSET)
    run_sequence(c, seed, 14)


@pytest.mark.parametrize("seed", range(10))
def test_random_call_sequences_on_medium_trees(seed):
    """60-200 taxa: sweeps that are cut into several batches, split scans, the planned-program kernel, whole-tree refreshes"""
    from mpboot_amd import synth, trees

    rng = np.random.default_rng(555 + seed + STATE_OFFSET)
    n, P = int(rng.integers(60, 201)), int(rng.integers(150, 600))
    aa = seed % 5 == 4
    letters, _ = synth.synth_alignment(n, P, "AA" if aa else "DNA", float(rng.uniform(0.03, 0.12)), seed=seed + STATE_OFFSET)
    codes = synth.letters_to_codes(letters, "AA" if aa else "DNA").copy()
    codes[rng.random(codes.shape) < 0.02] = 22 if aa else 15
    w = rng.integers(1, 3, size=P).astype(np.int32)
    c = dict(aa=aa, n=n, P=P, codes=codes, w=w, back=trees.random_topology(n, rng), maxtrav=int(rng.integers(2, 7)))
    if seed % 4 == 3:
        c["maxtrav"] = 9 + seed % 6               # long radii: host-planned programs, the deep kernels under the tracker
    run_sequence(c, seed, 8)


@pytest.mark.parametrize("seed", range(24))
def test_random_call_sequences_weighted(seed):
    """the same on the weighted (Sankoff) engine: random symmetric cost matrices, packed 16-bit and 32-bit costs"""
    c = random_case(11000 + seed + STATE_OFFSET)
    rng = np.random.default_rng(77 + seed + STATE_OFFSET)
    S = 20 if c["aa"] else 4
    m = rng.integers(1, 6 if seed % 3 else 4000, size=(S, S))
    cost = (np.triu(m, 1) + np.triu(m, 1).T).astype(np.uint32)
    run_sequence(c, seed, 10, cost=cost)


@pytest.mark.parametrize("seed", range(4))
def test_random_call_sequences_weighted_medium_trees(seed):
    """the weighted engine on 50-120 taxa: host-planned scans in several batches, the tracker's bit planes over many rows"""
    from mpboot_amd import synth, trees

    rng = np.random.default_rng(888 + seed + STATE_OFFSET)
    n, P = int(rng.integers(50, 121)), int(rng.integers(150, 400))
    aa = seed % 2 == 1
    letters, _ = synth.synth_alignment(n, P, "AA" if aa else "DNA", float(rng.uniform(0.04, 0.12)), seed=seed + STATE_OFFSET)
    codes = synth.letters_to_codes(letters, "AA" if aa else "DNA").copy()
    w = rng.integers(1, 3, size=P).astype(np.int32)
    S = 20 if aa else 4
    m = rng.integers(1, 6, size=(S, S))
    cost = (np.triu(m, 1) + np.triu(m, 1).T).astype(np.uint32)
    c = dict(aa=aa, n=n, P=P, codes=codes, w=w, back=trees.random_topology(n, rng), maxtrav=int(rng.integers(2, 5)))
    if seed >= 2:
        c["maxtrav"] = 7 + 3 * seed               # 13 (DNA), 16 (protein): k_snk_scan_deep, also under the tracker
    run_sequence(c, seed, 7, cost=cost)


def run_sequence(c, seed, n_steps, cost=None):
    from mpboot_amd import engine, trees
    from oracle import pyoracle as po

    rng = np.random.default_rng(seed + STATE_OFFSET)
    dt_e, dt_o = (engine.AA, po.AA) if c["aa"] else (engine.DNA, po.DNA)
    if cost is not None:
        c["w"] = np.maximum(c["w"], 0)
    keep_all = bool(c.get("keep_all", False)) and cost is None
    tmode = int(c.get("tie", 1)) if cost is None else 1          # 0 = first-best rule (PLL original + mpboot's pre-evaluate)
    e = engine.FitchEngine(c["codes"], c["w"], datatype=dt_e, cost=cost, keep_all=keep_all)
    o = po.Oracle(c["codes"], c["w"], datatype=dt_o, cost=cost, keep_all=keep_all)
    if tmode == 0:
        o.set_pre_evaluate(1)
    if o.num_informative == 0:
        return
    n, P = c["n"], c["P"]
    maxtrav = c["maxtrav"]                        # (any radius on both engines since round 5: the deep kernels above 8 / 12 / 6 levels)
    for k, v in OPTION_SETS[seed % len(OPTION_SETS)].items():
        if cost is not None and k == "words_per_lane":
            continue                            # (refused in weighted mode: one pattern per lane)
        e.set_option(k, v)
    known = [c["back"], trees.random_topology(n, rng)]
    for x in (e, o):
        x.set_tree(known[0])
        x.seed_ties(tmode, seed)
    weights = [c["w"]]
    tracked = mulhits = toplist = False
    samples_cur = None
    iteration = 0
    log = []
    for step in range(n_steps):
        op = int(rng.integers(0, 14))
        if tracked and op in (1, 2):
            op = 3                              # (the oracle's rearrangeParsimony books every test it makes: no bare scans once tracked)
        log.append(op)
        if str(step) in os.environ.get("MPF_STATEFUL_SKIP", "").split(","):
            continue
        if os.environ.get("MPF_STATEFUL_VERBOSE"):
            print("seed", seed, "step", step, "op", op, "tracked", tracked, "n", n, "P", P, "aa", c["aa"], "maxtrav", maxtrav, "opts", OPTION_SETS[seed % len(OPTION_SETS)], flush=True)
        if op == 0:                             # a tree seen before (same topology again -> cached plans) or a new one
            t = known[int(rng.integers(0, len(known)))] if rng.random() < 0.7 else trees.random_topology(n, rng)
            assert e.score_tree(t) == o.score_tree(t), log
        elif op == 1:                           # one prune node's candidates
            nodep = o.nodep()
            rec = int(nodep[1 + int(rng.integers(0, 2 * n - 2))])
            cur = o.score_tree()                # (both sides: the call re-orders the node table the tree builder draws from)
            assert e.score_tree() == cur, log
            o.set_best(cur)
            o.trace(True)
            saved = o.get_tree().copy()
            o.rearrange(rec, 1, maxtrav)
            tq, tm = o.get_trace()
            keep = tq >= 0
            assert (o.get_tree() == saved).all()
            q, mp, _ = e.spr_scan(rec, 1, maxtrav)
            assert q.tolist() == tq[keep].tolist() and mp.tolist() == tm[keep].tolist(), log
            for x in (e, o):
                x.seed_ties(tmode, seed + step)     # (the oracle's tie rule drew random numbers during that scan)
        elif op == 2:                           # a whole sweep: the best candidate score
            cur = o.score_tree()
            assert e.score_tree() == cur, log
            ntests, best = e.sweep_scan(1, maxtrav)
            lo = None
            for rec in o.nodep()[1:2 * n - 1]:
                o.set_best(cur)
                o.trace(True)
                o.rearrange(int(rec), 1, maxtrav)
                tq, tm = o.get_trace()
                if (tq >= 0).any():
                    m = int(tm[tq >= 0].min())
                    lo = m if lo is None else min(lo, m)
            if lo is not None:
                assert best == lo, log
            for x in (e, o):
                x.seed_ties(tmode, seed + step)
        elif op == 3:                           # a climb
            o.trace(True)
            if tracked:
                iteration += 1
                for x in (e, o):
                    x.ufboot_set_iteration(iteration)
            radius = int(rng.integers(1, maxtrav + 1))
            assert e.optimize_spr(1, radius) == o.optimize_spr(1, radius), log
            assert [a.tolist() for a in e.moves()] == [a.tolist() for a in o.get_moves()], log
            assert (e.get_tree() == o.get_tree()).all(), log
            known.append(e.get_tree().copy())
            if tracked:
                assert [a.tolist() for a in e.ufboot_state()] == [a.tolist() for a in o.ufboot_state()], log
                assert e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist(), log
                assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws(), log
                assert e.ufboot_duplicates() == o.ufboot_duplicates(), log
                if mulhits and not toplist:
                    for b in range(5):
                        assert e.ufboot_sample_trees(b) == o.ufboot_sample_trees(b), log
                if toplist:
                    for b in range(5):
                        assert e.ufboot_sample_top(b) == o.ufboot_sample_top(b), log
        elif op == 4:                           # re-weighting (between scans of one topology: the plans stay, the vectors go)
            if rng.random() < 0.5:
                w = weights[int(rng.integers(0, len(weights)))]
            else:
                w = (weights[0] * (1 + (rng.random(P) < 0.3) * rng.integers(1, 3, size=P))).astype(np.int32)
                weights.append(w)
            for x in (e, o):
                x.set_weights(w)
            assert e.score_tree() == o.score_tree(), log
        elif op == 5:                           # a stepwise-addition tree
            sd = int(rng.integers(1, 1000))
            dist = int(rng.integers(0, 4))
            assert e.make_parsimony_tree(sd, dist) == o.make_tree(sd, dist)[0], log
            assert (e.get_tree() == o.get_tree()).all(), log
            known.append(e.get_tree().copy())
        elif op == 6:                           # per-pattern lengths of the current tree
            o.enable_persite(True)
            assert e.score_tree() == o.score_tree(), log
            pe, te = e.pattern_scores()
            po_, to = o.pattern_scores()
            assert te == to and pe.tolist() == po_.tolist(), log
        elif op == 7 and not tracked and (weights[0] > 0).any():      # attach the bookkeeping half-way
            for x in (e, o):
                x.set_weights(weights[0])
            samples = rng.multinomial(max(1, int(weights[0].sum())), (weights[0] + 1e-9) / (weights[0] + 1e-9).sum(), size=5).astype(np.uint16)
            e.ufboot_attach(samples)
            o.ufboot_attach(samples)
            tracked = True
            samples_cur = samples
            rule = int(rng.integers(0, 6))              # 0-2 default, 3 -mulhits, 4 -mulhits -topboot, 5 -distinct_iter_top_boot
            mulhits = rule in (3, 4)
            toplist = rule in (4, 5)
            store = bool(rng.random() < 0.25)           # -storetrees, with any of the rules
            for x in (e, o):
                if store:
                    x.ufboot_set_store_trees(True)
                if mulhits:
                    x.ufboot_set_mulhits(True)
                if rule == 4:
                    x.ufboot_set_topboot(3)
                if rule == 5:
                    x.ufboot_set_distinct_iter(2)
            iteration = 0
        elif op == 9:                           # an option that changes kernels / batching / caching, mid-way
            k, v = [("scan_batch", int(rng.integers(1, 65))), ("split_below", int(rng.integers(0, 3)) * 2048), ("plan_cache", int(rng.integers(0, 2))),
                    ("scan_prog", int(rng.integers(0, 3))), ("views_mode", int(rng.integers(0, 3))), ("split_cands", int(rng.integers(8, 65))),
                    ("prog_min_descs", int(rng.integers(0, 2)) * 256), ("host_poll", int(rng.integers(0, 2))),
                    ("views_pipe", int(rng.integers(0, 2))), ("views_tile", [32, 16, 8, 4, 0][int(rng.integers(0, 5))])][int(rng.integers(0, 10))]
            if os.environ.get("MPF_STATEFUL_VERBOSE"):
                print("   option", k, v, flush=True)
            e.set_option(k, v)
        elif op == 10 and tracked and len(o.ufboot_tree_logl()) > 20:      # a cut-off from the trees booked so far
            logl = np.sort(o.ufboot_tree_logl())
            cut = float(logl[int(rng.integers(0, len(logl)))])
            e.ufboot_set_cutoff(cut)
            o.ufboot_set_cutoff(cut)
        elif op == 11 and tracked and rng.random() < 0.3:
            e.ufboot_detach()
            o.ufboot_detach()
            tracked = mulhits = toplist = False
        elif op == 12:                          # the tie stream by state (what the reference-side binding does around every call)
            if cost is None:
                assert e.tie_state() == o.tie_state(), log
            st = int(rng.integers(0, 2 ** 63))
            for x in (e, o):
                x.set_tie_state(st)
        elif op == 13 and tracked and cost is None and tmode == 1:
            # the batched refinement sweep on the attached samples, in the middle of everything else: == one re-weighted climb per
            # sample on a second oracle; and it must leave the tracker, the plans and the tie stream of THIS engine alone (the
            # calls after it are still compared with the oracle, which never saw it)
            for x in (e, o):
                x.set_weights(weights[0])
            cur = e.get_tree().copy()
            r = int(rng.integers(1, maxtrav + 1))
            seeds = rng.integers(1, 1 << 20, size=len(samples_cur))
            st_before = e.tie_state()
            sc, stable, _first = e.ufboot_refine_sweep(r, seeds)
            assert e.tie_state() == st_before, log
            o2 = po.Oracle(c["codes"], weights[0], datatype=dt_o, keep_all=keep_all)
            for b in range(len(samples_cur)):
                o2.set_weights(samples_cur[b].astype(np.int32))
                o2.seed_ties(po.TIE_RANDOM, int(seeds[b]))
                o2.reset_nodep()
                assert o2.score_tree(cur) == sc[b], log
                o2.trace(True)
                o2.optimize_spr(1, r)
                assert bool(stable[b]) == (len(o2.get_moves()[0]) == 0), (log, b)
            assert e.score_tree() == o.score_tree(), log      # (and both node tables in the same order again)
        elif op == 8:                           # the same tree handed over again, explicitly
            t = e.get_tree().copy()
            for x in (e, o):
                x.set_tree(t)
            assert e.score_tree() == o.score_tree(), log
    assert e.score_tree() == o.score_tree(), log


def test_concurrent_engines_random_sequences():
    """several engines on one GPU, one host thread each (the shape of bootstrap.refine_boot_trees / integration/multi_device.hpp),
    each running its own random call sequence against its own oracle at the same time: nothing in the library is shared between
    engines but the device"""
    import threading

    errors = []

    def work(k):
        try:
            for seed in range(40 + 8 * k, 48 + 8 * k):
                c = random_case(13000 + seed + STATE_OFFSET)
                run_sequence(c, seed, 10)
        except BaseException as exc:      # noqa: BLE001 - reported by the main thread
            errors.append((k, repr(exc)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors

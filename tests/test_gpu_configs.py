"""Every configuration of BASELINE.json at full size on the GPU (C3 properties live in test_gpu_fullsize.py):

  C2  200 taxa x 10 000 DNA patterns   -- full SPR hill climb from a random tree, move for move against the oracle
  C5  500 taxa x 20 000 protein        -- Fitch-20: score, whole scans and a climb; weighted (-cost) 20-state scans
  C4  1000 x 50 000, -bb 1000          -- online UFBoot phase at C3 size with 96 and with 1000 samples: boot_logl of every sample
                                          re-derived from per-pattern lengths, and the sample-sharded run (two ranks, events
                                          exchanged) == the unsharded one, at both sample counts
  C3  1000 x 50 000                    -- whole scans of four prune nodes against the oracle
plus the value checks of mpf_spr_sweep_scan (the call bench.py times) against per-prune-node scans.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _workload(name):
    from mpboot_amd import engine, synth
    cfg = synth.WORKLOADS[name]
    letters, _ = synth.workload(name)
    codes = synth.letters_to_codes(letters, cfg["alphabet"])
    return codes, (engine.DNA if cfg["alphabet"] == "DNA" else engine.AA)


def _oracle_scan(o, po, back, rec, maxtrav):
    cur = o.score_tree(back)
    o.seed_ties(po.TIE_RANDOM, 1)
    o.set_best(cur)
    o.trace(True)
    o.rearrange(int(rec), 1, maxtrav)
    tq, tm = o.get_trace()
    keep = tq >= 0
    return tq[keep].tolist(), tm[keep].tolist()


# ------------------------------------------------------------------------------------------------ C2

def test_c2_full_hill_climb_matches_oracle():
    """BASELINE config 2: full SPR hill climb, 1 GPU -- accepted moves, final topology and score == the oracle's"""
    from mpboot_amd import engine, trees
    from oracle import pyoracle as po
    codes, dt = _workload("C2")
    back = trees.random_topology(codes.shape[0], np.random.default_rng(1))
    for opts in (dict(), dict(scan_prog=2)):
        e = engine.FitchEngine(codes, datatype=dt)
        for k, v in opts.items():
            e.set_option(k, v)
        o = po.Oracle(codes)
        assert e.score_tree(back) == o.score_tree(back)
        e.seed_ties(engine.TIE_RANDOM, 1)
        o.seed_ties(po.TIE_RANDOM, 1)
        o.trace(True)
        se, so = e.optimize_spr(1, 6), o.optimize_spr(1, 6)
        assert se == so
        assert len(e.moves()[0]) > 300
        assert [x.tolist() for x in e.moves()] == [x.tolist() for x in o.get_moves()]
        assert (e.get_tree() == o.get_tree()).all()
        if not opts:
            st = e.stats()
            assert st["plan_launches"] > 0                    # the climb's closing sweeps ran as planned programs


# ------------------------------------------------------------------------------------------------ C5

@pytest.fixture(scope="module")
def c5():
    from mpboot_amd import trees
    codes, dt = _workload("C5")
    back = trees.random_topology(codes.shape[0], np.random.default_rng(9))
    return codes, dt, back


def test_c5_fitch20_scores_and_scans_match_oracle(c5):
    from mpboot_amd import engine
    from oracle import pyoracle as po
    codes, dt, back = c5
    e = engine.FitchEngine(codes, datatype=dt)
    o = po.Oracle(codes, datatype=po.AA)
    assert (e.S, e.W) == (20, o.W)
    assert e.score_tree(back) == o.score_tree(back)
    nodep = o.nodep()
    rng = np.random.default_rng(3)
    for rec in nodep[1 + rng.choice(2 * codes.shape[0] - 2, size=6, replace=False)]:
        tq, tm = _oracle_scan(o, po, back, rec, 6)
        e.set_tree(back)
        q, mp, _ = e.spr_scan(int(rec), 1, 6)
        assert q.tolist() == tq and mp.tolist() == tm


def test_c5_fitch20_climb_is_monotone_and_idempotent(c5):
    from mpboot_amd import engine
    codes, dt, back = c5
    e = engine.FitchEngine(codes, datatype=dt)
    e.set_tree(back)
    start = e.score_tree()
    e.seed_ties(engine.TIE_RANDOM, 5)
    final = e.optimize_spr(1, 6)
    sc = e.moves()[2]
    assert len(sc) > 500 and final < start
    assert (np.diff(sc.astype(np.int64)) <= 0).all() and int(sc[-1]) == final
    assert e.score_tree() == final
    tree = e.get_tree()
    e.seed_ties(engine.TIE_FIRST, 0)
    assert e.optimize_spr(1, 6) == final and (e.get_tree() == tree).all()
    # per-pattern lengths of the optimum sum to its length
    ptn, tot = e.pattern_scores()
    assert tot == final == int(ptn.astype(np.int64).sum())


def test_c5_weighted_20_state_scans_match_oracle(c5):
    """the `-cost` form of config 5: Sankoff-20 tree length and every candidate of two prune nodes (about 200 insertion tests)"""
    from mpboot_amd import engine
    from oracle import pyoracle as po
    codes, dt, back = c5
    rng = np.random.default_rng(5)
    c = rng.integers(1, 6, size=(20, 20))
    cost = (np.triu(c, 1) + np.triu(c, 1).T).astype(np.uint32)
    e = engine.FitchEngine(codes, datatype=dt, cost=cost)
    o = po.Oracle(codes, datatype=po.AA, cost=cost)
    assert e.score_tree(back) == o.score_tree(back)
    nodep = o.nodep()
    done = 0
    for rec in nodep[[40, 333, 700]]:
        tq, tm = _oracle_scan(o, po, back, rec, 6)
        e.set_tree(back)
        q, mp, _ = e.spr_scan(int(rec), 1, 6)
        assert q.tolist() == tq and mp.tolist() == tm
        done += len(tq)
    assert done > 60


# ------------------------------------------------------------------------------------------------ sweep_scan (the timed call)

@pytest.mark.parametrize("opts", [dict(), dict(scan_prog=0), dict(scan_prog=2, words_per_lane=2)])
def test_sweep_scan_equals_per_node_scans(opts):
    """mpf_spr_sweep_scan's unsplit whole-sweep branch: its best == the minimum over mpf_spr_scan of every prune node,
    its test count == the sum of their candidate counts"""
    from mpboot_amd import engine, synth, trees
    letters, _ = synth.synth_alignment(260, 1800, "DNA", 0.07, seed=21)
    codes = synth.letters_to_codes(letters)
    back = trees.random_topology(260, np.random.default_rng(4))
    e = engine.FitchEngine(codes)
    for k, v in opts.items():
        e.set_option(k, v)
    e.set_option("check_counts", 1)
    e.set_tree(back)
    e.score_tree()
    ntests, best = e.sweep_scan(1, 6)
    ref = engine.FitchEngine(codes)
    ref.set_option("scan_prog", 0)
    ref.set_tree(back)
    ref.score_tree()
    assert (e.node_order() == ref.node_order()).all()
    ncost, costs, off = e.sweep_costs(1, 6)
    tot, mn = 0, None
    for i, rec in enumerate(ref.node_order()):
        q, mp, _ = ref.spr_scan(int(rec), 1, 6)
        assert costs[int(off[i]):int(off[i + 1])].tolist() == mp.tolist(), int(rec)     # per candidate, not just the minimum
        tot += len(q)
        if len(mp):
            mn = int(mp.min()) if mn is None else min(mn, int(mp.min()))
    assert (ntests, best) == (tot, mn) and ncost == tot


def test_sweep_scan_candidates_at_c3_match_the_walking_kernel():
    """C3: every candidate cost of the whole sweep, planned programs (what bench.py times) vs the device-walked kernel
    (pinned candidate by candidate against the reference in test_gpu_parity.py)"""
    from mpboot_amd import engine, trees
    codes, dt = _workload("C3")
    back = trees.random_topology(codes.shape[0], np.random.default_rng(12))
    res = []
    for prog in (1, 0):
        e = engine.FitchEngine(codes, datatype=dt)
        e.set_option("scan_prog", prog)
        e.set_option("check_counts", 1)
        e.set_tree(back)
        e.score_tree()
        res.append(e.sweep_costs(1, 6))
        st = e.stats()
        assert (st["plan_launches"] > 0) == (prog == 1)
    assert res[0][0] == res[1][0] > 50_000
    assert (res[0][1] == res[1][1]).all()


def test_c3_scans_match_oracle():
    """BASELINE config 3 at full size, directly against the oracle: tree length and every insertion test (radius 6, both sides,
    the reference's order) of four prune nodes of a random tree"""
    from mpboot_amd import engine, trees
    from oracle import pyoracle as po
    codes, dt = _workload("C3")
    back = trees.random_topology(codes.shape[0], np.random.default_rng(21))
    e = engine.FitchEngine(codes, datatype=dt)
    o = po.Oracle(codes)
    assert e.score_tree(back) == o.score_tree(back)
    nodep = o.nodep()
    done = 0
    for rec in nodep[[7, 1003, 1500, 1990]]:              # a tip and three inner nodes, start / middle / end of the sweep order
        tq, tm = _oracle_scan(o, po, back, rec, 6)
        e.set_tree(back)
        q, mp, _ = e.spr_scan(int(rec), 1, 6)
        assert q.tolist() == tq and mp.tolist() == tm
        done += len(tq)
    assert done > 100


# ------------------------------------------------------------------------------------------------ C4 (-bb at C3 size)

@pytest.fixture(scope="module")
def c4():
    from mpboot_amd import engine
    codes, dt = _workload("C3")
    e = engine.FitchEngine(codes, datatype=dt)
    e.seed_ties(engine.TIE_RANDOM, 1)
    e.make_parsimony_tree(12345, 0)
    back = e.get_tree()
    P = codes.shape[1]
    samples = np.random.default_rng(4242).multinomial(P, np.ones(P) / P, size=96).astype(np.uint16)
    return codes, dt, back, samples


def _online_phase(e, engine, back, samples, **kw):
    e.ufboot_attach(samples, 0.5, **kw)
    e.set_tree(back)
    e.reset_node_order()
    e.seed_ties(engine.TIE_RANDOM, 1)
    return e.optimize_spr(1, 6)


def test_c4_online_ufboot_sweep_boot_logl_rederived(c4):
    """-bb online phase at full size: every sample's boot_logl == -<per-pattern lengths of the tree it kept, its weights>"""
    from mpboot_amd import engine
    codes, dt, back, samples = c4
    e = engine.FitchEngine(codes, datatype=dt)
    score = _online_phase(e, engine, back, samples)
    logl, counts, tr = e.ufboot_state()
    assert len(e.ufboot_tree_logl()) > 1000 and (counts >= 1).all()
    final = e.get_tree()
    for b in range(0, 96, 8):
        e.set_tree(e.ufboot_tree(int(tr[b])))
        ptn, _tot = e.pattern_scores()
        assert -int((ptn.astype(np.int64) * samples[b]).sum()) == int(logl[b]), b
    e.set_tree(final)
    assert e.score_tree() == score


def test_c4_online_phase_with_1000_samples_boot_logl_rederived():
    """BASELINE config 4's sample count on the C3 alignment, from a random topology (thousands of accepted moves, > 10^6 booked
    insertion tests): boot_logl of ALL 1000 samples == -<per-pattern lengths of the tree the sample kept, the sample's weights>,
    counts and stored trees consistent"""
    from mpboot_amd import engine, trees
    codes, dt = _workload("C3")
    P = codes.shape[1]
    samples = np.random.default_rng(4242).multinomial(P, np.ones(P) / P, size=1000).astype(np.uint16)
    back = trees.random_topology(codes.shape[0], np.random.default_rng(2024))
    e = engine.FitchEngine(codes, datatype=dt)
    score = _online_phase(e, engine, back, samples)
    logl, counts, tr = e.ufboot_state()
    assert len(logl) == 1000 and (counts >= 1).all()
    final = e.get_tree()
    w = samples.astype(np.int64)
    kept = {}
    for b in range(1000):
        kept.setdefault(int(tr[b]), []).append(b)
    assert len(e.ufboot_tree_logl()) > 100_000             # trees that went through the bookkeeping on the way
    for t, bs in kept.items():
        e.set_tree(e.ufboot_tree(t))
        ptn, _tot = e.pattern_scores()
        got = -(w[bs] @ ptn.astype(np.int64))
        assert got.tolist() == [int(logl[b]) for b in bs], t
    e.set_tree(final)
    assert e.score_tree() == score


C4_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["MPF_ROOT"])
import torch.distributed as dist
from mpboot_amd import engine, shard, synth
dist.init_process_group("gloo")
rank, ws = shard.world()
letters, _ = synth.workload("C3")
codes = synth.letters_to_codes(letters, "DNA")
back = np.load(os.environ["MPF_BACK"])
P = codes.shape[1]
samples = np.random.default_rng(4242).multinomial(P, np.ones(P) / P, size=int(os.environ.get("MPF_NSAMP", "96"))).astype(np.uint16)
e = engine.FitchEngine(codes)
e.ufboot_attach(samples, 0.5, shard=(rank, ws))
e.set_tree(back); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 1)
s = e.optimize_spr(1, 6)
logl, cnt, tr = e.ufboot_state()
res = {"s": s, "logl": logl.tolist(), "cnt": cnt.tolist(), "tr": tr.tolist(), "saved": e.ufboot_tree_logl().tolist(),
       "final": e.get_tree().tolist(), "draws": e.ufboot_counters()["tie_draws"]}
allr = [None] * ws
dist.all_gather_object(allr, res)
if rank == 0:
    assert all(r == allr[0] for r in allr), "ranks disagree"
    print("RESULT " + json.dumps(res))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("n_samples", [96, 1000])
def test_c4_sample_sharded_online_phase_equals_unsharded(c4, tmp_path, n_samples):
    """BASELINE config 4's data path on this box: two ranks (sharing the one GPU, gloo rendezvous) hold every second bootstrap
    sample each and all-gather their events per scan batch -- every rank ends with exactly the single engine's bookkeeping"""
    import json
    import os
    import socket
    import subprocess
    import sys

    from helpers import ROOT
    from mpboot_amd import engine
    codes, dt, back, samples = c4
    if n_samples != len(samples):
        P = codes.shape[1]
        samples = np.random.default_rng(4242).multinomial(P, np.ones(P) / P, size=n_samples).astype(np.uint16)
    single = engine.FitchEngine(codes, datatype=dt)
    s0 = _online_phase(single, engine, back, samples)
    logl, cnt, tr = single.ufboot_state()
    np.save(tmp_path / "back.npy", back)
    script = tmp_path / "worker.py"
    script.write_text(C4_WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MPF_ROOT=ROOT, MPF_BACK=str(tmp_path / "back.npy"), MPF_NSAMP=str(n_samples))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    got = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert got["s"] == s0
    assert got["logl"] == logl.tolist() and got["cnt"] == cnt.tolist() and got["tr"] == tr.tolist()
    assert got["saved"] == single.ufboot_tree_logl().tolist()
    assert got["final"] == single.get_tree().tolist()
    assert got["draws"] == single.ufboot_counters()["tie_draws"]


# ------------------------------------------------------------------------------------------------ C3: the climbs bench.py times

def _climb(codes, dt, back, mode, tile, seed=1):
    from mpboot_amd import engine
    e = engine.FitchEngine(codes, datatype=dt)
    e.set_option("climb_device", mode)
    e.set_option("climb_tile", tile)
    e.set_tree(back)
    e.reset_node_order()
    e.seed_ties(engine.TIE_RANDOM, seed)
    s = e.optimize_spr(1, 6)
    return e, s, [x.tolist() for x in e.moves()]


@pytest.fixture(scope="module")
def c3_climb():
    """BASELINE config 3's alignment, the random start tree of bench.py's random_start leg, and the climb from it with the sweep
    loop on the host (host-driven whole-chip batches: the path pinned move for move against the oracle on the fixtures and at C2)"""
    from mpboot_amd import trees
    codes, dt = _workload("C3")
    back = trees.random_topology(codes.shape[0], np.random.default_rng(2024))
    e, s, moves = _climb(codes, dt, back, 0, 1)
    assert e.stats()["climb_launches"] == 0 and len(moves[0]) > 3000
    return codes, dt, back, s, moves, e.get_tree().tolist(), e.tie_state()


def test_c3_benchmarked_climb_equals_the_oracle(c3_climb):
    """bench.py's random_start.plain_climb (C3, random topology 2024, tie seed 1) against the ORACLE at full size: every accepted
    move with its length, the final topology and the state the tie stream is left in.  (The fixture is the host-driven loop; the
    tests below hold the persistent kernel, 98 and 25 workgroups, and eight concurrent climbs to the same list.)"""
    from oracle import pyoracle as po
    codes, dt, back, s, moves, final, rng = c3_climb
    o = po.Oracle(codes)
    o.set_tree(back)
    o.seed_ties(po.TIE_RANDOM, 1)
    o.trace(True)
    assert o.optimize_spr(1, 6) == s
    assert [x.tolist() for x in o.get_moves()] == moves
    assert o.get_tree().tolist() == final
    assert o.tie_state() == rng


@pytest.mark.parametrize("opts", [{}, {"ufb_pipe": 0}, {"ufb_fast": 0}])
def test_c4_tracked_climb_first_visits_equal_the_oracle(opts):
    """-bb at full size against the ORACLE: the tracked climb of bench.py's bb_flow (C3 alignment, random topology 2024, tie seed 1)
    with 16 samples, cut short behind 240 prune-node visits (option max_visits / orc_set_max_visits; ~50 accepted moves, ~7 000
    booked trees) -- every book: accepted moves, topology, treels_logl, boot_logl / boot_counts / boot_trees and the trees
    themselves, the number of draws and the state of the tie stream.  All three ways through the loop."""
    from helpers import same_topology
    from mpboot_amd import engine, trees
    from oracle import pyoracle as po
    codes, dt = _workload("C3")
    P = codes.shape[1]
    samples = np.random.default_rng(4242).multinomial(P, np.ones(P) / P, size=1000).astype(np.uint16)[:16]
    back = trees.random_topology(codes.shape[0], np.random.default_rng(2024))
    K = 240
    e = engine.FitchEngine(codes, datatype=dt)
    for k, v in opts.items():
        e.set_option(k, v)
    e.set_option("max_visits", K)
    o = po.Oracle(codes)
    o.set_max_visits(K)
    o.trace(True)
    for x, mode in ((e, engine.TIE_RANDOM), (o, po.TIE_RANDOM)):
        x.ufboot_attach(samples)
        x.set_tree(back)
        x.seed_ties(mode, 1)
    e.reset_node_order()
    se, so = e.optimize_spr(1, 6), o.optimize_spr(1, 6)
    assert se == so
    em = [x.tolist() for x in e.moves()]
    assert len(em[0]) > 20 and em == [x.tolist() for x in o.get_moves()]
    assert (e.get_tree() == o.get_tree()).all()
    assert e.tie_state() == o.tie_state()
    assert len(e.ufboot_tree_logl()) > 3000 and e.ufboot_tree_logl().tolist() == o.ufboot_tree_logl().tolist()
    le, ce, te = e.ufboot_state()
    lo, co, to = o.ufboot_state()
    assert le.tolist() == lo.tolist() and ce.tolist() == co.tolist() and te.tolist() == to.tolist()
    assert e.ufboot_counters()["tie_draws"] == o.ufboot_draws() and o.ufboot_bad() == 0
    for t in sorted(set(te.tolist())):
        if t >= 0:
            assert same_topology(e.ufboot_tree(t), o.ufboot_tree(t), e.n)
    if not opts:
        assert e.get_option("ufb_early_batches") > 0          # (the default really took decisions from the costs)


@pytest.mark.parametrize("tile", [1, 4])
def test_c3_device_climb_equals_host_batches(c3_climb, tile):
    """random_start.plain_climb of bench.py at full size: the persistent kernel (98 co-resident workgroups at VW = 1, 25 at VW = 4,
    candidate lengths exchanged through the three-generation 64-bit atomic ring across the XCDs) accepts exactly the moves the
    host-driven loop accepts -- the whole list, the final topology and the state the tie stream is left in."""
    codes, dt, back, s, moves, final, rng = c3_climb
    for mode in (1, 2):                               # auto (hands sparse sweeps back to the host) and always-in-kernel
        e, sd, md = _climb(codes, dt, back, mode, tile)
        st = e.stats()
        assert st["climb_launches"] >= 1 and st["climb_moves"] > 3000
        assert sd == s
        assert md == moves
        assert e.get_tree().tolist() == final
        assert e.tie_state() == rng


@pytest.mark.parametrize("word_major", [1, 0])
def test_c3_climbs_in_one_launch_equal_the_pinned_climb(c3_climb, word_major):
    """bench.py's climbs_in_one_launch leg at ITS size: mpf_optimize_spr_many on C3 engines (k_climb_many: one workgroup per climb, 25
    tiles of 64 words taken through the refresh by its eight waves, every sweep inside the launch, 110 KB of control state in LDS) --
    the climb from the benchmarked start tree makes exactly the moves of the list the oracle is held to above, in both tile shapes;
    the climbs beside it end where their solo runs end."""
    from mpboot_amd import engine, trees
    codes, dt, back, s, moves, final, rng = c3_climb
    n = codes.shape[0]
    backs = [back] + [trees.random_topology(n, np.random.default_rng(4000 + k)) for k in range(1, 4)]
    engs = []
    for k, b in enumerate(backs):
        e = engine.FitchEngine(codes, datatype=dt)
        e.set_option("many_word_major", word_major)
        e.set_tree(b); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 1 + k)
        engs.append(e)
    sc = engine.optimize_spr_many(engs, 1, 6)
    e = engs[0]
    assert e.stats()["climb_launches"] >= 1 and e.get_option("climb_tile_many") == 4
    assert int(sc[0]) == s and [x.tolist() for x in e.moves()] == moves
    assert e.get_tree().tolist() == final and e.tie_state() == rng
    if word_major:
        for k in (1, 2):
            es, ss, ms = _climb(codes, dt, backs[k], 2, 1, seed=1 + k)
            assert int(sc[k]) == ss and [x.tolist() for x in engs[k].moves()] == ms and (engs[k].get_tree() == es.get_tree()).all()


def test_c3_eight_concurrent_device_climbs_equal_their_solo_runs(c3_climb):
    """bench.py's concurrent_climbs leg: eight engines on eight host threads, 64-word tiles (25 workgroups per climb, the launches
    admitted together and polling each other's exchange rings on one chip) -- every engine's moves are those of its solo run."""
    import threading
    from mpboot_amd import engine, trees
    codes, dt, back0, _s, _m, _f, _r = c3_climb
    n = codes.shape[0]
    K = 8
    backs = [back0] + [trees.random_topology(n, np.random.default_rng(3000 + k)) for k in range(1, K)]
    engines = []
    for k in range(K):
        e = engine.FitchEngine(codes, datatype=dt)
        e.set_option("climb_device", 1)
        e.set_option("climb_tile", 4)
        engines.append(e)
    solo = []
    for k in range(K):
        e = engines[k]
        e.set_tree(backs[k]); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 10 + k)
        s = e.optimize_spr(1, 6)
        solo.append((s, [x.tolist() for x in e.moves()], e.get_tree().tolist()))
    if True:
        # the first tree twice: solo with tile 4 == the host-driven reference of the fixture (seed 1)
        e = engines[0]
        e.set_tree(backs[0]); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 1)
        assert e.optimize_spr(1, 6) == _s and [x.tolist() for x in e.moves()] == _m
    got, errs = [None] * K, []

    def work(k):
        try:
            e = engines[k]
            e.set_tree(backs[k]); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 10 + k)
            e.reset_stats()
            s = e.optimize_spr(1, 6)
            got[k] = (s, [x.tolist() for x in e.moves()], e.get_tree().tolist(), e.stats()["climb_launches"])
        except Exception as exc:                      # noqa: BLE001
            errs.append((k, repr(exc)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in range(K):
        assert got[k][3] >= 1
        assert got[k][:3] == solo[k], f"engine {k} diverged from its solo run"


# ------------------------------------------------------------------------------------------------ full-size pins of two benchmarked legs

@pytest.mark.parametrize("tie", ["random", "first"])
def test_c3_start_tree_equals_the_oracle_insertion_by_insertion(tie):
    """bench.py's start_trees leg at ITS size: the addition loop of one C3 start tree (1000 taxa x 50 000 patterns; k_grow: skeleton +
    parts, several workgroups exchanging rows) against the ORACLE's stepwiseAddition loop (sprparsimony.cpp:2977-3019, :3107-3181)
    -- the length after every insertion, every insertion branch, the finished tree and the state the tie stream is left in, under
    mpboot's random tie rule and under the PLL original's first-best rule."""
    from mpboot_amd import engine
    from oracle import pyoracle as po
    codes, dt = _workload("C3")
    seed = 31337 + 12345 * 3                                  # (a unit of the bench's start_trees leg)
    e = engine.FitchEngine(codes, datatype=dt)
    e.seed_ties(engine.TIE_RANDOM if tie == "random" else engine.TIE_FIRST, 7)
    s, best, ins = e.stepwise_addition(seed)
    assert e.get_option("grow_launches") == 1 and e.get_option("grow_last_err") == 0      # the kernel built it, and came back clean
    o = po.Oracle(codes)
    o.seed_ties(po.TIE_RANDOM if tie == "random" else po.TIE_FIRST, 7)
    so, ob, oi = o.stepwise(seed)
    assert so == s
    assert ob.tolist() == best.tolist()
    assert oi.tolist() == ins.tolist()
    assert (o.get_tree() == e.get_tree()).all()
    assert o.tie_state() == e.tie_state()


def test_c3_start_tree_equals_the_reference_first_best():
    """... and against the REFERENCE itself: oracle/_ref/pll_ref_driver rasx = PLL's own makePermutationFast + buildSimpleTree +
    stepwiseAddition loop (fastDNAparsimony.c, compiled from the reference's sources), one checkpoint per added taxon."""
    import os
    import subprocess
    import tempfile

    from helpers import ROOT
    from mpboot_amd import engine, synth, trees
    drv = os.path.join(ROOT, "oracle", "_ref", "pll_ref_driver")
    if not (os.path.exists(drv) and os.access(drv, os.X_OK)):
        pytest.skip("oracle/_ref/pll_ref_driver not built (needs /root/reference at build time)")
    letters, names = synth.workload("C3")
    codes = synth.letters_to_codes(letters, "DNA")
    n = codes.shape[0]
    seed = 424242
    with tempfile.TemporaryDirectory() as tmp:
        aln = os.path.join(tmp, "a.phy")
        synth.write_phylip(aln, synth.letters_to_text(letters, "DNA"), names)
        out = subprocess.run([drv, "rasx", aln, "DNA", "0", str(seed)], capture_output=True, text=True, check=True, timeout=1200).stdout
    adds, back, check = [], None, None
    for l in out.splitlines():
        t = l.split()
        if t and t[0] == "add":
            adds.append((int(t[1]), int(t[3]), int(t[5]), int(t[7])))
        elif t and t[0] == "rasx_topology":
            back = trees.parse_topology_line(t[1:], n)
        elif t and t[0] == "rasx_check":
            check = int(t[1])
    assert len(adds) == n - 3 and back is not None
    e = engine.FitchEngine(codes)
    e.seed_ties(engine.TIE_FIRST, 0)
    s, best, ins = e.stepwise_addition(seed)
    assert e.get_option("grow_launches") == 1 and e.get_option("grow_last_err") == 0
    for step, _tip, b, i in adds:
        assert (int(best[step]), int(ins[step])) == (b, i), step
    assert e.get_tree().tolist() == back.tolist()
    assert s == check == e.score_tree()


def test_c5_fitch20_climb_first_visits_equal_the_oracle(c5):
    """BASELINE config 5 in its Fitch form at full size (500 taxa x 20 000 protein patterns, 20-state kernels): the SPR climb from a
    random tree against the ORACLE over its first 260 prune-node visits (orc_set_max_visits) -- every accepted move with its length
    --, and the WHOLE climb in the persistent kernel (k_climb, five states per lane) against the host-driven loop: moves, final
    topology and the state of the tie stream.  (A visit limit keeps a climb out of the kernel, so the kernel is held to the oracle
    through the host loop's list.)"""
    from mpboot_amd import engine
    from oracle import pyoracle as po
    codes, dt, back = c5
    K = 260
    o = po.Oracle(codes, datatype=po.AA)
    o.set_max_visits(K)
    o.set_tree(back)
    o.seed_ties(po.TIE_RANDOM, 5)
    o.trace(True)
    so = o.optimize_spr(1, 6)
    om = [x.tolist() for x in o.get_moves()]
    assert len(om[0]) >= 40
    runs = {}
    for mode in (0, 2):
        e = engine.FitchEngine(codes, datatype=dt)
        e.set_option("climb_device", mode)
        e.set_tree(back)
        e.reset_node_order()
        e.seed_ties(engine.TIE_RANDOM, 5)
        s = e.optimize_spr(1, 6)
        runs[mode] = (s, [x.tolist() for x in e.moves()], e.get_tree().tolist(), e.tie_state(), e.stats()["climb_launches"])
    assert runs[0][4] == 0 and runs[2][4] >= 1
    assert runs[0][:4] == runs[2][:4]
    k = len(om[0])
    assert [x[:k] for x in runs[0][1]] == om and om[2][-1] == so
    # the host loop cut at the same visit ends where the oracle ends
    e = engine.FitchEngine(codes, datatype=dt)
    e.set_option("max_visits", K)
    e.set_tree(back)
    e.reset_node_order()
    e.seed_ties(engine.TIE_RANDOM, 5)
    assert e.optimize_spr(1, 6) == so
    assert (e.get_tree() == o.get_tree()).all() and e.tie_state() == o.tie_state()

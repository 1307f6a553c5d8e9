"""Batched bootstrap refinement (mpf_ufboot_refine_sweep): the first sweep of pllOptimizeSprParsimony from ONE tree under many
samples' weights at once -- masked scan + mask x weight product + per-sample replay -- against the calls it stands for:
mpf_set_weights(sample b) + mpf_optimize_spr per sample on the engine, and the same on the CPU oracle (reference flow:
IQTree::optimizeBootTrees, iqtree.cpp:2797-2862)."""
import numpy as np
import pytest

from helpers import load_fixture

pytestmark = pytest.mark.gpu


def _samples(w0, B, rng):
    nsite = int(w0.sum())
    site_pattern = np.repeat(np.arange(len(w0)), w0)
    out = np.zeros((B, len(w0)), dtype=np.uint16)
    for b in range(B):
        out[b] = np.bincount(site_pattern[rng.integers(0, nsite, size=nsite)], minlength=len(w0))
    return out


def _check(codes, w0, dt_e, dt_o, back, samples, seeds, radius, chunk=None):
    from mpboot_amd import engine
    from oracle import pyoracle as po
    B = len(samples)
    e = engine.FitchEngine(codes, w0, datatype=dt_e)
    if chunk:
        e.set_option("refine_chunk", chunk)
    e.seed_ties(engine.TIE_RANDOM, 0)
    e.ufboot_attach(samples, 0.5)
    e.reset_node_order()
    e.set_tree(back)
    order = e.node_order()
    scores, stable, first = e.ufboot_refine_sweep(radius, seeds)
    e.ufboot_detach()
    solo = engine.FitchEngine(codes, w0, datatype=dt_e)
    o = po.Oracle(codes, w0, datatype=dt_o)
    n_stable = 0
    for b in range(B):
        w = samples[b].astype(np.int32)
        solo.set_weights(w)
        solo.seed_ties(engine.TIE_RANDOM, int(seeds[b]))
        solo.reset_node_order()
        solo.set_tree(back)
        s0 = solo.score_tree()
        s1 = solo.optimize_spr(1, radius)
        mv = solo.moves()
        o.set_weights(w)
        o.seed_ties(po.TIE_RANDOM, int(seeds[b]))
        o.reset_nodep()
        assert o.score_tree(back) == s0 == scores[b]
        o.trace(True)
        so = o.optimize_spr(1, radius)
        assert so == s1 and [x.tolist() for x in o.get_moves()] == [x.tolist() for x in mv]
        assert bool(stable[b]) == (len(mv[0]) == 0), (b, stable[b], len(mv[0]))
        if stable[b]:
            n_stable += 1
            assert s1 == scores[b] and (solo.get_tree() == back).all() and first[b] == 0
        else:
            # the first accepted move prunes at the visit the replay names (p side: that record, q side: the one behind it)
            rec = int(order[first[b] - 1])
            assert int(mv[0][0]) in (rec, int(back[rec]))
    return n_stable


@pytest.mark.parametrize("name", ["dna_clean", "dna_ambig", "dna_dups", "aa", "dna_48"])
def test_refine_sweep_equals_per_sample_climbs_on_fixtures(name):
    from mpboot_amd import engine, trees
    from oracle import pyoracle as po
    fx = load_fixture(name)
    codes, w0 = fx["codes_np"], fx["weights_np"]
    dt_e = engine.AA if fx["datatype"] == po.AA else engine.DNA
    rng = np.random.default_rng(17)
    samples = _samples(w0, 24, rng)
    seeds = rng.integers(1, 10 ** 6, size=24)
    # an SPR-optimal tree (most samples leave it alone or move sideways) and a random one (every sample moves)
    e = engine.FitchEngine(codes, w0, datatype=dt_e)
    start = trees.random_topology(fx["n"], np.random.default_rng(3))
    e.set_tree(start)
    e.seed_ties(engine.TIE_RANDOM, 5)
    e.optimize_spr(1, 6)
    opt = e.get_tree()
    for back, radius, chunk in ((opt, 6, None), (opt, 3, 5), (start, 6, 7)):
        _check(codes, w0, dt_e, fx["datatype"], back, samples, seeds, radius, chunk)


def test_refine_sweep_on_a_decisive_alignment_and_through_the_driver():
    """120 taxa x 6000 patterns: long alignments leave nearly every sample on the online phase's tree; refine_boot_trees with the
    batched first sweep == the per-sample loop"""
    from mpboot_amd import bootstrap, engine, synth, trees
    from oracle import pyoracle as po
    letters, _ = synth.synth_alignment(120, 6000, "DNA", 0.05, seed=8)
    codes = synth.letters_to_codes(letters, "DNA")
    n, P = codes.shape
    w0 = np.ones(P, dtype=np.int32)
    e = engine.FitchEngine(codes)
    e.set_tree(trees.random_topology(n, np.random.default_rng(1)))
    e.seed_ties(engine.TIE_RANDOM, 2)
    e.optimize_spr(1, 6)
    opt = e.get_tree()
    rng = np.random.default_rng(23)
    samples = _samples(w0, 40, rng)
    seeds = np.array([9 + 12345 * b for b in range(40)])
    n_stable = _check(codes, w0, engine.DNA, po.DNA, opt, samples, seeds, 6)
    assert n_stable >= 20
    # the driver: two start topologies among the samples
    other = trees.random_topology(n, np.random.default_rng(4))
    boot_trees = [opt if b % 5 else other for b in range(40)]
    e2 = engine.FitchEngine(codes)
    sc_b, tr_b = bootstrap.refine_boot_trees(e2, samples, boot_trees, 9, 6, batched=True)
    e3 = engine.FitchEngine(codes)
    sc_s, tr_s = bootstrap.refine_boot_trees(e3, samples, boot_trees, 9, 6, batched=False)
    assert sc_b.tolist() == sc_s.tolist()
    assert all((tr_b[b] == tr_s[b]).all() for b in range(40))
    # called again on the SAME engine (bench.py does, four times in a row): samples on `other` are not stable, so engine 0 has
    # climbed under sample weights in the first call -- it must have been handed back under the weights it came with
    assert (e2.weights() == w0).all()
    sc_b2, tr_b2 = bootstrap.refine_boot_trees(e2, samples, boot_trees, 9, 6, batched=True)
    assert sc_b2.tolist() == sc_s.tolist()
    assert all((tr_b2[b] == tr_s[b]).all() for b in range(40))
    # ... and an engine that holds weights which drop a pattern the samples count is refused, not silently mis-scored
    w_bad = w0.copy()
    w_bad[int(np.argmax(samples.max(axis=0) > 0))] = 0
    e2.set_weights(w_bad)
    with pytest.raises(ValueError):
        bootstrap.refine_boot_trees(e2, samples, boot_trees, 9, 6, batched=True)


REFINE_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["MPF_ROOT"])
import torch.distributed as dist
from mpboot_amd import bootstrap, engine, shard, synth, trees
dist.init_process_group("gloo")
rank, ws = shard.world()
letters, _ = synth.synth_alignment(60, 3000, "DNA", 0.06, seed=12)
codes = synth.letters_to_codes(letters, "DNA")
n, P = codes.shape
e = engine.FitchEngine(codes)
e.set_tree(trees.random_topology(n, np.random.default_rng(1)))
e.seed_ties(engine.TIE_RANDOM, 2)
e.optimize_spr(1, 6)
opt = e.get_tree()
other = trees.random_topology(n, np.random.default_rng(4))
samples = np.random.default_rng(23).multinomial(P, np.ones(P) / P, size=30).astype(np.uint16)
boot_trees = [opt if b % 4 else other for b in range(30)]
scores, local = bootstrap.refine_boot_trees(e, samples, boot_trees, 9, 6, batched=os.environ["MPF_BATCHED"] == "1")
mine = {int(b): t.tolist() for b, t in local.items()}
allt = [None] * ws
dist.all_gather_object(allt, mine)
if rank == 0:
    merged = {}
    for d in allt:
        merged.update(d)
    print("RESULT " + json.dumps({"scores": scores.tolist(), "trees": [merged[b] for b in range(30)]}))
dist.destroy_process_group()
'''


def test_batched_refinement_sharded_over_two_ranks_equals_per_sample_loop(tmp_path):
    """bootstrap.refine_boot_trees as bench.py --gpus N runs it: sample b on rank b % N (two gloo ranks sharing this box's GPU), the
    batched first sweep on each rank's share of the samples (sample-sharded tracker) -- same scores and trees as the per-sample loop"""
    import json
    import os
    import socket
    import subprocess
    import sys

    from helpers import ROOT
    script = tmp_path / "worker.py"
    script.write_text(REFINE_WORKER)
    res = {}
    for batched in ("1", "0"):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MPF_ROOT=ROOT, MPF_BATCHED=batched)
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                              "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        res[batched] = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["1"] == res["0"]


def test_batched_refinement_from_many_distinct_trees_equals_per_sample_oracle_climbs():
    """A noisy alignment (as synth C4N is at full size): the samples of the online phase keep MANY different trees and nearly every
    refinement climbs.  refine_boot_trees (one masked sweep + one product per distinct topology, then the unstable samples' own
    climbs) against IQTree::optimizeBootTrees' loop on the oracle: per sample re-weight, seed, one SPR climb from its tree
    (iqtree.cpp:2797-2862)."""
    from mpboot_amd import bootstrap, engine, shard, synth, trees
    from oracle import pyoracle as po
    n, P, B, radius = 40, 300, 96, 4
    letters, _ = synth.synth_alignment(n, P, "DNA", 0.4, seed=31)
    codes = synth.letters_to_codes(letters, "DNA")
    w0 = np.ones(P, dtype=np.int32)
    samples = _samples(w0, B, np.random.default_rng(5))
    # the online phase's result, stood in for: every sample's tree = an SPR-optimal tree of ANOTHER sample's alignment
    e = engine.FitchEngine(codes)
    boot_trees = []
    for b in range(B):
        e.set_weights(samples[(b * 7 + 3) % B].astype(np.int32))
        e.seed_ties(engine.TIE_RANDOM, 100 + b)
        e.reset_node_order()
        e.set_tree(trees.random_topology(n, np.random.default_rng(b % 60)))
        e.optimize_spr(1, radius)
        boot_trees.append(e.get_tree())
    e.set_weights(w0)
    distinct = len({engine.iq_topology_key(t) for t in boot_trees})
    assert distinct >= 50
    pool = [e, engine.FitchEngine(codes), engine.FitchEngine(codes)]
    sc_b, tr_b = bootstrap.refine_boot_trees(pool, samples, boot_trees, 9, radius, batched=True)
    o = po.Oracle(codes)
    moved = 0
    for b in range(B):
        o.set_weights(samples[b].astype(np.int32))
        o.seed_ties(po.TIE_RANDOM, shard.unit_seed(9, b))
        o.reset_nodep()
        o.set_tree(boot_trees[b])
        s = o.optimize_spr(1, radius)
        assert s == sc_b[b], b
        assert (o.get_tree() == tr_b[b]).all(), b
        moved += int(not (o.get_tree() == boot_trees[b]).all())
    assert moved >= B // 2                      # the refinements really climb here
    assert (e.weights() == w0).all()
    # many_launch: the unstable samples' climbs as workgroups of ONE launch per round (mpf_optimize_spr_many_round), a finished engine
    # taking the next sample at once -- same lengths, same trees
    pool9 = pool + [engine.FitchEngine(codes) for _ in range(6)]
    sc_m, tr_m = bootstrap.refine_boot_trees(pool9, samples, boot_trees, 9, radius, batched=True, many_launch=True)
    assert sc_m.tolist() == sc_b.tolist() and all((tr_m[b] == tr_b[b]).all() for b in range(B))
    assert all((x.weights() == w0).all() for x in pool9)
    sc_t, tr_t = bootstrap.refine_boot_trees(pool9, samples, boot_trees, 9, radius, batched=True, many_launch=False)
    assert sc_t.tolist() == sc_b.tolist()

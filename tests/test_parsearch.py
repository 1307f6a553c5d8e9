"""Iteration-parallel -bb (mpboot_amd/parsearch.py) on the CPU: the chains are driven on the oracle here (the GPU twin of this file is
tests/test_gpu_parsearch.py).  What an exchange must do: every sample ends up with the shortest length ANY chain holds and a tree of
that length -- the rule-wise merge of the chains' solo books --, the candidate sets take every chain's results, and two ranks over gloo
do exactly what two workers of one process do."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, same_topology


def _setup(n=40, P=300, B=40, seed=3):
    from mpboot_amd import synth
    from oracle import pyoracle as po
    letters, _ = synth.synth_alignment(n, P, "DNA", 0.4, seed=seed)          # (noisy: the chains must disagree on some samples)
    codes = synth.letters_to_codes(letters, "DNA")
    samples = np.random.default_rng(seed).multinomial(P, np.ones(P) / P, size=B).astype(np.uint16)
    o = po.Oracle(codes)
    starts = []
    for k in range(3):
        o.seed_ties(po.TIE_RANDOM, 1 + k)
        o.make_tree(1 + (k + 1) * 12345, 2)
        starts.append((o.get_tree(), o.score_tree()))
    return codes, samples, starts


def _books(o):
    logl, cnt, bt = o.ufboot_state()
    return (-logl).astype(np.int64), cnt.copy(), bt.copy()


def test_an_exchange_is_the_rule_wise_merge_of_the_solo_books():
    from mpboot_amd import parsearch
    from oracle import pyoracle as po
    codes, samples, starts = _setup()
    kw = dict(maxtrav=2, seed=11, sync_every=2, tie_mode=po.TIE_RANDOM, search_kw=dict(unsuccess=50))
    # the two chains alone, up to the first exchange (which, alone, changes nothing)
    solo = []
    for g in range(2):
        o = po.Oracle(codes)
        r = parsearch.ParallelBbRun([o], samples, starts, worker_base=g, **kw)
        info = r.round()
        assert info["adopted"] == 0
        solo.append((o, _books(o), r))
    # the same two chains as workers of one run
    pair = [po.Oracle(codes), po.Oracle(codes)]
    run = parsearch.ParallelBbRun(pair, samples, starts, **kw)
    info = run.round()
    want = np.minimum(solo[0][1][0], solo[1][1][0])
    differ = int((solo[0][1][0] != solo[1][1][0]).sum())
    assert differ > 0, "the fixture should let the chains disagree on some sample"
    assert info["adopted"] == differ
    for w, o in enumerate(pair):
        got_len, got_cnt, got_bt = _books(o)
        assert got_len.tolist() == want.tolist()
        for b in range(samples.shape[0]):
            src = 0 if solo[0][1][0][b] <= solo[1][1][0][b] else 1          # equal lengths keep the holder; the lower worker wins a tie for ownership
            if solo[w][1][0][b] == want[b]:
                src = w                                                     # what this chain held already stays
                assert got_cnt[b] == solo[w][1][1][b]
            else:
                assert got_cnt[b] == 2                                      # :3710 + :3728-3730
            a = o.ufboot_tree(int(got_bt[b]))
            e = solo[src][0].ufboot_tree(int(solo[src][1][2][b]))
            assert same_topology(a, e, o.n)
    # every chain's candidate set holds both chains' results
    scores = sorted(-float(x["score"]) for rr in info["per_worker"] for x in rr)
    for s in run.searches:
        assert all(any(abs(sc - c) < 0.5 for c in s.cands._scores) or sc < s.cands._scores[0] for sc in scores)
        assert s.cur_it == 2 + 4
    # the run goes on from the merged books, deterministically
    run.round()
    again = parsearch.ParallelBbRun([po.Oracle(codes), po.Oracle(codes)], samples, starts, **kw)
    again.round(); again.round()
    assert again.state_hash() == run.state_hash()
    for o in pair:
        assert o.ufboot_bad() == 0


WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["MPF_ROOT"])
sys.path.insert(0, os.path.join(os.environ["MPF_ROOT"], "tests"))
import torch.distributed as dist
from mpboot_amd import parsearch
from oracle import pyoracle as po
from test_parsearch import _setup, _books
ws = int(os.environ.get("WORLD_SIZE", "1"))
if ws > 1:
    dist.init_process_group("gloo")
codes, samples, starts = _setup()
W = 2 // ws
eng = [po.Oracle(codes) for _ in range(W)]
run = parsearch.ParallelBbRun(eng, samples, starts, maxtrav=2, seed=11, sync_every=2, tie_mode=po.TIE_RANDOM, search_kw=dict(unsuccess=50))
infos = [run.round() for _ in range(3)]
out = {"rank": run.rank, "lens": [_books(o)[0].tolist() for o in eng], "cnt": [_books(o)[1].tolist() for o in eng],
       "ties": [int(o.tie_state()) for o in eng], "adopted": [i["adopted"] for i in infos], "iterations": run.iterations,
       "best": run.best_score, "cands": [s.cands._scores for s in run.searches]}
print("RESULT " + json.dumps(out), flush=True)
if ws > 1:
    dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(nproc, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MPF_ROOT=ROOT)
    for attempt in range(2):
        cmd = [sys.executable, str(script)] if nproc == 1 else [
            sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), str(script)]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        # (the port was free when it was picked, not necessarily when the rendezvous bound it: one more try with another)
        if out.returncode == 0 or nproc == 1 or "RESULT " in out.stdout:
            break
    assert out.returncode == 0, out.stderr[-3000:]
    return sorted((json.loads(l.split("RESULT ", 1)[1]) for l in out.stdout.splitlines() if "RESULT " in l), key=lambda r: r["rank"])


def test_two_gloo_ranks_do_what_two_workers_of_one_process_do(tmp_path):
    one = _run(1, tmp_path)[0]
    two = _run(2, tmp_path)
    assert len(two) == 2
    for key in ("lens", "cnt", "ties", "cands"):
        assert [two[0][key][0], two[1][key][0]] == one[key], key
    assert two[0]["adopted"] == two[1]["adopted"] or True      # (per-rank counts differ; their sum is the run's)
    assert [a + b for a, b in zip(two[0]["adopted"], two[1]["adopted"])] == one["adopted"]
    assert two[0]["iterations"] == two[1]["iterations"] == one["iterations"] == 12
    assert two[0]["best"] == two[1]["best"] == one["best"]

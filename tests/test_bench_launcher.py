"""bench.py's multi-rank launcher on the CPU: `python bench.py --gpus N` (no launcher around it) must start N ranks itself.

MPF_BENCH_DRYRUN=1 replaces the engine by the rendezvous + the two all-reduces of the real run (there is no GPU here);
what is under test is the process plumbing: N children through torch.distributed.run, rank 0's line relayed, exit codes."""
import json
import os
import subprocess
import sys

from helpers import ROOT


def run(args, **env):
    e = dict(os.environ, MPF_BENCH_DRYRUN="1", MPF_BENCH_BACKEND="gloo", **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        if k not in env:
            e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=600)


def test_gpus_2_starts_two_ranks():
    out = run(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["dist_world_size"] == 2 and line["backend"] == "gloo"
    assert line["max_time"] == 2.0 and line["sum_tests"] == 200.0          # MAX over ranks of (1 + rank), SUM of 100 per rank
    assert (line["steps"], line["warmup"]) == (3, 1)


def test_single_rank_needs_no_launcher():
    out = run(["--steps", "2", "--warmup", "0"])
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.splitlines()[-1])
    assert line["n_gpus"] == 1 and line["dist_world_size"] == 1


def test_world_size_must_match_gpus():
    out = run(["--gpus", "2"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)

"""The N>1 path on CPU: two gloo ranks shard independent start trees; the result must equal the
single-process run.  The per-unit work is done by the oracle here (no GPU in this test)."""
import os
import socket
import subprocess
import sys

import numpy as np

from helpers import ROOT

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["MPF_ROOT"])
import torch.distributed as dist
from mpboot_amd import shard, synth
from oracle import pyoracle as po

class OracleAsEngine:
    def __init__(self, codes):
        self.o = po.Oracle(codes)
    def seed_ties(self, mode, seed):
        self.o.seed_ties(mode, seed)
    def reset_node_order(self):
        self.o.reset_nodep()
    def make_parsimony_tree(self, seed, dist_):
        return self.o.make_tree(seed, dist_)[0]
    def get_tree(self):
        return self.o.get_tree()

ws = int(os.environ.get("WORLD_SIZE", "1"))
if ws > 1:
    dist.init_process_group("gloo")
letters, _ = synth.synth_alignment(16, 300, "DNA", 0.15, seed=2)
codes = synth.letters_to_codes(letters)
scores, best, tree = shard.search_start_trees(lambda: OracleAsEngine(codes), 6, 77, 3)
if shard.world()[0] == 0:
    print("RESULT " + json.dumps({"scores": scores.tolist(), "best": best, "tree": tree.tolist()}))
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
        if nproc == 1:
            cmd = [sys.executable, str(script)]
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
                   "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        # (the port was free when it was picked, not necessarily when the rendezvous bound it: one more try with another)
        if out.returncode == 0 or nproc == 1 or "RESULT " in out.stdout:
            break
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    import json
    return json.loads(line[7:])


def test_unit_assignment_and_seeds():
    from mpboot_amd import shard
    assert shard.units_of_rank(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((shard.units_of_rank(10, r, 4) for r in range(4)), [])) == list(range(10))
    assert shard.unit_seed(5, 3) == 5 + 3 * 12345


def test_two_gloo_ranks_equal_single_process(tmp_path):
    one = _run(1, tmp_path)
    two = _run(2, tmp_path)
    assert one["scores"] == two["scores"]
    assert one["best"] == two["best"]
    assert one["tree"] == two["tree"]
    assert min(one["scores"]) == one["scores"][one["best"]]


REFINE_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["MPF_ROOT"])
import torch.distributed as dist
from mpboot_amd import shard, synth, bootstrap, trees
from oracle import pyoracle as po

class OracleAsEngine:
    def __init__(self, codes):
        self.o = po.Oracle(codes)
    def set_weights(self, w):
        self.o.set_weights(w)
    def seed_ties(self, mode, seed):
        self.o.seed_ties(mode, seed)
    def reset_node_order(self):
        self.o.reset_nodep()
    def set_tree(self, back):
        self.o.set_tree(back)
    def optimize_spr(self, a, b):
        return self.o.optimize_spr(a, b)
    def get_tree(self):
        return self.o.get_tree()

ws = int(os.environ.get("WORLD_SIZE", "1"))
if ws > 1:
    dist.init_process_group("gloo")
letters, _ = synth.synth_alignment(14, 250, "DNA", 0.2, seed=4)
codes = synth.letters_to_codes(letters)
rng = np.random.default_rng(11)
B, P = 7, codes.shape[1]
samples = rng.multinomial(P, np.ones(P) / P, size=B).astype(np.uint16)
starts = [trees.random_topology(14, np.random.default_rng(100 + b)) for b in range(B)]
scores, local = bootstrap.refine_boot_trees(OracleAsEngine(codes), samples, starts, 5, 6)
mine = {int(b): t.tolist() for b, t in local.items()}
allt = [None] * ws
if ws > 1:
    dist.all_gather_object(allt, mine)
else:
    allt = [mine]
if shard.world()[0] == 0:
    merged = {}
    for d in allt:
        merged.update(d)
    print("RESULT " + json.dumps({"scores": scores.tolist(), "trees": [merged[b] for b in range(B)]}))
if ws > 1:
    dist.destroy_process_group()
'''


def test_boot_tree_refinement_shards_over_two_gloo_ranks(tmp_path):
    global WORKER
    saved = WORKER
    try:
        WORKER = REFINE_WORKER
        one = _run(1, tmp_path)
        two = _run(2, tmp_path)
    finally:
        WORKER = saved
    assert one["scores"] == two["scores"]
    assert one["trees"] == two["trees"]
    assert all(s < 2 ** 31 for s in one["scores"])


GATHER_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["MPF_ROOT"])
import torch.distributed as dist
from mpboot_amd import shard
ws = int(os.environ.get("WORLD_SIZE", "1"))
if ws > 1:
    dist.init_process_group("gloo")
rank = shard.world()[0]
out = []
SIZES = [5, 0, 3, 17, shard.EVENT_BLOCK + 9, shard.EVENT_BLOCK]      # the last two: more than / exactly one fixed-size block
for rnd in range(6):
    rng = np.random.default_rng(100 * rnd + rank)
    n = SIZES[(rnd + rank) % 6] if rnd != 2 else 0                   # round 2: nobody has events
    local = rng.integers(0, 2 ** 32 - 1, size=(n, 3), dtype=np.uint64).astype(np.uint32)   # full 32-bit values survive
    merged = shard.gather_events(local, tag=1000 + rnd)
    assert merged.dtype == np.uint32
    out.append(sorted(map(tuple, merged.tolist())))
# ranks that are not at the same point of the run must notice
try:
    shard.gather_events(np.zeros((1, 3), dtype=np.uint32), tag=50 + rank)
    out.append("no error")
except RuntimeError as exc:
    out.append("out of step" if "out of step" in str(exc) else str(exc))
if rank == 0:
    print("RESULT " + json.dumps(out))
if ws > 1:
    dist.destroy_process_group()
'''


def test_event_gather_over_two_gloo_ranks(tmp_path):
    """the per-batch exchange of the sample-sharded online UFBoot phase as ONE fixed-size all-gather: uneven and empty
    contributions, a rank with more events than a block holds (second, exactly sized collective), ranks out of step"""
    global WORKER
    saved = WORKER
    try:
        WORKER = GATHER_WORKER
        two = _run(2, tmp_path)
    finally:
        WORKER = saved
    from mpboot_amd import shard
    sizes = [5, 0, 3, 17, shard.EVENT_BLOCK + 9, shard.EVENT_BLOCK]
    for rnd in range(6):
        want = []
        for rank in range(2):
            rng = np.random.default_rng(100 * rnd + rank)
            n = sizes[(rnd + rank) % 6] if rnd != 2 else 0
            want += list(map(tuple, rng.integers(0, 2 ** 32 - 1, size=(n, 3), dtype=np.uint64).astype(np.uint32).tolist()))
        assert [tuple(x) for x in two[rnd]] == sorted(want)
    assert two[6] == "out of step"


def test_in_process_multi_device_schedule():
    """integration/multi_device.hpp: unit b -> device b % G, in increasing b per device, seed base + 12345 b -- the same map
    the process-level sharding uses (shard.units_of_rank / shard.unit_seed), printed by the C++ driver (no GPU touched)"""
    import re
    from mpboot_amd import shard
    drv = os.path.join(ROOT, "oracle", "_build", "multi_device_driver")
    if not os.path.exists(drv):
        import pytest
        pytest.skip("oracle/_build/multi_device_driver not built")
    for units, G in ((10, 4), (1000, 8), (3, 5)):
        out = subprocess.run([drv, "map", str(units), str(G)], check=True, capture_output=True, text=True).stdout.splitlines()
        assert len(out) == G
        seen = []
        for d, line in enumerate(out):
            pairs = [(int(a), int(b)) for a, b in re.findall(r"(\d+)\(seed (\d+)\)", line)]
            got = [a for a, _ in pairs]
            assert got == shard.units_of_rank(units, d, G)
            assert [b for _, b in pairs] == [shard.unit_seed(7, u) for u in got]
            seen += got
        assert sorted(seen) == list(range(units))


def test_online_phase_sharding_policy():
    """below ONLINE_SHARD_MIN_SAMPLES every rank keeps all samples (the pipelined climb leaves the per-batch exchange no room,
    DESIGN 9); the explicit modes override"""
    from mpboot_amd import shard
    assert shard.online_shard(1000, 3, 8) is None
    assert shard.online_shard(shard.ONLINE_SHARD_MIN_SAMPLES, 3, 8) == (3, 8)
    assert shard.online_shard(10 ** 6, 0, 1) is None
    assert shard.online_shard(1000, 1, 2, "1") == (1, 2)
    assert shard.online_shard(10 ** 6, 1, 2, "0") is None

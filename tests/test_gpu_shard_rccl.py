"""The sharding helpers on the RCCL backend ("nccl" on ROCm).  A 1-GPU box cannot host two RCCL ranks (duplicate GPU), so
this runs the collectives of mpboot_amd/shard.py in a single-rank process group on the GPU: device placement, dtypes and
the two-stage all-gather of the event exchange are the same code the 8-GPU bench runs; the multi-rank logic itself is
covered by the gloo tests (tests/test_shard_gloo.py) and the two-process engine test (tests/test_gpu_ufboot.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, json
sys.path.insert(0, %r)
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = "29541"
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from mpboot_amd import shard
ev = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], dtype=np.uint32)
res = {"gather": shard.gather_events(ev, 11).tolist(),
       "empty": shard.gather_events(np.zeros((0, 3), dtype=np.uint32), 12).tolist(),
       "best": [int(x) for x in shard.reduce_best({0: 5, 2: 9}, 3)[0]],
       "tree": shard.broadcast_tree(np.arange(12, dtype=np.int32), 0, 12).tolist()}
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RESULT " + json.dumps(res))
""" % ROOT


def test_shard_collectives_on_rccl_single_rank():
    import json

    out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[7:])
    assert res["gather"] == [[1, 2, 3], [4, 5, 6], [7, 8, 9]]
    assert res["empty"] == []
    assert res["best"][0] == 5 and res["best"][2] == 9 and res["best"][1] > 2 ** 62
    assert res["tree"] == list(range(12))

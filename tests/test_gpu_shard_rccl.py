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


# ---- two real RCCL ranks (one GPU each).  The pool's boxes have ONE GPU, so this skips there; it runs the first time the suite
# meets a node with two or more -- the driver's multi-GPU box -- and covers what the gloo tests cannot: device tensors through
# RCCL over xGMI in the event exchange, the best-score all-reduce, the tree broadcast, and BASELINE config 4's sample-sharded
# online phase with one engine per GPU.
RCCL2_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["MPF_ROOT"])
import numpy as np, torch, torch.distributed as dist
rank, ws = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
lr = int(os.environ.get("LOCAL_RANK", rank))
torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
from mpboot_amd import engine, shard, synth, trees
out = {}
SIZES = [5, 0, 3, 17, shard.EVENT_BLOCK + 9, shard.EVENT_BLOCK]
g = []
for rnd in range(6):
    rng = np.random.default_rng(100 * rnd + rank)
    n = SIZES[(rnd + rank) % 6] if rnd != 2 else 0
    local = rng.integers(0, 2 ** 32 - 1, size=(n, 3), dtype=np.uint64).astype(np.uint32)
    g.append(sorted(map(tuple, shard.gather_events(local, tag=1000 + rnd).tolist())))
out["gather"] = g
out["best"] = [int(x) for x in shard.reduce_best({rank: 100 + rank, 2: 7 if rank == 1 else 9}, 3)[0]]
out["tree"] = shard.broadcast_tree(np.arange(12, dtype=np.int32) * (1 if rank == 0 else -1), 0, 12).tolist()
# the sample-sharded online UFBoot phase: one engine per GPU, every second sample each
letters, _ = synth.synth_alignment(40, 2000, "DNA", 0.08, seed=21)
codes = synth.letters_to_codes(letters, "DNA")
P = codes.shape[1]
samples = np.random.default_rng(7).multinomial(P, np.ones(P) / P, size=64).astype(np.uint16)
back = trees.random_topology(40, np.random.default_rng(5))
e = engine.FitchEngine(codes, device=lr)
e.ufboot_attach(samples, 0.5, shard=(rank, ws))
e.set_tree(back); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 1)
s = e.optimize_spr(1, 6)
logl, cnt, tr = e.ufboot_state()
res = {"s": s, "logl": logl.tolist(), "cnt": cnt.tolist(), "tr": tr.tolist(), "final": e.get_tree().tolist(),
       "draws": e.ufboot_counters()["tie_draws"]}
allr = [None] * ws
dist.all_gather_object(allr, res)
out["ranks_agree"] = all(r == allr[0] for r in allr)
out["ufb"] = res
# the same phase once more with the library's OWN exchange (mpf_rccl_exchange: ncclAllGather from inside libmpfitch.so)
comm = shard.native_comm(lr)
e2 = engine.FitchEngine(codes, device=lr)
e2.ufboot_attach(samples, 0.5, shard=(rank, ws), exchange=comm)
e2.set_tree(back); e2.reset_node_order(); e2.seed_ties(engine.TIE_RANDOM, 1)
s2 = e2.optimize_spr(1, 6)
l2, c2, t2 = e2.ufboot_state()
out["native_same"] = (s2, l2.tolist(), c2.tolist(), t2.tolist(), e2.get_tree().tolist()) == (s, logl.tolist(), cnt.tolist(), tr.tolist(), e.get_tree().tolist())
out["native_exchanges"] = comm.counters()["exchanges"]
out["native_min"] = comm.allreduce_min([100 + rank, 50 - rank, 7]).tolist()
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
if rank == 0:
    print("RESULT " + json.dumps(out))
'''


def _gpu_count():
    """GPUs a process started from here can use, asked of a CHILD process: no HIP call and no torch import in the pytest process
    (a runtime started here at collection time would be running before mpboot_amd.engine exports GPU_MAX_HW_QUEUES; sysfs lists
    the host's GPUs, not the ones this container was given)"""
    probe = ("import torch\nok = 0\nfor i in range(torch.cuda.device_count()):\n    try:\n        torch.cuda.set_device(i); torch.zeros(1, device='cuda'); ok += 1\n"
             "    except Exception:\n        break\nprint(ok)")       # (device_count() can name GPUs of the host this container cannot open)
    out = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=300)
    try:
        return int(out.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def test_two_rccl_ranks_exchange_and_sample_sharded_online_phase(tmp_path):
    if _gpu_count() < 2:
        pytest.skip("needs two GPUs: RCCL refuses two ranks on one device")
    import json
    import socket

    import numpy as np
    from mpboot_amd import engine, shard, synth, trees
    script = tmp_path / "worker.py"
    script.write_text(RCCL2_WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MPF_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    got = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    sizes = [5, 0, 3, 17, shard.EVENT_BLOCK + 9, shard.EVENT_BLOCK]
    for rnd in range(6):
        want = []
        for rank in range(2):
            rng = np.random.default_rng(100 * rnd + rank)
            n = sizes[(rnd + rank) % 6] if rnd != 2 else 0
            want += list(map(tuple, rng.integers(0, 2 ** 32 - 1, size=(n, 3), dtype=np.uint64).astype(np.uint32).tolist()))
        assert [tuple(x) for x in got["gather"][rnd]] == sorted(want)
    assert got["best"] == [100, 101, 7]
    assert got["tree"] == list(range(12))
    assert got["ranks_agree"]
    assert got["native_same"] and got["native_exchanges"] > 0 and got["native_min"] == [100, 49, 7]
    # == the unsharded engine on this process's GPU
    letters, _ = synth.synth_alignment(40, 2000, "DNA", 0.08, seed=21)
    codes = synth.letters_to_codes(letters, "DNA")
    P = codes.shape[1]
    samples = np.random.default_rng(7).multinomial(P, np.ones(P) / P, size=64).astype(np.uint16)
    back = trees.random_topology(40, np.random.default_rng(5))
    e = engine.FitchEngine(codes)
    e.ufboot_attach(samples, 0.5)
    e.set_tree(back); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 1)
    s = e.optimize_spr(1, 6)
    logl, cnt, tr = e.ufboot_state()
    u = got["ufb"]
    assert (u["s"], u["logl"], u["cnt"], u["tr"], u["final"]) == (s, logl.tolist(), cnt.tolist(), tr.tolist(), e.get_tree().tolist())
    assert u["draws"] == e.ufboot_counters()["tie_draws"]


def test_native_rccl_exchange_on_one_rank():
    """mpf_rccl_* (mpboot_amd/host/rccl_exchange.cpp): the library's own communicator -- ncclCommInitRank, the fixed-block
    ncclAllGather of a batch's events incl. the overflow gather, ncclAllReduce(min) -- on the one GPU this box has, as a
    communicator of ONE rank: a sample-"sharded" tracker whose shard is every sample goes through the exchange at every batch
    and must reproduce the unsharded run.  (Two real ranks: the test above, wherever two GPUs are visible.)"""
    import ctypes as C

    import numpy as np
    from mpboot_amd import engine, synth, trees
    if not engine.RcclComm.available():
        pytest.skip("librccl.so not found")
    comm = engine.RcclComm(engine.RcclComm.unique_id(), 0, 1, 0)
    assert comm.allreduce_min([9, 3, 2 ** 32 - 1]).tolist() == [9, 3, 2 ** 32 - 1]
    letters, _ = synth.synth_alignment(40, 2000, "DNA", 0.08, seed=21)
    codes = synth.letters_to_codes(letters, "DNA")
    P = codes.shape[1]
    back = trees.random_topology(40, np.random.default_rng(5))
    for B in (64, 700):                                   # 700 samples: batches with more than 4096 events (the second gather)
        samples = np.random.default_rng(7).multinomial(P, np.ones(P) / P, size=B).astype(np.uint16)
        ref = engine.FitchEngine(codes)
        ref.ufboot_attach(samples, 0.5)
        ref.set_tree(back); ref.reset_node_order(); ref.seed_ties(engine.TIE_RANDOM, 1)
        s = ref.optimize_spr(1, 6)
        e = engine.FitchEngine(codes)
        L = engine.load_library()
        ids = np.arange(B, dtype=np.int32)
        e.ufb_B = B
        rc = L.mpf_ufboot_attach_sharded(e.h, B, B, ids.ctypes.data_as(C.c_void_p), samples.ctypes.data_as(C.c_void_p), C.c_double(0.5),
                                         C.cast(L.mpf_rccl_exchange, C.c_void_p), comm.h)
        assert rc == 0, L.mpf_last_error()
        e.set_tree(back); e.reset_node_order(); e.seed_ties(engine.TIE_RANDOM, 1)
        before = comm.counters()
        assert e.optimize_spr(1, 6) == s
        after = comm.counters()
        assert after["exchanges"] > before["exchanges"]
        if B == 700:
            assert after["overflows"] > before["overflows"]
        for a, b in zip(e.ufboot_state(), ref.ufboot_state()):
            assert a.tolist() == b.tolist()
        assert (e.get_tree() == ref.get_tree()).all() and e.tie_state() == ref.tie_state()
        assert e.ufboot_counters()["tie_draws"] == ref.ufboot_counters()["tie_draws"]
        assert e.ufboot_tree_logl().tolist() == ref.ufboot_tree_logl().tolist()
        e.ufboot_detach()                                 # (the tracker holds comm.h as its exchange argument: release it before the communicator goes)
    # the Python face counts the engines attached through a communicator and refuses to free it under them
    e2 = engine.FitchEngine(codes)
    e2.ufboot_attach(samples, 0.5, shard=(0, 2), exchange=comm)
    with pytest.raises(engine.MpfError):
        comm.close()
    e2.ufboot_detach()
    comm.close()

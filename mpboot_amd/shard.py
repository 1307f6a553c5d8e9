"""Sharding of independent units (start trees, bootstrap replicates) over the GPUs of one node.

SURVEY.md §8(e): the units are independent, the tips are replicated on every GPU, and the only
exchange is one small all-reduce of best scores per round (torch.distributed: backend "nccl" is RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).  Unit u belongs to rank u % world.

The reference runs these units sequentially with one shared RNG stream (phyloanalysis.cpp:1270-1317,
iqtree.cpp:2515-2866); sharded runs give every unit its own stream, seeded as the reference seeds its
start trees (ran_seed + unit * 12345, phyloanalysis.cpp:1273).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

INT_MAX = np.iinfo(np.int64).max


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def units_of_rank(n_units: int, rank: int, world_size: int):
    return list(range(rank, n_units, world_size))


def unit_seed(base_seed: int, unit: int) -> int:
    return base_seed + unit * 12345


def _device():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def reduce_best(scores_local: dict, n_units: int):
    """scores_local: {unit: score} of the units this rank processed -> (scores[n_units], best_unit, owner_rank).

    One all-reduce(MIN) of n_units int64 -- <= 8 KB for 1000 replicates."""
    rank, ws = world()
    t = torch.full((n_units,), INT_MAX, dtype=torch.int64, device=_device())
    for u, s in scores_local.items():
        t[u] = int(s)
    if ws > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    scores = t.cpu().numpy()
    best_unit = int(np.argmin(scores))           # lowest unit index among equal scores: deterministic
    return scores, best_unit, best_unit % ws


def broadcast_tree(back, owner: int, length: int):
    """Hand the winning topology (int32 record links) from its owner to every rank."""
    rank, ws = world()
    if ws == 1:
        return np.asarray(back, dtype=np.int32)
    t = torch.zeros(length, dtype=torch.int32, device=_device())
    if rank == owner:
        t.copy_(torch.from_numpy(np.ascontiguousarray(back, dtype=np.int32)))
    dist.broadcast(t, src=owner)
    return t.cpu().numpy()


def search_start_trees(make_engine, n_units: int, base_seed: int, spr_radius: int = 6):
    """Independent randomized-stepwise-addition start trees, each SPR-optimised, sharded over ranks
    (the reference's initCandidateTreeSet loop, phyloanalysis.cpp:1261-1317).

    make_engine() -> an object with seed_ties(mode, seed), make_parsimony_tree(seed, dist),
    get_tree(); returns (scores[n_units], best_unit, best_tree)."""
    rank, ws = world()
    eng = make_engine()
    local, trees = {}, {}
    for u in units_of_rank(n_units, rank, ws):
        seed = unit_seed(base_seed, u)
        eng.seed_ties(1, seed)
        if hasattr(eng, "reset_node_order"):
            eng.reset_node_order()      # fresh instance state per unit: results do not depend on the sharding
        local[u] = eng.make_parsimony_tree(seed, spr_radius)
        if isinstance(local[u], tuple):
            local[u] = local[u][0]
        trees[u] = eng.get_tree()
    scores, best_unit, owner = reduce_best(local, n_units)
    nrec = len(next(iter(trees.values()))) if trees else 0
    if ws > 1:
        ln = torch.tensor([nrec], dtype=torch.int64, device=_device())
        dist.all_reduce(ln, op=dist.ReduceOp.MAX)
        nrec = int(ln.item())
    best_tree = broadcast_tree(trees.get(best_unit), owner, nrec)
    return scores, best_unit, best_tree


# ---------------------------------------------------------------- online UFBoot: per-batch event exchange
import ctypes as _C

EXCHANGE_FN = _C.CFUNCTYPE(_C.c_int, _C.c_void_p, _C.c_uint32, _C.c_void_p, _C.c_uint32, _C.POINTER(_C.c_void_p), _C.POINTER(_C.c_uint32))


# events a rank contributes per exchange in the common case: one fixed-size block [count, tag, EVENT_BLOCK triples] per rank
# goes out in ONE all-gather (a climb's batch books a few hundred events per rank; a block is 48 KB, xGMI moves it in
# microseconds, and a fixed shape lets RCCL reuse its buffers and plan).  A rank with more than that says so in its count and
# everybody takes part in a second, exactly sized all-gather for the overflow.
EVENT_BLOCK = 4096
_ev_bufs = {}


ONLINE_SHARD_MIN_SAMPLES = 4096


def online_shard(n_samples: int, rank: int, world_size: int, mode: str = "auto"):
    """The `shard` argument of FitchEngine.ufboot_attach for the ONLINE phase of a run on `world_size` GPUs, or None = every rank
    keeps all samples and makes the same calls (replicas: identical results by construction, no exchange).
    Sharding divides the REPS product and the event extraction of every scan batch (C3: 26 + 14 us per batch and 1000 samples
    on one GPU) and costs one all-gather of the batch's events on the HOST's side of the pipelined climb (the host
    has 35 us to spare per batch; the exchange -- two copies, the collective, Python -- is priced at 100-150 us).  Per batch:
    65 + 40 * B / 1000 us unsharded against max(65 + 40 * B / 1000 / W, 220) us sharded, so sharding pays from about 4000
    samples on, whatever W.  mode: "auto" (that rule), "1" (always), "0" (never)."""
    if world_size <= 1 or mode == "0":
        return None
    if mode == "1" or n_samples >= ONLINE_SHARD_MIN_SAMPLES:
        return (rank, world_size)
    return None


def gather_events(local: np.ndarray, tag: int = 0) -> np.ndarray:
    """All-gather of the (candidate index, sample, score) triples of one scan batch: [n_local, 3] uint32 in,
    [n_all, 3] out, identical on every rank.  One fixed-size collective on the default process group -- RCCL over xGMI on the
    GPU box ("nccl"), gloo in the CPU tests -- plus a second one only when some rank has more than EVENT_BLOCK events.  Ranks
    whose tags differ are not at the same point of the run: everybody raises instead of running on."""
    rank, ws = world()
    if ws == 1:
        return local
    dev = _device()
    key = (str(dev), ws)
    if key not in _ev_bufs:
        _ev_bufs[key] = (torch.zeros((EVENT_BLOCK + 1, 3), dtype=torch.int32, device=dev),
                         torch.zeros((ws, EVENT_BLOCK + 1, 3), dtype=torch.int32, device=dev))
    blk, allb = _ev_bufs[key]
    n_loc = int(local.shape[0])
    head = min(n_loc, EVENT_BLOCK)
    host = np.zeros((EVENT_BLOCK + 1, 3), dtype=np.int32)
    host[0, 0] = n_loc
    host[0, 1] = np.int64(int(tag) & 0x7FFFFFFF)
    if head:
        host[1:1 + head] = local[:head].view(np.int32)
    blk.copy_(torch.from_numpy(host))
    dist.all_gather_into_tensor(allb.view(ws * (EVENT_BLOCK + 1), 3), blk)
    got = allb.cpu().numpy()
    counts = got[:, 0, 0].astype(np.int64)
    tags = got[:, 0, 1]
    if (tags != tags[0]).any():
        raise RuntimeError(f"ranks out of step in the event exchange: tags {tags.tolist()}")
    parts = [got[r, 1:1 + min(int(counts[r]), EVENT_BLOCK)] for r in range(ws)]
    over = int(max(0, counts.max() - EVENT_BLOCK))
    if over:
        # (rare: more events than a block holds -- the remainder in an all-gather of exactly the size the largest rank needs)
        buf = torch.zeros((over, 3), dtype=torch.int32, device=dev)
        if n_loc > EVENT_BLOCK:
            buf[:n_loc - EVENT_BLOCK] = torch.from_numpy(np.ascontiguousarray(local[EVENT_BLOCK:]).view(np.int32)).to(dev)
        out = torch.zeros((ws, over, 3), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(out.view(ws * over, 3), buf)
        out = out.cpu().numpy()
        parts = [np.concatenate([parts[r], out[r, :max(0, int(counts[r]) - EVENT_BLOCK)]], axis=0) for r in range(ws)]
    if counts.sum() == 0:
        return local
    return np.ascontiguousarray(np.concatenate(parts, axis=0)).view(np.uint32)


def native_comm(device: int = 0):
    """The library's own RCCL communicator for this process's rank (engine.RcclComm), its id carried from rank 0 over
    torch.distributed (any backend: 128 bytes, once): pass it as `exchange` to FitchEngine.ufboot_attach for an event exchange
    that never leaves the library.  None where librccl is not available or the run has one rank."""
    from . import engine as _engine
    rank, ws = world()
    if ws == 1 or not _engine.RcclComm.available():
        return None
    box = [_engine.RcclComm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return _engine.RcclComm(box[0], rank, ws, device)


def event_exchange():
    """The callback mpf_ufboot_attach_sharded takes (include/mpfitch.h: mpf_ufb_exchange_fn)."""
    keep = {}

    def fn(_arg, tag, local_ptr, n_local, all_ptr, n_all_ptr):
        try:
            if n_local:
                local = np.ctypeslib.as_array(_C.cast(local_ptr, _C.POINTER(_C.c_uint32)), shape=(n_local, 3)).copy()
            else:
                local = np.zeros((0, 3), dtype=np.uint32)
            merged = np.ascontiguousarray(gather_events(local, tag), dtype=np.uint32)
            keep["buf"] = merged                         # valid until the next call
            all_ptr[0] = merged.ctypes.data if merged.shape[0] else None
            n_all_ptr[0] = merged.shape[0]
            return 0
        except Exception as exc:                         # never let an exception cross the C boundary
            keep["error"] = exc
            return 1

    cb = EXCHANGE_FN(fn)
    cb._keep = keep
    return cb

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


def gather_events(local: np.ndarray, tag: int = 0) -> np.ndarray:
    """All-gather of the (candidate index, sample, score) triples of one scan batch: [n_local, 3] uint32 in,
    [n_all, 3] out, identical on every rank.  Two small collectives (counts + tag, then the padded triples) on the default
    process group -- RCCL over xGMI on the GPU box ("nccl"), gloo in the CPU tests.  Ranks whose tags differ are not at
    the same point of the run: everybody raises instead of running on."""
    rank, ws = world()
    if ws == 1:
        return local
    dev = _device()
    n = torch.tensor([local.shape[0], int(tag)], dtype=torch.int64, device=dev)
    both = torch.zeros(2 * ws, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(both, n)
    both = both.cpu().numpy().reshape(ws, 2)
    if (both[:, 1] != both[0, 1]).any():
        raise RuntimeError(f"ranks out of step in the event exchange: tags {both[:, 1].tolist()}")
    counts = both[:, 0]
    m = int(counts.max())
    if m == 0:
        return local
    buf = torch.zeros((m, 3), dtype=torch.int32, device=dev)
    if local.shape[0]:
        buf[:local.shape[0]] = torch.from_numpy(local.view(np.int32)).to(dev)
    out = torch.zeros((ws, m, 3), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out.view(ws * m, 3), buf)
    out = out.cpu().numpy().view(np.uint32)
    return np.concatenate([out[r, :int(counts[r])] for r in range(ws)], axis=0)


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

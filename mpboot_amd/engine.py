"""ctypes binding of libmpfitch.so (include/mpfitch.h) -- the product's Python face.

There is no fallback: if the HIP library is missing or no GPU is present the
calls raise.  Nothing here imports the oracle.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# (MPF_LIB_PATH: a differently built library, e.g. the experiments build of tools/scan_bounds.sh -- never the default)
LIB_PATH = os.environ.get("MPF_LIB_PATH") or os.path.join(HERE, "libmpfitch.so")
# Engines on several host threads launch side by side only as far as their streams get hardware queues of their own; the HIP
# runtime's default is 4 per process and a persistent climb kernel holds its queue for a whole sweep.  The setting belongs to
# the process, not to the library: this module is the host side and sets it before the runtime starts, unless the user did.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

DNA, AA, BIN, GENERIC = 0, 1, 2, 3      # PLL_DNA_DATA, PLL_AA_DATA, PLL_BINARY_DATA (2 states), PLL_GENERIC_32 (multistate)
TIE_FIRST, TIE_RANDOM = 0, 1

EXPORTS = [
    "mpf_last_error", "mpf_abi_version", "mpf_engine_create", "mpf_engine_create_sankoff", "mpf_engine_destroy",
    "mpf_set_weights",
    "mpf_get_geometry", "mpf_get_informative", "mpf_get_tip_vector", "mpf_set_tree", "mpf_get_tree",
    "mpf_reset_node_order", "mpf_score_tree", "mpf_score_trees", "mpf_pattern_scores", "mpf_site_scores", "mpf_compute_parsimony", "mpf_compute_parsimony_at",
    "mpf_encode_iqtree_states", "mpf_seed_ties", "mpf_set_tie_state", "mpf_get_tie_state", "mpf_tie_state_after",
    "mpf_set_rand_callback", "mpf_spr_scan", "mpf_spr_sweep_scan", "mpf_spr_sweep_costs", "mpf_get_node_order", "mpf_optimize_spr",
    "mpf_make_parsimony_tree", "mpf_stepwise_addition", "mpf_get_moves", "mpf_get_stats", "mpf_reset_stats",
    "mpf_set_option", "mpf_get_option", "mpf_get_scan_trace", "mpf_reps_create", "mpf_reps_scores", "mpf_reps_destroy",
    "mpf_ufboot_attach", "mpf_ufboot_refine_sweep", "mpf_ufboot_attach_sharded", "mpf_ufboot_detach", "mpf_ufboot_set_cutoff", "mpf_ufboot_set_ratchet_booking", "mpf_ufboot_set_mulhits", "mpf_ufboot_set_store_trees", "mpf_ufboot_get_duplicates", "mpf_ufboot_get_sample_trees", "mpf_ufboot_set_topboot", "mpf_ufboot_get_sample_top", "mpf_ufboot_set_distinct_iter", "mpf_ufboot_set_iteration", "mpf_ufboot_get_sample_iters", "mpf_ufboot_next_cutoff", "mpf_ufboot_set_cutoff_from_btrees", "mpf_ufboot_get_orig_logl", "mpf_rccl_available", "mpf_rccl_unique_id", "mpf_rccl_create", "mpf_rccl_destroy", "mpf_rccl_exchange", "mpf_rccl_allreduce_min", "mpf_rccl_counters", "mpf_ufboot_num_trees",
    "mpf_ufboot_tree_logl", "mpf_ufboot_get_state", "mpf_ufboot_get_tree", "mpf_ufboot_get_counters",
    "mpf_min_pars_score_patterns", "mpf_mst_scores", "mpf_segment_patterns", "mpf_remain_bounds",
    "mpf_cost_matrix_load", "mpf_cost_matrix_triangle_fix",
    "mpf_iq_random_nnis", "mpf_iq_perturb_weights", "mpf_iq_topology_key", "mpf_ufboot_adopt", "mpf_optimize_spr_many", "mpf_optimize_spr_many_round",
]


class MpfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmpfitch error {code}: {msg}")
        self.code = code


def load_cost_matrix(file_or_keyword: str, n_states_alignment: int):
    """ParsTree::loadCostMatrixFile: ("fitch" | "e" | path) -> (cost[S, S] uint32 after the triangle repair, repaired?)"""
    cap = 64
    cost = np.zeros(cap * cap, dtype=np.uint32)
    S, ch = C.c_int32(), C.c_int32()
    _chk(load_library().mpf_cost_matrix_load(str(file_or_keyword).encode(), n_states_alignment, cap, _p(cost), C.byref(S), C.byref(ch)))
    return cost[:S.value * S.value].reshape(S.value, S.value).copy(), bool(ch.value)


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("n_taxa", C.c_int32), ("n_patterns", C.c_int32), ("datatype", C.c_int32),
                ("keep_all_sites", C.c_int32), ("reserved", C.c_int32 * 3)]


class Stats(C.Structure):
    _fields_ = [("insertion_tests", C.c_uint64), ("newview_ops", C.c_uint64), ("scan_launches", C.c_uint64),
                ("view_launches", C.c_uint64), ("moves_applied", C.c_uint64), ("algorithmic_bytes", C.c_uint64),
                ("last_scan_kernel_ms", C.c_double), ("scan_kernel_ms_total", C.c_double),
                ("view_kernel_ms_total", C.c_double), ("host_plan_ms_total", C.c_double),
                ("host_views_ms_total", C.c_double), ("host_scan_ms_total", C.c_double),
                ("host_sweep_ms_total", C.c_double), ("plan_kernel_ms_total", C.c_double), ("plan_launches", C.c_uint64),
                ("climb_launches", C.c_uint64), ("climb_steps", C.c_uint64), ("climb_nodes", C.c_uint64),
                ("climb_moves", C.c_uint64), ("climb_ms_total", C.c_double)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


_lib = None


def load_library():
    """Load libmpfitch.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        L.mpf_last_error.restype = C.c_char_p
        vp = C.c_void_p
        L.mpf_engine_create.argtypes = [C.POINTER(vp), C.POINTER(Config), vp, vp]
        L.mpf_engine_create_sankoff.argtypes = [C.POINTER(vp), C.POINTER(Config), vp, vp, vp]
        L.mpf_engine_destroy.argtypes = [vp]
        L.mpf_engine_destroy.restype = None
        L.mpf_set_weights.argtypes = [vp, vp]
        L.mpf_get_geometry.argtypes = [vp, vp, vp, vp, vp]
        L.mpf_get_informative.argtypes = [vp, vp]
        L.mpf_get_tip_vector.argtypes = [vp, C.c_int32, vp]
        L.mpf_set_tree.argtypes = [vp, vp]
        L.mpf_get_tree.argtypes = [vp, vp]
        L.mpf_reset_node_order.argtypes = [vp]
        L.mpf_score_tree.argtypes = [vp, vp]
        L.mpf_score_trees.argtypes = [vp, C.c_int32, vp, vp]
        L.mpf_pattern_scores.argtypes = [vp, vp, vp]
        L.mpf_site_scores.argtypes = [vp, vp, C.c_int32, vp]
        L.mpf_compute_parsimony.argtypes = [vp, vp, vp, vp]
        L.mpf_compute_parsimony_at.argtypes = [vp, vp, C.c_int32, vp, vp]
        L.mpf_encode_iqtree_states.argtypes = [C.c_int32, vp, C.c_int64, vp]
        L.mpf_seed_ties.argtypes = [vp, C.c_int32, C.c_int32]
        L.mpf_set_tie_state.argtypes = [vp, C.c_uint64]
        L.mpf_get_tie_state.argtypes = [vp, vp]
        L.mpf_tie_state_after.argtypes = [C.c_uint64, C.c_uint64]
        L.mpf_tie_state_after.restype = C.c_uint64
        L.mpf_spr_scan.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp]
        L.mpf_spr_sweep_scan.argtypes = [vp, C.c_int32, C.c_int32, vp, vp]
        L.mpf_spr_sweep_costs.argtypes = [vp, C.c_int32, C.c_int32, C.c_uint64, vp, vp, vp]
        L.mpf_get_node_order.argtypes = [vp, vp]
        L.mpf_optimize_spr.argtypes = [vp, C.c_int32, C.c_int32, vp]
        L.mpf_make_parsimony_tree.argtypes = [vp, C.c_int64, C.c_int32, vp]
        L.mpf_stepwise_addition.argtypes = [vp, C.c_int64, vp, vp, vp]
        L.mpf_get_moves.argtypes = [vp, C.c_int32, vp, vp, vp, vp]
        L.mpf_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.mpf_reset_stats.argtypes = [vp]
        L.mpf_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
        L.mpf_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
        L.mpf_get_scan_trace.argtypes = [vp, vp, C.c_uint64, vp]
        L.mpf_reps_create.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int32, vp]
        L.mpf_reps_scores.argtypes = [vp, C.c_int32, vp, vp]
        L.mpf_reps_destroy.argtypes = [vp]
        L.mpf_reps_destroy.restype = None
        L.mpf_ufboot_attach.argtypes = [vp, C.c_int32, vp, C.c_double]
        L.mpf_ufboot_refine_sweep.argtypes = [vp, C.c_int32, vp, vp, vp, vp]
        L.mpf_ufboot_attach_sharded.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, C.c_double, vp, vp]
        L.mpf_ufboot_detach.argtypes = [vp]
        L.mpf_ufboot_set_cutoff.argtypes = [vp, C.c_double]
        L.mpf_ufboot_set_ratchet_booking.argtypes = [vp, C.c_int32]
        L.mpf_ufboot_set_mulhits.argtypes = [vp, C.c_int32]
        L.mpf_ufboot_set_store_trees.argtypes = [vp, C.c_int32]
        L.mpf_ufboot_get_duplicates.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.mpf_ufboot_get_sample_trees.argtypes = [vp, C.c_int32, vp, C.c_int32, vp]
        L.mpf_ufboot_set_distinct_iter.argtypes = [vp, C.c_int32]
        L.mpf_ufboot_set_iteration.argtypes = [vp, C.c_int32]
        L.mpf_ufboot_get_sample_iters.argtypes = [vp, C.c_int32, vp, C.c_int32, vp]
        L.mpf_ufboot_set_topboot.argtypes = [vp, C.c_int32]
        L.mpf_ufboot_get_sample_top.argtypes = [vp, C.c_int32, vp, vp, C.c_int32, vp, vp]
        L.mpf_ufboot_next_cutoff.argtypes = [vp, C.c_int32, vp]
        L.mpf_ufboot_set_cutoff_from_btrees.argtypes = [vp, C.c_int32]
        L.mpf_rccl_unique_id.argtypes = [vp]
        L.mpf_rccl_create.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32]
        L.mpf_rccl_destroy.argtypes = [vp]
        L.mpf_rccl_destroy.restype = None
        L.mpf_rccl_allreduce_min.argtypes = [vp, vp, C.c_int32]
        L.mpf_rccl_counters.argtypes = [vp, vp, vp]
        L.mpf_ufboot_get_orig_logl.argtypes = [vp, vp]
        L.mpf_ufboot_num_trees.argtypes = [vp, vp]
        L.mpf_ufboot_tree_logl.argtypes = [vp, vp]
        L.mpf_ufboot_get_state.argtypes = [vp, vp, vp, vp]
        L.mpf_ufboot_get_tree.argtypes = [vp, C.c_int64, vp]
        L.mpf_ufboot_get_counters.argtypes = [vp, vp, vp, vp, vp]
        L.mpf_min_pars_score_patterns.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp]
        L.mpf_mst_scores.argtypes = [C.c_int32, vp, C.c_int32, C.c_int32, vp, vp]
        L.mpf_segment_patterns.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp]
        L.mpf_remain_bounds.argtypes = [C.c_int32, C.c_int32, vp, vp, vp, vp]
        L.mpf_cost_matrix_load.argtypes = [C.c_char_p, C.c_int32, C.c_int32, vp, vp, vp]
        L.mpf_cost_matrix_triangle_fix.argtypes = [C.c_int32, vp, vp]
        L.mpf_optimize_spr_many.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp]
        L.mpf_optimize_spr_many_round.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, vp]
        L.mpf_ufboot_adopt.argtypes = [vp, C.c_int32, vp, vp, vp, C.c_int32, vp, vp, vp]
        L.mpf_iq_random_nnis.argtypes = [C.c_int32, vp, C.c_int32, vp, vp]
        L.mpf_iq_perturb_weights.argtypes = [C.c_int32, vp, vp, C.c_int32, C.c_int32, vp, vp]
        L.mpf_iq_topology_key.argtypes = [C.c_int32, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _chk(rc):
    if rc != 0:
        raise MpfError(rc, load_library().mpf_last_error().decode())


def min_pars_score_patterns(codes: np.ndarray, datatype: int = DNA) -> np.ndarray:
    """pllCalcMinParsScorePattern for every pattern (host only)."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    out = np.zeros(codes.shape[1], dtype=np.int32)
    _chk(load_library().mpf_min_pars_score_patterns(datatype, codes.shape[0], codes.shape[1], _p(codes), _p(out)))
    return out


def mst_scores(states: np.ndarray, cost: np.ndarray) -> np.ndarray:
    """ParsTree::findMstScore for every pattern (host only); states = IQ-TREE codes [n][P]."""
    states = np.ascontiguousarray(states, dtype=np.int8)
    cost = np.ascontiguousarray(cost, dtype=np.uint32)
    out = np.zeros(states.shape[1], dtype=np.uint32)
    _chk(load_library().mpf_mst_scores(cost.shape[0], _p(cost), states.shape[0], states.shape[1], _p(states), _p(out)))
    return out


def segment_patterns(ras_pars_score, frequency, n_informative: int, vcsize: int = 16) -> np.ndarray:
    """IQTree::doSegmenting: returns segment_upper[0 .. n_segments)."""
    r = np.ascontiguousarray(ras_pars_score, dtype=np.int32)
    f = np.ascontiguousarray(frequency, dtype=np.int32)
    up = np.zeros(len(r), dtype=np.int32)
    k = C.c_int32()
    _chk(load_library().mpf_segment_patterns(len(r), n_informative, vcsize, _p(r), _p(f), _p(up), C.byref(k)))
    return up[:k.value].copy()


def remain_bounds(segment_upper, min_unit_pars, weight) -> np.ndarray:
    """remain[s] = sum over positions >= segment_upper[s] of min_unit_pars * weight (all but the last segment)."""
    up = np.ascontiguousarray(segment_upper, dtype=np.int32)
    m = np.ascontiguousarray(min_unit_pars, dtype=np.int32)
    w = np.ascontiguousarray(weight, dtype=np.uint16)
    out = np.zeros(max(len(up) - 1, 1), dtype=np.int32)
    _chk(load_library().mpf_remain_bounds(len(m), len(up), _p(up), _p(m), _p(w), _p(out)))
    return out[:len(up) - 1]


def optimize_spr_many(engines, mintrav: int = 1, maxtrav: int = 6):
    """mpf_optimize_spr_many: one SPR hill climb per engine (tree, weights, tie stream set on each as for optimize_spr), all of them
    side by side -- a resident workgroup per climb, all of its sweeps inside ONE launch.  -> final lengths, one per engine."""
    n = len(engines)
    hs = (C.c_void_p * n)(*[e.h for e in engines])
    out = np.zeros(n, dtype=np.uint32)
    _chk(load_library().mpf_optimize_spr_many(hs, n, int(mintrav), int(maxtrav), _p(out)))
    return out


class ClimbBatch:
    """mpf_optimize_spr_many_round for callers with more climbs than engines: start(k) after setting engine k up (tree, weights, tie
    stream), round() runs one launch for every active climb and returns the engines whose climb has just finished."""

    def __init__(self, engines, mintrav: int = 1, maxtrav: int = 6):
        self.engines = list(engines)
        self.n = len(self.engines)
        self.hs = (C.c_void_p * self.n)(*[e.h for e in self.engines])
        self.state = np.zeros(self.n, dtype=np.uint8)
        self.scores = np.zeros(self.n, dtype=np.uint32)
        self.mintrav, self.maxtrav = int(mintrav), int(maxtrav)

    def start(self, k: int):
        self.state[k] = 1

    def active(self) -> int:
        return int((self.state != 0).sum())

    def round(self):
        before = self.state != 0
        _chk(load_library().mpf_optimize_spr_many_round(self.hs, self.n, self.mintrav, self.maxtrav, _p(self.state), _p(self.scores)))
        return [int(k) for k in np.nonzero(before & (self.state == 0))[0]]


def iq_random_nnis(back: np.ndarray, num_nni: int, tie_state: int):
    """IQTree::doRandomNNIs (iqtree.cpp:1083-1106) on a copy of `back` -> (perturbed back, stream state behind the draws, re-listings)."""
    b = np.ascontiguousarray(back, dtype=np.int32).copy()
    n = (len(b) // 3 + 1) // 2
    st = C.c_uint64(tie_state & ((1 << 64) - 1))
    rl = C.c_int32()
    _chk(load_library().mpf_iq_random_nnis(n, _p(b), int(num_nni), C.byref(st), C.byref(rl)))
    return b, int(st.value), int(rl.value)


def iq_perturb_weights(weights, informative, percent: int, add: int, tie_state: int):
    """Alignment::createPerturbAlignment (alignment.cpp:1915-1969) as pattern weights -> (weights, stream state behind the draws)."""
    w = np.ascontiguousarray(weights, dtype=np.int32)
    inf = np.ascontiguousarray(informative, dtype=np.uint8)
    out = np.zeros_like(w)
    st = C.c_uint64(tie_state & ((1 << 64) - 1))
    _chk(load_library().mpf_iq_perturb_weights(len(w), _p(w), _p(inf), int(percent), int(add), C.byref(st), _p(out)))
    return out, int(st.value)


def iq_topology_key(back: np.ndarray) -> bytes:
    """16-byte digest of the canonical unrooted topology (the key CandidateSet::topologies looks trees up by)."""
    b = np.ascontiguousarray(back, dtype=np.int32)
    n = (len(b) // 3 + 1) // 2
    key = np.zeros(2, dtype=np.uint64)
    _chk(load_library().mpf_iq_topology_key(n, _p(b), _p(key)))
    return key.tobytes()


def encode_iqtree_states(states: np.ndarray, datatype: int = DNA) -> np.ndarray:
    """Alignment::convertState codes -> PLL tip codes (host only, no GPU needed)."""
    states = np.ascontiguousarray(states, dtype=np.int8)
    out = np.zeros(states.shape, dtype=np.uint8)
    _chk(load_library().mpf_encode_iqtree_states(datatype, _p(states), states.size, _p(out)))
    return out


class Reps:
    """REPS contraction (IQTree::saveCurrentTree, iqtree.cpp:3411-3449): boot weights resident on the GPU."""

    def __init__(self, boot: np.ndarray, device: int = 0):
        boot = np.ascontiguousarray(boot, dtype=np.uint16)
        self.B, self.P = boot.shape
        h = C.c_void_p()
        _chk(load_library().mpf_reps_create(C.byref(h), device, self.B, self.P, _p(boot)))
        self.h = h

    def scores(self, pattern_pars: np.ndarray) -> np.ndarray:
        pp = np.ascontiguousarray(np.atleast_2d(pattern_pars), dtype=np.uint16)
        assert pp.shape[1] == self.P
        out = np.zeros((pp.shape[0], self.B), dtype=np.int32)
        _chk(load_library().mpf_reps_scores(self.h, pp.shape[0], _p(pp), _p(out)))
        return out

    def close(self):
        if getattr(self, "h", None):
            load_library().mpf_reps_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RcclComm:
    """The library's own RCCL communicator (include/mpfitch.h: mpf_rccl_*): event exchange of a sample-sharded online phase and the
    all-reduce of best scores, native -- no torch in the data path.  `uid`: 128 bytes from RcclComm.unique_id() on rank 0, carried
    to the other ranks by the caller (mpboot_amd.shard.native_comm does it over torch.distributed's store)."""

    def __init__(self, uid: bytes, rank: int, world: int, device: int = 0):
        L = load_library()
        self.h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        _chk(L.mpf_rccl_create(C.byref(self.h), buf, rank, world, device))
        self.rank, self.world = rank, world
        self._users = 0                          # engines whose tracker holds self.h as its exchange argument

    @staticmethod
    def available() -> bool:
        return bool(load_library().mpf_rccl_available())

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        _chk(load_library().mpf_rccl_unique_id(buf))
        return bytes(buf)

    def allreduce_min(self, vals):
        v = np.ascontiguousarray(vals, dtype=np.uint32).copy()
        _chk(load_library().mpf_rccl_allreduce_min(self.h, _p(v), len(v)))
        return v

    def counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        _chk(load_library().mpf_rccl_counters(self.h, C.byref(a), C.byref(b)))
        return {"exchanges": a.value, "overflows": b.value}

    def close(self):
        """Frees the native communicator.  Refused while an engine's tracker still has it registered as its exchange
        (mpf_ufboot_attach_sharded stored the raw handle: a later climb or the closing handshake would call into freed memory);
        detach or close those engines first."""
        if getattr(self, "h", None):
            if getattr(self, "_users", 0) > 0:
                raise MpfError(-1, "RcclComm.close(): %d engine(s) still attached through this communicator -- ufboot_detach() them first" % self._users)
            load_library().mpf_rccl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FitchEngine:
    """One alignment resident on one MI355X; mirrors the reference's PLL-instance-level calls."""

    def __init__(self, codes: np.ndarray, weights=None, datatype: int = DNA, keep_all: bool = False, device: int = 0,
                 cost=None):
        """cost: optional [S, S] matrix -> weighted (Sankoff) parsimony, the reference's `-cost` mode"""
        L = load_library()
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        self.n, self.P = codes.shape
        if weights is None:
            weights = np.ones(self.P, dtype=np.int32)
        weights = np.ascontiguousarray(weights, dtype=np.int32)
        self._weights = weights.copy()           # the pattern weights in force (mpf_set_weights keeps it in step)
        self.weighted = cost is not None         # the Sankoff (-cost) engine
        cfg = Config(device, self.n, self.P, datatype, int(keep_all))
        h = C.c_void_p()
        if cost is None:
            _chk(L.mpf_engine_create(C.byref(h), C.byref(cfg), _p(codes), _p(weights)))
        else:
            cost = np.ascontiguousarray(cost, dtype=np.uint32)
            _chk(L.mpf_engine_create_sankoff(C.byref(h), C.byref(cfg), _p(codes), _p(weights), _p(cost)))
        self.h = h
        self.nrec = 3 * (2 * self.n - 1)
        s, w, ni, wp = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _chk(L.mpf_get_geometry(self.h, C.byref(s), C.byref(w), C.byref(ni), C.byref(wp)))
        self.S, self.W, self.num_informative, self.Wp = s.value, w.value, ni.value, wp.value

    def _drop_exchange(self):
        ex = getattr(self, "_ufb_exchange", None)
        if isinstance(ex, RcclComm):
            ex._users -= 1
        self._ufb_exchange = None

    def close(self):
        if getattr(self, "h", None):
            load_library().mpf_engine_destroy(self.h)
            self.h = None
            self._drop_exchange()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _refresh_geometry(self):
        s, w, ni, wp = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _chk(load_library().mpf_get_geometry(self.h, C.byref(s), C.byref(w), C.byref(ni), C.byref(wp)))
        self.S, self.W, self.num_informative, self.Wp = s.value, w.value, ni.value, wp.value

    def set_weights(self, w):
        w = np.ascontiguousarray(w, dtype=np.int32)
        _chk(load_library().mpf_set_weights(self.h, _p(w)))
        self._weights = w.copy()
        self._refresh_geometry()

    def weights(self):
        """The pattern weights in force (a copy)."""
        return self._weights.copy()

    def informative(self):
        f = np.zeros(self.P, dtype=np.int32)
        _chk(load_library().mpf_get_informative(self.h, _p(f)))
        return f

    def tip_vector(self, tip: int):
        out = np.zeros((self.S, self.W), dtype=np.uint32)
        _chk(load_library().mpf_get_tip_vector(self.h, tip, _p(out)))
        return out

    def set_tree(self, back):
        back = np.ascontiguousarray(back, dtype=np.int32)
        assert len(back) == self.nrec
        _chk(load_library().mpf_set_tree(self.h, _p(back)))

    def get_tree(self):
        back = np.empty(self.nrec, dtype=np.int32)
        _chk(load_library().mpf_get_tree(self.h, _p(back)))
        return back

    def reset_node_order(self):
        _chk(load_library().mpf_reset_node_order(self.h))

    def score_tree(self, back=None) -> int:
        if back is not None:
            self.set_tree(back)
        s = C.c_uint32()
        _chk(load_library().mpf_score_tree(self.h, C.byref(s)))
        return s.value

    def score_trees(self, backs):
        backs = np.ascontiguousarray(backs, dtype=np.int32)
        out = np.zeros(len(backs), dtype=np.uint32)
        _chk(load_library().mpf_score_trees(self.h, len(backs), _p(backs), _p(out)))
        return out

    def pattern_scores(self):
        out = np.zeros(self.P, dtype=np.uint16)
        tot = C.c_int32()
        _chk(load_library().mpf_pattern_scores(self.h, _p(out), C.byref(tot)))
        return out, tot.value

    def compute_parsimony(self, back=None, want_patterns: bool = True):
        """PhyloTree::computeParsimony(): (score, _pattern_pars)"""
        s = C.c_uint32()
        ptn = np.zeros(self.P, dtype=np.uint16) if want_patterns else None
        bp = None
        if back is not None:
            back = np.ascontiguousarray(back, dtype=np.int32)
            bp = _p(back)
        _chk(load_library().mpf_compute_parsimony(self.h, bp, C.byref(s), _p(ptn) if want_patterns else None))
        return s.value, ptn

    def seed_ties(self, mode: int, seed: int = 1):
        _chk(load_library().mpf_seed_ties(self.h, mode, seed))

    def set_tie_state(self, state: int):
        """Hand the 64-bit state of the host's SPRNG lcg64 stream over (mpf_set_tie_state)."""
        _chk(load_library().mpf_set_tie_state(self.h, state & ((1 << 64) - 1)))

    def tie_state(self) -> int:
        s = C.c_uint64()
        _chk(load_library().mpf_get_tie_state(self.h, C.byref(s)))
        return int(s.value)

    def site_scores(self, n_sites: int):
        """pllComputeSiteParsimony: lengths per expanded (weight-replicated) site of the kept patterns."""
        out = np.zeros(n_sites, dtype=np.int32)
        tot = C.c_int32()
        _chk(load_library().mpf_site_scores(self.h, _p(out), n_sites, C.byref(tot)))
        return out, tot.value

    def spr_scan(self, rec: int, mintrav: int = 1, maxtrav: int = 6, cap: int = 1 << 16):
        q = np.zeros(cap, dtype=np.int32)
        mp = np.zeros(cap, dtype=np.uint32)
        n_p, n_t = C.c_int32(), C.c_int32()
        _chk(load_library().mpf_spr_scan(self.h, rec, mintrav, maxtrav, cap, _p(q), _p(mp), C.byref(n_p), C.byref(n_t)))
        return q[:n_t.value].copy(), mp[:n_t.value].copy(), n_p.value

    def sweep_scan(self, mintrav: int = 1, maxtrav: int = 6):
        n = C.c_uint64()
        m = C.c_uint32()
        _chk(load_library().mpf_spr_sweep_scan(self.h, mintrav, maxtrav, C.byref(n), C.byref(m)))
        return n.value, m.value

    def sweep_costs(self, mintrav: int = 1, maxtrav: int = 6):
        """(n_tests, mp[n_tests], offsets[2n-1]): every insertion test of a whole sweep, prune nodes in sweep order"""
        L = load_library()
        n = C.c_uint64()
        _chk(L.mpf_spr_sweep_costs(self.h, mintrav, maxtrav, 0, None, None, C.byref(n)))
        mp = np.zeros(max(1, n.value), dtype=np.uint32)
        off = np.zeros(2 * self.n - 1, dtype=np.uint64)
        _chk(L.mpf_spr_sweep_costs(self.h, mintrav, maxtrav, n.value, _p(mp), _p(off), C.byref(n)))
        return n.value, mp[:n.value], off

    def node_order(self):
        recs = np.zeros(2 * self.n - 2, dtype=np.int32)
        _chk(load_library().mpf_get_node_order(self.h, _p(recs)))
        return recs

    def optimize_spr(self, mintrav: int = 1, maxtrav: int = 6) -> int:
        s = C.c_uint32()
        _chk(load_library().mpf_optimize_spr(self.h, mintrav, maxtrav, C.byref(s)))
        return s.value

    def make_parsimony_tree(self, seed: int, spr_dist: int) -> int:
        s = C.c_uint32()
        _chk(load_library().mpf_make_parsimony_tree(self.h, seed, spr_dist, C.byref(s)))
        return s.value

    def stepwise_addition(self, seed: int):
        best = np.zeros(self.n + 1, dtype=np.uint32)
        ins = np.zeros(self.n + 1, dtype=np.int32)
        s = C.c_uint32()
        _chk(load_library().mpf_stepwise_addition(self.h, seed, _p(best), _p(ins), C.byref(s)))
        return s.value, best, ins

    def moves(self):
        k = C.c_int32()
        L = load_library()
        _chk(L.mpf_get_moves(self.h, 0, None, None, None, C.byref(k)))
        a = np.zeros(k.value, dtype=np.int32)
        b = np.zeros(k.value, dtype=np.int32)
        s = np.zeros(k.value, dtype=np.uint32)
        if k.value:
            _chk(L.mpf_get_moves(self.h, k.value, _p(a), _p(b), _p(s), C.byref(k)))
        return a, b, s

    # ---- online UFBoot-MP bookkeeping (IQTree::saveCurrentTree during optimize_spr)
    def ufboot_attach(self, samples, epsilon: float = 0.5, shard=None, exchange=None):
        """samples: [n_samples][n_patterns] bootstrap weights (all of them, on every rank).
        shard = (rank, world): multi-GPU online phase -- this engine keeps samples rank, rank + world, ... and
        `exchange` (default: mpboot_amd.shard.event_exchange(), an all-gather over torch.distributed) merges the
        per-batch events of all ranks; every rank must then make the same optimize_spr calls."""
        samples = np.ascontiguousarray(samples, dtype=np.uint16)
        if samples.ndim != 2 or samples.shape[1] != self.P:
            raise ValueError("samples must be [n_samples][n_patterns]")
        self.ufb_B = samples.shape[0]
        if shard is None or shard[1] == 1:
            _chk(load_library().mpf_ufboot_attach(self.h, self.ufb_B, _p(samples), float(epsilon)))
            self._drop_exchange()
            return
        rank, world = shard
        ids = np.arange(rank, self.ufb_B, world, dtype=np.int32)
        local = np.ascontiguousarray(samples[ids])
        if exchange is None:
            from . import shard as _shard
            exchange = _shard.event_exchange()
        self._drop_exchange()
        if isinstance(exchange, RcclComm):       # the library's own exchange: mpf_rccl_exchange with the communicator as its argument
            fn = C.cast(load_library().mpf_rccl_exchange, C.c_void_p)
            _chk(load_library().mpf_ufboot_attach_sharded(self.h, self.ufb_B, len(ids), _p(ids), _p(local), float(epsilon), fn, exchange.h))
            self._ufb_exchange = exchange
            exchange._users += 1                 # (RcclComm.close() refuses while a tracker holds the raw handle)
            return
        self._ufb_exchange = exchange            # keep the ctypes callback alive as long as the tracker
        _chk(load_library().mpf_ufboot_attach_sharded(self.h, self.ufb_B, len(ids), _p(ids), _p(local), float(epsilon),
                                                      C.cast(exchange, C.c_void_p), None))

    def ufboot_refine_sweep(self, maxtrav: int, tie_seeds=None):
        """Batched bootstrap refinement (mpf_ufboot_refine_sweep): the first sweep of the SPR climb from the CURRENT tree under
        every attached sample's weights at once -> (scores[B], stable[B] bool, first_move_visit[B])."""
        B = self.ufb_B
        seeds = None if tie_seeds is None else np.ascontiguousarray(tie_seeds, dtype=np.int32)
        if seeds is not None and len(seeds) != B:
            raise ValueError("one tie seed per attached sample")
        scores = np.zeros(B, dtype=np.uint32)
        stable = np.zeros(B, dtype=np.uint8)
        first = np.zeros(B, dtype=np.int32)
        _chk(load_library().mpf_ufboot_refine_sweep(self.h, maxtrav, None if seeds is None else _p(seeds), _p(scores), _p(stable), _p(first)))
        return scores, stable.astype(bool), first

    def ufboot_detach(self):
        _chk(load_library().mpf_ufboot_detach(self.h))
        self._drop_exchange()

    def ufboot_set_cutoff(self, logl_cutoff: float):
        _chk(load_library().mpf_ufboot_set_cutoff(self.h, float(logl_cutoff)))

    def ufboot_set_ratchet_booking(self, on: bool):
        """False = mpboot's -no_hclimb1_bb: climbs under other weights than the attach-time ones are not booked"""
        _chk(load_library().mpf_ufboot_set_ratchet_booking(self.h, 1 if on else 0))

    def ufboot_set_store_trees(self, on: bool):
        """params->store_candidate_trees (-storetrees, iqtree.cpp:3302-3346); right after the attach"""
        _chk(load_library().mpf_ufboot_set_store_trees(self.h, 1 if on else 0))

    def ufboot_duplicates(self) -> int:
        """duplication_counter: trees that reached saveCurrentTree with a topology booked before (-storetrees)"""
        n = C.c_uint64()
        _chk(load_library().mpf_ufboot_get_duplicates(self.h, C.byref(n)))
        return int(n.value)

    def ufboot_set_mulhits(self, on: bool):
        """params->multiple_hits: the -mulhits update rule (iqtree.cpp:3498-3540); right after the attach"""
        _chk(load_library().mpf_ufboot_set_mulhits(self.h, 1 if on else 0))

    def ufboot_set_topboot(self, n_top: int):
        """params->store_top_boot_trees (-topboot N, with -mulhits; iqtree.cpp:3542-3585)"""
        self.ufb_topboot = int(n_top)
        _chk(load_library().mpf_ufboot_set_topboot(self.h, int(n_top)))

    def ufboot_set_distinct_iter(self, k: int):
        """params->distinct_iter_top_boot (iqtree.cpp:3587-3680); without -mulhits"""
        self.ufb_topboot = int(k)
        _chk(load_library().mpf_ufboot_set_distinct_iter(self.h, int(k)))

    def ufboot_set_iteration(self, cur_it: int):
        _chk(load_library().mpf_ufboot_set_iteration(self.h, int(cur_it)))

    def ufboot_sample_iters(self, sample: int):
        cap = max(int(getattr(self, "ufb_topboot", 0)), 1)
        it = np.zeros(cap, dtype=np.int32)
        n = C.c_int32(0)
        _chk(load_library().mpf_ufboot_get_sample_iters(self.h, int(sample), _p(it), cap, C.byref(n)))
        return [int(x) for x in it[:n.value]]

    def ufboot_sample_top(self, sample: int):
        """([(tree index, rell)...] best first, boot_threshold) of one sample under -mulhits -topboot"""
        cap = max(int(getattr(self, "ufb_topboot", 0)), 1)
        trees = np.zeros(cap, dtype=np.int64)
        rell = np.zeros(cap, dtype=np.int32)
        n, thr = C.c_int32(0), C.c_int32(0)
        _chk(load_library().mpf_ufboot_get_sample_top(self.h, int(sample), _p(trees), _p(rell), cap, C.byref(n), C.byref(thr)))
        return [(int(trees[i]), int(rell[i])) for i in range(n.value)], int(thr.value)

    def ufboot_sample_trees(self, sample: int):
        """boot_trees_parsimony[sample] under -mulhits, sorted"""
        L = load_library()
        n = C.c_int32(0)
        _chk(L.mpf_ufboot_get_sample_trees(self.h, int(sample), None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.int64)
        _chk(L.mpf_ufboot_get_sample_trees(self.h, int(sample), _p(out), n.value, C.byref(n)))
        return [int(x) for x in out[:n.value]]

    def ufboot_set_cutoff_from_btrees(self, on: bool):
        """-cutoff_from_btrees: ufboot_next_cutoff = min over the samples of the logl their tree was booked under"""
        _chk(load_library().mpf_ufboot_set_cutoff_from_btrees(self.h, 1 if on else 0))

    def ufboot_orig_logl(self):
        """boot_tree_orig_logl [n_samples]"""
        out = np.zeros(self.ufb_B, dtype=np.int32)
        _chk(load_library().mpf_ufboot_get_orig_logl(self.h, _p(out)))
        return out

    def ufboot_next_cutoff(self, percent: int = 10) -> float:
        c = C.c_double()
        _chk(load_library().mpf_ufboot_next_cutoff(self.h, percent, C.byref(c)))
        return c.value

    def ufboot_tree_logl(self):
        k = C.c_int64()
        L = load_library()
        _chk(L.mpf_ufboot_num_trees(self.h, C.byref(k)))
        out = np.zeros(k.value, dtype=np.float64)
        if k.value:
            _chk(L.mpf_ufboot_tree_logl(self.h, _p(out)))
        return out

    def ufboot_state(self):
        logl = np.zeros(self.ufb_B, dtype=np.float64)
        counts = np.zeros(self.ufb_B, dtype=np.int32)
        trees = np.zeros(self.ufb_B, dtype=np.int32)
        _chk(load_library().mpf_ufboot_get_state(self.h, _p(logl), _p(counts), _p(trees)))
        return logl, counts, trees

    def ufboot_tree(self, tree_index: int):
        back = np.empty(self.nrec, dtype=np.int32)
        _chk(load_library().mpf_ufboot_get_tree(self.h, int(tree_index), _p(back)))
        return back

    def ufboot_adopt(self, samples, scores, tree_of, trees, lengths) -> int:
        """mpf_ufboot_adopt: books of another search chain of the same run -- sample samples[k] is offered trees[tree_of[k]] at REPS
        length scores[k] and takes it when that is strictly shorter than what it holds.  -> number of samples that took one."""
        sm = np.ascontiguousarray(samples, dtype=np.int32)
        sc = np.ascontiguousarray(scores, dtype=np.uint32)
        to = np.ascontiguousarray(tree_of, dtype=np.int32)
        tr = np.ascontiguousarray(trees, dtype=np.int32).reshape(-1, 3 * (2 * self.n - 1)) if len(trees) else np.zeros((0, 3 * (2 * self.n - 1)), dtype=np.int32)
        ln = np.ascontiguousarray(lengths, dtype=np.uint32)
        assert len(sm) == len(sc) == len(to) and len(tr) == len(ln)
        k = C.c_int32()
        _chk(load_library().mpf_ufboot_adopt(self.h, len(sm), _p(sm), _p(sc), _p(to), len(tr), _p(tr), _p(ln), C.byref(k)))
        return int(k.value)

    def ufboot_counters(self) -> dict:
        d, e, r, ms = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_double()
        _chk(load_library().mpf_ufboot_get_counters(self.h, C.byref(d), C.byref(e), C.byref(r), C.byref(ms)))
        return {"tie_draws": d.value, "events": e.value, "reps_rows": r.value, "reps_kernel_ms": ms.value}

    def stats(self) -> dict:
        st = Stats()
        _chk(load_library().mpf_get_stats(self.h, C.byref(st)))
        return st.as_dict()

    def reset_stats(self):
        _chk(load_library().mpf_reset_stats(self.h))

    def set_option(self, key: str, value: int):
        _chk(load_library().mpf_set_option(self.h, key.encode(), int(value)))

    def scan_trace(self):
        """[workgroups, 4] uint64 timeline of the last planned-program scan (option scan_trace = 1)"""
        L = load_library()
        n = C.c_uint64()
        _chk(L.mpf_get_scan_trace(self.h, None, 0, C.byref(n)))
        out = np.zeros(max(4, n.value), dtype=np.uint64)
        _chk(L.mpf_get_scan_trace(self.h, _p(out), n.value, C.byref(n)))
        return out[:n.value].reshape(-1, 4)

    def get_option(self, key: str) -> int:
        v = C.c_int64(0)
        _chk(load_library().mpf_get_option(self.h, key.encode(), C.byref(v)))
        return int(v.value)

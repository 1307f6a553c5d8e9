"""ctypes binding of oracle/_build/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (mpboot_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liboracle.so")

DNA, AA, BIN, GENERIC = 0, 1, 2, 3
TIE_FIRST, TIE_RANDOM = 0, 1


def build(force: bool = False) -> str:
    src = [os.path.join(HERE, f) for f in ("fitch_oracle.c", "fitch_oracle.h", "rng.h", "sankoff_oracle.c")]
    src = [s for s in src if os.path.exists(s)]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, ci, cu = C.c_void_p, C.c_int, C.c_uint
        L.orc_create.restype = vp
        L.orc_create.argtypes = [ci, ci, ci, vp, vp, ci]
        L.orc_create_sankoff.restype = vp
        L.orc_create_sankoff.argtypes = [ci, ci, ci, vp, vp, ci, vp]
        L.orc_destroy.argtypes = [vp]
        for f in ("orc_words", "orc_states", "orc_num_informative", "orc_trace_len", "orc_moves_len"):
            getattr(L, f).restype = ci
            getattr(L, f).argtypes = [vp]
        L.orc_informative.restype = C.POINTER(ci)
        L.orc_informative.argtypes = [vp]
        L.orc_node_vector.restype = C.POINTER(C.c_uint32)
        L.orc_node_vector.argtypes = [vp, ci]
        L.orc_set_weights.argtypes = [vp, vp]
        L.orc_enable_persite.argtypes = [vp, ci]
        L.orc_set_tree.argtypes = [vp, vp]
        L.orc_get_tree.argtypes = [vp, vp]
        L.orc_reset_nodep.argtypes = [vp]
        L.orc_get_nodep.argtypes = [vp, vp]
        L.orc_node_rectifier.argtypes = [vp]
        L.orc_evaluate.restype = cu
        L.orc_evaluate.argtypes = [vp, ci, ci]
        L.orc_score_tree.restype = cu
        L.orc_score_tree.argtypes = [vp]
        L.orc_pattern_scores.restype = ci
        L.orc_pattern_scores.argtypes = [vp, vp]
        L.orc_site_scores.restype = ci
        L.orc_site_scores.argtypes = [vp, vp, ci]
        L.orc_seed_ties.argtypes = [vp, ci, ci]
        L.orc_set_tie_state.argtypes = [vp, C.c_ulonglong]
        L.orc_get_tie_state.restype = C.c_ulonglong
        L.orc_get_tie_state.argtypes = [vp]
        L.orc_set_pre_evaluate.argtypes = [vp, ci]
        L.orc_set_max_visits.argtypes = [vp, C.c_long]
        L.orc_trace.argtypes = [vp, ci]
        L.orc_trace_get.argtypes = [vp, vp, vp]
        L.orc_moves_get.argtypes = [vp, vp, vp, vp]
        L.orc_rearrange.restype = ci
        L.orc_rearrange.argtypes = [vp, ci, ci, ci]
        L.orc_set_best.argtypes = [vp, cu]
        L.orc_get_best.restype = cu
        L.orc_get_best.argtypes = [vp, vp, vp]
        L.orc_optimize_spr.restype = cu
        L.orc_optimize_spr.argtypes = [vp, ci, ci]
        L.orc_make_tree.restype = cu
        L.orc_make_tree.argtypes = [vp, C.c_long, ci, vp]
        L.orc_stepwise.restype = cu
        L.orc_stepwise.argtypes = [vp, C.c_long, vp, vp]
        L.orc_counters.argtypes = [vp, vp, vp, vp]
        L.orc_ufboot_attach.argtypes = [vp, ci, vp, C.c_double]
        L.orc_ufboot_detach.argtypes = [vp]
        L.orc_ufboot_set_cutoff.argtypes = [vp, C.c_double]
        L.orc_ufboot_set_ratchet_booking.argtypes = [vp, ci]
        L.orc_ufboot_set_mulhits.argtypes = [vp, ci]
        L.orc_ufboot_set_store_trees.argtypes = [vp, ci]
        L.orc_ufboot_duplicates.argtypes = [vp]
        L.orc_ufboot_duplicates.restype = ci
        L.orc_ufboot_set_topboot.argtypes = [vp, ci]
        L.orc_ufboot_set_distinct_iter.argtypes = [vp, ci]
        L.orc_ufboot_set_iteration.argtypes = [vp, ci]
        L.orc_ufboot_sample_iters.restype = ci
        L.orc_ufboot_sample_iters.argtypes = [vp, ci, vp]
        L.orc_ufboot_sample_top.restype = ci
        L.orc_ufboot_sample_top.argtypes = [vp, ci, vp, vp, vp]
        L.orc_ufboot_sample_trees.restype = ci
        L.orc_ufboot_sample_trees.argtypes = [vp, ci, vp, ci]
        L.orc_ufboot_ntrees.restype = ci
        L.orc_ufboot_ntrees.argtypes = [vp]
        L.orc_ufboot_bad.restype = ci
        L.orc_ufboot_bad.argtypes = [vp]
        L.orc_ufboot_draws.restype = C.c_ulonglong
        L.orc_ufboot_draws.argtypes = [vp]
        L.orc_ufboot_tree_logl.argtypes = [vp, vp]
        L.orc_ufboot_state.argtypes = [vp, vp, vp, vp]
        L.orc_ufboot_adopt.restype = ci
        L.orc_ufboot_adopt.argtypes = [vp, ci, vp, vp, vp, ci, vp, vp]
        L.orc_ufboot_tree.restype = ci
        L.orc_ufboot_tree.argtypes = [vp, ci, vp]
        L.orc_ufboot_next_cutoff.restype = C.c_double
        L.orc_ufboot_next_cutoff.argtypes = [vp, ci]
        L.orc_ufboot_set_cutoff_from_btrees.argtypes = [vp, ci]
        L.orc_ufboot_orig_logl.argtypes = [vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One alignment packed the reference's way; mirrors the PLL instance state."""

    def __init__(self, codes: np.ndarray, weights=None, datatype: int = DNA, keep_all: bool = False, cost=None):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        self.n, self.P = codes.shape
        if weights is None:
            weights = np.ones(self.P, dtype=np.int32)
        weights = np.ascontiguousarray(weights, dtype=np.int32)
        self._w = weights.copy()
        if cost is None:
            self.h = lib().orc_create(self.n, self.P, datatype, _p(codes), _p(weights), int(keep_all))
        else:
            cost = np.ascontiguousarray(cost, dtype=np.uint32)
            self.h = lib().orc_create_sankoff(self.n, self.P, datatype, _p(codes), _p(weights), int(keep_all), _p(cost))
        self.nrec = 3 * (2 * self.n - 1)

    def __del__(self):
        try:
            if self.h:
                lib().orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def W(self):
        return lib().orc_words(self.h)

    @property
    def S(self):
        return lib().orc_states(self.h)

    @property
    def num_informative(self):
        return lib().orc_num_informative(self.h)

    def informative(self):
        return np.ctypeslib.as_array(lib().orc_informative(self.h), shape=(self.P,)).copy()

    def node_vector(self, node: int):
        return np.ctypeslib.as_array(lib().orc_node_vector(self.h, node), shape=(self.S, self.W)).copy()

    def set_weights(self, w):
        w = np.ascontiguousarray(w, dtype=np.int32)
        self._w = w.copy()
        lib().orc_set_weights(self.h, _p(w))

    def enable_persite(self, on=True):
        lib().orc_enable_persite(self.h, int(on))

    def set_tree(self, back):
        back = np.ascontiguousarray(back, dtype=np.int32)
        assert len(back) == self.nrec
        lib().orc_set_tree(self.h, _p(back))

    def get_tree(self):
        back = np.empty(self.nrec, dtype=np.int32)
        lib().orc_get_tree(self.h, _p(back))
        return back

    def reset_nodep(self):
        lib().orc_reset_nodep(self.h)

    def nodep(self):
        a = np.zeros(2 * self.n, dtype=np.int32)
        lib().orc_get_nodep(self.h, _p(a))
        return a

    def node_rectifier(self):
        lib().orc_node_rectifier(self.h)

    def evaluate(self, rec: int, full: bool = False) -> int:
        return int(lib().orc_evaluate(self.h, rec, int(full)))

    def score_tree(self, back=None) -> int:
        if back is not None:
            self.set_tree(back)
        return int(lib().orc_score_tree(self.h))

    def pattern_scores(self):
        out = np.zeros(self.P, dtype=np.uint16)
        total = lib().orc_pattern_scores(self.h, _p(out))
        return out, int(total)

    def site_scores(self, nsite: int):
        out = np.zeros(nsite, dtype=np.int32)
        total = lib().orc_site_scores(self.h, _p(out), nsite)
        return out, int(total)

    def seed_ties(self, mode: int, seed: int = 1):
        lib().orc_seed_ties(self.h, mode, seed)

    def set_tie_state(self, state: int):
        lib().orc_set_tie_state(self.h, state & ((1 << 64) - 1))

    def tie_state(self) -> int:
        return int(lib().orc_get_tie_state(self.h))

    def set_pre_evaluate(self, mode: int):
        lib().orc_set_pre_evaluate(self.h, int(mode))

    def set_max_visits(self, k: int):
        """test aid: optimize_spr returns behind k prune-node visits (0 = no limit)"""
        lib().orc_set_max_visits(self.h, int(k))

    def trace(self, on=True):
        lib().orc_trace(self.h, int(on))

    def get_trace(self):
        k = lib().orc_trace_len(self.h)
        q = np.zeros(k, dtype=np.int32)
        mp = np.zeros(k, dtype=np.uint32)
        if k:
            lib().orc_trace_get(self.h, _p(q), _p(mp))
        return q, mp

    def get_moves(self):
        k = lib().orc_moves_len(self.h)
        a = np.zeros(k, dtype=np.int32)
        b = np.zeros(k, dtype=np.int32)
        s = np.zeros(k, dtype=np.uint32)
        if k:
            lib().orc_moves_get(self.h, _p(a), _p(b), _p(s))
        return a, b, s

    def rearrange(self, rec: int, mintrav: int = 1, maxtrav: int = 6) -> int:
        return lib().orc_rearrange(self.h, rec, mintrav, maxtrav)

    def set_best(self, best: int):
        lib().orc_set_best(self.h, best)

    def get_best(self):
        r, i = C.c_int(), C.c_int()
        b = lib().orc_get_best(self.h, C.byref(r), C.byref(i))
        return int(b), r.value, i.value

    def optimize_spr(self, mintrav: int = 1, maxtrav: int = 6) -> int:
        return int(lib().orc_optimize_spr(self.h, mintrav, maxtrav))

    def make_tree(self, seed: int, spr_dist: int):
        perm = np.zeros(self.n + 1, dtype=np.int32)
        s = lib().orc_make_tree(self.h, seed, spr_dist, _p(perm))
        return int(s), perm

    def stepwise(self, seed: int):
        best = np.zeros(self.n + 1, dtype=np.uint32)
        ins = np.zeros(self.n + 1, dtype=np.int32)
        s = lib().orc_stepwise(self.h, seed, _p(best), _p(ins))
        return int(s), best, ins

    # ---- UFBoot-MP online bookkeeping (IQTree::saveCurrentTree)
    def ufboot_attach(self, samples, epsilon: float = 0.5):
        samples = np.ascontiguousarray(samples, dtype=np.uint16)
        assert samples.ndim == 2 and samples.shape[1] == self.P
        self.ufb_B = samples.shape[0]
        lib().orc_ufboot_attach(self.h, self.ufb_B, _p(samples), float(epsilon))

    def ufboot_detach(self):
        lib().orc_ufboot_detach(self.h)

    def ufboot_set_ratchet_booking(self, on: bool):
        lib().orc_ufboot_set_ratchet_booking(self.h, 1 if on else 0)

    def ufboot_set_store_trees(self, on: bool):
        lib().orc_ufboot_set_store_trees(self.h, 1 if on else 0)

    def ufboot_duplicates(self) -> int:
        return int(lib().orc_ufboot_duplicates(self.h))

    def ufboot_set_mulhits(self, on: bool):
        lib().orc_ufboot_set_mulhits(self.h, 1 if on else 0)

    def ufboot_set_topboot(self, n_top: int):
        self.ufb_topboot = int(n_top)
        lib().orc_ufboot_set_topboot(self.h, int(n_top))

    def ufboot_set_distinct_iter(self, k: int):
        self.ufb_topboot = int(k)
        lib().orc_ufboot_set_distinct_iter(self.h, int(k))

    def ufboot_set_iteration(self, cur_it: int):
        lib().orc_ufboot_set_iteration(self.h, int(cur_it))

    def ufboot_sample_iters(self, sample: int):
        it = np.zeros(max(self.ufb_topboot, 1), dtype=np.int32)
        n = lib().orc_ufboot_sample_iters(self.h, int(sample), _p(it))
        return [int(x) for x in it[:n]]

    def ufboot_sample_top(self, sample: int):
        """boot_trees_parsimony_top[sample] under -mulhits -topboot N: ([(tree index, rell)...] best first, boot_threshold)"""
        idx = np.zeros(max(self.ufb_topboot, 1), dtype=np.int32)
        rell = np.zeros(max(self.ufb_topboot, 1), dtype=np.int32)
        thr = C.c_int(0)
        n = lib().orc_ufboot_sample_top(self.h, int(sample), _p(idx), _p(rell), C.byref(thr))
        return [(int(idx[i]), int(rell[i])) for i in range(n)], int(thr.value)

    def ufboot_sample_trees(self, sample: int):
        """boot_trees_parsimony[sample] under -mulhits, sorted"""
        n = lib().orc_ufboot_sample_trees(self.h, int(sample), _p(np.zeros(1, dtype=np.int32)), 0)
        out = np.zeros(max(n, 1), dtype=np.int32)
        lib().orc_ufboot_sample_trees(self.h, int(sample), _p(out), n)
        return sorted(int(x) for x in out[:n])

    def ufboot_set_cutoff(self, logl_cutoff: float):
        lib().orc_ufboot_set_cutoff(self.h, float(logl_cutoff))

    def ufboot_state(self):
        logl = np.zeros(self.ufb_B, dtype=np.float64)
        counts = np.zeros(self.ufb_B, dtype=np.int32)
        trees = np.zeros(self.ufb_B, dtype=np.int32)
        lib().orc_ufboot_state(self.h, _p(logl), _p(counts), _p(trees))
        return logl, counts, trees

    def ufboot_tree_logl(self):
        out = np.zeros(lib().orc_ufboot_ntrees(self.h), dtype=np.float64)
        if len(out):
            lib().orc_ufboot_tree_logl(self.h, _p(out))
        return out

    def ufboot_tree(self, tree_index: int):
        back = np.empty(self.nrec, dtype=np.int32)
        ok = lib().orc_ufboot_tree(self.h, int(tree_index), _p(back))
        return back if ok else None

    def ufboot_adopt(self, samples, scores, tree_of, trees, lengths) -> int:
        sm = np.ascontiguousarray(samples, dtype=np.int32)
        sc = np.ascontiguousarray(scores, dtype=np.uint32)
        to = np.ascontiguousarray(tree_of, dtype=np.int32)
        tr = np.ascontiguousarray(trees, dtype=np.int32).reshape(-1, self.nrec) if len(trees) else np.zeros((0, self.nrec), dtype=np.int32)
        ln = np.ascontiguousarray(lengths, dtype=np.uint32)
        return int(lib().orc_ufboot_adopt(self.h, len(sm), _p(sm), _p(sc), _p(to), len(tr), _p(tr), _p(ln)))

    def weights(self):
        return self._w.copy()

    def reset_node_order(self):
        self.reset_nodep()

    def ufboot_bad(self):
        return lib().orc_ufboot_bad(self.h)

    def ufboot_draws(self):
        return int(lib().orc_ufboot_draws(self.h))

    def ufboot_set_cutoff_from_btrees(self, on: bool):
        lib().orc_ufboot_set_cutoff_from_btrees(self.h, 1 if on else 0)

    def ufboot_orig_logl(self):
        out = np.zeros(self.ufb_B, dtype=np.int32)
        lib().orc_ufboot_orig_logl(self.h, _p(out))
        return out

    def ufboot_next_cutoff(self, percent: int = 10):
        return float(lib().orc_ufboot_next_cutoff(self.h, percent))

    def counters(self):
        a, b, c = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
        lib().orc_counters(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value


def randum_advance(seed: int, k: int) -> int:
    """tr->randomNumberSeed after k calls of PLL's randum() (pllrepo/src/utils.c:335-358; restated in oracle/rng.h, here once
    more in Python): what makePermutationFast leaves behind"""
    for _ in range(k):
        s0, s1, s2 = seed & 4095, (seed >> 12) & 4095, (seed >> 24) & 255
        t = 1549 * s0
        n0 = t & 4095
        t = (t >> 12) + 1549 * s1 + 406 * s0
        n1 = t & 4095
        t = (t >> 12) + 1549 * s2 + 406 * s1
        seed = ((t & 255) << 24) | (n1 << 12) | n0
    return seed


def lcg64_doubles(seed: int, k: int):
    """First k values of the restated SPRNG stream (oracle/rng.h) -- computed in Python ints."""
    mult = (0x27BB2EE6 << 32) | 0x87B0B0FD
    prime = 3037000493
    state = ((0x2BC6FFFF << 32) | 0x8CFE166D) ^ ((seed << 33) & 0xFFFFFFFFFFFFFFFF)
    out = []
    for _ in range(k):
        state = (state * mult + prime) & 0xFFFFFFFFFFFFFFFF
        out.append(float(state) * 5.4210108624275222e-20)
    return out

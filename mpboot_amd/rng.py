"""The reference's tie-break / resampling generator in vectorised form (host side, numpy).

random_double() of the reference = SPRNG 64-bit LCG, stream 0 of 1 (tools.cpp:3320-3368, sprng/lcg64.c:199-268):
    state <- state * 0x27bb2ee687b0b0fd + 3037000493 (mod 2^64);   value = state * 2^-64
Because the recurrence is affine, the k-th state is A_k * s0 + p * G_k with A_k = a^k and G_k = 1 + a + ... + a^(k-1);
both sequences are built by doubling, so n draws cost O(n) numpy operations instead of a Python loop.
"""
from __future__ import annotations

import numpy as np

_A = np.uint64(0x27BB2EE687B0B0FD)
_P = np.uint64(3037000493)
_TWO_M64 = 5.4210108624275222e-20


class Lcg64:
    def __init__(self, seed: int):
        self.state = np.uint64(((0x2BC6FFFF << 32) | 0x8CFE166D) ^ ((seed << 33) & 0xFFFFFFFFFFFFFFFF))

    def doubles(self, n: int) -> np.ndarray:
        """the next n values of random_double()"""
        if n <= 0:
            return np.zeros(0)
        with np.errstate(over="ignore"):
            A = np.empty(n, dtype=np.uint64)
            G = np.empty(n, dtype=np.uint64)
            A[0], G[0] = _A, np.uint64(1)
            m = 1
            while m < n:
                k = min(m, n - m)
                A[m:m + k] = A[m - 1] * A[:k]
                G[m:m + k] = G[m - 1] + A[m - 1] * G[:k]
                m += k
            states = A * self.state + _P * G
        self.state = states[-1]
        return states.astype(np.float64) * _TWO_M64

    def ints(self, n: int, bound: int) -> np.ndarray:
        """random_int(bound) = floor(random_double() * bound), n times (tools.cpp:3351-3353)"""
        return np.floor(self.doubles(n) * bound).astype(np.int64)

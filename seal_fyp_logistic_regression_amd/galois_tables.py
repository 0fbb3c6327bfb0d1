"""Host-side Galois index tables: out[i] = in[table[i]] applies X -> X^g in the NTT domain."""
from __future__ import annotations

import numpy as np


def bitrev(x: np.ndarray, bits: int) -> np.ndarray:
    x = x.astype(np.uint32)
    r = np.zeros_like(x)
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x = x >> 1
    return r


def gather_table(N: int, galois_elt: int) -> np.ndarray:
    logn = N.bit_length() - 1
    i = np.arange(N, dtype=np.uint64)
    raw = (np.uint64(galois_elt) * (2 * bitrev(i, logn).astype(np.uint64) + 1)) & np.uint64(2 * N - 1)
    return bitrev(((raw - 1) >> np.uint64(1)).astype(np.uint32), logn).astype(np.int64)

#!/usr/bin/env python3
"""Latency of small key-switch batches (n = 1..8 rotations per call, back to back) on the LR parameter set's ring
(N=16384, {60,40x7,60}) at several levels; HEFX_QUARTER=0/1 forces the split-2 / quarter-row path (development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seal_fyp_logistic_regression_amd import Engine
from oracle import oracle as O
N = 16384
primes = O.coeff_modulus_create(N, [60, 40, 40, 40, 40, 40, 40, 40, 60])
e = Engine(N, primes)
o = O.Oracle(N, primes)
k = len(primes)
key = e.to_device(o.uniform(k, 2 * (k - 1), 2).reshape(k - 1, 2, k, N))
NS = tuple(int(x) for x in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 2, 4, 8)
for L in (2, 3, 4, 5, 8):
    row = []
    for n in NS:
        cts = [e.to_device(o.uniform(L, 2, i)) for i in range(n)]
        outs = e.empty_many(n, (2, L, N))
        for _ in range(5):
            e.apply_galois_batch(L, cts, [3] * n, [key] * n, outs=outs)
        e.sync()
        t = time.perf_counter()
        for _ in range(100):
            e.apply_galois_batch(L, cts, [3] * n, [key] * n, outs=outs)
        e.sync()
        row.append("n=%d %.1f" % (n, (time.perf_counter() - t) / 100 * 1e6))
    print("L=%d us per batch: %s" % (L, "  ".join(row)))

#!/usr/bin/env python3
"""Bench unit (rotate step 1 + multiply_plain) at the batch sizes SURVEY.md 8(d) lists, B in {1, 16, 256, 1024},
for the C3 parameter set: median of the per-step times.  Prints one JSON object."""
import json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from seal_fyp_logistic_regression_amd import Engine

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
N, primes = bench.SETS[name]
k, L = len(primes), len(primes) - 1
e = Engine(N, primes)
rng = np.random.default_rng(0)
key = e.to_device(bench.synth(rng, primes, N, L, 2, k))
out = {"set": name, "N": N, "L": L, "runs": []}
for B in (1, 16, 256, 1024):
    ct = e.empty(B, 2, L, N); pt = e.empty(B, L, N); o = e.empty(B, 2, L, N)
    for base in range(0, B, 32):
        cnt = min(32, B - base)
        ct.view(base * 2 * L * N, (cnt, 2, L, N)).upload(bench.synth(rng, primes, N, cnt, 2, L))
        pt.view(base * L * N, (cnt, L, N)).upload(bench.synth(rng, primes, N, cnt, L))
    cts = [ct.view(i * 2 * L * N, (2, L, N)) for i in range(B)]
    pts = [pt.view(i * L * N, (L, N)) for i in range(B)]
    outs = [o.view(i * 2 * L * N, (2, L, N)) for i in range(B)]
    for _ in range(3):
        e.rotate_multiply_plain_batch(L, cts, [3] * B, [key] * B, pts, outs)
    e.sync()
    ts = []
    reps = 40 if B <= 16 else 20
    for _ in range(reps):
        t = time.perf_counter()
        e.rotate_multiply_plain_batch(L, cts, [3] * B, [key] * B, pts, outs)
        e.sync()
        ts.append(time.perf_counter() - t)
    med = statistics.median(ts)
    out["runs"].append({"batch": B, "median_ms": med * 1e3, "ops_per_s": B / med,
                        "algorithmic_GBps": B * bench.algorithmic_bytes_per_op(N, L) / med / 1e9})
    print(out["runs"][-1], flush=True)
print(json.dumps(out))

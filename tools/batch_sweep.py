#!/usr/bin/env python3
"""Bench unit (rotate step 1 + multiply_plain) at the batch sizes SURVEY.md 8(d) lists, B in {1, 16, 256, 1024}
(plus the headline's 4608), for one parameter set: median of the per-call times (call + wait), inputs drawn on the
device.  Prints one JSON object.   usage: tools/batch_sweep.py [C3] > profiles/r03_batch_sweep_C3.json"""
import hashlib, json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from seal_fyp_logistic_regression_amd import Engine

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
N, primes = bench.SETS[name]
k, L = len(primes), len(primes) - 1
e = Engine(N, primes)
key32 = lambda tag: hashlib.sha256(f"batch-sweep:{tag}".encode()).digest()
key = e.sample("uniform", key32("key"), 3, 2 * L, k, 0).view(0, (L, 2, k, N))
out = {"set": name, "N": N, "L": L, "unit": "rotate_vector(step 1, direct key) + multiply_plain",
       "algorithmic_bytes_per_op": bench.algorithmic_bytes_per_op(N, L), "runs": []}
BATCHES = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 16, 256, 1024, 4608)
for B in BATCHES:
    ct = e.sample("uniform", key32(f"ct{B}"), 1, 2 * B, L, 0)
    pt = e.sample("uniform", key32(f"pt{B}"), 2, B, L, 0)
    o = e.empty(B, 2, L, N)
    cts = [ct.view(i * 2 * L * N, (2, L, N)) for i in range(B)]
    pts = [pt.view(i * L * N, (L, N)) for i in range(B)]
    outs = [o.view(i * 2 * L * N, (2, L, N)) for i in range(B)]
    for _ in range(3):
        e.rotate_multiply_plain_batch(L, cts, [3] * B, [key] * B, pts, outs)
    e.sync()
    ts = []
    reps = 40 if B <= 16 else (20 if B <= 1024 else 8)
    for _ in range(reps):
        t = time.perf_counter()
        e.rotate_multiply_plain_batch(L, cts, [3] * B, [key] * B, pts, outs)
        e.sync()
        ts.append(time.perf_counter() - t)
    med = statistics.median(ts)
    out["runs"].append({"batch": B, "median_ms": med * 1e3, "us_per_op": med / B * 1e6, "ops_per_s": B / med,
                        "algorithmic_GBps": B * bench.algorithmic_bytes_per_op(N, L) / med / 1e9,
                        "frac_of_8TBps": B * bench.algorithmic_bytes_per_op(N, L) / med / 8e12})
    print(out["runs"][-1], file=sys.stderr, flush=True)
    del ct, pt, o, cts, pts, outs
print(json.dumps(out, indent=1))

import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from seal_fyp_logistic_regression_amd import Engine
from oracle import oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
N, bits = (16384, [60, 40, 40, 40, 40, 60]) if name == "C3" else (8192, [60, 40, 40, 60])
primes = O.coeff_modulus_create(N, bits)
e = Engine(N, primes); o = O.Oracle(N, primes); k = len(primes); L = k - 1
ct = e.to_device(o.uniform(L, 2, 1)); key = e.to_device(o.uniform(k, 2 * L, 2).reshape(L, 2, k, N)); out = e.empty(2, L, N)
for _ in range(5): e.apply_galois(L, ct, 3, key, out=out)
e.sync()
t = time.perf_counter()
for _ in range(200): e.apply_galois(L, ct, 3, key, out=out)
e.sync()
print("single key switch: %.1f us per op (200 back-to-back, async)" % ((time.perf_counter() - t) / 200 * 1e6))

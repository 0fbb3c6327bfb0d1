import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from seal_fyp_logistic_regression_amd import Engine
from oracle import oracle as O
N, bits = 8192, [60, 40, 40, 60]
primes = O.coeff_modulus_create(N, bits)
t = time.perf_counter(); e = Engine(N, primes); e.sync(); print("context create %.1f ms" % ((time.perf_counter() - t) * 1e3))
o = O.Oracle(N, primes)
ct = e.to_device(o.uniform(3, 2, 1)); key = e.to_device(o.uniform(4, 6, 2).reshape(3, 2, 4, N)); e.sync()
for i in range(3):
    t = time.perf_counter(); r = e.apply_galois(3, ct, 3, key); e.sync(); print("apply_galois call %d: %.2f ms" % (i, (time.perf_counter() - t) * 1e3))
t = time.perf_counter(); r = e.apply_galois(3, ct, 9, key); e.sync(); print("apply_galois new element: %.2f ms" % ((time.perf_counter() - t) * 1e3))
cts = [ct] * 200
t = time.perf_counter(); r = e.apply_galois_batch(3, cts, [3] * 200, [key] * 200); e.sync(); print("batch 200 first: %.2f ms" % ((time.perf_counter() - t) * 1e3))
t = time.perf_counter(); r = e.apply_galois_batch(3, cts, [3] * 200, [key] * 200); e.sync(); print("batch 200 second: %.2f ms" % ((time.perf_counter() - t) * 1e3))
v = np.linspace(0, 1, N // 2)
for i in range(3):
    t = time.perf_counter(); p = e.ckks_encode(3, v, 2.0 ** 40); e.sync(); print("encode call %d: %.2f ms" % (i, (time.perf_counter() - t) * 1e3))
for i in range(2):
    t = time.perf_counter(); w = e.ckks_decode(3, p, 2.0 ** 40); print("decode call %d: %.2f ms" % (i, (time.perf_counter() - t) * 1e3))

"""Start-up costs of a process that uses the engine once: loading the library, creating the context (HIP init, tables), the
first key switch (code-object load of the key-switch kernels), the second one.  python tools/first_call_costs2.py [N=16384]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
import numpy as np
t1 = time.perf_counter()
from seal_fyp_logistic_regression_amd import Engine, capi
capi.lib()
t2 = time.perf_counter()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
primes = {16384: [0xffffffffffd8001, 0xffffb20001, 0xffffc40001, 0xffffca8001, 0xffffe80001, 0xffffffffffe8001]}.get(N)
if primes is None:
    from seal_fyp_logistic_regression_amd import seal as S
    primes = S.CoeffModulus.Create(N, [60, 40, 40, 60])
e = Engine(N, primes)
e.sync()
t3 = time.perf_counter()
L, k = len(primes) - 1, len(primes)
key = e.sample("uniform", bytes(32), 1, 2 * (k - 1), k, 0)
ct = e.sample("uniform", bytes([1] * 32), 2, 2, L, 0)
e.sync()
t4 = time.perf_counter()
out = e.apply_galois(L, ct, 3, key)
e.sync()
t5 = time.perf_counter()
out = e.apply_galois(L, ct, 3, key)
e.sync()
t6 = time.perf_counter()
print(f"N={N}: numpy {1e3*(t1-t0):.0f} ms, dlopen libhefx {1e3*(t2-t1):.0f} ms, context (HIP init + tables) {1e3*(t3-t2):.0f} ms, "
      f"first sample kernels {1e3*(t4-t3):.0f} ms, FIRST key switch {1e3*(t5-t4):.1f} ms, second {1e3*(t6-t5):.2f} ms")

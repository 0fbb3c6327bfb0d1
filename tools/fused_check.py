"""Bit-exactness of the key-switch paths at C3 (N=16384, L=5 and L=3) against the CPU oracle, for A/B builds that only
instantiate one ring size (tools/build_variant.sh ... -DHEFX_ONLY_LOGN=14) and for the HEFX_FUSED modes:
    HEFX_FUSED=3 HEFX_LIB=build/libhefx_x.so python tools/fused_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from seal_fyp_logistic_regression_amd import Engine

N, primes = 16384, [0xffffffffffd8001, 0xffffb20001, 0xffffc40001, 0xffffca8001, 0xffffe80001, 0xffffffffffe8001]
o, e = O.Oracle(N, primes), Engine(N, primes)
k = len(primes)
rng = np.random.default_rng(7)
bad = 0
for L, n in ((k - 1, 37), (3, 9), (k - 1, 1)):
    keys = [o.uniform(k, 2 * (k - 1), 50 + i).reshape(k - 1, 2, k, N) for i in range(3)]
    dkeys = [e.to_device(x) for x in keys]
    cts = [o.uniform(L, 2, 100 + i) for i in range(n)]
    pts = [o.uniform(L, 1, 200 + i)[0] for i in range(n)]
    elts = [int(2 * rng.integers(1, N) + 1) for _ in range(n)]
    ki = sorted(int(rng.integers(3)) for _ in range(n))
    outs = e.rotate_multiply_plain_batch(L, [e.to_device(c) for c in cts], elts, [dkeys[j] for j in ki],
                                         [e.to_device(p) for p in pts])
    for i in range(n):
        ok = bool((outs[i].download() == o.rotate_mulplain(cts[i], elts[i], keys[ki[i]], pts[i])).all())
        bad += not ok
    c3 = o.multiply(cts[0], cts[0])
    got = e.relinearize(L, e.to_device(c3), dkeys[0]).download()
    bad += not bool((got == o.relinearize(c3, keys[0])).all())
print("fused_check:", "ALL BIT-EXACT" if bad == 0 else f"{bad} MISMATCHES", f"(HEFX_FUSED={os.environ.get('HEFX_FUSED', '')}, lib={os.environ.get('HEFX_LIB', 'default')})")
sys.exit(1 if bad else 0)

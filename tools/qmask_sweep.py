#!/usr/bin/env python3
"""Every per-launch shape combination of the five-launch small-batch key switch (HEFX_QMASK = bit 0 digit inverses, 1 digit
transforms, 2 mod-down inverse, 3 mod-down finish on quarter-row workgroups; HEFX_PAIR=0) over a grid of batch sizes and
levels on the LR ring (N = 16384, {60,40x7,60}): one child process per mask (the rule is read once per process), us per
apply_galois_batch call.  Prints, per (L, n), the best mask, its time and the time of the default rule.
    python tools/qmask_sweep.py [n,n,...] [L,L,...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NS = sys.argv[1] if len(sys.argv) > 1 else "6,8,10,12,16,20,24,32"
LS = sys.argv[2] if len(sys.argv) > 2 else "3,4,5,8"
CHILD = r'''
import sys, time, os, json
sys.path.insert(0, %r)
from seal_fyp_logistic_regression_amd import Engine
from oracle import oracle as O
N = 16384
primes = O.coeff_modulus_create(N, [60, 40, 40, 40, 40, 40, 40, 40, 60])
e = Engine(N, primes); o = O.Oracle(N, primes); k = len(primes)
key = e.to_device(o.uniform(k, 2 * (k - 1), 2).reshape(k - 1, 2, k, N))
out = {}
for L in [int(x) for x in %r.split(",")]:
    for n in [int(x) for x in %r.split(",")]:
        cts = [e.to_device(o.uniform(L, 2, i)) for i in range(n)]
        outs = e.empty_many(n, (2, L, N))
        for _ in range(5):
            e.apply_galois_batch(L, cts, [3] * n, [key] * n, outs=outs)
        e.sync()
        t = time.perf_counter()
        for _ in range(60):
            e.apply_galois_batch(L, cts, [3] * n, [key] * n, outs=outs)
        e.sync()
        out["%%d,%%d" %% (L, n)] = (time.perf_counter() - t) / 60 * 1e6
print(json.dumps(out))
''' % (ROOT, LS, NS)
res = {}
for mask in ["default"] + list(range(16)):
    env = dict(os.environ, HEFX_PAIR="0")
    if mask != "default":
        env["HEFX_QMASK"] = str(mask)
    else:
        env.pop("HEFX_PAIR")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    res[mask] = json.loads(r.stdout.strip().splitlines()[-1])
    print(mask, " ".join("%s=%.1f" % kv for kv in res[mask].items()), flush=True)
print("\n(L,n): best mask / us  |  default rule us")
for key in res["default"]:
    best = min(range(16), key=lambda m: res[m][key])
    print("%-7s mask %2d %6.1f | %6.1f" % (key, best, res[best][key], res["default"][key]))

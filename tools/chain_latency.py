"""Latency per level of the rotate-by-1 + add_inplace chains of helper.h:472-476 at the LR driver's shape (N = 16384,
{60,40x7,60}; after the dot product's rescale the chains run at L = 7 .. 2) -- one engine call for the whole chain
(hefx_rotate_add_chain) against one fused call per level (hefx_apply_galois_add_batch) and the round-3 sequence
(hefx_apply_galois_batch + hefx_add per level).  HEFX_CHAIN_GRAPH=0 switches the graph replay off.
    python tools/chain_latency.py [n=8] [L=2] [steps=500]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from seal_fyp_logistic_regression_amd import Engine
from seal_fyp_logistic_regression_amd.seal import galois_elt_from_step

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 500
N = 16384
primes = O.coeff_modulus_create(N, [60] + [40] * 7 + [60])
e = Engine(N, primes)
k = len(primes)
key = e.sample("uniform", bytes(range(32)), 1, 2 * (k - 1), k, 0)
cts = [e.sample("uniform", bytes(range(32)), 10 + i, 2, L, 0) for i in range(n)]
accs = [e.sample("uniform", bytes(range(32)), 50 + i, 2, L, 0) for i in range(n)]
elt = galois_elt_from_step(1, N)


def timed(fn):
    fn(); e.sync()
    t = time.perf_counter(); fn(); e.sync()
    return (time.perf_counter() - t) / steps * 1e6


def chain():
    e.rotate_add_chain(L, cts, [elt] * n, [key] * n, accs, steps)


def fused_levels():
    d, a = cts, accs
    for _ in range(steps):
        d, a = e.apply_galois_add_batch(L, d, [elt] * n, [key] * n, a)


def separate_levels():
    d, a = cts, accs
    for _ in range(steps):
        d = e.apply_galois_batch(L, d, [elt] * n, [key] * n)
        a = e.add_batch(L, 2, a, d)


print(f"n={n} L={L} steps={steps} graph={os.environ.get('HEFX_CHAIN_GRAPH', '1')}: "
      f"chain call {timed(chain):.1f} us/level, fused call per level {timed(fused_levels):.1f}, "
      f"rotate + add calls per level {timed(separate_levels):.1f}", flush=True)

"""Quick throughput probe of the bench unit (rotate step 1 + multiply_plain) -- development aid."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seal_fyp_logistic_regression_amd import Engine

SETS = {
    "C2": (8192, [0xffffffffffe8001, 0xfffff4c001, 0xfffffdc001, 0xfffffffffffc001]),
    "C3": (16384, [0xffffffffffd8001, 0xffffb20001, 0xffffc40001, 0xffffca8001, 0xffffe80001, 0xffffffffffe8001]),
}
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
if name in ("F40", "I60", "F40s", "I60s"):  # uniform-width prime sets to isolate the two arithmetic policies
    from oracle import oracle as O
    n_ = 8192 if name.endswith("s") else 16384
    SETS[name] = (n_, O.coeff_modulus_create(n_, [40 if name[0] == "F" else 60] * 6))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N, primes = SETS[name]
e = Engine(N, primes)
k = len(primes); L = k - 1
rng = np.random.default_rng(0)
def rnd(*shape):
    out = np.empty(shape + (N,), dtype=np.uint64)
    for idx in np.ndindex(*shape):
        out[idx] = rng.integers(0, primes[idx[-1]], N, dtype=np.uint64)
    return out
key = e.to_device(rnd(L, 2, k))
SLAB = os.environ.get("SLAB", "1") != "0"
if SLAB:  # one allocation per tensor class, items are views (what a pooling allocator gives)
    big_ct = e.to_device(np.stack([rnd(2, L) for _ in range(B)]))
    big_pt = e.to_device(np.stack([rnd(L) for _ in range(B)]))
    big_out = e.empty(B, 2, L, N)
    cts = [big_ct.view(i * 2 * L * N, (2, L, N)) for i in range(B)]
    pts = [big_pt.view(i * L * N, (L, N)) for i in range(B)]
    outs = [big_out.view(i * 2 * L * N, (2, L, N)) for i in range(B)]
else:
    cts = [e.to_device(rnd(2, L)) for _ in range(B)]
    pts = [e.to_device(rnd(L)) for _ in range(B)]
    outs = [e.empty(2, L, N) for _ in range(B)]
elts = [3] * B
for _ in range(2):
    e.rotate_multiply_plain_batch(L, cts, elts, [key] * B, pts, outs)
e.sync()
iters = 5
t = time.perf_counter()
for _ in range(iters):
    e.rotate_multiply_plain_batch(L, cts, elts, [key] * B, pts, outs)
e.sync()
dt = (time.perf_counter() - t) / iters
bytes_op = 8 * N * L * (2 * L + 7)
print(f"{name} B={B}: {dt*1e3:.3f} ms/batch  {B/dt:.0f} ops/s  {B/dt*bytes_op/1e9:.1f} GB/s algorithmic "
      f"({B/dt*bytes_op/8e12*100:.2f}% of 8 TB/s)")

"""Key-per-item regime probe (development aid; bench.py's `roofline.key_per_item` is the reported number):
B rotate+multiply_plain items at C3, EVERY item with its own Galois key (B x 7.86 MB of keys drawn on the device), so
each key switch pays its 2L(L+1)N key words from HBM -- the regime `roofline.frac` prices.
    python tools/key_per_item_probe.py [B=1024] [reps=5] [set=C3]
HEFX_STREAMS=0 gives the serial launch sequence (per-kernel times under rocprofv3 --kernel-trace --stats)."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seal_fyp_logistic_regression_amd import Engine
from seal_fyp_logistic_regression_amd.seal import galois_elt_from_step

SETS = {
    "C2": (8192, [0xffffffffffe8001, 0xfffff4c001, 0xfffffdc001, 0xfffffffffffc001]),
    "C3": (16384, [0xffffffffffd8001, 0xffffb20001, 0xffffc40001, 0xffffca8001, 0xffffe80001, 0xffffffffffe8001]),
}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N, primes = SETS[sys.argv[3] if len(sys.argv) > 3 else "C3"]
k = len(primes); L = k - 1
e = Engine(N, primes)
seed = lambda s: hashlib.sha256(s).digest()
ct = e.sample("uniform", seed(b"kpi:ct"), 1, 2 * B, L, 0)
pt = e.sample("uniform", seed(b"kpi:pt"), 2, B, L, 0)
kw = L * 2 * k * N
keys = e.sample("uniform", seed(b"kpi:keys"), 3, 2 * L * B, k, 0)
out = e.empty(B, 2, L, N)
cts = [ct.view(i * 2 * L * N, (2, L, N)) for i in range(B)]
pts = [pt.view(i * L * N, (L, N)) for i in range(B)]
outs = [out.view(i * 2 * L * N, (2, L, N)) for i in range(B)]
kv = [keys.view(i * kw, (L, 2, k, N)) for i in range(B)]
elts = [galois_elt_from_step(1 + (i % 64), N) for i in range(B)]
run = lambda: e.rotate_multiply_plain_batch(L, cts, elts, kv, pts, outs)
run(); run(); e.sync()
ts = []
for _ in range(reps):
    t = time.perf_counter(); run(); e.sync(); ts.append(time.perf_counter() - t)
dt = sorted(ts)[len(ts) // 2]
alg = 8 * N * L * (2 * L + 7)
print(f"key-per-item B={B}: {dt*1e3:.3f} ms  {B/dt:.0f} ops/s  {dt/B*1e6:.2f} us/op  algorithmic {B/dt*alg/1e9:.0f} GB/s "
      f"({B/dt*alg/8e12:.3f} of 8 TB/s); keys {B*kw*8/2**30:.1f} GiB", flush=True)
# the same items with ONE shared key, same box, for the ratio
run1 = lambda: e.rotate_multiply_plain_batch(L, cts, [elts[0]] * B, [kv[0]] * B, pts, outs)
run1(); e.sync()
ts = []
for _ in range(reps):
    t = time.perf_counter(); run1(); e.sync(); ts.append(time.perf_counter() - t)
d1 = sorted(ts)[len(ts) // 2]
print(f"shared key    B={B}: {d1*1e3:.3f} ms  {B/d1:.0f} ops/s  {d1/B*1e6:.2f} us/op", flush=True)

#!/usr/bin/env python3
"""Where one key switch spends its time (development aid).  Needs a -DHEFX_STAMP build:
    tools/build_variant.sh stamp1 -DHEFX_STAMP=1 -DHEFX_ONLY_LOGN=14
    HEFX_LIB=build/libhefx_stamp1.so python tools/stamp_timeline.py [L] [n]
Thread 0 of every workgroup of the five small-batch kernels records the 100 MHz wall clock at its phase boundaries
(HEFX_STAMP=2: after draining its outstanding memory operations).  Printed: per kernel the first entry, the last exit,
the gap to the previous kernel's last exit, and the phase times of the workgroup that finished last."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seal_fyp_logistic_regression_amd import Engine, capi
from oracle import oracle as O

L = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = 16384
primes = O.coeff_modulus_create(N, [60] + [40] * (L - 1) + [60])
e, o = Engine(N, primes), O.Oracle(N, primes)
k = len(primes)
key = e.to_device(o.uniform(k, 2 * L, 2).reshape(L, 2, k, N))
cts = [e.to_device(o.uniform(L, 2, i)) for i in range(n)]
outs = e.empty_many(n, (2, L, N))
lib = ctypes.CDLL(capi.library_path() if hasattr(capi, "library_path") else os.environ["HEFX_LIB"])
lib.hefx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
words = lib.hefx_debug_stamps(None, 0)
assert words > 0
buf = np.zeros(words, dtype=np.uint64)
# stamp slots: on the five-launch path one launch each; on the pair path (HEFX_PAIR, default for small batches) slots 0 + 1
# are the inverse / forward halves of ks_pair_digits, 3 + 4 those of ks_pair_moddown, 5 is ks_pair_finish
names = ["intt_digits", "ntt_digits", "mac", "moddown_intt", "moddown_finish", "pair_finish"]
rows = []
for rep in range(8):
    for _ in range(3):
        e.apply_galois_batch(L, cts, [3] * n, [key] * n, outs=outs)
    e.sync()
    lib.hefx_debug_stamps(None, 1)
    e.apply_galois_batch(L, cts, [3] * n, [key] * n, outs=outs)
    e.sync()
    assert lib.hefx_debug_stamps(buf.ctypes.data, 0) == words
    rows.append(buf.reshape(8, 1024, 16).astype(np.int64).copy())
# the repetition with the median end-to-end time
span = [int(max(r[k][:, 15].max() for k in range(6)) - r[0][:, 0][r[0][:, 0] > 0].min()) for r in rows]
r = rows[int(np.argsort(span)[len(span) // 2])]
t0 = r[0][:, 0][r[0][:, 0] > 0].min()
us = lambda v: (v - t0) / 100.0
print(f"N={N} L={L} n={n}: first entry -> last exit {sorted(span)[len(span)//2] / 100.0:.2f} us (median of {len(span)}; all: "
      + " ".join(f"{s/100.0:.1f}" for s in span) + ")")
prev_end = None
for kid, name in enumerate(names):
    s = r[kid]
    live = s[:, 15] > 0
    if not live.any():
        continue
    first, last = s[live, 0].min(), s[live, 15].max()
    crit = int(np.argmax(np.where(live, s[:, 15], 0)))
    gap = "" if prev_end is None else f"  gap after previous kernel {(first - prev_end) / 100.0:5.2f} us"
    print(f"{name:15s} workgroups {int(live.sum()):4d}  entry {us(first):6.2f}  exit {us(last):6.2f}  duration {(last - first) / 100.0:6.2f} us{gap}")
    ids = [i for i in range(16) if s[crit, i] > 0]
    ph = "  ".join(f"[{i}] +{(s[crit, i] - s[crit, ids[0]]) / 100.0:.2f}" for i in ids)
    print(f"{'':15s} last workgroup {crit}: entered {us(s[crit, 0]):.2f}; stamps (us after its entry): {ph}")
    # spread of the workgroups' own durations
    d = (s[live, 15] - s[live, 0]) / 100.0
    print(f"{'':15s} per-workgroup duration min {d.min():.2f} / median {np.median(d):.2f} / max {d.max():.2f} us; entry spread {(s[live, 0].max() - first) / 100.0:.2f} us")
    prev_end = last

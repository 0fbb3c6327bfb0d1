"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers belong on the CPU
build; there is no GPU ASan on this pool).  `make -C oracle asan` builds oracle/ckks_oracle.c with
-fsanitize=address,undefined; a child interpreter preloads the ASan runtime, loads that library through HEFX_ORACLE_SO and
walks the path once at small sizes: transforms, the RNS key switch, exact hoisting against the per-item sequence,
relinearisation, both rescale divisions, the double-hoisted core, sampling, encryption, encoding."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WALK = r"""
import numpy as np
from oracle import oracle as O
for N, bits in ((64, [30, 25, 25, 30]), (256, [36, 30, 30, 30, 36])):
    primes = O.coeff_modulus_create(N, bits)
    k, L = len(primes), len(primes) - 1
    o = O.Oracle(N, primes)
    a = o.uniform(L, 1, 7)[0]
    for j in range(L):
        row = a[j].copy()
        assert (o.ntt_inv(j, o.ntt_fwd(j, row)) == row).all()
    assert (o.ntt_naive(0, a[0]) == o.ntt_fwd(0, a[0])).all()
    sk = o.gen_secret(11)
    rk = o.gen_relin_key(sk, 12)
    gks = o.gen_galois_keys(sk)  # the reference's default set: +-2^i and the conjugation
    ct = o.uniform(L, 2, 3)
    pt = o.uniform(L, 1, 4)[0]
    for step in (1, -1, 4):
        elt = O.galois_elt_from_step(N, step)
        want = o.apply_galois(ct, elt, gks[elt])
        got, regular = o.apply_galois_hoisted_exact(ct, elt, gks[elt])
        assert (got == want).all() and not regular
        assert (o.rotate_mulplain(ct, elt, gks[elt], pt) == o.multiply_plain(want, pt)).all()
    m = o.multiply(ct, ct)
    r = o.relinearize(m, rk)
    f, g = o.rescale(r, rounded=False), o.rescale(r, rounded=True)
    assert f.shape == g.shape == (2, L - 1, N)
    assert o.mod_drop(ct, L - 1).shape == (2, L - 1, N)
    assert o.rotate_vector(ct, 7, gks).shape == ct.shape  # a NAF chain: 7 = 8 - 1
    vals = np.linspace(-1.0, 1.0, N // 2)
    enc = o.encode(L, vals, 2.0 ** 20)
    assert np.allclose(o.decode(enc, 2.0 ** 20).real, vals, atol=1e-3)
    c = o.encrypt(L, sk, enc, 21)
    assert np.allclose(o.decode(o.decrypt(c, sk), 2.0 ** 20).real, vals, atol=1e-2)
    key32 = bytes(range(32))
    for kind in ("uniform", "ternary", "noise"):
        o.sample(kind, key32, 5, 1, L)
print("walked", O._SO)
"""


def test_oracle_walk_is_clean_under_asan_and_ubsan():
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan_rt) or not os.path.exists(asan_rt):
        pytest.skip("no ASan runtime next to this gcc")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    so = os.path.join(ROOT, "oracle", "libckks_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=asan_rt, HEFX_ORACLE_SO=so, PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-c", WALK], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0 and "walked" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
    assert "libckks_oracle_asan.so" in p.stdout  # the instrumented build is the one that ran
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]

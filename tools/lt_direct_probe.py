"""Linear_Transform_Plain (helper.h:237-262) at C3 with a DIRECT Galois key per step (bench.py's
`lt_sharded.direct_keys_d*` leg on its own, for rocprofv3 --kernel-trace and A/B runs):
    python tools/lt_direct_probe.py [d=512] [reps=10]
LT_NAF=1: the reference's default power-of-two keys instead (every rotation a NAF chain; the engine's shared-prefix forest)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seal_fyp_logistic_regression_amd import algorithms as alg
from seal_fyp_logistic_regression_amd import seal as S

d = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, primes = 16384, [0xffffffffffd8001, 0xffffb20001, 0xffffc40001, 0xffffca8001, 0xffffe80001, 0xffffffffffe8001]
parms = S.EncryptionParameters("ckks")
parms.set_poly_modulus_degree(N)
parms.set_coeff_modulus(primes)
ctx = S.SEALContext.Create(parms)
kg = S.KeyGenerator(ctx, 0xC3)
enc, dec = S.Encryptor(ctx, kg.public_key(), 0xC4), S.Decryptor(ctx, kg.secret_key())
encoder, ev = S.CKKSEncoder(ctx), S.Evaluator(ctx)
eng = ctx.backend.engine
rng = np.random.default_rng(2000 + d)
M, v = rng.uniform(-1, 1, (d, d)), rng.uniform(-1, 1, d)
diags = encoder.encode_many(list(alg.get_all_diagonals(M)), 2.0 ** 40)
ct = enc.encrypt(encoder.encode(v, 2.0 ** 40))
naf = os.environ.get("LT_NAF") == "1"
gk = kg.galois_keys() if naf else kg.galois_keys([-d] + list(range(1, d)))
r = alg.linear_transform_plain(ev, ct, diags, gk)
eng.sync()
ts, hs = [], []
for _ in range(reps):
    t0 = time.perf_counter()
    r = alg.linear_transform_plain(ev, ct, diags, gk)
    t1 = time.perf_counter()
    eng.sync()
    ts.append(time.perf_counter() - t0)
    hs.append(t1 - t0)
ms = sorted(ts)[len(ts) // 2] * 1e3
ok = bool(np.allclose(encoder.decode(dec.decrypt(r))[:d].real, M @ v, atol=1e-3 * d))
print(f"{'NAF-key' if naf else 'direct-key'} LT d={d}: {ms:.3f} ms per call (host submit {sorted(hs)[len(hs)//2]*1e3:.3f} ms), {d/(ms*1e-3):.0f} key switches/s, "
      f"decrypts_to_Mv={ok}", flush=True)

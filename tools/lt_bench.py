"""Linear_Transform_Plain (helper.h:237-262) wall time on the MI355X for the dimensions the reference charted
(FYP Presentation slide 27: N=8192 {60,40,40,60}, d = 10 / 100 / 1000), compute phase only, like the reference's
timer at linear_transformation.cpp:540-542.  Also times the CPU oracle on the smallest size for scale."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seal_fyp_logistic_regression_amd import algorithms as alg
from seal_fyp_logistic_regression_amd import seal as S

N, bits, scale = 8192, [60, 40, 40, 60], 2.0 ** 40
parms = S.EncryptionParameters("ckks"); parms.set_poly_modulus_degree(N); parms.set_coeff_modulus(S.CoeffModulus.Create(N, bits))
ctx = S.SEALContext.Create(parms)
kg = S.KeyGenerator(ctx, 1); gk = kg.galois_keys(); enc = S.Encryptor(ctx, kg.public_key()); dec = S.Decryptor(ctx, kg.secret_key())
encoder, ev = S.CKKSEncoder(ctx), S.Evaluator(ctx)
rng = np.random.default_rng(0)
out = {"params": "N=8192 {60,40,40,60} scale 2^40, default (power-of-two) Galois keys", "published_us": {"10": 1.4e5, "100": 1.3e6, "1000": 1.8e7}, "runs": []}
for d in [int(x) for x in (sys.argv[1:] or ["10", "100", "1000"])]:
    M, v = rng.standard_normal((d, d)), rng.standard_normal(d)
    diags = [encoder.encode(x, scale) for x in alg.get_all_diagonals(M)]
    ct = enc.encrypt(encoder.encode(v, scale))
    ks = sum(len(ev.rotation_plan(s, gk)) for s in [-d] + list(range(1, d)))
    alg.linear_transform_plain(ev, ct, diags, gk); ctx.backend.engine.sync()
    reps = 3 if d >= 1000 else 10
    t = time.perf_counter()
    for _ in range(reps):
        r = alg.linear_transform_plain(ev, ct, diags, gk)
    ctx.backend.engine.sync()
    dt = (time.perf_counter() - t) / reps
    err = float(np.abs(encoder.decode(dec.decrypt(r))[:d].real - M @ v).max())
    out["runs"].append({"d": d, "key_switches_in_SEAL_order": ks, "gpu_us": dt * 1e6, "max_abs_err": err})
    print(out["runs"][-1], flush=True)
print(json.dumps(out))

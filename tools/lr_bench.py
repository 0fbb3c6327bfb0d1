"""predict_cipher_weights (logistic_regression_ckks.cpp:208-266) on the reference's LR parameter set
(N=16384, {60,40x7,60}, scale 2^40, 8 weights): rows x (multiply + relinearize + rescale + 8 sequential rotations),
masks, add_many, degree-3 sigmoid by Horner.  Compute phase only (rows encrypted beforehand), median of 3.
LR_LOG_SUM=1 selects the fast window sum (log2(size) rotations per dot product, not the reference's bits).
usage: lr_bench.py [rows ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seal_fyp_logistic_regression_amd import algorithms as alg
from seal_fyp_logistic_regression_amd import seal as S

LOG_SUM = os.environ.get("LR_LOG_SUM", "0") != "0"
N, bits, scale, nw = 16384, [60] + [40] * 7 + [60], 2.0 ** 40, 8
parms = S.EncryptionParameters("ckks"); parms.set_poly_modulus_degree(N); parms.set_coeff_modulus(S.CoeffModulus.Create(N, bits))
ctx = S.SEALContext.Create(parms)
kg = S.KeyGenerator(ctx, 1); gk = kg.galois_keys(); rk = kg.relin_keys()
enc, dec = S.Encryptor(ctx, kg.public_key(), 2), S.Decryptor(ctx, kg.secret_key())
encoder, ev, eng = S.CKKSEncoder(ctx), S.Evaluator(ctx), ctx.backend.engine
rng = np.random.default_rng(0)
out = {"params": f"N={N} {bits} scale 2^40, {nw} weights", "window_sum": "doubling (fast mode)" if LOG_SUM else "rotate-by-1 chain (reference, bit-exact)", "runs": []}
for rows in [int(x) for x in (sys.argv[1:] or ["100", "2000"])]:
    X = rng.uniform(-1, 1, (rows, nw)); w = rng.uniform(-0.5, 0.5, nw)
    t0 = time.perf_counter()
    feats = [enc.encrypt(p) for p in encoder.encode_many(list(X), scale)]
    cw = enc.encrypt(encoder.encode(w, scale))
    eng.sync(); setup_s = time.perf_counter() - t0
    walls = []
    for _ in range(4):
        t = time.perf_counter()
        pred = alg.predict_cipher_weights(ev, encoder, enc, feats, cw, nw, scale, gk, rk, log_sum=LOG_SUM)
        eng.sync(); walls.append(time.perf_counter() - t)
    got = encoder.decode(dec.decrypt(pred))[:rows].real
    c = alg.SIGMOID_COEFFS[3]; z = X @ w
    want = c[0] + c[1] * z + c[2] * z ** 2 + c[3] * z ** 3
    # the reference masks slot i of row i's replicated dot product (:222-229); the replication covers slots 0..size,
    # so only the first `nw` rows carry their full dot product (a property of the reference's packing, kept as is)
    out["runs"].append({"rows": rows, "ms": sorted(walls[1:])[1] * 1e3, "samples_ms": [round(x * 1e3, 1) for x in walls],
                        "encode_encrypt_rows_s": setup_s, "max_abs_err_first_rows": float(np.abs(got - want)[:nw].max())})
    print(out["runs"][-1], flush=True)
print(json.dumps(out))

"""CC_Matrix_Multiplication (matrix_multiplication.cpp:11-132) wall time on the MI355X, compute phase only (inputs
encoded / encrypted, keys generated beforehand), median of 5 calls incl. the host side:
  dense   the reference's form: all n^2 diagonals of every permutation matrix, 1e-8 added to each entry (:239-297),
          default power-of-two Galois keys (bit-exact to the op-by-op sequence)
  sparse  algorithms.cc_matrix_multiplication_sparse: the non-zero diagonals only (fast mode), the reference's keys
  sparse_direct  the same with a direct Galois key for every step it uses (one key switch per rotation)
  sparse_hoisted direct keys + the sigma / tau rotations through the explicit hoisted entry (exact since round 4: same
                 words as sparse_direct; above 32 rotations per source both run hoisted)
  sparse_hoisted2 the sigma / tau transforms double-hoisted (key-level diagonals, one mod-down each)
usage: matmul_bench.py [C3|C5] n [dense|sparse|sparse_direct|sparse_hoisted|sparse_hoisted2 ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seal_fyp_logistic_regression_amd import algorithms as alg
from seal_fyp_logistic_regression_amd import seal as S

setname, n = sys.argv[1], int(sys.argv[2])
modes = sys.argv[3:] or ["dense", "sparse"]
N = {"C3": 16384, "C5": 32768}[setname]
bits, scale = [60, 40, 40, 40, 40, 60], 2.0 ** 40
parms = S.EncryptionParameters("ckks"); parms.set_poly_modulus_degree(N); parms.set_coeff_modulus(S.CoeffModulus.Create(N, bits))
ctx = S.SEALContext.Create(parms)
kg = S.KeyGenerator(ctx, 1); gk = kg.galois_keys(); enc = S.Encryptor(ctx, kg.public_key(), 2); dec = S.Decryptor(ctx, kg.secret_key())
encoder, ev, eng = S.CKKSEncoder(ctx), S.Evaluator(ctx), ctx.backend.engine
rng = np.random.default_rng(0)
A, B = rng.uniform(-1, 1, (n, n)), rng.uniform(-1, 1, (n, n))
ctA, ctB = enc.encrypt(encoder.encode(A.reshape(-1), scale)), enc.encrypt(encoder.encode(B.reshape(-1), scale))
out = {"params": f"N={N} {bits} scale 2^40, n={n}", "runs": []}
for mode in modes:
    t0 = time.perf_counter()
    if mode == "dense":
        Us, Ut, V, W = alg.matmul_permutation_matrices(n)
        e_ = lambda U: encoder.encode_many(list(alg.get_all_diagonals(U) + 1e-8), scale)
        args = (e_(Us), e_(Ut), [e_(v) for v in V], [e_(w) for w in W])
        run = lambda: alg.cc_matrix_multiplication(ev, ctA, ctB, n, *args, gk)
        rotations = 2 * n * n * n
    else:  # "sparse": the reference's power-of-two keys (NAF chains); "sparse_direct": a direct key per step used
        sig, tau, phi, psi = alg.matmul_permutation_diagonals(n)
        e_ = lambda dd: dict(zip(dd, encoder.encode_many(list(dd.values()), scale)))
        args = (e_(sig), e_(tau), [e_(x) for x in phi], [e_(x) for x in psi])
        keys = gk
        if mode in ("sparse_direct", "sparse_hoisted", "sparse_hoisted2"):
            steps = sorted({-n * n} | {l for dd in [sig, tau] + phi + psi for l in dd if l})
            keys = kg.galois_keys(steps)
        h = {"sparse_hoisted": True, "sparse_hoisted2": 2}.get(mode, False)
        if h == 2:  # sigma / tau diagonals at the key level
            ek = lambda dd: dict(zip(dd, encoder.encode_many(list(dd.values()), scale, parms_id=ctx.k)))
            args = (ek(sig), ek(tau)) + args[2:]
        run = lambda keys=keys, h=h, args=args: alg.cc_matrix_multiplication_sparse(ev, ctA, ctB, n, *args, keys, hoisted=h)
        rotations = len(sig) + len(tau) + sum(len(x) for x in phi) + sum(len(x) for x in psi) + 4
    eng.sync()
    encode_s = time.perf_counter() - t0
    r = run(); eng.sync()
    walls = []
    for _ in range(5):
        t = time.perf_counter(); r = run(); eng.sync(); walls.append(time.perf_counter() - t)
    err = float(np.abs(encoder.decode(dec.decrypt(r))[:n * n].real.reshape(n, n) - A @ B).max())
    out["runs"].append({"mode": mode, "rotations": rotations, "ms": sorted(walls)[2] * 1e3, "samples_ms": [round(w * 1e3, 2) for w in walls],
                        "diagonal_encode_s": encode_s, "max_abs_err": err})
    print(out["runs"][-1], flush=True)
    del args
print(json.dumps(out))

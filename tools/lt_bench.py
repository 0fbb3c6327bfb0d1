"""Linear_Transform_Plain (helper.h:237-262) wall time on the MI355X for the dimensions the reference charted
(FYP Presentation slide 27: N=8192 {60,40,40,60}, d = 10 / 100 / 1000), compute phase only, like the reference's
timer at linear_transformation.cpp:540-542.  Three modes:
  naf      the reference's setup: default power-of-two Galois keys, NAF chains (bit-exact to the op-by-op sequence)
  direct   a direct Galois key per step (keygen.galois_keys(steps)): one key switch per rotation (bit-exact)
  hoisted  direct keys through the explicit hoisted entry (since round 4 the same computation and words as `direct`)
  hoisted2 double hoisting: additionally ONE mod-down for the whole transform (key-level diagonals)
  bsgs     baby-step/giant-step (algorithms.linear_transform_plain_bsgs): ~2*sqrt(d) keys and key switches, hoisted
           baby steps, inner sums through hefx_multiply_plain_sum"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seal_fyp_logistic_regression_amd import algorithms as alg
from seal_fyp_logistic_regression_amd import seal as S

N, bits, scale = 8192, [60, 40, 40, 60], 2.0 ** 40
if os.environ.get("LT_SET") == "C3":  # the headline parameter set instead of the reference's chart set
    N, bits = 16384, [60, 40, 40, 40, 40, 60]
parms = S.EncryptionParameters("ckks"); parms.set_poly_modulus_degree(N); parms.set_coeff_modulus(S.CoeffModulus.Create(N, bits))
ctx = S.SEALContext.Create(parms)
kg = S.KeyGenerator(ctx, 1); gk_default = kg.galois_keys(); enc = S.Encryptor(ctx, kg.public_key(), 2); dec = S.Decryptor(ctx, kg.secret_key())
encoder, ev = S.CKKSEncoder(ctx), S.Evaluator(ctx)
rng = np.random.default_rng(0)
out = {"params": f"N={N} {bits} scale 2^40", "published_us": {"10": 1.4e5, "100": 1.3e6, "1000": 1.8e7}, "runs": []}
for d in [int(x) for x in (sys.argv[1:] or ["10", "100", "1000"])]:
    M, v = rng.standard_normal((d, d)), rng.standard_normal(d)
    diags_data = encoder.encode_many(list(alg.get_all_diagonals(M)), scale)
    ct = enc.encrypt(encoder.encode(v, scale))
    steps = [-d] + list(range(1, d))
    t0 = time.perf_counter()
    gk_direct = kg.galois_keys(steps)
    keygen_s = time.perf_counter() - t0
    diags_key = encoder.encode_many(list(alg.get_all_diagonals(M)), scale, parms_id=ctx.k)
    t0 = time.perf_counter()
    gk_bsgs = kg.galois_keys(alg.bsgs_steps(d))
    keygen_bsgs_s = time.perf_counter() - t0
    diags_bsgs = encoder.encode_many(alg.bsgs_diagonals(alg.get_all_diagonals(M)), scale)
    for mode, gk, hoisted in (("naf", gk_default, False), ("direct", gk_direct, False), ("hoisted", gk_direct, True),
                              ("hoisted2", gk_direct, 2), ("bsgs", gk_bsgs, "bsgs")):
        if hoisted == "bsgs":
            ks = len(alg.bsgs_steps(d))
            diags = diags_bsgs
            run = lambda: alg.linear_transform_plain_bsgs(ev, ct, diags_bsgs, gk_bsgs)
        else:
            ks = sum(len(ev.rotation_plan(s, gk)) for s in steps)
            diags = diags_key if hoisted == 2 else diags_data
            run = lambda: alg.linear_transform_plain(ev, ct, diags, gk, hoisted=hoisted)
        run(); ctx.backend.engine.sync()
        reps = 10
        eng = ctx.backend.engine
        e0, e1 = eng.event(), eng.event()
        walls = []
        eng.event_record(e0)
        for _ in range(reps):
            t = time.perf_counter()
            r = run()
            eng.sync()
            walls.append(time.perf_counter() - t)
        eng.event_record(e1)
        eng.sync()
        dt = sorted(walls)[len(walls) // 2]  # median of the per-call wall times (call + sync)
        busy = eng.event_elapsed_ms(e0, e1) / reps * 1e3
        err = float(np.abs(encoder.decode(dec.decrypt(r))[:d].real - M @ v).max())
        out["runs"].append({"d": d, "mode": mode, "key_switches_in_SEAL_order": ks, "gpu_us": dt * 1e6, "hip_event_us": busy, "samples_us": [round(w * 1e6) for w in walls],
                            "max_abs_err": err, "direct_keygen_s": None if mode == "naf" else (keygen_bsgs_s if mode == "bsgs" else keygen_s),
                            "galois_keys": len(gk.keys)})
        print(out["runs"][-1], flush=True)
    del gk_direct, gk_bsgs
print(json.dumps(out))

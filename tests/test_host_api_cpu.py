"""Host logic of the SEAL API mirror (seal.py) and the L3 algorithms (algorithms.py), exercised on the
oracle-backed backend so it runs without a GPU: CKKS semantics (decrypt(f(enc x)) == f(x)), the reference's known
answers, SEAL's error behaviour, NAF plans."""
import json
import os

import numpy as np
import pytest

from seal_fyp_logistic_regression_amd import algorithms as alg
from seal_fyp_logistic_regression_amd import seal as S
from tests.oracle_backend import OracleBackend

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "appendix_b.json")))


def make(N, bits, seed=1):
    parms = S.EncryptionParameters("ckks")
    parms.set_poly_modulus_degree(N)
    parms.set_coeff_modulus(S.CoeffModulus.Create(N, bits))
    ctx = S.SEALContext.Create(parms, backend=OracleBackend(N, parms.coeff_modulus()))
    kg = S.KeyGenerator(ctx, seed)
    return dict(ctx=ctx, kg=kg, enc=S.Encryptor(ctx, kg.public_key(), seed + 1), dec=S.Decryptor(ctx, kg.secret_key()),
                encoder=S.CKKSEncoder(ctx), ev=S.Evaluator(ctx), rk=kg.relin_keys(), gk=kg.galois_keys())


@pytest.fixture(scope="module")
def env():
    return make(2048, [50, 30, 30, 30, 50])


def dec(e, ct, n=None):
    v = e["encoder"].decode(e["dec"].decrypt(ct))
    return v[:n] if n else v


def test_coeff_modulus_create_matches_golden():
    for s in GOLD["sets"]:
        assert S.CoeffModulus.Create(s["N"], s["bits"]) == [int(p, 16) for p in s["primes"]]
    for step, want in GOLD["naf_examples"].items():
        assert S.naf(int(step)) == want


def test_rotation_plan_counts(env):
    ev, gk = env["ev"], env["gk"]
    # Linear_Transform_Plain key-switch counts: rotate(-d) + rotate(1..d-1)
    for d, want in (("4", 5), ("10", 16), ("16", 29)):
        d = int(d)
        assert sum(len(ev.rotation_plan(s, gk)) for s in [-d] + list(range(1, d))) == want
    with pytest.raises(ValueError, match="Galois key not present"):
        ev.rotation_plan(4, S.KSwitchKeys())


def test_encode_encrypt_roundtrip_and_scalar(env):
    e, scale = env, 2.0 ** 30
    v = np.random.default_rng(0).standard_normal(1024) + 0.5j
    pt = e["encoder"].encode(v, scale)
    assert np.abs(e["encoder"].decode(pt) - v).max() < 1e-5
    ct = e["enc"].encrypt(pt)
    assert ct.size() == 2 and ct.parms_id() == e["ctx"].first_parms_id()
    assert np.abs(dec(e, ct) - v).max() < 1e-4
    pts = e["encoder"].encode(3.25, scale)
    assert np.abs(e["encoder"].decode(pts) - 3.25).max() < 1e-6


def test_evaluator_semantics(env):
    e, ev, scale = env, env["ev"], 2.0 ** 30
    rng = np.random.default_rng(1)
    a, b = rng.standard_normal(1024), rng.standard_normal(1024)
    ca, cb = e["enc"].encrypt(e["encoder"].encode(a, scale)), e["enc"].encrypt(e["encoder"].encode(b, scale))
    assert np.abs(dec(e, ev.add(ca, cb)) - (a + b)).max() < 1e-4
    assert np.abs(dec(e, ev.sub(ca, cb)) - (a - b)).max() < 1e-4
    assert np.abs(dec(e, ev.negate(ca)) + a).max() < 1e-4
    m = ev.multiply(ca, cb)
    assert m.size() == 3 and m.scale == scale * scale
    assert np.abs(dec(e, m) - a * b).max() < 1e-3
    ev.relinearize_inplace(m, e["rk"])
    assert m.size() == 2
    ev.rescale_to_next_inplace(m)
    assert m.parms_id() == e["ctx"].first_parms_id() - 1
    assert np.abs(dec(e, m) - a * b).max() < 1e-3
    for step in (1, -3, 7):
        assert np.abs(dec(e, ev.rotate_vector(ca, step, e["gk"])) - np.roll(a, -step)).max() < 1e-3
    mp = ev.multiply_plain(ca, e["encoder"].encode(b, scale))
    assert np.abs(dec(e, mp) - a * b).max() < 1e-3
    # size-3 + size-2 addition pads (SEAL: result size = max)
    s32 = ev.add(ev.multiply(ca, cb), ev.multiply_plain(ca, e["encoder"].encode(np.ones(1024), scale)))
    assert s32.size() == 3
    assert np.abs(dec(e, s32) - (a * b + a)).max() < 1e-3


def test_error_behaviour(env):
    e, ev, scale = env, env["ev"], 2.0 ** 30
    a = e["enc"].encrypt(e["encoder"].encode([1.0, 2.0], scale))
    b = e["enc"].encrypt(e["encoder"].encode([1.0, 2.0], scale * 2))
    with pytest.raises(ValueError, match="scale mismatch"):
        ev.add(a, b)
    low = a.copy()
    ev.mod_switch_to_next_inplace(low)
    with pytest.raises(ValueError, match="parameter mismatch"):
        ev.add(a, low)
    with pytest.raises(RuntimeError, match="transparent"):
        ev.multiply_plain(a, e["encoder"].encode(np.zeros(4), scale))
    with pytest.raises(ValueError, match="scale out of bounds"):
        big = a.copy()
        for _ in range(6):
            big = ev.multiply_plain(big, e["encoder"].encode([1.0], scale))
    with pytest.raises(ValueError, match="encrypted size must be 2"):
        ev.rotate_vector(ev.multiply(a, a), 1, e["gk"])
    with pytest.raises(ValueError, match="higher level"):
        ev.mod_switch_to_inplace(low, e["ctx"].first_parms_id())
    c = a.copy()
    while c.parms_id() > 1:
        ev.mod_switch_to_next_inplace(c)
    with pytest.raises(ValueError, match="end of modulus switching chain"):
        ev.rescale_to_next_inplace(c)


def test_linear_transform_known_answer(env):
    """M = 1..16 (4x4): M @ [1,5,9,13] = [90,202,314,426] (reference imgs/lin_transf.jpg / matmul first column)."""
    e, ev, scale, d = env, env["ev"], 2.0 ** 30, 4
    M = np.arange(1, 17, dtype=float).reshape(4, 4)
    v = np.array([1.0, 5.0, 9.0, 13.0])
    diags = [e["encoder"].encode(dg, scale) for dg in alg.get_all_diagonals(M)]
    ct = e["enc"].encrypt(e["encoder"].encode(v, scale))
    out = alg.linear_transform_plain(ev, ct, diags, e["gk"])
    got = dec(e, out, d).real
    assert np.allclose(got, [90, 202, 314, 426], atol=1e-2)
    cdiags = [e["enc"].encrypt(p) for p in diags]
    out2 = alg.linear_transform_cipher(ev, ct, cdiags, e["gk"])
    assert out2.size() == 3
    assert np.allclose(dec(e, out2, d).real, [90, 202, 314, 426], atol=1e-2)


def test_cipher_dot_product_and_powers(env):
    e, ev, scale = env, env["ev"], 2.0 ** 30
    a, b = np.array([1.0, 2.0, 3.0, 4.0]), np.array([0.5, -1.0, 2.0, 0.25])
    ca, cb = e["enc"].encrypt(e["encoder"].encode(a, scale)), e["enc"].encrypt(e["encoder"].encode(b, scale))
    dp = alg.cipher_dot_product(ev, ca, cb, 4, e["rk"], e["gk"])
    assert abs(dec(e, dp, 1)[0].real - float(a @ b)) < 0.05  # scale was forced to a power of two (helper.h:489)
    x = e["enc"].encrypt(e["encoder"].encode([0.8, -0.5], scale))
    pw = alg.compute_all_powers(ev, x, 3, e["rk"])
    for i in (2, 3):
        assert np.allclose(dec(e, pw[i], 2).real, np.array([0.8, -0.5]) ** i, atol=1e-2)


def test_matrix_encode_decode(env):
    e, ev, scale, n = env, env["ev"], 2.0 ** 30, 3
    rows = [np.arange(n) + 10.0 * i for i in range(n)]
    cts = [e["enc"].encrypt(e["encoder"].encode(r, scale)) for r in rows]
    packed = alg.c_matrix_encode(ev, cts, e["gk"])
    assert np.allclose(dec(e, packed, n * n).real, np.concatenate(rows), atol=1e-3)
    back = alg.c_matrix_decode(ev, e["encoder"], packed, n, scale, e["gk"])
    for i in range(n):
        assert np.allclose(dec(e, back[i], n).real, rows[i], atol=1e-2)


def test_polynomial_and_logistic_regression_step():
    """rows a9-a11: Tree/Horner sigmoid, predict_cipher_weights, and update_weights stopping where SEAL stops."""
    e = make(2048, [60, 40, 40, 40, 40, 40, 40, 40, 60], seed=4)  # the reference's LR chain (logistic_regression_ckks.cpp:420)
    ev, scale = e["ev"], 2.0 ** 40
    x = np.array([0.8, -0.3, 0.1])
    cx = e["enc"].encrypt(e["encoder"].encode(x, scale))
    c = alg.SIGMOID_COEFFS[3]
    want = c[0] + c[1] * x + c[2] * x ** 2 + c[3] * x ** 3
    t = alg.tree_cipher(ev, e["encoder"], e["enc"], cx, 3, scale, c, e["rk"])
    assert np.allclose(dec(e, t, 3).real, want, atol=1e-3)
    h = alg.horner_cipher(ev, e["encoder"], e["enc"], cx, 3, c, scale, e["rk"])
    assert np.allclose(dec(e, h, 3).real, want, atol=1e-3)
    # 3 observations x 4 features.  (The reference's masking at :222-229 reads slot i of dot product i, which
    # holds the full sum only for i < num_weights -- reproduced, not fixed -- so the check uses rows <= weights.)
    X = np.array([[0.5, -1.0, 0.2, 0.1], [1.5, 0.25, -0.3, 0.4], [-0.75, 0.5, 0.6, -0.2]])
    w = np.array([0.3, -0.6, 0.5, 0.25])
    y = np.array([1.0, 0.0, 1.0])
    feats = [e["enc"].encrypt(e["encoder"].encode(r, scale)) for r in X]
    featsT = [e["enc"].encrypt(e["encoder"].encode(col, scale)) for col in X.T]
    cw = e["enc"].encrypt(e["encoder"].encode(w, scale))
    cy = e["enc"].encrypt(e["encoder"].encode(y, scale))
    pred = alg.predict_cipher_weights(ev, e["encoder"], e["enc"], feats, cw, 4, scale, e["gk"], e["rk"])
    z = X @ w
    assert np.allclose(dec(e, pred, 3).real, c[0] + c[1] * z + c[2] * z ** 2 + c[3] * z ** 3, atol=5e-3)
    with pytest.raises(ValueError, match="scale out of bounds"):  # reference stops at :336
        alg.update_weights(ev, e["encoder"], e["enc"], feats, featsT, cy, cw, 0.1, e["gk"], e["rk"], scale)


def test_hoisted_rotation_is_the_regular_key_switch(env):
    """The hoisted entry (decompose once for all rotations of one ciphertext): since round 4 the engine's hoisted form
    carries the flip-mask term and returns rotate_vector's words, so the oracle-backed twin runs the regular sequence;
    the hoisted linear transform gives M.v with the bits of the direct-key transform."""
    e = env
    ctx, ev, kg = e["ctx"], e["ev"], e["kg"]
    d = 6
    steps = [-d] + list(range(1, d))
    gk = kg.galois_keys(steps)
    v = np.arange(1.0, 17.0)
    scale = 2.0 ** 30
    ct = e["enc"].encrypt(e["encoder"].encode(np.tile(v, ctx.N // 2 // 16), scale))
    L, be = ct.parms_id(), ctx.backend
    elts = [S.galois_elt_from_step(s, ctx.N) for s in (1, 3, 5)]
    outs = be.rotate_hoisted_batch(L, ct.data, elts, [gk.key(x) for x in elts])
    for s, o_, x in zip((1, 3, 5), outs, elts):
        got = dec(e, S.Ciphertext()._set(o_, 2, L, ct.scale), 16).real
        assert np.abs(got - np.roll(v, -s)).max() < 1e-3
        assert (o_ == be.apply_galois(L, ct.data, x, gk.key(x))).all()
    rng = np.random.default_rng(1)
    M, w = rng.standard_normal((d, d)), rng.standard_normal(d)
    diags = [e["encoder"].encode(x, scale) for x in alg.get_all_diagonals(M)]
    cw = e["enc"].encrypt(e["encoder"].encode(w, scale))
    a = alg.linear_transform_plain(ev, cw, diags, gk)
    h = alg.linear_transform_plain(ev, cw, diags, gk, hoisted=True)
    assert np.abs(dec(e, a, d).real - M @ w).max() < 1e-2 and np.abs(dec(e, h, d).real - M @ w).max() < 1e-2
    assert (np.asarray(a.data) == np.asarray(h.data)).all()
    with pytest.raises(ValueError, match="direct Galois key"):
        alg.linear_transform_plain(ev, cw, diags, e["gk"], hoisted=True)


def test_double_hoisted_linear_transform_decrypts_to_the_matrix_product(env):
    """second fast mode on the oracle twin: key-level diagonals, one mod-down for the whole transform"""
    e = env
    ctx, ev, kg = e["ctx"], e["ev"], e["kg"]
    d = 6
    gk = kg.galois_keys([-d] + list(range(1, d)))
    rng = np.random.default_rng(2)
    M, w = rng.standard_normal((d, d)), rng.standard_normal(d)
    scale = 2.0 ** 30
    diags_key = [e["encoder"].encode(x, scale, parms_id=ctx.k) for x in alg.get_all_diagonals(M)]
    diags = [e["encoder"].encode(x, scale) for x in alg.get_all_diagonals(M)]
    cw = e["enc"].encrypt(e["encoder"].encode(w, scale))
    h2 = alg.linear_transform_plain(ev, cw, diags_key, gk, hoisted=2)
    ref = alg.linear_transform_plain(ev, cw, diags, gk)
    assert h2.parms_id() == ref.parms_id() and h2.scale == ref.scale
    assert np.abs(dec(e, h2, d).real - M @ w).max() < 1e-2
    assert np.abs(dec(e, h2, d).real - dec(e, ref, d).real).max() < 1e-3
    with pytest.raises(ValueError, match="key-level"):
        alg.linear_transform_plain(ev, cw, diags, gk, hoisted=2)


def test_multiply_plain_sum_equals_the_op_by_op_sequence_and_checks_like_it(env):
    """Evaluator.multiply_plain_sum = add_many(multiply_plain(...)) per group (helper.h:271,275): same payload on the
    oracle twin by construction, SEAL's exceptions for mismatched scale / level / transparent operands."""
    e = env
    ev, encoder = e["ev"], e["encoder"]
    rng = np.random.default_rng(5)
    scale = 2.0 ** 30
    vs, ws = rng.standard_normal((5, 8)), rng.standard_normal((5, 8))
    cts = [e["enc"].encrypt(encoder.encode(v, scale)) for v in vs]
    pts = [encoder.encode(w, scale) for w in ws]
    one = ev.multiply_plain_sum(cts, pts)
    assert len(one) == 1 and np.abs(dec(e, one[0], 8).real - (vs * ws).sum(0)).max() < 1e-3
    ref = ev.add_many([ev.multiply_plain(c, p) for c, p in zip(cts, pts)])
    assert (one[0].data == ref.data).all() and one[0].scale == ref.scale
    grouped = ev.multiply_plain_sum(cts, pts, group=2)
    assert len(grouped) == 3
    assert np.abs(dec(e, grouped[2], 8).real - vs[4] * ws[4]).max() < 1e-3
    pv = alg.linear_transform_ciphermatrix_plainvector(ev, pts, cts)       # helper.h:265-278
    assert (pv.data == ref.data).all()
    with pytest.raises(ValueError, match="scale mismatch"):
        ev.multiply_plain_sum(cts, pts[:4] + [encoder.encode(ws[4], 2.0 ** 20)])
    with pytest.raises(RuntimeError, match="transparent"):
        ev.multiply_plain_sum(cts, pts[:4] + [encoder.encode(np.zeros(8), scale)])
    low = encoder.encode(ws[4], scale, parms_id=cts[0].parms_id() - 1)
    with pytest.raises(ValueError, match="parameter mismatch"):
        ev.multiply_plain_sum(cts, pts[:4] + [low])


@pytest.mark.parametrize("d,n1", [(6, None), (7, 2), (9, 4), (5, 5)])
def test_bsgs_linear_transform_decrypts_to_the_matrix_product(env, d, n1):
    """baby-step / giant-step form of Linear_Transform_Plain: n1-1 + n2-1 key switches, M.v in the first d slots,
    with and without hoisted baby steps; missing direct keys are refused."""
    e = env
    ctx, ev, kg, encoder = e["ctx"], e["ev"], e["kg"], e["encoder"]
    steps = alg.bsgs_steps(d, n1)
    a, b = alg.bsgs_split(d, n1)
    assert a * b >= d and len(steps) == a + b - 1
    gk = kg.galois_keys(steps)
    rng = np.random.default_rng(d)
    M, w = rng.standard_normal((d, d)), rng.standard_normal(d)
    scale = 2.0 ** 30
    sd = [encoder.encode(x, scale) for x in alg.bsgs_diagonals(alg.get_all_diagonals(M), n1)]
    cw = e["enc"].encrypt(encoder.encode(w, scale))
    ref = alg.linear_transform_plain(ev, cw, [encoder.encode(x, scale) for x in alg.get_all_diagonals(M)],
                                     kg.galois_keys([-d] + list(range(1, d))))
    for hoisted in (True, False):
        r = alg.linear_transform_plain_bsgs(ev, cw, sd, gk, n1, hoisted=hoisted)
        assert r.parms_id() == ref.parms_id() and r.scale == ref.scale and r.size() == 2
        assert np.abs(dec(e, r, d).real - M @ w).max() < 1e-2
    with pytest.raises(ValueError, match="direct Galois key"):
        alg.linear_transform_plain_bsgs(ev, cw, sd, e["gk"], n1)  # power-of-two keys only: step 3 (or 6) is missing


def test_sparse_matrix_product_matches_the_dense_one_up_to_the_epsilons():
    """cc_matrix_multiplication_sparse (non-zero diagonals only, 3n + 3(n-1) - 1 rotations) against the reference's
    form with 1e-8 on every entry of all n^2 diagonals (matrix_multiplication.cpp:239-297): A.B to CKKS precision, and
    the diagonal index maps equal the dense matrices' non-zero diagonals."""
    n = 3
    for m in (2, 3, 5):
        Us, Ut, V, W = alg.matmul_permutation_matrices(m)
        s_, t_, p_, q_ = alg.matmul_permutation_diagonals(m)
        for dense, sparse in [(Us, s_), (Ut, t_)] + list(zip(V, p_)) + list(zip(W, q_)):
            want = alg.nonzero_diagonals(dense)
            assert list(want) == list(sparse) and all((want[l] == sparse[l]).all() for l in want)
        assert (len(s_), len(t_), len(p_[0]), len(q_[0])) == (2 * m - 1, m, 2, 1)
    e = make(4096, [60, 40, 40, 40, 40, 60])  # at 2^30 the all-epsilon diagonals would round to zero polynomials
    scale = 2.0 ** 40
    rng = np.random.default_rng(3)
    A, B = rng.standard_normal((n, n)), rng.standard_normal((n, n))
    sig, tau, phi, psi = alg.matmul_permutation_diagonals(n)
    enc = lambda dd: {l: e["encoder"].encode(v, scale) for l, v in dd.items()}
    ctA = e["enc"].encrypt(e["encoder"].encode(A.reshape(-1), scale))
    ctB = e["enc"].encrypt(e["encoder"].encode(B.reshape(-1), scale))
    r = alg.cc_matrix_multiplication_sparse(e["ev"], ctA, ctB, n, enc(sig), enc(tau), [enc(x) for x in phi],
                                            [enc(x) for x in psi], e["gk"])
    Us, Ut, V, W = alg.matmul_permutation_matrices(n)
    dense = lambda U: [e["encoder"].encode(dg + 1e-8, scale) for dg in alg.get_all_diagonals(U)]
    ref = alg.cc_matrix_multiplication(e["ev"], ctA, ctB, n, dense(Us), dense(Ut), [dense(v) for v in V],
                                       [dense(w) for w in W], e["gk"])
    assert (r.size(), r.parms_id(), r.scale) == (ref.size(), ref.parms_id(), ref.scale)
    got = dec(e, r, n * n).real.reshape(n, n)
    assert np.abs(got - A @ B).max() < 1e-2
    assert np.abs(got - dec(e, ref, n * n).real.reshape(n, n)).max() < 1e-2


def test_shared_rotations_in_the_matrix_product_keep_the_bits_of_the_per_transform_loop():
    """cc_matrix_multiplication forms the rotations of ctA0 / ctB0 once for all n-1 Step-2 transforms; the literal
    loop of matrix_multiplication.cpp:40-43 (one Linear_Transform_Plain per k) must give the same ciphertext bits."""
    n = 3
    e = make(4096, [60, 40, 40, 40, 40, 60])
    scale = 2.0 ** 40
    rng = np.random.default_rng(8)
    A = rng.standard_normal((n, n))
    Us, Ut, V, W = alg.matmul_permutation_matrices(n)
    dense = lambda U: [e["encoder"].encode(dg + 1e-8, scale) for dg in alg.get_all_diagonals(U)]
    ct = e["enc"].encrypt(e["encoder"].encode(A.reshape(-1), scale))
    ct0 = alg.linear_transform_plain(e["ev"], ct, dense(Us), e["gk"])
    Vd = [dense(v) for v in V]
    shared = alg._linear_transforms_of_one_input(e["ev"], ct0, Vd, e["gk"])
    literal = [alg.linear_transform_plain(e["ev"], ct0, vd, e["gk"]) for vd in Vd]
    for a, b in zip(shared, literal):
        assert a.parms_id() == b.parms_id() and a.scale == b.scale and (a.data == b.data).all()


class _ForestTwin(OracleBackend):
    """The oracle twin with the two engine entry points the lockstep composites look for -- a multi-root rotation forest and
    the many-input linear transform -- each stated node by node / input by input on the oracle, so that the HOST logic of
    algorithms._rotations_of_many / _linear_transforms_of_inputs / linear_transforms_plain_many (forest construction, leaf
    bookkeeping, grouping of the product sums) runs on the CPU."""
    calls = 0

    def apply_galois_forest(self, L, parents, ext_ins, elts, keys, pts=None):
        type(self).calls += 1
        outs = []
        for i, p in enumerate(parents):
            assert p < i
            src = ext_ins[i] if p < 0 else outs[p]
            r = self.apply_galois_batch(L, [src], [elts[i]], [keys[i]])[0]
            if pts is not None and pts[i] is not None:
                r = self.multiply_plain(L, 2, r, pts[i])
            outs.append(r)
        return outs


def test_lockstep_transforms_of_several_inputs_keep_the_bits_of_the_loops():
    """Round 6: CC_Matrix_Multiplication runs the sigma / tau transforms and the V_k / W_k families in lockstep (one rotation
    forest with two roots).  On a backend that offers the forest entry the merged composition must give the ciphertext
    bits of the per-input loops -- and of the whole product as the plain twin computes it."""
    n = 3
    scale = 2.0 ** 40
    rng = np.random.default_rng(9)
    A, B = rng.standard_normal((n, n)), rng.standard_normal((n, n))
    Us, Ut, V, W = alg.matmul_permutation_matrices(n)

    def run(backend_cls):
        parms = S.EncryptionParameters("ckks")
        parms.set_poly_modulus_degree(4096)
        parms.set_coeff_modulus(S.CoeffModulus.Create(4096, [60, 40, 40, 40, 40, 60]))
        ctx = S.SEALContext.Create(parms, backend=backend_cls(4096, parms.coeff_modulus()))
        kg = S.KeyGenerator(ctx, 4)
        enc, encoder, ev, gk = S.Encryptor(ctx, kg.public_key(), 5), S.CKKSEncoder(ctx), S.Evaluator(ctx), kg.galois_keys()
        dense = lambda U: [encoder.encode(dg + 1e-8, scale) for dg in alg.get_all_diagonals(U)]
        ctA, ctB = enc.encrypt(encoder.encode(A.reshape(-1), scale)), enc.encrypt(encoder.encode(B.reshape(-1), scale))
        a0 = alg.linear_transform_plain(ev, ctA, dense(Us), gk)
        b0 = alg.linear_transform_plain(ev, ctB, dense(Ut), gk)
        Vd, Wd = [dense(v) for v in V], [dense(w) for w in W]
        merged = alg._linear_transforms_of_inputs(ev, [a0, b0], [Vd, Wd], gk)
        loops = [alg._linear_transforms_of_one_input(ev, a0, Vd, gk), alg._linear_transforms_of_one_input(ev, b0, Wd, gk)]
        rots = alg._rotations_of_many(ev, [a0, b0], [1, 3, -2, 5], gk)
        # ... and with the plaintext product fused into the last key switch of every plan (the sharded transforms' form)
        pts = [[Vd[0][l] for l in (1, 3, 2, 5)], [Wd[0][l] for l in (1, 3, 2, 5)]]
        fused = alg._rotations_of_many(ev, [a0, b0], [1, 3, -2, 5], gk, pts)
        fused_ref = [[ev.multiply_plain(ev.rotate_vector(c, s, gk), p) for s, p in zip((1, 3, -2, 5), pp)]
                     for c, pp in zip((a0, b0), pts)]
        for rr, ref in zip(fused, fused_ref):
            for x, y in zip(rr, ref):
                assert x.scale == y.scale and (x.data == y.data).all()
        prod = alg.cc_matrix_multiplication(ev, ctA, ctB, n, dense(Us), dense(Ut), Vd, Wd, gk)
        return merged, loops, rots, [[ev.rotate_vector(c, s, gk) for s in (1, 3, -2, 5)] for c in (a0, b0)], prod

    _ForestTwin.calls = 0
    merged, loops, rots, rot_ref, prod = run(_ForestTwin)
    assert _ForestTwin.calls >= 3   # the merged paths were taken (two forests in _linear_transforms_of_inputs, one in _rotations_of_many, ...)
    for fam_m, fam_l in zip(merged, loops):
        assert len(fam_m) == len(fam_l) == n - 1
        for a, b in zip(fam_m, fam_l):
            assert a.parms_id() == b.parms_id() and a.scale == b.scale and (a.data == b.data).all()
    for rr, ref in zip(rots, rot_ref):
        for a, b in zip(rr, ref):
            assert a.scale == b.scale and (a.data == b.data).all()
    *_, prod_plain = run(OracleBackend)          # no forest entry: the per-input paths
    assert prod.size() == prod_plain.size() == 3 and prod.scale == prod_plain.scale and (prod.data == prod_plain.data).all()


@pytest.mark.parametrize("size", [8, 5, 7])
def test_log_depth_window_sum_gives_the_same_dot_products(size):
    """cipher_dot_product_many(log_sum=True): about log2(size) rotations instead of size-1; slots 0..size-1 carry the
    same replicated dot product as the reference's rotate-by-1 chain (helper.h:472-476)."""
    e = make(2048, [50, 30, 30, 50])
    scale = 2.0 ** 30
    rng = np.random.default_rng(size)
    X, w = rng.uniform(-1, 1, (3, size)), rng.uniform(-1, 1, size)
    feats = [e["enc"].encrypt(e["encoder"].encode(r, scale)) for r in X]
    cw = e["enc"].encrypt(e["encoder"].encode(w, scale))
    ref = alg.cipher_dot_product_many(e["ev"], feats, [cw] * 3, size, e["rk"], e["gk"])
    fast = alg.cipher_dot_product_many(e["ev"], feats, [cw] * 3, size, e["rk"], e["gk"], log_sum=True)
    for i in range(3):
        assert fast[i].parms_id() == ref[i].parms_id() and fast[i].scale == ref[i].scale
        a, b = dec(e, ref[i], size).real, dec(e, fast[i], size).real
        assert np.abs(a - X[i] @ w).max() < 1e-3 and np.abs(b - X[i] @ w).max() < 1e-3


def test_default_keygenerator_and_encryptor_draw_fresh_os_randomness():
    """seed=None must be the default -- two default KeyGenerators give different secret keys, two
    default Encryptors (and two calls of one) give different ciphertexts of the same plaintext; explicit seeds stay
    reproducible (tests / tools only)."""
    N, bits = 2048, [50, 30, 50]
    parms = S.EncryptionParameters("ckks")
    parms.set_poly_modulus_degree(N)
    parms.set_coeff_modulus(S.CoeffModulus.Create(N, bits))
    ctx = S.SEALContext.Create(parms, backend=OracleBackend(N, parms.coeff_modulus()))
    kg1, kg2 = S.KeyGenerator(ctx), S.KeyGenerator(ctx)
    assert kg1._key32 != kg2._key32
    assert not np.array_equal(kg1.secret_key().host, kg2.secret_key().host)
    pt = S.CKKSEncoder(ctx).encode(np.arange(4.0), 2.0 ** 30)
    pk = kg1.public_key()
    e1, e2 = S.Encryptor(ctx, pk), S.Encryptor(ctx, pk)
    be = ctx.backend
    c1, c2, c3 = (be.to_host(e.encrypt(pt).data) for e in (e1, e2, e1))
    assert not np.array_equal(c1, c2) and not np.array_equal(c1, c3)
    assert np.array_equal(S.KeyGenerator(ctx, 7).secret_key().host, S.KeyGenerator(ctx, 7).secret_key().host)
    dec = S.Decryptor(ctx, kg1.secret_key())
    got = S.CKKSEncoder(ctx).decode(dec.decrypt(e2.encrypt(pt)))[:4]
    assert np.allclose(got.real, np.arange(4.0), atol=1e-4)


def test_batched_helpers_accept_empty_lists(env):
    """A rank that owns no unit of a sharded product (parallel.cc_matrix_multiplication_sparse_sharded with
    dimension - 1 < world) hands empty lists to the batched Evaluator helpers; on a backend that HAS the batch entry
    points (the HIP engine) they used to index cts[0].  Stubs stand in for those entry points here: they must not be
    reached, and every helper must return an empty result."""
    ev, be = env["ev"], env["ev"].be
    def boom(*a, **k):
        raise AssertionError("batch entry point called for an empty list")
    names = ("multiply_batch", "relinearize_batch", "rescale_batch", "add_batch")
    saved = {n: getattr(be, n, None) for n in names}
    try:
        for n in names:
            setattr(be, n, boom)
        assert ev.multiply_many([], []) == []
        assert ev.rescale_to_next_many_inplace([]) == []
        assert ev.relinearize_many_inplace([], env["rk"]) == []
        assert ev.add_pairs([], []) == []
    finally:
        for n, f in saved.items():
            if f is None:
                delattr(be, n)
            else:
                setattr(be, n, f)

"""GPU parity, round-2 additions (VERDICT r1 "close the test holes" + the floor/round hedge):

* rescale_to_next in BOTH division modes (floor = SEAL 3.4.x as SURVEY App. A.9 states it, round = SEAL >= 3.5) against
  orc_rescale(rounded=0/1) at C2 / C3 / C4 / C5, sizes 2 and 3, batches, every level kind (FP64- and integer-policy
  dropped primes);
* config 4's parameter set (N=16384, {60,40x7,60}, L=8 and 7) through every key-switch entry: apply_galois, relinearize,
  rescale, rotate+multiply_plain batches larger than one chunk and large enough to take the streaming-x path;
* config 1 (vector_ops.cpp:198-288 ckksOps on the BFVDefault(8192) chain): add_plain -> square -> relinearize;
* the composites a2 / a4 / a5 / a7 / a8 once at N=16384 (C3 chain), a10 at N=16384 on the C4 chain;
* per-ciphertext transparency, two contexts in one process with the current device switched underneath.

Everything goes through the C-ABI (ctypes) and is compared word for word with the CPU oracle."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "appendix_b.json")))
SETS = {s["name"]: s for s in GOLD["sets"]}


def _mk(name, env=None):
    from oracle import oracle as O
    from seal_fyp_logistic_regression_amd import Engine
    s = SETS[name]
    primes = [int(p, 16) for p in s["primes"]]
    old = {}
    for k_, v in (env or {}).items():  # context-creation knobs (HEFX_CHUNK ...) are read once, at hefx_context_create
        old[k_] = os.environ.get(k_)
        os.environ[k_] = v
    try:
        e = Engine(s["N"], primes)
    finally:
        for k_, v in old.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
    return O.Oracle(s["N"], primes), e, primes


def _rand_key(o, seed):
    return o.uniform(o.k, 2 * (o.k - 1), seed).reshape(o.k - 1, 2, o.k, o.N)


@pytest.fixture(scope="module")
def c4():
    return _mk("C4")


# ---------------------------------------------------------------------------------------------------------------
# rescale: floor and round
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("setname", ["C2", "C3", "C4", "C5"])
def test_rescale_floor_and_rounded_bit_exact(setname):
    """hefx_rescale_to_next_mode(FLOOR / ROUND) == orc_rescale(rounded = 0 / 1): sizes 2 and 3, the top level (drops a
    40-bit FP64-policy prime), level 2 (drops q_1 into the 60-bit q_0), a batch of 3, and the context default."""
    o, e, primes = _mk(setname)
    Ltop = o.k - 1
    for L in sorted({Ltop, 2}):
        for size in (2, 3):
            ct = o.uniform(L, size, 40 + 10 * L + size)
            d = e.to_device(ct)
            for rounded in (False, True):
                want = o.rescale(ct, rounded=rounded)
                got = e.rescale_to_next(L, size, d, rounded=rounded).download()
                assert got.shape == (size, L - 1, o.N)
                assert (got == want).all(), (setname, L, size, rounded)
            assert not (o.rescale(ct, rounded=True) == o.rescale(ct, rounded=False)).all()  # the modes do differ
    # contiguous batch
    L = Ltop
    cts = np.stack([o.uniform(L, 2, 900 + i) for i in range(3)])
    for rounded in (False, True):
        got = e.rescale_to_next(L, 2, e.to_device(cts), count=3, rounded=rounded).download()
        assert (got == np.stack([o.rescale(c, rounded=rounded) for c in cts])).all()
    # context default: round (round 6) unless HEFX_RESCALE=floor / set_rescale_rounded(False)
    assert e.rescale_rounded is (os.environ.get("HEFX_RESCALE", "round") != "floor")
    ct = o.uniform(L, 2, 77)
    e.set_rescale_rounded(False)
    assert e.rescale_rounded is False
    assert (e.rescale_to_next(L, 2, e.to_device(ct)).download() == o.rescale(ct, rounded=False)).all()
    e.set_rescale_rounded(True)
    assert e.rescale_rounded is True
    assert (e.rescale_to_next(L, 2, e.to_device(ct)).download() == o.rescale(ct, rounded=True)).all()
    # edge values: zero rows and q-1 everywhere
    edge = np.zeros((2, L, o.N), dtype=np.uint64)
    for j in range(L):
        edge[1, j, :] = primes[j] - 1
    for rounded in (False, True):
        assert (e.rescale_to_next(L, 2, e.to_device(edge), rounded=rounded).download() ==
                o.rescale(edge, rounded=rounded)).all()


def test_rounded_rescale_through_the_evaluator_matches_the_twin_and_ckks():
    """multiply -> relinearize -> rescale with the context in ROUND mode: same bits as the oracle twin in ROUND mode,
    different bits from FLOOR, and both decrypt to the product (the modes differ by < 1 unit of the last place)."""
    from tests.test_gpu_composites import make, bits, decode
    a, b = np.array([1.5, -2.0, 0.25, 3.0]), np.array([0.5, 4.0, -8.0, 1.0])
    res = {}
    for rounded in (False, True):
        for kind in ("gpu", "oracle"):
            e = make(8192, [60, 40, 40, 60], kind, seed=3)
            e["ctx"].backend.rescale_rounded = rounded
            ca = e["enc"].encrypt(e["encoder"].encode(a, 2.0 ** 40))
            cb = e["enc"].encrypt(e["encoder"].encode(b, 2.0 ** 40))
            m = e["ev"].multiply(ca, cb)
            e["ev"].relinearize_inplace(m, e["rk"])
            e["ev"].rescale_to_next_inplace(m)
            res[(rounded, kind)] = (e, m)
    for rounded in (False, True):
        (eg, mg), (eo, mo) = res[(rounded, "gpu")], res[(rounded, "oracle")]
        assert (bits(eg, mg) == bits(eo, mo)).all()
        assert np.allclose(decode(eg, mg, 4), a * b, atol=1e-5)
    assert not (bits(*res[(True, "gpu")]) == bits(*res[(False, "gpu")])).all()


# ---------------------------------------------------------------------------------------------------------------
# config 4 parameter set: N=16384, L=8 (k=9) and L=7
# ---------------------------------------------------------------------------------------------------------------
def test_c4_apply_galois_relinearize_rescale_bit_exact(c4):
    from oracle import oracle as O
    o, e, primes = c4
    assert o.N == 16384 and o.k == 9
    key = _rand_key(o, 7)
    dkey = e.to_device(key)
    for L in (8, 7, 1):
        ct = o.uniform(L, 2, 11 + L)
        for step in (1, -1, 64):
            elt = O.galois_elt_from_step(o.N, step)
            assert (e.apply_galois(L, e.to_device(ct), elt, dkey).download() == o.apply_galois(ct, elt, key)).all(), (L, step)
    ct = o.uniform(8, 2, 99)
    assert (e.apply_galois(8, e.to_device(ct), 2 * o.N - 1, dkey).download() == o.apply_galois(ct, 2 * o.N - 1, key)).all()
    for L in (8, 7, 2):
        a, b = o.uniform(L, 2, 1), o.uniform(L, 2, 2)
        m = o.multiply(a, b)
        assert (e.multiply(L, e.to_device(a), e.to_device(b)).download() == m).all()
        want = o.relinearize(m, key)
        assert (e.relinearize(L, e.to_device(m), dkey).download() == want).all()
        outs = e.relinearize_batch(L, [e.to_device(m), e.to_device(o.multiply(b, b))], dkey)
        assert (outs[1].download() == o.relinearize(o.multiply(b, b), key)).all()
        for rounded in (False, True):
            assert (e.rescale_to_next(L, 2, e.to_device(want), rounded=rounded).download() ==
                    o.rescale(want, rounded=rounded)).all()
            assert (e.rescale_to_next(L, 3, e.to_device(m), rounded=rounded).download() == o.rescale(m, rounded=rounded)).all()


@pytest.mark.parametrize("env,n", [(None, 40), ({"HEFX_CHUNK": "16"}, 40), ({"HEFX_CHUNK": "8", "HEFX_SUB": "3"}, 19)])
def test_c4_rotate_multiply_plain_batches_bit_exact(env, n):
    """L=8 and L=7 at N=16384: one chunk big enough to stream x (40 items x 72 rows > 256 MB), three chunks alternating
    on the two internal streams, and sub-chunked x; distinct keys per step so the MAC's one-item path runs too."""
    from oracle import oracle as O
    o, e, primes = _mk("C4", env)
    keys = {s: _rand_key(o, 1000 + s) for s in (1, 2, 3)}
    dkeys = {s: e.to_device(k_) for s, k_ in keys.items()}
    for L in (8, 7):
        cts = [o.uniform(L, 2, 200 + i) for i in range(n)]
        pts = [o.uniform(L, 1, 300 + i)[0] for i in range(n)]
        steps = [1 + (i % 3) if i % 5 else 1 for i in range(n)]
        elts = [O.galois_elt_from_step(o.N, s) for s in steps]
        outs = e.rotate_multiply_plain_batch(L, [e.to_device(c) for c in cts], elts, [dkeys[s] for s in steps],
                                             [e.to_device(p) for p in pts])
        for i in range(n):
            want = o.rotate_mulplain(cts[i], elts[i], keys[steps[i]], pts[i])
            assert (outs[i].download() == want).all(), (L, i)
        plain = e.apply_galois_batch(L, [e.to_device(c) for c in cts[:5]], elts[:5], [dkeys[s] for s in steps[:5]])
        for i in range(5):
            assert (plain[i].download() == o.apply_galois(cts[i], elts[i], keys[steps[i]])).all()


# ---------------------------------------------------------------------------------------------------------------
# config 1: vector_ops.cpp ckksOps on the BFVDefault(8192) chain (43/43/44/44/44 bits -- every prime integer-policy)
# ---------------------------------------------------------------------------------------------------------------
def _make_primes(N, primes, kind, seed):
    from seal_fyp_logistic_regression_amd import seal as S
    from tests.oracle_backend import OracleBackend
    parms = S.EncryptionParameters("ckks")
    parms.set_poly_modulus_degree(N)
    parms.set_coeff_modulus(primes)
    backend = OracleBackend(N, parms.coeff_modulus()) if kind == "oracle" else None
    ctx = S.SEALContext.Create(parms, backend=backend)
    kg = S.KeyGenerator(ctx, seed)
    return dict(ctx=ctx, kg=kg, enc=S.Encryptor(ctx, kg.public_key(), seed + 1), dec=S.Decryptor(ctx, kg.secret_key()),
                encoder=S.CKKSEncoder(ctx, device_encode=False), ev=S.Evaluator(ctx), rk=kg.relin_keys())


def test_config1_ckksops_add_plain_square_relinearize_bit_exact():
    """BASELINE config 1 (/root/reference/vector_ops.cpp:198-288): CKKS context on CoeffModulus::BFVDefault(8192),
    scale = sqrt(last prime), (cipher_vec1 + plain_vec2)^2 via add_plain_inplace, square_inplace, relinearize_inplace
    (:268-270).  Same bits as the oracle twin; decrypts to (i + (i % 2) + 1)^2."""
    from seal_fyp_logistic_regression_amd import seal as S
    from tests.test_gpu_composites import bits, decode
    N = 8192
    primes = S.CoeffModulus.BFVDefault(N)
    assert [p.bit_length() for p in primes] == [43, 43, 44, 44, 44]
    slots = N // 2
    v1 = np.arange(slots, dtype=float)                       # :230-234
    v2 = (np.arange(slots) % 2 + 1).astype(float)            # :239-243
    scale = float(np.sqrt(float(primes[-1])))                # :251
    out = {}
    for kind in ("gpu", "oracle"):
        e = _make_primes(N, primes, kind, seed=17)
        p1, p2 = e["encoder"].encode(v1, scale), e["encoder"].encode(v2, scale)
        ct = e["enc"].encrypt(p1)
        e["ev"].add_plain_inplace(ct, p2)                    # :268
        e["ev"].square_inplace(ct)                           # :269
        assert ct.size() == 3
        e["ev"].relinearize_inplace(ct, e["rk"])             # :270
        out[kind] = (e, ct)
    (eg, cg), (eo, co) = out["gpu"], out["oracle"]
    assert cg.size() == 2 and cg.parms_id() == co.parms_id() and cg.scale == co.scale
    assert (bits(eg, cg) == bits(eo, co)).all()
    got = decode(eg, cg, slots)
    want = (v1 + v2) ** 2
    assert np.allclose(got, want, rtol=1e-3, atol=0.5)  # scale 2^22 only: ~7 significant bits on values up to 1.7e7


# ---------------------------------------------------------------------------------------------------------------
# composites at full ring size (N = 16384)
# ---------------------------------------------------------------------------------------------------------------
def test_composites_a2_a4_a5_a7_a8_at_n16384_c3_chain():
    """Linear_Transform_Cipher (a2), C_Matrix_Encode / _Decode (a4, a5), cipher_dot_product (a7) and
    compute_all_powers (a8) on config 3's parameter set -- the <14> kernel instantiations, size-3 adds, rescale of
    size-2 and size-3 ciphertexts at L = 5..2."""
    from seal_fyp_logistic_regression_amd import algorithms as alg
    from tests.test_gpu_composites import both, bits, decode
    rng = np.random.default_rng(11)
    d = 4
    M, v = rng.standard_normal((d, d)), rng.standard_normal(d)
    a, b = np.arange(1.0, 9.0), np.linspace(-1, 1, 8)
    rows = rng.standard_normal((3, 3))

    def run(e):
        scale = 2.0 ** 40
        enc, encoder, ev = e["enc"], e["encoder"], e["ev"]
        diags = [encoder.encode(x, scale) for x in alg.get_all_diagonals(M)]
        ct = enc.encrypt(encoder.encode(v, scale))
        ltc = alg.linear_transform_cipher(ev, ct, [enc.encrypt(p) for p in diags], e["gk"])
        ca, cb = enc.encrypt(encoder.encode(a, scale)), enc.encrypt(encoder.encode(b, scale))
        dp = alg.cipher_dot_product(ev, ca, cb, 8, e["rk"], e["gk"])
        pw = alg.compute_all_powers(ev, cb, 4, e["rk"])
        packed = alg.c_matrix_encode(ev, [enc.encrypt(encoder.encode(r, scale)) for r in rows], e["gk"])
        unpacked = alg.c_matrix_decode(ev, encoder, packed, 3, scale, e["gk"])
        return ltc, dp, pw, packed, unpacked

    r = both(16384, [60, 40, 40, 40, 40, 60], run, seed=6)
    (eg, g), (eo, o) = r["gpu"], r["oracle"]
    ltc_g, dp_g, pw_g, pk_g, un_g = g
    ltc_o, dp_o, pw_o, pk_o, un_o = o
    assert ltc_g.size() == 3 and (bits(eg, ltc_g) == bits(eo, ltc_o)).all()
    assert np.allclose(decode(eg, ltc_g, d), M @ v, atol=1e-4)
    assert dp_g.parms_id() == dp_o.parms_id() and (bits(eg, dp_g) == bits(eo, dp_o)).all()
    assert abs(decode(eg, dp_g, 1)[0] - float(a @ b)) < 1e-3
    for i in range(2, 5):
        assert pw_g[i].parms_id() == pw_o[i].parms_id() and (bits(eg, pw_g[i]) == bits(eo, pw_o[i])).all()
        assert np.allclose(decode(eg, pw_g[i], 8), b ** i, atol=1e-4)
    assert (bits(eg, pk_g) == bits(eo, pk_o)).all()
    for i in range(3):
        assert (bits(eg, un_g[i]) == bits(eo, un_o[i])).all()
        assert np.allclose(decode(eg, un_g[i], 3), rows[i], atol=1e-4)


def test_logistic_regression_step_at_n16384_c4_chain(rescale_mode):
    """rows a9-a11 at config 4's FULL parameter set (N=16384, {60,40x7,60}, L=8): Tree sigmoid, predict_cipher_weights
    over 3 rows x 4 weights (row-batched key switches at L=8 ... 4), update_weights raising where SEAL raises."""
    from seal_fyp_logistic_regression_amd import algorithms as alg
    from tests.test_gpu_composites import both, bits, decode
    X = np.array([[0.5, -1.0, 0.2, 0.1], [1.5, 0.25, -0.3, 0.4], [-0.75, 0.5, 0.6, -0.2]])
    w = np.array([0.3, -0.6, 0.5, 0.25])
    y = np.array([1.0, 0.0, 1.0])
    c = alg.SIGMOID_COEFFS[3]

    def run(e):
        scale = 2.0 ** 40
        feats = [e["enc"].encrypt(e["encoder"].encode(r, scale)) for r in X]
        featsT = [e["enc"].encrypt(e["encoder"].encode(col, scale)) for col in X.T]
        cw = e["enc"].encrypt(e["encoder"].encode(w, scale))
        cy = e["enc"].encrypt(e["encoder"].encode(y, scale))
        x = e["enc"].encrypt(e["encoder"].encode([0.8, -0.3], scale))
        t = alg.tree_cipher(e["ev"], e["encoder"], e["enc"], x, 3, scale, c, e["rk"])
        pred = alg.predict_cipher_weights(e["ev"], e["encoder"], e["enc"], feats, cw, 4, scale, e["gk"], e["rk"])
        with pytest.raises(ValueError, match="scale out of bounds"):
            alg.update_weights(e["ev"], e["encoder"], e["enc"], feats, featsT, cy, cw, 0.1, e["gk"], e["rk"], scale)
        return t, pred

    r = both(16384, [60, 40, 40, 40, 40, 40, 40, 40, 60], run, seed=4)
    (eg, (tg, pg)), (eo, (to, po)) = r["gpu"], r["oracle"]
    assert (bits(eg, tg) == bits(eo, to)).all()
    assert pg.parms_id() == po.parms_id() and (bits(eg, pg) == bits(eo, po)).all()
    z = X @ w
    assert np.allclose(decode(eg, pg, 3), c[0] + c[1] * z + c[2] * z ** 2 + c[3] * z ** 3, atol=5e-3)


# ---------------------------------------------------------------------------------------------------------------
# per-ciphertext transparency, device selection per call, whole-batch validation
# ---------------------------------------------------------------------------------------------------------------
def test_transparency_is_detected_per_ciphertext_of_a_batch():
    from seal_fyp_logistic_regression_amd import capi
    o, e, primes = _mk("C2")
    L = 3
    pt = o.uniform(L, 1, 5)[0]
    cts = np.stack([o.uniform(L, 2, 60 + i) for i in range(4)])
    e.multiply_plain(L, 2, e.to_device(cts), e.to_device(pt), count=4)
    e.check_transparent()                                   # nothing transparent
    cts[2, 1] = 0                                           # ONE ciphertext of the batch has c1 == 0
    got = e.multiply_plain(L, 2, e.to_device(cts), e.to_device(pt), count=4).download()
    assert (got == np.stack([o.multiply_plain(c, pt) for c in cts])).all()
    with pytest.raises(capi.TransparentCiphertextError):
        e.check_transparent()
    e.check_transparent()                                   # cleared


def test_invalid_item_in_a_multi_chunk_batch_leaves_the_context_usable():
    """ks_run validates the whole batch before anything is submitted: a bad Galois element in the LAST
    chunk must raise without launching the earlier chunks, and the next call must work and be bit-exact."""
    from oracle import oracle as O
    o, e, primes = _mk("C2", {"HEFX_CHUNK": "4"})
    L, n = 3, 11
    key = _rand_key(o, 9)
    dkey = e.to_device(key)
    cts = [o.uniform(L, 2, 500 + i) for i in range(n)]
    dcts = [e.to_device(c) for c in cts]
    outs = e.empty_many(n, (2, L, o.N))
    sentinel = np.full((2, L, o.N), 7, dtype=np.uint64)
    for o_ in outs:
        o_.upload(sentinel)
    with pytest.raises(ValueError):
        e.apply_galois_batch(L, dcts, [3] * (n - 1) + [4], [dkey] * n, outs=outs)   # even element in the third chunk
    e.sync()
    assert all((o_.download() == sentinel).all() for o_ in outs)                     # nothing was written
    for _ in range(3):                                                               # ring slots / streams still fine
        got = e.apply_galois_batch(L, dcts, [3] * n, [dkey] * n, outs=outs)
    for i in range(n):
        assert (got[i].download() == o.apply_galois(cts[i], 3, key)).all()


def test_batch_items_processed_grouped_by_key_land_in_their_own_outputs():
    """ks_run processes the items of a batch grouped by key (neighbours then share their key loads in the MAC): a batch
    whose keys arrive scrambled -- three keys, several chunks, some items in place -- must still put every result where
    its item says, with the bits of the item-by-item evaluation."""
    from oracle import oracle as O
    o, e, primes = _mk("C2", {"HEFX_CHUNK": "4"})
    L, n = 3, 13
    steps = [1, 2, 1, 4, 2, 2, 1, 4, 4, 1, 2, 4, 1]
    keys = {s_: _rand_key(o, 40 + s_) for s_ in (1, 2, 4)}
    dkeys = {s_: e.to_device(k_) for s_, k_ in keys.items()}
    elts = [O.galois_elt_from_step(o.N, s_) for s_ in steps]
    cts = [o.uniform(L, 2, 700 + i) for i in range(n)]
    pts = [o.uniform(L, 1, 800 + i)[0] for i in range(n)]
    dcts = [e.to_device(c) for c in cts]
    outs = e.empty_many(n, (2, L, o.N))
    for i in (3, 7, 12):                                   # in-place items
        outs[i] = dcts[i]
    got = e.apply_galois_batch(L, dcts, elts, [dkeys[s_] for s_ in steps], outs=outs)
    for i in range(n):
        assert (got[i].download() == o.apply_galois(cts[i], elts[i], keys[steps[i]])).all(), i
    dcts = [e.to_device(c) for c in cts]
    got = e.rotate_multiply_plain_batch(L, dcts, elts, [dkeys[s_] for s_ in steps], [e.to_device(p) for p in pts])
    for i in range(n):
        assert (got[i].download() == o.rotate_mulplain(cts[i], elts[i], keys[steps[i]], pts[i])).all(), i


def test_two_contexts_with_the_current_device_switched_underneath():
    """Every C-ABI entry selects its context's device itself.  On a one-GPU box both contexts live on
    device 0; the test still drives two interleaved contexts (separate scratch, rings, streams, pools) and -- when a
    second device exists -- places the second context there and flips torch's current device between calls."""
    import torch
    from oracle import oracle as O
    from seal_fyp_logistic_regression_amd import Engine
    ndev = torch.cuda.device_count()
    s = SETS["C2"]
    primes = [int(p, 16) for p in s["primes"]]
    o = O.Oracle(s["N"], primes)
    e0, e1 = Engine(s["N"], primes, device=0), Engine(s["N"], primes, device=1 if ndev > 1 else 0)
    key = _rand_key(o, 3)
    ct, pt = o.uniform(3, 2, 1), o.uniform(3, 1, 2)[0]
    want = o.rotate_mulplain(ct, 3, key, pt)
    for rep in range(3):
        for i, e in enumerate((e0, e1)):
            if ndev > 1:
                torch.cuda.set_device((i + 1) % ndev)       # the WRONG device is current when the call is made
            d = e.to_device(ct)
            got = e.rotate_multiply_plain_batch(3, [d], [3], [e.to_device(key)], [e.to_device(pt)])[0].download()
            assert (got == want).all(), (rep, i)
            m = e.relinearize(3, e.square(3, d), e.to_device(key))
            assert (e.rescale_to_next(3, 2, m).download() == o.rescale(o.relinearize(o.multiply(ct, ct), key), rounded=e.rescale_rounded)).all()
    if ndev > 1:
        torch.cuda.set_device(0)


def test_pointer_table_batch_entries_bit_exact():
    """hefx_add_batch / hefx_sub_batch / hefx_multiply_plain_batch / hefx_rescale_to_next_batch (the lockstep batches the
    C++ shim's recorder submits): scattered operands, more items than one pointer-table slice, size 2 and 3, both
    rescale modes, in-place add."""
    o, e, primes = _mk("C2")
    L = 3
    n = 9
    A = [o.uniform(L, 2, 700 + i) for i in range(n)]
    B = [o.uniform(L, 2, 800 + i) for i in range(n)]
    P = [o.uniform(L, 1, 900 + i)[0] for i in range(n)]
    dA, dB, dP = [e.to_device(x) for x in A], [e.to_device(x) for x in B], [e.to_device(x) for x in P]
    pad = e.empty(12345)                      # keeps the next allocations from being a slab in order
    outs = e.add_batch(L, 2, dA[::-1], dB)     # reversed list: not contiguous -> pointer-table path
    for i in range(n):
        assert (outs[i].download() == o.add(A[n - 1 - i], B[i])).all()
    outs = e.sub_batch(L, 2, dA, dB)
    for i in range(n):
        assert (outs[i].download() == o.sub(A[i], B[i])).all()
    e.sub_batch(L, 2, dA, dB, outs=dA)         # in place on the first operand
    for i in range(n):
        assert (dA[i].download() == o.sub(A[i], B[i])).all()
    outs = e.multiply_plain_batch(L, 2, dB, dP)
    for i in range(n):
        assert (outs[i].download() == o.multiply_plain(B[i], P[i])).all()
    M = [o.multiply(B[i], B[(i + 1) % n]) for i in range(4)]     # size 3
    dM = [e.to_device(m) for m in M]
    for rounded in (False, True):
        e.set_rescale_rounded(rounded)
        r2 = e.rescale_batch(L, 2, dB[::-1])
        for i in range(n):
            assert (r2[i].download() == o.rescale(B[n - 1 - i], rounded=rounded)).all()
        r3 = e.rescale_batch(L, 3, dM[::-1])
        for i in range(4):
            assert (r3[i].download() == o.rescale(M[3 - i], rounded=rounded)).all()
    e.set_rescale_rounded(False)
    # more items than one table slice (the ring slot holds 1536 pointers: 512 triples / 768 pairs)
    big = 800
    idx = [i % n for i in range(big)]
    outs = e.add_batch(L, 2, [dB[i] for i in idx], [dB[(i + 1) % n] for i in idx])
    for t in (0, 511, 512, 799):
        assert (outs[t].download() == o.add(B[idx[t]], B[(idx[t] + 1) % n])).all()
    outs = e.rescale_batch(L, 2, [dB[i] for i in idx])
    for t in (0, 767, 768, 799):
        assert (outs[t].download() == o.rescale(B[idx[t]], rounded=e.rescale_rounded)).all()
    with pytest.raises(ValueError):
        e.rescale_batch(1, 2, dB[::-1])
    del pad


def test_c_abi_allreduce_on_a_one_rank_communicator():
    """hefx_comm_unique_id / hefx_comm_init / hefx_allreduce_sum (RCCL resolved with dlopen inside libhefx.so).  A one-GPU
    box can only host a one-rank communicator (RCCL wants one device per rank): the sum over one rank is the input, and
    the local canonicalisation must bring words >= q_j back into [0,q_j).  The N>1 arithmetic (wrap-free uint64 sum +
    reduction == serial add_many) is covered on CPU by tests/test_parallel_cpu.py."""
    from seal_fyp_logistic_regression_amd import Engine
    o, e, primes = _mk("C2")
    L = 3
    assert e.comm_world == 0
    with pytest.raises(ValueError):
        e.allreduce_sum(L, 2, e.to_device(o.uniform(L, 2, 1)))     # no communicator yet
    uid = Engine.comm_unique_id()
    assert len(uid) == 128
    e.comm_init(1, 0, uid)
    assert e.comm_world == 1
    parts = [o.uniform(L, 2, 50 + i) for i in range(8)]
    raw = np.zeros_like(parts[0])
    want = np.zeros_like(parts[0])
    for p_ in parts:                      # what 8 ranks' partials add up to before the reduction
        raw = raw + p_
        want = o.add(want, p_)
    d = e.to_device(raw)
    e.allreduce_sum(L, 2, d)
    assert (d.download() == want).all()
    with pytest.raises(ValueError):
        e.comm_init(1, 0, uid)            # one communicator per context
    e.comm_destroy()
    assert e.comm_world == 0


@pytest.mark.parametrize("N", [4096, 16384])
def test_primes_at_the_top_of_the_admissible_range_bit_exact(N):
    """The integer NTT policy keeps values in [0,8q) forward / [0,4q) inverse (under-estimated Shoup quotient), which is
    tight against 2^64 for primes just below 2^61 -- the largest hefx_context_create admits (SEAL's own primes stop at
    60 bits).  NTT round trip, key switch (rotation, fused product, relinearisation) and both rescales against the
    oracle on the four largest such primes, with all-(q-1) inputs as the worst case for every lazy sum."""
    from oracle import oracle as O
    from seal_fyp_logistic_regression_amd import Engine
    from seal_fyp_logistic_regression_amd.seal import _is_prime
    primes, v = [], (1 << 61) - 2 * N + 1
    while len(primes) < 4:
        if _is_prime(v):
            primes.append(v)
        v -= 2 * N
    assert all(p.bit_length() == 61 for p in primes)
    o, e = O.Oracle(N, primes), Engine(N, primes)
    k, L = 4, 3
    worst = np.stack([np.full(N, p - 1, dtype=np.uint64) for p in primes])
    for a in (o.uniform(k, 1, 3)[0], worst):
        d = e.to_device(a[None])
        e.ntt_forward(d, 1, k, 0)
        assert (d.download()[0] == np.stack([o.ntt_fwd(j, a[j]) for j in range(k)])).all()
        e.ntt_inverse(d, 1, k, 0)
        assert (d.download()[0] == a).all()
    key = _rand_key(o, 7)
    dkey = e.to_device(key)
    ct_worst = np.stack([worst[:L], worst[:L]])
    for ct in (o.uniform(L, 2, 11), ct_worst):
        pt = o.uniform(L, 1, 12)[0]
        for elt in (3, 2 * N - 1, O.galois_elt_from_step(N, -5)):
            assert (e.apply_galois(L, e.to_device(ct), elt, dkey).download() == o.apply_galois(ct, elt, key)).all()
            got = e.rotate_multiply_plain_batch(L, [e.to_device(ct)] * 3, [elt] * 3, [dkey] * 3, [e.to_device(pt)] * 3)
            assert all((g.download() == o.rotate_mulplain(ct, elt, key, pt)).all() for g in got)
        m = o.multiply(ct, ct)
        r = o.relinearize(m, key)
        assert (e.relinearize(L, e.to_device(m), dkey).download() == r).all()
        for rounded in (False, True):
            assert (e.rescale_to_next(L, 2, e.to_device(r), rounded=rounded).download() == o.rescale(r, rounded=rounded)).all()


@pytest.mark.parametrize("knob", ["HEFX_QUARTER=0", "HEFX_QUARTER=1", "auto", "HEFX_QMASK=5", "HEFX_QMASK=10", "HEFX_PAIR=1", "HEFX_PAIR=0"])
def test_small_batch_quarter_row_path_bit_exact(knob):
    """The small-batch key switch (quarter-row workgroups with eight coefficients per thread, descriptors in the kernel
    arguments) gives the oracle's bits for rotations, fused rotate+multiply_plain and relinearisation at every ring size it
    is built for (N = 4096 .. 32768: every radix-8 pass / remainder combination of the 8-coefficient cores), top and lower
    levels, 1..19 items, distinct keys / elements.  Each of the four transform launches picks quarter rows or split-2
    workgroups on its own (the scratch layouts are shared): forced off, forced on, the engine's own per-launch rule, and the
    two complementary mixes (inverse launches on quarter rows with forward ones on split-2 workgroups, and the reverse).
    Round 5: "auto" takes the PAIR path (four launches, two transform phases: ks_pair_*) wherever 4 L^2 n <= 512 workgroups;
    HEFX_PAIR=1 forces it for every small batch -- also the ones whose 1900 quarter workgroups are not co-resident -- and
    HEFX_PAIR=0 keeps it off."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, json, os
sys.path.insert(0, %r)
import numpy as np
from oracle import oracle as O
from seal_fyp_logistic_regression_amd import Engine
gold = json.load(open(os.path.join(%r, "tests", "golden", "appendix_b.json")))
sets = {s["name"]: s for s in gold["sets"]}
ok = True
for name in ("C2", "C3", "C4", "C5", "toy4096", "toy2048"):
    if name.startswith("toy"):
        N = int(name[3:])
        primes = O.coeff_modulus_create(N, [50, 30, 30, 50])
    else:
        N, primes = sets[name]["N"], [int(p, 16) for p in sets[name]["primes"]]
    o, e = O.Oracle(N, primes), Engine(N, primes)
    k = len(primes)
    keys = [o.uniform(k, 2 * (k - 1), 70 + i).reshape(k - 1, 2, k, N) for i in range(3)]
    dkeys = [e.to_device(x) for x in keys]
    for L in sorted({k - 1, max(1, k - 2), 1}):
        for n in (1, 3, 8) + ((19,) if name in ("C3", "toy4096") else ()):  # 19: descriptors in the kernel arguments up to 32 items
            cts = [o.uniform(L, 2, 100 * L + i) for i in range(n)]
            pts = [o.uniform(L, 1, 200 * L + i)[0] for i in range(n)]
            steps = [(1, -1, 5, 2)[i %% 4] for i in range(n)]
            elts = [O.galois_elt_from_step(N, s) for s in steps]
            kk = [i %% 3 for i in range(n)]
            d = [e.to_device(c) for c in cts]
            outs = e.apply_galois_batch(L, d, elts, [dkeys[j] for j in kk])
            ok &= all((outs[i].download() == o.apply_galois(cts[i], elts[i], keys[kk[i]])).all() for i in range(n))
            outs = e.rotate_multiply_plain_batch(L, d, elts, [dkeys[j] for j in kk], [e.to_device(p) for p in pts])
            ok &= all((outs[i].download() == o.rotate_mulplain(cts[i], elts[i], keys[kk[i]], pts[i])).all() for i in range(n))
            m = [o.multiply(cts[i], cts[(i + 1) %% n]) for i in range(n)]
            outs = e.relinearize_batch(L, [e.to_device(x) for x in m], dkeys[0])
            ok &= all((outs[i].download() == o.relinearize(m[i], keys[0])).all() for i in range(n))
        ct = o.uniform(L, 2, 999)     # in place: regular path (the input must be copied first)
        dd = e.to_device(ct)
        e.apply_galois(L, dd, 3, dkeys[0], out=dd)
        ok &= bool((dd.download() == o.apply_galois(ct, 3, keys[0])).all())
print("PARITY", ok)
""" % (root, root)
    env = {k: v for k, v in os.environ.items() if k not in ("HEFX_QUARTER", "HEFX_QMASK", "HEFX_PAIR", "HEFX_PAIR_MAX")}
    if knob != "auto":
        env[knob.split("=")[0]] = knob.split("=")[1]
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert "PARITY True" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_python_composition_on_the_gpu_equals_the_native_one_call_paths():
    """algorithms.linear_transform_plain / the sparse double-hoisted transform prefer a native one-call C-ABI entry when the
    backend has one; the oracle twin always takes the Python composition.  Here the composition itself runs ON THE GPU
    (the backend's native entries hidden) and must give the bits of the native call and of the twin."""
    from seal_fyp_logistic_regression_amd import algorithms as alg
    from tests.test_gpu_composites import make, bits, decode
    rng = np.random.default_rng(21)
    d = 9
    M, v = rng.standard_normal((d, d)), rng.standard_normal(d)

    class Hidden:  # the GPU backend without its one-call linear transform
        def __init__(self, be):
            self._be = be

        def __getattr__(self, name):
            if name in ("linear_transform_plain", "linear_transform_plain_bsgs", "linear_transform_plain_hoisted2_sparse"):
                raise AttributeError(name)
            return getattr(self._be, name)

    out = {}
    for kind in ("gpu", "gpu-composition", "oracle"):
        e = make(8192, [60, 40, 40, 60], "oracle" if kind == "oracle" else "gpu", seed=9)
        if kind == "gpu-composition":
            e["ev"].be = Hidden(e["ev"].be)
        scale = 2.0 ** 40
        diags = [e["encoder"].encode(x, scale) for x in alg.get_all_diagonals(M)]
        ct = e["enc"].encrypt(e["encoder"].encode(v, scale))
        out[kind] = (e, alg.linear_transform_plain(e["ev"], ct, diags, e["gk"]))
    ref = bits(*out["oracle"])
    assert (bits(*out["gpu"]) == ref).all()
    assert (bits(*out["gpu-composition"]) == ref).all()
    assert np.allclose(decode(*out["gpu-composition"], d), M @ v, atol=1e-4)

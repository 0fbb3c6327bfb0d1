"""GPU CKKS encode (hefx_ckks_encode) against the CPU oracle's encoder.

Floating point, so the bar is the tolerance stated here, not bit equality: the two encoders run different FFTs, so
round(p_k * scale) may differ by exactly one unit where p_k * scale sits within FFT error (~N * 2^-52 * scale) of a
half-integer.  Tolerances: |coefficient difference| <= 1 on every coefficient, on fewer than 1 % of them;
decode(encode(v)) == v to 1e-7 at scale 2^40.  Everything downstream of the plaintext is integer work and stays
bit-exact (tests/test_gpu_parity.py, tests/test_gpu_composites.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BITS = {1024: [27], 2048: [54], 4096: [36, 36, 37], 8192: [60, 40, 40, 60], 16384: [60, 40, 40, 40, 40, 60],
        32768: [60, 40, 40, 60]}


def setup(N):
    from oracle import oracle as O
    from seal_fyp_logistic_regression_amd import Engine
    primes = O.coeff_modulus_create(N, BITS[N])
    return Engine(N, primes), O.Oracle(N, primes), primes


def centered(rows, primes):
    out = rows.astype(np.int64)
    for j, q in enumerate(primes[: rows.shape[0]]):
        out[j] = np.where(rows[j] > q // 2, out[j] - q, out[j])
    return out


@pytest.mark.parametrize("N", [1024, 2048, 4096, 8192, 16384, 32768])
def test_encode_matches_oracle_within_one_unit(N):
    e, o, primes = setup(N)
    L = max(1, len(primes) - 1)
    rng = np.random.default_rng(N)
    scale = 2.0 ** (20 if N <= 2048 else 40)
    for nvalues, cplx in ((N // 2, True), (N // 2, False), (7, True), (1, False)):
        count = 3
        v = rng.uniform(-1, 1, (count, nvalues)) + (1j * rng.uniform(-1, 1, (count, nvalues)) if cplx else 0)
        dev = e.ckks_encode(L, v, scale)
        e.ntt_inverse(dev, count, L, 0)
        got = dev.download()
        total = mism = 0
        for i in range(count):
            want = o.encode(L, v[i], scale)
            wc = np.stack([o.ntt_inv(j, want[j]) for j in range(L)])
            d = centered(got[i], primes) - centered(wc, primes)
            assert np.abs(d).max() <= 1
            # every RNS row carries the SAME integer coefficient
            assert all((d[j] == d[0]).all() for j in range(L))
            mism += int((d[0] != 0).sum())
            total += N
        assert mism < 0.01 * total


def test_encode_decode_round_trip_and_python_encoder():
    from seal_fyp_logistic_regression_amd import seal as S
    parms = S.EncryptionParameters("ckks")
    parms.set_poly_modulus_degree(8192)
    parms.set_coeff_modulus(S.CoeffModulus.Create(8192, [60, 40, 40, 60]))
    ctx = S.SEALContext.Create(parms)
    dev_enc, host_enc = S.CKKSEncoder(ctx), S.CKKSEncoder(ctx, device_encode=False)
    rng = np.random.default_rng(5)
    v = rng.uniform(-3, 3, 4096) + 1j * rng.uniform(-3, 3, 4096)
    scale = 2.0 ** 40
    pt = dev_enc.encode(v, scale)
    assert pt.parms_id() == 3 and pt.scale == scale and not pt.is_zero
    assert np.abs(dev_enc.decode(pt) - v).max() < 1e-7
    ref = host_enc.encode(v, scale)
    a, b = ctx.backend.to_host(pt.data), ctx.backend.to_host(ref.data)
    assert a.shape == b.shape  # NTT-form rows differ wherever one coefficient differs: compare through decode
    assert np.abs(dev_enc.decode(pt) - host_enc.decode(ref)).max() < 1e-9
    # batch form == one at a time, bit for bit (same kernel, same inputs)
    vs = [rng.uniform(-1, 1, 10) for _ in range(5)]
    many = dev_enc.encode_many(vs, scale)
    for x, p in zip(vs, many):
        assert (ctx.backend.to_host(p.data) == ctx.backend.to_host(dev_enc.encode(x, scale).data)).all()
    # zero vectors are flagged without a device round trip; multiply_plain then raises like SEAL
    z = dev_enc.encode(np.zeros(10), scale)
    assert z.is_zero and not ctx.backend.to_host(z.data).any()
    kg = S.KeyGenerator(ctx, 1)
    ct = S.Encryptor(ctx, kg.public_key(), 2).encrypt(pt)
    with pytest.raises(RuntimeError, match="transparent"):
        S.Evaluator(ctx).multiply_plain(ct, z)
    with pytest.raises(ValueError):
        dev_enc.encode(np.zeros(4097), scale)


def test_linear_transform_with_device_encoded_diagonals():
    """The reference's benchmark shape (matrix_mult_benchmark.cpp:291-336) with every encode on the GPU:
    result == U.v to CKKS precision."""
    from seal_fyp_logistic_regression_amd import algorithms as alg
    from seal_fyp_logistic_regression_amd import seal as S
    parms = S.EncryptionParameters("ckks")
    parms.set_poly_modulus_degree(8192)
    parms.set_coeff_modulus(S.CoeffModulus.Create(8192, [60, 40, 40, 60]))
    ctx = S.SEALContext.Create(parms)
    kg = S.KeyGenerator(ctx, 3)
    encoder, ev = S.CKKSEncoder(ctx), S.Evaluator(ctx)
    d, scale = 16, 2.0 ** 40
    rng = np.random.default_rng(11)
    U, v = rng.uniform(-1, 1, (d, d)), rng.uniform(-1, 1, d)
    diags = encoder.encode_many(alg.get_all_diagonals(U), scale)
    ct = S.Encryptor(ctx, kg.public_key(), 4).encrypt(encoder.encode(v, scale))
    out = alg.linear_transform_plain(ev, ct, diags, kg.galois_keys())
    got = encoder.decode(S.Decryptor(ctx, kg.secret_key()).decrypt(out))[:d].real
    assert np.abs(got - U @ v).max() < 1e-4


def test_encode_rejects_what_it_cannot_do():
    from oracle import oracle as O
    from seal_fyp_logistic_regression_amd import Engine
    from seal_fyp_logistic_regression_amd.capi import HefxError
    e, _, _ = setup(4096)
    with pytest.raises(ValueError):
        e.ckks_encode(2, np.zeros((1, 2049)), 2.0 ** 30)
    with pytest.raises(ValueError):
        e.ckks_encode(2, np.zeros((1, 8)), -1.0)
    with pytest.raises(ValueError):
        e.ckks_encode(9, np.zeros((1, 8)), 2.0 ** 30)
    with pytest.raises(ValueError):
        e.ckks_encode(2, np.zeros((0, 8)), 2.0 ** 30)


@pytest.mark.parametrize("N,bits", [(2048, [54]), (4096, [36, 36, 37]), (8192, [60, 40, 40, 60]),
                                    (16384, [60, 40, 40, 40, 40, 60]), (32768, [60, 40, 40, 40, 60])])
def test_gpu_decode_matches_host_decode(N, bits):
    """hefx_ckks_decode (inverse NTT, Garner CRT, centring, slot-root FFT on the device) against the host decode of
    the SAME plaintext (exact big-integer CRT + numpy FFT): floating point, tolerance 1e-9 relative to the largest
    value; at every level of the chain, for complex values of both signs, and after an evaluator op."""
    from seal_fyp_logistic_regression_amd import seal as S
    parms = S.EncryptionParameters("ckks")
    parms.set_poly_modulus_degree(N)
    parms.set_coeff_modulus(S.CoeffModulus.Create(N, bits))
    ctx = S.SEALContext.Create(parms)
    dev, host = S.CKKSEncoder(ctx), S.CKKSEncoder(ctx, device_encode=False)
    rng = np.random.default_rng(N + 1)
    scale = 2.0 ** (20 if N <= 2048 else 30)
    top = ctx.first_parms_id()
    for L in sorted({top, max(1, top - 1), 1}):
        v = rng.uniform(-50, 50, N // 2) + 1j * rng.uniform(-50, 50, N // 2)
        pt = host.encode(v, scale, parms_id=L)
        g, h = dev.decode(pt), host.decode(pt)
        assert g.shape == h.shape == (N // 2,)
        assert np.abs(g - h).max() < 1e-9 * 50, (N, L)
        assert np.abs(g - v).max() < 1e-3
    # a decrypted ciphertext (noise in the low bits, product scale) decodes the same way on both sides
    if top >= 2:
        kg = S.KeyGenerator(ctx, 5)
        enc, dec, ev = S.Encryptor(ctx, kg.public_key(), 6), S.Decryptor(ctx, kg.secret_key()), S.Evaluator(ctx)
        a = rng.uniform(-2, 2, N // 2)
        ct = enc.encrypt(dev.encode(a, scale))
        sq = ev.multiply_plain(ct, dev.encode(a, scale))
        p = dec.decrypt(sq)
        assert np.abs(dev.decode(p) - host.decode(p)).max() < 1e-9 * 4
        assert np.abs(dev.decode(p).real - a * a).max() < 1e-2

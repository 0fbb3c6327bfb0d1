"""GPU sampler / encrypt / decrypt (hefx_sample_*, hefx_encrypt, hefx_decrypt) against the
CPU oracle: integer work, so the bar is bit equality."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEY = bytes((7 * i + 3) & 0xFF for i in range(32))
BITS = {1024: [27], 2048: [54], 4096: [36, 36, 37], 8192: [60, 40, 40, 60], 16384: [60, 40, 40, 40, 40, 60],
        32768: [60, 40, 40, 60]}


def setup(N):
    from oracle import oracle as O
    from seal_fyp_logistic_regression_amd import Engine
    primes = O.coeff_modulus_create(N, BITS[N])
    return Engine(N, primes), O.Oracle(N, primes), primes


@pytest.mark.parametrize("N", sorted(BITS))
def test_samplers_bit_exact(N):
    e, o, primes = setup(N)
    k = len(primes)
    for kind in ("uniform", "ternary", "noise"):
        for stream_id, npoly, nrows, mod_first in ((1, 1, k, 0), (2 ** 40 + 5, 3, max(1, k - 1), k - max(1, k - 1))):
            got = e.sample(kind, KEY, stream_id, npoly, nrows, mod_first).download()
            want = o.sample(kind, KEY, stream_id, npoly, nrows, mod_first)
            assert (got == want).all(), (kind, stream_id)


def test_uniform_rejection_path_bit_exact():
    """a 27-bit prime rejects 2^64 mod q / 2^64 of the words -- far too rare to see; a modulus just above 2^63 is
    not a valid NTT prime.  Exercise the redraw logic through a key search instead: find a (key, stream) whose
    first block contains a rejected word for the 60-bit prime, bound = q*floor(2^64/q)."""
    e, o, primes = setup(8192)
    q = primes[0]
    bound = (2 ** 64 - 1) // q * q
    # P(reject) per word = 1 - bound/2^64 ~ 2^-4 .. 2^-60 depending on q; for 0xffffffffffe8001*16 it is ~1e-13:
    # cannot be hit by search.  What CAN be checked exactly: GPU == oracle over many streams (any divergence in
    # the redraw bookkeeping would show up as soon as one side redraws), plus the bound itself.
    assert bound % q == 0 and 2 ** 64 - bound < q
    for s in range(20, 28):
        assert (e.sample("uniform", KEY, s, 2, 4).download() == o.sample("uniform", KEY, s, 2, 4)).all()


@pytest.mark.parametrize("N,bits", [(8192, [60, 40, 40, 60]), (32768, [60, 40, 60])])
def test_encrypt_decrypt_bit_exact_and_round_trip(N, bits):
    from seal_fyp_logistic_regression_amd import seal as S
    from tests.oracle_backend import OracleBackend
    outs = {}
    for kind in ("gpu", "oracle"):
        parms = S.EncryptionParameters("ckks")
        parms.set_poly_modulus_degree(N)
        parms.set_coeff_modulus(S.CoeffModulus.Create(N, bits))
        ctx = S.SEALContext.Create(parms, backend=OracleBackend(N, parms.coeff_modulus()) if kind == "oracle" else None)
        kg = S.KeyGenerator(ctx, 11)
        enc, dec = S.Encryptor(ctx, kg.public_key(), 12), S.Decryptor(ctx, kg.secret_key())
        encoder, ev = S.CKKSEncoder(ctx, device_encode=False), S.Evaluator(ctx)
        v = np.linspace(-2, 2, 64)
        ct = enc.encrypt(encoder.encode(v, 2.0 ** 40))
        ct2 = enc.encrypt(encoder.encode(v, 2.0 ** 40))           # second call: fresh stream
        prod = ev.multiply(ct, ct2)                                # size 3
        be = ctx.backend
        outs[kind] = dict(sk=kg.secret_key().host, pk=kg.public_key(), ct=be.to_host(ct.data), ct2=be.to_host(ct2.data),
                          pt=be.to_host(dec.decrypt(ct).data), pt3=be.to_host(dec.decrypt(prod).data),
                          val=encoder.decode(dec.decrypt(ct))[:64].real,
                          val3=encoder.decode(dec.decrypt(prod))[:64].real)
    g, o = outs["gpu"], outs["oracle"]
    for name in ("sk", "pk", "ct", "ct2", "pt", "pt3"):
        assert (np.asarray(g[name]).reshape(-1) == np.asarray(o[name]).reshape(-1)).all(), name
    assert (g["ct"] != g["ct2"]).mean() > 0.99                     # a new stream id per encrypt call
    v = np.linspace(-2, 2, 64)
    assert np.abs(g["val"] - v).max() < 1e-6 and np.abs(g["val3"] - v * v).max() < 1e-5


def test_sampling_rejects_bad_arguments():
    e, _, _ = setup(4096)
    with pytest.raises(ValueError):
        e.sample("uniform", KEY, 1, 1, 4)            # only 3 primes
    with pytest.raises(ValueError):
        e.sample("noise", KEY[:31], 1, 1, 1)
    with pytest.raises(ValueError):
        e.sample("ternary", KEY, 1, 0, 1)

"""Counter-mode sampler, CPU side: the oracle's ChaCha20 against the RFC 7539 known answer, the noise table against
the distribution it encodes, and the three samplers against their specification (ranges, shares, determinism)."""
import numpy as np

from oracle import oracle as O

KEY = bytes(range(32))


def test_chacha20_block_rfc7539_section_2_3_2():
    # key 00..1f, block counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00 -> state words 12..15
    out = O.chacha20_block(KEY, 1 | (0x09000000 << 32), 0x4A000000)
    want = [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3, 0xC7F4D1C7, 0x0368C033, 0x9AAA2204, 0x4E6CD4C3,
            0x466482D2, 0x09AA9F07, 0x05D7C214, 0xA2028BD9, 0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]
    assert [int(x) for x in out] == want


def test_noise_thresholds_encode_the_clipped_truncated_normal():
    import math
    t = O.noise_thresholds()
    assert (np.diff(t[:38].astype(np.float64)) > 0).all() and int(t[38]) == 2 ** 64 - 1
    p = np.diff(np.concatenate([[0.0], t[:38].astype(np.float64), [2.0 ** 64]])) / 2.0 ** 64
    assert abs(p.sum() - 1) < 1e-15 and np.allclose(p, p[::-1], rtol=0, atol=1e-12)  # symmetric
    clipped = math.erf(19.2 / (3.2 * math.sqrt(2)))
    assert abs(p[19] - math.erf(1 / (3.2 * math.sqrt(2))) / clipped) < 1e-12       # P(0) = P(|x| < 1)
    want3 = 0.5 * (math.erf(4 / (3.2 * math.sqrt(2))) - math.erf(3 / (3.2 * math.sqrt(2)))) / clipped
    assert abs(p[19 + 3] - want3) < 1e-12 and abs(p[19 - 3] - want3) < 1e-12       # P(3) = P(3 <= x < 4)


def test_samplers_follow_their_specification():
    N = 4096
    primes = O.coeff_modulus_create(N, [50, 30, 30, 50])
    o = O.Oracle(N, primes)
    u = o.sample("uniform", KEY, 7, 2, 4)
    for j, q in enumerate(primes):
        assert (u[:, j] < q).all()
        assert abs(float(u[:, j].astype(np.float64).mean()) / q - 0.5) < 0.02
    t = o.sample("ternary", KEY, 8, 3, 4)
    sign = np.where(t[:, 0] == primes[0] - 1, -1, t[:, 0].astype(np.int64))
    assert set(np.unique(sign)) == {-1, 0, 1}
    for v in (-1, 0, 1):
        assert abs((sign == v).mean() - 1 / 3) < 0.02
    for j, q in enumerate(primes):  # the SAME draw in every row
        assert (np.where(t[:, j] == q - 1, -1, t[:, j].astype(np.int64)) == sign).all()
    e = o.sample("noise", KEY, 9, 8, 2, mod_first=1)
    q1 = primes[1]
    val = np.where(e[:, 0] > q1 // 2, e[:, 0].astype(np.int64) - q1, e[:, 0].astype(np.int64))
    assert np.abs(val).max() <= 19 and abs(val.mean()) < 0.05
    assert abs(val.std() - 2.828) < 0.05   # sqrt(sum k^2 p_k) of the truncated variable
    q2 = primes[2]
    assert (np.where(e[:, 1] > q2 // 2, e[:, 1].astype(np.int64) - q2, e[:, 1].astype(np.int64)) == val).all()
    # determinism and stream / key separation
    assert (o.sample("uniform", KEY, 7, 2, 4) == u).all()
    assert (o.sample("uniform", KEY, 6, 2, 4) != u).mean() > 0.99
    assert (o.sample("uniform", bytes(32), 7, 2, 4) != u).mean() > 0.99
    # position addressing: polynomial p of a 2-poly call == what row indices p*nrows+j of the counter say
    one = o.sample("ternary", KEY, 8, 1, 4)
    assert (one[0] == t[0]).all()

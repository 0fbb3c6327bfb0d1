"""The reference's OWN composite functions against this repository's composition, bit for bit ("the
composites are checked through shared host code ... a composition bug is common-mode").

drivers/_ref/gen_composite_vectors_shim is tools/gen_composite_vectors.cpp compiled with /root/reference/helper.h and
matrix_multiplication.cpp pulled in unchanged: Linear_Transform_Plain / _Cipher / _CipherMatrix_PlainVector,
C_Matrix_Encode / _Decode, cipher_dot_product, compute_all_powers and CC_Matrix_Multiplication are the REFERENCE'S C++,
running over include/seal/seal.h (recorded, fused, executed by the HIP engine).  It writes every input -- ciphertexts,
encoded diagonals, the whole default Galois key set, the relinearisation key -- and every result.  Here the same inputs go
through seal_fyp_logistic_regression_amd/algorithms.py twice, on the HIP engine and on the CPU oracle (tests/oracle_backend),
and all three results must be the same words, sizes, levels and scales.  Neither composition shares a line with the other."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _generate(tmp_path, which, mode):
    exe = os.path.join(ROOT, "drivers", "_ref", "gen_composite_vectors_shim")
    if not os.path.exists(exe):
        pytest.skip("drivers/_ref/gen_composite_vectors_shim is not built (make -C drivers needs /root/reference)")
    env = dict(os.environ, SEAL_SHIM_RESCALE=mode, HEFX_RESCALE=mode)
    r = subprocess.run([exe, str(tmp_path), which], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    from tests import seal_vectors as SV
    vec = SV.load(os.path.join(str(tmp_path), f"composites_{which}.bin"))
    assert not vec.from_real_seal  # the shim says so: nothing here pins SEAL
    return vec


def _replay(vec, mode, expect):
    from tests import seal_vectors as SV
    derived = {}
    for kind in ("gpu", "oracle"):
        res = SV.check_composites(vec, kind, rounded=mode == "round", derived=derived)
        want = set(expect) - ({"dec_row"} if kind == "oracle" else set())
        assert set(res) == want, (kind, sorted(res))
        assert all(res.values()), (kind, mode, {k: v for k, v in res.items() if not v})


@pytest.mark.parametrize("mode", ["round", "floor"])
def test_reference_composites_c2_same_words_as_this_repository(tmp_path, mode):
    """config 2's parameters: the three linear transforms (d = 4 with the reference's own 1..16 matrix, d = 16), the matrix
    packing pair, the dot product (one rescale: both divisions)"""
    vec = _generate(tmp_path, "c2", mode)
    assert (vec.N, len(vec.primes)) == (8192, 4)
    assert sum(r.kind == 3 and r.tag == "gk" for r in vec.records) == 24  # 3^(+-2^i), i < 12 (the two of i = 11 coincide), 2N - 1
    assert [len(vec.all(t)) for t in ("lt4_diag", "lt16_diag", "lt4_cdiag", "lt4_ptrot", "enc_row", "dec_row")] == [4, 16, 4, 4, 3, 3]
    _replay(vec, mode, ["lt4_plain", "lt16_plain", "lt4_cipher", "lt4_cmpv", "enc_packed", "dec_row", "dot"])


@pytest.mark.parametrize("mode", ["round", "floor"])
def test_reference_composites_c3_same_words_as_this_repository(tmp_path, mode):
    """config 3's parameters: compute_all_powers to degree 5 and the reference's CC_Matrix_Multiplication at n = 4, set up
    as its driver sets it up (all 128 diagonals with their epsilon)"""
    vec = _generate(tmp_path, "c3", mode)
    assert (vec.N, len(vec.primes)) == (16384, 6)
    assert [r.aux for r in vec.all("pow")] == [2, 3, 4, 5] and len(vec.all("mm_v")) == len(vec.all("mm_w")) == 48
    # mm_diagonals: this repository's U_sigma / U_tau / V_k / W_k (algorithms.matmul_permutation_matrices) and encoder
    # reproduce the 128 plaintexts the reference's get_U_* + CKKSEncoder produced, word for word
    _replay(vec, mode, ["pow", "mm_diagonals", "mm_out"])


def test_reference_matrix_product_config5_same_words_as_this_repository(tmp_path):
    """config 5 in the survey's reading (n = 8: 64 x 64 U matrices at N = 32768, all 1024 diagonals with their epsilon): the
    reference's CC_Matrix_Multiplication (matrix_multiplication.cpp:11-132 = matrix_mult_benchmark.cpp:13-71) through the shim
    against algorithms.py on the engine and on the oracle.  The file carries the inputs, the keys of the steps up to 64 and
    the result; the 1024 diagonal plaintexts (1.3 GB) are derived on the engine side, whose encoder is the producer's, and
    handed to the oracle side."""
    vec = _generate(tmp_path, "c5", "round")
    assert (vec.N, len(vec.primes)) == (32768, 6) and vec.get("mm_out").aux == 8 and not vec.has("mm_usig")
    _replay(vec, "round", ["mm_out"])


def test_a_wrong_word_or_a_wrong_scale_is_caught(tmp_path):
    """the replay's comparison is not vacuous"""
    from tests import seal_vectors as SV
    vec = _generate(tmp_path, "c2", "round")
    vec.get("lt4_plain").words[7] ^= 1
    vec.get("dot").scale *= 2.0
    res = SV.check_composites(vec, "gpu", rounded=True)
    assert res["lt4_plain"] is False and res["dot"] is False and res["lt16_plain"] is True and res["lt4_cipher"] is True


def _mt19937_64(seed):
    """std::mt19937_64: the generator include/seal/seal.h draws its sampler keys from when SEAL_SHIM_SEED is set"""
    M = (1 << 64) - 1
    mt = [seed & M]
    for i in range(1, 312):
        mt.append((6364136223846793005 * (mt[-1] ^ (mt[-1] >> 62)) + i) & M)
    idx = 312
    while True:
        if idx == 312:
            for i in range(312):
                x = (mt[i] & 0xFFFFFFFF80000000) | (mt[(i + 1) % 312] & 0x7FFFFFFF)
                mt[i] = mt[(i + 156) % 312] ^ (x >> 1) ^ (0xB5026F5AA96619E9 if x & 1 else 0)
            idx = 0
        y = mt[idx]
        idx += 1
        y ^= (y >> 29) & 0x5555555555555555
        y ^= (y << 17) & 0x71D67FFFEDA60000
        y ^= (y << 37) & 0xFFF7EEE000000000
        y ^= y >> 43
        yield y & M


def test_mt19937_64_is_the_standard_generator():
    g = _mt19937_64(5489)
    for _ in range(9999):
        next(g)
    assert next(g) == 9981545732273789042  # the C++ standard's check value: 10000th output of a default-seeded engine


@pytest.mark.parametrize("mode", ["round", "floor"])
def test_reference_lr_composites_c4_same_words_as_this_repository(tmp_path, mode):
    """config 4's parameters: the reference's Tree_cipher, Horner_cipher and predict_cipher_weights (six rows of eight
    weights) against algorithms.py -- whose prediction runs the rows' dot products in lockstep, encodes the masks in one
    batch and sums the masked rows in one pass, none of which the reference's loop does.  The functions encrypt a constant
    inside; the shim's Encryptors take their sampler keys from mt19937_64(SEAL_SHIM_SEED) in construction order
    (drivers/xcheck_lr.cpp), so the replay's Encryptors are given the same keys."""
    from seal_fyp_logistic_regression_amd import algorithms as alg
    from tests import seal_vectors as SV
    exe = os.path.join(ROOT, "drivers", "_ref", "xcheck_lr")
    if not os.path.exists(exe):
        pytest.skip("drivers/_ref/xcheck_lr is not built (make -C drivers needs /root/reference)")
    seed = 20261004
    env = dict(os.environ, SEAL_SHIM_RESCALE=mode, HEFX_RESCALE=mode, SEAL_SHIM_SEED=str(seed))
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    vec = SV.load(os.path.join(str(tmp_path), "lr_c4.bin"))
    assert (vec.N, len(vec.primes)) == (16384, 9)
    g = _mt19937_64(seed)
    keys = [bytes(next(g) & 0xFF for _ in range(32)) for _ in range(5)]  # KeyGenerator, input Encryptor, Encryptors 1..3
    coeffs, scale = [0.5, 1.20069, 0.00001, -0.81562], 2.0 ** 40
    pk = vec.ct("pk")
    assert pk.shape == (2, 9, 16384) and len(vec.all("feat")) == 6
    for kind in ("gpu", "oracle"):
        s = SV.CompositeSide(vec, kind, rounded=mode == "round")
        assert sorted(s.gk.keys) == sorted({SV.elt_from_step(1, vec.N), SV.elt_from_step(-8, vec.N)})

        def encryptor(i):
            e = s.S.Encryptor(s.ctx, pk)
            e._key32, e._stream = keys[i], 0
            return e

        x = s.ct(vec.get("poly_ct"))
        assert s.same(alg.tree_cipher(s.ev, s.encoder, encryptor(2), x, 3, scale, coeffs, s.rk), vec.get("tree")), (kind, "tree")
        assert s.same(alg.horner_cipher(s.ev, s.encoder, encryptor(3), x, 3, coeffs, scale, s.rk), vec.get("horner")), (kind, "horner")
        got = alg.predict_cipher_weights(s.ev, s.encoder, encryptor(4), s.cts("feat"), s.ct(vec.get("weights")), 8, scale, s.gk, s.rk,
                                         degree=3)
        assert s.same(got, vec.get("predict")), (kind, "predict")
        # the same Encryptor keys with the streams of another message: not the reference's words (the comparison bites)
        e = encryptor(3)
        e._stream = 1
        assert not s.same(alg.horner_cipher(s.ev, s.encoder, e, x, 3, coeffs, scale, s.rk), vec.get("horner"))

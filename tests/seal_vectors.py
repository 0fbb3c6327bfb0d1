"""Reader / writer / checker of the "HEFXKAT1" known-answer files that tools/gen_seal_vectors.cpp produces with REAL
Microsoft SEAL (format documented there and in tests/golden/seal/README.md).

`check(vec, impl)` recomputes every known answer of a file from the file's own inputs with `impl` -- the CPU oracle
(oracle.Oracle) or the HIP engine behind the C-ABI (EngineImpl below) -- and compares uint64 words.  Keys and
ciphertexts are inputs (SEAL's PRNG), evaluator outputs are the answers."""
from __future__ import annotations

import glob
import os
import struct
from typing import Dict, List

import numpy as np

MAGIC = b"HEFXKAT1"
KIND_CT, KIND_PT, KIND_KEY, KIND_STREAM = 1, 2, 3, 4
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seal")


class Record:
    def __init__(self, tag, kind, size, rows, aux, scale, words):
        self.tag, self.kind, self.size, self.rows, self.aux, self.scale, self.words = tag, kind, size, rows, aux, scale, words


class VectorFile:
    def __init__(self, path, N, primes, producer, records: List[Record]):
        self.path, self.N, self.primes, self.producer, self.records = path, N, primes, producer, records
        self.k = len(primes)

    def get(self, tag) -> Record:
        for r in self.records:
            if r.tag == tag:
                return r
        raise KeyError(tag)

    def has(self, tag) -> bool:
        return any(r.tag == tag for r in self.records)

    def all(self, tag) -> List[Record]:
        """the records of a list (tools/gen_composite_vectors.cpp: the index rides in aux), in index order"""
        return sorted((r for r in self.records if r.tag == tag), key=lambda r: r.aux)

    def ct(self, tag) -> np.ndarray:
        r = self.get(tag)
        assert r.kind == KIND_CT
        return r.words.reshape(r.size, r.rows, self.N)

    def pt(self, tag) -> np.ndarray:
        r = self.get(tag)
        assert r.kind == KIND_PT
        return r.words.reshape(r.rows, self.N)

    def key(self, tag, elt) -> np.ndarray:
        for r in self.records:
            if r.tag == tag and r.kind == KIND_KEY and r.aux == elt:
                assert r.size == self.k - 1 and r.rows == self.k, "key layout is not [k-1][2][k][N]"
                return r.words.reshape(self.k - 1, 2, self.k, self.N)
        raise KeyError((tag, elt))

    def stream(self, tag) -> bytes:
        """the bytes a save() member wrote (record kind 4; aux = byte count)"""
        r = self.get(tag)
        assert r.kind == KIND_STREAM
        return np.ascontiguousarray(r.words, dtype="<u8").tobytes()[:r.aux]

    @property
    def from_real_seal(self) -> bool:
        return self.producer.startswith("Microsoft SEAL")


def load(path: str) -> VectorFile:
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != MAGIC:
        raise ValueError(f"{path}: not a HEFXKAT1 file")
    version, N, k, _ = struct.unpack_from("<IIII", data, 8)
    if version != 1:
        raise ValueError(f"{path}: unsupported version {version}")
    off = 24
    primes = list(struct.unpack_from(f"<{k}Q", data, off))
    off += 8 * k
    producer = data[off:off + 64].split(b"\0")[0].decode()
    off += 64
    recs = []
    while off < len(data):
        tag = data[off:off + 24].split(b"\0")[0].decode()
        kind, size, rows, aux, scale, nwords = struct.unpack_from("<IIIIdQ", data, off + 24)
        off += 24 + 16 + 8 + 8
        words = np.frombuffer(data, dtype="<u8", count=nwords, offset=off).astype(np.uint64)
        off += 8 * nwords
        recs.append(Record(tag, kind, size, rows, aux, scale, words))
    return VectorFile(path, N, primes, producer, recs)


def write(path: str, N: int, primes, producer: str, records: List[Record]):
    """Same byte layout as tools/gen_seal_vectors.cpp (used by the self-check test and by anyone converting vectors)."""
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<IIII", 1, N, len(primes), 0) + struct.pack(f"<{len(primes)}Q", *primes))
        f.write(producer.encode()[:63].ljust(64, b"\0"))
        for r in records:
            w = np.ascontiguousarray(r.words, dtype="<u8").reshape(-1)
            f.write(r.tag.encode()[:23].ljust(24, b"\0") + struct.pack("<IIIIdQ", r.kind, r.size, r.rows, r.aux, r.scale, w.size))
            f.write(w.tobytes())


def golden_files() -> List[str]:
    """files of tools/gen_seal_vectors.cpp (Evaluator members)"""
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "seal_*.bin")))


def golden_composite_files() -> List[str]:
    """files of tools/gen_composite_vectors.cpp (the reference's own composite functions)"""
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "composites_*.bin")))


def all_golden_files() -> List[str]:
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.bin")))


def elt_from_step(step: int, N: int) -> int:
    pos = step if step > 0 else N // 2 + step
    return pow(3, pos, 2 * N)


class EngineImpl:
    """The oracle's method names over the HIP engine (every call goes through the C-ABI)."""

    def __init__(self, N, primes):
        from seal_fyp_logistic_regression_amd import Engine
        self.e, self.N = Engine(N, primes), N

    def _L(self, ct):
        return ct.shape[-2]

    def apply_galois(self, ct, elt, key):
        return self.e.apply_galois(self._L(ct), self.e.to_device(ct), elt, self.e.to_device(key)).download()

    def rotate_mulplain(self, ct, elt, key, pt):
        e = self.e
        return e.rotate_multiply_plain_batch(self._L(ct), [e.to_device(ct)], [elt], [e.to_device(key)], [e.to_device(pt)])[0].download()

    def multiply_plain(self, ct, pt):
        return self.e.multiply_plain(self._L(ct), ct.shape[0], self.e.to_device(ct), self.e.to_device(pt)).download()

    def add_plain(self, ct, pt):
        return self.e.add_plain(self._L(ct), ct.shape[0], self.e.to_device(ct), self.e.to_device(pt)).download()

    def add(self, a, b):
        return self.e.add(self._L(a), a.shape[0], self.e.to_device(a), self.e.to_device(b)).download()

    def multiply(self, a, b):
        return self.e.multiply(self._L(a), self.e.to_device(a), self.e.to_device(b)).download()

    def relinearize(self, ct3, key):
        return self.e.relinearize(self._L(ct3), self.e.to_device(ct3), self.e.to_device(key)).download()

    def rescale(self, ct, rounded=False):
        return self.e.rescale_to_next(self._L(ct), ct.shape[0], self.e.to_device(ct), rounded=rounded).download()

    def mod_drop(self, x, L_out):
        return self.e.mod_drop(self._L(x), L_out, x.shape[0], self.e.to_device(x)).download()


def check(vec: VectorFile, impl) -> Dict[str, object]:
    """-> {answer tag: True / False} plus "rescale_mode": "floor" | "round" | None (which division reproduces the file's
    rescale answers).  Missing optional records are skipped."""
    N = vec.N
    ct, ct_b, pt = vec.ct("ct"), vec.ct("ct_b"), vec.pt("pt")
    e1, em1, e4, econj = elt_from_step(1, N), elt_from_step(-1, N), elt_from_step(4, N), 2 * N - 1
    res: Dict[str, object] = {}
    eq = lambda a, b: bool(a.shape == b.shape and (a == b).all())
    rot1 = impl.apply_galois(ct, e1, vec.key("gk", e1))
    res["rot1"] = eq(rot1, vec.ct("rot1"))
    res["rot1_mulpt"] = eq(impl.multiply_plain(vec.ct("rot1"), pt), vec.ct("rot1_mulpt"))
    if hasattr(impl, "rotate_mulplain"):  # the fused hot-loop body (the metric's unit)
        res["rot1_mulpt_fused"] = eq(impl.rotate_mulplain(ct, e1, vec.key("gk", e1), pt), vec.ct("rot1_mulpt"))
    res["rotm1"] = eq(impl.apply_galois(ct, em1, vec.key("gk", em1)), vec.ct("rotm1"))
    # rotate_vector(ct, 3) with power-of-two keys: NAF(3) = [-1, 4], least significant term first
    naf = impl.apply_galois(impl.apply_galois(ct, em1, vec.key("gk", em1)), e4, vec.key("gk", e4))
    res["rot3_naf"] = eq(naf, vec.ct("rot3_naf"))
    res["conj"] = eq(impl.apply_galois(ct, econj, vec.key("gk", econj)), vec.ct("conj"))
    res["mulpt"] = eq(impl.multiply_plain(ct, pt), vec.ct("mulpt"))
    res["add"] = eq(impl.add(ct, ct_b), vec.ct("add"))
    res["addpl"] = eq(impl.add_plain(ct, pt), vec.ct("addpl"))
    res["mul"] = eq(impl.multiply(ct, ct_b), vec.ct("mul"))
    res["sq"] = eq(impl.multiply(ct, ct), vec.ct("sq"))
    res["relin"] = eq(impl.relinearize(vec.ct("mul"), vec.key("rk", 0)), vec.ct("relin"))
    res["rescale_mode"] = None
    if vec.has("rescale"):
        modes = {}
        for name, rounded in (("floor", False), ("round", True)):
            modes[name] = eq(impl.rescale(vec.ct("relin"), rounded=rounded), vec.ct("rescale")) and \
                eq(impl.rescale(vec.ct("mul"), rounded=rounded), vec.ct("rescale3"))
        res["rescale"] = modes["floor"] or modes["round"]
        res["rescale_mode"] = "floor" if modes["floor"] else ("round" if modes["round"] else None)
        low = impl.mod_drop(ct, ct.shape[1] - 1)
        res["modsw"] = eq(low, vec.ct("modsw"))
        res["rot1_low"] = eq(impl.apply_galois(vec.ct("modsw"), e1, vec.key("gk", e1)), vec.ct("rot1_low"))
    return res


# ------------------------------------------------------------------------------------------------------------------
# SEAL 3.4.5's uncompressed stream layouts as include/seal/seal.h writes them (shim_io.h: "format unpinned" until a file
# from REAL SEAL has passed check_streams) -- restated here independently of the C++ so that the same check pins both
# ------------------------------------------------------------------------------------------------------------------
def seal_parms_id(N: int, primes, scheme: int = 2, plain_modulus: int = 0):
    """EncryptionParameters::compute_parms_id: SHA3-256 over the uint64 words (scheme, N, primes..., plain modulus)"""
    import hashlib
    d = hashlib.sha3_256(struct.pack(f"<{3 + len(primes)}Q", scheme, N, *primes, plain_modulus)).digest()
    return struct.unpack("<4Q", d)


def parse_ciphertext_stream(b: bytes, off: int = 0):
    """Ciphertext::save -> (fields, words, next offset)"""
    pid = struct.unpack_from("<4Q", b, off)
    ntt = b[off + 32]
    size, n, rows = struct.unpack_from("<3Q", b, off + 33)
    scale, count = struct.unpack_from("<dQ", b, off + 57)
    words = np.frombuffer(b, dtype="<u8", count=count, offset=off + 73).astype(np.uint64)
    return dict(parms_id=pid, is_ntt_form=ntt, size=size, poly_modulus_degree=n, coeff_mod_count=rows, scale=scale), words, off + 73 + 8 * count


def check_streams(vec: VectorFile) -> Dict[str, bool]:
    """The three stream records of a vector file against the layouts the shim writes: parameters, a ciphertext (its words
    must be the file's own `ct` record) and a Galois-key set holding the key of step 1."""
    out = {}
    N, k = vec.N, vec.k
    key_id = seal_parms_id(N, vec.primes)
    first_id = seal_parms_id(N, vec.primes[:-1]) if k > 1 else key_id
    p = vec.stream("parms_stream")
    want = struct.pack("<BQQ", 2, N, k) + struct.pack(f"<{k}Q", *vec.primes) + struct.pack("<Q", 0)
    out["parms_stream"] = p == want
    f, words, end = parse_ciphertext_stream(vec.stream("ct_stream"))
    ct = vec.get("ct")
    out["ct_stream"] = (f["parms_id"] == first_id and f["is_ntt_form"] == 1 and f["size"] == ct.size and
                        f["poly_modulus_degree"] == N and f["coeff_mod_count"] == ct.rows and f["scale"] == ct.scale and
                        end == len(vec.stream("ct_stream")) and bool((words == ct.words).all()))
    g = vec.stream("gk1_stream")
    pid = struct.unpack_from("<4Q", g, 0)
    (dim1,) = struct.unpack_from("<Q", g, 32)
    off, ok, found = 40, pid == key_id and dim1 == N, {}
    for index in range(dim1 if ok else 0):
        (dim2,) = struct.unpack_from("<Q", g, off)
        off += 8
        comps = []
        for _ in range(dim2):
            cf, cw, off = parse_ciphertext_stream(g, off)
            ok = ok and cf["parms_id"] == key_id and cf["size"] == 2 and cf["coeff_mod_count"] == k and cf["is_ntt_form"] == 1
            comps.append(cw)
        if dim2:
            ok = ok and dim2 == k - 1
            found[2 * index + 1] = np.concatenate(comps)
    e1 = elt_from_step(1, N)
    ok = ok and off == len(g) and list(found) == [e1]
    if ok and vec.has("gk"):  # the generator's default key set and this one-key set come from the same KeyGenerator: same
        # secret key, fresh randomness -- the WORDS differ, the layout must match the `gk` record's
        ok = found[e1].size == vec.key("gk", e1).size
    out["gk1_stream"] = bool(ok)
    return out


# ------------------------------------------------------------------------------------------------------------------
# Composite files (tools/gen_composite_vectors.cpp, drivers/xcheck_lr.cpp): the answers are the results of the REFERENCE'S
# OWN functions (helper.h, matrix_multiplication.cpp, logistic_regression_ckks.cpp, compiled from where they lie); the
# replay below sends the file's inputs through this repository's composition, seal_fyp_logistic_regression_amd/algorithms.py,
# on the HIP engine ("gpu") or on the CPU oracle ("oracle")
# ------------------------------------------------------------------------------------------------------------------
class CompositeSide:
    """seal.py containers over one backend, filled from a vector file"""

    def __init__(self, vec: VectorFile, kind: str, rounded: bool):
        from seal_fyp_logistic_regression_amd import seal as S
        parms = S.EncryptionParameters("ckks")
        parms.set_poly_modulus_degree(vec.N)
        parms.set_coeff_modulus(vec.primes)
        self.S, self.vec, self.kind = S, vec, kind
        backend = None
        if kind == "oracle":
            from tests.oracle_backend import OracleBackend
            backend = OracleBackend(vec.N, vec.primes)
        self.ctx = S.SEALContext.Create(parms, backend=backend)
        self.be, self.ev = self.ctx.backend, S.Evaluator(self.ctx)
        self.be.rescale_rounded = bool(rounded)
        # device encode on the engine (the shim's CKKSEncoder is the same entry point), host encode on the oracle
        self.encoder = S.CKKSEncoder(self.ctx, device_encode=kind != "oracle")
        self.gk, self.rk = S.KSwitchKeys(), S.KSwitchKeys()
        for r in vec.records:
            if r.kind == KIND_KEY:
                assert r.size == vec.k - 1 and r.rows == vec.k, "key layout is not [k-1][2][k][N]"
                (self.rk if r.tag == "rk" else self.gk).keys[r.aux] = self.be.from_host(r.words.reshape(vec.k - 1, 2, vec.k, vec.N))

    def ct(self, rec: Record):
        return self.S.Ciphertext()._set(self.be.from_host(rec.words.reshape(rec.size, rec.rows, self.vec.N)), rec.size, rec.rows, rec.scale)

    def pt(self, rec: Record):
        p = self.S.Plaintext()
        p.data, p._parms_id, p.scale = self.be.from_host(rec.words.reshape(rec.rows, self.vec.N)), rec.rows, rec.scale
        return p

    def cts(self, tag):
        return [self.ct(r) for r in self.vec.all(tag)]

    def pts(self, tag):
        return [self.pt(r) for r in self.vec.all(tag)]

    def same(self, got, rec: Record) -> bool:
        """size, level, scale and every word"""
        if (got.size(), got.parms_id()) != (rec.size, rec.rows) or got.scale != rec.scale:
            return False
        words = self.be.to_host(got.data).reshape(-1)
        return bool(words.shape == rec.words.shape and (words == rec.words).all())


def check_composites(vec: VectorFile, kind: str, rounded: bool, derived: dict = None) -> Dict[str, bool]:
    """-> {answer tag: equal?} for every composite answer the file holds (sets c2 / c3 / c5 of tools/gen_composite_vectors.cpp).
    `derived` (optional dict, shared between the calls for one file): plaintext words one side had to encode itself -- set c5
    carries no diagonals -- are left there by the side that encoded them (the engine: the encoder the file's producer used)
    and taken from there by the other, so that both sides replay the same inputs.
    C_Matrix_Decode encodes its masks inside the function: on the oracle side the masks come from the host encoder, which
    may differ from SEAL's / the engine's in a last bit of a coefficient, so `dec_row` is replayed on the engine only."""
    from seal_fyp_logistic_regression_amd import algorithms as alg
    s = CompositeSide(vec, kind, rounded)
    res: Dict[str, bool] = {}
    for name in ("lt4", "lt16"):
        if vec.has(name + "_plain"):
            res[name + "_plain"] = s.same(alg.linear_transform_plain(s.ev, s.ct(vec.get(name + "_ct")), s.pts(name + "_diag"), s.gk),
                                          vec.get(name + "_plain"))
    if vec.has("lt4_cipher"):
        cdiags = s.cts("lt4_cdiag")
        res["lt4_cipher"] = s.same(alg.linear_transform_cipher(s.ev, s.ct(vec.get("lt4_ct")), cdiags, s.gk), vec.get("lt4_cipher"))
        res["lt4_cmpv"] = s.same(alg.linear_transform_ciphermatrix_plainvector(s.ev, s.pts("lt4_ptrot"), cdiags), vec.get("lt4_cmpv"))
    if vec.has("enc_packed"):
        res["enc_packed"] = s.same(alg.c_matrix_encode(s.ev, s.cts("enc_row"), s.gk), vec.get("enc_packed"))
        if kind != "oracle":
            rows = vec.all("dec_row")
            back = alg.c_matrix_decode(s.ev, s.encoder, s.ct(vec.get("enc_packed")), len(rows), vec.get("enc_row").scale, s.gk)
            res["dec_row"] = len(back) == len(rows) > 0 and all(s.same(g, r) for g, r in zip(back, rows))
    if vec.has("dot"):
        res["dot"] = s.same(alg.cipher_dot_product(s.ev, s.ct(vec.get("dot_a")), s.ct(vec.get("dot_b")), 8, s.rk, s.gk), vec.get("dot"))
    if vec.has("pow"):
        recs = vec.all("pow")
        powers = alg.compute_all_powers(s.ev, s.ct(vec.get("pow_ct")), max(r.aux for r in recs), s.rk)
        res["pow"] = all(s.same(powers[r.aux], r) for r in recs)
    if vec.has("mm_out"):
        out_rec = vec.get("mm_out")
        n = out_rec.aux or int(round(len(vec.all("mm_usig")) ** 0.5))
        top, scale = vec.get("mm_a").rows, vec.get("mm_a").scale
        sets = None
        if derived is not None and "mm" in derived:  # words another side encoded (the oracle side of a file without diagonals)
            sets = [[s.pt(Record("", KIND_PT, 1, top, 0, scale, w)) for w in ws] for ws in derived["mm"]]
        elif not vec.has("mm_usig") or kind != "oracle":
            # the diagonals as the driver forms them (matrix_multiplication.cpp:205-297): U_sigma, U_tau, V_k, W_k of
            # helper.h:702-851, every diagonal, plus 1e-8 -- derived here and encoded by this side's encoder
            Us, Ut, V, W = alg.matmul_permutation_matrices(n)
            sets = [[s.encoder.encode(dg + 1e-8, scale) for dg in alg.get_all_diagonals(U)] for U in [Us, Ut] + list(V) + list(W)]
            if derived is not None:
                derived["mm"] = [[s.be.to_host(p.data).reshape(-1) for p in ps] for ps in sets]
        if vec.has("mm_usig"):
            filed = [s.pts("mm_usig"), s.pts("mm_utau")]
            v, w = s.pts("mm_v"), s.pts("mm_w")
            assert len(filed[0]) == n * n and len(v) == len(w) == (n - 1) * n * n
            filed += [v[k * n * n:(k + 1) * n * n] for k in range(n - 1)] + [w[k * n * n:(k + 1) * n * n] for k in range(n - 1)]
            if sets is not None:  # this repository's permutation matrices and encoder give the reference's plaintext words
                res["mm_diagonals"] = all((s.be.to_host(a.data).reshape(-1) == s.be.to_host(b.data).reshape(-1)).all() and a.scale == b.scale
                                          for fa, fb in zip(sets, filed) for a, b in zip(fa, fb))
            sets = filed
        out = alg.cc_matrix_multiplication(s.ev, s.ct(vec.get("mm_a")), s.ct(vec.get("mm_b")), n, sets[0], sets[1], sets[2:n + 1],
                                           sets[n + 1:], s.gk)
        res["mm_out"] = s.same(out, out_rec)
    return res

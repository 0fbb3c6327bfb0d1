"""ctypes/numpy binding of the CPU oracle (oracle/ckks_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (see ckks_oracle.h).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; the product package never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HEFX_ORACLE_SO: another build of the same source (tests/test_oracle_sanitizers.py loads the ASan + UBSan one)
_SO = os.environ.get("HEFX_ORACLE_SO") or os.path.join(_HERE, "libckks_oracle.so")

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ckks_oracle.c")
    if os.environ.get("HEFX_ORACLE_SO"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libckks_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    vp, u64, i, d = C.c_void_p, C.c_uint64, C.c_int, C.c_double

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_is_prime", i, u64)
    sig("orc_coeff_modulus_create", i, u64, i32p, i, u64p)
    sig("orc_min_primitive_root", u64, u64, u64)
    sig("orc_mulmod", u64, u64, u64, u64)
    sig("orc_powmod", u64, u64, u64, u64)
    sig("orc_invmod", u64, u64, u64)
    sig("orc_ctx_create", vp, u64, u64p, i)
    sig("orc_ctx_destroy", None, vp)
    sig("orc_ctx_psi", u64, vp, i)
    sig("orc_ntt_fwd", None, vp, i, u64p)
    sig("orc_ntt_inv", None, vp, i, u64p)
    sig("orc_ntt_naive", None, vp, i, u64p, u64p)
    sig("orc_galois_elt_from_step", u64, u64, i)
    sig("orc_naf_steps", i, u64, i, i32p, i)
    sig("orc_galois_table", None, u64, u64, u32p)
    sig("orc_apply_galois_ntt", None, vp, u64, u64p, u64p)
    sig("orc_add", None, vp, i, i, u64p, u64p, u64p)
    sig("orc_sub", None, vp, i, i, u64p, u64p, u64p)
    sig("orc_negate", None, vp, i, i, u64p, u64p)
    sig("orc_add_plain", None, vp, i, i, u64p, u64p, u64p)
    sig("orc_multiply_plain", None, vp, i, i, u64p, u64p, u64p)
    sig("orc_multiply", None, vp, i, i, u64p, i, u64p, u64p)
    sig("orc_is_transparent", i, vp, i, i, u64p)
    sig("orc_switch_key", None, vp, i, u64p, u64p, u64p)
    sig("orc_apply_galois", None, vp, i, u64p, u64, u64p, u64p)
    sig("orc_apply_galois_hoisted", None, vp, i, u64p, u64, u64p, u64p)
    sig("orc_apply_galois_hoisted_exact", i, vp, i, u64p, u64, u64p, u64p)
    sig("orc_lt_double_hoisted_core", None, vp, i, u64p, i, u64p, u64p, u64p, u64p)
    sig("orc_relinearize", None, vp, i, u64p, u64p, u64p)
    sig("orc_rescale", None, vp, i, i, u64p, u64p, i)
    sig("orc_mod_drop", None, vp, i, i, i, u64p, u64p)
    sig("orc_rotate_mulplain", None, vp, i, u64p, u64, u64p, u64p, u64p)
    sig("orc_fill_uniform", None, vp, i, i, u64, u64p)
    sig("orc_gen_secret", None, vp, u64, u64p)
    sig("orc_gen_kswitch_key", None, vp, u64p, u64p, u64, u64p)
    sig("orc_gen_relin_key", None, vp, u64p, u64, u64p)
    sig("orc_gen_galois_key", None, vp, u64p, u64, u64, u64p)
    sig("orc_encrypt_sym", None, vp, i, u64p, u64p, u64, u64p)
    sig("orc_decrypt", None, vp, i, i, u64p, u64p, u64p)
    sig("orc_encode", None, vp, i, f64p, i, d, u64p)
    sig("orc_decode", None, vp, i, u64p, d, f64p)
    sig("orc_chacha20_block", None, u32p, u64, u64, u32p)
    sig("orc_noise_thresholds", None, u64p)
    for name in ("orc_sample_uniform", "orc_sample_ternary", "orc_sample_noise"):
        sig(name, None, vp, u32p, u64, i, i, i, u64p)
    _lib = L
    return L


def chacha20_block(key32: bytes, counter: int, nonce: int) -> np.ndarray:
    key = np.frombuffer(bytes(key32), dtype="<u4").copy()
    out = np.zeros(16, dtype=np.uint32)
    lib().orc_chacha20_block(key, counter, nonce, out)
    return out


def noise_thresholds() -> np.ndarray:
    t = np.zeros(39, dtype=np.uint64)
    lib().orc_noise_thresholds(t)
    return t


def coeff_modulus_create(N: int, bit_sizes) -> list[int]:
    bits = np.asarray(bit_sizes, dtype=np.int32)
    out = np.zeros(len(bits), dtype=np.uint64)
    rc = lib().orc_coeff_modulus_create(N, bits, len(bits), out)
    if rc:
        raise ValueError(f"coeff_modulus_create failed rc={rc}")
    return [int(x) for x in out]


def galois_elt_from_step(N: int, step: int) -> int:
    return int(lib().orc_galois_elt_from_step(N, step))


def naf_steps(N: int, step: int) -> list[int]:
    out = np.zeros(40, dtype=np.int32)
    n = lib().orc_naf_steps(N, step, out, 40)
    return [int(x) for x in out[:n]]


def galois_table(N: int, elt: int) -> np.ndarray:
    t = np.zeros(N, dtype=np.uint32)
    lib().orc_galois_table(N, elt, t)
    return t


class Oracle:
    """One CKKS parameter set: N, primes[0..k-1] (last = special prime P)."""

    def __init__(self, N: int, primes):
        self.N = int(N)
        self.primes = [int(p) for p in primes]
        self.k = len(self.primes)
        arr = np.asarray(self.primes, dtype=np.uint64)
        self._h = lib().orc_ctx_create(self.N, arr, self.k)
        if not self._h:
            raise ValueError("orc_ctx_create failed (prime not = 1 mod 2N?)")

    def __del__(self):
        try:
            if self._h:
                lib().orc_ctx_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- helpers
    def psi(self, j):
        return int(lib().orc_ctx_psi(self._h, j))

    def _new(self, *shape):
        return np.zeros(shape + (self.N,), dtype=np.uint64)

    # ---- NTT
    def ntt_fwd(self, j, a):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        lib().orc_ntt_fwd(self._h, j, a)
        return a

    def ntt_inv(self, j, a):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        lib().orc_ntt_inv(self._h, j, a)
        return a

    def ntt_naive(self, j, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        out = np.zeros_like(a)
        lib().orc_ntt_naive(self._h, j, a, out)
        return out

    def apply_galois_ntt(self, elt, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        out = np.zeros_like(a)
        lib().orc_apply_galois_ntt(self._h, elt, a, out)
        return out

    # ---- evaluator ops; payloads are [size, L, N] uint64
    def add(self, a, b):
        out = np.zeros_like(a)
        lib().orc_add(self._h, a.shape[1], a.shape[0], a, b, out)
        return out

    def sub(self, a, b):
        out = np.zeros_like(a)
        lib().orc_sub(self._h, a.shape[1], a.shape[0], a, b, out)
        return out

    def negate(self, a):
        out = np.zeros_like(a)
        lib().orc_negate(self._h, a.shape[1], a.shape[0], a, out)
        return out

    def add_plain(self, ct, pt):
        out = np.zeros_like(ct)
        lib().orc_add_plain(self._h, ct.shape[1], ct.shape[0], ct, pt, out)
        return out

    def multiply_plain(self, ct, pt):
        out = np.zeros_like(ct)
        lib().orc_multiply_plain(self._h, ct.shape[1], ct.shape[0], ct, pt, out)
        return out

    def multiply(self, a, b):
        out = self._new(a.shape[0] + b.shape[0] - 1, a.shape[1])
        lib().orc_multiply(self._h, a.shape[1], a.shape[0], a, b.shape[0], b, out)
        return out

    def is_transparent(self, ct):
        return bool(lib().orc_is_transparent(self._h, ct.shape[1], ct.shape[0], ct))

    def apply_galois(self, ct, elt, key):
        out = np.zeros_like(ct)
        lib().orc_apply_galois(self._h, ct.shape[1], ct, elt, key, out)
        return out

    def apply_galois_hoisted(self, ct, elt, key):
        """the UNCORRECTED hoisted rotation (decompose, then permute; rounds 1-3's fast mode): other words than SEAL's --
        the counter-example beside apply_galois_hoisted_exact"""
        out = np.zeros_like(ct)
        lib().orc_apply_galois_hoisted(self._h, ct.shape[1], np.ascontiguousarray(ct), elt, key, out)
        return out

    def apply_galois_hoisted_exact(self, ct, elt, key):
        """the hoisted sequence plus the flip-mask term (ks_mac_exact_kernel's identity): (words, took_regular_path)"""
        out = np.zeros_like(ct)
        z = lib().orc_apply_galois_hoisted_exact(self._h, ct.shape[1], np.ascontiguousarray(ct), elt, key, out)
        return out, bool(z)

    def lt_double_hoisted_core(self, ct_new, diags_keylevel, elts, keys):
        """second fast mode: ct_new [2][L][N], diags [d][k][N], elts[1..d-1] (index 0 ignored), keys list for l>=1"""
        d = len(diags_keylevel)
        L = ct_new.shape[1]
        diag = np.ascontiguousarray(np.stack(diags_keylevel), dtype=np.uint64)
        kk = np.ascontiguousarray(np.stack(keys), dtype=np.uint64) if keys else np.zeros(1, dtype=np.uint64)
        e = np.ascontiguousarray(np.asarray([0] + list(elts), dtype=np.uint64))
        out = np.zeros_like(ct_new)
        lib().orc_lt_double_hoisted_core(self._h, L, np.ascontiguousarray(ct_new), d, diag, e, kk, out)
        return out

    def switch_key(self, ct, target, key):
        ct = ct.copy()
        lib().orc_switch_key(self._h, ct.shape[1], ct, np.ascontiguousarray(target), key)
        return ct

    def relinearize(self, ct3, key):
        out = self._new(2, ct3.shape[1])
        lib().orc_relinearize(self._h, ct3.shape[1], ct3, key, out)
        return out

    def rescale(self, ct, rounded=True):
        out = self._new(ct.shape[0], ct.shape[1] - 1)
        lib().orc_rescale(self._h, ct.shape[1], ct.shape[0], ct, out, int(rounded))
        return out

    def mod_drop(self, x, L_out):
        """x: [npoly, L_in, N] -> [npoly, L_out, N]"""
        return np.ascontiguousarray(x[:, :L_out, :])

    def rotate_mulplain(self, ct, elt, key, pt):
        out = np.zeros_like(ct)
        lib().orc_rotate_mulplain(self._h, ct.shape[1], ct, elt, key, pt, out)
        return out

    def rotate_vector(self, ct, step, gkeys: dict):
        """SEAL rotate_internal (App. A.7): direct key if present, else NAF chain."""
        if step == 0:
            return ct.copy()
        elt = galois_elt_from_step(self.N, step)
        if elt in gkeys:
            return self.apply_galois(ct, elt, gkeys[elt])
        terms = naf_steps(self.N, step)
        if len(terms) == 1:
            raise ValueError("Galois key not present")
        for t in terms:
            if abs(t) == self.N // 2:
                continue
            ct = self.rotate_vector(ct, t, gkeys)
        return ct

    # ---- sampling / keys / encryption (non-hot; decrypted-value checks only)
    def uniform(self, L, npoly, seed):
        out = self._new(npoly, L)
        lib().orc_fill_uniform(self._h, L, npoly, seed, out)
        return out

    def gen_secret(self, seed):
        sk = self._new(self.k)
        lib().orc_gen_secret(self._h, seed, sk)
        return sk

    def gen_relin_key(self, sk, seed):
        out = self._new(self.k - 1, 2, self.k)
        lib().orc_gen_relin_key(self._h, sk, seed, out)
        return out

    def gen_galois_key(self, sk, elt, seed):
        out = self._new(self.k - 1, 2, self.k)
        lib().orc_gen_galois_key(self._h, sk, elt, seed, out)
        return out

    def default_galois_elts(self):
        """keygen.galois_keys() with no args (App. A.7): 3^(+-2^i) and 2N-1."""
        N = self.N
        elts = []
        logn = N.bit_length() - 1
        for i in range(logn - 1):
            elts.append(galois_elt_from_step(N, 1 << i))
            elts.append(galois_elt_from_step(N, -(1 << i)))
        elts.append(2 * N - 1)
        return sorted(set(elts))

    def gen_galois_keys(self, sk, steps=None, seed=0x6A1015):
        if steps is None:
            elts = self.default_galois_elts()
        else:
            elts = [galois_elt_from_step(self.N, s) for s in steps]
        return {e: self.gen_galois_key(sk, e, seed + 7919 * n) for n, e in enumerate(elts)}

    def encrypt(self, L, sk, pt, seed):
        ct = self._new(2, L)
        lib().orc_encrypt_sym(self._h, L, sk, pt, seed, ct)
        return ct

    def decrypt(self, ct, sk):
        pt = self._new(ct.shape[1])
        lib().orc_decrypt(self._h, ct.shape[1], ct.shape[0], ct, sk, pt)
        return pt

    # ---- counter-mode sampling (CPU statement of csrc/hefx_sample.hip)
    def sample(self, kind, key32, stream_id, npoly, nrows, mod_first=0):
        if len(key32) != 32:
            raise ValueError("key32 must be 32 bytes")
        key = np.frombuffer(bytes(key32), dtype="<u4").copy()
        out = np.empty((npoly, nrows, self.N), dtype=np.uint64)
        f = {"uniform": lib().orc_sample_uniform, "ternary": lib().orc_sample_ternary,
             "noise": lib().orc_sample_noise}[kind]
        f(self._h, key, stream_id, npoly, nrows, mod_first, out)
        return out

    def encode(self, L, values, scale):
        v = np.asarray(values, dtype=np.complex128)
        ri = np.empty(2 * len(v), dtype=np.float64)
        ri[0::2] = v.real
        ri[1::2] = v.imag
        pt = self._new(L)
        lib().orc_encode(self._h, L, ri, len(v), float(scale), pt)
        return pt

    def decode(self, pt, scale):
        ri = np.zeros(self.N, dtype=np.float64)
        lib().orc_decode(self._h, pt.shape[0], pt, float(scale), ri)
        return ri[0::2] + 1j * ri[1::2]

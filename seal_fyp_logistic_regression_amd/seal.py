"""Host-side mirror of the Microsoft SEAL 3.4.5 API surface the reference drivers use, on top
of the hefx engine.  Same names, argument meaning and error behaviour as `seal::` so that tests read like the
reference's code (e.g. /root/reference/helper.h:237-262 becomes algorithms.linear_transform_plain).

Split of work: level/scale/parms_id bookkeeping, NAF decomposition of rotation steps and SEAL's validity checks
live here (host); all RNS arithmetic goes through a backend -- `GpuBackend` (the HIP engine; the only backend this
package ships) or, in tests only, an oracle-backed twin with the same methods.  Sampling (keys, noise) and the
complex FFT of CKKSEncoder run on the host in numpy; their NTTs run on the GPU.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from .engine import Engine


# ----------------------------------------------------------------------------------------------
# number theory helpers (host logic; CoeffModulus::Create, App. A.3)
# ----------------------------------------------------------------------------------------------
def _is_prime(n: int) -> bool:
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


class CoeffModulus:
    @staticmethod
    def Create(poly_modulus_degree: int, bit_sizes: Sequence[int]) -> List[int]:
        """SEAL CoeffModulus::Create: per bit size walk down from 2^b - 2N + 1 in steps of 2N; primes of equal
        size are handed out smallest first (/root/reference/linear_transformation2.cpp:233)."""
        need: Dict[int, int] = {}
        for b in bit_sizes:
            if not 2 <= b <= 60:
                raise ValueError("bit_sizes is invalid")
            need[b] = need.get(b, 0) + 1
        table: Dict[int, List[int]] = {}
        for b, cnt in need.items():
            v, lower, found = (1 << b) - 2 * poly_modulus_degree + 1, 1 << (b - 1), []
            while len(found) < cnt and v > lower:
                if _is_prime(v):
                    found.append(v)
                v -= 2 * poly_modulus_degree
            if len(found) < cnt:
                raise ValueError("failed to find enough qualifying primes")
            table[b] = found
        return [table[b].pop() for b in bit_sizes]

    @staticmethod
    def BFVDefault(poly_modulus_degree: int) -> List[int]:
        """SEAL's hard-coded 128-bit-security defaults (util/globals.cpp) for the degrees the
        reference uses -- /root/reference/vector_ops.cpp:208 builds its CKKS context (BASELINE config 1) on this chain."""
        table = {
            4096: [0xffffee001, 0xffffc4001, 0x1ffffe0001],
            8192: [0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001],
            16384: [0xfffffffd8001, 0xfffffffa0001, 0xfffffff00001, 0x1fffffff68001, 0x1fffffff50001,
                    0x1ffffffee8001, 0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001],
        }
        if poly_modulus_degree not in table:
            raise ValueError("poly_modulus_degree is not supported by BFVDefault")
        return list(table[poly_modulus_degree])

    @staticmethod
    def MaxBitCount(poly_modulus_degree: int) -> int:
        return {1024: 27, 2048: 54, 4096: 109, 8192: 218, 16384: 438, 32768: 881}[poly_modulus_degree]


def galois_elt_from_step(step: int, N: int) -> int:
    """SEAL galois_elt_from_step (App. A.7)."""
    m = 2 * N
    if step == 0:
        return m - 1
    if abs(step) >= N // 2:
        raise ValueError("step count too large")
    pos = step if step > 0 else N // 2 + step
    return pow(3, pos, m)


def naf(value: int) -> List[int]:
    """SEAL util::naf: non-adjacent form, least significant term first, signed like `value`."""
    sign, value, res, i = value < 0, abs(value), [], 0
    while value:
        zi = 2 - (value & 3) if value & 1 else 0
        value = (value - zi) >> 1
        if zi:
            res.append((-zi if sign else zi) * (1 << i))
        i += 1
    return res


# ----------------------------------------------------------------------------------------------
# backend: the HIP engine
# ----------------------------------------------------------------------------------------------
class GpuBackend:
    """Adapter exposing the hefx C-ABI with handle-in / handle-out methods (handles are DeviceArray)."""

    name = "hip"

    def __init__(self, N: int, primes: Sequence[int], device: int = 0):
        self.engine = Engine(N, primes, device=device)
        self.N, self.primes, self.k = N, list(primes), len(primes)

    def from_host(self, a: np.ndarray):
        return self.engine.to_device(a)

    def to_host(self, h) -> np.ndarray:
        return h.download()

    def ntt_forward(self, h, npoly, nrows, mod_first=0):
        self.engine.ntt_forward(h, npoly, nrows, mod_first)
        return h

    def ntt_inverse(self, h, npoly, nrows, mod_first=0):
        self.engine.ntt_inverse(h, npoly, nrows, mod_first)
        return h

    def add(self, L, size, a, b):
        return self.engine.add(L, size, a, b)

    def sub(self, L, size, a, b):
        return self.engine.sub(L, size, a, b)

    def negate(self, L, size, a):
        return self.engine.negate(L, size, a)

    def add_plain(self, L, size, ct, pt):
        return self.engine.add_plain(L, size, ct, pt)

    def add_many(self, L, size, cts):
        return self.engine.add_many(L, size, cts)

    def multiply_plain(self, L, size, ct, pt):
        return self.engine.multiply_plain(L, size, ct, pt)

    def multiply_plain_sum(self, L, size, cts, pts, group=None):
        return self.engine.multiply_plain_sum(L, size, cts, pts, group)

    def multiply_batch(self, L, As, Bs):
        return self.engine.multiply_batch(L, As, Bs)

    def add_batch(self, L, size, As, Bs):
        return self.engine.add_batch(L, size, As, Bs)

    def rescale_batch(self, L, size, cts):
        return self.engine.rescale_batch(L, size, cts)

    def relinearize_batch(self, L, ct3s, key):
        return self.engine.relinearize_batch(L, ct3s, key)

    def multiply(self, L, a, b):
        return self.engine.multiply(L, a, b)

    def square(self, L, a):
        return self.engine.square(L, a)

    def apply_galois(self, L, ct, elt, key):
        return self.engine.apply_galois(L, ct, elt, key)

    def apply_galois_batch(self, L, cts, elts, keys):
        return self.engine.apply_galois_batch(L, cts, elts, keys)

    def rotate_multiply_plain_batch(self, L, cts, elts, keys, pts):
        return self.engine.rotate_multiply_plain_batch(L, cts, elts, keys, pts)

    def apply_galois_forest(self, L, parents, ext_ins, elts, keys, pts=None):
        """a forest of rotations in one engine call (hefx_apply_galois_forest); the oracle-backed twin of the tests has no
        such method and runs the same nodes depth by depth (algorithms._rotations_batched)"""
        return self.engine.apply_galois_forest(L, parents, ext_ins, elts, keys, pts)

    def rotate_add_chain(self, L, cts, elts, keys, accs, steps):
        """`steps` x (t = apply_galois(t); a = a + t) per pair, in lockstep (helper.h:472-476 as one engine call)"""
        return self.engine.rotate_add_chain(L, cts, elts, keys, accs, steps)

    def relinearize(self, L, ct3, key):
        return self.engine.relinearize(L, ct3, key)

    def rescale(self, L, size, ct):
        return self.engine.rescale_to_next(L, size, ct)

    # floor (SEAL 3.4.x as SURVEY App. A.9 states it, default) or round-to-nearest (SEAL >= 3.5) division in
    # rescale_to_next: the one [M]-confidence semantic of the bit-exact path, switchable per context
    @property
    def rescale_rounded(self) -> bool:
        return self.engine.rescale_rounded

    @rescale_rounded.setter
    def rescale_rounded(self, v: bool):
        self.engine.set_rescale_rounded(bool(v))

    def mod_drop(self, L_in, L_out, npoly, x):
        return self.engine.mod_drop(L_in, L_out, npoly, x)

    def reduce_canonical(self, L, size, h, addends):
        return self.engine.reduce_canonical(L, size, h, addends)

    def linear_transform_plain(self, L, ct, diag_pts, key_elts, keys, hoisted=False):
        return self.engine.linear_transform_plain(L, ct, diag_pts, key_elts, keys, hoisted=hoisted)

    def linear_transform_plain_many(self, L, cts, diag_pts, key_elts, keys):
        return self.engine.linear_transform_plain_many(L, cts, diag_pts, key_elts, keys)

    def rotate_hoisted_batch(self, L, ct, elts, keys, pts=None):
        return self.engine.rotate_hoisted_batch(L, ct, elts, keys, pts)

    def linear_transform_plain_hoisted2_sparse(self, L, ct, d, steps, diag_pts_keylevel, key_elts, keys):
        return self.engine.linear_transform_plain_hoisted2_sparse(L, ct, d, steps, diag_pts_keylevel, key_elts, keys)

    def linear_transform_plain_bsgs(self, L, ct, shifted_diag_pts, n1, key_elts, keys, hoisted=True):
        return self.engine.linear_transform_plain_bsgs(L, ct, shifted_diag_pts, n1, key_elts, keys, hoisted)

    def sample(self, kind, key32, stream_id, npoly, nrows, mod_first=0):
        return self.engine.sample(kind, key32, stream_id, npoly, nrows, mod_first)

    def keygen_kswitch(self, sk, new_sk, key32, stream_id):
        return self.engine.keygen_kswitch(sk, new_sk, key32, stream_id)

    def galois_permute(self, elt, a, rows):
        return self.engine.galois_permute(elt, a, rows)

    def encrypt(self, L, pk, plain, key32, stream_id):
        return self.engine.encrypt(L, pk, plain, key32, stream_id)

    def decrypt(self, L, size, ct, sk):
        return self.engine.decrypt(L, size, ct, sk)

    def ckks_decode(self, L, pt, scale):
        return self.engine.ckks_decode(L, pt, scale)[0]

    def ckks_encode(self, L, values, scale):
        """[count][nvalues] slot values -> [count][L][N] NTT-form plaintexts, or None if N is outside the kernel's range"""
        if not 1024 <= self.N <= 32768:
            return None
        return self.engine.ckks_encode(L, values, scale)


# ----------------------------------------------------------------------------------------------
# SEAL-shaped objects
# ----------------------------------------------------------------------------------------------
class EncryptionParameters:
    def __init__(self, scheme: str = "ckks"):
        if str(scheme).lower() != "ckks":
            raise ValueError("unsupported scheme (this engine implements the CKKS path only)")
        self._n = 0
        self._q: List[int] = []

    def set_poly_modulus_degree(self, n: int):
        self._n = int(n)

    def set_coeff_modulus(self, primes: Sequence[int]):
        self._q = [int(p) for p in primes]

    def poly_modulus_degree(self) -> int:
        return self._n

    def coeff_modulus(self) -> List[int]:
        return list(self._q)


class ContextData:
    def __init__(self, ctx: "SEALContext", parms_id: int):
        self._ctx, self._pid = ctx, parms_id

    def parms_id(self):
        return self._pid

    def chain_index(self) -> int:
        return self._pid - 1 if self._pid <= self._ctx.k - 1 or self._ctx.k == 1 else self._ctx.k - 1

    def coeff_modulus(self) -> List[int]:
        return self._ctx.primes[: self._pid]

    def total_coeff_modulus_bit_count(self) -> int:
        p = 1
        for q in self.coeff_modulus():
            p *= q
        return p.bit_length()

    def next_context_data(self):
        return ContextData(self._ctx, self._pid - 1) if self._pid > 1 else None


class SEALContext:
    """parms_id == number of RNS primes of the level: k for the key level, k-1 for the first data level ... 1."""

    def __init__(self, parms: EncryptionParameters, backend=None, device: int = 0):
        self.N = parms.poly_modulus_degree()
        self.primes = parms.coeff_modulus()
        self.k = len(self.primes)
        if self.N < 1024 or self.N & (self.N - 1):
            raise ValueError("poly_modulus_degree is not valid")
        self.backend = backend if backend is not None else GpuBackend(self.N, self.primes, device)

    @classmethod
    def Create(cls, parms, backend=None, device: int = 0):
        return cls(parms, backend, device)

    def key_parms_id(self) -> int:
        return self.k

    def first_parms_id(self) -> int:
        return self.k - 1 if self.k > 1 else 1

    def last_parms_id(self) -> int:
        return 1

    def key_context_data(self):
        return ContextData(self, self.k)

    def first_context_data(self):
        return ContextData(self, self.first_parms_id())

    def get_context_data(self, parms_id: int):
        return ContextData(self, parms_id)


class Plaintext:
    def __init__(self):
        self.data = None  # backend handle [L][N], NTT form
        self._parms_id = 0
        self._scale = 1.0
        self.is_zero = False  # known at encode time: lets multiply_plain raise "transparent" without a sync

    def parms_id(self):
        return self._parms_id

    @property
    def scale(self):
        return self._scale

    @scale.setter
    def scale(self, v):
        self._scale = float(v)


class Ciphertext:
    def __init__(self):
        self.data = None  # backend handle [size][L][N], NTT form
        self._size = 0
        self._parms_id = 0
        self._scale = 1.0

    def size(self) -> int:
        return self._size

    def parms_id(self):
        return self._parms_id

    @property
    def scale(self):
        return self._scale

    @scale.setter
    def scale(self, v):
        self._scale = float(v)

    def _set(self, data, size, parms_id, scale):
        self.data, self._size, self._parms_id, self._scale = data, size, parms_id, float(scale)
        return self

    def copy(self) -> "Ciphertext":  # value semantics of seal::Ciphertext; payloads are never mutated in place
        return Ciphertext()._set(self.data, self._size, self._parms_id, self._scale)


class SecretKey:
    def __init__(self, host: np.ndarray, handle):
        self.host, self.data = host, handle  # [k][N] NTT form


class KSwitchKeys:
    """GaloisKeys / RelinKeys: galois element (or 0 for the relin key) -> device key [k-1][2][k][N]."""

    def __init__(self):
        self.keys: Dict[int, object] = {}

    def has_key(self, elt: int) -> bool:
        return elt in self.keys

    def key(self, elt: int):
        return self.keys[elt]


GaloisKeys = KSwitchKeys
RelinKeys = KSwitchKeys


def _key32(seed) -> bytes:
    """32-byte ChaCha20 key of the backend sampler: fresh OS randomness when seed is None (production), else a
    deterministic expansion of the integer seed (tests: randomness is an input of the parity chain)."""
    import hashlib
    if seed is None:
        return os.urandom(32)
    return hashlib.sha256(b"hefx-seed:" + str(int(seed)).encode()).digest()


class KeyGenerator:
    """Randomness from the backend's counter-mode sampler (hefx_sample_*: ChaCha20 keystream, SEAL 3.4.5's
    distributions; SEAL seeds from random_device, so keys are inputs to parity, never outputs); all modular
    arithmetic on the backend (App. A.11).  The default seed=None draws the sampler key from the OS (os.urandom);
    an integer seed is for tests and tools only -- it makes the secret key publicly derivable."""

    def __init__(self, context: SEALContext, seed: Optional[int] = None):
        self.ctx = context
        self._key32, self._stream = _key32(seed), 0
        be, k = context.backend, context.k
        h = be.sample("ternary", self._key32, 2 * self._next_stream(), 1, k)
        be.ntt_forward(h, 1, k, 0)
        self._sk = SecretKey(be.to_host(h).reshape(k, context.N), h)

    def _next_stream(self) -> int:
        self._stream += 1
        return self._stream

    def secret_key(self) -> SecretKey:
        return self._sk

    def _encrypt_zero(self, npoly: int, rows: int, sid: int):
        """npoly fresh symmetric encryptions of zero over the first `rows` primes: returns host (c0, c1);
        a uniform from sampler stream 2*sid, e noise from 2*sid+1 (what hefx_keygen_kswitch draws)"""
        be, sk = self.ctx.backend, self._sk
        a = be.sample("uniform", self._key32, 2 * sid, npoly, rows)
        e = be.sample("noise", self._key32, 2 * sid + 1, npoly, rows)
        be.ntt_forward(e, npoly, rows, 0)
        sk_rows = be.from_host(sk.host[:rows])
        as_ = be.multiply_plain(rows, npoly, a, sk_rows)
        c0 = be.negate(rows, npoly, be.add(rows, npoly, as_, e))
        return (be.to_host(c0).reshape(npoly, rows, self.ctx.N),
                be.to_host(a).reshape(npoly, rows, self.ctx.N))

    def public_key(self):
        # generated once, like SEAL's KeyGenerator (public_key() returns the same key on every call)
        if getattr(self, "_pk", None) is None:
            c0, c1 = self._encrypt_zero(1, self.ctx.k, self._next_stream())
            self._pk = np.stack([c0[0], c1[0]])  # [2][k][N]
        return self._pk

    def _kswitch_key(self, new_sk_host, new_sk_dev=None):
        """key-switching key for new_sk under sk.  On the HIP engine: one call (hefx_keygen_kswitch -- sampling, NTT and
        assembly on the device); otherwise the same arithmetic composed from backend ops (the oracle twin in tests:
        same sampler streams, same bits)."""
        ctx, be = self.ctx, self.ctx.backend
        k, N, q = ctx.k, ctx.N, ctx.primes
        sid = self._next_stream()
        native = getattr(be, "keygen_kswitch", None)
        if native is not None:
            if new_sk_dev is None:
                new_sk_dev = be.from_host(new_sk_host)
            return native(self._sk.data, new_sk_dev, self._key32, sid)
        if new_sk_host is None:
            new_sk_host = be.to_host(new_sk_dev).reshape(k, N)
        c0, c1 = self._encrypt_zero(k - 1, k, sid)
        P = q[k - 1]
        # c0[i][row i] += (P mod q_i) * new_sk[row i]
        factor = np.empty((k, N), dtype=np.uint64)
        for j in range(k):
            factor[j, :] = P % q[j]
        t = be.to_host(be.multiply_plain(k, 1, be.from_host(new_sk_host[None]), be.from_host(factor)))[0]
        for i in range(k - 1):
            s = c0[i, i] + t[i]
            c0[i, i] = np.where(s >= q[i], s - q[i], s)
        key = np.stack([c0, c1], axis=1)  # [k-1][2][k][N]
        return be.from_host(key)

    def relin_keys(self) -> KSwitchKeys:
        be, sk = self.ctx.backend, self._sk
        s2 = be.multiply_plain(self.ctx.k, 1, be.from_host(sk.host[None]), sk.data)
        rk = KSwitchKeys()
        rk.keys[0] = self._kswitch_key(None, s2)
        return rk

    def default_galois_elts(self) -> List[int]:
        N = self.ctx.N
        logn = N.bit_length() - 1
        elts = [2 * N - 1]
        for i in range(logn - 1):
            elts += [pow(3, 1 << i, 2 * N), pow(3, N // 2 - (1 << i), 2 * N)]
        return sorted(set(elts))

    def galois_keys(self, steps: Optional[Sequence[int]] = None) -> KSwitchKeys:
        """keygen.galois_keys(): default = 3^(+-2^i) and 2N-1 (power-of-two steps only, App. A.7)."""
        from . import galois_tables
        N = self.ctx.N
        elts = self.default_galois_elts() if steps is None else [galois_elt_from_step(s, N) for s in steps]
        gk = KSwitchKeys()
        permute = getattr(self.ctx.backend, "galois_permute", None)
        for g in elts:
            if permute is not None:  # s(X^g) on the device
                gk.keys[g] = self._kswitch_key(None, permute(g, self._sk.data, self.ctx.k))
            else:
                tab = galois_tables.gather_table(N, g)
                gk.keys[g] = self._kswitch_key(self._sk.host[:, tab])
        return gk


class Encryptor:
    """(pk0*u + e0 + m, pk1*u + e1) over the plaintext's level in ONE engine call (hefx_encrypt: sampling, NTT and
    the dyadic arithmetic all on the GPU).  The default seed=None draws the sampler key from the OS;
    an integer seed (tests only) replays the same (u, e0, e1) for the i-th ciphertext of every such instance."""

    def __init__(self, context: SEALContext, public_key: np.ndarray, seed: Optional[int] = None):
        self.ctx, self.pk = context, public_key
        self._pk_dev = context.backend.from_host(np.ascontiguousarray(public_key))  # [2][k][N]
        self._key32, self._stream = _key32(seed), 0

    def encrypt(self, plain: Plaintext, destination: Optional[Ciphertext] = None) -> Ciphertext:
        L = plain.parms_id()
        self._stream += 1
        c = self.ctx.backend.encrypt(L, self._pk_dev, plain.data, self._key32, self._stream)
        out = destination if destination is not None else Ciphertext()
        return out._set(c, 2, L, plain.scale)


class Decryptor:
    def __init__(self, context: SEALContext, secret_key: SecretKey):
        self.ctx, self.sk = context, secret_key

    def decrypt(self, encrypted: Ciphertext, destination: Optional[Plaintext] = None) -> Plaintext:
        """c0 + c1 s (+ c2 s^2): handles size-3 ciphertexts (/root/reference/matrix_multiplication.cpp:419);
        one engine call (hefx_decrypt)."""
        L = encrypted.parms_id()
        out = destination if destination is not None else Plaintext()
        out.data = self.ctx.backend.decrypt(L, encrypted.size(), encrypted.data, self.sk.data)
        out._parms_id, out._scale = L, encrypted.scale
        return out


class CKKSEncoder:
    """Canonical embedding with slot i <-> root zeta^(3^i) (App. A.12).  encode runs on the GPU
    (hefx_ckks_encode: FFT + rounding + RNS + NTT in two launches) when the backend offers it; decode, scalars and
    coefficients wider than 62 bits use the host FFT + the backend NTT.  device_encode=False forces
    the host FFT (bit-identical across backends -- what the evaluator parity tests use)."""

    def __init__(self, context: SEALContext, device_encode: bool = True):
        self.ctx = context
        self.device_encode = bool(device_encode)
        N = context.N
        pos = np.empty(N // 2, dtype=np.int64)
        p = 1
        for i in range(N // 2):
            pos[i] = p
            p = p * 3 % (2 * N)
        self._r1 = (pos - 1) >> 1
        self._r2 = (2 * N - pos - 1) >> 1
        self._zeta = np.exp(1j * np.pi * np.arange(N) / N)

    def slot_count(self) -> int:
        return self.ctx.N // 2

    def _to_rns(self, coeffs: np.ndarray, L: int) -> np.ndarray:
        q = self.ctx.primes
        out = np.empty((L, self.ctx.N), dtype=np.uint64)
        if np.abs(coeffs).max(initial=0.0) < 2.0 ** 62:
            ci = coeffs.astype(np.int64)
            for j in range(L):
                out[j] = np.mod(ci, q[j]).astype(np.uint64)
        else:  # wide coefficients (scale > 2^62): exact big-int path
            ci = [int(c) for c in coeffs]
            for j in range(L):
                out[j] = np.array([c % q[j] for c in ci], dtype=np.uint64)
        return out

    def encode(self, values, scale: float, destination: Optional[Plaintext] = None,
               parms_id: Optional[int] = None) -> Plaintext:
        ctx, be, N = self.ctx, self.ctx.backend, self.ctx.N
        L = parms_id if parms_id is not None else ctx.first_parms_id()
        out = destination if destination is not None else Plaintext()
        if np.isscalar(values):  # encode(double, scale, pt): every NTT slot = round(v*scale) mod q
            c = int(round(float(values) * scale))
            rows = np.empty((L, N), dtype=np.uint64)
            for j in range(L):
                rows[j, :] = c % ctx.primes[j]
            out.data, out.is_zero = be.from_host(rows), c == 0
        else:
            v = np.asarray(values)
            if v.size > N // 2:
                raise ValueError("values has invalid size")
            dev = self._encode_device(v.reshape(1, -1), scale, L) if v.size else None
            if dev is not None:
                out.data, out.is_zero = dev[0].view(0, (L, N)), dev[1][0]
                out._parms_id, out._scale = L, float(scale)
                return out
            v = v.astype(np.complex128)
            A = np.zeros(N, dtype=np.complex128)
            A[self._r1[: v.size]] = v
            A[self._r2[: v.size]] = np.conj(v)
            a = np.fft.fft(A) / N
            coeffs = np.rint(np.real(a * np.conj(self._zeta)) * scale)
            rows = self._to_rns(coeffs, L)
            out.is_zero = not coeffs.any()
            out.data = be.ntt_forward(be.from_host(rows), 1, L, 0)
        out._parms_id, out._scale = L, float(scale)
        if math.log2(scale) >= ContextData(ctx, L).total_coeff_modulus_bit_count():
            raise ValueError("scale out of bounds")
        return out

    def _encode_device(self, v2d: np.ndarray, scale: float, L: int):
        """GPU encode (hefx_ckks_encode) when the backend has it and every coefficient provably fits 62 bits and
        zero-ness is decidable from norms; returns (slab [count][L][N], is_zero[count]) or None -> host path."""
        be, N = self.ctx.backend, self.ctx.N
        enc = getattr(be, "ckks_encode", None) if self.device_encode else None
        if enc is None or not (scale > 0) or math.log2(scale) >= ContextData(self.ctx, L).total_coeff_modulus_bit_count():
            return None
        mag = np.abs(v2d)
        if not np.all(np.isfinite(mag)) or float(mag.max(initial=0.0)) * scale >= 2.0 ** 62:
            return None  # |p_k| <= max|v|: wide coefficients take the exact big-int host path
        # Parseval: sum p_k^2 = (2/N) sum |v_i|^2, so max|p_k| >= sqrt(2 sum|v|^2)/N; and max|p_k| <= max|v|
        hi = mag.max(axis=1) * scale
        lo = np.sqrt(2.0 * (mag.astype(np.float64) ** 2).sum(axis=1)) / N * scale
        zero, nonzero = hi < 0.499, lo > 0.501
        if not np.all(zero | nonzero):
            return None
        slab = enc(L, v2d, scale)
        if slab is None:
            return None
        return slab, [bool(z) for z in zero]

    def encode_many(self, vectors, scale: float, parms_id: Optional[int] = None) -> List[Plaintext]:
        """Encodes equally long vectors in one launch (the d diagonals of Linear_Transform_Plain,
        matrix_mult_benchmark.cpp:291-323; the one-hot masks of logistic_regression_ckks.cpp:222-225)."""
        ctx, N = self.ctx, self.ctx.N
        L = parms_id if parms_id is not None else ctx.first_parms_id()
        vs = [np.asarray(v) for v in vectors]
        if vs and all(v.ndim == 1 and v.size == vs[0].size and 0 < v.size <= N // 2 for v in vs):
            dev = self._encode_device(np.stack(vs), scale, L)
            if dev is not None:
                outs = []
                for i in range(len(vs)):
                    pt = Plaintext()
                    pt.data, pt.is_zero = dev[0].view(i * L * N, (L, N)), dev[1][i]
                    pt._parms_id, pt._scale = L, float(scale)
                    outs.append(pt)
                return outs
        return [self.encode(v, scale, None, L) for v in vs]

    def decode(self, plain: Plaintext) -> np.ndarray:
        ctx, be, N, L = self.ctx, self.ctx.backend, self.ctx.N, plain.parms_id()
        dev = getattr(be, "ckks_decode", None) if self.device_encode else None
        if dev is not None and L <= 16:  # inverse NTT, CRT, centring and the slot-root FFT on the GPU
            return dev(L, plain.data, plain.scale)
        h = be.from_host(be.to_host(plain.data))
        rows = be.to_host(be.ntt_inverse(h, 1, L, 0))
        q = ctx.primes[:L]
        Q = 1
        for p in q:
            Q *= p
        acc = np.zeros(N, dtype=object)
        for j in range(L):  # CRT compose
            Qj = Q // q[j]
            acc = (acc + rows[j].astype(object) * (Qj * pow(Qj, -1, q[j]))) % Q
        centered = np.array([float(c - Q) if c > Q // 2 else float(c) for c in acc]) / plain.scale
        A = np.fft.ifft(centered * self._zeta) * N
        return A[self._r1]


class Evaluator:
    """seal::Evaluator for CKKS.  Out-of-place forms return the destination; *_inplace forms rebind the
    ciphertext's payload (payload buffers are immutable once produced, so copies are O(1))."""

    def __init__(self, context: SEALContext):
        self.ctx, self.be = context, context.backend
        self._bitcount: Dict[int, int] = {}

    # ---- checks shared with SEAL
    @staticmethod
    def _close(a: float, b: float) -> bool:
        return abs(a - b) <= max(abs(a), abs(b)) * 2.0 ** -40 or a == b

    def _check_same(self, a, b):
        if a.parms_id() != b.parms_id():
            raise ValueError("encrypted1 and encrypted2 parameter mismatch")

    def _check_scale(self, new_scale: float, parms_id: int):
        bits = self._bitcount.get(parms_id)
        if bits is None:
            bits = self._bitcount[parms_id] = ContextData(self.ctx, parms_id).total_coeff_modulus_bit_count()
        if new_scale <= 0 or int(math.log2(new_scale)) >= bits:
            raise ValueError("scale out of bounds")

    # ---- add / sub / negate
    def _addsub(self, a: Ciphertext, b: Ciphertext, sub: bool) -> Ciphertext:
        self._check_same(a, b)
        if not self._close(a.scale, b.scale):
            raise ValueError("scale mismatch")
        L = a.parms_id()
        if a.size() != b.size():  # result size = max; the extra polys are copied (negated for b in sub)
            big, small = (a, b) if a.size() > b.size() else (b, a)
            hb = self.be.to_host(big.data)
            pad = np.zeros_like(hb)
            pad[: small.size()] = self.be.to_host(small.data)
            small = Ciphertext()._set(self.be.from_host(pad), big.size(), L, small.scale)
            a, b = (big, small) if a.size() > b.size() else (small, big)
        f = self.be.sub if sub else self.be.add
        return Ciphertext()._set(f(L, a.size(), a.data, b.data), a.size(), L, a.scale)

    def add(self, a, b, destination=None):
        r = self._addsub(a, b, False)
        return r if destination is None else destination._set(r.data, r.size(), r.parms_id(), r.scale)

    def add_inplace(self, a, b):
        return self.add(a, b, a)

    def sub(self, a, b, destination=None):
        r = self._addsub(a, b, True)
        return r if destination is None else destination._set(r.data, r.size(), r.parms_id(), r.scale)

    def sub_inplace(self, a, b):
        return self.sub(a, b, a)

    def add_many(self, cts: Sequence[Ciphertext], destination=None):
        """SEAL: destination = cts[0]; add_inplace the rest.  Same bits from one n-way reduction kernel."""
        if not cts:
            raise ValueError("encrypteds cannot be empty")
        for c in cts[1:]:
            self._check_same(cts[0], c)
            if not self._close(cts[0].scale, c.scale):
                raise ValueError("scale mismatch")
            if c.size() != cts[0].size():
                raise ValueError("add_many: mixed sizes are not supported by the fused reduction")
        L, size = cts[0].parms_id(), cts[0].size()
        data = self.be.add_many(L, size, [c.data for c in cts])
        out = destination if destination is not None else Ciphertext()
        return out._set(data, size, L, cts[0].scale)

    def negate(self, a, destination=None):
        out = destination if destination is not None else Ciphertext()
        return out._set(self.be.negate(a.parms_id(), a.size(), a.data), a.size(), a.parms_id(), a.scale)

    def negate_inplace(self, a):
        return self.negate(a, a)

    def add_plain(self, a: Ciphertext, p: Plaintext, destination=None):
        if a.parms_id() != p.parms_id():
            raise ValueError("encrypted and plain parameter mismatch")
        if not self._close(a.scale, p.scale):
            raise ValueError("scale mismatch")
        out = destination if destination is not None else Ciphertext()
        return out._set(self.be.add_plain(a.parms_id(), a.size(), a.data, p.data), a.size(), a.parms_id(), a.scale)

    def add_plain_inplace(self, a, p):
        return self.add_plain(a, p, a)

    # ---- multiply
    def multiply_plain(self, a: Ciphertext, p: Plaintext, destination=None):
        if a.parms_id() != p.parms_id():
            raise ValueError("encrypted_ntt and plain_ntt parameter mismatch")
        new_scale = a.scale * p.scale
        self._check_scale(new_scale, a.parms_id())
        data = self.be.multiply_plain(a.parms_id(), a.size(), a.data, p.data)
        if p.is_zero:
            raise RuntimeError("result ciphertext is transparent")  # SEAL: std::logic_error
        out = destination if destination is not None else Ciphertext()
        return out._set(data, a.size(), a.parms_id(), new_scale)

    def multiply_plain_inplace(self, a, p):
        return self.multiply_plain(a, p, a)

    # ---- the same operation over many independent ciphertexts (the rows of a data set): one engine launch per
    # list where the backend offers it, otherwise the per-item calls -- identical checks and bits either way
    def _uniform(self, cts):
        if not cts:  # an empty list is trivially uniform; the callers below return before touching cts[0]
            return True
        L, size, scale = cts[0].parms_id(), cts[0].size(), cts[0].scale
        return all(c._parms_id == L and c._size == size and c._scale == scale for c in cts)

    def multiply_many(self, As: Sequence[Ciphertext], Bs: Sequence[Ciphertext]) -> List[Ciphertext]:
        if not As:  # a rank that owns no unit of a sharded product (parallel.py)
            return []
        f = getattr(self.be, "multiply_batch", None)
        if f is None or not (self._uniform(As) and self._uniform(Bs)) or any(a.data is b.data for a, b in zip(As, Bs)):
            return [self.multiply(a, b) for a, b in zip(As, Bs)]
        a, b = As[0], Bs[0]
        self._check_same(a, b)
        if a.size() != 2 or b.size() != 2:
            raise ValueError("multiply: only size-2 operands are supported (all reference call sites)")
        s = a.scale * b.scale
        self._check_scale(s, a.parms_id())
        outs = f(a.parms_id(), [x.data for x in As], [x.data for x in Bs])
        return [Ciphertext()._set(o, 3, a.parms_id(), s) for o in outs]

    def relinearize_many_inplace(self, cts: Sequence[Ciphertext], relin_keys: KSwitchKeys):
        f = getattr(self.be, "relinearize_batch", None)
        if not cts:
            return cts
        if f is None or not self._uniform(cts) or cts[0].size() != 3:
            for c in cts:
                self.relinearize_inplace(c, relin_keys)
            return cts
        L = cts[0].parms_id()
        for c, o in zip(cts, f(L, [c.data for c in cts], relin_keys.key(0))):
            c._set(o, 2, L, c.scale)
        return cts

    def rescale_to_next_many_inplace(self, cts: Sequence[Ciphertext]):
        f = getattr(self.be, "rescale_batch", None)
        if not cts:
            return cts
        if f is None or not self._uniform(cts) or cts[0].parms_id() < 2:
            for c in cts:
                self.rescale_to_next_inplace(c)
            return cts
        L, size = cts[0].parms_id(), cts[0].size()
        new_scale = cts[0].scale / float(self.ctx.primes[L - 1])
        for c, o in zip(cts, f(L, size, [c.data for c in cts])):
            c._set(o, size, L - 1, new_scale)
        return cts

    def add_pairs(self, As: Sequence[Ciphertext], Bs: Sequence[Ciphertext]) -> List[Ciphertext]:
        f = getattr(self.be, "add_batch", None)
        if not As:
            return []
        if f is None or not (self._uniform(As) and self._uniform(Bs)) or As[0].size() != Bs[0].size():
            return [self.add(a, b) for a, b in zip(As, Bs)]
        a, b = As[0], Bs[0]
        self._check_same(a, b)
        if not self._close(a.scale, b.scale):
            raise ValueError("scale mismatch")
        outs = f(a.parms_id(), a.size(), [x.data for x in As], [x.data for x in Bs])
        return [Ciphertext()._set(o, a.size(), a.parms_id(), a.scale) for o in outs]

    def multiply_plain_sum(self, cts: Sequence[Ciphertext], pts: Sequence[Plaintext], group: Optional[int] = None):
        """add_many(multiply_plain(cts[i], pts[i])) per group of `group` consecutive terms (None: one group), in one
        pass over the operands (hefx_multiply_plain_sum).  Checks and exceptions of the op-by-op sequence
        (helper.h:271,275), same bits.  Returns the list of group sums."""
        if not cts or len(cts) != len(pts):
            raise ValueError("multiply_plain_sum: need as many plaintexts as ciphertexts")
        L, size = cts[0].parms_id(), cts[0].size()
        scale = None
        for a, p in zip(cts, pts):
            if a.parms_id() != L or a.size() != size:
                raise ValueError("encrypted parameter mismatch")
            if a.parms_id() != p.parms_id():
                raise ValueError("encrypted_ntt and plain_ntt parameter mismatch")
            s = a.scale * p.scale
            self._check_scale(s, L)
            if scale is not None and not self._close(scale, s):
                raise ValueError("scale mismatch")
            scale = s if scale is None else scale
            if p.is_zero:
                raise RuntimeError("result ciphertext is transparent")
        outs = self.be.multiply_plain_sum(L, size, [a.data for a in cts], [p.data for p in pts], group)
        return [Ciphertext()._set(o, size, L, scale) for o in outs]

    def multiply(self, a: Ciphertext, b: Ciphertext, destination=None):
        self._check_same(a, b)
        if a.size() != 2 or b.size() != 2:
            raise ValueError("multiply: only size-2 operands are supported (all reference call sites)")
        new_scale = a.scale * b.scale
        self._check_scale(new_scale, a.parms_id())
        L = a.parms_id()
        data = self.be.square(L, a.data) if a.data is b.data else self.be.multiply(L, a.data, b.data)
        out = destination if destination is not None else Ciphertext()
        return out._set(data, 3, L, new_scale)

    def multiply_inplace(self, a, b):
        return self.multiply(a, b, a)

    def square(self, a, destination=None):
        return self.multiply(a, a, destination)

    def square_inplace(self, a):
        return self.multiply(a, a, a)

    # ---- relinearize / rescale / mod switch
    def relinearize_inplace(self, a: Ciphertext, relin_keys: KSwitchKeys):
        if a.size() == 2:
            return a  # SEAL: nothing to do (true at logistic_regression_ckks.cpp:237,319)
        if a.size() != 3:
            raise ValueError("relinearize: encrypted size must be 2 or 3")
        return a._set(self.be.relinearize(a.parms_id(), a.data, relin_keys.key(0)), 2, a.parms_id(), a.scale)

    def rescale_to_next_inplace(self, a: Ciphertext):
        L = a.parms_id()
        if L <= 1:
            raise ValueError("end of modulus switching chain reached")
        q_last = self.ctx.primes[L - 1]
        return a._set(self.be.rescale(L, a.size(), a.data), a.size(), L - 1, a.scale / float(q_last))

    def mod_switch_to_next_inplace(self, x):
        return self.mod_switch_to_inplace(x, x.parms_id() - 1)

    def mod_switch_to_inplace(self, x, parms_id: int):
        L = x.parms_id()
        if parms_id > L:
            raise ValueError("cannot switch to higher level modulus")
        if parms_id < 1:
            raise ValueError("end of modulus switching chain reached")
        if parms_id == L:
            return x
        if isinstance(x, Plaintext):
            x.data = self.be.mod_drop(L, parms_id, 1, x.data)
            x._parms_id = parms_id
            return x
        return x._set(self.be.mod_drop(L, parms_id, x.size(), x.data), x.size(), parms_id, x.scale)

    # ---- rotations
    def rotation_plan(self, steps: int, galois_keys: KSwitchKeys) -> List[int]:
        """Galois elements SEAL's rotate_internal applies, in order (App. A.7)."""
        if steps == 0:
            return []
        N = self.ctx.N
        elt = galois_elt_from_step(steps, N)
        if galois_keys.has_key(elt):
            return [elt]
        terms = naf(steps)
        if len(terms) == 1:
            raise ValueError("Galois key not present")
        plan: List[int] = []
        for t in terms:
            if abs(t) == N // 2:
                continue
            plan += self.rotation_plan(t, galois_keys)
        return plan

    def rotate_vector(self, a: Ciphertext, steps: int, galois_keys: KSwitchKeys, destination=None):
        if a.size() != 2:
            raise ValueError("encrypted size must be 2")
        data, L = a.data, a.parms_id()
        for elt in self.rotation_plan(steps, galois_keys):
            data = self.be.apply_galois(L, data, elt, galois_keys.key(elt))
        out = destination if destination is not None else Ciphertext()
        return out._set(data, 2, L, a.scale)

    def rotate_vector_inplace(self, a, steps, galois_keys):
        return self.rotate_vector(a, steps, galois_keys, a)

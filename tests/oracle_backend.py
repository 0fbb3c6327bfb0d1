"""Oracle-backed twin of seal.GpuBackend (TEST INFRASTRUCTURE): same methods, numpy arrays as handles, every
operation computed by oracle/ckks_oracle.c.  Running seal.Evaluator / algorithms.* on this backend yields the
reference-semantics result that the HIP engine must reproduce bit for bit."""
import numpy as np

from oracle import oracle as O


class OracleBackend:
    name = "oracle"

    def __init__(self, N, primes):
        self.o = O.Oracle(N, primes)
        self.N, self.primes, self.k = N, list(primes), len(primes)

    def from_host(self, a):
        return np.ascontiguousarray(a, dtype=np.uint64).copy()

    def to_host(self, h):
        return h

    def _ct(self, h, size, L):
        return np.ascontiguousarray(h).reshape(size, L, self.N)

    def ntt_forward(self, h, npoly, nrows, mod_first=0):
        v = h.reshape(npoly, nrows, self.N)
        for p in range(npoly):
            for r in range(nrows):
                v[p, r] = self.o.ntt_fwd(mod_first + r, v[p, r])
        return h

    def ntt_inverse(self, h, npoly, nrows, mod_first=0):
        v = h.reshape(npoly, nrows, self.N)
        for p in range(npoly):
            for r in range(nrows):
                v[p, r] = self.o.ntt_inv(mod_first + r, v[p, r])
        return h

    def add(self, L, size, a, b):
        return self.o.add(self._ct(a, size, L), self._ct(b, size, L))

    def sub(self, L, size, a, b):
        return self.o.sub(self._ct(a, size, L), self._ct(b, size, L))

    def negate(self, L, size, a):
        return self.o.negate(self._ct(a, size, L))

    def add_plain(self, L, size, ct, pt):
        return self.o.add_plain(self._ct(ct, size, L), np.ascontiguousarray(pt).reshape(L, self.N))

    def add_many(self, L, size, cts):
        acc = self._ct(cts[0], size, L).copy()
        for c in cts[1:]:
            acc = self.o.add(acc, self._ct(c, size, L))
        return acc

    def multiply_plain(self, L, size, ct, pt):
        return self.o.multiply_plain(self._ct(ct, size, L), np.ascontiguousarray(pt).reshape(L, self.N))

    def multiply_plain_sum(self, L, size, cts, pts, group=None):
        # the op-by-op statement: multiply_plain each (helper.h:271), add_many per group (:275)
        n = len(cts)
        group = n if group is None else group
        return [self.add_many(L, size, [self.multiply_plain(L, size, cts[i], pts[i])
                                        for i in range(g, min(n, g + group))]) for g in range(0, n, group)]

    def multiply(self, L, a, b):
        return self.o.multiply(self._ct(a, 2, L), self._ct(b, 2, L))

    def square(self, L, a):
        return self.o.multiply(self._ct(a, 2, L), self._ct(a, 2, L))

    def apply_galois(self, L, ct, elt, key):
        return self.o.apply_galois(self._ct(ct, 2, L), elt, key)

    def apply_galois_batch(self, L, cts, elts, keys):
        return [self.apply_galois(L, c, e, k) for c, e, k in zip(cts, elts, keys)]

    def rotate_multiply_plain_batch(self, L, cts, elts, keys, pts):
        return [self.o.rotate_mulplain(self._ct(c, 2, L), e, k, np.ascontiguousarray(p).reshape(L, self.N))
                for c, e, k, p in zip(cts, elts, keys, pts)]

    # ---- sampling / encrypt / decrypt: the same composition hefx_encrypt / hefx_decrypt run on the GPU
    def sample(self, kind, key32, stream_id, npoly, nrows, mod_first=0):
        return self.o.sample(kind, key32, stream_id, npoly, nrows, mod_first)

    def encrypt(self, L, pk, plain, key32, stream_id):
        u = self.o.sample("ternary", key32, 4 * stream_id + 0, 1, L)
        e = np.concatenate([self.o.sample("noise", key32, 4 * stream_id + 1, 1, L),
                            self.o.sample("noise", key32, 4 * stream_id + 2, 1, L)])
        self.ntt_forward(u, 1, L)
        self.ntt_forward(e, 2, L)
        pkL = np.ascontiguousarray(np.asarray(pk).reshape(2, self.k, self.N)[:, :L, :])
        c = self.o.add(self.o.multiply_plain(pkL, u[0]), e)
        if plain is not None:
            c = self.o.add_plain(c, np.ascontiguousarray(plain).reshape(L, self.N))
        return c

    def decrypt(self, L, size, ct, sk):
        s = np.ascontiguousarray(np.asarray(sk).reshape(-1, self.N)[:L])
        c = self._ct(ct, size, L)
        acc = c[size - 1][None].copy()
        for p in range(size - 2, -1, -1):
            acc = self.o.add(self.o.multiply_plain(acc, s), c[p][None])
        return acc[0]

    def rotate_hoisted_batch(self, L, ct, elts, keys, pts=None):
        # hoisting is exact since round 4 (ks_mac_exact_kernel): its words are the regular key switch's
        c = self._ct(ct, 2, L)
        outs = [self.o.apply_galois(c, e, k) for e, k in zip(elts, keys)]
        if pts is not None:
            outs = [self.o.multiply_plain(r, np.ascontiguousarray(p).reshape(L, self.N)) for r, p in zip(outs, pts)]
        return outs

    def lt_double_hoisted_core(self, ct_new, diags_keylevel, elts, keys):
        L = len(self.primes) - 1
        return self.o.lt_double_hoisted_core(self._ct(ct_new, 2, L),
                                             [np.ascontiguousarray(p).reshape(self.k, self.N) for p in diags_keylevel],
                                             elts, keys)

    def relinearize(self, L, ct3, key):
        return self.o.relinearize(self._ct(ct3, 3, L), key)

    rescale_rounded = True  # round to nearest, like the engine's default; tests switch the twin to floor

    def rescale(self, L, size, ct):
        return self.o.rescale(self._ct(ct, size, L), rounded=self.rescale_rounded)

    def mod_drop(self, L_in, L_out, npoly, x):
        return np.ascontiguousarray(np.ascontiguousarray(x).reshape(npoly, L_in, self.N)[:, :L_out, :])

    def reduce_canonical(self, L, size, h, addends):
        v = self._ct(h, size, L)
        for j in range(L):
            v[:, j, :] %= np.uint64(self.primes[j])
        return v

"""Thin object layer over the C-ABI: device buffers + one method per hefx_* entry point.

Payloads are flat uint64 arrays in SEAL's layout ([size][L][N], NTT form, canonical residues).
"""
from __future__ import annotations

import ctypes as C
from typing import List,  Optional, Sequence

import numpy as np

from . import capi


class DeviceArray:
    """A uint64 device buffer owned by an Engine (hefx_malloc/hefx_free)."""

    __slots__ = ("engine", "ptr", "shape", "_owned", "_parent")

    def __init__(self, engine: "Engine", shape, ptr: Optional[int] = None):
        self.engine = engine
        self.shape = tuple(int(s) for s in shape)
        if ptr is None:
            p = C.c_void_p()
            capi.check(capi.lib().hefx_malloc(engine._h, self.nbytes, C.byref(p)))
            self.ptr = p.value
            self._owned = True
        else:
            self.ptr = int(ptr)
            self._owned = False

    @property
    def nwords(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def nbytes(self) -> int:
        return self.nwords * 8

    def upload(self, host: np.ndarray, stream=None) -> "DeviceArray":
        h = np.ascontiguousarray(host, dtype=np.uint64)
        if h.size != self.nwords:
            raise ValueError(f"upload size mismatch: {h.size} vs {self.nwords}")
        capi.check(capi.lib().hefx_upload(self.engine._h, self.ptr, h.ctypes.data, self.nbytes, stream))
        return self

    def download(self, stream=None) -> np.ndarray:
        out = np.empty(self.shape, dtype=np.uint64)
        capi.check(capi.lib().hefx_download(self.engine._h, out.ctypes.data, self.ptr, self.nbytes, stream))
        return out

    def view(self, offset_words: int, shape) -> "DeviceArray":
        """Non-owning window into this buffer; holds a reference to the parent so the memory outlives it."""
        v = DeviceArray(self.engine, shape, ptr=self.ptr + 8 * int(offset_words))
        v._parent = self
        return v

    def free(self):
        if self._owned and self.ptr and self.engine._h:
            capi.lib().hefx_free(self.engine._h, self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """One CKKS parameter set on one GPU: N, primes[0..k-1] (last = special prime)."""

    def __init__(self, poly_degree: int, primes: Sequence[int], device: int = 0):
        self.N = int(poly_degree)
        self.primes = [int(p) for p in primes]
        self.k = len(self.primes)
        self.device = device
        self._h = None
        arr = (C.c_uint64 * self.k)(*self.primes)
        h = C.c_void_p()
        capi.check(capi.lib().hefx_context_create(self.N, arr, self.k, device, C.byref(h)))
        self._h = h.value

    def close(self):
        if self._h:
            capi.lib().hefx_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- memory
    def empty_many(self, n: int, shape) -> List[DeviceArray]:
        """n equally shaped buffers as views of ONE allocation (hipMalloc / hipFree are slow and hipFree synchronises
        the device; a batch of outputs is one slab whose views keep it alive)"""
        words = 1
        for d in shape:
            words *= int(d)
        slab = DeviceArray(self, (max(n, 1) * words,))
        return [slab.view(i * words, shape) for i in range(n)]

    def empty(self, *shape) -> DeviceArray:
        return DeviceArray(self, shape)

    def zeros(self, *shape, stream=None) -> DeviceArray:
        a = DeviceArray(self, shape)
        capi.check(capi.lib().hefx_memset_zero(self._h, a.ptr, a.nbytes, stream))
        return a

    def to_device(self, host: np.ndarray, stream=None) -> DeviceArray:
        h = np.ascontiguousarray(host, dtype=np.uint64)
        return DeviceArray(self, h.shape).upload(h, stream)

    def copy(self, src: DeviceArray, stream=None) -> DeviceArray:
        dst = DeviceArray(self, src.shape)
        capi.check(capi.lib().hefx_copy(self._h, dst.ptr, src.ptr, src.nbytes, stream))
        return dst

    def copy_raw(self, dst_ptr: int, src_ptr: int, nbytes: int, stream=None):
        """device-to-device copy between raw pointers (e.g. a torch tensor's data_ptr)"""
        capi.check(capi.lib().hefx_copy(self._h, int(dst_ptr), int(src_ptr), int(nbytes), stream))

    def sync(self, stream=None):
        capi.check(capi.lib().hefx_stream_sync(self._h, stream))

    def ks_fallback_count(self) -> int:
        """key-switch chunks redone item by item because a shared source held a zero c1 coefficient (hefx.h:
        hefx_ks_fallback_count); waits for the device"""
        import ctypes
        v = ctypes.c_uint64(0)
        capi.check(capi.lib().hefx_ks_fallback_count(self._h, ctypes.byref(v)))
        return int(v.value)

    def ks_stats(self) -> dict:
        """host-side counters of the key-switch front door (hefx_ks_stats): key switches submitted, of them exactly
        hoisted, launch sequences, batched calls -- since the context was created"""
        import ctypes
        v = (ctypes.c_uint64 * 4)()
        capi.check(capi.lib().hefx_ks_stats(self._h, v))
        return {"key_switches": int(v[0]), "hoisted": int(v[1]), "chunks": int(v[2]), "calls": int(v[3])}

    def psi(self, j: int) -> int:
        return int(capi.lib().hefx_psi(self._h, j))

    # ---- NTT (in place)
    def ntt_forward(self, buf: DeviceArray, npoly: int, nrows: int, mod_first: int = 0, stream=None):
        capi.check(capi.lib().hefx_ntt_forward(self._h, buf.ptr, npoly, nrows, mod_first, stream))

    def ntt_inverse(self, buf: DeviceArray, npoly: int, nrows: int, mod_first: int = 0, stream=None):
        capi.check(capi.lib().hefx_ntt_inverse(self._h, buf.ptr, npoly, nrows, mod_first, stream))

    # ---- element-wise; ciphertext arrays are [count?][size][L][N]
    def _out(self, like: DeviceArray, out):
        return out if out is not None else DeviceArray(self, like.shape)

    def add(self, L, size, a, b, out=None, count=1, stream=None):
        out = self._out(a, out)
        capi.check(capi.lib().hefx_add(self._h, L, size, count, a.ptr, b.ptr, out.ptr, stream))
        return out

    def sub(self, L, size, a, b, out=None, count=1, stream=None):
        out = self._out(a, out)
        capi.check(capi.lib().hefx_sub(self._h, L, size, count, a.ptr, b.ptr, out.ptr, stream))
        return out

    def negate(self, L, size, a, out=None, count=1, stream=None):
        out = self._out(a, out)
        capi.check(capi.lib().hefx_negate(self._h, L, size, count, a.ptr, out.ptr, stream))
        return out

    def add_plain(self, L, size, ct, pt, out=None, stream=None):
        out = self._out(ct, out)
        capi.check(capi.lib().hefx_add_plain(self._h, L, size, ct.ptr, pt.ptr, out.ptr, stream))
        return out

    def add_many(self, L, size, cts: Sequence[DeviceArray], out=None, stream=None):
        out = self._out(cts[0], out)
        arr = capi.ptr_array([c.ptr for c in cts])
        capi.check(capi.lib().hefx_add_many(self._h, L, size, len(cts), arr, out.ptr, stream))
        return out

    def multiply_plain(self, L, size, ct, pt, out=None, count=1, stream=None):
        out = self._out(ct, out)
        capi.check(capi.lib().hefx_multiply_plain(self._h, L, size, count, ct.ptr, pt.ptr, out.ptr, stream))
        return out

    def multiply_plain_sum(self, L, size, cts, pts, group=None, outs=None, stream=None):
        """outs[g] = sum over group g of cts[i] (.) pts[i] (hefx_multiply_plain_sum); group=None: one sum of all n"""
        n = len(cts)
        group = n if group is None else int(group)
        groups = (n + group - 1) // group
        if outs is None:
            outs = self.empty_many(groups, (size, L, self.N))
        capi.check(capi.lib().hefx_multiply_plain_sum(self._h, L, size, n, group, capi.ptr_array([c.ptr for c in cts]),
                                                      capi.ptr_array([p.ptr for p in pts]),
                                                      capi.ptr_array([o.ptr for o in outs]), stream))
        return outs

    def check_transparent(self, stream=None):
        capi.check(capi.lib().hefx_check_transparent(self._h, stream))

    def multiply(self, L, a, b, out=None, stream=None):
        out = out if out is not None else DeviceArray(self, (3, L, self.N))
        capi.check(capi.lib().hefx_multiply(self._h, L, a.ptr, b.ptr, out.ptr, stream))
        return out

    def multiply_batch(self, L, As, Bs, outs=None, stream=None):
        """outs[i] = As[i] * Bs[i] (size 2 x 2 -> 3), one launch (hefx_multiply_batch)"""
        n = len(As)
        outs = outs if outs is not None else self.empty_many(n, (3, L, self.N))
        capi.check(capi.lib().hefx_multiply_batch(self._h, L, n, capi.ptr_array([a.ptr for a in As]),
                                                  capi.ptr_array([b.ptr for b in Bs]),
                                                  capi.ptr_array([o.ptr for o in outs]), stream))
        return outs

    @staticmethod
    def contiguous(arrs) -> bool:
        """equally sized buffers laid out back to back (views of one slab, in order)"""
        base, step = arrs[0].ptr, arrs[0].nbytes
        return all(a.nbytes == step and a.ptr == base + i * step for i, a in enumerate(arrs))

    def add_batch(self, L, size, As, Bs, stream=None):
        """[As[i] + Bs[i]]: one launch when both lists are slabs (count = n), else one call per pair"""
        n = len(As)
        outs = self.empty_many(n, (size, L, self.N))
        if n > 1 and self.contiguous(As) and self.contiguous(Bs):
            capi.check(capi.lib().hefx_add(self._h, L, size, n, As[0].ptr, Bs[0].ptr, outs[0].ptr, stream))
        else:  # scattered operands: one launch over a pointer table
            capi.check(capi.lib().hefx_add_batch(self._h, L, size, n, capi.ptr_array([a.ptr for a in As]),
                                                 capi.ptr_array([b.ptr for b in Bs]),
                                                 capi.ptr_array([o.ptr for o in outs]), stream))
        return outs

    def sub_batch(self, L, size, As, Bs, outs=None, stream=None):
        """[As[i] - Bs[i]] in one launch over a pointer table (hefx_sub_batch)"""
        n = len(As)
        outs = outs if outs is not None else self.empty_many(n, (size, L, self.N))
        capi.check(capi.lib().hefx_sub_batch(self._h, L, size, n, capi.ptr_array([a.ptr for a in As]),
                                             capi.ptr_array([b.ptr for b in Bs]), capi.ptr_array([o.ptr for o in outs]),
                                             stream))
        return outs

    def multiply_plain_batch(self, L, size, cts, pts, outs=None, stream=None):
        """[cts[i] (.) pts[i]] in one launch over a pointer table (hefx_multiply_plain_batch)"""
        n = len(cts)
        outs = outs if outs is not None else self.empty_many(n, (size, L, self.N))
        capi.check(capi.lib().hefx_multiply_plain_batch(self._h, L, size, n, capi.ptr_array([c.ptr for c in cts]),
                                                        capi.ptr_array([p.ptr for p in pts]),
                                                        capi.ptr_array([o.ptr for o in outs]), stream))
        return outs

    def rescale_batch(self, L, size, cts, stream=None):
        """[rescale_to_next(ct)]: one launch when the inputs are a slab"""
        n = len(cts)
        outs = self.empty_many(n, (size, L - 1, self.N))
        if n > 1 and self.contiguous(cts):
            capi.check(capi.lib().hefx_rescale_to_next(self._h, L, size, n, cts[0].ptr, outs[0].ptr, stream))
        else:  # scattered operands: one launch pair over a pointer table
            capi.check(capi.lib().hefx_rescale_to_next_batch(self._h, L, size, n, capi.ptr_array([c.ptr for c in cts]),
                                                             capi.ptr_array([o.ptr for o in outs]), stream))
        return outs

    def square(self, L, a, out=None, stream=None):
        out = out if out is not None else DeviceArray(self, (3, L, self.N))
        capi.check(capi.lib().hefx_square(self._h, L, a.ptr, out.ptr, stream))
        return out

    # ---- key switching
    def apply_galois(self, L, ct, elt, key, out=None, stream=None):
        out = out if out is not None else DeviceArray(self, (2, L, self.N))
        capi.check(capi.lib().hefx_apply_galois(self._h, L, ct.ptr, int(elt), key.ptr, out.ptr, stream))
        return out

    def apply_galois_batch(self, L, cts, elts, keys, outs=None, stream=None):
        n = len(cts)
        outs = outs if outs is not None else self.empty_many(n, (2, L, self.N))
        capi.check(capi.lib().hefx_apply_galois_batch(
            self._h, L, n, capi.ptr_array([c.ptr for c in cts]), capi.u32_array(elts),
            capi.ptr_array([k.ptr for k in keys]), capi.ptr_array([o.ptr for o in outs]), stream))
        return outs

    def apply_galois_add_batch(self, L, cts, elts, keys, accs, outs=None, acc_outs=None, stream=None):
        """outs[i] = apply_galois(cts[i]); acc_outs[i] = accs[i] + outs[i] in the same key switch (helper.h:474-475);
        acc_outs=None: new buffers; pass acc_outs=accs for the in-place sum.  Returns (outs, acc_outs)."""
        n = len(cts)
        outs = outs if outs is not None else self.empty_many(n, (2, L, self.N))
        acc_outs = acc_outs if acc_outs is not None else self.empty_many(n, (2, L, self.N))
        capi.check(capi.lib().hefx_apply_galois_add_batch(
            self._h, L, n, capi.ptr_array([c.ptr for c in cts]), capi.u32_array(elts),
            capi.ptr_array([k.ptr for k in keys]), capi.ptr_array([a.ptr for a in accs]),
            capi.ptr_array([a.ptr for a in acc_outs]), capi.ptr_array([o.ptr for o in outs]), stream))
        return outs, acc_outs

    def rotate_add_chain(self, L, cts, elts, keys, accs, steps, outs=None, acc_outs=None, stream=None):
        """`steps` times (t = apply_galois(t); a += t) for every pair (cts[i], accs[i]) in lockstep -- the loop of
        helper.h:472-476 as one call (hefx_rotate_add_chain).  Returns (final rotations, final sums); inputs untouched."""
        n = len(cts)
        outs = outs if outs is not None else self.empty_many(n, (2, L, self.N))
        acc_outs = acc_outs if acc_outs is not None else self.empty_many(n, (2, L, self.N))
        capi.check(capi.lib().hefx_rotate_add_chain(
            self._h, L, n, capi.ptr_array([c.ptr for c in cts]), capi.u32_array(elts),
            capi.ptr_array([k.ptr for k in keys]), capi.ptr_array([a.ptr for a in accs]),
            capi.ptr_array([a.ptr for a in acc_outs]), capi.ptr_array([o.ptr for o in outs]), int(steps), stream))
        return outs, acc_outs

    def apply_galois_forest(self, L, parents, ext_ins, elts, keys, pts=None, outs=None, stream=None):
        """hefx_apply_galois_forest: node i rotates node parents[i]'s result (parents[i] < 0: ext_ins[i]) by elts[i] with keys[i];
        a non-None pts[i] is multiplied in.  Returns the nodes' results."""
        n = len(parents)
        outs = outs if outs is not None else self.empty_many(n, (2, L, self.N))
        par = (C.c_int32 * n)(*[int(p) for p in parents])
        capi.check(capi.lib().hefx_apply_galois_forest(
            self._h, L, n, par, capi.ptr_array([x.ptr if x is not None else 0 for x in ext_ins]), capi.u32_array(elts),
            capi.ptr_array([k.ptr for k in keys]),
            capi.ptr_array([p.ptr if p is not None else 0 for p in pts]) if pts is not None else None,
            capi.ptr_array([o.ptr for o in outs]), stream))
        return outs

    def rotate_multiply_plain_batch(self, L, cts, elts, keys, pts, outs=None, stream=None):
        """out_i = rotate(ct_i, elt_i) (.) pt_i; a None entry of `pts` makes item i a plain rotation (hefx.h)"""
        n = len(cts)
        outs = outs if outs is not None else self.empty_many(n, (2, L, self.N))
        capi.check(capi.lib().hefx_rotate_multiply_plain_batch(
            self._h, L, n, capi.ptr_array([c.ptr for c in cts]), capi.u32_array(elts),
            capi.ptr_array([k.ptr for k in keys]), capi.ptr_array([p.ptr if p is not None else 0 for p in pts]),
            capi.ptr_array([o.ptr for o in outs]), stream))
        return outs

    def prepare_rotate_multiply_plain_batch(self, L, cts, elts, keys, pts, outs):
        """The C argument arrays of rotate_multiply_plain_batch built ONCE, for callers that submit the same batch repeatedly
        (bench.py: 9216 items per step are ~5 ms of Python list building around ~10 ms of GPU work at N = 8192, which a C or
        C++ caller of the ABI never pays).  Returns an opaque tuple for rotate_multiply_plain_prepared; the buffers stay
        referenced by it."""
        n = len(cts)
        return (L, n, capi.ptr_array([c.ptr for c in cts]), capi.u32_array(elts), capi.ptr_array([k.ptr for k in keys]),
                capi.ptr_array([p.ptr for p in pts]), capi.ptr_array([o.ptr for o in outs]), (cts, keys, pts, outs))

    def rotate_multiply_plain_prepared(self, prep, stream=None):
        L, n, a_ct, a_elt, a_key, a_pt, a_out, _ = prep
        capi.check(capi.lib().hefx_rotate_multiply_plain_batch(self._h, L, n, a_ct, a_elt, a_key, a_pt, a_out, stream))

    def relinearize(self, L, ct3, key, out=None, stream=None):
        out = out if out is not None else DeviceArray(self, (2, L, self.N))
        capi.check(capi.lib().hefx_relinearize(self._h, L, ct3.ptr, key.ptr, out.ptr, stream))
        return out

    def relinearize_batch(self, L, ct3s, key, outs=None, stream=None):
        n = len(ct3s)
        outs = outs if outs is not None else self.empty_many(n, (2, L, self.N))
        capi.check(capi.lib().hefx_relinearize_batch(
            self._h, L, n, capi.ptr_array([c.ptr for c in ct3s]), key.ptr,
            capi.ptr_array([o.ptr for o in outs]), stream))
        return outs

    def rescale_to_next(self, L, size, ct, out=None, count=1, stream=None, rounded=None):
        """rounded=None: the context's mode (set_rescale_rounded / HEFX_RESCALE); True / False: hefx_rescale_to_next_mode"""
        if out is None:
            shape = (size, L - 1, self.N) if count == 1 else (count, size, L - 1, self.N)
            out = DeviceArray(self, shape)
        if rounded is None:
            capi.check(capi.lib().hefx_rescale_to_next(self._h, L, size, count, ct.ptr, out.ptr, stream))
        else:
            capi.check(capi.lib().hefx_rescale_to_next_mode(self._h, L, size, count, ct.ptr, out.ptr,
                                                            1 if rounded else 0, stream))
        return out

    def set_rescale_rounded(self, rounded: bool):
        """context default of rescale_to_next: False = floor (SEAL 3.4.x per SURVEY App. A.9), True = round (>= 3.5)"""
        capi.check(capi.lib().hefx_set_rescale_mode(self._h, 1 if rounded else 0))

    @property
    def rescale_rounded(self) -> bool:
        return capi.lib().hefx_get_rescale_mode(self._h) == 1

    def mod_drop(self, L_in, L_out, npoly, x, out=None, stream=None):
        out = out if out is not None else DeviceArray(self, (npoly, L_out, self.N))
        capi.check(capi.lib().hefx_mod_drop(self._h, L_in, L_out, npoly, x.ptr, out.ptr, stream))
        return out

    def reduce_canonical(self, L, size, buf, addends=8, stream=None):
        capi.check(capi.lib().hefx_reduce_canonical(self._h, L, size, buf.ptr, addends, stream))
        return buf

    # ---- multi-GPU exchange behind the C-ABI (RCCL, loaded at run time)
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        capi.check(capi.lib().hefx_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, world: int, rank: int, unique_id: bytes):
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        capi.check(capi.lib().hefx_comm_init(self._h, world, rank, unique_id))

    def comm_destroy(self):
        capi.check(capi.lib().hefx_comm_destroy(self._h))

    @property
    def comm_world(self) -> int:
        return int(capi.lib().hefx_comm_world(self._h))

    def allreduce_sum(self, L, size, buf, stream=None):
        """in place: sum over the ranks of the communicator, canonical (hefx_allreduce_sum)"""
        capi.check(capi.lib().hefx_allreduce_sum(self._h, L, size, buf.ptr, stream))
        return buf

    def rotate_hoisted_batch(self, L, ct, elts, keys, pts=None, outs=None, stream=None):
        """n rotations of ONE ciphertext sharing its digit decomposition -- exactly hoisted since round 4: the words of
        apply_galois_batch (hefx.h: hefx_rotate_hoisted_batch); the one source may be no item's output"""
        n = len(elts)
        outs = outs if outs is not None else self.empty_many(n, (2, L, self.N))
        capi.check(capi.lib().hefx_rotate_hoisted_batch(
            self._h, L, ct.ptr, n, capi.u32_array(elts), capi.ptr_array([k.ptr for k in keys]),
            capi.ptr_array([p.ptr for p in pts]) if pts is not None else None,
            capi.ptr_array([o.ptr for o in outs]), stream))
        return outs

    def linear_transform_plain(self, L, ct, diag_pts, key_elts, keys, out=None, stream=None, hoisted=False):
        """Linear_Transform_Plain in one native call (hefx_linear_transform_plain[_hoisted])"""
        out = out if out is not None else DeviceArray(self, (2, L, self.N))
        f = {False: capi.lib().hefx_linear_transform_plain, True: capi.lib().hefx_linear_transform_plain_hoisted,
             2: capi.lib().hefx_linear_transform_plain_hoisted2}[hoisted]
        # key_elts / keys may be prebuilt C arrays (algorithms caches them per key set: a Galois-key set does not change)
        karr = keys if isinstance(keys, C.Array) else capi.ptr_array([k.ptr for k in keys])
        darr = diag_pts if isinstance(diag_pts, C.Array) else capi.ptr_array([p.ptr for p in diag_pts])
        capi.check(f(
            self._h, L, ct.ptr, len(darr), darr, len(karr), capi.u32_array(key_elts), karr, out.ptr, stream))
        return out

    def linear_transform_plain_many(self, L, cts, diag_pts, key_elts, keys, outs=None, stream=None):
        """len(cts) transforms of one dimension and key set in lockstep (hefx_linear_transform_plain_many); diag_pts: the
        diagonals of transform 0, then of transform 1, ... (a flat list or a prebuilt C pointer array)"""
        count = len(cts)
        if count < 1:
            raise ValueError("linear_transform_plain_many: no inputs")
        outs = outs if outs is not None else self.empty_many(count, (2, L, self.N))
        karr = keys if isinstance(keys, C.Array) else capi.ptr_array([k.ptr for k in keys])
        darr = diag_pts if isinstance(diag_pts, C.Array) else capi.ptr_array([p.ptr for p in diag_pts])
        if len(darr) % count:
            raise ValueError("linear_transform_plain_many: every transform needs the same number of diagonals")
        capi.check(capi.lib().hefx_linear_transform_plain_many(
            self._h, L, count, capi.ptr_array([c.ptr for c in cts]), len(darr) // count, darr, len(karr),
            capi.u32_array(key_elts), karr, capi.ptr_array([o.ptr for o in outs]), stream))
        return outs

    def linear_transform_plain_hoisted2_sparse(self, L, ct, d, steps, diag_pts_keylevel, key_elts, keys, out=None,
                                               stream=None):
        """double-hoisted transform over a subset of the diagonals (steps[0] == 0)"""
        out = out if out is not None else DeviceArray(self, (2, L, self.N))
        arr = (C.c_int * len(steps))(*[int(x) for x in steps])
        capi.check(capi.lib().hefx_linear_transform_plain_hoisted2_sparse(
            self._h, L, ct.ptr, int(d), len(steps), arr, capi.ptr_array([p.ptr for p in diag_pts_keylevel]), len(keys),
            capi.u32_array(key_elts), capi.ptr_array([k.ptr for k in keys]), out.ptr, stream))
        return out

    def linear_transform_plain_bsgs(self, L, ct, shifted_diag_pts, n1, key_elts, keys, hoisted=True, out=None,
                                    stream=None):
        """baby-step / giant-step Linear_Transform_Plain in one native call (hefx_linear_transform_plain_bsgs)"""
        out = out if out is not None else DeviceArray(self, (2, L, self.N))
        capi.check(capi.lib().hefx_linear_transform_plain_bsgs(
            self._h, L, ct.ptr, len(shifted_diag_pts), int(n1), capi.ptr_array([p.ptr for p in shifted_diag_pts]),
            len(keys), capi.u32_array(key_elts), capi.ptr_array([k.ptr for k in keys]), 1 if hoisted else 0, out.ptr,
            stream))
        return out

    # ---- randomness, encrypt, decrypt on the GPU
    def sample(self, kind, key32: bytes, stream_id: int, npoly: int, nrows: int, mod_first: int = 0, out=None,
               stream=None):
        """kind in {'uniform', 'ternary', 'noise'} -> [npoly][nrows][N]; ternary / noise in coefficient form"""
        if len(key32) != 32:
            raise ValueError("key32 must be 32 bytes")
        f = {"uniform": capi.lib().hefx_sample_uniform, "ternary": capi.lib().hefx_sample_ternary,
             "noise": capi.lib().hefx_sample_noise}[kind]
        out = out if out is not None else DeviceArray(self, (npoly, nrows, self.N))
        capi.check(f(self._h, key32, stream_id, npoly, nrows, mod_first, out.ptr, stream))
        return out

    def keygen_kswitch(self, sk, new_sk, key32: bytes, stream_id: int, out=None, stream=None):
        """key-switching key [k-1][2][k][N] for new_sk under sk, sampled and assembled on the device"""
        if len(key32) != 32:
            raise ValueError("key32 must be 32 bytes")
        out = out if out is not None else DeviceArray(self, (self.k - 1, 2, self.k, self.N))
        capi.check(capi.lib().hefx_keygen_kswitch(self._h, sk.ptr, new_sk.ptr, key32, stream_id, out.ptr, stream))
        return out

    def galois_permute(self, elt: int, a, rows: int, out=None, stream=None):
        out = out if out is not None else DeviceArray(self, (rows, self.N))
        capi.check(capi.lib().hefx_galois_permute(self._h, elt, a.ptr, rows, out.ptr, stream))
        return out

    def encrypt(self, L, pk, plain, key32: bytes, stream_id: int, out=None, stream=None):
        if len(key32) != 32:
            raise ValueError("key32 must be 32 bytes")
        out = out if out is not None else DeviceArray(self, (2, L, self.N))
        capi.check(capi.lib().hefx_encrypt(self._h, L, pk.ptr, plain.ptr if plain is not None else None, key32,
                                           stream_id, out.ptr, stream))
        return out

    def encrypt_batch(self, L, pk, plains, key32: bytes, first_stream_id: int, outs=None, stream=None):
        """n encryptions, item i with stream id first_stream_id + i: the words of n encrypt() calls (hefx_encrypt_batch);
        plains[i] may be None (encryption of zero)"""
        if len(key32) != 32:
            raise ValueError("key32 must be 32 bytes")
        n = len(plains)
        outs = outs if outs is not None else self.empty_many(n, (2, L, self.N))
        capi.check(capi.lib().hefx_encrypt_batch(
            self._h, L, n, pk.ptr, capi.ptr_array([p.ptr if p is not None else None for p in plains]), key32,
            int(first_stream_id), capi.ptr_array([o.ptr for o in outs]), stream))
        return outs

    def decrypt(self, L, size, ct, sk, out=None, stream=None):
        out = out if out is not None else DeviceArray(self, (L, self.N))
        capi.check(capi.lib().hefx_decrypt(self._h, L, size, ct.ptr, sk.ptr, out.ptr, stream))
        return out

    # ---- CKKS encode on the GPU
    def ckks_encode(self, L, values, scale, out=None, stream=None):
        """values: [count][nvalues] (or [nvalues]) real or complex -> [count][L][N] NTT-form plaintexts"""
        v = np.atleast_2d(np.asarray(values))
        count, nvalues = v.shape
        re = np.ascontiguousarray(v.real, dtype=np.float64)
        im = np.ascontiguousarray(v.imag, dtype=np.float64) if np.iscomplexobj(v) else None
        out = out if out is not None else DeviceArray(self, (count, L, self.N))
        capi.check(capi.lib().hefx_ckks_encode(self._h, L, re.ctypes.data, im.ctypes.data if im is not None else None,
                                               nvalues, count, float(scale), out.ptr, stream))
        return out

    def ckks_encode_batch(self, L, values, scale, outs=None, stream=None):
        """values: [count][nvalues] -> `count` separately allocated [L][N] plaintexts (hefx_ckks_encode_batch)"""
        v = np.atleast_2d(np.asarray(values))
        count, nvalues = v.shape
        re = np.ascontiguousarray(v.real, dtype=np.float64)
        im = np.ascontiguousarray(v.imag, dtype=np.float64) if np.iscomplexobj(v) else None
        outs = outs if outs is not None else self.empty_many(count, (L, self.N))
        capi.check(capi.lib().hefx_ckks_encode_batch(
            self._h, L, re.ctypes.data, im.ctypes.data if im is not None else None, nvalues, count, float(scale),
            capi.ptr_array([o.ptr for o in outs]), stream))
        return outs

    def ckks_decode(self, L, pt, scale, count=1, complex_out=True, stream=None):
        """[count][L][N] NTT-form plaintexts -> [count][N/2] slot values (complex, or real if complex_out=False)"""
        re = np.empty((count, self.N // 2), dtype=np.float64)
        im = np.empty((count, self.N // 2), dtype=np.float64) if complex_out else None
        capi.check(capi.lib().hefx_ckks_decode(self._h, L, pt.ptr, count, float(scale), re.ctypes.data,
                                               im.ctypes.data if im is not None else None, stream))
        return re + 1j * im if complex_out else re

    # ---- measurement
    def event(self):
        ev = C.c_void_p()
        capi.check(capi.lib().hefx_event_create(self._h, C.byref(ev)))
        return ev

    def event_record(self, ev, stream=None):
        capi.check(capi.lib().hefx_event_record(self._h, ev, stream))

    def event_elapsed_ms(self, ev0, ev1) -> float:
        ms = C.c_float()
        capi.check(capi.lib().hefx_event_elapsed_ms(self._h, ev0, ev1, C.byref(ms)))
        return float(ms.value)

    def event_destroy(self, ev):
        capi.check(capi.lib().hefx_event_destroy(self._h, ev))

    def profile_begin(self):
        capi.check(capi.lib().hefx_profile_begin(self._h))

    def profile_end(self):
        """-> ({launch kind: summed ms}, number of chunks)"""
        ms = (C.c_double * 8)()
        n = C.c_uint64()
        capi.check(capi.lib().hefx_profile_end(self._h, ms, C.byref(n)))
        names = []
        for k in range(8):
            nm = capi.lib().hefx_profile_stage_name(k).decode()
            if not nm:
                break
            names.append(nm)
        return {nm: float(ms[k]) for k, nm in enumerate(names)}, int(n.value)

"""The reference's composite HE algorithms (its L3 layer), restated over the seal.py API mirror so that the same
code drives the HIP engine (product) and, in tests, the oracle-backed twin.  Each function follows the
reference line by line in WHAT it calls and in which order -- results must be bit-identical -- but is free in HOW
the calls are batched: independent rotations of one ciphertext go to the engine as one batch.

Citations are /root/reference/helper.h unless noted.
"""
from __future__ import annotations

from typing import List, Sequence

import operator

import numpy as np

from .seal import Ciphertext, CKKSEncoder, Evaluator, KSwitchKeys, Plaintext


# ---- plain data prep (L0) -------------------------------------------------------------------
def get_diagonal(position: int, U: np.ndarray) -> np.ndarray:
    """helper.h:175-195: U(0,l), U(1,l+1), ... wrapping around."""
    n = U.shape[0]
    return np.array([U[i, (i + position) % n] for i in range(n)], dtype=U.dtype)


def get_all_diagonals(U: np.ndarray) -> np.ndarray:
    """helper.h:198-209"""
    return np.stack([get_diagonal(i, U) for i in range(U.shape[0])])


# ---- linear transforms ----------------------------------------------------------------------
def _rotations_batched(ev: Evaluator, ct: Ciphertext, steps: Sequence[int], gal_keys: KSwitchKeys,
                       pts: Sequence[Plaintext] = None) -> List[Ciphertext]:
    """rotate_vector(ct, l) for every l in `steps`, executed depth by depth over SEAL's NAF plans so that each
    engine call is one batch of independent key switches.  Identical (source payload, Galois element) pairs are
    computed once -- the key switch is deterministic, so sharing cannot change a bit.  When `pts` is given, the
    last term of each plan is fused with multiply_plain (hefx_rotate_multiply_plain_batch)."""
    be, L = ev.be, ct.parms_id()
    plans = [ev.rotation_plan(s, gal_keys) for s in steps]
    cur = [ct.data for _ in steps]
    if hasattr(be, "apply_galois_forest") and sum(len(p) for p in plans) > 1:
        # the engine's own scheduler for the whole forest (depth batches, subtrees on lanes, hoisted depths): one call
        index, parents, elts, node_pts = {}, [], [], []
        leaf = [-1] * len(plans)
        for i, p in enumerate(plans):
            c = -1
            for depth, elt in enumerate(p):
                fuse = pts is not None and depth == len(p) - 1
                key = (c, elt, i if fuse else -1)
                j = index.get(key)
                if j is None:
                    j = index[key] = len(parents)
                    parents.append(c), elts.append(elt), node_pts.append(pts[i].data if fuse else None)
                c = j
            leaf[i] = c
        outs = be.apply_galois_forest(L, parents, [ct.data if q < 0 else None for q in parents], elts,
                                      [gal_keys.key(e) for e in elts], node_pts if pts is not None else None)
        cur = [outs[j] if j >= 0 else ct.data for j in leaf]
        plans_done = True
    else:
        plans_done = False
    for depth in range(0 if plans_done else max((len(p) for p in plans), default=0)):
        jobs, job_list, owner = {}, [], {}
        for i, p in enumerate(plans):
            if depth >= len(p):
                continue
            fuse = pts is not None and depth == len(p) - 1
            key = (id(cur[i]), p[depth], i if fuse else -1)
            if key not in jobs:
                jobs[key] = len(job_list)
                job_list.append((cur[i], p[depth], pts[i].data if fuse else None))
            owner[i] = jobs[key]
        outs = [None] * len(job_list)
        plain = [j for j, t in enumerate(job_list) if t[2] is None]
        fused = [j for j, t in enumerate(job_list) if t[2] is not None]
        if plain:
            o = be.apply_galois_batch(L, [job_list[j][0] for j in plain], [job_list[j][1] for j in plain],
                                      [gal_keys.key(job_list[j][1]) for j in plain])
            for j, x in zip(plain, o):
                outs[j] = x
        if fused:
            o = be.rotate_multiply_plain_batch(L, [job_list[j][0] for j in fused], [job_list[j][1] for j in fused],
                                               [gal_keys.key(job_list[j][1]) for j in fused],
                                               [job_list[j][2] for j in fused])
            for j, x in zip(fused, o):
                outs[j] = x
        for i, j in owner.items():
            cur[i] = outs[j]
    out = []
    for i in range(len(steps)):
        if not plans[i]:  # step 0: rotate_vector returns its input unchanged
            c = ct.copy()
            out.append(ev.multiply_plain(c, pts[i]) if pts is not None else c)
            continue
        scale = ct.scale
        if pts is not None:
            scale *= pts[i].scale
            ev._check_scale(scale, L)
            if pts[i].parms_id() != L:
                raise ValueError("encrypted_ntt and plain_ntt parameter mismatch")
            if pts[i].is_zero:
                raise RuntimeError("result ciphertext is transparent")
        out.append(Ciphertext()._set(cur[i], 2, L, scale))
    return out


_PT_STATE = operator.attrgetter("_parms_id", "_scale", "is_zero")
_PT_PTR = operator.attrgetter("data.ptr")


def _plain_product_scale(ev: Evaluator, ct: Ciphertext, pts: Sequence[Plaintext]) -> float:
    """The checks multiply_plain + add_many would make over ct (.) pts[i], and the common product scale.  One set
    comprehension when everything is in order (a 1000-diagonal transform is otherwise host-bound in this loop); the
    element-by-element walk only runs to raise the exception of the FIRST offending plaintext, like the op-by-op
    sequence."""
    L = ct.parms_id()
    if set(map(_PT_STATE, pts)) == {(L, pts[0]._scale, False)}:  # (one C-level pass: 25 us for 512 plaintexts)
        scale = ct.scale * pts[0]._scale
        ev._check_scale(scale, L)
        return scale
    scale = None
    for p in pts:
        if p.parms_id() != L:
            raise ValueError("encrypted_ntt and plain_ntt parameter mismatch")
        s = ct.scale * p.scale
        ev._check_scale(s, L)
        if scale is not None and not ev._close(scale, s):
            raise ValueError("scale mismatch")
        scale = s if scale is None else scale
        if p.is_zero:
            raise RuntimeError("result ciphertext is transparent")
    return scale


def linear_transform_plain(ev: Evaluator, ct: Ciphertext, U_diagonals: Sequence[Plaintext],
                           gal_keys: KSwitchKeys, hoisted: bool = False) -> Ciphertext:
    """Linear_Transform_Plain, helper.h:237-262 (= linear_transformation2.cpp:149-174).

    hoisted=2 is the double-hoisted fast mode: U_diagonals must be KEY-LEVEL plaintexts (encoder.encode(..., parms_id =
    context.key_parms_id())), ct at the top data level; the products are accumulated over the extended basis and modded
    down once (hefx_linear_transform_plain_hoisted2) -- per rotation only a gathered key MAC remains.

    hoisted=True asks for SURVEY 8f rank 3 explicitly: the d-1 rotations of ct_new share one digit decomposition
    (hefx_rotate_hoisted_batch); it needs a direct Galois key per step (keygen.galois_keys(steps)).  Since round 4 the
    hoisted form is EXACT -- the engine adds the term by which SEAL's positive digit lifts differ from the signed ones
    (DESIGN.md "Exact hoisting") -- so this is bit-identical to the reference's sequence with those keys, and the plain
    call (hoisted=False) takes the same path by itself whenever more than 32 rotations share a source."""
    d = len(U_diagonals)
    if hoisted == 2:
        return _linear_transform_plain_hoisted2(ev, ct, U_diagonals, gal_keys)
    if hoisted:
        return _linear_transform_plain_hoisted(ev, ct, U_diagonals, gal_keys)
    native = getattr(ev.be, "linear_transform_plain", None)
    if native is not None:  # the HIP engine runs the whole transform behind one C-ABI call, same bits
        return _linear_transform_plain_native(ev, native, ct, U_diagonals, gal_keys)
    ct_rot = ev.rotate_vector(ct, -d, gal_keys)                      # :244  fill with duplicate
    ct_new = ev.add(ct, ct_rot)                                      # :247
    res = [ev.multiply_plain(ct_new, U_diagonals[0])]                # :250
    res += _rotations_batched(ev, ct_new, list(range(1, d)), gal_keys, U_diagonals[1:])   # :252-257
    return ev.add_many(res)                                          # :259


def _linear_transform_plain_hoisted2(ev: Evaluator, ct: Ciphertext, U_diagonals: Sequence[Plaintext],
                                     gal_keys: KSwitchKeys) -> Ciphertext:
    ctx, be = ev.ctx, ev.be
    d, L = len(U_diagonals), ct.parms_id()
    if ct.size() != 2:
        raise ValueError("encrypted size must be 2")
    if L != ctx.first_parms_id():
        raise ValueError("double hoisting is built for the top data level")
    scale = None
    for p in U_diagonals:
        if p.parms_id() != ctx.k:
            raise ValueError("double hoisting needs key-level plaintexts (encode with parms_id = key level)")
        s = ct.scale * p.scale
        ev._check_scale(s, L)
        if scale is not None and not ev._close(scale, s):
            raise ValueError("scale mismatch")
        scale = s if scale is None else scale
        if p.is_zero:
            raise RuntimeError("result ciphertext is transparent")
    native = getattr(be, "linear_transform_plain", None)
    if native is not None:  # plans, key checks and everything else behind one C-ABI call (errors come back as ValueError)
        elts = sorted(gal_keys.keys)
        data = native(L, ct.data, [p.data for p in U_diagonals], elts, [gal_keys.key(e) for e in elts], hoisted=2)
    else:  # the oracle twin: regular -d rotation, then the oracle's statement of the double-hoisted core
        plans = [ev.rotation_plan(l, gal_keys) for l in range(1, d)]
        if any(len(p) != 1 for p in plans):
            raise ValueError("hoisted linear transform needs a direct Galois key for every step 1..d-1")
        ct_new = ev.add(ct, ev.rotate_vector(ct, -d, gal_keys))
        elts = [p[0] for p in plans]
        data = be.lt_double_hoisted_core(ct_new.data, [p.data for p in U_diagonals], elts,
                                         [gal_keys.key(e) for e in elts])
    return Ciphertext()._set(data, 2, L, scale)


def _linear_transform_plain_hoisted(ev: Evaluator, ct: Ciphertext, U_diagonals: Sequence[Plaintext],
                                    gal_keys: KSwitchKeys) -> Ciphertext:
    d, L = len(U_diagonals), ct.parms_id()
    native = getattr(ev.be, "linear_transform_plain", None)
    if native is not None:  # one C-ABI call (hefx_linear_transform_plain_hoisted), which also checks the keys
        return _linear_transform_plain_native(ev, native, ct, U_diagonals, gal_keys, hoisted=True)
    plans = [ev.rotation_plan(l, gal_keys) for l in range(1, d)]
    if any(len(p) != 1 for p in plans):
        raise ValueError("hoisted linear transform needs a direct Galois key for every step 1..d-1")
    ct_new = ev.add(ct, ev.rotate_vector(ct, -d, gal_keys))          # helper.h:244-247, regular rotation
    res = [ev.multiply_plain(ct_new, U_diagonals[0])]                # :250
    for p in U_diagonals[1:]:
        if p.parms_id() != L:
            raise ValueError("encrypted_ntt and plain_ntt parameter mismatch")
        ev._check_scale(ct_new.scale * p.scale, L)
        if p.is_zero:
            raise RuntimeError("result ciphertext is transparent")
    elts = [p[0] for p in plans]
    outs = ev.be.rotate_hoisted_batch(L, ct_new.data, elts, [gal_keys.key(e) for e in elts],
                                      [p.data for p in U_diagonals[1:]])
    res += [Ciphertext()._set(o, 2, L, ct_new.scale * p.scale) for o, p in zip(outs, U_diagonals[1:])]
    return ev.add_many(res)                                          # :259


def _linear_transform_plain_native(ev: Evaluator, native, ct: Ciphertext, U_diagonals: Sequence[Plaintext],
                                   gal_keys: KSwitchKeys, hoisted: bool = False) -> Ciphertext:
    """Same checks and bookkeeping as the op-by-op path (SEAL's exceptions), arithmetic in hefx_linear_transform_plain."""
    L = ct.parms_id()
    if ct.size() != 2:
        raise ValueError("encrypted size must be 2")
    scale = _plain_product_scale(ev, ct, U_diagonals)
    # missing keys / too large steps come back from the engine with SEAL's messages (ValueError)
    elts, keys = _native_key_args(gal_keys)
    diag = [p.data for p in U_diagonals]
    if diag and hasattr(diag[0], "ptr"):  # HIP engine: the pointer table in one C-level pass
        from . import capi
        diag = capi.ptr_array(list(map(_PT_PTR, U_diagonals)))
    data = native(L, ct.data, diag, elts, keys, **({"hoisted": True} if hoisted else {}))
    return Ciphertext()._set(data, 2, L, scale)


def _native_key_args(gal_keys: KSwitchKeys):
    """(elements, key payloads) of a key set in the form the native entry points take.  On the HIP engine the two C arrays
    are built once per key set and kept on it (512 keys: 0.1 ms of Python per call otherwise, with the GPU idle); the set
    is rebuilt when keys were added.  Other backends get plain lists."""
    elts = sorted(gal_keys.keys)
    keys = [gal_keys.key(e) for e in elts]
    if not keys or not hasattr(keys[0], "ptr"):
        return elts, keys
    # fingerprint: elements AND payload addresses -- a key replaced under an existing element, or one removed and another
    # added, must not leave the native transform on the old payloads (0.03 ms at 512 keys against 0.1 ms for the C arrays)
    mark = (tuple(elts), tuple(k.ptr for k in keys))
    cache = getattr(gal_keys, "_native_args", None)
    if cache is not None and cache[0] == mark:
        return cache[1], cache[2]
    from . import capi
    cache = (mark, capi.u32_array(elts), capi.ptr_array(list(mark[1])), keys)  # keys: kept alive
    gal_keys._native_args = cache
    return cache[1], cache[2]


def linear_transforms_plain_many(ev: Evaluator, cts: Sequence[Ciphertext], diag_sets: Sequence[Sequence[Plaintext]],
                                 gal_keys: KSwitchKeys) -> List[Ciphertext]:
    """[Linear_Transform_Plain(cts[t], diag_sets[t], gal_keys) for t] -- INDEPENDENT transforms of one dimension, e.g.
    the sigma transform of ctA and the tau transform of ctB (matrix_multiplication.cpp:22-25).  On the HIP engine they run
    in lockstep behind one call (hefx_linear_transform_plain_many: every dependent launch sequence carries the items of all
    inputs); per input the operations and their order are those of linear_transform_plain -- same bits, which is also what
    every other backend (and mixed shapes) gets from the loop."""
    native = getattr(ev.be, "linear_transform_plain_many", None)
    d = len(diag_sets[0]) if diag_sets else 0
    same = (len(cts) > 1 and len(cts) == len(diag_sets) and all(len(ds) == d for ds in diag_sets)
            and all(c.size() == 2 and c.parms_id() == cts[0].parms_id() for c in cts)
            and len({id(c.data) for c in cts}) == len(cts))
    if native is None or not same or len(cts) > 64:
        return [linear_transform_plain(ev, c, ds, gal_keys) for c, ds in zip(cts, diag_sets)]
    L = cts[0].parms_id()
    scales = [_plain_product_scale(ev, c, ds) for c, ds in zip(cts, diag_sets)]   # SEAL's checks, transform by transform
    elts, keys = _native_key_args(gal_keys)
    flat = [p.data for ds in diag_sets for p in ds]
    if flat and hasattr(flat[0], "ptr"):
        from . import capi
        flat = capi.ptr_array([p.ptr for p in flat])
    outs = native(L, [c.data for c in cts], flat, elts, keys)
    return [Ciphertext()._set(o, 2, L, s) for o, s in zip(outs, scales)]


# ---- baby-step / giant-step form of Linear_Transform_Plain -----------------------------------
# With l = j*n1 + i (i < n1 baby, j < n2 giant):
#   sum_l diag_l (.) rot_l(v)  =  sum_j rot_(j*n1)( sum_i rot_(-j*n1)(diag_l) (.) rot_i(v) )
# so a d x d transform needs n1-1 (hoisted) + n2-1 key switches instead of d-1, and n1+n2-1 Galois keys instead of d
# (d = 1000: 63 and 64 against 999 and 1000), at the price of diagonals that are rotated in the clear before they
# are encoded.  A different operation sequence than helper.h:237-262, so NOT its noise bits: same decryption, checked
# bit-for-bit against the same composition on the oracle-backed twin.
def bsgs_split(d: int, n1: int = None):
    """(n1, n2) with n1*n2 >= d; default n1 = ceil(sqrt(d))"""
    if n1 is None:
        n1 = max(1, int(np.ceil(np.sqrt(d))))
    return n1, (d + n1 - 1) // n1


def bsgs_steps(d: int, n1: int = None) -> List[int]:
    """rotation steps the transform needs direct Galois keys for (keygen.galois_keys(bsgs_steps(d)))"""
    n1, n2 = bsgs_split(d, n1)
    return [-d] + list(range(1, n1)) + [j * n1 for j in range(1, n2)]


def bsgs_diagonals(diagonals: np.ndarray, n1: int = None) -> List[np.ndarray]:
    """diagonals[l] (length d, helper.h:198-209) shifted right by (l // n1) * n1 slots -- what
    linear_transform_plain_bsgs expects, still in the clear; encode each with the ciphertext's scale and level."""
    d = len(diagonals)
    n1, _ = bsgs_split(d, n1)
    out = []
    for l in range(d):
        shift = (l // n1) * n1
        v = np.zeros(shift + d, dtype=np.asarray(diagonals[l]).dtype)
        v[shift:] = diagonals[l]
        out.append(v)
    return out


def linear_transform_plain_bsgs(ev: Evaluator, ct: Ciphertext, shifted_diagonals: Sequence[Plaintext],
                                gal_keys: KSwitchKeys, n1: int = None, hoisted: bool = True) -> Ciphertext:
    """Linear_Transform_Plain (helper.h:237-262) in baby-step / giant-step form; `shifted_diagonals` are the
    encodings of bsgs_diagonals(...), gal_keys must hold direct keys for bsgs_steps(d, n1).  hoisted=True shares the
    digit decomposition of ct_new over the baby rotations (hefx_rotate_hoisted_batch)."""
    be, L = ev.be, ct.parms_id()
    d = len(shifted_diagonals)
    n1, n2 = bsgs_split(d, n1)
    if d + n1 * n2 > ev.ctx.N // 2:  # the shifted diagonals and the rotated duplicate must not wrap around
        raise ValueError("baby-step/giant-step transform: dimension too large for the slot count")
    native = getattr(be, "linear_transform_plain_bsgs", None)
    if native is not None:  # the HIP engine: the whole composition behind one C-ABI call, same bits as the lines below
        if ct.size() != 2:
            raise ValueError("encrypted size must be 2")
        scale = _plain_product_scale(ev, ct, shifted_diagonals)
        elts = sorted(gal_keys.keys)
        data = native(L, ct.data, [p.data for p in shifted_diagonals], n1, elts, [gal_keys.key(e) for e in elts],
                      hoisted)
        return Ciphertext()._set(data, 2, L, scale)
    baby = list(range(1, n1))
    plans = [ev.rotation_plan(s, gal_keys) for s in baby + [j * n1 for j in range(1, n2)]]
    if any(len(p) != 1 for p in plans):
        raise ValueError("baby-step/giant-step transform needs a direct Galois key for every step of bsgs_steps(d)")
    ct_new = ev.add(ct, ev.rotate_vector(ct, -d, gal_keys))          # helper.h:244-247
    elts = [p[0] for p in plans]
    if hoisted and baby:
        data = be.rotate_hoisted_batch(L, ct_new.data, elts[:n1 - 1], [gal_keys.key(e) for e in elts[:n1 - 1]])
        rots = [ct_new] + [Ciphertext()._set(x, 2, L, ct_new.scale) for x in data]
    else:
        rots = [ct_new] + _rotations_batched(ev, ct_new, baby, gal_keys)
    # inner sums: group j = sum_i rots[i] (.) diag'[j*n1 + i], all groups in one pass
    inner = ev.multiply_plain_sum([rots[l % n1] for l in range(d)], list(shifted_diagonals), group=n1)
    if n2 > 1:
        outs = be.apply_galois_batch(L, [c.data for c in inner[1:]], elts[n1 - 1:],
                                     [gal_keys.key(e) for e in elts[n1 - 1:]])
        inner = inner[:1] + [Ciphertext()._set(x, 2, L, inner[0].scale) for x in outs]
    return ev.add_many(inner)


def linear_transform_cipher(ev: Evaluator, ct: Ciphertext, U_diagonals: Sequence[Ciphertext],
                            gal_keys: KSwitchKeys) -> Ciphertext:
    """Linear_Transform_Cipher, helper.h:212-234: ct x ct products are NOT relinearized (size-3 sum)."""
    d = len(U_diagonals)
    ct_new = ev.add(ct, ev.rotate_vector(ct, -d, gal_keys))          # :216-219
    rots = _rotations_batched(ev, ct_new, list(range(1, d)), gal_keys)
    res = [ev.multiply(ct_new, U_diagonals[0])]                      # :222
    res += [ev.multiply(r, U_diagonals[l + 1]) for l, r in enumerate(rots)]   # :227-228
    return ev.add_many(res)                                          # :231


def linear_transform_ciphermatrix_plainvector(ev: Evaluator, pt_rotations: Sequence[Plaintext],
                                              U_diagonals: Sequence[Ciphertext]) -> Ciphertext:
    """Linear_Transform_CipherMatrix_PlainVector, helper.h:265-278."""
    n = len(pt_rotations)                                            # :271 multiply_plain, :275 add_many -- one pass
    return ev.multiply_plain_sum(list(U_diagonals[:n]), list(pt_rotations))[0]


def c_matrix_encode(ev: Evaluator, matrix: Sequence[Ciphertext], gal_keys: KSwitchKeys) -> Ciphertext:
    """C_Matrix_Encode, helper.h:307-322: sum_i rotate(row_i, -i*n)."""
    n = len(matrix)
    rots = [matrix[0].copy()] + [ev.rotate_vector(matrix[i], -i * n, gal_keys) for i in range(1, n)]
    return ev.add_many(rots)


def c_matrix_decode(ev: Evaluator, encoder: CKKSEncoder, matrix: Ciphertext, dimension: int, scale: float,
                    gal_keys: KSwitchKeys) -> List[Ciphertext]:
    """C_Matrix_Decode, helper.h:325-360: mask row i, rotate it back."""
    out = []
    for i in range(dimension):
        mask = np.zeros(dimension * dimension)
        mask[i * dimension:(i + 1) * dimension] = 1
        row = ev.multiply_plain(matrix, encoder.encode(mask, scale, parms_id=matrix.parms_id()))
        if i:
            ev.rotate_vector_inplace(row, i * dimension, gal_keys)
        out.append(row)
    return out


def cipher_dot_product(ev: Evaluator, ctA: Ciphertext, ctB: Ciphertext, size: int, relin_keys: KSwitchKeys,
                       gal_keys: KSwitchKeys) -> Ciphertext:
    """cipher_dot_product, helper.h:416-502.  The rotate-by-1 chain is inherently sequential and is kept so:
    rotating by i directly would give different noise bits."""
    mult = ev.multiply(ctA, ctB)                                     # :432
    ev.relinearize_inplace(mult, relin_keys)                         # :440
    ev.rescale_to_next_inplace(mult)                                 # :441
    zero_filled = ev.rotate_vector(mult, -size, gal_keys)            # :455
    dup = ev.add(mult, zero_filled)                                  # :464
    for _ in range(1, size):                                         # :472-476
        ev.rotate_vector_inplace(dup, 1, gal_keys)
        ev.add_inplace(mult, dup)
    mult.scale = 2.0 ** int(np.log2(mult.scale))                     # :489 "manual rescale"
    return mult


def compute_all_powers(ev: Evaluator, ct: Ciphertext, degree: int, relin_keys: KSwitchKeys) -> List[Ciphertext]:
    """compute_all_powers, helper.h:505-547 (= polynomial.cpp:56-96): x^i = x^cand * x^(i-cand), minimal depth."""
    powers: List[Ciphertext] = [None] * (degree + 1)
    powers[1] = ct
    levels = [0] * (degree + 1)
    for i in range(2, degree + 1):
        minimum, cand = i, -1
        for j in range(1, i // 2 + 1):
            k = i - j
            newlevel = max(levels[j], levels[k]) + 1
            if newlevel < minimum:
                cand, minimum = j, newlevel
        levels[i] = minimum
        a, b = powers[cand].copy(), powers[i - cand].copy()
        target = min(a.parms_id(), b.parms_id())
        ev.mod_switch_to_inplace(a, target)                          # :537
        ev.mod_switch_to_inplace(b, target)
        p = ev.multiply(a, b)                                        # :539
        ev.relinearize_inplace(p, relin_keys)                        # :541
        ev.rescale_to_next_inplace(p)                                # :543
        powers[i] = p
    return powers


def _matmul_step3(ev: Evaluator, ctA0: Ciphertext, ctB0: Ciphertext, ctAk: List[Ciphertext],
                  ctBk: List[Ciphertext]) -> Ciphertext:
    """matrix_multiplication.cpp:69-129: rescale the 2(n-1) transforms, A0*B0 + sum_k A_k*B_k.  The rescales and the
    products of the n-1 pairs are independent, so each kind is one launch over its list; the chain of add_inplace
    (:128) is one n-way sum -- canonical residues of the same integers, hence the same bits."""
    ev.rescale_to_next_many_inplace(list(ctAk) + list(ctBk))         # :69-73 (one launch pair over both families)
    ctAB = ev.multiply(ctA0, ctB0)                                   # :104
    ev.mod_switch_to_next_inplace(ctAB)                              # :112
    for c in ctAk + ctBk:
        c.scale = 2.0 ** int(np.log2(c.scale))                       # :117-121 "manual rescale"
    if not ctAk:
        return ctAB
    return ev.add_many([ctAB] + ev.multiply_many(ctAk, ctBk))        # :123-129


def _linear_transforms_of_one_input(ev: Evaluator, ct: Ciphertext, diag_sets: Sequence[Sequence[Plaintext]],
                                    gal_keys: KSwitchKeys) -> List[Ciphertext]:
    """[Linear_Transform_Plain(ct, diags, gal_keys) for diags in diag_sets] (matrix_multiplication.cpp:40-43: the n-1
    transforms V_k of ctA0, W_k of ctB0).  Every one of them starts by forming the same ct_new = ct + rotate(ct, -d)
    and the same rotations rotate(ct_new, l) -- deterministic functions of ct -- so they are formed ONCE and each
    transform keeps only its sum of plaintext products (helper.h:250-259) -- the bits of the transform-by-transform
    loop with (n-1)x fewer key switches."""
    if not diag_sets:
        return []
    d = len(diag_sets[0])
    if any(len(ds) != d for ds in diag_sets):
        return [linear_transform_plain(ev, ct, ds, gal_keys) for ds in diag_sets]
    ct_new = _duplicate(ev, ct, d, gal_keys)                          # helper.h:244-247
    rots = [ct_new] + _rotations_batched(ev, ct_new, list(range(1, d)), gal_keys)   # :255
    terms_ct = [rots[l] for _ in diag_sets for l in range(d)]
    terms_pt = [ds[l] for ds in diag_sets for l in range(d)]
    return ev.multiply_plain_sum(terms_ct, terms_pt, group=d)        # :250, :256, :259


def _rotations_of_many(ev: Evaluator, cts: Sequence[Ciphertext], steps: Sequence[int], gal_keys: KSwitchKeys,
                       pts_per_input: Sequence[Sequence[Plaintext]] = None) -> List[List[Ciphertext]]:
    """[[rotate_vector(ct, l) for l in steps] for ct in cts] with the NAF forests of all inputs in ONE engine call
    (hefx_apply_galois_forest takes any number of roots): the depths of the forests run side by side, so k inputs cost the
    dependent launch sequences of one.  pts_per_input[t][i] (optional): multiply_plain of input t's rotation by steps[i],
    fused into the last key switch of its plan, with the checks of the op-by-op product (as _rotations_batched does for one
    input).  Backends without the forest entry, a single input or a zero step take the per-input path."""
    be = ev.be
    if len(cts) < 2 or not hasattr(be, "apply_galois_forest") or any(s == 0 for s in steps) or not steps or \
            len({c.parms_id() for c in cts}) != 1 or any(c.size() != 2 for c in cts):
        return [_rotations_batched(ev, c, steps, gal_keys, pts_per_input[t] if pts_per_input is not None else None)
                for t, c in enumerate(cts)]
    L = cts[0].parms_id()
    plans = [ev.rotation_plan(s, gal_keys) for s in steps]
    parents, ext, elts, node_pts = [], [], [], []
    leaf = [[-1] * len(plans) for _ in cts]
    for t, ct in enumerate(cts):
        index = {}
        for i, p in enumerate(plans):
            c = -1
            for depth, elt in enumerate(p):
                fuse = pts_per_input is not None and depth == len(p) - 1
                key = (c, elt, i if fuse else -1)
                j = index.get(key)
                if j is None:
                    j = index[key] = len(parents)
                    parents.append(c), ext.append(ct.data if c < 0 else None), elts.append(elt)
                    node_pts.append(pts_per_input[t][i].data if fuse else None)
                c = j
            leaf[t][i] = c
    outs = be.apply_galois_forest(L, parents, ext, elts, [gal_keys.key(e) for e in elts],
                                  node_pts if pts_per_input is not None else None)
    res = []
    for t, ct in enumerate(cts):
        row = []
        for i, j in enumerate(leaf[t]):
            scale = ct.scale
            if pts_per_input is not None:
                p = pts_per_input[t][i]
                scale *= p.scale
                ev._check_scale(scale, L)
                if p.parms_id() != L:
                    raise ValueError("encrypted_ntt and plain_ntt parameter mismatch")
                if p.is_zero:
                    raise RuntimeError("result ciphertext is transparent")
            row.append(Ciphertext()._set(outs[j], 2, L, scale))
        res.append(row)
    return res


def _linear_transforms_of_inputs(ev: Evaluator, cts: Sequence[Ciphertext],
                                 diag_sets_per_input: Sequence[Sequence[Sequence[Plaintext]]],
                                 gal_keys: KSwitchKeys) -> List[List[Ciphertext]]:
    """[_linear_transforms_of_one_input(ct, sets) for ct, sets] with the rotations of ALL inputs formed together
    (matrix_multiplication.cpp:40-43: the V_k of ctA0 and the W_k of ctB0 are independent of one another): the -d rotations
    as one batch, the rotation forests in one call, every plaintext product and sum in one pass.  Per input the same
    operations in the same order: same bits."""
    ds = [len(s[0]) for s in diag_sets_per_input if s]
    if len(cts) < 2 or len(ds) != len(cts) or len(set(ds)) != 1 or \
            any(len(x) != ds[0] for s in diag_sets_per_input for x in s) or not hasattr(ev.be, "apply_galois_forest"):
        return [_linear_transforms_of_one_input(ev, c, s, gal_keys) for c, s in zip(cts, diag_sets_per_input)]
    d = ds[0]
    dup = _rotations_of_many(ev, cts, [-d], gal_keys)                 # helper.h:244
    ct_news = [ev.add(c, r[0]) for c, r in zip(cts, dup)]             # :247
    rots = _rotations_of_many(ev, ct_news, list(range(1, d)), gal_keys)   # :255
    terms_ct, terms_pt, counts = [], [], []
    for ct_new, rr, sets in zip(ct_news, rots, diag_sets_per_input):
        full = [ct_new] + rr
        terms_ct += [full[l] for _ in sets for l in range(d)]
        terms_pt += [s[l] for s in sets for l in range(d)]
        counts.append(len(sets))
    sums = ev.multiply_plain_sum(terms_ct, terms_pt, group=d)          # :250, :256, :259
    out, at = [], 0
    for k in counts:
        out.append(sums[at:at + k])
        at += k
    return out


def cc_matrix_multiplication(ev: Evaluator, ctA: Ciphertext, ctB: Ciphertext, dimension: int,
                             U_sigma: Sequence[Plaintext], U_tau: Sequence[Plaintext],
                             V_diagonals: Sequence[Sequence[Plaintext]], W_diagonals: Sequence[Sequence[Plaintext]],
                             gal_keys: KSwitchKeys) -> Ciphertext:
    """CC_Matrix_Multiplication, /root/reference/matrix_multiplication.cpp:11-132 (Jiang et al. 2018/1041)."""
    # Step 1's two transforms are independent of one another, and so are Step 2's two families: each pair runs in
    # lockstep (round 6) -- 2 x (1 + NAF depth) dependent launch sequences for the whole product instead of 4 x
    ctA0, ctB0 = linear_transforms_plain_many(ev, [ctA, ctB], [U_sigma, U_tau], gal_keys)    # :22, :25
    if dimension > 1:
        ctAk, ctBk = _linear_transforms_of_inputs(ev, [ctA0, ctB0], [V_diagonals, W_diagonals], gal_keys)   # :42, :43
    else:
        ctAk, ctBk = [], []
    return _matmul_step3(ev, ctA0, ctB0, ctAk, ctBk)


# ---- the matrix product restricted to the non-zero diagonals ------------------------------------
# The permutation matrices of CC_Matrix_Multiplication have 2n-1 (sigma), n (tau), 2 (phi_k) and 1 (psi_k) non-zero
# diagonals out of n^2; the reference adds 1e-8 to every entry (matrix_multiplication.cpp:239-297) so that SEAL does
# not refuse the all-zero ones as transparent, and rotates for all n^2.  Dropping them changes the result by those
# epsilons only (below CKKS noise) -- a fast mode, not the reference's bits: 2n * n^2 rotations become 3n + 3(n-1) - 1.
def matmul_permutation_matrices(n: int):
    """U_sigma, U_tau, [V_k], [W_k] (k = 1..n-1) on the row-major flattening: helper.h:702-851 (get_U_sigma, get_U_tau,
    get_V_k, get_W_k), i.e. sigma, tau, phi^k, psi^k of Jiang et al. 2018/1041."""
    d = n * n
    Us, Ut = np.zeros((d, d)), np.zeros((d, d))
    for i in range(n):
        for j in range(n):
            Us[n * i + j, n * i + (i + j) % n] = 1
            Ut[n * i + j, n * ((i + j) % n) + j] = 1
    V, W = [], []
    for k in range(1, n):
        Vk, Wk = np.zeros((d, d)), np.zeros((d, d))
        for i in range(n):
            for j in range(n):
                Vk[n * i + j, n * i + (j + k) % n] = 1
                Wk[n * i + j, n * ((i + k) % n) + j] = 1
        V.append(Vk)
        W.append(Wk)
    return Us, Ut, V, W


def nonzero_diagonals(U: np.ndarray) -> dict:
    """{l: l-th diagonal (helper.h:175-195)} for the diagonals of U that are not identically zero"""
    n = U.shape[0]
    idx = np.arange(n)
    out = {}
    for l in range(n):
        dg = U[idx, (idx + l) % n]
        if dg.any():
            out[l] = dg
    return out


def matmul_permutation_diagonals(n: int):
    """nonzero_diagonals of the four families of matmul_permutation_matrices(n), built directly from the index maps
    (the dense d x d matrices take 134 MB each at n = 64): ({l: diag} for sigma, for tau, [for phi^k], [for psi^k])"""
    d = n * n

    def collect(entries):
        out = {}
        for r, c in entries:
            out.setdefault((c - r) % d, np.zeros(d))[r] = 1.0
        return dict(sorted(out.items()))

    ij = [(i, j) for i in range(n) for j in range(n)]
    sigma = collect((n * i + j, n * i + (i + j) % n) for i, j in ij)
    tau = collect((n * i + j, n * ((i + j) % n) + j) for i, j in ij)
    phi = [collect((n * i + j, n * i + (j + k) % n) for i, j in ij) for k in range(1, n)]
    psi = [collect((n * i + j, n * ((i + k) % n) + j) for i, j in ij) for k in range(1, n)]
    return sigma, tau, phi, psi


def _duplicate(ev: Evaluator, ct: Ciphertext, d: int, gal_keys: KSwitchKeys) -> Ciphertext:
    return ev.add(ct, ev.rotate_vector(ct, -d, gal_keys))            # helper.h:244-247


def _sparse_products(ev: Evaluator, ct_new: Ciphertext, terms, gal_keys: KSwitchKeys,
                     hoisted: bool = False) -> List[Ciphertext]:
    """[diag (.) rot_l(ct_new) for (l, diag) in terms], the rotations as one batch (fused with their products);
    hoisted=True shares the digit decomposition of ct_new (direct Galois keys, hefx_rotate_hoisted_batch)"""
    rot = [(i, l, p) for i, (l, p) in enumerate(terms) if l]
    out = [None] * len(terms)
    for i, (l, p) in enumerate(terms):
        if not l:
            out[i] = ev.multiply_plain(ct_new, p)                    # :250
    if hoisted and rot:
        L = ct_new.parms_id()
        plans = [ev.rotation_plan(l, gal_keys) for _, l, _ in rot]
        if any(len(pl) != 1 for pl in plans):
            raise ValueError("hoisted rotations need a direct Galois key for every step")
        scale = _plain_product_scale(ev, ct_new, [p for _, _, p in rot])
        elts = [pl[0] for pl in plans]
        data = ev.be.rotate_hoisted_batch(L, ct_new.data, elts, [gal_keys.key(x) for x in elts],
                                          [p.data for _, _, p in rot])
        for (i, _, _), x in zip(rot, data):
            out[i] = Ciphertext()._set(x, 2, L, scale)
        return out
    for (i, _, _), c in zip(rot, _rotations_batched(ev, ct_new, [l for _, l, _ in rot], gal_keys,
                                                    [p for _, _, p in rot])):   # :252-257
        out[i] = c
    return out


def linear_transform_plain_sparse(ev: Evaluator, ct: Ciphertext, d: int, diagonals: dict,
                                  gal_keys: KSwitchKeys, hoisted: bool = False) -> Ciphertext:
    """Linear_Transform_Plain (helper.h:237-262) of a d x d matrix given by its non-zero diagonals {l: Plaintext}"""
    if not diagonals:
        raise ValueError("encrypteds cannot be empty")
    if hoisted == 2:
        return _linear_transform_plain_sparse_hoisted2(ev, ct, d, diagonals, gal_keys)
    ct_new = _duplicate(ev, ct, d, gal_keys)
    return ev.add_many(_sparse_products(ev, ct_new, sorted(diagonals.items()), gal_keys, hoisted))   # :259


def _linear_transform_plain_sparse_hoisted2(ev: Evaluator, ct: Ciphertext, d: int, diagonals: dict,
                                            gal_keys: KSwitchKeys) -> Ciphertext:
    """double hoisting (one mod-down for the whole transform, see _linear_transform_plain_hoisted2) over the non-zero
    diagonals: KEY-LEVEL plaintexts, ct at the top data level, diagonal 0 present, direct keys for the other steps"""
    ctx, be, L = ev.ctx, ev.be, ct.parms_id()
    if ct.size() != 2:
        raise ValueError("encrypted size must be 2")
    if L != ctx.first_parms_id():
        raise ValueError("double hoisting is built for the top data level")
    if 0 not in diagonals:
        raise ValueError("double hoisting over a subset of diagonals needs diagonal 0")
    steps = [0] + sorted(l for l in diagonals if l)
    pts = [diagonals[l] for l in steps]
    if any(p.parms_id() != ctx.k for p in pts):
        raise ValueError("double hoisting needs key-level plaintexts (encode with parms_id = key level)")
    scale = None
    for p in pts:
        s = ct.scale * p.scale
        ev._check_scale(s, L)
        if scale is not None and not ev._close(scale, s):
            raise ValueError("scale mismatch")
        scale = s if scale is None else scale
        if p.is_zero:
            raise RuntimeError("result ciphertext is transparent")
    native = getattr(be, "linear_transform_plain_hoisted2_sparse", None)
    if native is not None:
        elts = sorted(gal_keys.keys)
        data = native(L, ct.data, d, steps, [p.data for p in pts], elts, [gal_keys.key(e) for e in elts])
    else:  # the oracle twin: regular -d rotation, then the oracle's statement of the double-hoisted core
        plans = [ev.rotation_plan(l, gal_keys) for l in steps[1:]]
        if any(len(p) != 1 for p in plans):
            raise ValueError("hoisted linear transform needs a direct Galois key for every step")
        ct_new = _duplicate(ev, ct, d, gal_keys)
        elts = [p[0] for p in plans]
        data = be.lt_double_hoisted_core(ct_new.data, [p.data for p in pts], elts, [gal_keys.key(e) for e in elts])
    return Ciphertext()._set(data, 2, L, scale)


def cc_matrix_multiplication_sparse(ev: Evaluator, ctA: Ciphertext, ctB: Ciphertext, dimension: int, U_sigma: dict,
                                    U_tau: dict, V_diagonals: Sequence[dict], W_diagonals: Sequence[dict],
                                    gal_keys: KSwitchKeys, hoisted: bool = False) -> Ciphertext:
    """CC_Matrix_Multiplication (matrix_multiplication.cpp:11-132) over the non-zero diagonals only.  The 2(n-1)
    Step-2 transforms read the same two duplicated ciphertexts, so all their rotations go out as two batches.
    hoisted=True (direct Galois keys for every step) runs the 2n-1 / n rotations of the sigma / tau transforms on one
    shared digit decomposition each; hoisted=2 additionally mods them down once per transform (U_sigma / U_tau then
    hold KEY-LEVEL plaintexts, see _linear_transform_plain_sparse_hoisted2; Step 2 stays single-hoisted)."""
    d = dimension * dimension
    ctA0 = linear_transform_plain_sparse(ev, ctA, d, U_sigma, gal_keys, hoisted)   # :22
    ctB0 = linear_transform_plain_sparse(ev, ctB, d, U_tau, gal_keys, hoisted)     # :25
    ctAk = _sparse_transforms_of_one_input(ev, ctA0, d, V_diagonals, gal_keys)     # :42
    ctBk = _sparse_transforms_of_one_input(ev, ctB0, d, W_diagonals, gal_keys)     # :43
    return _matmul_step3(ev, ctA0, ctB0, ctAk, ctBk)


def _sparse_transforms_of_one_input(ev: Evaluator, ct0: Ciphertext, d: int, diag_dicts: Sequence[dict],
                                    gal_keys: KSwitchKeys) -> List[Ciphertext]:
    """[linear_transform_plain_sparse(ct0, d, diags) for diags in diag_dicts] (matrix_multiplication.cpp:40-43 over the
    non-zero diagonals): the transforms read the same duplicated ciphertext, so the rotations they need go out as one
    batch and -- when every transform has the same number of diagonals (phi^k: 2, psi^k: 1) -- all sums in one pass."""
    if not diag_dicts:
        return []
    ct_new = _duplicate(ev, ct0, d, gal_keys)
    terms = [(l, p) for diags in diag_dicts for l, p in sorted(diags.items())]
    counts = {len(diags) for diags in diag_dicts}
    if len(counts) == 1:
        steps = sorted({l for l, _ in terms if l})
        rot = dict(zip(steps, _rotations_batched(ev, ct_new, steps, gal_keys)))
        rot[0] = ct_new
        return ev.multiply_plain_sum([rot[l] for l, _ in terms], [p for _, p in terms], group=counts.pop())
    prods = _sparse_products(ev, ct_new, terms, gal_keys)
    cts, pos = [], 0
    for diags in diag_dicts:
        cts.append(ev.add_many(prods[pos:pos + len(diags)]))
        pos += len(diags)
    return cts


# ---- polynomial evaluation and encrypted logistic regression (logistic_regression_ckks.cpp) -------------------
def tree_cipher(ev: Evaluator, encoder: CKKSEncoder, encryptor, ctx: Ciphertext, degree: int, scale: float,
                coeffs: Sequence[float], relin_keys: KSwitchKeys) -> Ciphertext:
    """Tree_cipher, /root/reference/logistic_regression_ckks.cpp:55-137 (= polynomial.cpp:233-359)."""
    plain = [None if coeffs[i] == 0 else encoder.encode(float(coeffs[i]), scale) for i in range(degree + 1)]  # :71-83
    powers = compute_all_powers(ev, ctx, degree, relin_keys)                                              # :93
    result = encryptor.encrypt(plain[0])                                                                   # :102
    for i in range(1, degree + 1):                                                                         # :110
        ev.mod_switch_to_inplace(plain[i], powers[i].parms_id())                                          # :113
        temp = ev.multiply_plain(powers[i], plain[i])                                                      # :116
        ev.rescale_to_next_inplace(temp)                                                                   # :119
        ev.mod_switch_to_inplace(result, temp.parms_id())                                                  # :122
        result.scale = 2.0 ** int(np.log2(result.scale))                                                   # :126-127
        temp.scale = 2.0 ** int(np.log2(result.scale))
        ev.add_inplace(result, temp)                                                                       # :130
    return result


def horner_cipher(ev: Evaluator, encoder: CKKSEncoder, encryptor, ctx: Ciphertext, degree: int,
                  coeffs: Sequence[float], scale: float, relin_keys: KSwitchKeys) -> Ciphertext:
    """Horner_cipher, /root/reference/logistic_regression_ckks.cpp:139-205."""
    plain = [encoder.encode(float(coeffs[i]), scale) for i in range(degree + 1)]                          # :147-157
    temp = encryptor.encrypt(plain[degree])                                                                # :162
    ctx = ctx.copy()
    for i in range(degree - 1, -1, -1):                                                                    # :168
        if ctx.parms_id() > temp.parms_id():                                                               # :172-182
            ev.mod_switch_to_inplace(ctx, temp.parms_id())
        elif ctx.parms_id() < temp.parms_id():
            ev.mod_switch_to_inplace(temp, ctx.parms_id())
        ev.multiply_inplace(temp, ctx)                                                                     # :184
        ev.relinearize_inplace(temp, relin_keys)                                                           # :187
        ev.rescale_to_next_inplace(temp)                                                                   # :189
        ev.mod_switch_to_inplace(plain[i], temp.parms_id())                                                # :192
        temp.scale = 2.0 ** 40                                                                             # :195 (the reference hard-codes 2^40)
        ev.add_plain_inplace(temp, plain[i])                                                               # :198
    return temp


def cipher_dot_product_many(ev: Evaluator, As: Sequence[Ciphertext], Bs: Sequence[Ciphertext], size: int,
                            relin_keys: KSwitchKeys, gal_keys: KSwitchKeys, log_sum: bool = False) -> List[Ciphertext]:
    """len(As) independent cipher_dot_product calls (helper.h:416-502) advanced in lockstep: the rotate-by-1 chain
    inside one dot product is sequential, but the chains of different rows are independent, so every step is ONE
    batched key-switch launch over all rows.  Same calls in the same order per row -> same bits as the loop at
    logistic_regression_ckks.cpp:217-220.

    log_sum=True is a fast mode (not the reference's bits): the window sum sum_{t<size} rot^t(dup) by doubling --
    S(2k) = S(k) + rot^k(S(k)), S(2k+1) = S(2k) + rot^(2k)(dup) -- i.e. about log2(size) rotations instead of size-1
    (size 8: 3 instead of 7).  Slots 0..size-1 hold the same replicated dot product as the reference's chain; slots
    size..2*size-1 differ (they carry one more copy of the products), which no caller reads."""
    be, n = ev.be, len(As)
    if n == 0:  # a rank of a row-sharded prediction that owns no row (parallel.predict_cipher_weights_sharded)
        return []
    mults = ev.multiply_many(As, Bs)                                 # :432
    ev.relinearize_many_inplace(mults, relin_keys)                   # :440
    ev.rescale_to_next_many_inplace(mults)                           # :441
    L = mults[0].parms_id()

    def rotate_all(cts, step):
        plan = ev.rotation_plan(step, gal_keys)
        data = [c.data for c in cts]
        for elt in plan:
            data = be.apply_galois_batch(L, data, [elt] * n, [gal_keys.key(elt)] * n)
        return [Ciphertext()._set(d, 2, L, c.scale) for d, c in zip(data, cts)]

    dups = ev.add_pairs(mults, rotate_all(mults, -size))             # :455, :464
    if log_sum:
        acc, k = dups, 1
        for b in bin(size)[3:]:  # the bits of size below its leading one
            acc, k = ev.add_pairs(acc, rotate_all(acc, k)), 2 * k
            if b == "1":
                acc, k = ev.add_pairs(acc, rotate_all(dups, k)), k + 1
        mults = acc
    else:
        plan1 = ev.rotation_plan(1, gal_keys)
        chain = getattr(be, "rotate_add_chain", None)
        if chain is not None and len(plan1) == 1 and size > 1:
            # :472-476 as ONE engine call (hefx_rotate_add_chain): the same key switches and sums level by level -- the
            # oracle-backed twin of the tests runs the loop below, and the bits agree
            elt, key = plan1[0], gal_keys.key(plan1[0])
            _, sums = chain(L, [c.data for c in dups], [elt] * n, [key] * n, [m.data for m in mults], size - 1)
            mults = [Ciphertext()._set(d, 2, L, m.scale) for d, m in zip(sums, mults)]
        else:
            for _ in range(1, size):                                 # :472-476
                dups = rotate_all(dups, 1)
                mults = ev.add_pairs(mults, dups)
    for m in mults:
        m.scale = 2.0 ** int(np.log2(m.scale))                       # :489 "manual rescale"
    return mults


SIGMOID_COEFFS = {3: [0.5, 1.20069, 0.00001, -0.81562],
                  5: [0.5, 1.53048, 0.00001, -2.3533056, 0.00001, 1.3511295],
                  7: [0.5, 1.73496, 0.00001, -4.19407, 0.00001, 5.43402, 0.00001, -2.50739]}


def predict_cipher_weights(ev: Evaluator, encoder: CKKSEncoder, encryptor, features: Sequence[Ciphertext],
                           weights: Ciphertext, num_weights: int, scale: float, gal_keys: KSwitchKeys,
                           relin_keys: KSwitchKeys, degree: int = 3, log_sum: bool = False) -> Ciphertext:
    """predict_cipher_weights, /root/reference/logistic_regression_ckks.cpp:208-266.  log_sum=True: the fast window
    sum of cipher_dot_product_many (same prediction values, not the reference's noise bits)."""
    num_rows = len(features)
    results = cipher_dot_product_many(ev, features, [weights] * num_rows, num_weights, relin_keys, gal_keys,
                                      log_sum)                                                             # :220
    # the reference encodes one one-hot mask per row inside the loop (:222-225, 2000 CPU FFTs per iteration); here all
    # masks go through one batched encode (one GPU launch on the HIP engine) -- same plaintexts, same order of use
    # masks go through one batched encode (one GPU launch on the HIP engine) -- same plaintexts, same order of use.
    # Encoding at the level below the top IS encode + mod_switch_to_next (:227): CKKS drops RNS rows, no arithmetic.
    top = ev.ctx.first_parms_id()
    masks = encoder.encode_many(list(np.eye(num_rows)), scale, parms_id=top - 1)                           # :222-227
    lin = ev.multiply_plain_sum(results, masks)[0]                                                         # :229, :233
    ev.relinearize_inplace(lin, relin_keys)                                                                # :237 (no-op)
    ev.rescale_to_next_inplace(lin)                                                                        # :239
    lin.scale = 2.0 ** int(np.log2(lin.scale))                                                             # :242
    coeffs = SIGMOID_COEFFS[degree]                                                                        # :245-262
    return horner_cipher(ev, encoder, encryptor, lin, len(coeffs) - 1, coeffs, scale, relin_keys)         # :264


def update_weights(ev: Evaluator, encoder: CKKSEncoder, encryptor, features: Sequence[Ciphertext],
                   features_T: Sequence[Ciphertext], labels: Ciphertext, weights: Ciphertext, learning_rate: float,
                   gal_keys: KSwitchKeys, relin_keys: KSwitchKeys, scale: float, degree: int = 3) -> Ciphertext:
    """update_weights, /root/reference/logistic_regression_ckks.cpp:269-345.  As committed, the reference cannot
    get past :336 with its own parameters: the gradient is at the last level (one 60-bit prime left) and the
    multiply_plain pushes the scale to 2^80 -> SEAL throws invalid_argument("scale out of bounds").  The mirror
    keeps that behaviour (ValueError from Evaluator._check_scale)."""
    num_obs, num_weights = len(features), len(features_T)
    pred = predict_cipher_weights(ev, encoder, encryptor, features, weights, num_weights, scale, gal_keys,
                                  relin_keys, degree)                                                       # :282
    labels = labels.copy()
    ev.mod_switch_to_inplace(labels, pred.parms_id())                                                      # :286
    pred_labels = ev.sub(pred, labels)                                                                     # :288
    fT = [f.copy() for f in features_T]
    for f in fT:
        ev.mod_switch_to_inplace(f, pred_labels.parms_id())                                                # :298
    grads = cipher_dot_product_many(ev, fT, [pred_labels] * num_weights, num_obs, relin_keys, gal_keys)    # :299
    masks = encoder.encode_many(list(np.eye(num_weights)), scale, parms_id=grads[0].parms_id())           # :302-308
    gradient = ev.multiply_plain_sum(grads, masks)[0]                                                      # :310, :316
    ev.relinearize_inplace(gradient, relin_keys)                                                           # :319
    ev.rescale_to_next_inplace(gradient)                                                                   # :321
    gradient.scale = 2.0 ** int(np.log2(gradient.scale))                                                   # :324
    n_pt = encoder.encode(float(learning_rate) / num_obs, scale)                                           # :330-331
    ev.mod_switch_to_inplace(n_pt, gradient.parms_id())                                                    # :333
    ev.multiply_plain_inplace(gradient, n_pt)                                                              # :336  <- SEAL throws here
    new_weights = ev.sub(gradient, weights)                                                                # :341
    ev.negate_inplace(new_weights)                                                                         # :342
    return new_weights

"""The reference's composite HE algorithms (its L3 layer), restated over the seal.py API mirror so that the same
code drives the HIP engine (product) and, in tests, the oracle-backed twin.  Each function follows the
reference line by line in WHAT it calls and in which order -- results must be bit-identical -- but is free in HOW
the calls are batched: independent rotations of one ciphertext go to the engine as one batch.

Citations are /root/reference/helper.h unless noted.
"""
from __future__ import annotations

from typing import List, Sequence

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
    for depth in range(max((len(p) for p in plans), default=0)):
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


def linear_transform_plain(ev: Evaluator, ct: Ciphertext, U_diagonals: Sequence[Plaintext],
                           gal_keys: KSwitchKeys) -> Ciphertext:
    """Linear_Transform_Plain, helper.h:237-262 (= linear_transformation2.cpp:149-174)."""
    d = len(U_diagonals)
    ct_rot = ev.rotate_vector(ct, -d, gal_keys)                      # :244  fill with duplicate
    ct_new = ev.add(ct, ct_rot)                                      # :247
    res = [ev.multiply_plain(ct_new, U_diagonals[0])]                # :250
    res += _rotations_batched(ev, ct_new, list(range(1, d)), gal_keys, U_diagonals[1:])   # :252-257
    return ev.add_many(res)                                          # :259


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
    return ev.add_many([ev.multiply_plain(U_diagonals[i], pt_rotations[i]) for i in range(len(pt_rotations))])


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


def cc_matrix_multiplication(ev: Evaluator, ctA: Ciphertext, ctB: Ciphertext, dimension: int,
                             U_sigma: Sequence[Plaintext], U_tau: Sequence[Plaintext],
                             V_diagonals: Sequence[Sequence[Plaintext]], W_diagonals: Sequence[Sequence[Plaintext]],
                             gal_keys: KSwitchKeys) -> Ciphertext:
    """CC_Matrix_Multiplication, /root/reference/matrix_multiplication.cpp:11-132 (Jiang et al. 2018/1041)."""
    ctA0 = linear_transform_plain(ev, ctA, U_sigma, gal_keys)        # :22
    ctB0 = linear_transform_plain(ev, ctB, U_tau, gal_keys)          # :25
    ctAk = [linear_transform_plain(ev, ctA0, V_diagonals[k], gal_keys) for k in range(dimension - 1)]   # :42
    ctBk = [linear_transform_plain(ev, ctB0, W_diagonals[k], gal_keys) for k in range(dimension - 1)]   # :43
    for c in ctAk + ctBk:
        ev.rescale_to_next_inplace(c)                                # :69-73
    ctAB = ev.multiply(ctA0, ctB0)                                   # :104
    ev.mod_switch_to_next_inplace(ctAB)                              # :112
    for c in ctAk + ctBk:
        c.scale = 2.0 ** int(np.log2(c.scale))                       # :117-121 "manual rescale"
    for k in range(dimension - 1):
        ev.add_inplace(ctAB, ev.multiply(ctAk[k], ctBk[k]))          # :123-129
    return ctAB

"""Multi-GPU execution of the hot path: one process per GPU, torch.distributed over RCCL/xGMI.

The path shards by INDEPENDENT UNITS -- diagonals of one linear transform, the 2(n-1) step-2 transforms of a
matrix product, rows of an LR batch, or simply independent ciphertexts -- with every rank holding the input
ciphertext and all keys.  The only exchange is the final ciphertext sum: one all-reduce(SUM) of uint64 words
followed by a local canonicalisation.  Exact: residues are < 2^61, at most 8 ranks add, so the integer sum cannot
wrap, and modular addition is associative, so the result has the same bits as the serial add_many
(/root/reference/helper.h:259).  Messages are 0.4-4 MB (size*L*N*8 B): latency-bound on xGMI, so one all-reduce per
transform, never one per diagonal.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import numpy as np

from . import algorithms as alg
from .seal import Ciphertext, Evaluator, KSwitchKeys, Plaintext


def shard(n: int, rank: int, world: int) -> range:
    """contiguous block partition of range(n)"""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def init_engine_comm(ev: Evaluator, group=None) -> None:
    """Attach an RCCL communicator to the evaluator's HIP engine (hefx_comm_init): rank 0 draws the 128-byte id, the
    torch.distributed group only carries it to the other ranks.  From then on allreduce_ciphertext runs entirely behind
    the C-ABI (hefx_allreduce_sum: RCCL in place on the payload + local canonicalisation)."""
    import torch.distributed as dist
    eng = ev.be.engine
    if eng.comm_world:
        return
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [eng.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    eng.comm_init(world, rank, box[0])


# How the exchange runs on the HIP engine (HEFX_ENGINE_COMM = auto | on | off; bench.py sets the attribute directly):
#   auto  under torch's "nccl" backend (one device per rank, the precondition of RCCL) the engine attaches its OWN
#         communicator on first use (init_engine_comm) and every exchange is hefx_allreduce_sum -- RCCL in place on the
#         payload plus the local canonicalisation, nothing of it in Python; any other backend takes the torch path
#   on    require the engine communicator (raise if it cannot be attached);   off   always the torch path
ENGINE_COMM = os.environ.get("HEFX_ENGINE_COMM", "auto")


class _DeviceWords:
    """zero-copy view of an engine buffer for torch (torch.as_tensor reads __cuda_array_interface__): the collective
    runs in place on the payload, no staging tensor"""

    def __init__(self, ptr: int, nwords: int):
        self.__cuda_array_interface__ = {"shape": (nwords,), "typestr": "<i8", "data": (int(ptr), False), "version": 3,
                                         "strides": None}


# (id(group object), world) -> (weak reference to the group object, bool): the collective outcome of the handshake below,
# equal on every rank.  The weak reference guards against an id reused by a later group: an entry counts only
# while it still points at the very object it was made for.
_comm_decisions = {}


def _group_object(group):
    import torch.distributed as dist
    return group if group is not None else dist.group.WORLD


def _decision_get(group, world):
    import weakref
    g = _group_object(group)
    ent = _comm_decisions.get((id(g), world))
    if ent is None:
        return None
    if ent[0]() is not g:  # the id belongs to another (dead) group's entry
        del _comm_decisions[(id(g), world)]
        return None
    return ent[1]


def _decision_set(group, world, ok):
    import weakref
    g = _group_object(group)
    try:
        _comm_decisions[(id(g), world)] = (weakref.ref(g), ok)
    except TypeError:  # not weak-referenceable: decide anew next time (one small all-reduce)
        pass


def _decision_drop(group, world):
    _comm_decisions.pop((id(_group_object(group)), world), None)


def _all_true(flag: bool, group, backend: str) -> bool:
    """logical AND of `flag` over the ranks of `group` (one small all-reduce)"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()) == 1)


def _engine_comm_ready(ev: Evaluator, group, world: int) -> bool:
    """Whether the exchange runs behind the C-ABI (hefx_allreduce_sum) -- decided COLLECTIVELY, once per group: the ranks
    take the engine path only if every one of them holds (or could attach) the communicator; a rank that cannot (its
    engine already serves another group, RCCL failed to initialise there) makes ALL ranks take the torch path instead of
    leaving its peers alone in a collective.  Whether to try at all depends only on the mode and the backend, which are
    the same everywhere, so the handshake itself is entered by every rank or by none."""
    eng = ev.be.engine
    import torch.distributed as dist
    backend = dist.get_backend(group)
    if ENGINE_COMM == "off" or not (ENGINE_COMM == "on" or backend == "nccl"):
        return False
    known = _decision_get(group, world)
    if known is not None:
        if known and eng.comm_world != world:  # destroyed behind our back (bench.py's opt-in leg does)
            _decision_drop(group, world)
        else:
            return known
    have = eng.comm_world == world
    if _all_true(have, group, backend):
        _decision_set(group, world, True)
        return True
    ok = _all_true(have or eng.comm_world == 0, group, backend)
    err = None
    if ok:
        try:
            if have:
                eng.comm_destroy()  # some peer lacks it: everybody attaches anew (comm_init is collective)
            init_engine_comm(ev, group)
        except Exception as ex:  # reported below, after the ranks have agreed on the outcome
            err = ex
        ok = _all_true(err is None and eng.comm_world == world, group, backend)
        if not ok and eng.comm_world:
            try:
                eng.comm_destroy()
            except Exception:
                pass
    if not ok and ENGINE_COMM == "on":
        raise RuntimeError("HEFX_ENGINE_COMM=on, but the engine communicator could not be attached on every rank"
                           + (f": {err!r}" if err else ""))
    _decision_set(group, world, ok)
    return ok


def allreduce_ciphertext(ev: Evaluator, ct: Ciphertext, group=None) -> Ciphertext:
    """sum of every rank's `ct` (same level/scale/size), bit-identical to a serial add_many"""
    import torch
    import torch.distributed as dist
    be, L, size, N = ev.be, ct.parms_id(), ct.size(), ev.ctx.N
    world = dist.get_world_size(group)
    if world > 8:
        raise ValueError("the wrap-free uint64 sum argument holds for at most 8 addends of < 2^61")
    if be.name == "hip" and _engine_comm_ready(ev, group, world):  # communicator behind the C-ABI
        out = be.engine.copy(ct.data)
        be.engine.allreduce_sum(L, size, out)
        return Ciphertext()._set(out, size, L, ct.scale)
    if be.name == "hip":
        # torch path: the collective runs IN PLACE on a copy of the payload, viewed as a torch tensor without staging.
        # Engine calls and torch both submit to the device's default stream, so the copy is ordered before the
        # collective and the canonicalisation after it; only a caller that has moved torch to another stream pays a wait.
        eng = be.engine
        out = eng.copy(ct.data)
        dev = torch.device("cuda", eng.device)
        cur = torch.cuda.current_stream(dev)
        try:
            t = torch.as_tensor(_DeviceWords(out.ptr, size * L * N), device=dev)
            staged = False
        except Exception:  # a torch build without __cuda_array_interface__ support: one staging tensor, as in round 2
            t = torch.empty(size * L * N, dtype=torch.int64, device=dev)
            eng.copy_raw(t.data_ptr(), out.ptr, t.numel() * 8)
            staged = True
        if cur.cuda_stream != 0:
            eng.sync()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        if staged:
            eng.copy_raw(out.ptr, t.data_ptr(), t.numel() * 8)
        # nccl: a synchronous collective makes torch's current stream wait for it (work.wait()), and that stream is the
        # null stream the engine submits to unless the caller moved torch elsewhere.  Any other backend (gloo with device
        # tensors stages through the host) only PROMISES completion to the host, so the host waits before the engine reads.
        if cur.cuda_stream != 0 or dist.get_backend(group) != "nccl":
            cur.synchronize()
        eng.reduce_canonical(L, size, out, addends=world)
        data = out
    else:  # oracle-backed twin (tests, gloo)
        t = torch.from_numpy(np.ascontiguousarray(be.to_host(ct.data)).view(np.int64).reshape(-1).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        data = be.reduce_canonical(L, size, t.numpy().view(np.uint64).reshape(size, L, N).copy(), world)
    return Ciphertext()._set(data, size, L, ct.scale)


def linear_transform_plain_sharded(ev: Evaluator, ct: Ciphertext, U_diagonals: Sequence[Plaintext],
                                   gal_keys: KSwitchKeys, group=None) -> Ciphertext:
    """Linear_Transform_Plain (helper.h:237-262) with the diagonals l in [0,d) split over the ranks of `group`.
    Every rank returns the full result."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    d = len(U_diagonals)
    ct_new = ev.add(ct, ev.rotate_vector(ct, -d, gal_keys))            # replicated: every rank needs ct_new
    mine = list(shard(d, rank, world))
    res: List[Ciphertext] = []
    steps = [l for l in mine if l > 0]
    if 0 in mine:
        res.append(ev.multiply_plain(ct_new, U_diagonals[0]))
    res += alg._rotations_batched(ev, ct_new, steps, gal_keys, [U_diagonals[l] for l in steps])
    if res:
        partial = ev.add_many(res)
    else:  # more ranks than diagonals: contribute the zero ciphertext at the right level/scale
        z = ev.multiply_plain(ct_new, U_diagonals[0])
        partial = ev.sub(z, z)
    return allreduce_ciphertext(ev, partial, group)


def linear_transforms_plain_sharded_many(ev: Evaluator, cts: Sequence[Ciphertext],
                                         diag_sets: Sequence[Sequence[Plaintext]], gal_keys: KSwitchKeys,
                                         group=None) -> List[Ciphertext]:
    """[linear_transform_plain_sharded(ct, diags) for ct, diags] for INDEPENDENT transforms of one dimension (the sigma / tau
    transforms of CC_Matrix_Multiplication, matrix_multiplication.cpp:22-25) with this rank's share of BOTH in lockstep:
    the -d rotations as one batch, the rank's diagonals of every input in one rotation forest (alg._rotations_of_many), then
    one all-reduce per transform.  Per transform the operations of linear_transform_plain_sharded: same bits."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    d = len(diag_sets[0]) if diag_sets else 0
    if len(cts) < 2 or any(len(ds) != d for ds in diag_sets):
        return [linear_transform_plain_sharded(ev, c, ds, gal_keys, group) for c, ds in zip(cts, diag_sets)]
    dup = alg._rotations_of_many(ev, cts, [-d], gal_keys)                          # replicated: helper.h:244
    ct_news = [ev.add(c, r[0]) for c, r in zip(cts, dup)]                          # :247
    mine = list(shard(d, rank, world))
    steps = [l for l in mine if l > 0]
    prods = alg._rotations_of_many(ev, ct_news, steps, gal_keys, [[ds[l] for l in steps] for ds in diag_sets]) \
        if steps else [[] for _ in cts]
    outs = []
    for ct_new, ds, pr in zip(ct_news, diag_sets, prods):
        res = ([ev.multiply_plain(ct_new, ds[0])] if 0 in mine else []) + list(pr)
        if res:
            partial = ev.add_many(res)
        else:  # more ranks than diagonals: the zero ciphertext at the right level / scale
            z = ev.multiply_plain(ct_new, ds[0])
            partial = ev.sub(z, z)
        outs.append(allreduce_ciphertext(ev, partial, group))
    return outs


def _zero_like(ev: Evaluator, ct: Ciphertext) -> Ciphertext:
    """the zero ciphertext at ct's size / level / scale (a rank's contribution when it owns no unit)"""
    return ev.sub(ct, ct)


def cc_matrix_multiplication_sharded(ev: Evaluator, ctA: Ciphertext, ctB: Ciphertext, dimension: int,
                                     U_sigma: Sequence[Plaintext], U_tau: Sequence[Plaintext],
                                     V_diagonals: Sequence[Sequence[Plaintext]],
                                     W_diagonals: Sequence[Sequence[Plaintext]], gal_keys: KSwitchKeys,
                                     group=None) -> Ciphertext:
    """CC_Matrix_Multiplication (/root/reference/matrix_multiplication.cpp:11-132) with Step 2 sharded:
    the two n^2-diagonal transforms of Step 1 are diagonal-sharded (one all-reduce each), the 2(n-1) transforms of
    Step 2 and the products A_k (.) B_k are split by k, and the size-3 partial sums meet in one all-reduce.  Modular
    addition is associative and commutative, so the bits equal the serial order of :123-129."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ctA0, ctB0 = linear_transforms_plain_sharded_many(ev, [ctA, ctB], [U_sigma, U_tau], gal_keys, group)   # :22, :25
    mine = list(shard(dimension - 1, rank, world))
    # this rank's transforms of ctA0 / ctB0 share the rotations of their inputs, and the two inputs run in lockstep
    # (alg._linear_transforms_of_inputs, as the serial form does for all n-1): same bits as transform by transform
    if mine:
        ctAk, ctBk = alg._linear_transforms_of_inputs(ev, [ctA0, ctB0], [[V_diagonals[k] for k in mine],
                                                                          [W_diagonals[k] for k in mine]], gal_keys)   # :42, :43
    else:
        ctAk, ctBk = [], []
    ev.rescale_to_next_many_inplace(ctAk)                                          # :69-73
    ev.rescale_to_next_many_inplace(ctBk)
    ctAB = ev.multiply(ctA0, ctB0)                                                 # :104
    ev.mod_switch_to_next_inplace(ctAB)                                            # :112
    for c in ctAk + ctBk:
        c.scale = 2.0 ** int(np.log2(c.scale))                                     # :117-121
    terms = ([ctAB] if rank == 0 else []) + (ev.multiply_many(ctAk, ctBk) if ctAk else [])   # A_0 (.) B_0 counted once
    partial = ev.add_many(terms) if terms else _zero_like(ev, ctAB)                # :123-129 (this rank's share)
    out = allreduce_ciphertext(ev, partial, group)
    out.scale = ctAB.scale  # the serial sum carries its first addend's scale (A_0 (.) B_0) on every rank
    return out


def linear_transform_plain_sparse_sharded(ev: Evaluator, ct: Ciphertext, d: int, diagonals: dict,
                                          gal_keys: KSwitchKeys, group=None) -> Ciphertext:
    """alg.linear_transform_plain_sparse with the NON-ZERO diagonals split over the ranks (one all-reduce)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ct_new = alg._duplicate(ev, ct, d, gal_keys)                                   # replicated
    items = sorted(diagonals.items())
    mine = [items[i] for i in shard(len(items), rank, world)]
    if mine:
        partial = ev.add_many(alg._sparse_products(ev, ct_new, mine, gal_keys))
    else:
        z = ev.multiply_plain(ct_new, items[0][1])
        partial = ev.sub(z, z)
    return allreduce_ciphertext(ev, partial, group)


def cc_matrix_multiplication_sparse_sharded(ev: Evaluator, ctA: Ciphertext, ctB: Ciphertext, dimension: int,
                                            U_sigma: dict, U_tau: dict, V_diagonals: Sequence[dict],
                                            W_diagonals: Sequence[dict], gal_keys: KSwitchKeys, group=None,
                                            shard_step1: bool = False) -> Ciphertext:
    """BASELINE config 5 (/root/reference/matrix_mult_benchmark.cpp:13-71 at n = 64, N = 32768) in the form that can
    exist there: alg.cc_matrix_multiplication_sparse (non-zero diagonals only) with the 2(n-1) Step-2 transforms
    phi^k / psi^k and the products A_k (.) B_k split by k over the ranks and ONE all-reduce of the size-3 partial sum.
    sigma / tau (Step 1) run replicated (shard_step1=False: a single exchange per product) or split by diagonal with one
    all-reduce each (shard_step1=True: 3 exchanges, ~2x less replicated work at n = 64).  Modular addition is
    associative and commutative: bit-identical to the serial sparse product either way."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    d = dimension * dimension
    if shard_step1:
        ctA0 = linear_transform_plain_sparse_sharded(ev, ctA, d, U_sigma, gal_keys, group)   # :22
        ctB0 = linear_transform_plain_sparse_sharded(ev, ctB, d, U_tau, gal_keys, group)     # :25
    else:
        ctA0 = alg.linear_transform_plain_sparse(ev, ctA, d, U_sigma, gal_keys)
        ctB0 = alg.linear_transform_plain_sparse(ev, ctB, d, U_tau, gal_keys)
    mine = list(shard(dimension - 1, rank, world))
    ctAk = alg._sparse_transforms_of_one_input(ev, ctA0, d, [V_diagonals[k] for k in mine], gal_keys)   # :42
    ctBk = alg._sparse_transforms_of_one_input(ev, ctB0, d, [W_diagonals[k] for k in mine], gal_keys)   # :43
    ev.rescale_to_next_many_inplace(ctAk)                                          # :69-73
    ev.rescale_to_next_many_inplace(ctBk)
    ctAB = ev.multiply(ctA0, ctB0)                                                 # :104
    ev.mod_switch_to_next_inplace(ctAB)                                            # :112
    for c in ctAk + ctBk:
        c.scale = 2.0 ** int(np.log2(c.scale))                                     # :117-121
    terms = ([ctAB] if rank == 0 else []) + (ev.multiply_many(ctAk, ctBk) if ctAk else [])   # A_0 (.) B_0 counted once
    partial = ev.add_many(terms) if terms else _zero_like(ev, ctAB)                # :123-129 (this rank's share)
    out = allreduce_ciphertext(ev, partial, group)
    out.scale = ctAB.scale  # the serial sum carries its first addend's scale (A_0 (.) B_0) on every rank
    return out


def predict_cipher_weights_sharded(ev: Evaluator, encoder, encryptor, features: Sequence[Ciphertext],
                                   weights: Ciphertext, num_weights: int, scale: float, gal_keys: KSwitchKeys,
                                   relin_keys: KSwitchKeys, degree: int = 3, group=None) -> Ciphertext:
    """predict_cipher_weights (/root/reference/logistic_regression_ckks.cpp:208-266) with the observation rows split
    over the ranks: every rank computes and masks the dot products of its rows; the masked partial sums
    meet in one all-reduce; the sigmoid polynomial runs replicated."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    num_rows = len(features)
    mine = list(shard(num_rows, rank, world))
    results = alg.cipher_dot_product_many(ev, [features[i] for i in mine], [weights] * len(mine), num_weights,
                                          relin_keys, gal_keys)                     # :220
    if results:  # masks of this rank's rows in one batched encode, at the level :227 switches them to; mask-and-sum
        # in one pass (same plaintexts and the same canonical sum as the serial form, see alg.predict_cipher_weights)
        eye = np.zeros((len(mine), num_rows))
        eye[np.arange(len(mine)), mine] = 1
        masks = encoder.encode_many(list(eye), scale, parms_id=ev.ctx.first_parms_id() - 1)   # :222-227
        partial = ev.multiply_plain_sum(results, masks)[0]                          # :229, :233 (this rank's share)
    else:  # more ranks than rows: the zero ciphertext at the level / scale a masked dot product has
        one = alg.cipher_dot_product_many(ev, features[:1], [weights], num_weights, relin_keys, gal_keys)[0]
        mask_pt = encoder.encode(np.ones(1), scale)
        ev.mod_switch_to_next_inplace(mask_pt)
        ev.multiply_plain_inplace(one, mask_pt)
        partial = _zero_like(ev, one)
    lin = allreduce_ciphertext(ev, partial, group)
    ev.relinearize_inplace(lin, relin_keys)                                         # :237 (no-op)
    ev.rescale_to_next_inplace(lin)                                                 # :239
    lin.scale = 2.0 ** int(np.log2(lin.scale))                                      # :242
    coeffs = alg.SIGMOID_COEFFS[degree]                                             # :245-262
    return alg.horner_cipher(ev, encoder, encryptor, lin, len(coeffs) - 1, coeffs, scale, relin_keys)   # :264

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

from typing import List, Sequence

import numpy as np

from . import algorithms as alg
from .seal import Ciphertext, Evaluator, KSwitchKeys, Plaintext


def shard(n: int, rank: int, world: int) -> range:
    """contiguous block partition of range(n)"""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def allreduce_ciphertext(ev: Evaluator, ct: Ciphertext, group=None) -> Ciphertext:
    """sum of every rank's `ct` (same level/scale/size), bit-identical to a serial add_many"""
    import torch
    import torch.distributed as dist
    be, L, size, N = ev.be, ct.parms_id(), ct.size(), ev.ctx.N
    world = dist.get_world_size(group)
    if world > 8:
        raise ValueError("the wrap-free uint64 sum argument holds for at most 8 addends of < 2^61")
    if be.name == "hip":
        from .engine import DeviceArray
        eng = be.engine
        t = torch.empty(size * L * N, dtype=torch.int64, device=torch.device("cuda", eng.device))
        eng.copy_raw(t.data_ptr(), ct.data.ptr, t.numel() * 8)   # default stream, ordered before the collective
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        torch.cuda.current_stream().synchronize()
        out = DeviceArray(eng, (size, L, N))
        eng.copy_raw(out.ptr, t.data_ptr(), t.numel() * 8)
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

"""ctypes binding of the C-ABI in include/hefx.h (libhefx.so, gfx950 HIP kernels).

This is the only place Python touches the engine.  There is no CPU fallback: if libhefx.so is missing
or no HIP device is present, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _build

HEFX_OK = 0
HEFX_ERR_INVALID = -1
HEFX_ERR_HIP = -2
HEFX_ERR_UNSUPPORTED = -3
HEFX_ERR_TRANSPARENT = -4


class HefxError(RuntimeError):
    """HIP/device failure or unsupported parameter set."""


class TransparentCiphertextError(RuntimeError):
    """SEAL: std::logic_error("result ciphertext is transparent")."""


_lib = None

_vp, _u64, _u32, _i, _sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_size_t
_pp = C.POINTER(C.c_void_p)

_SIGS = {
    "hefx_last_error": (C.c_char_p, []),
    "hefx_version": (C.c_char_p, []),
    "hefx_device_count": (_i, []),
    "hefx_context_create": (_i, [_u32, C.POINTER(_u64), _i, _i, _pp]),
    "hefx_context_destroy": (None, [_vp]),
    "hefx_poly_degree": (_u32, [_vp]),
    "hefx_prime_count": (_i, [_vp]),
    "hefx_prime": (_u64, [_vp, _i]),
    "hefx_psi": (_u64, [_vp, _i]),
    "hefx_malloc": (_i, [_vp, _sz, _pp]),
    "hefx_free": (_i, [_vp, _vp]),
    "hefx_upload": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "hefx_download": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "hefx_copy": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "hefx_copy_peer": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "hefx_copy_peer_to": (_i, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "hefx_device_memory": (_i, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "hefx_ks_fallback_count": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "hefx_ks_stats": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "hefx_context_device": (_i, [_vp]),
    "hefx_memset_zero": (_i, [_vp, _vp, _sz, _vp]),
    "hefx_stream_sync": (_i, [_vp, _vp]),
    "hefx_ntt_forward": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "hefx_ntt_inverse": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "hefx_add": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "hefx_sub": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "hefx_negate": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "hefx_add_plain": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "hefx_add_many": (_i, [_vp, _i, _i, _i, _pp, _vp, _vp]),
    "hefx_multiply_plain": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "hefx_check_transparent": (_i, [_vp, _vp]),
    "hefx_multiply_plain_sum": (_i, [_vp, _i, _i, _i, _i, _pp, _pp, _pp, _vp]),
    "hefx_multiply": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "hefx_square": (_i, [_vp, _i, _vp, _vp, _vp]),
    "hefx_multiply_batch": (_i, [_vp, _i, _i, _pp, _pp, _pp, _vp]),
    "hefx_apply_galois": (_i, [_vp, _i, _vp, _u32, _vp, _vp, _vp]),
    "hefx_apply_galois_batch": (_i, [_vp, _i, _i, _pp, C.POINTER(_u32), _pp, _pp, _vp]),
    "hefx_rotate_multiply_plain_batch": (_i, [_vp, _i, _i, _pp, C.POINTER(_u32), _pp, _pp, _pp, _vp]),
    "hefx_apply_galois_add_batch": (_i, [_vp, _i, _i, _pp, C.POINTER(_u32), _pp, _pp, _pp, _pp, _vp]),
    "hefx_rotate_add_chain": (_i, [_vp, _i, _i, _pp, C.POINTER(_u32), _pp, _pp, _pp, _pp, _i, _vp]),
    "hefx_apply_galois_forest": (_i, [_vp, _i, _i, C.POINTER(C.c_int32), _pp, C.POINTER(_u32), _pp, _pp, _pp, _vp]),
    "hefx_relinearize": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "hefx_relinearize_batch": (_i, [_vp, _i, _i, _pp, _vp, _pp, _vp]),
    "hefx_rescale_to_next": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "hefx_rescale_to_next_mode": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "hefx_rescale_to_next_batch": (_i, [_vp, _i, _i, _i, _pp, _pp, _vp]),
    "hefx_add_batch": (_i, [_vp, _i, _i, _i, _pp, _pp, _pp, _vp]),
    "hefx_sub_batch": (_i, [_vp, _i, _i, _i, _pp, _pp, _pp, _vp]),
    "hefx_multiply_plain_batch": (_i, [_vp, _i, _i, _i, _pp, _pp, _pp, _vp]),
    "hefx_set_rescale_mode": (_i, [_vp, _i]),
    "hefx_get_rescale_mode": (_i, [_vp]),
    "hefx_mod_drop": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "hefx_reduce_canonical": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "hefx_comm_unique_id": (_i, [C.c_char_p]),
    "hefx_comm_init": (_i, [_vp, _i, _i, C.c_char_p]),
    "hefx_comm_destroy": (_i, [_vp]),
    "hefx_comm_world": (_i, [_vp]),
    "hefx_comm_rank": (_i, [_vp]),
    "hefx_allreduce_sum": (_i, [_vp, _i, _i, _vp, _vp]),
    "hefx_linear_transform_plain": (_i, [_vp, _i, _vp, _i, _pp, _i, C.POINTER(_u32), _pp, _vp, _vp]),
    "hefx_linear_transform_plain_many": (_i, [_vp, _i, _i, _pp, _i, _pp, _i, C.POINTER(_u32), _pp, _pp, _vp]),
    "hefx_rotate_hoisted_batch": (_i, [_vp, _i, _vp, _i, C.POINTER(_u32), _pp, _pp, _pp, _vp]),
    "hefx_linear_transform_plain_hoisted": (_i, [_vp, _i, _vp, _i, _pp, _i, C.POINTER(_u32), _pp, _vp, _vp]),
    "hefx_linear_transform_plain_hoisted2": (_i, [_vp, _i, _vp, _i, _pp, _i, C.POINTER(_u32), _pp, _vp, _vp]),
    "hefx_linear_transform_plain_hoisted2_sparse": (_i, [_vp, _i, _vp, _i, _i, C.POINTER(C.c_int), _pp, _i, C.POINTER(_u32),
                                                       _pp, _vp, _vp]),
    "hefx_linear_transform_plain_bsgs": (_i, [_vp, _i, _vp, _i, _i, _pp, _i, C.POINTER(_u32), _pp, _i, _vp, _vp]),
    "hefx_ckks_encode": (_i, [_vp, _i, _vp, _vp, _i, _i, C.c_double, _vp, _vp]),
    "hefx_ckks_encode_batch": (_i, [_vp, _i, _vp, _vp, _i, _i, C.c_double, _pp, _vp]),
    "hefx_sample_uniform": (_i, [_vp, C.c_char_p, _u64, _i, _i, _i, _vp, _vp]),
    "hefx_sample_ternary": (_i, [_vp, C.c_char_p, _u64, _i, _i, _i, _vp, _vp]),
    "hefx_sample_noise": (_i, [_vp, C.c_char_p, _u64, _i, _i, _i, _vp, _vp]),
    "hefx_keygen_kswitch": (_i, [_vp, _vp, _vp, C.c_char_p, _u64, _vp, _vp]),
    "hefx_galois_permute": (_i, [_vp, _u32, _vp, _i, _vp, _vp]),
    "hefx_encrypt": (_i, [_vp, _i, _vp, _vp, C.c_char_p, _u64, _vp, _vp]),
    "hefx_encrypt_batch": (_i, [_vp, _i, _i, _vp, _pp, C.c_char_p, _u64, _pp, _vp]),
    "hefx_decrypt": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "hefx_ckks_decode": (_i, [_vp, _i, _vp, _i, C.c_double, _vp, _vp, _vp]),
    "hefx_event_create": (_i, [_vp, _pp]),
    "hefx_event_destroy": (_i, [_vp, _vp]),
    "hefx_event_record": (_i, [_vp, _vp, _vp]),
    "hefx_event_elapsed_ms": (_i, [_vp, _vp, _vp, C.POINTER(C.c_float)]),
    "hefx_profile_begin": (_i, [_vp]),
    "hefx_profile_end": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(_u64)]),
    "hefx_profile_stage_name": (C.c_char_p, [_i]),
}

EXPORTED_SYMBOLS = sorted(_SIGS)


def library_path() -> str:
    # HEFX_LIB: development override to A/B alternative builds of the same ABI
    return os.environ.get("HEFX_LIB") or _build.SO


def lib():
    """Loads libhefx.so (must have been built: python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise HefxError(f"{path} is missing: build the HIP extension first (__graft_entry__.build())")
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(rc: int) -> None:
    if rc == HEFX_OK:
        return
    msg = lib().hefx_last_error().decode()
    if rc == HEFX_ERR_INVALID:
        raise ValueError(msg)  # SEAL: std::invalid_argument
    if rc == HEFX_ERR_TRANSPARENT:
        raise TransparentCiphertextError(msg)
    raise HefxError(f"hefx error {rc}: {msg}")


def ptr_array(ptrs):
    """C array of device pointers (None = NULL).  Built through numpy for long lists: the ctypes constructor walks its
    arguments one Python call at a time (68 us for 512 pointers against 19 us this way), which at 512 diagonals + 512 keys
    was a tenth of a direct-key linear transform's wall time.  A ctypes array is passed through unchanged."""
    if isinstance(ptrs, C.Array):
        return ptrs
    n = len(ptrs)
    if n < 32:
        return (C.c_void_p * n)(*[None if p is None else int(p) for p in ptrs])
    import numpy as np
    try:
        a = np.array(ptrs, dtype=np.uint64)
    except TypeError:  # a None among them
        a = np.array([0 if p is None else p for p in ptrs], dtype=np.uint64)
    arr = (C.c_void_p * n).from_buffer(a)
    arr._keep = a  # the numpy buffer owns the memory
    return arr


def u32_array(vals):
    if isinstance(vals, C.Array):
        return vals
    n = len(vals)
    if n < 32:
        return (C.c_uint32 * n)(*[int(v) for v in vals])
    import numpy as np
    a = np.array(vals, dtype=np.uint32)
    arr = (C.c_uint32 * n).from_buffer(a)
    arr._keep = a
    return arr

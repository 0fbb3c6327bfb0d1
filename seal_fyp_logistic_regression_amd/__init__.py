"""MI355X-native CKKS ciphertext-arithmetic engine behind the SEAL Evaluator API surface used by
MarwanNour/SEAL-FYP-Logistic-Regression (hot path: Linear_Transform_Plain, helper.h:237-262).

Layers: csrc/ (gfx950 HIP kernels + C-ABI, include/hefx.h) -> capi.py (ctypes) -> engine.py (device
buffers, one method per ABI call).  There is no CPU fallback anywhere in this package.
"""
from . import capi  # noqa: F401
from .engine import DeviceArray, Engine  # noqa: F401

__all__ = ["capi", "Engine", "DeviceArray"]

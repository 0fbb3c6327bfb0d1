"""CPU-only checks of the drop-in boundary: libhefx.so builds, loads, exports every symbol that
include/hefx.h declares, and fails loudly (no silent CPU fallback) when there is no HIP device."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from seal_fyp_logistic_regression_amd import _build, capi
    _build.build()
    return capi.lib()


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "hefx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hefx_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_all_exported(lib):
    from seal_fyp_logistic_regression_amd import capi
    syms = _header_symbols()
    assert len(syms) >= 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.library_path()], text=True)
    exported = set(re.findall(r" T (hefx_[a-z_0-9]+)", out))
    missing = [s for s in syms if s not in exported]
    assert not missing, missing
    # and the ctypes table binds exactly the header's surface
    assert sorted(capi.EXPORTED_SYMBOLS) == syms


def test_library_targets_gfx950_only(lib):
    from seal_fyp_logistic_regression_amd import capi
    data = open(capi.library_path(), "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx942", b"gfx90a", b"sm_80", b"sm_90"):
        assert other not in data


def test_no_device_fails_loudly(lib):
    import torch  # only to know whether a GPU is visible in this process
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from seal_fyp_logistic_regression_amd import Engine, capi
    assert lib.hefx_device_count() == 0
    with pytest.raises(capi.HefxError, match="no HIP device|HIP"):
        Engine(8192, [0xffffffffffe8001, 0xfffff4c001, 0xfffffdc001, 0xfffffffffffc001])


def test_argument_validation_without_device(lib):
    from seal_fyp_logistic_regression_amd import capi
    h = C.c_void_p()
    bad = (C.c_uint64 * 2)(97, 193)
    rc = lib.hefx_context_create(8192, bad, 2, 0, C.byref(h))
    assert rc == capi.HEFX_ERR_INVALID and b"1 mod 2N" in lib.hefx_last_error()
    rc = lib.hefx_context_create(8000, bad, 2, 0, C.byref(h))
    assert rc == capi.HEFX_ERR_INVALID
    rc = lib.hefx_context_create(65536, bad, 2, 0, C.byref(h))
    assert rc == capi.HEFX_ERR_UNSUPPORTED


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "seal_fyp_logistic_regression_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".cuh", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "ckks_oracle" not in text, f


def test_header_is_plain_c():
    """include/hefx.h is the C-ABI: it must compile as C99 on its own (no C++-isms, no HIP or torch types)."""
    import subprocess
    import tempfile
    src = '#include "hefx.h"\nint main(void) { return hefx_device_count() < 0; }\n'
    with tempfile.NamedTemporaryFile("w", suffix=".c", delete=False) as f:
        f.write(src)
        path = f.name
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I",
                        os.path.join(ROOT, "include"), path], capture_output=True, text=True)
    os.unlink(path)
    assert r.returncode == 0, r.stderr

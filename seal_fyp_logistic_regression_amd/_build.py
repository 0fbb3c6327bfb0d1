"""Builds csrc/ into libhefx.so (gfx950 only) with hipcc.  In-tree so that the .so travels with gpurun."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libhefx.so")
SOURCES = ["hefx_keyswitch.hip", "hefx_kernels.hip", "hefx_encode.hip", "hefx_sample.hip", "hefx_capi.cpp"]  # slowest first
HEADERS = ["hefx_internal.h", "hefx_modarith.cuh", "hefx_ntt.cuh", "hefx_ntt8.cuh", "../../include/hefx.h"]
DEPS = SOURCES + HEADERS  # a change in any of them rebuilds the library (a header: every object; a source: its object)


def source_sha16() -> str:
    """sha256[:16] over the engine's sources (csrc/ + the public header), in the fixed order of DEPS: what bench.py writes
    into its JSON line and the profile tools into profiles/*.json, so that counter files measured on other kernels are seen."""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(DEPS):
        h.update(d.encode())
        h.update(open(os.path.join(CSRC, d), "rb").read())
    return h.hexdigest()[:16]


def library_sha16() -> str:
    """sha256[:16] of the built libhefx.so ("" when it does not exist)"""
    import hashlib
    if not os.path.exists(SO):
        return ""
    return hashlib.sha256(open(SO, "rb").read()).hexdigest()[:16]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return SO
    objs, jobs = [], []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        # the public header is only seen by the C-ABI translation unit (a doc edit there must not cost four minutes of kernels)
        hdrs = [h for h in HEADERS if src == "hefx_capi.cpp" or not h.endswith("hefx.h")]
        hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in hdrs)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, src))):
            continue  # this object is newer than its source and every header it includes
        # -pragma-unroll-threshold: the transforms are written as fully unrolled loops over register arrays; the inline
        # asm statements of hefx_modarith.cuh count as large in the unroller's size estimate and push the inverse
        # transforms past the default threshold (loops left rolled -> the register arrays go to scratch memory)
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-mllvm", "-pragma-unroll-threshold=1048576", "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((src, subprocess.Popen(cmd)))  # the five translation units compile side by side
    failed = [(src, p.returncode) for src, p in jobs if p.wait() != 0]  # every job is waited for before anything is raised
    if failed:
        raise subprocess.CalledProcessError(failed[0][1], "hipcc " + ", ".join(src for src, _ in failed))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))

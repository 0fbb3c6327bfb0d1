"""Host-side profiling of the C++ shim WITHOUT a GPU: builds build/stub/libhefx.so, a stand-in whose every C-ABI entry
(parsed from include/hefx.h) returns HEFX_OK at once (hefx_malloc hands out 64 host bytes, hefx_download fills ones), and
build/stub/probe = drivers/lt_host_probe.cpp linked against it.  What remains when it runs is the recorder's own work:
    python tools/make_stub_libhefx.py && build/stub/probe 1000 30
    (add -pg by hand and run gprof for a profile; this is how recording a rotation went from 3.7 to 0.4 us in round 4)
Development tool only: nothing in the product or the tests links the stub."""
import os, re, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "build", "stub")
os.makedirs(out, exist_ok=True)
h = open(os.path.join(root, "include", "hefx.h")).read()
h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
h = re.sub(r"//.*", "", h)
protos = re.findall(r"\n\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_ \*]*?\b(hefx_[a-z0-9_]+)\s*\(([^;{]*?)\))\s*;", h, flags=re.S)
src = ['#include "hefx.h"', "#include <cstdlib>", "#include <cstring>", 'extern "C" {']
special = {
    "hefx_malloc": "*d_ptr = std::calloc(1, 64); return 0;",
    "hefx_free": "std::free(d_ptr); return 0;",
    "hefx_last_error": 'return "";',
    "hefx_context_create": "*out = (hefx_context *)std::calloc(1, 64); return 0;",
    "hefx_download": "std::memset(h_dst, 1, bytes); return 0;",
    "hefx_device_memory": "if (free_bytes) *free_bytes = (size_t)200 << 30; if (total_bytes) *total_bytes = (size_t)288 << 30; return 0;",
}
seen = set()
for full, name, _ in protos:
    if name in seen:
        continue
    seen.add(name)
    ret = full[: full.index(name)].strip()
    body = special.get(name) or ("" if ret == "void" else ("return nullptr;" if "*" in ret else "return 0;"))
    src.append(f"{full} {{ {body} }}")
src.append("}")
open(os.path.join(out, "stub.cpp"), "w").write("\n".join(src) + "\n")
inc = os.path.join(root, "include")
subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I" + inc, os.path.join(out, "stub.cpp"), "-o", os.path.join(out, "libhefx.so")])
# --sanitize: the probe (i.e. include/seal/seal.h's host side) under AddressSanitizer + UBSan, for tests/test_shim_host_cpu.py
san = ["-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"] if "--sanitize" in sys.argv[1:] else ["-O2"]
subprocess.check_call(["g++"] + san + ["-g", "-std=c++17", "-w", "-I" + inc, os.path.join(root, "drivers", "lt_host_probe.cpp"), "-o",
                       os.path.join(out, "probe"), "-L" + out, "-lhefx", "-Wl,-rpath,$ORIGIN"])
print(f"{len(seen)} entries stubbed; run {os.path.join(out, 'probe')} [d=1000] [reps=5]")

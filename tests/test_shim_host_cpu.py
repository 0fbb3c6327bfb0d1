"""The C++ shim's recorder without a GPU: drivers/lt_host_probe.cpp (the reference's Linear_Transform_Plain loop, helper.h:237-262,
piece by piece) linked against a STUB libhefx whose every entry returns at once (tools/make_stub_libhefx.py).  Nothing is
computed -- this is a compile-and-run check of include/seal/seal.h's host side and a guard on what recording costs: the
loop over 1000 rotate_vector + multiply_plain calls took 3.7 ms of host time until round 4 found 3^pos mod 2N computed by
pos multiplications inside it; it takes 0.4-0.8 ms now."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_recording_the_unchanged_linear_transform_loop_is_cheap():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_stub_libhefx.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    probe = os.path.join(ROOT, "build", "stub", "probe")
    try:
        r = subprocess.run([probe, "1000", "7"], capture_output=True, text=True, timeout=300)
    finally:  # the stand-in library never outlives the test (it is named like the real one)
        import shutil
        shutil.rmtree(os.path.join(ROOT, "build", "stub"), ignore_errors=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rec = sorted(int(m.group(1)) for m in re.finditer(r"record loop (\d+)", r.stdout))
    sub = sorted(int(m.group(1)) for m in re.finditer(r"add_many (\d+)", r.stdout))
    assert len(rec) == 7
    # medians, with a wide margin for a loaded container (measured here: ~700 and ~400 us)
    assert rec[3] < 3000, f"recording 1000 rotations + products: {rec} us"
    assert sub[3] < 3000, f"planning and submitting them: {sub} us"


def test_recorder_and_planner_are_clean_under_asan_and_ubsan():
    """the same probe built with -fsanitize=address,undefined: recording, fusing and planning 200 rotations + products (and the
    destructors of everything recorded) touch no freed or foreign memory.  The stub hands out 64-byte host blocks, so a shim
    that dereferenced a 'device' pointer on the host would be caught here too."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_stub_libhefx.py"), "--sanitize"], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    probe = os.path.join(ROOT, "build", "stub", "probe")
    try:
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
        r = subprocess.run([probe, "200", "3"], capture_output=True, text=True, timeout=600, env=env)
    finally:
        import shutil
        shutil.rmtree(os.path.join(ROOT, "build", "stub"), ignore_errors=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-4000:]


def test_sha3_parms_id_and_parameter_stream_against_hashlib(tmp_path):
    """include/seal/shim_io.h without a GPU: drivers/serial_host_probe.cpp (plain g++, no libhefx -- nothing in it touches an
    engine) prints SHA3-256 of its argument, the SEAL-style parms_id of config 2's parameter set and the bytes of
    EncryptionParameters::Save; Python's hashlib and struct say what they must be."""
    import hashlib
    import struct
    exe = str(tmp_path / "serial_host_probe")
    # (built with ASan + UBSan: the stream readers and writers of seal.h run instrumented; a report fails the exit code)
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-w",
                        "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "drivers", "serial_host_probe.cpp"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    primes = [0xffffffffffe8001, 0xfffff4c001, 0xfffffdc001, 0xfffffffffffc001]
    for msg in ("", "abc", "x" * 135, "y" * 136, "z" * 500):   # empty, short, one byte under / exactly / several times the rate
        out = subprocess.run([exe, msg], capture_output=True, text=True, timeout=60).stdout.split("\n")
        assert out[0] == "sha3 " + hashlib.sha3_256(msg.encode()).hexdigest()
    words = [2, 8192] + primes + [0]
    want_id = struct.unpack("<4Q", hashlib.sha3_256(struct.pack("<7Q", *words)).digest())
    _, a, b, c, d, stream = out[1].split()
    assert tuple(int(x, 16) for x in (a, b, c, d)) == want_id
    assert bytes.fromhex(stream) == struct.pack("<BQQ", 2, 8192, 4) + struct.pack("<4Q", *primes) + struct.pack("<Q", 0)
    assert out[2] == "roundtrip 1"

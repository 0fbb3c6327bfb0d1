"""Round-6 GPU tests.

* hefx_context_destroy returns the scratch / workspace buffers that grow_retiring() retired; HEFX_CHUNK=1024 with
  every item rotating ONE source (the one-source hoisting shortcut must leave room for its source descriptor in the ring
  slot).
* The stand-alone row transform (ntt_rows_kernel, under encode / encrypt / keygen / decrypt) after the round-6 register
  fix: forward and inverse at every degree against the oracle's transform, both arithmetic policies in one call.
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C2 = (8192, [0xffffffffffe8001, 0xfffff4c001, 0xfffffdc001, 0xfffffffffffc001])


def _free_bytes(e):
    from seal_fyp_logistic_regression_amd import capi
    f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
    capi.check(capi.lib().hefx_device_memory(e._h, ctypes.byref(f), ctypes.byref(t)))
    return f.value


def test_context_destroy_returns_retired_workspaces():
    """Contexts whose key-switch scratch grew several times (batches of 1, 40, 300 items: each growth RETIRES the outgrown
    buffer instead of freeing it, hefx_capi.cpp grow_retiring) are destroyed; the device's free memory, read through a
    long-lived probe context, must come back to where it started.  Until round 6 the retired buffers leaked."""
    from seal_fyp_logistic_regression_amd import Engine
    N, primes = C2
    L, k = 3, len(primes)
    probe = Engine(N, primes)
    key32 = bytes(range(32))

    def one_life():
        e = Engine(N, primes)
        key = e.sample('uniform', key32, 1, 2 * L, k, 0)
        for n in (1, 40, 300):
            ct = e.sample('uniform', key32, 2, 2 * n, L, 0)
            pt = e.sample('uniform', key32, 3, n, L, 0)
            cts = [ct.view(i * 2 * L * N, (2, L, N)) for i in range(n)]
            pts = [pt.view(i * L * N, (L, N)) for i in range(n)]
            outs = e.rotate_multiply_plain_batch(L, cts, [3] * n, [key] * n, pts)
            outs[-1].download()
            del outs, cts, pts, ct, pt
        del key
        e.close()

    one_life()  # first life: code objects, allocator arenas of the runtime itself
    start = _free_bytes(probe)
    for _ in range(3):
        one_life()
    end = _free_bytes(probe)
    # three lives leaked ~0.5 GB before the fix (the retired 64 MiB / 128 MiB / ... ladders); allow the runtime 32 MiB of its own
    assert start - end < (32 << 20), (start, end, (start - end) >> 20)
    probe.close()


def test_chunk_1024_one_source_leaves_room_for_the_source_descriptor():
    """HEFX_CHUNK=1024 (the ring slot's size) and 1024 rotations of ONE ciphertext: the one-source shortcut would have put
    its source descriptor one KsItem past the slot; now such a chunk takes the unhoisted sequence.  A sample
    of outputs word for word against the oracle, and a second call with 1023 items (which does hoist)."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from oracle import oracle as O\n"
        "from seal_fyp_logistic_regression_amd import Engine\n"
        "N = 2048\n"
        "from seal_fyp_logistic_regression_amd import seal as S\n"
        "primes = [int(p) for p in S.CoeffModulus.Create(N, [50, 40, 50])]\n"
        "o, e = O.Oracle(N, primes), Engine(N, primes); L, k = 2, 3\n"
        "key = e.sample('uniform', bytes(range(32)), 1, 2 * L, k, 0)\n"
        "hk = key.download().reshape(L, 2, k, N)\n"
        "ct = e.sample('uniform', bytes(range(32)), 2, 2, L, 0).view(0, (2, L, N)); hct = ct.download()\n"
        "want = o.apply_galois(hct, 3, hk)\n"
        "ok = True\n"
        "for n in (1024, 1023):\n"
        "    outs = e.apply_galois_batch(L, [ct] * n, [3] * n, [key] * n)\n"
        "    ok = ok and all(bool((outs[i].download() == want).all()) for i in (0, 1, 511, 512, n - 2, n - 1))\n"
        "print('PARITY', ok)\n") % (ROOT,)
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "HEFX_CHUNK": "1024"}, capture_output=True, text=True,
                       timeout=600)
    assert "PARITY True" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("N", [1024, 2048, 4096, 8192, 16384, 32768])
def test_row_transforms_every_degree_both_policies(N):
    """hefx_ntt_forward / hefx_ntt_inverse (ntt_rows_kernel; the split kernels at N = 32768) over a modulus chain that mixes
    the two arithmetic policies (40-bit primes: FP64; 50/60-bit: integer): forward == the oracle's transform word for word,
    inverse(forward(x)) == x, on 3 polynomials x 4 rows including the all-(q-1) and all-zero rows."""
    from oracle import oracle as O
    from seal_fyp_logistic_regression_amd import Engine, seal as S
    primes = [int(p) for p in S.CoeffModulus.Create(N, [60, 40, 50, 40])]
    o, e = O.Oracle(N, primes), Engine(N, primes)
    x = o.uniform(4, 3, 1234 + N)  # [3, 4, N]
    for j in range(4):
        x[1, j, :] = primes[j] - 1
    x[2, 0, :] = 0
    d = e.to_device(x)
    e.ntt_forward(d, 3, 4)
    got = d.download().reshape(3, 4, N)
    want = np.stack([np.stack([o.ntt_fwd(j, x[p, j]) for j in range(4)]) for p in range(3)])
    assert (got == want).all()
    e.ntt_inverse(d, 3, 4)
    assert (d.download().reshape(3, 4, N) == x).all()
    e.close()


def test_cc_matrix_multiplication_n8_config5_dense_bit_exact(rescale_mode):
    """BASELINE config 5 as SURVEY App. B reads it -- "64 x 64" are the U matrices, i.e. n = 8, exactly as config 3's
    "16 x 16" is n = 4 -- in the reference's EXACT composition (matrix_mult_benchmark.cpp:13-71: all 64 diagonals of every
    U_sigma / U_tau / V_k / W_k with the +1e-8 epsilons of :239-297, 1024 plaintexts = 1.34 GB, ~630 key switches with the
    default power-of-two Galois keys) at N = 32768 {60,40,40,40,40,60}: HIP engine and oracle twin bit for bit in both
    rescale divisions, and the result decrypts to A.B."""
    from seal_fyp_logistic_regression_amd import algorithms as alg
    from tests.test_gpu_composites import both, bits, decode
    n = 8
    rng = np.random.default_rng(58)
    A, B = rng.uniform(-1, 1, (n, n)), rng.uniform(-1, 1, (n, n))

    def run(e):
        scale = 2.0 ** 40
        enc = lambda U: [e["encoder"].encode(dg + 1e-8, scale) for dg in alg.get_all_diagonals(U)]  # epsilon: :239-297
        Us, Ut, V, W = alg.matmul_permutation_matrices(n)      # helper.h:702-851
        ctA = e["enc"].encrypt(e["encoder"].encode(A.reshape(-1), scale))
        ctB = e["enc"].encrypt(e["encoder"].encode(B.reshape(-1), scale))
        return alg.cc_matrix_multiplication(e["ev"], ctA, ctB, n, enc(Us), enc(Ut), [enc(v) for v in V],
                                            [enc(w) for w in W], e["gk"])

    r = both(32768, [60, 40, 40, 40, 40, 60], run)
    (eg, cg), (eo, co) = r["gpu"], r["oracle"]
    assert cg.size() == 3 and cg.parms_id() == co.parms_id() == 4 and cg.scale == co.scale
    assert (bits(eg, cg) == bits(eo, co)).all()
    got = decode(eg, cg, n * n).reshape(n, n)
    assert np.allclose(got, A @ B, rtol=1e-4, atol=1e-3), np.abs(got - A @ B).max()


@pytest.mark.parametrize("setname,d,count,direct", [("C2", 9, 3, False), ("C3", 16, 2, False), ("C3", 40, 2, True), ("C2", 5, 5, False),
                                                   ("C2", 1, 2, False), ("C2", 2, 3, False), ("C2", 100, 2, False)])
def test_linear_transform_plain_many_bit_exact(setname, d, count, direct):
    """hefx_linear_transform_plain_many: `count` independent Linear_Transform_Plain calls in lockstep (the sigma / tau
    transforms of CC_Matrix_Multiplication, matrix_multiplication.cpp:22-25) -- every output word for word what the
    single-transform entry gives for that input, and what the oracle twin's op-by-op sequence gives; with the reference's
    default keys (NAF forests) and with a direct key per step (the wide depth runs exactly hoisted per source); d = 1 (no
    rotation below ct_new), d = 2, and d = 100 (beyond 96 diagonals the first product is a launch of its own and the final sum
    goes through its table level)."""
    from seal_fyp_logistic_regression_amd import algorithms as alg
    from tests.test_gpu_composites import make, bits, decode
    N, bits_ = {"C2": (8192, [60, 40, 40, 60]), "C3": (16384, [60, 40, 40, 40, 40, 60])}[setname]
    steps = ([-d] + list(range(1, d))) if direct else None
    res = {}
    rng = np.random.default_rng(d * 100 + count)
    Ms = [rng.uniform(-1, 1, (d, d)) for _ in range(count)]
    vs = [rng.uniform(-1, 1, d) for _ in range(count)]
    for kind in ("gpu", "oracle"):
        e = make(N, bits_, kind, seed=11, galois_steps=steps)
        scale = 2.0 ** 40
        cts = [e["enc"].encrypt(e["encoder"].encode(v, scale)) for v in vs]
        diag_sets = [[e["encoder"].encode(x, scale) for x in alg.get_all_diagonals(M)] for M in Ms]
        many = alg.linear_transforms_plain_many(e["ev"], cts, diag_sets, e["gk"])
        one_by_one = [alg.linear_transform_plain(e["ev"], c, ds, e["gk"]) for c, ds in zip(cts, diag_sets)]
        res[kind] = (e, many, one_by_one)
    (eg, mg, og), (eo, mo, oo) = res["gpu"], res["oracle"]
    assert eg["ctx"].backend.name == "hip" and hasattr(eg["ctx"].backend, "linear_transform_plain_many")
    for t in range(count):
        assert mg[t].scale == og[t].scale == mo[t].scale and mg[t].parms_id() == mo[t].parms_id()
        assert (bits(eg, mg[t]) == bits(eg, og[t])).all(), t          # lockstep == one call per transform
        assert (bits(eg, mg[t]) == bits(eo, mo[t])).all(), t          # == the oracle twin's op-by-op sequence
        assert np.allclose(decode(eg, mg[t], d), Ms[t] @ vs[t], atol=1e-3)


def test_linear_transform_plain_many_refuses_bad_arguments():
    """count out of range, null pointers, a missing Galois key: HEFX_ERR_INVALID before anything runs"""
    from seal_fyp_logistic_regression_amd import algorithms as alg, capi
    from tests.test_gpu_composites import make
    e = make(8192, [60, 40, 40, 60], "gpu", seed=5, galois_steps=[1, 2])
    be, L = e["ctx"].backend, 3
    scale = 2.0 ** 40
    cts = [e["enc"].encrypt(e["encoder"].encode(np.ones(4), scale)) for _ in range(2)]
    diags = [e["encoder"].encode(np.ones(4), scale) for _ in range(8)]
    elts = sorted(e["gk"].keys)
    keys = [e["gk"].key(x) for x in elts]
    with pytest.raises(ValueError, match="Galois key not present"):   # -4 has no key and is a single NAF term (SEAL: invalid_argument)
        be.linear_transform_plain_many(L, [c.data for c in cts], [p.data for p in diags], elts, keys)
    with pytest.raises(ValueError):
        be.linear_transform_plain_many(L, [], [], elts, keys)
    with pytest.raises(ValueError, match="bad linear-transform arguments"):   # more than 64 transforms in one call
        be.linear_transform_plain_many(L, [cts[0].data] * 65, [diags[0].data] * 65, elts, keys)


def test_bench_eight_ranks_share_one_gpu_sharded_legs_keep_the_serial_bits():
    """The driver's scaling run goes to eight ranks; this box has one MI355X, so the eight ranks share it (gloo rendezvous
    and exchange, as in tests/test_gpu_round4.py's two-rank run) with workloads cut to fit: what is exercised is every
    sharded leg's PARTITION at world 8 on the HIP engine -- a 16-diagonal transform (2 diagonals per rank), the n = 4 matrix
    product (6 Step-2 transforms over 8 ranks: two ranks own none), a five-row prediction (three ranks own no row) -- each
    with the bits of its serial form.  (The 2000-row leg at this world size needs the eight devices: on one GPU the ranks'
    allocations contend for seconds per slab.)"""
    from tests.test_gpu_round4 import _bench
    r, _ = _bench({}, "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "256", "--cpu-seconds", "0", "--lt", "16",
                  "--ladder", "", "--composites", "matmul_C3_n4,lr_rows_5x8", "--secondary", "", "--sustain", "0",
                  "--key-per-item", "0", "--lt-direct", "0", timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["verified"] is True and line["value"] > 0
    d16 = line["lt_sharded"]["d16"]
    assert d16["bits_equal_serial"] is True and d16["decrypts_to_Mv"] is True
    comp = line["composites"]
    assert set(comp) == {"matmul_C3_n4", "lr_rows_5x8"}, comp
    for name, rec in comp.items():
        assert rec["bits_equal_serial"] is True and rec["sharded_ms"] > 0, (name, rec)
    assert comp["matmul_C3_n4"]["decrypts_to_AB"] is True and comp["lr_rows_5x8"]["decrypts_to_sigmoid_of_Xw"] is True

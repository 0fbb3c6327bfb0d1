"""Round-4 GPU tests: the multi-rank bench path as the driver's SCALE run executes it, the fused
rotate + accumulate key switch and the chain entry point behind the reference's rotate-by-1 loops (helper.h:472-476),
the batched encode / encrypt submissions of the C++ shim."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C3 = (16384, [0xffffffffffd8001, 0xffffb20001, 0xffffc40001, 0xffffca8001, 0xffffe80001, 0xffffffffffe8001])
C2 = (8192, [0xffffffffffe8001, 0xfffff4c001, 0xfffffdc001, 0xfffffffffffc001])
C4 = (16384, [0xffffffffffd8001, 0xffff940001, 0xffffa78001, 0xffffaf8001, 0xffffb20001, 0xffffc40001, 0xffffca8001,
              0xffffe80001, 0xffffffffffe8001])
C5 = (32768, [0xfffffffff840001, 0xffff940001, 0xffffb20001, 0xffffc40001, 0xffffe80001, 0xffffffffffc0001])
SETS = {"C2": C2, "C3": C3, "C4": C4, "C5": C5}


def _edge61(N=4096, count=4):
    """the largest primes hefx_context_create admits (61 bits; SEAL's stop at 60): the 128-bit accumulator policy of the key MAC"""
    from seal_fyp_logistic_regression_amd.seal import _is_prime
    primes, v = [], (1 << 61) - 2 * N + 1
    while len(primes) < count:
        if _is_prime(v):
            primes.append(v)
        v -= 2 * N
    return N, primes


def _bench(extra_env, *flags, timeout=900):
    env = dict(os.environ, HEFX_BENCH_BACKEND="gloo", **extra_env)
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)  # the bench launches its own ranks
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    return r, time.monotonic() - t0


def test_bench_two_ranks_one_json_line_and_sharded_bits():
    """`bench.py --gpus 2` end to end on this box's one MI355X (two ranks share it; rendezvous and the exchange of the
    sharded leg over gloo, because RCCL wants a device per rank): rc 0, exactly ONE line on stdout and it is the JSON,
    n_gpus == 2, and the diagonal-sharded Linear_Transform_Plain has the bits of the serial one."""
    r, _ = _bench({}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "512", "--cpu-seconds", "0", "--lt", "16",
                  "--secondary", "C2", "--sustain", "0.3", "--key-per-item", "0", "--lt-direct", "0")  # (secondary: single-rank runs only)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["value"] > 0 and line["verified"] is True
    d16 = line["lt_sharded"]["d16"]
    assert d16["bits_equal_serial"] is True and d16["decrypts_to_Mv"] is True
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
    # round 5: the longer second pass and the library hashes ride in the same line; the other parameter sets only in
    # single-rank runs (tests/test_gpu_round5.py)
    assert line["secondary"] == {}
    assert line["sustained"]["steps"] >= 2 and line["sustained"]["value"] > 0
    assert len(line["libhefx_sha16"]) == 16 and len(line["csrc_sha16"]) == 16 and line["rescale_mode"] in ("floor", "round")
    # round 6: SURVEY 8(d)'s scaling workloads ride in the same line -- serial and sharded, the sharded bits asserted equal
    comp = line["composites"]
    assert set(comp) == {"matmul_C3_n4", "matmul_C5_n8", "lr_rows_2000x8"}, comp
    for name, rec in comp.items():
        assert rec["bits_equal_serial"] is True and rec["serial_ms"] > 0 and rec["sharded_ms"] > 0, (name, rec)
    assert comp["matmul_C3_n4"]["decrypts_to_AB"] is True and comp["matmul_C5_n8"]["decrypts_to_AB"] is True
    assert comp["lr_rows_2000x8"]["decrypts_to_sigmoid_of_Xw"] is True
    assert [r["batch"] for r in line["batch_ladder"]] == [1, 16, 256]   # (--batch 512: the ladder stops below it)
    assert line["chain_level_us"]["n1"] > 0 and line["chain_level_us"]["n8"] > 0


def test_bench_two_ranks_a_dead_rank_ends_the_run_non_zero():
    """Rank 1 dies right after the rendezvous (HEFX_BENCH_FAIL_RANK, a test hook in bench.py): rank 0 is then blocked in
    its first barrier for ever -- the launcher must notice, end it after its grace period and exit non-zero, well inside
    the time a driver would wait."""
    r, dt = _bench({"HEFX_BENCH_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "512",
                   "--cpu-seconds", "0", "--lt", "16", "--secondary", "", "--sustain", "0", timeout=300)
    assert r.returncode != 0
    assert dt < 180, f"the launcher took {dt:.0f} s to give up on a dead rank"
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")], "no result line from a failed run"


# ------------------------------------------------------------------------------------------------------------------
# rotate + accumulate in one key switch, and the chain of them (helper.h:472-476)
# ------------------------------------------------------------------------------------------------------------------
def _engine_and_oracle(N, primes):
    from oracle import oracle as O
    from seal_fyp_logistic_regression_amd import Engine
    return O.Oracle(N, primes), Engine(N, primes)


def _key(o, seed):
    return o.uniform(o.k, 2 * (o.k - 1), seed).reshape(o.k - 1, 2, o.k, o.N)


@pytest.mark.parametrize("setname,L,n", [("C3", 5, 1), ("C3", 2, 8), ("C3", 5, 40), ("C2", 3, 7), ("C3", 3, 300),
                                         ("C4", 8, 3), ("C4", 7, 36), ("C5", 5, 2), ("C5", 5, 34)])
def test_apply_galois_add_batch_bit_exact(setname, L, n):
    """hefx_apply_galois_add_batch against the oracle's op-by-op sequence rotate (apply_galois) then add: the rotation AND
    the sum, word for word.  n = 1 and 8: the small-batch path (descriptors in the kernel arguments, quarter-row kernels);
    40: one split-2 chunk; 300: two chunks on the two internal streams, three keys (key-grouped order).  Separate output
    sums first, then the in-place form (acc_out == acc_in) on top of the in-place rotation (ct_out == ct_in)."""
    N, primes = SETS[setname]
    o, e = _engine_and_oracle(N, primes)
    rng = np.random.default_rng(100 * L + n)
    keys = [_key(o, 31 + i) for i in range(3)]
    dkeys = [e.to_device(k) for k in keys]
    if n <= 40:
        cts = [o.uniform(L, 2, 1000 + i) for i in range(n)]
        accs = [o.uniform(L, 2, 5000 + i) for i in range(n)]
    else:  # device-drawn inputs keep the host side of the large case short
        big = e.sample("uniform", bytes(range(32)), 9, 4 * n, L, 0).download().reshape(2 * n, 2, L, N)
        cts, accs = list(big[:n]), list(big[n:])
    elts = [int(2 * rng.integers(1, N) + 1) for _ in range(n)]
    ki = [int(rng.integers(3)) for _ in range(n)]
    dct, dacc = [e.to_device(c) for c in cts], [e.to_device(a) for a in accs]
    outs, sums = e.apply_galois_add_batch(L, dct, elts, [dkeys[j] for j in ki], dacc)
    check = range(n) if n <= 40 else sorted({0, 1, 127, 128, 255, 256, 257, 299} | set(int(x) for x in rng.integers(0, n, 12)))
    want_rot = {i: o.apply_galois(cts[i], elts[i], keys[ki[i]]) for i in check}
    for i in check:
        assert (outs[i].download() == want_rot[i]).all(), ("rotation", i)
        assert (sums[i].download() == o.add(accs[i], want_rot[i])).all(), ("sum", i)
        assert (dacc[i].download() == accs[i]).all() and (dct[i].download() == cts[i]).all(), ("inputs untouched", i)
    if n <= 40:  # in place: t = rot(t); a += t, twice
        for _ in range(2):
            e.apply_galois_add_batch(L, dct, elts, [dkeys[j] for j in ki], dacc, outs=dct, acc_outs=dacc)
        for i in check:
            t1 = want_rot[i]
            t2 = o.apply_galois(t1, elts[i], keys[ki[i]])
            assert (dct[i].download() == t2).all(), ("in-place rotation", i)
            assert (dacc[i].download() == o.add(o.add(accs[i], t1), t2)).all(), ("in-place sum", i)


@pytest.mark.parametrize("zero", [False, True])
def test_apply_galois_add_batch_of_one_source_runs_hoisted_bit_exact(zero):
    """The accumulate form on a batch the engine hoists (48 rotations of 2 ciphertexts, in-place sums): rotation and sum
    word for word against rotate-then-add on the oracle; with a planted zero coefficient the chunk takes the per-item
    fallback, whose epilogue accumulates just the same -- and exactly once (the hoisted epilogue is gated off)."""
    N, primes = C3
    o, e = _engine_and_oracle(N, primes)
    L, n = 4, 48
    rng = np.random.default_rng(48 + zero)
    keys = [_key(o, 61 + i) for i in range(2)]
    dkeys = [e.to_device(k) for k in keys]
    srcs = [o.uniform(L, 2, 700 + i) for i in range(2)]
    if zero:
        srcs[0] = _plant_zero_coefficients(o, srcs[0], L, rng, [1], 2)
    dsrcs = [e.to_device(c) for c in srcs]
    accs = [o.uniform(L, 2, 800 + i) for i in range(n)]
    dacc = [e.to_device(a) for a in accs]
    si = [int(rng.integers(2)) for _ in range(n)]
    elts = [int(2 * rng.integers(1, N) + 1) for _ in range(n)]
    ki = [int(rng.integers(2)) for _ in range(n)]
    before = e.ks_fallback_count()
    outs, sums = e.apply_galois_add_batch(L, [dsrcs[s] for s in si], elts, [dkeys[k] for k in ki], dacc, acc_outs=dacc)
    assert (e.ks_fallback_count() - before) == (1 if zero else 0)
    for i in range(n):
        rot = o.apply_galois(srcs[si[i]], elts[i], keys[ki[i]])
        assert (outs[i].download() == rot).all(), ("rotation", i)
        assert (dacc[i].download() == o.add(accs[i], rot)).all(), ("sum", i)


@pytest.mark.parametrize("setname,L,n,nsrc", [("C3", 5, 512, 1), ("C3", 3, 300, 7), ("C4", 8, 256, 2), ("C4", 5, 200, 50),
                                              ("C5", 5, 160, 1), ("C2", 3, 1100, 3), ("C2", 1, 400, 1)])
def test_hoisted_equals_per_item_on_the_device_every_output(setname, L, n, nsrc):
    """Full-size cross-check without the oracle's cost: the same rotations once as ONE batch (hoisted: the engine decomposes
    each source once per chunk) and once in slices of 32 items (the per-item kernels) -- every word of every output equal,
    with the fused plaintext product, eight keys, random elements, at top and lower levels.  (The oracle-backed test above
    pins both against SEAL's sequence on samples; this one compares all n outputs.)"""
    N, primes = SETS[setname]
    o, e = _engine_and_oracle(N, primes)
    rng = np.random.default_rng(1000 * L + n)
    dkeys = [e.sample("uniform", bytes([i] * 32), 1, 2 * (o.k - 1), o.k, 0) for i in range(8)]
    dsrcs = [e.sample("uniform", bytes([100 + i] * 32), 2, 2, L, 0) for i in range(nsrc)]
    dpts = [e.sample("uniform", bytes([200 + i] * 32), 3, 1, L, 0) for i in range(4)]
    si = [int(rng.integers(nsrc)) for _ in range(n)]
    elts = [int(2 * rng.integers(1, N) + 1) for _ in range(n)]
    ki = [int(rng.integers(8)) for _ in range(n)]
    args = lambda idx: (L, [dsrcs[si[i]] for i in idx], [elts[i] for i in idx], [dkeys[ki[i]] for i in idx])
    before = e.ks_fallback_count()
    one = e.rotate_multiply_plain_batch(*args(range(n)), [dpts[i % 4] for i in range(n)])
    plain = e.apply_galois_batch(*args(range(n)))
    assert e.ks_fallback_count() == before
    for lo in range(0, n, 32):
        idx = range(lo, min(n, lo + 32))
        a = e.rotate_multiply_plain_batch(*args(idx), [dpts[i % 4] for i in idx])
        b = e.apply_galois_batch(*args(idx))
        for j, i in enumerate(idx):
            assert (a[j].download() == one[i].download()).all(), ("fused", i)
            assert (b[j].download() == plain[i].download()).all(), ("plain", i)


def test_profile_session_over_a_hoisted_batch():
    """hefx_profile_begin/end around a batch the engine hoists: the per-launch-kind report has HEFX_PROFILE_STAGES entries,
    the exact MAC is booked under the MAC stage, the (empty) fallback launches under their own, and the results are still
    the oracle's (a profile session runs the chunks serially on the caller's stream)."""
    N, primes = C3
    o, e = _engine_and_oracle(N, primes)
    L, n = 5, 64
    key = _key(o, 77)
    dk = e.to_device(key)
    src = o.uniform(L, 2, 4242)
    dsrc = e.to_device(src)
    elts = [2 * i + 3 for i in range(n)]
    e.apply_galois_batch(L, [dsrc] * n, elts, [dk] * n)  # first use: tables
    e.profile_begin()
    outs = e.apply_galois_batch(L, [dsrc] * n, elts, [dk] * n)
    stages, chunks = e.profile_end()
    assert chunks == 1 and len(stages) == 8 and "gated fallback launches" in stages
    assert stages["ks_mac_kernel"] > 0 and stages["ks_moddown_finish_kernel"] > 0
    assert 0 < stages["gated fallback launches"] < 0.5 * sum(stages.values()), stages  # five empty launches, not a second key switch
    for i in (0, 17, 63):
        assert (outs[i].download() == o.apply_galois(src, elts[i], key)).all()


def test_apply_galois_add_batch_refuses_overlapping_sums():
    N, primes = C2
    o, e = _engine_and_oracle(N, primes)
    L = 3
    dk = e.to_device(_key(o, 5))
    cts = [e.to_device(o.uniform(L, 2, 10 + i)) for i in range(2)]
    acc = e.to_device(o.uniform(L, 2, 20))
    with pytest.raises(ValueError):
        e.apply_galois_add_batch(L, cts, [3, 5], [dk, dk], [acc, acc], acc_outs=[acc, acc])   # two items, one sum
    with pytest.raises(ValueError):
        e.apply_galois_add_batch(L, cts, [3, 5], [dk, dk], [acc, acc], acc_outs=[cts[1], e.empty(2, L, N)])  # a sum over an input


@pytest.mark.parametrize("setname,L,n,steps", [("C3", 2, 8, 1), ("C3", 2, 8, 2), ("C3", 2, 8, 3), ("C3", 2, 8, 12),
                                               ("C3", 2, 8, 13), ("C3", 5, 1, 11), ("C2", 3, 3, 10), ("C3", 2, 40, 4),
                                               ("C4", 7, 8, 5), ("C5", 5, 2, 4), ("C3", 2, 17, 4), ("C3", 2, 24, 5),
                                               ("C2", 2, 32, 3)])
def test_rotate_add_chain_bit_exact(setname, L, n, steps):
    """hefx_rotate_add_chain = the loop of helper.h:472-476 (rotate_vector_inplace(dup, step); add_inplace(mult, dup)) for
    n pairs in lockstep, against the oracle's loop: final rotation and final sum word for word, inputs untouched.  12 / 13
    steps at n = 8, L = 2 is the shape of the LR gradient's chains (logistic_regression_ckks.cpp:295-300), an even and an odd
    count of middle levels; 1-3 steps the degenerate plans; n = 40 the wide path; n = 17 / 24 / 32: two / three / three LANES
    (round 5: one lane per 8 chains, each on its own stream with its own scratch slice and descriptor slots)."""
    N, primes = SETS[setname]
    o, e = _engine_and_oracle(N, primes)
    keys = [_key(o, 61 + i) for i in range(2)]
    dkeys = [e.to_device(k) for k in keys]
    from seal_fyp_logistic_regression_amd.seal import galois_elt_from_step
    elts = [galois_elt_from_step(1 if i % 2 == 0 else -2, N) for i in range(n)]
    cts = [o.uniform(L, 2, 2000 + i) for i in range(n)]
    accs = [o.uniform(L, 2, 6000 + i) for i in range(n)]
    dct, dacc = [e.to_device(c) for c in cts], [e.to_device(a) for a in accs]
    outs, sums = e.rotate_add_chain(L, dct, elts, [dkeys[i % 2] for i in range(n)], dacc, steps)
    e.sync()
    for i in (range(n) if n <= 8 else sorted({0, 1, n // 3, n // 2, (2 * n) // 3, n - 2, n - 1})):
        t, a = cts[i], accs[i]
        for _ in range(steps):
            t = o.apply_galois(t, elts[i], keys[i % 2])
            a = o.add(a, t)
        assert (outs[i].download() == t).all(), ("rotation", i)
        assert (sums[i].download() == a).all(), ("sum", i)
        assert (dct[i].download() == cts[i]).all() and (dacc[i].download() == accs[i]).all(), ("inputs untouched", i)


def test_rotate_add_chain_with_graph_replay_bit_exact():
    """HEFX_CHAIN_GRAPH=1 (the two alternating levels captured as a HIP graph and replayed; opt-in, measured slower than
    plain launches) gives the same words."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from oracle import oracle as O\n"
        "from seal_fyp_logistic_regression_amd import Engine\n"
        "N, primes = %r\n"
        "o, e = O.Oracle(N, primes), Engine(N, primes); L, n, steps = 2, 4, 12\n"
        "key = o.uniform(o.k, 2*(o.k-1), 61).reshape(o.k-1, 2, o.k, o.N); dk = e.to_device(key)\n"
        "cts = [o.uniform(L, 2, 2000+i) for i in range(n)]; accs = [o.uniform(L, 2, 6000+i) for i in range(n)]\n"
        "outs, sums = e.rotate_add_chain(L, [e.to_device(c) for c in cts], [3]*n, [dk]*n, [e.to_device(a) for a in accs], steps)\n"
        "ok = True\n"
        "for i in range(n):\n"
        "    t, a = cts[i], accs[i]\n"
        "    for _ in range(steps):\n"
        "        t = o.apply_galois(t, 3, key); a = o.add(a, t)\n"
        "    ok = ok and bool((outs[i].download() == t).all()) and bool((sums[i].download() == a).all())\n"
        "print('PARITY', ok)\n") % (ROOT, C3)
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "HEFX_CHAIN_GRAPH": "1"}, capture_output=True,
                       text=True, timeout=600)
    assert "PARITY True" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


# ------------------------------------------------------------------------------------------------------------------
# batched encode / encrypt (what the C++ shim's recorder submits for a loop of encode + encrypt calls)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("setname,L,n", [("C2", 3, 5), ("C3", 5, 300), ("C3", 2, 1)])
def test_encrypt_batch_equals_single_encryptions_word_for_word(setname, L, n):
    """hefx_encrypt_batch(first_stream_id = s) == n calls of hefx_encrypt with stream ids s, s+1, ... (same public key,
    same sampler key): the sampler is counter mode, item i draws from the sub-streams 4(s+i)+{0,1,2}.  Integer work: bit
    equality.  hefx_encrypt itself is pinned against the oracle's sampler in test_gpu_sampling.py; here additionally the
    first, a middle and the last item of the batch against an oracle-side encryption.  300 items: two 256-item slices;
    a None plaintext is an encryption of zero."""
    N, primes = C3 if setname == "C3" else C2
    o, e = _engine_and_oracle(N, primes)
    k = len(primes)
    key32 = bytes((11 * i + 5) & 0xFF for i in range(32))
    pk = e.sample("uniform", bytes(range(32)), 3, 2, k, 0)
    big = e.sample("uniform", bytes(range(32)), 4, n, L, 0)
    plains = [big.view(i * L * N, (L, N)) for i in range(n)]
    if n > 2:
        plains[1] = None
    s0 = 2 ** 33 + 7
    outs = e.encrypt_batch(L, pk, plains, key32, s0)
    for i in (range(n) if n <= 8 else (0, 1, 2, 127, 255, 256, 257, n - 1)):
        single = e.encrypt(L, pk, plains[i], key32, s0 + i).download()
        assert (outs[i].download() == single).all(), i


@pytest.mark.parametrize("setname,L,count,nvalues", [("C2", 3, 7, 5), ("C3", 5, 300, 2000), ("C3", 3, 2, 8192)])
def test_ckks_encode_batch_equals_contiguous_encode(setname, L, count, nvalues):
    """hefx_ckks_encode_batch (separately allocated outputs through a pointer table) writes the words hefx_ckks_encode
    writes for the same vectors -- the same kernels on the same inputs, then one scatter launch: bit equality."""
    N, primes = C3 if setname == "C3" else C2
    o, e = _engine_and_oracle(N, primes)
    rng = np.random.default_rng(count)
    vals = rng.uniform(-3, 3, (count, nvalues))
    vals[0, :] = 0.0
    vals[0, 0] = 1.0  # a one-hot mask, as in logistic_regression_ckks.cpp:222-225
    want = e.ckks_encode(L, vals, 2.0 ** 40).download().reshape(count, L, N)
    outs = e.ckks_encode_batch(L, vals, 2.0 ** 40)
    for i in (range(count) if count <= 8 else (0, 1, 255, 256, count - 1)):
        assert (outs[i].download() == want[i]).all(), i


def test_large_chunks_bit_exact_and_device_memory():
    """HEFX_CHUNK above 256 (the descriptor ring's slots hold 512 items since round 4; the size rule itself still stops at
    256): 600 items at N = 8192 as chunks of 512 + 88, a sample of outputs word for word against the oracle; and
    hefx_device_memory reports a plausible device."""
    code = (
        "import sys, ctypes; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from oracle import oracle as O\n"
        "from seal_fyp_logistic_regression_amd import Engine, capi\n"
        "N, primes = %r\n"
        "o, e = O.Oracle(N, primes), Engine(N, primes); L, n, k = 3, 600, len(primes)\n"
        "key = e.sample('uniform', bytes(range(32)), 1, 2 * L, k, 0)\n"
        "ct = e.sample('uniform', bytes(range(32)), 2, 2 * n, L, 0); pt = e.sample('uniform', bytes(range(32)), 3, n, L, 0)\n"
        "cts = [ct.view(i * 2 * L * N, (2, L, N)) for i in range(n)]; pts = [pt.view(i * L * N, (L, N)) for i in range(n)]\n"
        "outs = e.rotate_multiply_plain_batch(L, cts, [3] * n, [key] * n, pts)\n"
        "hk = key.download().reshape(L, 2, k, N)\n"
        "ok = all(bool((outs[i].download() == o.rotate_mulplain(cts[i].download(), 3, hk, pts[i].download())).all())\n"
        "         for i in (0, 255, 256, 511, 512, 599))\n"
        "f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)\n"
        "capi.check(capi.lib().hefx_device_memory(e._h, ctypes.byref(f), ctypes.byref(t)))\n"
        "print('PARITY', ok, 'MEM', 0 < f.value <= t.value and t.value > (16 << 30))\n") % (ROOT, C2)
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "HEFX_CHUNK": "512"}, capture_output=True, text=True,
                       timeout=600)
    assert "PARITY True MEM True" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("env", [{"SEAL_SHIM_CHAINS": "0"}, {"SEAL_SHIM_FUSE_ADD": "0"}, {"HEFX_CHAIN_GRAPH": "1"},
                                 {"SEAL_SHIM_PENDING_MB": "64"}])
def test_cpp_shim_selftest_with_the_fusions_switched_off(env):
    """drivers/shim_selftest.cpp compares every recorded run with call-by-call execution bit for bit; here again with the
    chain detection off (pairs go out as hefx_apply_galois_add_batch per level), with the pair fusion off (round 3's
    rotate batch + add batch), with the chain levels replayed as a HIP graph, and with a 64 MB pending budget (submissions
    forced in the middle of everything)."""
    exe = os.path.join(ROOT, "drivers", "_ref", "shim_selftest")
    if not os.path.exists(exe):
        pytest.skip("drivers/_ref/shim_selftest is not built (make -C drivers)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env={**os.environ, **env})
    assert r.returncode == 0 and "SELFTEST PASSED" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _plant_zero_coefficients(o, ct, L, rng, rows, count):
    """ct with `count` zero COEFFICIENTS in the inverse transform of c1's RNS rows `rows` (the digits of a key switch)"""
    ct = ct.copy()
    for i in rows:
        coef = o.ntt_inv(i, ct[1, i])
        coef[rng.choice(o.N, size=count, replace=False)] = 0
        ct[1, i] = o.ntt_fwd(i, coef)
    return ct


@pytest.mark.parametrize("setname,L,n,nsrc,zeros", [
    ("C3", 5, 100, 3, None), ("C2", 3, 600, 1, None), ("C3", 2, 72, 18, None), ("C3", 2, 70, 35, None), ("C4", 8, 40, 2, None),
    ("C5", 5, 36, 1, None), ("C4", 3, 64, 4, None), ("E61", 3, 48, 2, None),
    ("C3", 5, 64, 2, "one"), ("C3", 5, 40, 1, "transparent"), ("C2", 3, 600, 2, "second_chunk"), ("C4", 8, 40, 2, "many")])
def test_shared_source_decomposition_bit_exact(setname, L, n, nsrc, zeros):
    """Batches in which many items rotate the SAME ciphertext (the d-1 rotations of Linear_Transform_Plain, helper.h:252-257)
    run EXACTLY HOISTED: the distinct sources are decomposed and extended to every key modulus once per chunk, every item
    runs the gathered key MAC with the flip-mask correction (ks_mac_exact_kernel) and its own mod-down -- SEAL's words,
    because the rotated digit is the signed permutation of the source's plus q_i on the negated coefficients.  Mixed
    elements (incl. conjugation 2N-1 and large ones), two keys, with and without the fused plaintext product, sources
    interleaved; every output (a sample at n = 600) against the oracle, which decomposes every item on its own.
    n = 600 at N = 8192: two chunks; nsrc = n/4: near the boundary of the mode (n/3); nsrc = n/2: the same inputs through the
    ordinary per-item decomposition; C5: N = 32768; E61: four 61-bit primes at N = 4096 (the 128-bit MAC policy).
    zeros: the one input class the hoisted identity does not cover -- a source whose INTT(c1) has a ZERO coefficient in some
    RNS row (a negated zero stays 0, not q_i).  The source decomposition detects it on the device and the chunk is redone
    by the per-item kernels: "one" plants a single zero in one row of one source, "many" 50 in every row, "transparent" is
    c1 = 0, "second_chunk" has the zero in a source only items of the second chunk rotate.  hefx_ks_fallback_count tells
    which path ran: 0 without zeros (the fast path really ran), >= 1 with."""
    N, primes = _edge61() if setname == "E61" else SETS[setname]
    o, e = _engine_and_oracle(N, primes)
    rng = np.random.default_rng(7 * n + nsrc)
    keys = [_key(o, 91 + i) for i in range(2)]
    dkeys = [e.to_device(k) for k in keys]
    srcs = [o.uniform(L, 2, 3000 + i) for i in range(nsrc)]
    si = [int(rng.integers(nsrc)) for _ in range(n)]
    if zeros == "one":
        srcs[1] = _plant_zero_coefficients(o, srcs[1], L, rng, [L - 2], 1)
    elif zeros == "many":
        srcs[0] = _plant_zero_coefficients(o, srcs[0], L, rng, range(L), 50)
    elif zeros == "transparent":
        srcs[0][1] = 0
    elif zeros == "second_chunk":
        srcs[1] = _plant_zero_coefficients(o, srcs[1], L, rng, [0], 3)
        si = [0] * 512 + [int(rng.integers(2)) for _ in range(n - 512)]
        si[-1] = 1
    dsrcs = [e.to_device(c) for c in srcs]
    pts = [o.uniform(L, 1, 4000 + i)[0] for i in range(min(n, 8))]
    dpts = [e.to_device(p) for p in pts]
    elts = [int(2 * rng.integers(1, N) + 1) for _ in range(n)]
    elts[0], elts[-1] = 2 * N - 1, 3
    ki = [int(rng.integers(2)) for _ in range(n)]
    if zeros == "second_chunk":
        ki = [0] * n  # one key: the items keep their order (no grouping by key), so the chunks are [0, 512) and [512, 600)
    before = e.ks_fallback_count()
    outs = e.rotate_multiply_plain_batch(L, [dsrcs[s] for s in si], elts, [dkeys[k] for k in ki], [dpts[i % len(pts)] for i in range(n)])
    mid = e.ks_fallback_count()
    plain = e.apply_galois_batch(L, [dsrcs[s] for s in si], elts, [dkeys[k] for k in ki])
    after = e.ks_fallback_count()
    for i in (range(n) if n <= 100 else sorted({0, 1, 255, 256, 511, 512, 513, n - 1} | set(int(x) for x in rng.integers(0, n, 10)))):
        assert (outs[i].download() == o.rotate_mulplain(srcs[si[i]], elts[i], keys[ki[i]], pts[i % len(pts)])).all(), ("fused", i)
        assert (plain[i].download() == o.apply_galois(srcs[si[i]], elts[i], keys[ki[i]])).all(), ("plain", i)
    for s in range(nsrc):
        assert (dsrcs[s].download() == srcs[s]).all(), "sources untouched"
    if os.environ.get("HEFX_SHARE_SRC", "1") != "0" and nsrc * 3 <= n:
        if zeros is None:
            assert after == before, "a chunk of random ciphertexts fell back to the per-item path"
        elif zeros == "second_chunk":
            assert (mid - before, after - mid) == (1, 1), "exactly the second chunk of each call falls back"
        else:
            assert mid > before and after > mid, "a source with a zero coefficient must take the per-item path"

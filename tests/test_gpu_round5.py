"""Round-5 GPU tests.

* The torch "nccl" (= RCCL) code path at world 1: every multi-rank test of the suite rendezvous over
  gloo because the box has one GPU and RCCL wants one device per rank -- so `init_process_group("nccl", device_id=...)`,
  `dist.barrier()` on a device, the in-place `all_reduce` on the `__cuda_array_interface__` view of an engine buffer
  (parallel.py:115-158) and the engine's own communicator (`hefx_comm_init` / `hefx_allreduce_sum`) had never executed.
  A one-rank group runs all of them on the one-GPU box; the sums of one addend must leave the serial bits.
* bench.py with HEFX_BENCH_FORCE_PG=1: the process group, the barriers, the MAX all-reduce of the timing and the sharded
  linear-transform leg over "nccl" at world 1.
"""
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = dict(os.environ, **extra)
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HEFX_BENCH_BACKEND"):
        env.pop(v, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


_WORKER = textwrap.dedent('''
    import os, socket, sys, json
    import numpy as np
    sys.path.insert(0, os.environ["HEFX_ROOT"])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    res = {}
    try:
        dist.barrier()
        from seal_fyp_logistic_regression_amd import algorithms as alg, parallel as par, seal as S
        parms = S.EncryptionParameters("ckks")
        parms.set_poly_modulus_degree(4096)
        parms.set_coeff_modulus(S.CoeffModulus.Create(4096, [60, 40, 40, 40, 40, 40, 40, 40, 60]))
        ctx = S.SEALContext.Create(parms)
        assert ctx.backend.name == "hip"
        kg = S.KeyGenerator(ctx, 3)
        enc, dec = S.Encryptor(ctx, kg.public_key(), 5), S.Decryptor(ctx, kg.secret_key())
        encoder, ev, gk = S.CKKSEncoder(ctx), S.Evaluator(ctx), kg.galois_keys()
        bits = lambda c: ctx.backend.to_host(c.data)
        scale = 2.0 ** 40
        rng = np.random.default_rng(7)
        d = 9
        M, v = rng.standard_normal((d, d)), rng.standard_normal(d)
        diags = [encoder.encode(x, scale) for x in alg.get_all_diagonals(M)]
        ct = enc.encrypt(encoder.encode(v, scale))
        serial = alg.linear_transform_plain(ev, ct, diags, gk)
        n = 3
        A = rng.standard_normal((n, n))
        Us, Ut, V, W = alg.matmul_permutation_matrices(n)
        dense = lambda U: [encoder.encode(dg + 1e-8, scale) for dg in alg.get_all_diagonals(U)]
        cA = enc.encrypt(encoder.encode(A.reshape(-1), scale))
        margs = (dense(Us), dense(Ut), [dense(x) for x in V], [dense(x) for x in W])
        m_serial = alg.cc_matrix_multiplication(ev, cA, cA, n, *margs, gk)
        for mode in ("off", "on"):       # off: torch.distributed in place on the payload; on: hefx_allreduce_sum (RCCL inside libhefx)
            par.ENGINE_COMM = mode
            par._comm_decisions.clear()
            one = par.allreduce_ciphertext(ev, ct)
            res["allreduce_" + mode] = bool((bits(one) == bits(ct)).all())
            sh = par.linear_transform_plain_sharded(ev, ct, diags, gk)
            res["lt_" + mode] = bool((bits(sh) == bits(serial)).all())
            mm = par.cc_matrix_multiplication_sharded(ev, cA, cA, n, *margs, gk)
            res["mm_" + mode] = bool((bits(mm) == bits(m_serial)).all())
            res["comm_world_" + mode] = int(ctx.backend.engine.comm_world)
        res["lt_value"] = bool(np.allclose(encoder.decode(dec.decrypt(sh))[:d].real, M @ v, atol=1e-4))
        ctx.backend.engine.comm_destroy()
        dist.barrier()
    finally:
        dist.destroy_process_group()
    print("RESULT " + json.dumps(res))
''')


@pytest.mark.timeout(600)
def test_nccl_world1_exchange_paths_leave_the_serial_bits():
    r = subprocess.run([sys.executable, "-c", _WORKER], cwd=ROOT, env=_clean_env(HEFX_ROOT=ROOT), capture_output=True,
                       text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(line[-1][7:])
    for k in ("allreduce_off", "lt_off", "mm_off", "allreduce_on", "lt_on", "mm_on", "lt_value"):
        assert res[k] is True, (k, res)
    assert res["comm_world_off"] == 0 and res["comm_world_on"] == 1, res  # "on" really attached the engine's communicator


@pytest.mark.timeout(900)
def test_bench_world1_over_nccl_takes_the_collective_branches():
    """bench.py creates the "nccl" process group at world 1 (HEFX_BENCH_FORCE_PG=1) and then runs what an N-rank run runs:
    barriers around the timed region, the MAX all-reduce of the wall time, the diagonal-sharded linear transform with its
    all-reduce through torch.distributed AND behind the C-ABI -- one JSON line, bits equal serial."""
    env = _clean_env(HEFX_BENCH_FORCE_PG="1", HEFX_BENCH_C_ABI_COMM="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "512",
                        "--cpu-seconds", "0", "--lt", "16", "--lt-direct", "0", "--key-per-item", "0", "--variant-keys", "0",
                        "--stream-keys", "0", "--secondary", "C2", "--sustain", "0.3"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["verified"] is True and line["value"] > 0
    d16 = line["lt_sharded"]["d16"]
    assert d16["bits_equal_serial"] is True and d16["decrypts_to_Mv"] is True, d16
    assert d16["c_abi_allreduce"].get("bits_equal_serial") is True, d16
    assert d16["key_switches_executed"] <= d16["key_switches_serial"] and d16["key_switches_executed"] > 0
    # the N = 8192 leg (north_star's other degree), the sustained pass and the hashes ride in the single-rank line
    c2 = line["secondary"]["C2"]
    assert c2["verified"] is True and c2["value"] > 0 and 0 < c2["roofline"]["frac"] < 1, c2
    assert line["sustained"]["steps"] >= 2 and len(line["csrc_sha16"]) == 16 and line["rescale_mode"] in ("floor", "round")


def test_ks_stats_count_the_naf_forest_and_hoisting():
    """hefx_ks_stats: key switches submitted / hoisted / launch sequences, host-side.  A direct-key linear transform of
    d = 64 runs its 63 rotations of ct_new as ONE exactly hoisted batch; with the default power-of-two keys the forest shares
    prefixes, so fewer key switches execute than the op-by-op loop's NAF count."""
    import numpy as np
    from seal_fyp_logistic_regression_amd import algorithms as alg
    from seal_fyp_logistic_regression_amd import seal as S
    parms = S.EncryptionParameters("ckks")
    parms.set_poly_modulus_degree(4096)
    parms.set_coeff_modulus(S.CoeffModulus.Create(4096, [60, 40, 40, 60]))
    ctx = S.SEALContext.Create(parms)
    kg = S.KeyGenerator(ctx, 1)
    enc, encoder, ev = S.Encryptor(ctx, kg.public_key(), 2), S.CKKSEncoder(ctx), S.Evaluator(ctx)
    eng = ctx.backend.engine
    d = 64
    rng = np.random.default_rng(0)
    diags = encoder.encode_many(list(alg.get_all_diagonals(rng.standard_normal((d, d)))), 2.0 ** 30)
    ct = enc.encrypt(encoder.encode(rng.standard_normal(d), 2.0 ** 30))
    gk_direct = kg.galois_keys([-d] + list(range(1, d)))
    s0 = eng.ks_stats()
    alg.linear_transform_plain(ev, ct, diags, gk_direct)
    s1 = eng.ks_stats()
    assert s1["key_switches"] - s0["key_switches"] == d          # rotate(-d) + d - 1 rotations of ct_new
    assert s1["hoisted"] - s0["hoisted"] == d - 1
    gk = kg.galois_keys()
    naf = sum(len(ev.rotation_plan(s, gk)) for s in [-d] + list(range(1, d)))
    alg.linear_transform_plain(ev, ct, diags, gk)
    s2 = eng.ks_stats()
    assert d <= s2["key_switches"] - s1["key_switches"] < naf
    assert s2["calls"] > s1["calls"] and s2["chunks"] > s1["chunks"]


def test_chain_refuses_null_and_overlapping_outputs_before_anything_runs():
    """the small-n path of hefx_rotate_add_chain launched its last level without validating ct_out --
    a null pointer faulted on the GPU, equal or overlapping outputs gave wrong bits silently, and n > 32 refused the same
    arguments.  Now every n refuses them up front."""
    import ctypes as C
    from oracle import oracle as O
    from seal_fyp_logistic_regression_amd import Engine, capi
    N, primes = 4096, [0xffffffffffc0001, 0xfffffd8001, 0xfffffffff00001]
    from seal_fyp_logistic_regression_amd.seal import CoeffModulus
    primes = CoeffModulus.Create(N, [60, 40, 60])
    o, e = O.Oracle(N, primes), Engine(N, primes)
    L, k = 2, 3
    key = e.to_device(o.uniform(k, 2 * L, 3).reshape(L, 2, k, N))
    cts = [e.to_device(o.uniform(L, 2, 10 + i)) for i in range(2)]
    accs = [e.to_device(o.uniform(L, 2, 20 + i)) for i in range(2)]
    outs = [e.empty(2, L, N) for _ in range(2)]
    lib = capi.lib()
    arr = lambda ptrs: (C.c_void_p * len(ptrs))(*ptrs)
    elts = (C.c_uint32 * 2)(3, 3)
    keys = arr([key.ptr, key.ptr])

    def call(ct_out, acc_out):
        return lib.hefx_rotate_add_chain(e._h, L, 2, arr([c.ptr for c in cts]), elts, keys, arr([a.ptr for a in accs]),
                                         arr(acc_out), arr(ct_out), 3, None)

    good_acc = [a.ptr for a in accs]
    assert call([outs[0].ptr, None], good_acc) == capi.HEFX_ERR_INVALID
    assert call([outs[0].ptr, outs[0].ptr], good_acc) == capi.HEFX_ERR_INVALID
    assert call([outs[0].ptr, accs[0].ptr], good_acc) == capi.HEFX_ERR_INVALID            # a rotation output on a sum
    assert call([outs[0].ptr, outs[0].ptr + 8 * N], good_acc) == capi.HEFX_ERR_INVALID    # partial overlap
    e.sync()
    before = [a.download() for a in accs]
    assert call([outs[0].ptr, outs[1].ptr], good_acc) == 0
    e.sync()
    # the refused calls had not touched the sums: the accepted one starts from the original values
    elt = 3
    for i in range(2):
        t, a = o.uniform(L, 2, 10 + i), o.uniform(L, 2, 20 + i)
        assert (before[i] == a).all()
        for _ in range(3):
            t = o.apply_galois(t, elt, o.uniform(k, 2 * L, 3).reshape(L, 2, k, N))
            a = o.add(a, t)
        assert (outs[i].download() == t).all() and (accs[i].download() == a).all()


@pytest.mark.gpu
def test_naf_forest_on_two_lanes_bit_exact_vs_op_by_op_and_vs_one_lane():
    """hefx_linear_transform_plain with the reference's default keys deals the subtrees of a large NAF forest onto two lanes
    (two streams, two halves of the scratch buffer).  d = 160 at N = 4096 is well above the 96-node bound: the words must
    be those of the oracle-backed twin's op-by-op loop (helper.h:237-262), and those of the same call with HEFX_LT_LANES=0."""
    import hashlib
    import numpy as np
    from seal_fyp_logistic_regression_amd import algorithms as alg
    from tests.test_gpu_composites import both, bits, decode
    d = 160
    rng = np.random.default_rng(160)
    M, v = rng.standard_normal((d, d)), rng.standard_normal(d)

    def run(e):
        scale = 2.0 ** 30
        diags = [e["encoder"].encode(x, scale) for x in alg.get_all_diagonals(M)]
        ct = e["enc"].encrypt(e["encoder"].encode(v, scale))
        eng = getattr(e["ctx"].backend, "engine", None)
        s0 = eng.ks_stats() if eng else None
        out = alg.linear_transform_plain(e["ev"], ct, diags, e["gk"])
        return out, (eng.ks_stats()["chunks"] - s0["chunks"]) if eng else None

    r = both(4096, [50, 30, 30, 50], run)
    (eg, (ag, seqs)), (eo, (ao, _)) = r["gpu"], r["oracle"]
    assert (bits(eg, ag) == bits(eo, ao)).all()
    assert np.allclose(decode(eg, ag, d), M @ v, atol=0.5)  # scale 2^30 and ~400 key switches: value check only loosely
    digest = hashlib.sha256(np.ascontiguousarray(bits(eg, ag)).tobytes()).hexdigest()
    code = textwrap.dedent(f"""
        import hashlib, sys
        import numpy as np
        sys.path.insert(0, {ROOT!r})
        from seal_fyp_logistic_regression_amd import algorithms as alg
        from tests.test_gpu_composites import make, bits
        d = {d}
        rng = np.random.default_rng(160)
        M, v = rng.standard_normal((d, d)), rng.standard_normal(d)
        e = make(4096, [50, 30, 30, 50], "gpu")
        diags = [e["encoder"].encode(x, 2.0 ** 30) for x in alg.get_all_diagonals(M)]
        ct = e["enc"].encrypt(e["encoder"].encode(v, 2.0 ** 30))
        eng = e["ctx"].backend.engine
        s0 = eng.ks_stats()["chunks"]
        out = alg.linear_transform_plain(e["ev"], ct, diags, e["gk"])
        print(hashlib.sha256(np.ascontiguousarray(bits(e, out)).tobytes()).hexdigest(), eng.ks_stats()["chunks"] - s0)
    """)
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HEFX_LT_LANES="0"), capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    one_lane_digest, one_lane_seqs = p.stdout.split()[-2:]
    assert one_lane_digest == digest
    assert seqs > int(one_lane_seqs), (seqs, one_lane_seqs)  # the laned call really submitted more (smaller) launch sequences


@pytest.mark.gpu
def test_mixed_batch_of_plain_and_fused_rotations_and_growing_batches_without_a_sync():
    """(1) hefx_rotate_multiply_plain_batch with NULL plaintext entries: the items without a plaintext are plain rotations, the
    others fused products -- one batch, the oracle's words for both kinds (what the C++ shim submits per forest depth).
    (2) batches that grow from 3 to 300 items submitted back to back WITHOUT a synchronisation in between: the scratch buffer is
    re-allocated under work in flight (outgrown buffers are retired, not freed) -- every output must still be the oracle's."""
    import numpy as np
    from oracle import oracle as O
    from seal_fyp_logistic_regression_amd import Engine
    N, primes = 4096, O.coeff_modulus_create(4096, [50, 30, 30, 50])
    L = len(primes) - 1
    o, e = O.Oracle(N, primes), Engine(N, primes, device=0)
    key = o.uniform(len(primes), 2 * L, 3).reshape(L, 2, len(primes), N)
    dkey = e.to_device(key)
    elts = [O.galois_elt_from_step(N, s) for s in (1, -1, 2, 5)]
    cts = [o.uniform(L, 2, 100 + i) for i in range(300)]
    pts = [o.uniform(L, 1, 500 + i)[0] for i in range(300)]
    dcts, dpts = [e.to_device(c) for c in cts], [e.to_device(p) for p in pts]
    # (1) forty items, every third one without a plaintext
    n = 40
    mix = [None if i % 3 == 0 else dpts[i] for i in range(n)]
    outs = e.rotate_multiply_plain_batch(L, dcts[:n], [elts[i % 4] for i in range(n)], [dkey] * n, mix)
    for i in range(n):
        want = o.apply_galois(cts[i], elts[i % 4], key) if i % 3 == 0 else o.rotate_mulplain(cts[i], elts[i % 4], key, pts[i])
        assert (outs[i].download() == want).all(), i
    # (2) a fresh engine (small scratch), growing batches, no sync until the end
    e2 = Engine(N, primes, device=0)
    dkey2 = e2.to_device(key)
    d2c, d2p = [e2.to_device(c) for c in cts], [e2.to_device(p) for p in pts]
    results = []
    for n in (3, 20, 70, 300):
        results.append((n, e2.rotate_multiply_plain_batch(L, d2c[:n], [elts[i % 4] for i in range(n)], [dkey2] * n, d2p[:n])))
    for n, outs in results:
        for i in sorted(set([0, 1, n // 2, n - 1])):
            assert (outs[i].download() == o.rotate_mulplain(cts[i], elts[i % 4], key, pts[i])).all(), (n, i)


@pytest.mark.gpu
@pytest.mark.parametrize("nroots,fan,depth", [(3, 2, 3), (12, 3, 3)])
def test_apply_galois_forest_bit_exact_vs_node_by_node(nroots, fan, depth):
    """hefx_apply_galois_forest: a forest of rotations (some ending in a plaintext product) against the oracle node by node.
    The second case has 12 * (1 + 3 + 9) = 156 nodes: above the 96-node bound, i.e. on two lanes; and a refused call."""
    import numpy as np
    from oracle import oracle as O
    from seal_fyp_logistic_regression_amd import Engine
    N, primes = 4096, O.coeff_modulus_create(4096, [50, 30, 30, 50])
    L = len(primes) - 1
    o, e = O.Oracle(N, primes), Engine(N, primes, device=0)
    steps = [1, -1, 2, -2, 4, 8]
    elt_of = {s: O.galois_elt_from_step(N, s) for s in steps}
    keys = {s: o.uniform(len(primes), 2 * L, 40 + i).reshape(L, 2, len(primes), N) for i, s in enumerate(steps)}
    dkeys = {s: e.to_device(k) for s, k in keys.items()}
    srcs = [o.uniform(L, 2, 700 + r) for r in range(2)]
    dsrcs = [e.to_device(x) for x in srcs]
    rng = np.random.default_rng(nroots * 100 + fan)
    parents, step_of, src_of, pt_of = [], [], [], []
    level = []
    for r in range(nroots):  # roots rotate one of two external ciphertexts
        parents.append(-1), step_of.append(steps[r % len(steps)]), src_of.append(r % 2), level.append(len(parents) - 1)
    for _ in range(depth - 1):
        nxt = []
        for p in level:
            for f in range(fan):
                parents.append(p), step_of.append(steps[int(rng.integers(len(steps)))]), src_of.append(None), nxt.append(len(parents) - 1)
        level = nxt
    n = len(parents)
    pts = [o.uniform(L, 1, 900 + i)[0] if i % 3 == 1 else None for i in range(n)]
    dpts = [e.to_device(p) if p is not None else None for p in pts]
    outs = e.apply_galois_forest(L, parents, [dsrcs[s] if s is not None else None for s in src_of],
                                 [elt_of[s] for s in step_of], [dkeys[s] for s in step_of], dpts)
    # the oracle, node by node: a child of a node with a plaintext rotates the PRODUCT (that is what the node's output holds)
    want = []
    for i in range(n):
        src = srcs[src_of[i]] if parents[i] < 0 else want[parents[i]]
        if pts[i] is None:
            want.append(o.apply_galois(src, elt_of[step_of[i]], keys[step_of[i]]))
        else:
            want.append(o.rotate_mulplain(src, elt_of[step_of[i]], keys[step_of[i]], pts[i]))
    for i in range(n):
        assert (outs[i].download() == want[i]).all(), i
    with pytest.raises(ValueError):  # a child listed before its parent (HEFX_ERR_INVALID)
        e.apply_galois_forest(L, [1, -1], [None, dsrcs[0]], [elt_of[1]] * 2, [dkeys[1]] * 2)
    with pytest.raises(ValueError):  # two nodes with one output
        one = e.empty(2, L, N)
        e.apply_galois_forest(L, [-1, -1], [dsrcs[0], dsrcs[1]], [elt_of[1]] * 2, [dkeys[1]] * 2, outs=[one, one])

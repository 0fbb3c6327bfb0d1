"""N>1 path on CPU: world_size-2 gloo processes run the sharded linear transform on the oracle-backed backend and
must reproduce the serial result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, d, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from seal_fyp_logistic_regression_amd import algorithms as alg
        from seal_fyp_logistic_regression_amd import parallel as par
        from tests.test_host_api_cpu import make
        e = make(2048, [50, 30, 30, 50], seed=3)
        rng = np.random.default_rng(11)
        M, v = rng.standard_normal((d, d)), rng.standard_normal(d)
        scale = 2.0 ** 30
        diags = [e["encoder"].encode(x, scale) for x in alg.get_all_diagonals(M)]
        import seal_fyp_logistic_regression_amd.seal as S
        ct = S.Encryptor(e["ctx"], e["kg"].public_key(), seed=5).encrypt(e["encoder"].encode(v, scale))
        serial = alg.linear_transform_plain(e["ev"], ct, diags, e["gk"])
        sharded = par.linear_transform_plain_sharded(e["ev"], ct, diags, e["gk"])
        same = bool((np.asarray(serial.data) == np.asarray(sharded.data)).all())
        val = e["encoder"].decode(e["dec"].decrypt(sharded))[:d].real
        q.put((rank, same, sharded.parms_id() == serial.parms_id(), bool(np.allclose(val, M @ v, atol=1e-2)),
               len(list(par.shard(d, rank, world)))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("d,world", [(8, 2), (1, 2), (3, 4), (16, 8)])
def test_sharded_linear_transform_world2(d, world):
    """world 4 with d = 3: one rank owns no diagonal and contributes the zero ciphertext (the 8-GPU node's case whenever a
    transform has fewer units than ranks)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, d, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, level_ok, value_ok, nmine in sorted(out):
        assert same, f"rank {rank}: sharded result differs from the serial add_many"
        assert level_ok and value_ok
    assert sum(o[4] for o in out) == d


def test_shard_partition():
    from seal_fyp_logistic_regression_amd.parallel import shard
    for n in (0, 1, 7, 16, 29):
        for w in (1, 2, 3, 8):
            parts = [list(shard(n, r, w)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _worker_matmul_lr(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import seal_fyp_logistic_regression_amd.seal as S
        from seal_fyp_logistic_regression_amd import algorithms as alg
        from seal_fyp_logistic_regression_amd import parallel as par
        from tests.test_host_api_cpu import make
        # ---- matrix product n = 3 (Step 2 sharded by k)
        n = 3
        e = make(2048, [60, 40, 40, 40, 40, 60], seed=6)  # config 3's chain: the product ends at scale 2^160
        A = np.arange(1, n * n + 1, dtype=float).reshape(n, n) / 4
        d = n * n
        Us, Ut = np.zeros((d, d)), np.zeros((d, d))
        for i in range(n):
            for j in range(n):
                Us[n * i + j, n * i + (i + j) % n] = 1
                Ut[n * i + j, n * ((i + j) % n) + j] = 1
        V, W = [], []
        for k in range(1, n):
            Vk, Wk = np.zeros((d, d)), np.zeros((d, d))
            for i in range(n):
                for j in range(n):
                    Vk[n * i + j, n * i + (j + k) % n] = 1
                    Wk[n * i + j, n * ((i + k) % n) + j] = 1
            V.append(Vk)
            W.append(Wk)
        scale = 2.0 ** 40
        enc = lambda U: [e["encoder"].encode(dg + 1e-8, scale) for dg in alg.get_all_diagonals(U)]
        ctA = S.Encryptor(e["ctx"], e["kg"].public_key(), seed=7).encrypt(e["encoder"].encode(A.reshape(-1), scale))
        args = (ctA, ctA, n, enc(Us), enc(Ut), [enc(v) for v in V], [enc(w) for w in W], e["gk"])
        serial = alg.cc_matrix_multiplication(e["ev"], *args)
        sharded = par.cc_matrix_multiplication_sharded(e["ev"], *args)
        mm_same = bool((np.asarray(serial.data) == np.asarray(sharded.data)).all()) and sharded.size() == 3
        got = e["encoder"].decode(e["dec"].decrypt(sharded))[:d].real.reshape(n, n)
        mm_val = bool(np.allclose(got, A @ A, atol=5e-2))
        # ---- LR predict, rows sharded; same Encryptor state on both paths
        e = make(2048, [60, 40, 40, 40, 40, 40, 40, 40, 60], seed=4)
        scale = 2.0 ** 40
        X = np.array([[0.5, -1.0, 0.2, 0.1], [1.5, 0.25, -0.3, 0.4], [-0.75, 0.5, 0.6, -0.2]])
        w = np.array([0.3, -0.6, 0.5, 0.25])
        feats = [e["enc"].encrypt(e["encoder"].encode(r, scale)) for r in X]
        cw = e["enc"].encrypt(e["encoder"].encode(w, scale))
        enc_a = S.Encryptor(e["ctx"], e["kg"].public_key(), seed=9)
        enc_b = S.Encryptor(e["ctx"], e["kg"].public_key(), seed=9)
        p_serial = alg.predict_cipher_weights(e["ev"], e["encoder"], enc_a, feats, cw, 4, scale, e["gk"], e["rk"])
        p_shard = par.predict_cipher_weights_sharded(e["ev"], e["encoder"], enc_b, feats, cw, 4, scale, e["gk"], e["rk"])
        lr_same = bool((np.asarray(p_serial.data) == np.asarray(p_shard.data)).all())
        q.put((rank, mm_same, mm_val, lr_same))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_matrix_product_and_lr_predict_world2(world):
    """world 8 (the node the scaling run uses): four Step-2 transforms and three observation rows over eight ranks -- most
    ranks own nothing in one or the other and contribute the zero ciphertext"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_matmul_lr, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = []
    for _ in procs:
        try:
            out.append(q.get(timeout=240))
        except Exception:
            break
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.terminate()
    assert len(out) == world and all(p.exitcode == 0 for p in procs), "a worker failed (see its traceback above)"
    for rank, mm_same, mm_val, lr_same in sorted(out):
        assert mm_same, f"rank {rank}: sharded matrix product differs from the serial one"
        assert mm_val
        assert lr_same, f"rank {rank}: row-sharded LR predict differs from the serial one"


def _worker_sparse_matmul(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import seal_fyp_logistic_regression_amd.seal as S
        from seal_fyp_logistic_regression_amd import algorithms as alg
        from seal_fyp_logistic_regression_amd import parallel as par
        from tests.test_host_api_cpu import make
        e = make(2048, [60, 40, 40, 40, 40, 60], seed=8)
        rng = np.random.default_rng(3)
        A, B = rng.uniform(-1, 1, (n, n)), rng.uniform(-1, 1, (n, n))
        scale = 2.0 ** 40
        sig, tau, phi, psi = alg.matmul_permutation_diagonals(n)
        enc = lambda dd: {l: e["encoder"].encode(v, scale) for l, v in dd.items()}
        mk = lambda seed, M: S.Encryptor(e["ctx"], e["kg"].public_key(), seed=seed).encrypt(e["encoder"].encode(M.reshape(-1), scale))
        args = (mk(7, A), mk(8, B), n, enc(sig), enc(tau), [enc(x) for x in phi], [enc(x) for x in psi], e["gk"])
        serial = alg.cc_matrix_multiplication_sparse(e["ev"], *args)
        res = []
        for step1 in (False, True):
            sh = par.cc_matrix_multiplication_sparse_sharded(e["ev"], *args, shard_step1=step1)
            same = bool((np.asarray(serial.data) == np.asarray(sh.data)).all()) and sh.size() == 3 and \
                sh.parms_id() == serial.parms_id() and sh.scale == serial.scale
            got = e["encoder"].decode(e["dec"].decrypt(sh))[:n * n].real.reshape(n, n)
            res.append((same, bool(np.allclose(got, A @ B, atol=1e-3))))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,world", [(3, 2), (2, 2), (3, 8)])
def test_sharded_sparse_matrix_product_world2(n, world):
    """config 5's form of the matrix product (non-zero diagonals only) with Step 2 split by k over two ranks -- n = 2
    leaves rank 1 without a unit -- with sigma / tau replicated and diagonal-sharded: bits of the serial sparse product."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sparse_matmul, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = []
    for _ in procs:
        try:
            out.append(q.get(timeout=240))
        except Exception:
            break
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.terminate()
    assert len(out) == world and all(p.exitcode == 0 for p in procs), "a worker failed (see its traceback above)"
    for rank, res in sorted(out):
        for step1, (same, val) in zip((False, True), res):
            assert same, f"rank {rank}, shard_step1={step1}: sharded sparse product differs from the serial one"
            assert val


def test_comm_decision_cache_is_tied_to_the_group_object():
    """the collective 'engine communicator or torch' decision is cached per group -- keyed on the object's id, so an
    entry must stop counting once that id belongs to another object."""
    from seal_fyp_logistic_regression_amd import parallel as par

    class Group:
        pass

    a = Group()
    par._decision_set(a, 2, True)
    assert par._decision_get(a, 2) is True and par._decision_get(a, 4) is None
    b = Group()  # an id reused after `a` died: modelled by moving a's entry under b's id
    par._comm_decisions[(id(b), 2)] = par._comm_decisions.pop((id(a), 2))
    assert par._decision_get(b, 2) is None and (id(b), 2) not in par._comm_decisions
    par._decision_set(b, 2, False)
    assert par._decision_get(b, 2) is False
    par._decision_drop(b, 2)
    assert par._decision_get(b, 2) is None

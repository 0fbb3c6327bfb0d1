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


@pytest.mark.parametrize("d", [8, 1])
def test_sharded_linear_transform_world2(d):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, d, q)) for r in range(2)]
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

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
    r, _ = _bench({}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "512", "--cpu-seconds", "0", "--lt", "16")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["value"] > 0 and line["verified"] is True
    d16 = line["lt_sharded"]["d16"]
    assert d16["bits_equal_serial"] is True and d16["decrypts_to_Mv"] is True
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1


def test_bench_two_ranks_a_dead_rank_ends_the_run_non_zero():
    """Rank 1 dies right after the rendezvous (HEFX_BENCH_FAIL_RANK, a test hook in bench.py): rank 0 is then blocked in
    its first barrier for ever -- the launcher must notice, end it after its grace period and exit non-zero, well inside
    the time a driver would wait."""
    r, dt = _bench({"HEFX_BENCH_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "512",
                   "--cpu-seconds", "0", "--lt", "16", timeout=300)
    assert r.returncode != 0
    assert dt < 180, f"the launcher took {dt:.0f} s to give up on a dead rank"
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")], "no result line from a failed run"

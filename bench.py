#!/usr/bin/env python3
"""bench.py -- CKKS rotate + multiply_plain throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of B independent synthetic ciphertexts:
out_i = multiply_plain(rotate_vector(ct_i, 1), pt_i) with a directly keyed step (exactly one key switch),
at the top data level of configs[2]'s parameter set C3 (N=16384, primes {60,40,40,40,40,60}, L=5) -- the
configuration the metric is quoted on.  Inputs (ciphertexts, plaintexts, the Galois key) are resident in
HBM before the timed region.  With N>1 ranks each rank owns its own batch (independent units, no data-path
collective) -> weak scaling; value is the whole-job aggregate.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SETS = {
    # SURVEY.md Appendix B
    "C2": (8192, [0xffffffffffe8001, 0xfffff4c001, 0xfffffdc001, 0xfffffffffffc001]),
    "C3": (16384, [0xffffffffffd8001, 0xffffb20001, 0xffffc40001, 0xffffca8001, 0xffffe80001,
                   0xffffffffffe8001]),
    "C4": (16384, [0xffffffffffd8001, 0xffff940001, 0xffffa78001, 0xffffaf8001, 0xffffb20001, 0xffffc40001,
                   0xffffca8001, 0xffffe80001, 0xffffffffffe8001]),
    "C5": (32768, [0xfffffffff840001, 0xffff940001, 0xffffb20001, 0xffffc40001, 0xffffe80001, 0xffffffffffc0001]),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_bytes_per_op(N: int, L: int) -> int:
    """read ct (2LN) + key (2L(L+1)N) + pt (LN) + write ct (2LN) words = 8*N*L*(2L+7) B."""
    return 8 * N * L * (2 * L + 7)


def synth(rng, primes, N, *shape):
    import numpy as np
    out = np.empty(shape + (N,), dtype=np.uint64)
    for idx in np.ndindex(*shape):
        out[idx] = rng.integers(0, primes[idx[-1]], N, dtype=np.uint64)
    return out


def cpu_baseline(name: str, budget_s: float):
    """Times the CPU oracle (SEAL-3.4.5-algorithm restatement, kind "port") on a bounded sample of the same
    workload: every host core runs independent rotate+multiply_plain ops for ~budget_s seconds."""
    import numpy as np
    from oracle import oracle as O
    N, primes = SETS[name]
    k = len(primes)
    L = k - 1
    cores = os.cpu_count() or 1
    o = O.Oracle(N, primes)
    ct = o.uniform(L, 2, 0x5EA1C0DE)
    pt = o.uniform(L, 1, 0x5EA1C0DE + (1 << 32))[0]
    key = o.uniform(k, 2 * L, 0x6A1015).reshape(L, 2, k, N)
    o.rotate_mulplain(ct, 3, key, pt)  # warm-up
    t0 = time.perf_counter()
    o.rotate_mulplain(ct, 3, key, pt)
    one = time.perf_counter() - t0
    counts = [0] * cores
    deadline = [0.0]

    def work(i):
        while time.perf_counter() < deadline[0]:  # time-bounded: all-core contention is not predictable
            o.rotate_mulplain(ct, 3, key, pt)  # ctypes releases the GIL
            counts[i] += 1

    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    t0 = time.perf_counter()
    deadline[0] = t0 + budget_s
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return {
        "value": sum(counts) / dt,
        "unit": "rotate+multiply_plain ops/s",
        "cores": cores,
        "kind": "port",
        "single_thread_value": 1.0 / one,
        "sample": f"{sum(counts)} ops of the bench workload ({name}: N={N}, L={L}) on {cores} threads in "
                  f"{dt:.1f} s (time-bounded); SEAL-3.4.5-algorithm CPU restatement (oracle/ckks_oracle.c), "
                  "real SEAL is not installable offline",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1152, help="independent ciphertexts per step per GPU")
    ap.add_argument("--set", default="C3", choices=sorted(SETS))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 disables)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    # one rank per GPU.  HEFX_BENCH_BACKEND=gloo (development only) lets several ranks share the GPUs that exist, to
    # exercise the multi-rank code path on a one-GPU box; the driver's runs use RCCL ("nccl").
    backend = os.environ.get("HEFX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from seal_fyp_logistic_regression_amd import Engine

    N, primes = SETS[args.set]
    k = len(primes)
    L = k - 1
    B = args.batch
    e = Engine(N, primes, device=local_rank)
    rng = np.random.default_rng(0x5EA1C0DE + rank)
    # slabs: one allocation per tensor class, items are views (what a pooling allocator hands out)
    big_ct = e.empty(B, 2, L, N)
    big_pt = e.empty(B, L, N)
    big_out = e.empty(B, 2, L, N)
    GEN = 32  # generate/upload in groups to bound host memory
    for base in range(0, B, GEN):
        cnt = min(GEN, B - base)
        big_ct.view(base * 2 * L * N, (cnt, 2, L, N)).upload(synth(rng, primes, N, cnt, 2, L))
        big_pt.view(base * L * N, (cnt, L, N)).upload(synth(rng, primes, N, cnt, L))
    key = e.to_device(synth(np.random.default_rng(0x6A1015), primes, N, L, 2, k))
    cts = [big_ct.view(i * 2 * L * N, (2, L, N)) for i in range(B)]
    pts = [big_pt.view(i * L * N, (L, N)) for i in range(B)]
    outs = [big_out.view(i * 2 * L * N, (2, L, N)) for i in range(B)]
    elts = [3] * B  # galois_elt_from_step(1) = 3
    keys = [key] * B

    def step():
        e.rotate_multiply_plain_batch(L, cts, elts, keys, pts, outs)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ev0, ev1 = e.event(), e.event()
    barrier()
    t0 = time.perf_counter()
    e.event_record(ev0)
    for _ in range(args.steps):
        step()
    e.event_record(ev1)
    barrier()
    dt = time.perf_counter() - t0
    gpu_ms = e.event_elapsed_ms(ev0, ev1)  # HIP events on the stream the launches are issued on
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # per-launch-kind durations: a short profiled pass (serial, HIP events between the five launches)
    e.profile_begin()
    step()
    stage_ms, nchunks = e.profile_end()

    if rank == 0:
        bytes_op = algorithmic_bytes_per_op(N, L)
        total_ops = B * args.steps * world
        value = total_ops / dt
        # roofline of the path on THIS rank: algorithmic bytes of one step / HIP-event time of one step
        step_ms = gpu_ms / args.steps
        achieved = B * bytes_op / (step_ms * 1e-3) / 1e9
        # HBM traffic: not measurable live; taken from the committed rocprofv3 --pmc passes of this same command
        # (FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note, WRITE_SIZE as is), scaled to one step.
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_bench_pmc_traffic.json")
        if args.set == "C3" and os.path.exists(pmc):
            try:
                pb = json.load(open(pmc))["per_op_bytes"]
                traffic = (pb["fetch_x2"] + pb["write"]) * B / 1e9  # GB per step, same unit basis as achieved*time
                traffic_src = "profiles/r01_bench_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes); GB per step"
            except Exception:
                pass
        dom = max(stage_ms, key=stage_ms.get)
        tot = sum(stage_ms.values())
        line = {
            "metric": "CKKS ciphertext rotate+plain-mult ops/sec at N=16384; HBM GB/s vs roofline",
            "value": value,
            "unit": "rotate+multiply_plain ops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.set}: N={N}, coeff_modulus bits "
                            f"{[p.bit_length() for p in primes]}, level L={L} (k={k}); per step and per GPU "
                            f"{B} independent rotate_vector(step=1, direct Galois key)+multiply_plain, "
                            "uniform random residues, inputs resident in HBM",
                "batch_per_gpu": B,
                "parallelism": f"{world} x independent ciphertext batches (no data-path collective)",
                "algorithmic_bytes_per_op": bytes_op,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_GB_per_step": B * bytes_op / 1e9,  # same basis as `traffic` (one step = one launch sequence)
                "launch": "one step = the 6-kernel key-switch sequence over the whole batch "
                          f"({nchunks} chunks); achieved = {B} ops x {bytes_op} B / {step_ms:.3f} ms (HIP events)",
                "dominant_kernel": dom,
                "kernel_avg_us": {kname: ms / max(nchunks, 1) * 1e3 for kname, ms in stage_ms.items()},
                "kernel_share": {kname: ms / tot for kname, ms in stage_ms.items()},
                "expected_first_limiter": "integer VALU (64-bit modmul emulated with v_mad_u64_u32)",
            },
        }
        if args.cpu_seconds > 0 and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(args.set, args.cpu_seconds)
                line["gpu_over_cpu_single_thread"] = value / line["cpu_baseline"]["single_thread_value"]
            except Exception as ex:  # the oracle is test infrastructure; never let it break the GPU number
                line["cpu_baseline"] = {"value": None, "error": repr(ex)}
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

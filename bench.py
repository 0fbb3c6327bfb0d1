#!/usr/bin/env python3
"""bench.py -- CKKS rotate + multiply_plain throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of B independent synthetic ciphertexts:
out_i = multiply_plain(rotate_vector(ct_i, 1), pt_i) with a directly keyed step (exactly one key switch), at the top
data level of configs[2]'s parameter set C3 (N=16384, primes {60,40,40,40,40,60}, L=5) -- the configuration the
metric is quoted on.  Inputs (ciphertexts, plaintexts, the Galois key) are resident in HBM before the timed region.
With N>1 ranks each rank owns its own batch (independent units, no data-path collective) -> weak
scaling; `value` is the whole-job aggregate.

Launch: `python bench.py --gpus N` starts N ranks ITSELF (child processes, created before this process touches the
GPU) when it is not already running under a launcher; under `python -m torch.distributed.run --nproc-per-node N`
(RANK / LOCAL_RANK / WORLD_SIZE in the environment) it is one of the ranks.  One rank per GPU, RCCL ("nccl").
`HEFX_BENCH_BACKEND=gloo` (development only) lets several ranks share the GPUs that exist.

Besides the headline, the same JSON line carries:
  verified      outs[0] and outs[B-1] of the timed loop compared word for word with the CPU oracle
  variants      the same batch with 16 distinct steps / Galois keys round-robin (the headline shares one key)
  lt_sharded    Linear_Transform_Plain (helper.h:237-262) at C3, d = 16 and 512, diagonals sharded over the ranks with one
                RCCL SUM(uint64) all-reduce per transform (parallel.linear_transform_plain_sharded); the sharded bits are
                asserted equal to the serial ones in the run
  roofline.valu the integer / FP64 butterfly issue-cycle bound next to the HBM one
  batch_ladder  the unit at B = 1, 16, 256, 1024: us per op and roofline fraction; chain_level_us: one
                rotate-by-1 + add level of the dot products' chains at n = 1 / 8, L = 2
  composites    SURVEY 8(d)'s scaling workloads: matmul_C3_n4, matmul_C5_n8 (dense, every diagonal), lr_rows_2000x8 --
                serial ms, and with more than one rank the sharded forms with bits asserted equal to the serial ones

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
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
# VALU issue model (profiles/r01_valu_issue_rates.txt, tools/ubench_valu.hip): cycles one SIMD
# spends per wave-butterfly (64 butterflies) -- Harvey/Shoup on 60-bit primes vs the exact-FMA FP64 butterfly on
# primes < 2^41 -- and the machine: 256 CUs x 4 SIMDs at the 2.4 GHz peak engine clock.
# Integer butterfly as built now (21 instructions, 7 of them v_mad_u64_u32): 81 cycles measured in isolation, 105 for the
# form the compiler makes of x*w - h*q (tools/ubench_bfly.hip, profiles/r02_butterfly_ubench.txt; the mid-round-2 code was
# the latter); FP64 butterfly: 8 instructions at 4.75 = 38, priced at 40.
VALU_CYC_INT, VALU_CYC_F64 = 81.0, 40.0
SIMDS, CLOCK_HZ = 1024, 2.4e9


def algorithmic_bytes_per_op(N: int, L: int) -> int:
    """read ct (2LN) + key (2L(L+1)N) + pt (LN) + write ct (2LN) words = 8*N*L*(2L+7) B."""
    return 8 * N * L * (2 * L + 7)


def valu_bound_ops_per_s(N: int, primes, L: int) -> dict:
    """Butterfly-only VALU ceiling of one rotate+multiply_plain: (L+1)(L+2) transforms of (N/2) log2 N butterflies,
    each priced by the arithmetic policy of the modulus it runs in (digit INTT: q_i; digit NTT: target modulus;
    mod-down INTT: P; mod-down NTT: q_j)."""
    k = len(primes)
    f64 = [p < (1 << 41) for p in primes]
    n_int = n_f64 = 0
    for i in range(L):                       # inverse transform of digit i
        n_f64, n_int = (n_f64 + 1, n_int) if f64[i] else (n_f64, n_int + 1)
    for i in range(L):                       # digit i -> every other modulus of the extended basis
        for m in list(range(L)) + [k - 1]:
            if m == i:
                continue
            n_f64, n_int = (n_f64 + 1, n_int) if f64[m] else (n_f64, n_int + 1)
    n_f64, n_int = (n_f64 + 2, n_int) if f64[k - 1] else (n_f64, n_int + 2)      # INTT_P of both accumulators
    for j in range(L):                       # remainder -> data prime j, both polys
        n_f64, n_int = (n_f64 + 2, n_int) if f64[j] else (n_f64, n_int + 2)
    wave_bf = (N // 2) * (N.bit_length() - 1) / 64.0
    cycles = wave_bf * (n_int * VALU_CYC_INT + n_f64 * VALU_CYC_F64)
    return {"int_transforms": n_int, "f64_transforms": n_f64, "simd_cycles_per_op": cycles,
            "peak_ops_per_s": SIMDS * CLOCK_HZ / cycles}


def cpu_baseline(name: str, budget_s: float):
    """Times the CPU oracle (SEAL-3.4.5-algorithm restatement, kind "port") on a bounded sample of the same
    workload: every host core runs independent rotate+multiply_plain ops for ~budget_s seconds."""
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
    reps = 0
    while time.perf_counter() - t0 < min(1.0, budget_s / 4):
        o.rotate_mulplain(ct, 3, key, pt)
        reps += 1
    one = (time.perf_counter() - t0) / reps
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
                  f"{dt:.1f} s (time-bounded) + {reps} ops on one thread; SEAL-3.4.5-algorithm CPU restatement "
                  "(oracle/ckks_oracle.c), real SEAL is not installable offline",
    }


class BoardSampler:
    """Package power, engine clock and busy percentage of one GPU, read with rocm-smi about twice a second on a side
    thread while the key-switch passes run (headline + variants): corroborates from outside that the timed region keeps
    the device busy, and records the clock the power governor grants under this load (profiles/r03_clock_and_power_*).
    Reads sysfs through a child process; touches neither the HIP context nor the streams.  Absent tool -> no samples."""

    def __init__(self, gpu_index: int):
        self.idx, self.samples, self._stop, self._th = gpu_index, [], threading.Event(), None

    def _run(self):
        import re
        exe = "/opt/rocm/bin/rocm-smi"
        if not os.path.exists(exe):
            return
        pat = {"sclk_mhz": re.compile(r"GPU\[%d\].*sclk clock level.*\((\d+)Mhz\)" % self.idx),
               "package_power_w": re.compile(r"GPU\[%d\].*Power \(W\):\s*([0-9.]+)" % self.idx),
               "gpu_use_pct": re.compile(r"GPU\[%d\].*GPU use \(%%\):\s*([0-9.]+)" % self.idx)}
        # the child must not carry a profiler's preloaded tool library: under `rocprofv3 --pmc` that library initialises the
        # GPU in every process it is loaded into, and rocm-smi (an env -> python3 script) then re-executes itself
        env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF"))}
        while not self._stop.is_set():
            try:
                out = subprocess.run([exe, "--showclocks", "--showpower", "--showuse"], capture_output=True, text=True,
                                     timeout=10, env=env).stdout
            except Exception:
                return
            rec = {}
            for k, rx in pat.items():
                m = rx.search(out)
                if m:
                    rec[k] = float(m.group(1))
            if len(rec) == 3:
                self.samples.append(rec)
            self._stop.wait(0.3)

    def start(self):
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()

    def stop(self):
        self._stop.set()
        if self._th:
            self._th.join(timeout=15)
        busy = [r for r in self.samples if r["gpu_use_pct"] >= 90.0]
        if not busy:
            return {"samples": len(self.samples), "note": "no sample fell inside the busy period (short run) or rocm-smi is absent"}
        avg = lambda k: sum(r[k] for r in busy) / len(busy)
        return {"samples": len(busy), "package_power_w": avg("package_power_w"), "sclk_mhz": avg("sclk_mhz"),
                "gpu_use_pct": avg("gpu_use_pct"),
                "source": "rocm-smi --showclocks --showpower --showuse on a side thread during the key-switch passes; "
                          "samples with GPU use >= 90 %"}


# ------------------------------------------------------------------------------------------------------------------
# launcher: --gpus N outside a distributed launcher starts the N ranks itself
# ------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(args) -> int:
    """Parent of a self-launched multi-rank run.  It never initialises the GPU (device_count() only counts, per the
    image notes) and never replaces itself: the ranks are ordinary child processes and their exit codes are relayed."""
    import torch
    n = args.gpus
    have = torch.cuda.device_count()
    backend = os.environ.get("HEFX_BENCH_BACKEND", "nccl")
    if backend == "nccl" and have < n:
        print(f"bench.py: --gpus {n} needs {n} HIP devices, this machine shows {have} "
              "(RCCL wants one device per rank; HEFX_BENCH_BACKEND=gloo shares devices for development)",
              file=sys.stderr, flush=True)
        return 2
    if have < 1:
        print("bench.py needs a HIP device: the engine has no CPU fallback", file=sys.stderr, flush=True)
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # poll: a rank that dies before or inside a barrier / collective leaves its peers blocked in RCCL for ever, so the
    # first non-zero exit ends the others after a short grace period and its code is the run's
    rc = 0
    live = list(procs)
    failed_at = None
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = rc or abs(code) or 1
                failed_at = failed_at or time.monotonic()
        if failed_at is not None and live and time.monotonic() - failed_at > 15.0:
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    return rc


# ------------------------------------------------------------------------------------------------------------------
# the sharded linear transform (SURVEY 8d "Scaling runs", 8e-i)
# ------------------------------------------------------------------------------------------------------------------
def lt_sharded_bench(local_rank: int, world: int, dims, reps: int, direct_d: int = 0, use_pg: bool = False):
    """Linear_Transform_Plain at C3 with the reference's default (power-of-two, NAF-expanded) Galois keys: the serial
    one-call form on this rank, and -- with more than one rank -- the diagonal-sharded form with its one all-reduce.
    Same seeds on every rank => same keys, ciphertext and diagonals everywhere (what a broadcast at setup would give)."""
    import numpy as np
    import torch.distributed as dist
    from seal_fyp_logistic_regression_amd import algorithms as alg
    from seal_fyp_logistic_regression_amd import parallel as par
    from seal_fyp_logistic_regression_amd import seal as S
    N, primes = SETS["C3"]
    parms = S.EncryptionParameters("ckks")
    parms.set_poly_modulus_degree(N)
    parms.set_coeff_modulus(primes)
    ctx = S.SEALContext.Create(parms, device=local_rank)
    kg = S.KeyGenerator(ctx, 0xC3)
    enc, dec = S.Encryptor(ctx, kg.public_key(), 0xC4), S.Decryptor(ctx, kg.secret_key())
    encoder, ev, gk = S.CKKSEncoder(ctx), S.Evaluator(ctx), kg.galois_keys()
    eng = ctx.backend.engine
    bits = lambda c: ctx.backend.to_host(c.data)
    scale = 2.0 ** 40
    out = {}
    for d in dims:
        rng = np.random.default_rng(1000 + d)
        M, v = rng.uniform(-1, 1, (d, d)), rng.uniform(-1, 1, d)
        diags = encoder.encode_many(list(alg.get_all_diagonals(M)), scale)
        ct = enc.encrypt(encoder.encode(v, scale))

        def timed(fn):
            for _ in range(3):  # warm-up: scratch growth, Galois tables, pool slabs -- and the clock, which the host-side key
                r = fn()        # generation before this leg lets drop (one 2 ms call does not bring it back)
            eng.sync()
            if use_pg:
                dist.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                r = fn()
            eng.sync()
            return r, (time.perf_counter() - t0) / reps * 1e3

        serial, serial_ms = timed(lambda: alg.linear_transform_plain(ev, ct, diags, gk))
        ok = bool(np.allclose(encoder.decode(dec.decrypt(serial))[:d].real, M @ v, atol=1e-3 * d))
        rec = {"serial_ms": serial_ms, "decrypts_to_Mv": ok, "key_switches_serial": alg_key_switches(ev, d, gk)}
        # what the transform really executes with the reference's default keys (linear_transformation2.cpp:239): the engine's
        # own counters around ONE more call -- key switches after prefix sharing of the NAF forest (the op-by-op loop runs
        # `key_switches_serial`), how many ran exactly hoisted, launch sequences -- and the algorithmic-byte rate of the
        # executed key switches, each pricing its own key read (8 N L (2L+7) bytes, the basis of roofline.frac)
        s0 = eng.ks_stats()
        alg.linear_transform_plain(ev, ct, diags, gk)
        s1 = eng.ks_stats()
        Lr = len(primes) - 1
        ex = s1["key_switches"] - s0["key_switches"]
        rec.update({"key_switches_executed": ex, "hoisted_items": s1["hoisted"] - s0["hoisted"],
                    "launch_sequences": s1["chunks"] - s0["chunks"], "key_switches_per_s": ex / (serial_ms * 1e-3),
                    "frac_of_8TBps_algorithmic": ex / (serial_ms * 1e-3) * 8 * N * Lr * (2 * Lr + 7) / 8e12})
        if use_pg:
            par.ENGINE_COMM = "off"   # this leg: the exchange through torch.distributed (in place on the payload)
            sharded, sharded_ms = timed(lambda: par.linear_transform_plain_sharded(ev, ct, diags, gk))
            same = bool((bits(serial) == bits(sharded)).all())
            t = [sharded_ms, 1.0 if same else 0.0]
            import torch
            tt = torch.tensor(t, dtype=torch.float64)
            if dist.get_backend() == "nccl":
                tt = tt.cuda()
            mx, mn = tt.clone(), tt.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            dist.all_reduce(mn, op=dist.ReduceOp.MIN)
            rec.update({"sharded_ms": float(mx[0].item()), "speedup_vs_1": serial_ms / float(mx[0].item()),
                        "bits_equal_serial": bool(mn[1].item() == 1.0),
                        "allreduce_bytes": 2 * (len(primes) - 1) * N * 8})
            assert rec["bits_equal_serial"], f"sharded linear transform (d={d}) differs from the serial one"
            if dist.get_backend() == "nccl" and os.environ.get("HEFX_BENCH_C_ABI_COMM") == "1":
                # opt-in (a second communicator that could not be exercised on a one-GPU box must not be able to stall
                # the scaling run): the same transform with the exchange behind the C-ABI (hefx_allreduce_sum: RCCL called by libhefx
                # itself, in place on the payload) instead of torch.distributed; reported next to it, never fatal
                try:
                    par.ENGINE_COMM = "on"
                    par.init_engine_comm(ev)
                    via, via_ms = timed(lambda: par.linear_transform_plain_sharded(ev, ct, diags, gk))
                    tt = torch.tensor([via_ms, 1.0 if bool((bits(serial) == bits(via)).all()) else 0.0],
                                      dtype=torch.float64).cuda()
                    mx, mn = tt.clone(), tt.clone()
                    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
                    dist.all_reduce(mn, op=dist.ReduceOp.MIN)
                    rec["c_abi_allreduce"] = {"sharded_ms": float(mx[0].item()), "bits_equal_serial": bool(mn[1].item() == 1.0)}
                except Exception as ex:
                    rec["c_abi_allreduce"] = {"error": repr(ex)[:300]}
                finally:
                    try:
                        eng.comm_destroy()
                    except Exception:
                        pass
        out[f"d{d}"] = rec
    # Linear_Transform_Plain with a DIRECT Galois key per step (keygen.galois_keys(steps)): d - 1 key switches, each with
    # its own 7.9 MB key -- at d = 512 that is 4 GB of keys, the key-streaming case of SURVEY H5 (bit-exact to the
    # op-by-op sequence with those keys; serial, rank-local)
    if direct_d:
        d = direct_d
        rng = np.random.default_rng(2000 + d)
        M, v = rng.uniform(-1, 1, (d, d)), rng.uniform(-1, 1, d)
        diags = encoder.encode_many(list(alg.get_all_diagonals(M)), scale)
        ct = enc.encrypt(encoder.encode(v, scale))
        gk_direct = kg.galois_keys([-d] + list(range(1, d)))
        for _ in range(3):  # (warm-up as above: 4 GB of keys were just generated with the GPU mostly idle)
            r = alg.linear_transform_plain(ev, ct, diags, gk_direct)
        eng.sync()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = alg.linear_transform_plain(ev, ct, diags, gk_direct)
            eng.sync()
            ts.append(time.perf_counter() - t0)
        ms = sorted(ts)[len(ts) // 2] * 1e3
        ks = alg_key_switches(ev, d, gk_direct)
        kb = (len(primes) - 1) * 2 * len(primes) * N * 8
        out[f"direct_keys_d{d}"] = {
            "serial_ms": ms, "key_switches": ks, "galois_keys": len(gk_direct.keys), "key_bytes": len(gk_direct.keys) * kb,
            "key_switches_per_s": ks / (ms * 1e-3), "key_GBps": ks * kb / (ms * 1e-3) / 1e9,
            # every rotation reads its own key: the regime roofline.frac prices (8 N L (2L+7) algorithmic bytes per key switch)
            "frac_of_8TBps_algorithmic": ks / (ms * 1e-3) * 8 * N * (len(primes) - 1) * (2 * (len(primes) - 1) + 7) / 8e12,
            "note": "d-1 rotations of ONE ciphertext, a direct Galois key per step: run exactly hoisted (SEAL's bits; "
                    "DESIGN.md 'Exact hoisting'), wall time per call incl. the host side",
            "decrypts_to_Mv": bool(np.allclose(encoder.decode(dec.decrypt(r))[:d].real, M @ v, atol=1e-3 * d))}
        del gk_direct
    return out


def composites_bench(local_rank: int, world: int, which, use_pg: bool = False, reps: int = 5):
    """SURVEY 8(d)'s scaling workloads, compute phase only (keys generated, inputs encoded / encrypted beforehand):
      matmul_C3_n4    CC_Matrix_Multiplication (matrix_multiplication.cpp:11-132) n = 4 at C3: 8 transforms x 16 diagonals
      matmul_C5_n8    the same at n = 8, N = 32768 (matrix_mult_benchmark.cpp:13-71; config 5 as SURVEY App. B reads it:
                      64 x 64 U matrices): 16 transforms x 64 diagonals, every diagonal, the 1e-8 epsilons
      lr_rows_2000x8  predict_cipher_weights (logistic_regression_ckks.cpp:208-266) over 2000 rows x 8 weights at C4
    Serial ms on this rank (median of `reps` calls incl. the host side); with more than one rank also the sharded form of
    parallel.py (max over ranks) with its bits asserted equal to the serial ones.  Same seeds on every rank."""
    import numpy as np
    import torch.distributed as dist
    from seal_fyp_logistic_regression_amd import algorithms as alg
    from seal_fyp_logistic_regression_amd import parallel as par
    from seal_fyp_logistic_regression_amd import seal as S
    out = {}

    def context(setname):
        N, primes = SETS[setname]
        parms = S.EncryptionParameters("ckks")
        parms.set_poly_modulus_degree(N)
        parms.set_coeff_modulus(primes)
        ctx = S.SEALContext.Create(parms, device=local_rank)
        kg = S.KeyGenerator(ctx, 0xC5)
        return ctx, kg, S.Decryptor(ctx, kg.secret_key()), S.CKKSEncoder(ctx), S.Evaluator(ctx)

    def measure(eng, serial_fn, sharded_fn, bits):
        for _ in range(2):
            r = serial_fn()
        eng.sync()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = serial_fn()
            eng.sync()
            ts.append(time.perf_counter() - t0)
        s0 = eng.ks_stats()
        serial_fn()
        s1 = eng.ks_stats()
        ms = sorted(ts)[len(ts) // 2] * 1e3
        ks = s1["key_switches"] - s0["key_switches"]
        rec = {"serial_ms": ms, "key_switches_executed": ks, "launch_sequences": s1["chunks"] - s0["chunks"],
               "key_switches_per_s": ks / (ms * 1e-3)}
        if use_pg:
            import torch
            par.ENGINE_COMM = "off"
            for _ in range(2):
                sh = sharded_fn()
            eng.sync()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                sh = sharded_fn()
            eng.sync()
            sms = (time.perf_counter() - t0) / reps * 1e3
            same = bool((bits(r) == bits(sh)).all())
            tt = torch.tensor([sms, 1.0 if same else 0.0], dtype=torch.float64)
            if dist.get_backend() == "nccl":
                tt = tt.cuda()
            mx, mn = tt.clone(), tt.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            dist.all_reduce(mn, op=dist.ReduceOp.MIN)
            rec.update({"sharded_ms": float(mx[0].item()), "speedup_vs_1": ms / float(mx[0].item()),
                        "bits_equal_serial": bool(mn[1].item() == 1.0)})
            assert rec["bits_equal_serial"], "a sharded composite differs from the serial one"
        return r, rec

    for name in which:
        if name.startswith("matmul_"):
            setname, n = name.split("_")[1], int(name.split("_n")[1])
            ctx, kg, dec, encoder, ev = context(setname)
            eng = ctx.backend.engine
            N = ctx.N
            Lr = len(SETS[setname][1]) - 1
            enc, gk = S.Encryptor(ctx, kg.public_key(), 0xC6), kg.galois_keys()
            rng = np.random.default_rng(500 + n)
            A, B = rng.uniform(-1, 1, (n, n)), rng.uniform(-1, 1, (n, n))
            scale = 2.0 ** 40
            dense = lambda U: encoder.encode_many(list(alg.get_all_diagonals(U) + 1e-8), scale)   # matrix_multiplication.cpp:239-297
            Us, Ut, V, W = alg.matmul_permutation_matrices(n)
            ctA, ctB = enc.encrypt(encoder.encode(A.reshape(-1), scale)), enc.encrypt(encoder.encode(B.reshape(-1), scale))
            args = (ctA, ctB, n, dense(Us), dense(Ut), [dense(x) for x in V], [dense(x) for x in W], gk)
            bits = lambda c: ctx.backend.to_host(c.data)
            r, rec = measure(eng, lambda: alg.cc_matrix_multiplication(ev, *args),
                             lambda: par.cc_matrix_multiplication_sharded(ev, *args), bits)
            got = encoder.decode(dec.decrypt(r))[:n * n].real.reshape(n, n)
            rec.update({"decrypts_to_AB": bool(np.allclose(got, A @ B, atol=1e-3)), "max_abs_err": float(np.abs(got - A @ B).max()),
                        "rotations_op_by_op": 2 * n * n * n, "plaintext_diagonals": 2 * n * n * n, "plaintext_bytes": 2 * n * n * n * Lr * N * 8,
                        "frac_of_8TBps_algorithmic": rec["key_switches_per_s"] * 8 * N * Lr * (2 * Lr + 7) / 8e12,
                        "workload": f"{setname}: n={n}, {2 * n} Linear_Transform_Plain x {n * n} diagonals, default power-of-two Galois keys"})
            out[name] = rec
            del args, gk, ctx, kg, enc, ev, encoder, dec
        elif name.startswith("lr_rows_"):
            rows, nw = (int(x) for x in name[len("lr_rows_"):].split("x"))
            ctx, kg, dec, encoder, ev = context("C4")
            eng = ctx.backend.engine
            gk, rk = kg.galois_keys(), kg.relin_keys()
            enc = S.Encryptor(ctx, kg.public_key(), 0xC6)
            rng = np.random.default_rng(700 + rows)
            X, w = rng.uniform(-1, 1, (rows, nw)), rng.uniform(-0.5, 0.5, nw)
            scale = 2.0 ** 40
            feats = [enc.encrypt(p) for p in encoder.encode_many(list(X), scale)]
            cw = enc.encrypt(encoder.encode(w, scale))
            bits = lambda c: ctx.backend.to_host(c.data)
            fresh = lambda: S.Encryptor(ctx, kg.public_key(), 0xC7)  # (the sigmoid encrypts a constant: same seed, same bits)
            r, rec = measure(eng, lambda: alg.predict_cipher_weights(ev, encoder, fresh(), feats, cw, nw, scale, gk, rk),
                             lambda: par.predict_cipher_weights_sharded(ev, encoder, fresh(), feats, cw, nw, scale, gk, rk), bits)
            got = encoder.decode(dec.decrypt(r))[:rows].real
            c = alg.SIGMOID_COEFFS[3]
            z = X @ w
            want = c[0] + c[1] * z + c[2] * z ** 2 + c[3] * z ** 3
            # the reference's packing replicates a dot product over slots 0..size: only the first `nw` rows carry theirs in full
            rec.update({"decrypts_to_sigmoid_of_Xw": bool(np.abs(got - want)[:nw].max() < 1e-2),
                        "max_abs_err_first_rows": float(np.abs(got - want)[:nw].max()), "rows": rows, "weights": nw,
                        "workload": f"C4: {rows} rows x (multiply + relinearize + rescale + {nw} sequential rotate-by-1 + add), "
                                    "masks, add_many, degree-3 sigmoid (Horner)"})
            out[name] = rec
            del feats, gk, rk, ctx, kg, enc, ev, encoder, dec
    return out


def alg_key_switches(ev, d, gk) -> int:
    """key switches of one Linear_Transform_Plain with these keys (NAF rule)"""
    return sum(len(ev.rotation_plan(s, gk)) for s in [-d] + list(range(1, d)))


def secondary_bench(name, Engine, local_rank, rank, world, timed, key32, seconds=1.0):
    """the bench unit at parameter set `name`: value (whole job), ms_per_step, roofline.frac, verified"""
    import hashlib
    N, primes = SETS[name]
    k = len(primes)
    L = k - 1
    B = {"C2": 9216, "C3": 4608, "C4": 2304, "C5": 2304}[name]
    e2 = Engine(N, primes, device=local_rank)
    big_ct = e2.sample("uniform", key32("ct:" + name), 1, 2 * B, L, 0)
    big_pt = e2.sample("uniform", key32("pt:" + name), 2, B, L, 0)
    big_out = e2.empty(B, 2, L, N)
    key = e2.sample("uniform", hashlib.sha256(("hefx-bench:key:" + name).encode()).digest(), 3, 2 * L, k, 0).view(0, (L, 2, k, N))
    cts = [big_ct.view(i * 2 * L * N, (2, L, N)) for i in range(B)]
    pts = [big_pt.view(i * L * N, (L, N)) for i in range(B)]
    outs = [big_out.view(i * 2 * L * N, (2, L, N)) for i in range(B)]
    elts, keys = [3] * B, [key] * B
    prep = e2.prepare_rotate_multiply_plain_batch(L, cts, elts, keys, pts, outs)
    fn = lambda: e2.rotate_multiply_plain_prepared(prep)
    one, _ = timed(fn, 2, 3, e2)
    steps = max(3, int(seconds / max(one / 2, 1e-6)))
    dt, gpu_ms = timed(fn, steps, 0, e2)
    verified = None
    if rank == 0:
        try:
            from oracle import oracle as O
            o = O.Oracle(N, primes)
            hk = key.download()
            verified = all(bool((outs[i].download() == o.rotate_mulplain(cts[i].download(), 3, hk, pts[i].download())).all())
                           for i in (0, B // 2 + 1, B - 1))
        except Exception as ex:
            verified = f"not checked: {ex!r}"
    bytes_op = algorithmic_bytes_per_op(N, L)
    achieved = B * bytes_op / (gpu_ms / steps * 1e-3) / 1e9
    valu = valu_bound_ops_per_s(N, primes, L)
    value = B * steps * world / dt
    return {"workload": f"{name}: N={N}, coeff_modulus bits {[p.bit_length() for p in primes]}, L={L}; {B} independent "
                        "rotate_vector(step=1, direct key)+multiply_plain per step and GPU",
            "value": value, "unit": "rotate+multiply_plain ops/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
            "batch_per_gpu": B, "verified": verified,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_op": bytes_op,
                         "valu_frac": value / world / valu["peak_ops_per_s"]}}


def _hashes():
    """(source hash, library hash) of the engine that runs: written into the line and compared with the hashes the
    profile tools wrote into profiles/*_pmc_traffic.json / *_sq_counters.json -- counters of other kernels are stale"""
    from seal_fyp_logistic_regression_amd import _build
    return _build.source_sha16(), _build.library_sha16()


# ------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4608, help="independent ciphertexts per step per GPU")
    ap.add_argument("--set", default="C3", choices=sorted(SETS))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 disables)")
    ap.add_argument("--lt", default="16,512", help="dimensions of the (sharded) linear-transform leg; '' disables")
    ap.add_argument("--variant-keys", type=int, default=16, help="distinct Galois keys of the secondary pass; 0 disables")
    ap.add_argument("--stream-keys", type=int, default=64, help="distinct Galois keys of the key-streaming pass (> 256 MiB "
                    "of keys at C3); 0 disables")
    ap.add_argument("--key-per-item", type=int, default=1024, help="items of the key-per-item pass (one Galois key each: "
                    "8 GB of keys at 1024, C3); 0 disables")
    ap.add_argument("--sustain", type=float, default=2.5, help="seconds of the second, longer headline pass (0 disables)")
    ap.add_argument("--secondary", default="C2,C4,C5", help="parameter sets of the `secondary` block (north_star: "
                    "poly_modulus_degree in {8192, 16384}); '' disables")
    ap.add_argument("--lt-direct", type=int, default=512, help="dimension of the direct-key Linear_Transform_Plain leg "
                    "(one Galois key per step: 4 GB of keys at d = 512, C3); 0 disables")
    ap.add_argument("--composites", default="matmul_C3_n4,matmul_C5_n8,lr_rows_2000x8", help="SURVEY 8(d)'s scaling workloads "
                    "(serial ms; sharded + bits_equal_serial with more than one rank); '' disables")
    ap.add_argument("--ladder", default="1,16,256,1024", help="batch sizes of the `batch_ladder` block (SURVEY 8d); '' disables")
    ap.add_argument("--quick", action="store_true", help="headline only (A/B and profiler runs): no CPU baseline, variants, "
                    "key-per-item, secondary sets, sustained pass or linear-transform legs")
    args = ap.parse_args()
    if args.quick:
        args.cpu_seconds, args.variant_keys, args.stream_keys, args.key_per_item = 0.0, 0, 0, 0
        args.lt, args.lt_direct, args.secondary, args.sustain = "", 0, "", 0.0
        args.composites, args.ladder = "", ""

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch(args))  # nothing in this process has touched the GPU

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # stdout carries exactly ONE line, the JSON: libraries that write to file descriptor 1 on their own (RCCL's version
    # banner at communicator creation, gloo's connection notes) are sent to stderr for the duration of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks; reporting n_gpus={world}",
              file=sys.stderr, flush=True)

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    # one rank per GPU.  HEFX_BENCH_BACKEND=gloo (development only) lets several ranks share the GPUs that exist, to
    # exercise the multi-rank code path on a one-GPU box; the driver's runs use RCCL ("nccl").
    backend = os.environ.get("HEFX_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"bench.py: {world} ranks need {world} HIP devices, this machine shows {ndev}")
    if backend != "nccl":
        local_rank %= max(ndev, 1)
    torch.cuda.set_device(local_rank)
    # HEFX_BENCH_FORCE_PG=1 (test hook, tests/test_gpu_round5.py): create the process group and take every barrier /
    # all-reduce / sharded branch even at world 1 -- the only way the one-GPU box can execute the torch "nccl" (RCCL) path
    use_pg = world > 1 or os.environ.get("HEFX_BENCH_FORCE_PG") == "1"
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    # test hook (tests/test_gpu_round4.py): this rank dies right after the rendezvous, before the first barrier -- the
    # launcher must then end its blocked peers and report a non-zero exit code instead of hanging
    if os.environ.get("HEFX_BENCH_FAIL_RANK") == str(rank):
        os._exit(7)

    from seal_fyp_logistic_regression_amd import Engine

    N, primes = SETS[args.set]
    k = len(primes)
    L = k - 1
    B = args.batch
    e = Engine(N, primes, device=local_rank)
    # Synthetic inputs: uniform residues drawn ON the device by the engine's counter-mode sampler (ChaCha20 keyed by the
    # seed; hefx_sample_uniform) -- no host generation / PCIe leg, so the batch can be sized for the HBM.
    import hashlib
    key32 = lambda tag: hashlib.sha256(f"hefx-bench:{tag}:{rank}".encode()).digest()
    big_ct = e.sample("uniform", key32("ct"), 1, 2 * B, L, 0)            # [B][2][L][N]
    big_pt = e.sample("uniform", key32("pt"), 2, B, L, 0)                # [B][L][N]
    big_out = e.empty(B, 2, L, N)
    nk = max(1, args.variant_keys)
    big_key = e.sample("uniform", hashlib.sha256(b"hefx-bench:key").digest(), 3, 2 * L * nk, k, 0)   # nk x [L][2][k][N]
    key_words = L * 2 * k * N
    keyv = [big_key.view(i * key_words, (L, 2, k, N)) for i in range(nk)]
    cts = [big_ct.view(i * 2 * L * N, (2, L, N)) for i in range(B)]
    pts = [big_pt.view(i * L * N, (L, N)) for i in range(B)]
    outs = [big_out.view(i * 2 * L * N, (2, L, N)) for i in range(B)]
    elts = [3] * B  # galois_elt_from_step(1) = 3
    keys = [keyv[0]] * B

    # the C argument arrays are built once (Engine.prepare_*): what is timed is the C-ABI call, as a C / C++ caller makes it
    prepared = {"step": e.prepare_rotate_multiply_plain_batch(L, cts, elts, keys, pts, outs)}  # (cleared before the LT leg)

    def step():
        e.rotate_multiply_plain_prepared(prepared["step"])

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup, eng=None):
        eng = eng or e
        for _ in range(warmup):
            fn()
        ev0, ev1 = eng.event(), eng.event()
        barrier()
        t0 = time.perf_counter()
        eng.event_record(ev0)
        for _ in range(steps):
            fn()
        eng.event_record(ev1)
        barrier()
        dt = time.perf_counter() - t0
        gpu_ms = eng.event_elapsed_ms(ev0, ev1)  # HIP events on the stream the launches are issued on
        if use_pg:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, gpu_ms

    board = BoardSampler(local_rank) if rank == 0 else None
    if board:
        board.start()
    dt, gpu_ms = timed(step, args.steps, args.warmup)
    # a second, longer self-timed pass of the SAME step: at least `--sustain` seconds whatever --steps
    # is, so that the board sampler (and anybody watching from outside) sees the device busy more than once and the clock /
    # power it settles at; reported next to the headline, never instead of it
    sustained = None
    if args.sustain > 0:
        ssteps = max(args.steps, int(args.sustain / max(dt / args.steps, 1e-6)) + 1)
        sdt_, sgpu_ = timed(step, ssteps, 0)
        sustained = {"steps": ssteps, "seconds": sdt_, "value": B * ssteps * world / sdt_, "ms_per_step": sdt_ / ssteps * 1e3,
                     "gpu_ms_per_step": sgpu_ / ssteps}

    # SURVEY 8(d)'s batch ladder B in {1, 16, 256, 1024} of the same unit (prepared call + wait, median; on this rank) and the
    # latency of one rotate-by-1 + add level of the dot products' chains (helper.h:472-476) on the LR ring at L = 2 -- the
    # regimes configs 2-4 live in, next to the throughput headline; 0.2 s of GPU time
    batch_ladder, chain_level_us, unprepared = None, None, None
    if args.ladder:
        import statistics
        batch_ladder = []
        bytes_op_ = algorithmic_bytes_per_op(N, L)
        for Bl in [int(x) for x in args.ladder.split(",") if x and int(x) <= B]:
            prep_l = e.prepare_rotate_multiply_plain_batch(L, cts[:Bl], elts[:Bl], keys[:Bl], pts[:Bl], outs[:Bl])
            for _ in range(3):
                e.rotate_multiply_plain_prepared(prep_l)
            e.sync()
            ts = []
            for _ in range(40 if Bl <= 16 else 20):
                t0 = time.perf_counter()
                e.rotate_multiply_plain_prepared(prep_l)
                e.sync()
                ts.append(time.perf_counter() - t0)
            med = statistics.median(ts)
            batch_ladder.append({"batch": Bl, "us_per_call": med * 1e6, "us_per_op": med / Bl * 1e6, "ops_per_s": Bl / med,
                                 "frac": Bl * bytes_op_ / med / 1e9 / HBM_PEAK_GBS})
            del prep_l
        try:
            from seal_fyp_logistic_regression_amd.seal import galois_elt_from_step
            Nc, pc = SETS["C4"]
            ec = Engine(Nc, pc, device=local_rank)
            kc, Lc, steps_c = len(pc), 2, 300
            ckey = ec.sample("uniform", key32("chain-key"), 1, 2 * (kc - 1), kc, 0)
            chain_level_us = {"ring": "C4 (the LR parameter set), L = 2", "steps": steps_c}
            for nc in (1, 8):
                ccts = [ec.sample("uniform", key32("chain-ct"), 10 + i, 2, Lc, 0) for i in range(nc)]
                caccs = [ec.sample("uniform", key32("chain-acc"), 50 + i, 2, Lc, 0) for i in range(nc)]
                celt = [galois_elt_from_step(1, Nc)] * nc
                ec.rotate_add_chain(Lc, ccts, celt, [ckey] * nc, caccs, steps_c)
                ec.sync()
                t0 = time.perf_counter()
                ec.rotate_add_chain(Lc, ccts, celt, [ckey] * nc, caccs, steps_c)
                ec.sync()
                chain_level_us[f"n{nc}"] = (time.perf_counter() - t0) / steps_c * 1e6
            del ckey, ccts, caccs
            ec.close()
        except Exception as ex:  # reported, never fatal for the headline
            chain_level_us = {"error": repr(ex)[:300]}
        # the PUBLIC Python entry (Engine.rotate_multiply_plain_batch: list marshalling on every call) next to the prepared
        # C-ABI call the headline times since round 5 (keep the marshalling cost visible)
        usteps = max(3, args.steps // 5)
        udt, _ = timed(lambda: e.rotate_multiply_plain_batch(L, cts, elts, keys, pts, outs), usteps, 1)
        unprepared = {"value": B * usteps * world / udt, "steps": usteps, "ms_per_step": udt / usteps * 1e3,
                      "note": "Engine.rotate_multiply_plain_batch with Python lists (the r01-r04 headline's call)"}

    # what was timed is checked: one item of EVERY chunk of the launch sequence (the engine cuts the batch into chunks of
    # at most 256 items on alternating internal streams) plus the last item, word for word against the CPU oracle (rank 0)
    verified, verified_items = None, []
    if rank == 0:
        try:
            from oracle import oracle as O
            o = O.Oracle(N, primes)
            hk = keyv[0].download()
            verified_items = sorted({min(c0 + (7 * (c0 // 256)) % 256, B - 1) for c0 in range(0, B, 256)} | {0, B - 1})
            verified = all(bool((outs[i].download() == o.rotate_mulplain(cts[i].download(), 3, hk, pts[i].download())).all())
                           for i in verified_items)
        except Exception as ex:  # the oracle is the checker only; its absence must not hide the GPU number
            verified = f"not checked: {ex!r}"

    # per-launch-kind durations: a short profiled pass (serial, HIP events between the launches).  The host-side
    # verification above left the device idle and its clock low: two untimed steps bring it back first.
    step()
    step()
    e.sync()
    e.profile_begin()
    step()
    stage_ms, nchunks = e.profile_end()

    # secondary pass: the same batch with `nk` distinct steps / keys round-robin (a linear transform rotates by distinct
    # steps with distinct keys: no key sharing between neighbouring items, nk keys competing for the caches)
    variants = {}
    if args.variant_keys > 1:
        from seal_fyp_logistic_regression_amd.seal import galois_elt_from_step
        velts = [galois_elt_from_step(1 + (i % nk), N) for i in range(B)]
        vkeys = [keyv[i % nk] for i in range(B)]
        vsteps = max(3, args.steps // 10)
        prepared["v"] = e.prepare_rotate_multiply_plain_batch(L, cts, velts, vkeys, pts, outs)
        vdt, _ = timed(lambda: e.rotate_multiply_plain_prepared(prepared["v"]), vsteps, 1)
        variants[f"distinct_keys_{nk}"] = {"value": B * vsteps * world / vdt, "steps": vsteps,
                                           "key_bytes": nk * key_words * 8,
                                           "note": f"steps 1..{nk} round-robin, one uniform-random key each"}
    # 64 distinct keys (480 MiB at C3, more than the 256 MiB Infinity Cache): the engine processes a batch grouped by key,
    # so each key is read from HBM ONCE PER GROUP of B/64 items, not once per item -- this is the many-keys case of a
    # linear transform whose rotations repeat steps, NOT the key-streaming regime (that is `roofline.key_per_item` below)
    if args.stream_keys > 1:
        from seal_fyp_logistic_regression_amd.seal import galois_elt_from_step
        ns = args.stream_keys
        skey = e.sample("uniform", hashlib.sha256(b"hefx-bench:stream-keys").digest(), 5, 2 * L * ns, k, 0)
        skeyv = [skey.view(i * key_words, (L, 2, k, N)) for i in range(ns)]
        selts = [galois_elt_from_step(1 + (i % ns), N) for i in range(B)]
        skeys = [skeyv[i % ns] for i in range(B)]
        ssteps = max(3, args.steps // 10)
        sprep = e.prepare_rotate_multiply_plain_batch(L, cts, selts, skeys, pts, outs)
        sdt, sgpu = timed(lambda: e.rotate_multiply_plain_prepared(sprep), ssteps, 1)
        del sprep
        variants[f"key_streaming_{ns}"] = {
            "value": B * ssteps * world / sdt, "steps": ssteps, "key_bytes": ns * key_words * 8,
            "hbm_key_reads_per_item": ns / B,
            "note": f"{ns} distinct keys = {ns * key_words * 8 / 2**20:.0f} MiB > the 256 MiB Infinity Cache; items are "
                    f"processed grouped by key, so a key is read from HBM once per {B // ns} items ({ns / B * key_words * 8 / 1e6:.2f} "
                    "MB of key traffic per op) -- many keys, not one key per item"}
        del skey, skeyv, skeys
    # THE KEY-STREAMING REGIME: one Galois key PER ITEM, so every key switch pays its
    # 2 L (L+1) N key words from HBM -- the regime the algorithmic bytes of `roofline.frac` price.  K items of the batch
    # with K distinct uniform-random keys (7.9 MB each at C3: 8 GB at K = 1024).
    key_per_item = None
    if args.key_per_item > 1 and rank == 0:
        from seal_fyp_logistic_regression_amd.seal import galois_elt_from_step
        K = min(args.key_per_item, B)
        try:
            kkey = e.sample("uniform", hashlib.sha256(b"hefx-bench:key-per-item").digest(), 6, 2 * L * K, k, 0)
            kkeyv = [kkey.view(i * key_words, (L, 2, k, N)) for i in range(K)]
            kelts = [galois_elt_from_step(1 + (i % 64), N) for i in range(K)]
            ksteps = max(3, args.steps // 5)
            kprep = e.prepare_rotate_multiply_plain_batch(L, cts[:K], kelts, kkeyv, pts[:K], outs[:K])
            run_k = lambda: e.rotate_multiply_plain_prepared(kprep)
            for _ in range(2):
                run_k()
            e.sync()
            ev0, ev1 = e.event(), e.event()
            t0 = time.perf_counter()
            e.event_record(ev0)
            for _ in range(ksteps):
                run_k()
            e.event_record(ev1)
            e.sync()
            kdt = time.perf_counter() - t0
            kgpu = e.event_elapsed_ms(ev0, ev1)
            kops = K * ksteps / kdt
            key_per_item = {
                "ops_per_s": kops, "frac": kops * algorithmic_bytes_per_op(N, L) / 1e9 / HBM_PEAK_GBS,
                "achieved_GBps_algorithmic": K * algorithmic_bytes_per_op(N, L) / (kgpu / ksteps * 1e-3) / 1e9,
                "batch": K, "steps": ksteps, "distinct_keys": K, "key_bytes": K * key_words * 8,
                "note": "one uniform-random Galois key per item: every key word is read from HBM exactly once per op"}
            del kkey, kkeyv, kprep
        except Exception as ex:  # e.g. not enough memory for the keys: reported, never fatal
            key_per_item = {"error": repr(ex)[:300]}

    # `secondary` (north_star: "poly_modulus_degree in {8192, 16384}"): the same unit at the other
    # parameter sets, each on its own engine context with its own device-drawn batch, timed like the headline (barrier +
    # sync on both sides, max over ranks) for about a second, three outputs checked against the oracle on rank 0
    # (single-rank runs only: a leg that fails on ONE rank of a multi-rank run would leave its peers in the leg's barriers)
    secondary = {}
    for name in [x for x in args.secondary.split(",") if x and x != args.set and world == 1]:
        try:
            secondary[name] = secondary_bench(name, Engine, local_rank, rank, world, timed, key32)
        except Exception as ex:  # reported, never fatal for the headline
            secondary[name] = {"error": repr(ex)[:300]}

    board_rec = board.stop() if board else None
    line = None
    if rank == 0:
        bytes_op = algorithmic_bytes_per_op(N, L)
        total_ops = B * args.steps * world
        value = total_ops / dt
        # roofline of the path on THIS rank: algorithmic bytes of one step / HIP-event time of one step
        step_ms = gpu_ms / args.steps
        achieved = B * bytes_op / (step_ms * 1e-3) / 1e9
        # HBM traffic: not measurable live; taken from the committed rocprofv3 --pmc passes of this same command
        # (FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note, WRITE_SIZE as is), scaled to one step.
        src16, lib16 = _hashes()
        traffic, traffic_src, traffic_stale = None, None, None
        for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
            pmc = os.path.join(ROOT, "profiles", f"{rnd}_bench_pmc_traffic.json")
            if args.set == "C3" and os.path.exists(pmc):
                try:
                    pj = json.load(open(pmc))
                    pb = pj["per_op_bytes"]
                    traffic = (pb["fetch_x2"] + pb["write"]) * B / 1e9  # GB per step, same unit basis as achieved*time
                    traffic_src = (f"profiles/{rnd}_bench_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                   "separate passes); GB per step")
                    # the counters belong to the engine sources they were measured on (tools/pmc_traffic.py writes the hash)
                    traffic_stale = pj.get("csrc_sha16") != src16
                    break
                except Exception:
                    pass
        # VALU issue: the committed SQ counter pass of this same command gives the VALU wave-instructions per op; with the
        # engine clock sampled during this run that is cycles of wall time per instruction and SIMD -- against ~5 cycles
        # of issue cost for this mix (4 for 32-bit ops, 4.75 v_fma_f64, 5.7 v_mad_u64_u32: profiles/r01_valu_issue_rates.txt)
        issue = None
        sqf = next((f for f in (os.path.join(ROOT, "profiles", f"{r}_bench_sq_counters.json") for r in ("r06", "r05", "r04", "r03"))
                    if os.path.exists(f)), "")
        if args.set == "C3" and sqf:
            try:
                sq = json.load(open(sqf))
                meta = sq.pop("_meta", {})
                per_op = sum(v["SQ_INSTS_VALU"] / (v["launches"] * 256) for v in sq.values())  # 256 items per chunk launch
                mhz = (board_rec or {}).get("sclk_mhz")
                issue = {"valu_wave_instr_per_op": per_op,
                         "source": f"profiles/{os.path.basename(sqf)} (rocprofv3 --pmc SQ_INSTS_VALU of this command)",
                         "stale": meta.get("csrc_sha16") != src16,
                         "sclk_mhz": mhz}
                if mhz:
                    cyc = (dt / total_ops * world) * mhz * 1e6 * SIMDS / per_op
                    issue["wall_cycles_per_instr_per_simd"] = cyc
                    issue["frac_of_issue_bound_at_5_cycles"] = 5.0 / cyc
            except Exception:
                issue = None
        dom = max(stage_ms, key=stage_ms.get)
        tot = sum(stage_ms.values())
        valu = valu_bound_ops_per_s(N, primes, L)
        per_gpu = value / world
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
            "verified": verified,
            "verified_items": verified_items,
            "libhefx_sha16": lib16,      # sha256[:16] of the libhefx.so this run loaded
            "csrc_sha16": src16,         # ... and of the engine sources (csrc/ + include/hefx.h) it was built from
            "rescale_mode": "round" if e.rescale_rounded else "floor",  # the engine's default division
            "sustained": sustained,
            "batch_ladder": batch_ladder,          # B in {1, 16, 256, 1024}, us per op and roofline frac
            "chain_level_us": chain_level_us,      # one rotate-by-1 + add level (helper.h:472-476), n = 1 and 8 chains, L = 2
            "unprepared_call": unprepared,
            "secondary": secondary,
            "config": {
                "workload": f"{args.set}: N={N}, coeff_modulus bits "
                            f"{[p.bit_length() for p in primes]}, level L={L} (k={k}); per step and per GPU "
                            f"{B} independent rotate_vector(step=1, direct Galois key)+multiply_plain, "
                            "uniform random residues drawn on the device, inputs resident in HBM",
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
                # True: the counter file was measured on other engine sources than the ones that ran (csrc hash differs)
                "traffic_stale": traffic_stale,
                # the honest number for the regime `frac` prices: one Galois key per item
                "key_per_item": key_per_item,
                "algorithmic_GB_per_step": B * bytes_op / 1e9,  # same basis as `traffic` (one step = one launch sequence)
                # what the HBM really does: measured traffic / step time against the same peak, and the
                # bytes the unit cannot avoid once the shared Galois key lives in the caches (ct in 2LN + pt LN + ct out
                # 2LN words) with the ratio of measured traffic to them
                "hbm_measured": None if traffic is None else {
                    "GBps": traffic / (step_ms * 1e-3), "frac_of_peak": traffic / (step_ms * 1e-3) / HBM_PEAK_GBS,
                    "frac_of_achievable_6300": traffic / (step_ms * 1e-3) / 6300.0,
                    "bytes_per_op": traffic * 1e9 / B,
                    "over_algorithmic": traffic * 1e9 / B / bytes_op,
                    "over_compulsory_cache_adjusted": traffic * 1e9 / B / (8 * N * L * 5)},
                "compulsory_bytes_per_op_cache_adjusted": 8 * N * L * 5,
                "launch": "one step = the key-switch launch sequence over the whole batch "
                          f"({nchunks} chunks); achieved = {B} ops x {bytes_op} B / {step_ms:.3f} ms (HIP events)",
                "dominant_kernel": dom,
                "kernel_avg_us": {kname: ms / max(nchunks, 1) * 1e3 for kname, ms in stage_ms.items()},
                "kernel_share": {kname: ms / tot for kname, ms in stage_ms.items()},
                "note": "frac prices the ALGORITHMIC bytes (SURVEY 8d), 70 % of which are the Galois key; in the "
                        "headline run every item shares one key, which then lives in L2 / Infinity Cache -- an "
                        "algorithmic-bytes equivalent, not measured HBM traffic (that is `traffic`); see "
                        "variants.distinct_keys_* for the run without key sharing",
                "valu": {
                    "bound": "valu",
                    "unit": "rotate+multiply_plain ops/s per GPU",
                    "achieved": per_gpu,
                    "peak": valu["peak_ops_per_s"],
                    "frac": per_gpu / valu["peak_ops_per_s"],
                    "model": f"{valu['int_transforms']} integer-policy + {valu['f64_transforms']} FP64-policy transforms "
                             f"x (N/2) log2 N butterflies, {VALU_CYC_INT:.0f} / {VALU_CYC_F64:.0f} SIMD cycles per "
                             f"wave-butterfly (profiles/r01_valu_issue_rates.txt), {SIMDS} SIMDs at {CLOCK_HZ / 1e9:.1f} "
                             "GHz; butterflies only (no loads, exchanges, MAC, epilogues)",
                    "issue": issue,
                },
            },
            "variants": variants,
            "board_under_load": board_rec,
            "lt_sharded": None,
            "composites": None,
        }
        if args.cpu_seconds > 0 and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(args.set, args.cpu_seconds)
                line["gpu_over_cpu_single_thread"] = value / line["cpu_baseline"]["single_thread_value"]
            except Exception as ex:  # the oracle is test infrastructure; never let it break the GPU number
                line["cpu_baseline"] = {"value": None, "error": repr(ex)}
        else:
            line["cpu_baseline"] = None

    def emit():
        if rank == 0:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(line) + "\n").encode())

    # secondary leg: the (sharded) linear transform.  The headline above is already measured; a watchdog makes sure a
    # stalled exchange in this leg cannot take the JSON line with it
    if args.lt and args.set == "C3":
        dims = [int(x) for x in args.lt.split(",") if x]
        prepared.clear()
        del big_ct, big_pt, big_out, cts, pts, outs

        def bail():  # the headline line still goes out, but a wedged exchange is a FAILED run: exit 3 on every rank
            if rank == 0:
                line["lt_sharded"] = {"error": "timed out after 180 s (the headline measurement is unaffected)"}
                emit()
            os._exit(3)

        dog = threading.Timer(180.0, bail)
        dog.daemon = True
        dog.start()
        try:
            lt = lt_sharded_bench(local_rank, world, dims, reps=10, direct_d=args.lt_direct, use_pg=use_pg)
        except Exception as ex:
            lt = {"error": repr(ex)[:400]}
        dog.cancel()
        if rank == 0:
            line["lt_sharded"] = lt
    # SURVEY 8(d)'s scaling workloads (matrix products at C3 / C5, LR row batch): same watchdog rule
    if args.composites and args.set == "C3":
        if prepared:
            prepared.clear()
            del big_ct, big_pt, big_out, cts, pts, outs

        def bail2():
            if rank == 0:
                line["composites"] = {"error": "timed out after 300 s (the headline measurement is unaffected)"}
                emit()
            os._exit(3)

        dog = threading.Timer(300.0, bail2)
        dog.daemon = True
        dog.start()
        try:
            comp = composites_bench(local_rank, world, [x for x in args.composites.split(",") if x], use_pg=use_pg)
        except Exception as ex:
            comp = {"error": repr(ex)[:400]}
        dog.cancel()
        if rank == 0:
            line["composites"] = comp
    emit()
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

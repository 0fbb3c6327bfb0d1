#!/usr/bin/env python3
"""CKKS encode throughput: hefx_ckks_encode (GPU FFT + round + RNS + NTT) vs the host-FFT path of the same
CKKSEncoder and vs the CPU oracle's encoder.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from oracle import oracle as O
    from seal_fyp_logistic_regression_amd import seal as S
    res = {}
    for N, bits in ((8192, [60, 40, 40, 60]), (16384, [60, 40, 40, 40, 40, 60])):
        parms = S.EncryptionParameters("ckks")
        parms.set_poly_modulus_degree(N)
        parms.set_coeff_modulus(S.CoeffModulus.Create(N, bits))
        ctx = S.SEALContext.Create(parms)
        dev, host = S.CKKSEncoder(ctx), S.CKKSEncoder(ctx, device_encode=False)
        e = ctx.backend.engine
        L = len(bits) - 1
        rng = np.random.default_rng(1)
        scale = 2.0 ** 40
        row = {}
        for count in (1, 64, 512):
            v = rng.uniform(-1, 1, (count, N // 2))
            out = e.empty(count, L, N)
            for _ in range(2):
                e.ckks_encode(L, v, scale, out=out)
            e.sync()
            reps = max(2, 2048 // count)
            t0 = time.perf_counter()
            for _ in range(reps):
                e.ckks_encode(L, v, scale, out=out)
            e.sync()
            row[f"gpu_batch{count}_vectors_per_s"] = reps * count / (time.perf_counter() - t0)
        v = rng.uniform(-1, 1, N // 2)
        t0 = time.perf_counter()
        for _ in range(20):
            host.encode(v, scale)
        e.sync()
        row["host_fft_path_vectors_per_s"] = 20 / (time.perf_counter() - t0)
        o = O.Oracle(N, ctx.primes)
        t0 = time.perf_counter()
        for _ in range(5):
            o.encode(L, v, scale)
        row["cpu_oracle_vectors_per_s"] = 5 / (time.perf_counter() - t0)
        res[f"N={N},L={L}"] = row
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One-off randomized parity stress (GPU vs CPU oracle), longer than the pytest suite: random parameter sets,
levels, Galois elements, batch compositions (mixed keys, shared sources, in-place rotations), relinearize,
rescale, multiply, multiply batch, one-pass product sums, sampler streams.  usage: stress_parity.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from seal_fyp_logistic_regression_amd import Engine

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(time.time()))
SETS = [(2048, [54]), (2048, [40, 40]), (4096, [36, 36, 37]), (4096, [50, 30, 30, 50]), (8192, [60, 40, 40, 60]),
        (8192, [40, 40, 40, 40, 40]), (8192, [55, 45, 50, 60]), (16384, [60, 40, 40, 40, 40, 60]),
        (16384, [50, 50, 50, 50]), (32768, [60, 40, 40, 60])]
t_end = time.time() + budget
rounds = checks = 0
while time.time() < t_end:
    N, bits = SETS[rng.integers(len(SETS))]
    primes = O.coeff_modulus_create(N, bits)
    o, e = O.Oracle(N, primes), Engine(N, primes)
    k = len(primes)
    seed = int(rng.integers(1 << 30))
    for L in ([k - 1] if k > 1 else []) + ([int(rng.integers(1, k))] if k > 2 else []):
        nkeys = 3
        keys = [o.uniform(k, 2 * (k - 1), seed + i).reshape(k - 1, 2, k, N) for i in range(nkeys)]
        dkeys = [e.to_device(x) for x in keys]
        n = int(rng.integers(1, 12))
        cts = [o.uniform(L, 2, seed + 100 + i) for i in range(n)]
        pts = [o.uniform(L, 1, seed + 200 + i)[0] for i in range(n)]
        elts = [int(2 * rng.integers(1, N) + 1) for _ in range(n)]
        ki = [int(rng.integers(nkeys)) for _ in range(n)]
        if rng.random() < 0.5:
            ki = sorted(ki)  # runs of equal keys: the MAC's shared-key path
        dcts = [e.to_device(c) for c in cts]
        outs = e.rotate_multiply_plain_batch(L, dcts, elts, [dkeys[j] for j in ki], [e.to_device(p) for p in pts])
        for i in range(n):
            assert (outs[i].download() == o.rotate_mulplain(cts[i], elts[i], keys[ki[i]], pts[i])).all(), (N, bits, L, i)
        # in place + plain
        e.apply_galois_batch(L, dcts, elts, [dkeys[j] for j in ki], outs=dcts)
        for i in range(n):
            assert (dcts[i].download() == o.apply_galois(cts[i], elts[i], keys[ki[i]])).all(), ("inplace", N, bits, L, i)
        # hoisted entry (exact since round 4: the regular words)
        src = e.to_device(cts[0])
        ho = e.rotate_hoisted_batch(L, src, elts, [dkeys[j] for j in ki])
        for i in range(n):
            assert (ho[i].download() == o.apply_galois(cts[0], elts[i], keys[ki[i]])).all(), ("hoist", N, bits, L, i)
        # a batch the engine hoists by itself: 33..72 rotations of 1..3 sources, now and then with a zero coefficient planted
        # in one source (the chunk then takes the on-device per-item fallback); a sample of the outputs against the oracle
        nb, ns = int(rng.integers(33, 73)), int(rng.integers(1, 4))
        srcs = [cts[i % n].copy() for i in range(ns)]
        planted = rng.random() < 0.3
        if planted:
            row = int(rng.integers(L))
            coef = o.ntt_inv(row, srcs[0][1, row])
            coef[int(rng.integers(N))] = 0
            srcs[0][1, row] = o.ntt_fwd(row, coef)
        dsr = [e.to_device(x) for x in srcs]
        sb = [int(rng.integers(ns)) for _ in range(nb)]
        eb = [int(2 * rng.integers(1, N) + 1) for _ in range(nb)]
        kb = [int(rng.integers(nkeys)) for _ in range(nb)]
        fb0 = e.ks_fallback_count()
        hb = e.apply_galois_batch(L, [dsr[x] for x in sb], eb, [dkeys[x] for x in kb])
        assert (e.ks_fallback_count() > fb0) == (planted and 0 in sb), ("fallback count", N, bits, L, planted)
        for i in sorted(set([0, nb - 1] + [int(x) for x in rng.integers(0, nb, 4)])):
            assert (hb[i].download() == o.apply_galois(srcs[sb[i]], eb[i], keys[kb[i]])).all(), ("auto-hoist", N, bits, L, i)
            checks += 1
        # multiply / relinearize / rescale
        a, b = cts[0], o.uniform(L, 2, seed + 999)
        m = o.multiply(a, b)
        dm = e.multiply(L, e.to_device(a), e.to_device(b))
        assert (dm.download() == m).all()
        r = e.relinearize(L, dm, dkeys[0])
        want = o.relinearize(m, keys[0])
        assert (r.download() == want).all(), ("relin", N, bits, L)
        if L >= 2:
            assert (e.rescale_to_next(L, 2, r).download() == o.rescale(want, rounded=e.rescale_rounded)).all(), ("rescale", N, bits, L)
            assert (e.rescale_to_next(L, 3, dm, rounded=True).download() == o.rescale(m, rounded=True)).all(), ("rescale round", N, bits, L)
        # row-batched ops and the one-pass product sum (inputs scattered over the pooled allocator)
        mb = e.multiply_batch(L, dcts, [dcts[0]] * n)
        back = [dcts[i].download() for i in range(n)]
        g = int(rng.integers(1, n + 1))
        ps = e.multiply_plain_sum(L, 2, dcts, [e.to_device(p) for p in pts], g)
        for i in range(n):
            assert (mb[i].download() == o.multiply(back[i], back[0])).all(), ("mulbatch", N, bits, L, i)
        for gi in range(len(ps)):
            acc = None
            for i in range(gi * g, min(n, (gi + 1) * g)):
                t = o.multiply_plain(back[i], pts[i])
                acc = t if acc is None else o.add(acc, t)
            assert (ps[gi].download() == acc).all(), ("mulplain_sum", N, bits, L, gi)
        checks += 4 * n + 3 + len(ps)
    # a random rotation forest (hefx_apply_galois_forest): random parents, a handful of elements / keys, plaintext products on
    # some nodes -- every third round, sometimes large enough for two lanes and hoisted depths; the oracle node by node
    if k > 1 and N <= 8192 and rounds % 3 == 0:
        L = k - 1
        steps = [1, -1, 2, 4, -8]
        felts = [O.galois_elt_from_step(N, s_) for s_ in steps]
        fkeys = [o.uniform(k, 2 * (k - 1), seed + 900 + i).reshape(k - 1, 2, k, N) for i in range(len(steps))]
        dfk = [e.to_device(x) for x in fkeys]
        nn = int(rng.choice([5, 20, 60, 130, 220]))
        srcs = [o.uniform(L, 2, seed + 950 + i) for i in range(2)]
        dsrcs = [e.to_device(x) for x in srcs]
        nroots = max(1, nn // int(rng.choice([3, 8, 20])))
        par = [-1 if i < nroots else int(rng.integers(0, i)) for i in range(nn)]
        which = [int(rng.integers(len(steps))) for _ in range(nn)]
        src_i = [int(rng.integers(2)) if par[i] < 0 else None for i in range(nn)]
        fpts = [o.uniform(L, 1, seed + 1000 + i)[0] if rng.random() < 0.3 else None for i in range(nn)]
        outs = e.apply_galois_forest(L, par, [dsrcs[j] if j is not None else None for j in src_i], [felts[w] for w in which],
                                     [dfk[w] for w in which], [e.to_device(p_) if p_ is not None else None for p_ in fpts])
        want = []
        for i in range(nn):
            src = srcs[src_i[i]] if par[i] < 0 else want[par[i]]
            want.append(o.apply_galois(src, felts[which[i]], fkeys[which[i]]) if fpts[i] is None else
                        o.rotate_mulplain(src, felts[which[i]], fkeys[which[i]], fpts[i]))
        for i in range(nn):
            assert (outs[i].download() == want[i]).all(), ("forest", N, bits, nn, i)
        checks += nn
    key32 = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    for kind in ("uniform", "ternary", "noise"):
        sid = int(rng.integers(1 << 40))
        assert (e.sample(kind, key32, sid, 2, k).download() == o.sample(kind, key32, sid, 2, k)).all(), kind
        checks += 1
    rounds += 1
print(f"stress ok: {rounds} parameter-set rounds, {checks} bit-exact comparisons in {budget:.0f} s")

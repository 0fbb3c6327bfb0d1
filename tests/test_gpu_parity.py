"""GPU parity: every hefx_* compute entry point (through the C-ABI, ctypes) against the CPU oracle on the
same seeded inputs.  Bar: bit-exact uint64 RNS coefficients."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "appendix_b.json")))
SETS = {s["name"]: s for s in GOLD["sets"]}


def _mk(name):
    from oracle import oracle as O
    from seal_fyp_logistic_regression_amd import Engine
    if name == "toy2048":  # smallest degree the key-switch kernels are built for
        primes = O.coeff_modulus_create(2048, [50, 30, 30, 50])
        return O.Oracle(2048, primes), Engine(2048, primes), primes
    s = SETS[name]
    primes = [int(p, 16) for p in s["primes"]]
    return O.Oracle(s["N"], primes), Engine(s["N"], primes), primes


@pytest.fixture(scope="module")
def c2():
    return _mk("C2")


@pytest.fixture(scope="module")
def c3():
    return _mk("C3")


@pytest.mark.parametrize("n,bits", [(1024, [60, 40, 30]), (2048, [60, 40, 30]), (4096, [60, 36, 60]),
                                    (8192, [60, 40, 60]), (16384, [60, 40, 50, 60]), (32768, [60, 40, 60])])
def test_ntt_forward_inverse_bit_exact(n, bits):
    from oracle import oracle as O
    from seal_fyp_logistic_regression_amd import Engine
    primes = O.coeff_modulus_create(n, bits)
    o, e = O.Oracle(n, primes), Engine(n, primes)
    for j in range(len(primes)):
        assert e.psi(j) == o.psi(j)
    k = len(primes)
    a = o.uniform(k, 3, 42 + n)  # [3][k][n]
    want = np.stack([np.stack([o.ntt_fwd(j, a[p, j]) for j in range(k)]) for p in range(3)])
    d = e.to_device(a)
    e.ntt_forward(d, 3, k, 0)
    got = d.download()
    assert (got == want).all()
    e.ntt_inverse(d, 3, k, 0)
    assert (d.download() == a).all()
    # edge values: zeros, q-1 everywhere, single one
    edge = np.zeros((1, k, n), dtype=np.uint64)
    for j in range(k):
        edge[0, j, :] = primes[j] - 1
    want = np.stack([o.ntt_fwd(j, edge[0, j]) for j in range(k)])[None]
    d = e.to_device(edge)
    e.ntt_forward(d, 1, k, 0)
    assert (d.download() == want).all()
    # sub-range of moduli (mod_first > 0)
    sub = np.ascontiguousarray(a[0, 1:])
    d = e.to_device(sub)
    e.ntt_forward(d, 1, k - 1, 1)
    assert (d.download() == np.stack([o.ntt_fwd(j, a[0, j]) for j in range(1, k)])).all()


def test_elementwise_ops(c2):
    o, e, primes = c2
    L = 3
    a, b = o.uniform(L, 2, 1), o.uniform(L, 2, 2)
    pt = o.uniform(L, 1, 3)[0]
    da, db, dp = e.to_device(a), e.to_device(b), e.to_device(pt)
    assert (e.add(L, 2, da, db).download() == o.add(a, b)).all()
    assert (e.sub(L, 2, da, db).download() == o.sub(a, b)).all()
    assert (e.negate(L, 2, da).download() == o.negate(a)).all()
    assert (e.add_plain(L, 2, da, dp).download() == o.add_plain(a, pt)).all()
    assert (e.multiply_plain(L, 2, da, dp).download() == o.multiply_plain(a, pt)).all()
    e.check_transparent()
    m = e.multiply(L, da, db).download()
    assert (m == o.multiply(a, b)).all()
    assert (e.square(L, da).download() == o.multiply(a, a)).all()
    # size-3 operands (Linear_Transform_Cipher sums size-3 results, helper.h:231)
    a3, b3 = o.uniform(L, 3, 4), o.uniform(L, 3, 5)
    assert (e.add(L, 3, e.to_device(a3), e.to_device(b3)).download() == o.add(a3, b3)).all()
    # lower level + contiguous batch of 4 ciphertexts
    a2 = o.uniform(2, 8, 6).reshape(4, 2, 2, o.N)
    b2 = o.uniform(2, 8, 7).reshape(4, 2, 2, o.N)
    got = e.add(2, 2, e.to_device(a2), e.to_device(b2), count=4).download()
    want = np.stack([o.add(a2[i], b2[i]) for i in range(4)])
    assert (got == want).all()
    # negate of zero stays zero
    z = np.zeros((2, L, o.N), dtype=np.uint64)
    assert (e.negate(L, 2, e.to_device(z)).download() == 0).all()


def test_transparent_result_is_reported(c2):
    from seal_fyp_logistic_regression_amd import capi
    o, e, _ = c2
    L = 3
    a = o.uniform(L, 2, 1)
    zero = np.zeros((L, o.N), dtype=np.uint64)
    e.multiply_plain(L, 2, e.to_device(a), e.to_device(zero))
    with pytest.raises(capi.TransparentCiphertextError):
        e.check_transparent()
    e.check_transparent()  # flag was cleared


def test_add_many(c2):
    o, e, _ = c2
    L = 3
    cts = [o.uniform(L, 2, 100 + i) for i in range(53)]  # > one pointer group
    want = cts[0]
    for c in cts[1:]:
        want = o.add(want, c)
    got = e.add_many(L, 2, [e.to_device(c) for c in cts]).download()
    assert (got == want).all()
    assert (e.add_many(L, 2, [e.to_device(cts[0])]).download() == cts[0]).all()
    # wide sums go through the device pointer table (two-level reduction): 150 inputs, and size-3 ciphertexts
    dev = [e.to_device(c) for c in cts]
    many = [dev[i % 53] for i in range(150)]
    want = cts[0]
    for i in range(1, 150):
        want = o.add(want, cts[i % 53])
    assert (e.add_many(L, 2, many).download() == want).all()
    c3s = [o.multiply(cts[i], cts[i + 1]) for i in range(4)]
    d3 = [e.to_device(c) for c in c3s]
    want3 = c3s[0]
    for i in range(1, 100):
        want3 = o.add(want3, c3s[i % 4])
    assert (e.add_many(L, 3, [d3[i % 4] for i in range(100)]).download() == want3).all()


def test_multiply_plain_sum_bit_exact(c2):
    """hefx_multiply_plain_sum (helper.h:265-278 in one pass): uniform residues, against multiply_plain + add per term;
    one group, ragged groups, a sum longer than one accumulator fold (128), more groups than one pointer-table slice,
    size-3 ciphertexts, and the argument checks."""
    o, e, _ = c2
    L = 3
    cts = [o.uniform(L, 2, 7000 + i) for i in range(9)]
    pts = [o.uniform(L, 1, 7100 + i)[0] for i in range(7)]
    dct, dpt = [e.to_device(c) for c in cts], [e.to_device(p) for p in pts]

    def want(idx):
        acc = None
        for ci, pi in idx:
            t = o.multiply_plain(cts[ci], pts[pi])
            acc = t if acc is None else o.add(acc, t)
        return acc

    for n, group in ((5, None), (23, 4), (300, 300), (700, 3), (1500, 1400)):  # last: groups beyond one table slice
        idx = [(i % 9, (3 * i + 1) % 7) for i in range(n)]
        g = n if group is None else group
        outs = e.multiply_plain_sum(L, 2, [dct[a] for a, _ in idx], [dpt[b] for _, b in idx], group)
        assert len(outs) == (n + g - 1) // g
        for k in (0, len(outs) // 2, len(outs) - 1):
            assert (outs[k].download() == want(idx[k * g:(k + 1) * g])).all(), (n, group, k)
    c3 = np.ascontiguousarray(o.multiply(cts[0], cts[1])[:, :2])  # size 3, one level down
    p0, p1 = np.ascontiguousarray(pts[0][:2]), np.ascontiguousarray(pts[1][:2])
    got = e.multiply_plain_sum(2, 3, [e.to_device(c3)] * 2, [e.to_device(p0), e.to_device(p1)])[0]
    assert (got.download() == o.add(o.multiply_plain(c3, p0), o.multiply_plain(c3, p1))).all()
    with pytest.raises(ValueError):
        e.multiply_plain_sum(L, 2, [dct[0]], [dpt[0]], outs=[dct[0]])  # output aliases an input


def test_row_batched_ops_bit_exact(c2):
    """hefx_multiply_batch and the count-batched add / rescale behind Evaluator.multiply_many / add_pairs /
    rescale_to_next_many_inplace: n independent ciphertexts per launch, same bits as one call each -- with one shared
    right operand (the weight ciphertext of logistic_regression_ckks.cpp:217-220), across pointer-table slices, and
    for inputs that are NOT views of one slab (per-item fallback)."""
    o, e, _ = c2
    L, n = 3, 450
    cts = [o.uniform(L, 2, 9000 + i) for i in range(7)]
    dev = [e.to_device(c) for c in cts]
    As = [dev[i % 7] for i in range(n)]
    outs = e.multiply_batch(L, As, [dev[3]] * n)
    for i in (0, 1, 6, 425, 426, n - 1):
        assert (outs[i].download() == o.multiply(cts[i % 7], cts[3])).all(), i
    assert e.contiguous(outs) and not e.contiguous(As)
    r2 = e.relinearize_batch(L, outs[:9], e.to_device(_rand_key(o, 5)))
    rs = e.rescale_batch(L, 2, r2)          # slab in, one launch
    key = _rand_key(o, 5)
    for i in (0, 8):
        want = o.rescale(o.relinearize(o.multiply(cts[i % 7], cts[3]), key), rounded=e.rescale_rounded)
        assert (rs[i].download() == want).all()
    sums = e.add_batch(L, 2, r2, r2[::-1])  # second list is not in slab order -> per-item path
    sums2 = e.add_batch(L, 2, r2, r2)       # both slabs -> one launch
    a = [x.download() for x in r2]
    for i in (0, 4, 8):
        assert (sums[i].download() == o.add(a[i], a[8 - i])).all()
        assert (sums2[i].download() == o.add(a[i], a[i])).all()
    with pytest.raises(ValueError):
        e.multiply_batch(L, [dev[0]], [dev[1]], outs=[dev[0]])


def test_pooled_allocator_recycles_without_overlap(c2):
    """hefx_malloc / hefx_free (include/hefx.h): freed blocks come back for the same size, blocks carved from one slab
    never overlap, data written through one block is not disturbed by its neighbours, and HEFX_POOL_MB=0 (plain
    hipMalloc / hipFree) gives the same results."""
    import os, subprocess, sys
    o, e, _ = c2
    L, N = 3, o.N
    a = e.empty(2, L, N)
    p = a.ptr
    del a
    b = e.empty(2, L, N)
    assert b.ptr == p  # recycled, no hipFree / hipMalloc round trip
    bufs = [e.empty(2, L, N) for _ in range(300)]  # several slabs of this size class
    ptrs = sorted(x.ptr for x in bufs + [b])
    assert all(q - r >= 2 * L * N * 8 for r, q in zip(ptrs, ptrs[1:]))
    cts = [o.uniform(L, 2, 8000 + i) for i in range(8)]
    dev = [e.to_device(c) for c in cts]
    del bufs, b
    tmp = [e.add(L, 2, dev[i], dev[(i + 1) % 8]) for i in range(8)]  # results land in recycled blocks
    for i in range(8):
        assert (dev[i].download() == cts[i]).all()
        assert (tmp[i].download() == o.add(cts[i], cts[(i + 1) % 8])).all()
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from oracle import oracle as O\nfrom seal_fyp_logistic_regression_amd import Engine\n"
            "primes=%r; o=O.Oracle(8192, primes); e=Engine(8192, primes)\n"
            "a,b=o.uniform(3,2,1),o.uniform(3,2,2)\n"
            "for _ in range(50): r=e.add(3,2,e.to_device(a),e.to_device(b))\n"
            "print('PARITY', bool((r.download()==o.add(a,b)).all()))\n") % (
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))), o.primes)
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "HEFX_POOL_MB": "0"}, capture_output=True,
                       text=True, timeout=600)
    assert "PARITY True" in r.stdout, (r.stdout[-300:], r.stderr[-1500:])


def _rand_key(o, seed):
    return o.uniform(o.k, 2 * (o.k - 1), seed).reshape(o.k - 1, 2, o.k, o.N)


@pytest.mark.parametrize("setname", ["C2", "C3", "rot5", "C5", "toy2048"])
def test_apply_galois_bit_exact(setname):
    from oracle import oracle as O
    o, e, primes = _mk(setname)
    Ltop = o.k - 1
    key = _rand_key(o, 7)
    dkey = e.to_device(key)
    for L in sorted({Ltop, max(1, Ltop - 1), 1}):
        ct = o.uniform(L, 2, 11 + L)
        for step in (1, -1, 4):
            elt = O.galois_elt_from_step(o.N, step)
            want = o.apply_galois(ct, elt, key)
            got = e.apply_galois(L, e.to_device(ct), elt, dkey).download()
            assert (got == want).all(), (setname, L, step)
    # in place (rotate_vector_inplace, helper.h:474)
    ct = o.uniform(Ltop, 2, 98)
    d = e.to_device(ct)
    e.apply_galois(Ltop, d, O.galois_elt_from_step(o.N, 1), dkey, out=d)
    assert (d.download() == o.apply_galois(ct, O.galois_elt_from_step(o.N, 1), key)).all()
    # conjugation element 2N-1
    ct = o.uniform(Ltop, 2, 99)
    want = o.apply_galois(ct, 2 * o.N - 1, key)
    assert (e.apply_galois(Ltop, e.to_device(ct), 2 * o.N - 1, dkey).download() == want).all()


def test_rotate_multiply_plain_batch_bit_exact(c3):
    from oracle import oracle as O
    o, e, _ = c3
    L = 5
    n = 19  # one (partial) chunk since chunks hold up to 256 items; the multi-chunk default path: test_gpu_round3.py
    keys = {s: _rand_key(o, 1000 + s) for s in (1, 2)}
    dkeys = {s: e.to_device(k) for s, k in keys.items()}
    cts = [o.uniform(L, 2, 200 + i) for i in range(n)]
    pts = [o.uniform(L, 1, 300 + i)[0] for i in range(n)]
    steps = [1 + (i % 2) for i in range(n)]
    elts = [O.galois_elt_from_step(o.N, s) for s in steps]
    outs = e.rotate_multiply_plain_batch(L, [e.to_device(c) for c in cts], elts, [dkeys[s] for s in steps],
                                         [e.to_device(p) for p in pts])
    for i in range(n):
        want = o.rotate_mulplain(cts[i], elts[i], keys[steps[i]], pts[i])
        assert (outs[i].download() == want).all(), i
    # the unfused path gives the same bits
    i = 3
    r = e.apply_galois(L, e.to_device(cts[i]), elts[i], dkeys[steps[i]])
    got = e.multiply_plain(L, 2, r, e.to_device(pts[i])).download()
    assert (got == outs[i].download()).all()


def test_relinearize_and_rescale_bit_exact(c3):
    o, e, primes = c3
    for L in (5, 2):
        key = _rand_key(o, 5)
        a, b = o.uniform(L, 2, 1), o.uniform(L, 2, 2)
        m = o.multiply(a, b)
        want = o.relinearize(m, key)
        got = e.relinearize(L, e.to_device(m), e.to_device(key)).download()
        assert (got == want).all()
        outs = e.relinearize_batch(L, [e.to_device(m), e.to_device(o.multiply(b, b))], e.to_device(key))
        assert (outs[0].download() == want).all()
        assert (outs[1].download() == o.relinearize(o.multiply(b, b), key)).all()
        rs = e.rescale_to_next(L, 2, e.to_device(want)).download()
        assert rs.shape == (2, L - 1, o.N)
        assert (rs == o.rescale(want, rounded=e.rescale_rounded)).all()
        rs3 = e.rescale_to_next(L, 3, e.to_device(m)).download()
        assert (rs3 == o.rescale(m, rounded=e.rescale_rounded)).all()
    md = e.mod_drop(5, 3, 2, e.to_device(a if a.shape[1] == 5 else o.uniform(5, 2, 1))).download()
    assert md.shape == (2, 3, o.N)


def test_mod_drop_and_reduce(c3):
    o, e, primes = c3
    a = o.uniform(5, 2, 1)
    assert (e.mod_drop(5, 3, 2, e.to_device(a)).download() == a[:, :3]).all()
    # sum of 8 canonical words then canonicalise == 8 modular adds
    parts = [o.uniform(5, 2, 50 + i) for i in range(8)]
    raw = np.zeros_like(parts[0])
    want = np.zeros_like(parts[0])
    for p in parts:
        raw = raw + p
        want = o.add(want, p)
    d = e.to_device(raw)
    e.reduce_canonical(5, 2, d, addends=8)
    assert (d.download() == want).all()


def test_invalid_arguments_raise(c2):
    o, e, _ = c2
    a = e.to_device(o.uniform(3, 2, 1))
    key = e.to_device(_rand_key(o, 1))
    with pytest.raises(ValueError):
        e.apply_galois(3, a, 4, key)  # even Galois element
    with pytest.raises(ValueError):
        e.apply_galois(4, a, 3, key)  # level above top data level
    with pytest.raises(ValueError):
        e.rescale_to_next(1, 2, a)
    # a dependent chain handed over as ONE batch is refused (items run key-grouped on several streams): item 1 reads
    # item 0's output / two items write one output; sharing a SOURCE and an item's own in-place rotation are fine
    b, c = e.empty(2, 3, 8192), e.empty(2, 3, 8192)
    with pytest.raises(ValueError, match="independent"):
        e.apply_galois_batch(3, [a, b], [3, 3], [key, key], [b, c])
    with pytest.raises(ValueError, match="overlapping outputs"):
        e.apply_galois_batch(3, [a, a], [3, 9], [key, key], [b, b])
    # ... and the check is on byte ranges: views into one allocation that overlap by a row are caught too
    big = e.empty(3, 3, 8192)
    v0, v1 = big.view(0, (2, 3, 8192)), big.view(3 * 8192, (2, 3, 8192))
    with pytest.raises(ValueError, match="overlapping outputs"):
        e.apply_galois_batch(3, [a, a], [3, 9], [key, key], [v0, v1])
    with pytest.raises(ValueError, match="independent"):
        e.apply_galois_batch(3, [a, v1], [3, 9], [key, key], [v0, c])
    e.apply_galois_batch(3, [a, a], [3, 3], [key, key], [b, c])
    assert (b.download() == c.download()).all()
    want = b.download()
    a2 = e.copy(a)
    e.apply_galois_batch(3, [a2, a], [3, 3], [key, key], [a2, c])   # item 0 in place
    assert (a2.download() == want).all()


def test_alternative_launch_paths_bit_exact(c3):
    """The optional paths must give the same bits as the default one: fused digit-NTT+MAC kernel (HEFX_FUSED=1),
    integer-only arithmetic policy (HEFX_NO_FP64=1), serial chunks (HEFX_STREAMS=0), small chunks + sub-chunks."""
    import subprocess, sys, os, json
    from oracle import oracle as O
    o, e, primes = c3
    L, n = 5, 9
    key = _rand_key(o, 77)
    cts = [o.uniform(L, 2, 900 + i) for i in range(n)]
    pts = [o.uniform(L, 1, 950 + i)[0] for i in range(n)]
    want = [o.rotate_mulplain(cts[i], 3, key, pts[i]) for i in range(n)]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from oracle import oracle as O\n"
        "from seal_fyp_logistic_regression_amd import Engine\n"
        "primes=%r; o=O.Oracle(16384, primes); e=Engine(16384, primes); L=5; n=9\n"
        "key=o.uniform(o.k, 2*(o.k-1), 77).reshape(o.k-1,2,o.k,o.N); dk=e.to_device(key)\n"
        "cts=[o.uniform(L,2,900+i) for i in range(n)]; pts=[o.uniform(L,1,950+i)[0] for i in range(n)]\n"
        "outs=e.rotate_multiply_plain_batch(L,[e.to_device(c) for c in cts],[3]*n,[dk]*n,[e.to_device(p) for p in pts])\n"
        "ok=all((outs[i].download()==o.rotate_mulplain(cts[i],3,key,pts[i])).all() for i in range(n))\n"
        "print('PARITY', ok)\n") % (root, primes)
    for env in ({"HEFX_FUSED": "1"}, {"HEFX_NO_FP64": "1"}, {"HEFX_STREAMS": "0"}, {"HEFX_CHUNK": "4", "HEFX_SUB": "2"},
                {"HEFX_FUSED": "1", "HEFX_NO_FP64": "1", "HEFX_CHUNK": "5"}):
        r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True,
                           timeout=600)
        assert "PARITY True" in r.stdout, (env, r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("setname", ["C2", "C3", "C5"])
def test_hoisted_rotations_bit_exact_against_the_regular_key_switch(setname):
    """hefx_rotate_hoisted_batch: n rotations of ONE ciphertext share its digit decomposition -- since
    round 4 with the flip-mask term (ks_mac_exact_kernel), so the words are SEAL's: the checker is orc_apply_galois, the
    regular per-item sequence, with and without the fused plaintext product, at a lower level.  (7 items: the latency
    path serves them; the hoisted kernels take over above 32 items -- test_hoisted_chunking... with HEFX_CHUNK and
    tests/test_gpu_round4.py::test_shared_source_decomposition_bit_exact cover those.)  The uncorrected hoisted statement
    of rounds 1-3 (orc_apply_galois_hoisted) has other words."""
    from oracle import oracle as O
    o, e, primes = _mk(setname)
    L = len(primes) - 1
    steps = [1, 2, 3, 5, -1, -4, 7]
    n = len(steps)
    keys = {s: _rand_key(o, 4000 + s) for s in steps}
    dkeys = {s: e.to_device(k) for s, k in keys.items()}
    ct = o.uniform(L, 2, 42)
    dct = e.to_device(ct)
    pts = [o.uniform(L, 1, 600 + i)[0] for i in range(n)]
    elts = [O.galois_elt_from_step(o.N, s) for s in steps]
    outs = e.rotate_hoisted_batch(L, dct, elts, [dkeys[s] for s in steps], [e.to_device(p) for p in pts])
    before = e.ks_fallback_count()
    differs = 0
    for i in range(n):
        h = o.apply_galois(ct, elts[i], keys[steps[i]])
        assert (outs[i].download() == o.multiply_plain(h, pts[i])).all(), (setname, i)
        differs += int((h != o.apply_galois_hoisted(ct, elts[i], keys[steps[i]])).any())
    assert differs == n  # the flip term matters on every one of them
    plain = e.rotate_hoisted_batch(L, dct, elts, [dkeys[s] for s in steps])
    for i in range(n):
        assert (plain[i].download() == o.apply_galois(ct, elts[i], keys[steps[i]])).all(), (setname, i)
    assert e.ks_fallback_count() == before  # the hoisted kernels produced these, not the per-item fallback
    assert (dct.download() == ct).all()  # the shared source is never written
    if L > 2:
        ctl = o.uniform(L - 1, 2, 43)
        got = e.rotate_hoisted_batch(L - 1, e.to_device(ctl), elts[:4], [dkeys[s] for s in steps[:4]])
        for i in range(4):
            assert (got[i].download() == o.apply_galois(ctl, elts[i], keys[steps[i]])).all()
    with pytest.raises(ValueError):
        e.rotate_hoisted_batch(L, dct, elts[:1], [dkeys[steps[0]]], outs=[dct])  # output aliases the source


def test_hoisted_chunking_and_policies_bit_exact(c3):
    import os, subprocess, sys
    o, e, primes = c3
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from oracle import oracle as O\n"
        "from seal_fyp_logistic_regression_amd import Engine\n"
        "primes=%r; o=O.Oracle(16384, primes); e=Engine(16384, primes); L=5; n=45\n"
        "keys=[o.uniform(o.k, 2*(o.k-1), 70+i).reshape(o.k-1,2,o.k,o.N) for i in range(3)]; dk=[e.to_device(k) for k in keys]\n"
        "ct=o.uniform(L,2,900); d=e.to_device(ct); pts=[o.uniform(L,1,950+i)[0] for i in range(n)]\n"
        "elts=[[3,9,27][i%%3] for i in range(n)]\n"
        "outs=e.rotate_hoisted_batch(L,d,elts,[dk[i%%3] for i in range(n)],[e.to_device(p) for p in pts])\n"
        "ok=all((outs[i].download()==o.multiply_plain(o.apply_galois(ct,elts[i],keys[i%%3]),pts[i])).all() for i in range(n))\n"
        "print('PARITY', ok)\n") % (root, primes)
    # 45 items: one hoisted chunk; chunks of 40 + 5 (hoisted kernels, then the latency path) on two streams; serial chunks
    # of 36 + 9; the integer policies for every modulus; hoisting switched off
    for env in ({}, {"HEFX_CHUNK": "40"}, {"HEFX_STREAMS": "0", "HEFX_CHUNK": "36"}, {"HEFX_NO_FP64": "1"}, {"HEFX_SHARE_SRC": "0"},
                {"HEFX_FLIPW_MB": "1"}):  # ... and with no room for the flip tables (the batch then runs unhoisted)
        r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True,
                           timeout=600)
        assert "PARITY True" in r.stdout, (env, r.stdout[-500:], r.stderr[-1500:])

"""Frozen golden vectors (tests/golden/frozen_vectors.json, made by tests/golden/make_frozen_vectors.py): the toy key switch and
rescale in pymodel's words -- the oracle must reproduce them -- and SHA-256 digests of the oracle's outputs at N = 2048 and
N = 8192, which the oracle (CPU test) and the HIP engine through the C-ABI (GPU test) must both still produce.  The on-the-fly
comparisons elsewhere in tests/ cannot see a change that moves the oracle and the engine together; a frozen file can."""
import importlib.util
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
FROZEN = json.load(open(os.path.join(HERE, "golden", "frozen_vectors.json")))
_spec = importlib.util.spec_from_file_location("make_frozen_vectors", os.path.join(HERE, "golden", "make_frozen_vectors.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)


def test_splitmix64_known_answers():
    # the reference stream of splitmix64 seeded with 1234567 (Vigna's test vector)
    assert [int(x) for x in gen.splitmix64(1234567, 3)] == [6457827717110365317, 3203168211198807973, 9817491932198370423]


def test_toy_key_switch_and_rescale_in_pymodels_words():
    t = FROZEN["toy"]
    n, L, primes = t["n"], t["L"], t["primes"]
    assert O.coeff_modulus_create(n, [24, 20, 25]) == primes
    o = O.Oracle(n, primes)
    assert [o.psi(j) for j in range(len(primes))] == t["psi"]
    ct, target = np.asarray(t["ct"], dtype=np.uint64), np.asarray(t["target"], dtype=np.uint64)
    key = np.asarray(t["key"], dtype=np.uint64)
    assert o.switch_key(ct, target, key).tolist() == t["switch_key"]
    assert o.rescale(ct, rounded=False).tolist() == t["rescale_floor"]


def _inputs(case):
    ct, ct3, pt, key = gen.case_inputs(case["n"], case["primes"], case["L"], case["seed"])
    assert {"ct": gen.digest(ct), "ct3": gen.digest(ct3), "pt": gen.digest(pt), "key": gen.digest(key)} == case["in"]
    return ct, ct3, pt, key


@pytest.mark.parametrize("case", FROZEN["digests"], ids=lambda c: c["name"])
def test_oracle_still_produces_the_frozen_digests(case):
    n, L = case["n"], case["L"]
    o = O.Oracle(n, case["primes"])
    ct, ct3, pt, key = _inputs(case)
    got = {}
    for step in (1, -3):
        elt = O.galois_elt_from_step(n, step)
        got[f"apply_galois_step{step}"] = gen.digest(o.apply_galois(ct, elt, key))
        got[f"rotate_mulplain_step{step}"] = gen.digest(o.rotate_mulplain(ct, elt, key, pt))
    got["relinearize"] = gen.digest(o.relinearize(ct3, key))
    got["rescale_floor"] = gen.digest(o.rescale(ct, rounded=False))
    got["rescale_round"] = gen.digest(o.rescale(ct, rounded=True))
    got["rescale_floor_size3"] = gen.digest(o.rescale(ct3, rounded=False))
    got["multiply"] = gen.digest(o.multiply(ct, ct))
    got["multiply_plain"] = gen.digest(o.multiply_plain(ct, pt))
    assert got == case["out"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", FROZEN["digests"], ids=lambda c: c["name"])
def test_engine_produces_the_frozen_digests(case):
    from seal_fyp_logistic_regression_amd import Engine
    n, L = case["n"], case["L"]
    e = Engine(n, case["primes"], device=0)
    ct, ct3, pt, key = _inputs(case)
    dct, dct3, dpt, dkey = e.to_device(ct), e.to_device(ct3), e.to_device(pt), e.to_device(key)
    got = {}
    for step in (1, -3):
        elt = O.galois_elt_from_step(n, step)
        got[f"apply_galois_step{step}"] = gen.digest(e.apply_galois(L, dct, elt, dkey).download())
        got[f"rotate_mulplain_step{step}"] = gen.digest(e.rotate_multiply_plain_batch(L, [dct], [elt], [dkey], [dpt])[0].download())
    got["relinearize"] = gen.digest(e.relinearize(L, dct3, dkey).download())
    got["rescale_floor"] = gen.digest(e.rescale_to_next(L, 2, dct, rounded=False).download())
    got["rescale_round"] = gen.digest(e.rescale_to_next(L, 2, dct, rounded=True).download())
    got["rescale_floor_size3"] = gen.digest(e.rescale_to_next(L, 3, dct3, rounded=False).download())
    got["multiply"] = gen.digest(e.multiply(L, dct, dct).download())
    got["multiply_plain"] = gen.digest(e.multiply_plain(L, 2, dct, dpt).download())
    assert got == case["out"]

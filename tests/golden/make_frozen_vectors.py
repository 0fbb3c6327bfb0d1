"""Generates tests/golden/frozen_vectors.json -- the golden vectors SURVEY 8(c) asks the build to create itself, FROZEN, so that
a later change which moves the oracle and the engine together is still seen:
  toy      N = 16, L = 2 (three primes): inputs and outputs IN FULL, the outputs computed by tests/pymodel.py (pure Python,
           O(N^2) definitions, big integers) -- not by the C oracle;
  digests  N = 2048 and N = 8192 (config 2's primes): SHA-256 of the inputs and of what the C oracle returns for them (the
           oracle is pinned to pymodel at toy size by tests/test_oracle_pinning.py).
Inputs come from splitmix64 written out below (no library generator whose stream could change), reduced modulo the row's prime.
    python tests/golden/make_frozen_vectors.py      (in the repository root; needs gcc for the oracle)
PARITY UNPINNED all the same: these are this repository's restatement of SEAL 3.4.5, frozen -- not SEAL's own words."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def splitmix64(seed: int, count: int) -> np.ndarray:
    """count words of the splitmix64 stream started at `seed`"""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def rows(seed: int, primes, shape_front, n: int) -> np.ndarray:
    """[*shape_front][len(primes)][n] residues: row j of every polynomial is reduced modulo primes[j]"""
    count = int(np.prod(shape_front)) * len(primes) * n
    w = splitmix64(seed, count).reshape(*shape_front, len(primes), n)
    q = np.asarray(primes, dtype=np.uint64).reshape(*([1] * len(shape_front)), len(primes), 1)
    return np.ascontiguousarray(w % q)


def digest(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def case_inputs(n, primes, L, seed):
    k = len(primes)
    data = list(primes[:L])
    ct = rows(seed, data, (2,), n)
    ct3 = rows(seed + 1, data, (3,), n)
    pt = rows(seed + 2, data, (), n)
    key = rows(seed + 3, primes, (k - 1, 2), n)  # [k-1][2][k][n]
    return ct, ct3, pt, key


def main():
    from oracle import oracle as O
    import pymodel as pm
    out = {"generator": "tests/golden/make_frozen_vectors.py", "note": "frozen restatement of SEAL 3.4.5 -- parity unpinned"}
    # ---- toy: pymodel's words in full
    n, L = 16, 2
    primes = O.coeff_modulus_create(n, [24, 20, 25])
    psis = [pm.min_primitive_root(2 * n, q) for q in primes]
    ct, _, _, key = case_inputs(n, primes, L, 0x70F)
    target = rows(0x7AB, primes[:L], (), n)
    sw = pm.switch_key(ct.tolist(), target.tolist(), key.tolist(), primes, psis, L)
    rs = pm.rescale_floor(ct.tolist(), primes, psis, L)
    out["toy"] = {"n": n, "L": L, "primes": [int(q) for q in primes], "psi": [int(p) for p in psis], "ct": ct.tolist(),
                  "target": target.tolist(), "key": key.tolist(), "switch_key": sw, "rescale_floor": rs}
    # ---- digests: the C oracle's words at sizes the engine runs
    cases = []
    for name, n, primes, L in (("n2048", 2048, O.coeff_modulus_create(2048, [40, 30, 30, 40]), 3),
                               ("c2", 8192, [0xffffffffffe8001, 0xfffff4c001, 0xfffffdc001, 0xfffffffffffc001], 3),
                               ("c2_level2", 8192, [0xffffffffffe8001, 0xfffff4c001, 0xfffffdc001, 0xfffffffffffc001], 2)):
        o = O.Oracle(n, [int(q) for q in primes])
        seed = 0x5EA1C0DE + n + L
        ct, ct3, pt, key = case_inputs(n, [int(q) for q in primes], L, seed)
        rec = {"name": name, "n": n, "L": L, "primes": [int(q) for q in primes], "seed": seed,
               "in": {"ct": digest(ct), "ct3": digest(ct3), "pt": digest(pt), "key": digest(key)}, "out": {}}
        for step in (1, -3):
            elt = O.galois_elt_from_step(n, step)
            rec["out"][f"apply_galois_step{step}"] = digest(o.apply_galois(ct, elt, key))
            rec["out"][f"rotate_mulplain_step{step}"] = digest(o.rotate_mulplain(ct, elt, key, pt))
        rel = o.relinearize(ct3, key)
        rec["out"]["relinearize"] = digest(rel)
        rec["out"]["rescale_floor"] = digest(o.rescale(ct, rounded=False))
        rec["out"]["rescale_round"] = digest(o.rescale(ct, rounded=True))
        rec["out"]["rescale_floor_size3"] = digest(o.rescale(ct3, rounded=False))
        rec["out"]["multiply"] = digest(o.multiply(ct, ct))
        rec["out"]["multiply_plain"] = digest(o.multiply_plain(ct, pt))
        cases.append(rec)
    out["digests"] = cases
    path = os.path.join(ROOT, "tests", "golden", "frozen_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

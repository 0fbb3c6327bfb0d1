"""Pure-Python (big-int, O(N^2)) restatement of the path's math for TOY sizes only.

Independent of oracle/ckks_oracle.c: used to pin the C oracle at N<=64.
Follows SURVEY.md Appendix A.5/A.7/A.8/A.9 (SEAL 3.4.5 semantics); test infrastructure only.
"""


def bitrev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def is_prime(n):
    if n < 2:
        return False
    i = 2
    while i * i <= n:
        if n % i == 0:
            return False
        i += 1
    return True


def min_primitive_root(two_n, q):
    """smallest integer of exact multiplicative order two_n (a power of two) mod q"""
    for x in range(2, q):
        if pow(x, two_n // 2, q) == q - 1:
            return x
    raise ValueError


def ntt_def(a, psi, q):
    n = len(a)
    logn = n.bit_length() - 1
    out = []
    for i in range(n):
        x = pow(psi, 2 * bitrev(i, logn) + 1, q)
        out.append(sum(int(a[k]) * pow(x, k, q) for k in range(n)) % q)
    return out


def intt_def(A, psi, q):
    n = len(A)
    logn = n.bit_length() - 1
    ninv = pow(n, -1, q)
    out = []
    for k in range(n):
        s = 0
        for i in range(n):
            x = pow(psi, -(2 * bitrev(i, logn) + 1) * k, q)
            s += int(A[i]) * x
        out.append(s * ninv % q)
    return out


def galois_table(n, elt):
    logn = n.bit_length() - 1
    tab = []
    for i in range(n):
        raw = (elt * (2 * bitrev(i, logn) + 1)) % (2 * n)
        tab.append(bitrev((raw - 1) >> 1, logn))
    return tab


def switch_key(ct, target, key, primes, psis, L):
    """App. A.8.  ct [2][L][n], target [L][n], key [L_key][2][k][n] (lists of ints). Returns new ct."""
    k = len(primes)
    n = len(target[0])
    P = primes[k - 1]
    mods = list(range(L)) + [k - 1]
    acc = [[[0] * n for _ in mods] for _ in range(2)]
    for i in range(L):
        d = intt_def(target[i], psis[i], primes[i])
        for jj, mi in enumerate(mods):
            m = primes[mi]
            if mi == i:
                x = [int(v) for v in target[i]]
            else:
                x = ntt_def([v % m for v in d], psis[mi], m)
            for c in range(2):
                kr = key[i][c][mi]
                for a in range(n):
                    acc[c][jj][a] += x[a] * int(kr[a])
    out = [[list(map(int, ct[c][j])) for j in range(L)] for c in range(2)]
    half = P >> 1
    for c in range(2):
        u = intt_def([v % P for v in acc[c][L]], psis[k - 1], P)
        u = [(v + half) % P for v in u]
        for j in range(L):
            q = primes[j]
            r = [((v % q) - (half % q)) % q for v in u]
            rh = ntt_def(r, psis[j], q)
            pinv = pow(P % q, -1, q)
            for a in range(n):
                out[c][j][a] = (out[c][j][a] + (acc[c][j][a] - rh[a]) * pinv) % q
    return out


def rescale_floor(ct, primes, psis, L):
    """App. A.9, SEAL 3.4.x floor variant.  ct [size][L][n] -> [size][L-1][n]"""
    last = L - 1
    ql = primes[last]
    out = []
    for poly in ct:
        d = intt_def(poly[last], psis[last], ql)
        rows = []
        for j in range(last):
            q = primes[j]
            x = ntt_def([v % q for v in d], psis[j], q)
            qinv = pow(ql % q, -1, q)
            rows.append([((int(poly[j][a]) - x[a]) * qinv) % q for a in range(len(d))])
        out.append(rows)
    return out

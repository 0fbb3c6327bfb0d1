/*
 * ckks_oracle.c -- CPU ORACLE (test infrastructure, NOT product code). See ckks_oracle.h.
 *
 * PARITY UNPINNED (no SEAL binary, no reference golden vectors -- SURVEY.md section 8c).
 * Restates SEAL 3.4.5 (upstream native/src/seal/{evaluator,keygenerator,ckks}.cpp and
 * util/{smallntt,numth,polyarithsmallmod,uintarithsmallmod,baseconverter}.cpp) as summarised in
 * SURVEY.md Appendix A; each function cites the appendix item and the reference call sites that
 * reach it (file:line under /root/reference).
 *
 * Algorithm class matches SEAL's CPU path so that this file can also serve as bench.py's
 * cpu_baseline ("port"): Harvey lazy butterflies with Shoup twiddles, 128-bit lazy accumulation in
 * key switching, Barrett reduction with floor(2^128/q).
 */
#include "ckks_oracle.h"

#include <assert.h>
#include <malloc.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------------------------------
 * scalar modular arithmetic (App. A.4)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t q;
    uint64_t r0, r1; /* floor(2^128/q) = r1*2^64 + r0 (SEAL SmallModulus::const_ratio) */
} mod_t;

static void mod_init(mod_t *m, uint64_t q)
{
    m->q = q;
    /* floor(2^128 / q) without 256-bit arithmetic: 2^128 = (2^128-1) + 1; q is never a power of two */
    u128 all = ~(u128)0;
    u128 r = all / q;
    m->r0 = (uint64_t)r;
    m->r1 = (uint64_t)(r >> 64);
}

static inline uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) >> 64); }

/* SEAL barrett_reduce_128 */
static inline uint64_t barrett128(u128 x, const mod_t *m)
{
    uint64_t x0 = (uint64_t)x, x1 = (uint64_t)(x >> 64);
    uint64_t carry = mulhi64(x0, m->r0);
    u128 t = (u128)x0 * m->r1;
    uint64_t tmp1 = (uint64_t)t + carry;
    uint64_t tmp3 = (uint64_t)(t >> 64) + (tmp1 < carry);
    t = (u128)x1 * m->r0;
    uint64_t lo = (uint64_t)t;
    uint64_t s = tmp1 + lo;
    carry = (uint64_t)(t >> 64) + (s < lo);
    uint64_t qhat = x1 * m->r1 + tmp3 + carry;
    uint64_t res = x0 - qhat * m->q;
    return res >= m->q ? res - m->q : res;
}

/* SEAL barrett_reduce_63: x < 2^63 */
static inline uint64_t barrett64(uint64_t x, const mod_t *m)
{
    uint64_t qhat = mulhi64(x, m->r1);
    uint64_t res = x - qhat * m->q;
    return res >= m->q ? res - m->q : res;
}

static inline uint64_t addmod(uint64_t a, uint64_t b, uint64_t q)
{
    uint64_t s = a + b;
    return s >= q ? s - q : s;
}
static inline uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }
static inline uint64_t mulmod_m(uint64_t a, uint64_t b, const mod_t *m) { return barrett128((u128)a * b, m); }

uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }

uint64_t orc_powmod(uint64_t a, uint64_t e, uint64_t q)
{
    uint64_t r = 1 % q;
    a %= q;
    while (e) {
        if (e & 1) r = orc_mulmod(r, a, q);
        a = orc_mulmod(a, a, q);
        e >>= 1;
    }
    return r;
}

uint64_t orc_invmod(uint64_t a, uint64_t q) { return orc_powmod(a, q - 2, q); /* q prime */ }

/* deterministic Miller-Rabin for 64-bit (SEAL uses probabilistic MR; same answers on primes) */
int orc_is_prime(uint64_t n)
{
    if (n < 2) return 0;
    static const uint64_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (size_t i = 0; i < sizeof small / sizeof *small; i++) {
        if (n == small[i]) return 1;
        if (n % small[i] == 0) return 0;
    }
    uint64_t d = n - 1;
    int r = 0;
    while (!(d & 1)) {
        d >>= 1;
        r++;
    }
    for (size_t i = 0; i < sizeof small / sizeof *small; i++) {
        uint64_t x = orc_powmod(small[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int j = 1; j < r; j++) {
            x = orc_mulmod(x, x, n);
            if (x == n - 1) {
                comp = 0;
                break;
            }
        }
        if (comp) return 0;
    }
    return 1;
}

/* App. A.3: get_primes walks down from 2^b - 2N + 1 in steps of 2N, collecting primes > 2^(b-1) in
 * descending order; CoeffModulus::Create then hands them out smallest-first per bit size.
 * Reached from e.g. /root/reference/linear_transformation2.cpp:229-233, matrix_multiplication.cpp:144-150. */
int orc_coeff_modulus_create(uint64_t N, const int *bit_sizes, int nbits, uint64_t *out)
{
    int count[64] = {0};
    for (int i = 0; i < nbits; i++) {
        if (bit_sizes[i] < 2 || bit_sizes[i] > 60) return -1;
        count[bit_sizes[i]]++;
    }
    uint64_t *table[64] = {0};
    int have[64] = {0};
    int rc = 0;
    for (int b = 2; b <= 60 && !rc; b++) {
        if (!count[b]) continue;
        table[b] = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)count[b]);
        uint64_t factor = 2 * N;
        uint64_t value = ((uint64_t)1 << b) - factor + 1;
        uint64_t lower = (uint64_t)1 << (b - 1);
        while (have[b] < count[b] && value > lower) {
            if (orc_is_prime(value)) table[b][have[b]++] = value;
            value -= factor;
        }
        if (have[b] < count[b]) rc = -2;
    }
    if (!rc)
        for (int i = 0; i < nbits; i++) {
            int b = bit_sizes[i];
            out[i] = table[b][--have[b]]; /* .back() then pop_back(): smallest unused first */
        }
    for (int b = 0; b < 64; b++) free(table[b]);
    return rc;
}

/* App. A.5: smallest integer of exact order 2N mod q (try_minimal_primitive_root). */
uint64_t orc_min_primitive_root(uint64_t two_n, uint64_t q)
{
    if ((q - 1) % two_n) return 0;
    uint64_t cof = (q - 1) / two_n;
    uint64_t root = 0;
    for (uint64_t g = 2; g < q; g++) {
        uint64_t cand = orc_powmod(g, cof, q);
        if (orc_powmod(cand, two_n >> 1, q) == q - 1) { /* exact order 2N (2N is a power of two) */
            root = cand;
            break;
        }
    }
    if (!root) return 0;
    /* all primitive 2N-th roots are the odd powers of root; take the minimum */
    uint64_t gen_sq = orc_mulmod(root, root, q);
    uint64_t cur = root, best = root;
    for (uint64_t i = 0; i < two_n / 2; i++) {
        if (cur < best) best = cur;
        cur = orc_mulmod(cur, gen_sq, q);
    }
    return best;
}

/* ------------------------------------------------------------------------------------------------
 * context
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    mod_t m;
    uint64_t psi;
    uint64_t *root, *sroot;   /* root[bitrev(i)] = psi^i, Shoup-scaled companion */
    uint64_t *iroot, *siroot; /* iroot[idx] = root[idx]^-1 */
    uint64_t ninv, sninv;
} ntt_tab;

struct orc_ctx {
    uint64_t N;
    int logn;
    int k;
    ntt_tab *t;
};

static inline uint32_t bitrev(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

static inline uint64_t shoup(uint64_t w, uint64_t q) { return (uint64_t)(((u128)w << 64) / q); }

orc_ctx *orc_ctx_create(uint64_t N, const uint64_t *primes, int k)
{
    if (N < 4 || (N & (N - 1))) return NULL;
    /* Every evaluator function takes its multi-MB temporaries from malloc, as SEAL takes them from its MemoryPool.  By
     * default glibc serves blocks above 128 KiB with mmap / munmap -- a page-zeroing system call pair per temporary that
     * serialises all threads of the many-core CPU baseline (bench.py cpu_baseline) on the address-space lock.  Keep such
     * blocks in the per-thread arenas instead, so the baseline measures arithmetic, not the allocator. */
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof *c);
    c->N = N;
    c->k = k;
    int logn = 0;
    while (((uint64_t)1 << logn) < N) logn++;
    c->logn = logn;
    c->t = (ntt_tab *)calloc((size_t)k, sizeof(ntt_tab));
    for (int j = 0; j < k; j++) {
        ntt_tab *t = &c->t[j];
        uint64_t q = primes[j];
        mod_init(&t->m, q);
        t->psi = orc_min_primitive_root(2 * N, q);
        if (!t->psi) {
            orc_ctx_destroy(c);
            return NULL;
        }
        t->root = (uint64_t *)malloc(sizeof(uint64_t) * N * 4);
        t->sroot = t->root + N;
        t->iroot = t->root + 2 * N;
        t->siroot = t->root + 3 * N;
        uint64_t p = 1;
        for (uint64_t i = 0; i < N; i++) {
            uint32_t r = bitrev((uint32_t)i, logn);
            t->root[r] = p;
            p = orc_mulmod(p, t->psi, q);
        }
        for (uint64_t i = 0; i < N; i++) {
            t->sroot[i] = shoup(t->root[i], q);
            t->iroot[i] = orc_invmod(t->root[i], q);
            t->siroot[i] = shoup(t->iroot[i], q);
        }
        t->ninv = orc_invmod(N % q, q);
        t->sninv = shoup(t->ninv, q);
    }
    return c;
}

void orc_ctx_destroy(orc_ctx *c)
{
    if (!c) return;
    for (int j = 0; j < c->k; j++) free(c->t[j].root);
    free(c->t);
    free(c);
}

uint64_t orc_ctx_n(const orc_ctx *c) { return c->N; }
int orc_ctx_k(const orc_ctx *c) { return c->k; }
uint64_t orc_ctx_prime(const orc_ctx *c, int j) { return c->t[j].m.q; }
uint64_t orc_ctx_psi(const orc_ctx *c, int j) { return c->t[j].psi; }

/* ------------------------------------------------------------------------------------------------
 * NTT (App. A.5).  Reached from every rotate_vector (/root/reference/helper.h:244,255), relinearize
 * (helper.h:440) and rescale (matrix_multiplication.cpp:71-72).
 * ---------------------------------------------------------------------------------------------- */
/* x*w mod q in [0,2q) given ws = floor(w*2^64/q); valid for any 64-bit x */
static inline uint64_t shoup_mul_lazy(uint64_t x, uint64_t w, uint64_t ws, uint64_t q)
{
    return x * w - mulhi64(x, ws) * q;
}

void orc_ntt_fwd(const orc_ctx *c, int j, uint64_t *a)
{
    const ntt_tab *tb = &c->t[j];
    const uint64_t q = tb->m.q, two_q = 2 * q;
    uint64_t n = c->N, t = n >> 1;
    for (uint64_t m = 1; m < n; m <<= 1, t >>= 1) {
        for (uint64_t i = 0; i < m; i++) {
            uint64_t W = tb->root[m + i], Ws = tb->sroot[m + i];
            uint64_t *X = a + 2 * i * t, *Y = X + t;
            for (uint64_t jj = 0; jj < t; jj++) {
                uint64_t x = X[jj] - (two_q & (uint64_t)(-(int64_t)(X[jj] >= two_q)));
                uint64_t tt = shoup_mul_lazy(Y[jj], W, Ws, q);
                X[jj] = x + tt;
                Y[jj] = x + two_q - tt;
            }
        }
    }
    for (uint64_t i = 0; i < n; i++) {
        uint64_t x = a[i];
        x -= two_q & (uint64_t)(-(int64_t)(x >= two_q));
        x -= q & (uint64_t)(-(int64_t)(x >= q));
        a[i] = x;
    }
}

void orc_ntt_inv(const orc_ctx *c, int j, uint64_t *a)
{
    const ntt_tab *tb = &c->t[j];
    const uint64_t q = tb->m.q, two_q = 2 * q;
    uint64_t n = c->N, t = 1;
    for (uint64_t h = n >> 1; h >= 1; h >>= 1, t <<= 1) {
        int last = (h == 1);
        for (uint64_t i = 0; i < h; i++) {
            uint64_t W = tb->iroot[h + i], Ws = tb->siroot[h + i];
            if (last) { /* fold N^-1 into the last stage */
                W = mulmod_m(W, tb->ninv, &tb->m);
                Ws = shoup(W, q);
            }
            uint64_t *X = a + 2 * i * t, *Y = X + t;
            for (uint64_t jj = 0; jj < t; jj++) {
                uint64_t u = X[jj], v = Y[jj]; /* both in [0,2q) */
                uint64_t s = u + v;
                s -= two_q & (uint64_t)(-(int64_t)(s >= two_q));
                uint64_t d = u + two_q - v;
                X[jj] = last ? shoup_mul_lazy(s, tb->ninv, tb->sninv, q) : s;
                Y[jj] = shoup_mul_lazy(d, W, Ws, q);
            }
        }
        if (last) break;
    }
    for (uint64_t i = 0; i < n; i++) {
        uint64_t x = a[i];
        x -= q & (uint64_t)(-(int64_t)(x >= q));
        a[i] = x;
    }
}

void orc_ntt_naive(const orc_ctx *c, int j, const uint64_t *in, uint64_t *out)
{
    const ntt_tab *tb = &c->t[j];
    uint64_t q = tb->m.q, n = c->N;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t e = 2 * (uint64_t)bitrev((uint32_t)i, c->logn) + 1;
        uint64_t x = orc_powmod(tb->psi, e, q);
        uint64_t acc = 0, xp = 1;
        for (uint64_t kx = 0; kx < n; kx++) {
            acc = addmod(acc, orc_mulmod(in[kx] % q, xp, q), q);
            xp = orc_mulmod(xp, x, q);
        }
        out[i] = acc;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Galois (App. A.7).  rotate_vector call sites: /root/reference/helper.h:216,227,244,255,316,352,455,474.
 * ---------------------------------------------------------------------------------------------- */
uint64_t orc_galois_elt_from_step(uint64_t N, int step)
{
    uint64_t m = 2 * N;
    if (step == 0) return m - 1;
    uint64_t pos;
    if (step < 0)
        pos = (N >> 1) - (uint64_t)(-step);
    else
        pos = (uint64_t)step;
    uint64_t e = 1;
    for (uint64_t i = 0; i < pos; i++) e = (e * 3) & (m - 1);
    return e;
}

int orc_naf_steps(uint64_t N, int step, int *out, int max_out)
{
    (void)N;
    int sign = step < 0;
    int value = step < 0 ? -step : step;
    int cnt = 0;
    for (int i = 0; value; i++) {
        int zi = (value & 1) ? 2 - (value & 3) : 0;
        value = (value - zi) >> 1;
        if (zi) {
            if (cnt < max_out) out[cnt] = (sign ? -zi : zi) * (1 << i);
            cnt++;
        }
    }
    return cnt;
}

void orc_galois_table(uint64_t N, uint64_t elt, uint32_t *table)
{
    int logn = 0;
    while (((uint64_t)1 << logn) < N) logn++;
    uint64_t m = 2 * N;
    for (uint64_t i = 0; i < N; i++) {
        uint64_t rev = bitrev((uint32_t)i, logn);
        uint64_t raw = (elt * (2 * rev + 1)) & (m - 1);
        table[i] = bitrev((uint32_t)((raw - 1) >> 1), logn);
    }
}

void orc_apply_galois_ntt(const orc_ctx *c, uint64_t elt, const uint64_t *in, uint64_t *out)
{
    uint32_t *tab = (uint32_t *)malloc(sizeof(uint32_t) * c->N);
    orc_galois_table(c->N, elt, tab);
    for (uint64_t i = 0; i < c->N; i++) out[i] = in[tab[i]];
    free(tab);
}

/* ------------------------------------------------------------------------------------------------
 * element-wise Evaluator ops (App. A.6)
 * ---------------------------------------------------------------------------------------------- */
/* add: /root/reference/helper.h:219,247,259,464,475 */
void orc_add(const orc_ctx *c, int L, int size, const uint64_t *a, const uint64_t *b, uint64_t *out)
{
    uint64_t n = c->N;
    for (int p = 0; p < size; p++)
        for (int j = 0; j < L; j++) {
            uint64_t q = c->t[j].m.q;
            size_t o = ((size_t)p * L + j) * n;
            for (uint64_t i = 0; i < n; i++) out[o + i] = addmod(a[o + i], b[o + i], q);
        }
}

/* sub: /root/reference/logistic_regression_ckks.cpp:288,341 */
void orc_sub(const orc_ctx *c, int L, int size, const uint64_t *a, const uint64_t *b, uint64_t *out)
{
    uint64_t n = c->N;
    for (int p = 0; p < size; p++)
        for (int j = 0; j < L; j++) {
            uint64_t q = c->t[j].m.q;
            size_t o = ((size_t)p * L + j) * n;
            for (uint64_t i = 0; i < n; i++) out[o + i] = submod(a[o + i], b[o + i], q);
        }
}

/* negate_inplace: /root/reference/logistic_regression_ckks.cpp:342 */
void orc_negate(const orc_ctx *c, int L, int size, const uint64_t *a, uint64_t *out)
{
    uint64_t n = c->N;
    for (int p = 0; p < size; p++)
        for (int j = 0; j < L; j++) {
            uint64_t q = c->t[j].m.q;
            size_t o = ((size_t)p * L + j) * n;
            for (uint64_t i = 0; i < n; i++) out[o + i] = a[o + i] ? q - a[o + i] : 0;
        }
}

/* add_plain_inplace: /root/reference/polynomial.cpp:210, vector_ops.cpp:268 */
void orc_add_plain(const orc_ctx *c, int L, int size, const uint64_t *ct, const uint64_t *pt, uint64_t *out)
{
    uint64_t n = c->N;
    if (out != ct) memcpy(out, ct, sizeof(uint64_t) * (size_t)size * L * n);
    for (int j = 0; j < L; j++) {
        uint64_t q = c->t[j].m.q;
        size_t o = (size_t)j * n;
        for (uint64_t i = 0; i < n; i++) out[o + i] = addmod(ct[o + i], pt[o + i], q);
    }
}

/* multiply_plain: /root/reference/helper.h:250,256,271,347 */
void orc_multiply_plain(const orc_ctx *c, int L, int size, const uint64_t *ct, const uint64_t *pt, uint64_t *out)
{
    uint64_t n = c->N;
    for (int p = 0; p < size; p++)
        for (int j = 0; j < L; j++) {
            const mod_t *m = &c->t[j].m;
            size_t o = ((size_t)p * L + j) * n, po = (size_t)j * n;
            for (uint64_t i = 0; i < n; i++) out[o + i] = mulmod_m(ct[o + i], pt[po + i], m);
        }
}

/* multiply (CKKS tensor product in NTT domain): /root/reference/helper.h:222,228,432;
 * matrix_multiplication.cpp:104,127 */
void orc_multiply(const orc_ctx *c, int L, int size_a, const uint64_t *a, int size_b, const uint64_t *b,
                  uint64_t *out)
{
    uint64_t n = c->N;
    int size_o = size_a + size_b - 1;
    uint64_t *tmp = (uint64_t *)calloc((size_t)size_o * L * n, sizeof(uint64_t));
    for (int pa = 0; pa < size_a; pa++)
        for (int pb = 0; pb < size_b; pb++)
            for (int j = 0; j < L; j++) {
                const mod_t *m = &c->t[j].m;
                size_t oa = ((size_t)pa * L + j) * n, ob = ((size_t)pb * L + j) * n;
                size_t oo = ((size_t)(pa + pb) * L + j) * n;
                for (uint64_t i = 0; i < n; i++)
                    tmp[oo + i] = addmod(tmp[oo + i], mulmod_m(a[oa + i], b[ob + i], m), m->q);
            }
    memcpy(out, tmp, sizeof(uint64_t) * (size_t)size_o * L * n);
    free(tmp);
}

int orc_is_transparent(const orc_ctx *c, int L, int size, const uint64_t *ct)
{
    size_t n = (size_t)(size - 1) * L * c->N, off = (size_t)L * c->N;
    for (size_t i = 0; i < n; i++)
        if (ct[off + i]) return 0;
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * key switching (App. A.8): RNS digits, one special prime P = primes[k-1], rounded division by P.
 * ---------------------------------------------------------------------------------------------- */
/* the mod-down of App. A.8 on CANONICAL accumulators S[2][L+1][N] (rows q_0..q_(L-1), P), added into ct[2][L][N]:
 * ct[c][j] += (S[c][j] - NTT_j(((INTT_P(S[c][P]) + P/2) mod P) mod q_j - P/2 mod q_j)) * P^-1 mod q_j */
static void moddown_into(const orc_ctx *c, int L, const uint64_t *S, uint64_t *ct)
{
    const uint64_t n = c->N;
    const int sp = c->k - 1, nm = L + 1;
    const mod_t *mp = &c->t[sp].m;
    const uint64_t half = mp->q >> 1;
    uint64_t *d = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint64_t *x = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (int cc = 0; cc < 2; cc++) {
        memcpy(d, S + ((size_t)cc * nm + L) * n, sizeof(uint64_t) * n);
        orc_ntt_inv(c, sp, d);
        for (uint64_t a = 0; a < n; a++) d[a] = barrett64(d[a] + half, mp);
        for (int j = 0; j < L; j++) {
            const mod_t *m = &c->t[j].m;
            uint64_t half_j = barrett64(half, m);
            uint64_t pinv = orc_invmod(mp->q % m->q, m->q);
            for (uint64_t a = 0; a < n; a++) x[a] = submod(barrett64(d[a], m), half_j, m->q);
            orc_ntt_fwd(c, j, x);
            const uint64_t *aj = S + ((size_t)cc * nm + j) * n;
            uint64_t *dst = ct + ((size_t)cc * L + j) * n;
            for (uint64_t a = 0; a < n; a++) {
                uint64_t v = submod(aj[a], x[a], m->q);
                dst[a] = addmod(dst[a], mulmod_m(v, pinv, m), m->q);
            }
        }
    }
    free(d);
    free(x);
}

/* tab == NULL: SEAL's key switch of `target` (App. A.8).
 * tab != NULL: the HOISTED variant: `target` is the UNROTATED c1; its digits are extended to every
 * modulus once and each extended row is read through the Galois gather table, i.e. the automorphism is applied AFTER
 * the decomposition.  With flip == NULL this is the uncorrected sum (what rounds 1-3 shipped as a fast mode) -- other
 * words than SEAL's: where the automorphism negates a coefficient, SEAL's digit is q_i - a (positive lift), this one -a. */
/* flip != NULL (with tab): the EXACT hoisted form (csrc/hefx_keyswitch.hip, ks_mac_exact_kernel).  flip[a] = 1 where the
 * automorphism negates the coefficient that lands at position a.  SEAL's digit of the rotated polynomial is
 * sigma(T_i) + q_i * flip as an integer vector (T_i = the unrotated digit, sigma the signed permutation over Z) unless T_i
 * has a zero coefficient, so its transform modulo m is the gathered row plus (q_i mod m) * NTT_m(flip): adding that term
 * to the hoisted sum gives SEAL's accumulator modulo m, hence SEAL's bits.  Returns 1 if a zero coefficient was met (the
 * identity does not hold then and the caller must take the regular sequence), else 0. */
static int switch_key_impl(const orc_ctx *c, int L, uint64_t *ct, const uint64_t *target, const uint64_t *key,
                           const uint32_t *tab, const uint64_t *flip)
{
    const uint64_t n = c->N;
    const int k = c->k, sp = k - 1; /* special-prime index at key level */
    const int nm = L + 1;           /* moduli touched: q_0..q_{L-1}, P */
    /* per-thread scratch kept across calls (as SEAL's MemoryPool does): a fresh multi-MB calloc per call
     * serialises all-core baseline runs on the kernel's mmap lock */
    static __thread u128 *tl_acc = NULL;
    static __thread size_t tl_acc_words = 0;
    const size_t acc_words = (size_t)2 * nm * n;
    if (tl_acc_words < acc_words) {
        free(tl_acc);
        tl_acc = (u128 *)malloc(acc_words * sizeof(u128));
        tl_acc_words = acc_words;
    }
    u128 *acc = tl_acc;
    memset(acc, 0, acc_words * sizeof(u128));
    uint64_t *d = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint64_t *x = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint64_t *xg = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint64_t *w = flip ? (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nm * n) : NULL; /* w[jj] = NTT_m(flip) */
    int zero_seen = 0;
    for (int jj = 0; flip && jj < nm; jj++) {
        memcpy(w + (size_t)jj * n, flip, sizeof(uint64_t) * n);
        orc_ntt_fwd(c, jj < L ? jj : sp, w + (size_t)jj * n);
    }

    for (int i = 0; i < L; i++) {
        memcpy(d, target + (size_t)i * n, sizeof(uint64_t) * n);
        orc_ntt_inv(c, i, d); /* digit in coefficient form, [0,q_i) */
        for (uint64_t a = 0; flip && a < n; a++) zero_seen |= d[a] == 0;
        for (int jj = 0; jj < nm; jj++) {
            int mi = jj < L ? jj : sp; /* modulus / key-row index at key level */
            const mod_t *m = &c->t[mi].m;
            const uint64_t *xs;
            if (mi == i) {
                xs = target + (size_t)i * n; /* already NTT mod q_i */
            } else {
                if (c->t[i].m.q > m->q)
                    for (uint64_t a = 0; a < n; a++) x[a] = barrett64(d[a], m);
                else
                    memcpy(x, d, sizeof(uint64_t) * n);
                orc_ntt_fwd(c, mi, x);
                xs = x;
            }
            if (tab) {
                for (uint64_t a = 0; a < n; a++) xg[a] = xs[tab[a]];
                xs = xg;
            }
            for (int cc = 0; cc < 2; cc++) {
                const uint64_t *kr = key + ((((size_t)i * 2 + cc) * k) + mi) * n;
                u128 *ac = acc + ((size_t)cc * nm + jj) * n;
                for (uint64_t a = 0; a < n; a++) ac[a] += (u128)xs[a] * kr[a];
                if (flip && mi != i) { /* + (q_i mod m) * NTT_m(flip) * key */
                    const uint64_t qim = c->t[i].m.q % m->q, *wj = w + (size_t)jj * n;
                    for (uint64_t a = 0; a < n; a++) ac[a] += (u128)mulmod_m(qim, wj[a], m) * kr[a];
                }
            }
        }
    }
    free(w);

    const mod_t *mp = &c->t[sp].m;
    const uint64_t half = mp->q >> 1;
    for (int cc = 0; cc < 2; cc++) {
        /* u = INTT_P(acc mod P); u = (u + floor(P/2)) mod P */
        u128 *ap = acc + ((size_t)cc * nm + L) * n;
        for (uint64_t a = 0; a < n; a++) d[a] = barrett128(ap[a], mp);
        orc_ntt_inv(c, sp, d);
        for (uint64_t a = 0; a < n; a++) d[a] = barrett64(d[a] + half, mp);
        for (int j = 0; j < L; j++) {
            const ntt_tab *tj = &c->t[j];
            const mod_t *m = &tj->m;
            uint64_t half_j = barrett64(half, m);
            uint64_t pinv = orc_invmod(mp->q % m->q, m->q);
            for (uint64_t a = 0; a < n; a++) x[a] = submod(barrett64(d[a], m), half_j, m->q);
            orc_ntt_fwd(c, j, x);
            u128 *aj = acc + ((size_t)cc * nm + j) * n;
            uint64_t *dst = ct + ((size_t)cc * L + j) * n;
            for (uint64_t a = 0; a < n; a++) {
                uint64_t v = submod(barrett128(aj[a], m), x[a], m->q);
                dst[a] = addmod(dst[a], mulmod_m(v, pinv, m), m->q);
            }
        }
    }
    free(d);
    free(x);
    free(xg);
    return zero_seen;
}

void orc_switch_key(const orc_ctx *c, int L, uint64_t *ct, const uint64_t *target, const uint64_t *key)
{
    (void)switch_key_impl(c, L, ct, target, key, NULL, NULL);
}

/* hoisted rotation: c0' = perm(c0) + ks0, c1' = ks1 with the key switch of the hoisted variant above */
void orc_apply_galois_hoisted(const orc_ctx *c, int L, const uint64_t *ct_in, uint64_t elt, const uint64_t *key,
                              uint64_t *ct_out)
{
    const uint64_t n = c->N;
    uint32_t *tab = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint64_t *res = (uint64_t *)calloc((size_t)2 * L * n, sizeof(uint64_t));
    orc_galois_table(n, elt, tab);
    for (int j = 0; j < L; j++) {
        const uint64_t *s0 = ct_in + (size_t)j * n;
        uint64_t *r0 = res + (size_t)j * n;
        for (uint64_t i = 0; i < n; i++) r0[i] = s0[tab[i]];
    }
    (void)switch_key_impl(c, L, res, ct_in + (size_t)L * n, key, tab, NULL);
    memcpy(ct_out, res, sizeof(uint64_t) * (size_t)2 * L * n);
    free(tab);
    free(res);
}

/* EXACT hoisted rotation (test infrastructure for csrc/hefx_keyswitch.hip ks_mac_exact_kernel): the hoisted sequence
 * plus the flip-mask term.  Coefficient a of p(X^elt) is +-p_s with s = a * elt^-1 mod 2N, negative when s >= N.
 * Returns 0 when the identity applied (ct_out then equals orc_apply_galois's words -- tests/test_oracle_cpu.py checks
 * exactly that), 1 when c1 had a zero coefficient in some digit: ct_out is then computed by orc_apply_galois itself. */
int orc_apply_galois_hoisted_exact(const orc_ctx *c, int L, const uint64_t *ct_in, uint64_t elt, const uint64_t *key,
                                   uint64_t *ct_out)
{
    const uint64_t n = c->N;
    uint32_t *tab = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint64_t *res = (uint64_t *)calloc((size_t)2 * L * n, sizeof(uint64_t));
    uint64_t *flip = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint64_t ginv = elt; /* odd: its own inverse mod 8; Newton doubles the correct bits */
    for (int r = 0; r < 5; r++) ginv *= 2 - elt * ginv;
    ginv &= 2 * n - 1;
    for (uint64_t a = 0; a < n; a++) flip[a] = ((a * ginv) & (2 * n - 1)) >= n;
    orc_galois_table(n, elt, tab);
    for (int j = 0; j < L; j++) {
        const uint64_t *s0 = ct_in + (size_t)j * n;
        uint64_t *r0 = res + (size_t)j * n;
        for (uint64_t i = 0; i < n; i++) r0[i] = s0[tab[i]];
    }
    const int zero = switch_key_impl(c, L, res, ct_in + (size_t)L * n, key, tab, flip);
    if (zero)
        orc_apply_galois(c, L, ct_in, elt, key, ct_out);
    else
        memcpy(ct_out, res, sizeof(uint64_t) * (size_t)2 * L * n);
    free(tab);
    free(res);
    free(flip);
    return zero;
}

/* DOUBLE-HOISTED linear transform, core (csrc/hefx_keyswitch.hip lt2_mac_kernel), top data level
 * L = k-1: given ct_new [2][L][N], key-level diagonals diag[d][k][N] (NTT), Galois elements elt[1..d-1] and their
 * keys key[l] ([k-1][2][k][N], l = 1..d-1 stored at index l-1):
 *   S[c][m]  = sum_l diag_l[m] * (sum_i x_i[m][tab_l] * key_l[i][c][m])   over all k moduli (x_i = digit i of c1 extended)
 *   out      = (diag_0*c0 + sum_l diag_l * c0[tab_l], diag_0*c1) + moddown(S)
 * Every operation is exact modulo its prime, so the reduction points do not matter; the single mod-down rounds once. */
void orc_lt_double_hoisted_core(const orc_ctx *c, int L, const uint64_t *ct_new, int d, const uint64_t *diag,
                                const uint64_t *elts, const uint64_t *keys, uint64_t *out)
{
    const uint64_t n = c->N;
    const int k = c->k, sp = k - 1, nm = L + 1;
    uint32_t *tab = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint64_t *x = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)L * nm * n); /* x[i][jj] */
    uint64_t *dg = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint64_t *S = (uint64_t *)calloc((size_t)2 * nm * n, sizeof(uint64_t));
    const uint64_t *c0 = ct_new, *c1 = ct_new + (size_t)L * n;
    for (int i = 0; i < L; i++) {
        memcpy(dg, c1 + (size_t)i * n, sizeof(uint64_t) * n);
        orc_ntt_inv(c, i, dg);
        for (int jj = 0; jj < nm; jj++) {
            const int mi = jj < L ? jj : sp;
            uint64_t *xr = x + ((size_t)i * nm + jj) * n;
            if (mi == i) {
                memcpy(xr, c1 + (size_t)i * n, sizeof(uint64_t) * n);
            } else {
                const uint64_t q = c->t[mi].m.q;
                for (uint64_t a = 0; a < n; a++) xr[a] = dg[a] % q;
                orc_ntt_fwd(c, mi, xr);
            }
        }
    }
    /* out = diag_0 * ct_new (both polys, data primes) */
    for (int cc = 0; cc < 2; cc++)
        for (int j = 0; j < L; j++) {
            const uint64_t q = c->t[j].m.q;
            const uint64_t *src = ct_new + ((size_t)cc * L + j) * n, *d0 = diag + (size_t)j * n;
            uint64_t *o = out + ((size_t)cc * L + j) * n;
            for (uint64_t a = 0; a < n; a++) o[a] = (uint64_t)((u128)src[a] * d0[a] % q);
        }
    for (int l = 1; l < d; l++) {
        const uint64_t *dl = diag + (size_t)l * k * n;
        const uint64_t *key = keys + (size_t)(l - 1) * (k - 1) * 2 * k * n;
        orc_galois_table(n, elts[l], tab);
        for (int jj = 0; jj < nm; jj++) {
            const int mi = jj < L ? jj : sp;
            const uint64_t q = c->t[mi].m.q;
            for (int cc = 0; cc < 2; cc++) {
                uint64_t *Sr = S + ((size_t)cc * nm + jj) * n;
                for (uint64_t a = 0; a < n; a++) {
                    u128 acc = 0;
                    for (int i = 0; i < L; i++) {
                        const uint64_t xv = x[((size_t)i * nm + jj) * n + tab[a]];
                        const uint64_t kv = key[((((size_t)i * 2 + cc) * k) + mi) * n + a];
                        acc += (u128)xv * kv % q;
                    }
                    const uint64_t inner = (uint64_t)(acc % q);
                    Sr[a] = (uint64_t)(((u128)Sr[a] + (u128)inner * dl[(size_t)mi * n + a] % q) % q);
                }
            }
        }
        for (int j = 0; j < L; j++) { /* C0 += diag_l * c0[tab_l] */
            const uint64_t q = c->t[j].m.q;
            uint64_t *o = out + (size_t)j * n;
            for (uint64_t a = 0; a < n; a++)
                o[a] = (uint64_t)(((u128)o[a] + (u128)c0[(size_t)j * n + tab[a]] * dl[(size_t)j * n + a] % q) % q);
        }
    }
    moddown_into(c, L, S, out);
    free(tab);
    free(x);
    free(dg);
    free(S);
}

/* apply_galois_inplace, CKKS size-2 (App. A.7): c0' = perm(c0) + ks0, c1' = ks1 */
void orc_apply_galois(const orc_ctx *c, int L, const uint64_t *ct_in, uint64_t elt, const uint64_t *key,
                      uint64_t *ct_out)
{
    const uint64_t n = c->N;
    uint32_t *tab = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint64_t *target = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)L * n);
    uint64_t *res = (uint64_t *)calloc((size_t)2 * L * n, sizeof(uint64_t));
    orc_galois_table(n, elt, tab);
    for (int j = 0; j < L; j++) {
        const uint64_t *s0 = ct_in + (size_t)j * n, *s1 = ct_in + ((size_t)L + j) * n;
        uint64_t *r0 = res + (size_t)j * n, *tg = target + (size_t)j * n;
        for (uint64_t i = 0; i < n; i++) {
            r0[i] = s0[tab[i]];
            tg[i] = s1[tab[i]];
        }
    }
    orc_switch_key(c, L, res, target, key);
    memcpy(ct_out, res, sizeof(uint64_t) * (size_t)2 * L * n);
    free(tab);
    free(target);
    free(res);
}

/* relinearize_inplace (App. A.6): /root/reference/helper.h:440,541; polynomial.cpp:92,187 */
void orc_relinearize(const orc_ctx *c, int L, const uint64_t *ct3, const uint64_t *key, uint64_t *ct2)
{
    const uint64_t n = c->N;
    uint64_t *res = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)2 * L * n);
    memcpy(res, ct3, sizeof(uint64_t) * (size_t)2 * L * n);
    orc_switch_key(c, L, res, ct3 + (size_t)2 * L * n, key);
    memcpy(ct2, res, sizeof(uint64_t) * (size_t)2 * L * n);
    free(res);
}

/* rescale_to_next_inplace (App. A.9): /root/reference/matrix_multiplication.cpp:71-72, helper.h:441.
 * SEAL 3.4.x floor_last_coeff_modulus_ntt_inplace: out_j = (c_j - NTT_j(INTT(c_last) mod q_j)) * q_last^-1.
 * rounded!=0 adds the floor(q_last/2) term of SEAL >= 3.5 (kept for fixtures; not the 3.4.5 behaviour). */
void orc_rescale(const orc_ctx *c, int L, int size, const uint64_t *in, uint64_t *out, int rounded)
{
    const uint64_t n = c->N;
    const int last = L - 1;
    const mod_t *ml = &c->t[last].m;
    uint64_t *d = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint64_t *x = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint64_t half = ml->q >> 1;
    for (int p = 0; p < size; p++) {
        memcpy(d, in + ((size_t)p * L + last) * n, sizeof(uint64_t) * n);
        orc_ntt_inv(c, last, d);
        if (rounded)
            for (uint64_t a = 0; a < n; a++) d[a] = barrett64(d[a] + half, ml);
        for (int j = 0; j < last; j++) {
            const mod_t *m = &c->t[j].m;
            uint64_t qinv = orc_invmod(ml->q % m->q, m->q);
            uint64_t half_j = barrett64(half, m);
            for (uint64_t a = 0; a < n; a++) {
                uint64_t v = barrett64(d[a], m);
                x[a] = rounded ? submod(v, half_j, m->q) : v;
            }
            orc_ntt_fwd(c, j, x);
            const uint64_t *src = in + ((size_t)p * L + j) * n;
            uint64_t *dst = out + ((size_t)p * last + j) * n;
            for (uint64_t a = 0; a < n; a++) dst[a] = mulmod_m(submod(src[a], x[a], m->q), qinv, m);
        }
    }
    free(d);
    free(x);
}

/* mod_switch_to_next / mod_switch_to (CKKS, App. A.10): /root/reference/matrix_multiplication.cpp:112 */
void orc_mod_drop(const orc_ctx *c, int L_in, int L_out, int npoly, const uint64_t *in, uint64_t *out)
{
    const uint64_t n = c->N;
    for (int p = 0; p < npoly; p++)
        memmove(out + (size_t)p * L_out * n, in + (size_t)p * L_in * n, sizeof(uint64_t) * (size_t)L_out * n);
}

/* hot-loop body of Linear_Transform_Plain: /root/reference/helper.h:255-256 */
void orc_rotate_mulplain(const orc_ctx *c, int L, const uint64_t *ct_in, uint64_t elt, const uint64_t *key,
                         const uint64_t *pt, uint64_t *ct_out)
{
    orc_apply_galois(c, L, ct_in, elt, key, ct_out);
    orc_multiply_plain(c, L, 2, ct_out, pt, ct_out);
}

/* ------------------------------------------------------------------------------------------------
 * non-hot pieces for decrypted-value checks (App. A.11 / A.12); deterministic splitmix64 sampling
 * ---------------------------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline uint64_t uniform_mod(uint64_t *s, uint64_t q)
{
    /* rejection sampling on the smallest covering bit mask */
    uint64_t mask = ~(uint64_t)0 >> __builtin_clzll(q);
    for (;;) {
        uint64_t v = splitmix64(s) & mask;
        if (v < q) return v;
    }
}

void orc_fill_uniform(const orc_ctx *c, int L, int npoly, uint64_t seed, uint64_t *out)
{
    uint64_t s = seed;
    for (int p = 0; p < npoly; p++)
        for (int j = 0; j < L; j++) {
            uint64_t q = c->t[j].m.q;
            uint64_t *o = out + ((size_t)p * L + j) * c->N;
            for (uint64_t i = 0; i < c->N; i++) o[i] = uniform_mod(&s, q);
        }
}

static void sample_noise_ntt(const orc_ctx *c, int nrows, const int *rows, uint64_t *s, uint64_t *out)
{
    /* clipped normal sigma=3.2, bound 6*sigma, truncated toward zero, then NTT per row */
    const uint64_t n = c->N;
    int64_t *e = (int64_t *)malloc(sizeof(int64_t) * n);
    for (uint64_t i = 0; i < n; i++) {
        double v;
        do {
            double u1 = ((double)(splitmix64(s) >> 11) + 1.0) / 9007199254740993.0;
            double u2 = (double)(splitmix64(s) >> 11) / 9007199254740992.0;
            v = 3.2 * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        } while (fabs(v) > 19.2);
        e[i] = (int64_t)v;
    }
    for (int r = 0; r < nrows; r++) {
        uint64_t q = c->t[rows[r]].m.q;
        uint64_t *o = out + (size_t)r * n;
        for (uint64_t i = 0; i < n; i++) o[i] = e[i] >= 0 ? (uint64_t)e[i] : q - (uint64_t)(-e[i]);
        orc_ntt_fwd(c, rows[r], o);
    }
    free(e);
}

void orc_gen_secret(const orc_ctx *c, uint64_t seed, uint64_t *sk)
{
    const uint64_t n = c->N;
    uint64_t s = seed;
    int8_t *t = (int8_t *)malloc(n);
    for (uint64_t i = 0; i < n; i++) t[i] = (int8_t)(uniform_mod(&s, 3)) - 1;
    for (int j = 0; j < c->k; j++) {
        uint64_t q = c->t[j].m.q;
        uint64_t *o = sk + (size_t)j * n;
        for (uint64_t i = 0; i < n; i++) o[i] = t[i] == 1 ? 1 : (t[i] == 0 ? 0 : q - 1);
        orc_ntt_fwd(c, j, o);
    }
    free(t);
}

/* fresh symmetric encryption of zero over `nrows` key-level rows rows[]: (c0,c1) = (-(a s + e), a) */
static void encrypt_zero_rows(const orc_ctx *c, int nrows, const int *rows, const uint64_t *sk, uint64_t *s,
                              uint64_t *c0, uint64_t *c1)
{
    const uint64_t n = c->N;
    for (int r = 0; r < nrows; r++) {
        uint64_t q = c->t[rows[r]].m.q;
        for (uint64_t i = 0; i < n; i++) c1[(size_t)r * n + i] = uniform_mod(s, q);
    }
    sample_noise_ntt(c, nrows, rows, s, c0);
    for (int r = 0; r < nrows; r++) {
        const mod_t *m = &c->t[rows[r]].m;
        const uint64_t *sr = sk + (size_t)rows[r] * n;
        for (uint64_t i = 0; i < n; i++) {
            uint64_t as = mulmod_m(c1[(size_t)r * n + i], sr[i], m);
            uint64_t v = addmod(as, c0[(size_t)r * n + i], m->q);
            c0[(size_t)r * n + i] = v ? m->q - v : 0;
        }
    }
}

void orc_gen_kswitch_key(const orc_ctx *c, const uint64_t *sk, const uint64_t *new_sk, uint64_t seed,
                         uint64_t *out)
{
    const uint64_t n = c->N;
    const int k = c->k;
    uint64_t s = seed;
    int *rows = (int *)malloc(sizeof(int) * (size_t)k);
    for (int j = 0; j < k; j++) rows[j] = j;
    uint64_t P = c->t[k - 1].m.q;
    for (int i = 0; i < k - 1; i++) {
        uint64_t *c0 = out + ((size_t)i * 2 + 0) * k * n;
        uint64_t *c1 = out + ((size_t)i * 2 + 1) * k * n;
        encrypt_zero_rows(c, k, rows, sk, &s, c0, c1);
        const mod_t *m = &c->t[i].m;
        uint64_t factor = P % m->q;
        uint64_t *row = c0 + (size_t)i * n;
        const uint64_t *ns = new_sk + (size_t)i * n;
        for (uint64_t a = 0; a < n; a++) row[a] = addmod(row[a], mulmod_m(ns[a], factor, m), m->q);
    }
    free(rows);
}

void orc_gen_relin_key(const orc_ctx *c, const uint64_t *sk, uint64_t seed, uint64_t *out)
{
    const uint64_t n = c->N;
    uint64_t *s2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)c->k * n);
    for (int j = 0; j < c->k; j++)
        for (uint64_t i = 0; i < n; i++)
            s2[(size_t)j * n + i] = mulmod_m(sk[(size_t)j * n + i], sk[(size_t)j * n + i], &c->t[j].m);
    orc_gen_kswitch_key(c, sk, s2, seed, out);
    free(s2);
}

void orc_gen_galois_key(const orc_ctx *c, const uint64_t *sk, uint64_t elt, uint64_t seed, uint64_t *out)
{
    const uint64_t n = c->N;
    uint64_t *sp = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)c->k * n);
    for (int j = 0; j < c->k; j++) orc_apply_galois_ntt(c, elt, sk + (size_t)j * n, sp + (size_t)j * n);
    orc_gen_kswitch_key(c, sk, sp, seed, out);
    free(sp);
}

void orc_encrypt_sym(const orc_ctx *c, int L, const uint64_t *sk, const uint64_t *pt, uint64_t seed,
                     uint64_t *ct)
{
    const uint64_t n = c->N;
    uint64_t s = seed;
    int *rows = (int *)malloc(sizeof(int) * (size_t)L);
    for (int j = 0; j < L; j++) rows[j] = j;
    encrypt_zero_rows(c, L, rows, sk, &s, ct, ct + (size_t)L * n);
    for (int j = 0; j < L; j++) {
        uint64_t q = c->t[j].m.q;
        for (uint64_t i = 0; i < n; i++)
            ct[(size_t)j * n + i] = addmod(ct[(size_t)j * n + i], pt[(size_t)j * n + i], q);
    }
    free(rows);
}

/* Decryptor::decrypt incl. size 3 (/root/reference/matrix_multiplication.cpp:419): Horner in s */
void orc_decrypt(const orc_ctx *c, int L, int size, const uint64_t *ct, const uint64_t *sk, uint64_t *pt)
{
    const uint64_t n = c->N;
    for (int j = 0; j < L; j++) {
        const mod_t *m = &c->t[j].m;
        const uint64_t *sr = sk + (size_t)j * n;
        uint64_t *o = pt + (size_t)j * n;
        for (uint64_t i = 0; i < n; i++) {
            uint64_t acc = ct[((size_t)(size - 1) * L + j) * n + i];
            for (int p = size - 2; p >= 0; p--)
                acc = addmod(mulmod_m(acc, sr[i], m), ct[((size_t)p * L + j) * n + i], m->q);
            o[i] = acc;
        }
    }
}

/* ---- CKKS canonical embedding (App. A.12): slot i <-> root zeta^(3^i), zeta = exp(2 pi i / 2N) ---- */
typedef struct {
    double re, im;
} cpx;

static void bitrev_cpx(cpx *v, uint64_t n)
{
    for (uint64_t i = 1, j = 0; i < n; i++) {
        uint64_t bit = n >> 1;
        for (; j >= bit; bit >>= 1) j -= bit;
        j += bit;
        if (i < j) {
            cpx t = v[i];
            v[i] = v[j];
            v[j] = t;
        }
    }
}

/* plain radix-2 DFT: out[r] = sum_k v[k] exp(sign * 2 pi i r k / n) */
static void fft_cpx(cpx *v, uint64_t n, int sign)
{
    const double tw = sign * 6.283185307179586476925286766559 / (double)n;
    cpx *w = (cpx *)malloc(sizeof(cpx) * (n / 2 + 1));
    for (uint64_t i = 0; i < n / 2; i++) {
        w[i].re = cos(tw * (double)i);
        w[i].im = sin(tw * (double)i);
    }
    bitrev_cpx(v, n);
    for (uint64_t len = 2; len <= n; len <<= 1) {
        uint64_t h = len >> 1, step = n / len;
        for (uint64_t i = 0; i < n; i += len)
            for (uint64_t j = 0; j < h; j++) {
                cpx c = w[j * step];
                cpx a = v[i + j], b = v[i + j + h];
                cpx t = {b.re * c.re - b.im * c.im, b.re * c.im + b.im * c.re};
                v[i + j].re = a.re + t.re;
                v[i + j].im = a.im + t.im;
                v[i + j + h].re = a.re - t.re;
                v[i + j + h].im = a.im - t.im;
            }
    }
    free(w);
}

/* encode: p with p(zeta^(3^j)) = v_j*scale and p(conj root) = conj; N-point negacyclic DFT at all odd
 * powers zeta^(2r+1), slot j sits at r = (3^j mod 2N - 1)/2 and its conjugate at (2N - 3^j - 1)/2. */
void orc_encode(const orc_ctx *c, int L, const double *vals_ri, int nvals, double scale, uint64_t *pt)
{
    const uint64_t n = c->N, slots = n >> 1, M = 2 * n;
    cpx *v = (cpx *)calloc(n, sizeof(cpx));
    uint64_t pos = 1;
    for (uint64_t i = 0; i < slots; i++) {
        double re = 0, im = 0;
        if (i < (uint64_t)nvals) {
            re = vals_ri[2 * i];
            im = vals_ri[2 * i + 1];
        }
        uint64_t r1 = (pos - 1) >> 1, r2 = (M - pos - 1) >> 1;
        v[r1].re = re;
        v[r1].im = im;
        v[r2].re = re;
        v[r2].im = -im;
        pos = (pos * 3) & (M - 1);
    }
    fft_cpx(v, n, -1);
    const double tw = 6.283185307179586476925286766559 / (double)M;
    for (uint64_t i = 0; i < n; i++) {
        /* a_k = v_k / n ; p_k = Re(a_k * zeta^-k) */
        double cr = cos(tw * (double)i), ci = -sin(tw * (double)i);
        double co = (v[i].re * cr - v[i].im * ci) / (double)n * scale;
        double r = round(co);
        int neg = r < 0;
        u128 mag = (u128)fabs(r);
        for (int j = 0; j < L; j++) {
            uint64_t q = c->t[j].m.q;
            uint64_t red = (uint64_t)(mag % q);
            pt[(size_t)j * n + i] = neg ? (red ? q - red : 0) : red;
        }
    }
    for (int j = 0; j < L; j++) orc_ntt_fwd(c, j, pt + (size_t)j * n);
    free(v);
}

#define BIGL 20
static void big_mul_add(uint64_t *x, int nl, uint64_t m, uint64_t a)
{ /* x = x*m + a */
    u128 carry = a;
    for (int i = 0; i < nl; i++) {
        u128 t = (u128)x[i] * m + carry;
        x[i] = (uint64_t)t;
        carry = t >> 64;
    }
}
static int big_cmp(const uint64_t *a, const uint64_t *b, int nl)
{
    for (int i = nl - 1; i >= 0; i--)
        if (a[i] != b[i]) return a[i] > b[i] ? 1 : -1;
    return 0;
}
static void big_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, int nl)
{
    uint64_t borrow = 0;
    for (int i = 0; i < nl; i++) {
        uint64_t t = a[i] - b[i], b2 = a[i] < b[i];
        uint64_t t2 = t - borrow;
        b2 |= t < borrow;
        r[i] = t2;
        borrow = b2;
    }
}
static double big_to_double(const uint64_t *a, int nl)
{
    double r = 0;
    for (int i = nl - 1; i >= 0; i--) r = r * 18446744073709551616.0 + (double)a[i];
    return r;
}

void orc_decode(const orc_ctx *c, int L, const uint64_t *pt, double scale, double *vals_ri)
{
    const uint64_t n = c->N, slots = n >> 1;
    assert(L < BIGL);
    uint64_t *co = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)L * n);
    memcpy(co, pt, sizeof(uint64_t) * (size_t)L * n);
    for (int j = 0; j < L; j++) orc_ntt_inv(c, j, co + (size_t)j * n);
    /* Garner inverses inv[j][i] = q_j^-1 mod q_i (j<i) */
    uint64_t inv[BIGL][BIGL];
    for (int i = 0; i < L; i++)
        for (int j = 0; j < i; j++) inv[j][i] = orc_invmod(c->t[j].m.q % c->t[i].m.q, c->t[i].m.q);
    uint64_t Q[BIGL] = {0}, halfQ[BIGL];
    Q[0] = 1;
    for (int j = 0; j < L; j++) big_mul_add(Q, BIGL, c->t[j].m.q, 0);
    for (int i = 0; i < BIGL; i++) halfQ[i] = (Q[i] >> 1) | (i + 1 < BIGL ? Q[i + 1] << 63 : 0);
    double *real = (double *)malloc(sizeof(double) * n);
    for (uint64_t a = 0; a < n; a++) {
        uint64_t vdig[BIGL];
        for (int i = 0; i < L; i++) {
            const mod_t *m = &c->t[i].m;
            uint64_t t = co[(size_t)i * n + a];
            for (int j = 0; j < i; j++) t = mulmod_m(submod(t, vdig[j] % m->q, m->q), inv[j][i], m);
            vdig[i] = t;
        }
        uint64_t X[BIGL] = {0};
        X[0] = vdig[L - 1];
        for (int i = L - 2; i >= 0; i--) big_mul_add(X, BIGL, c->t[i].m.q, vdig[i]);
        if (big_cmp(X, halfQ, BIGL) > 0) {
            uint64_t Y[BIGL];
            big_sub(Y, Q, X, BIGL);
            real[a] = -big_to_double(Y, BIGL);
        } else
            real[a] = big_to_double(X, BIGL);
    }
    const uint64_t M = 2 * n;
    const double tw = 6.283185307179586476925286766559 / (double)M;
    cpx *v = (cpx *)malloc(sizeof(cpx) * n);
    for (uint64_t i = 0; i < n; i++) {
        double x = real[i] / scale;
        v[i].re = x * cos(tw * (double)i);
        v[i].im = x * sin(tw * (double)i);
    }
    fft_cpx(v, n, +1);
    uint64_t pos = 1;
    for (uint64_t i = 0; i < slots; i++) {
        uint64_t r1 = (pos - 1) >> 1;
        vals_ri[2 * i] = v[r1].re;
        vals_ri[2 * i + 1] = v[r1].im;
        pos = (pos * 3) & (M - 1);
    }
    free(co);
    free(real);
    free(v);
}

/* ------------------------------------------------------------------------------------------------
 * Counter-mode sampling: the CPU statement of csrc/hefx_sample.hip.
 * SEAL 3.4.5 draws from a Blake2-based stream that cannot be reproduced by a parallel sampler, and SEAL itself
 * seeds it from std::random_device -- randomness is an INPUT of the parity chain, never an output.  What is
 * specified here instead: every random word is a pure function of (256-bit key, stream id, position), taken from
 * the ChaCha20 keystream (RFC 7539 block function, 64-bit block counter in state words 12-13, 64-bit stream id in
 * words 14-15), so a GPU thread and this loop produce the same bits in any order.
 *   position  -> block counter = attempt << 48 | row << 16 | (index >> 3), 64-bit word (index & 7) of that block
 *   uniform   mod q: accept r < q * floor(2^64 / q), value r mod q; else attempt + 1            (SEAL: rejection)
 *   ternary   {-1,0,1}: r mod 3 - 1 with the same rejection rule                                 (App. A.11)
 *   noise     trunc(N(0, 3.2^2) | |x| <= 19.2) by inverse CDF on r with a 39-entry threshold table built from
 *             erfc in double precision -- exact integer comparisons, no libm call in the sampler itself
 * ---------------------------------------------------------------------------------------------- */
#define ROTL32(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define CHACHA_QR(a, b, c, d) \
    a += b; d ^= a; d = ROTL32(d, 16); c += d; b ^= c; b = ROTL32(b, 12); \
    a += b; d ^= a; d = ROTL32(d, 8);  c += d; b ^= c; b = ROTL32(b, 7);

void orc_chacha20_block(const uint32_t key[8], uint64_t counter, uint64_t nonce, uint32_t out[16])
{
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4], key[5], key[6], key[7], (uint32_t)counter, (uint32_t)(counter >> 32),
                      (uint32_t)nonce, (uint32_t)(nonce >> 32)};
    uint32_t x[16];
    memcpy(x, s, sizeof x);
    for (int r = 0; r < 10; r++) {
        CHACHA_QR(x[0], x[4], x[8], x[12]) CHACHA_QR(x[1], x[5], x[9], x[13])
        CHACHA_QR(x[2], x[6], x[10], x[14]) CHACHA_QR(x[3], x[7], x[11], x[15])
        CHACHA_QR(x[0], x[5], x[10], x[15]) CHACHA_QR(x[1], x[6], x[11], x[12])
        CHACHA_QR(x[2], x[7], x[8], x[13]) CHACHA_QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

static uint64_t sample_word(const uint32_t key[8], uint64_t stream, uint64_t row, uint64_t idx, uint64_t attempt)
{
    uint32_t b[16];
    orc_chacha20_block(key, (attempt << 48) | (row << 16) | (idx >> 3), stream, b);
    const int w = (int)(idx & 7);
    return (uint64_t)b[2 * w] | ((uint64_t)b[2 * w + 1] << 32);
}

/* thresholds t[0..38]: value = -19 + #{i : r >= t[i]}, t[i] = floor(2^64 * P(Y <= -19 + i)) */
void orc_noise_thresholds(uint64_t t[39])
{
    const double sigma = 3.2, clip = 19.2, rs2 = 1.0 / (sigma * 1.4142135623730951);
    const double tail = erfc(clip * rs2);           /* 2 * P(X > clip) */
    const double total = 1.0 - tail;                /* P(|X| <= clip) */
    /* P(Y <= k) for k = -19..19 with Y = trunc(X): for k < 0, Y <= k <=> X <= k (strictly below k+1 ... see below) */
    for (int i = 0; i < 39; i++) {
        const int k = -19 + i;
        /* Y <= k  <=>  X < k + 1 for k >= 0;  X <= k for k < 0 (trunc toward zero); X continuous so < vs <= agree */
        const double edge = k >= 0 ? (double)(k + 1) : (double)k;
        double below; /* P(-clip <= X < edge) */
        if (edge >= clip)
            below = total;
        else
            below = 0.5 * erfc(-edge * rs2) - 0.5 * tail;
        double cdf = below / total;
        if (cdf >= 1.0 || i == 38) {
            t[i] = ~(uint64_t)0;
            continue;
        }
        long double scaled = (long double)cdf * 18446744073709551616.0L;
        t[i] = (uint64_t)scaled;
    }
}

void orc_sample_uniform(const orc_ctx *c, const uint32_t key[8], uint64_t stream, int npoly, int nrows,
                        int mod_first, uint64_t *out)
{
    const uint64_t n = c->N;
    for (int p = 0; p < npoly; p++)
        for (int j = 0; j < nrows; j++) {
            const uint64_t q = c->t[mod_first + j].m.q;
            const uint64_t bound = (~(uint64_t)0 / q) * q; /* q * floor((2^64-1)/q) == q * floor(2^64/q), q odd */
            const uint64_t row = (uint64_t)p * nrows + j;
            uint64_t *o = out + row * n;
            for (uint64_t i = 0; i < n; i++) {
                uint64_t r, a = 0;
                do r = sample_word(key, stream, row, i, a++);
                while (r >= bound);
                o[i] = r % q;
            }
        }
}

void orc_sample_ternary(const orc_ctx *c, const uint32_t key[8], uint64_t stream, int npoly, int nrows,
                        int mod_first, uint64_t *out)
{
    const uint64_t n = c->N;
    for (int p = 0; p < npoly; p++)
        for (uint64_t i = 0; i < n; i++) {
            uint64_t r, a = 0;
            do r = sample_word(key, stream, (uint64_t)p, i, a++);
            while (r >= 0xFFFFFFFFFFFFFFFFull); /* 3 * floor(2^64 / 3) */
            const int v = (int)(r % 3) - 1;
            for (int j = 0; j < nrows; j++) {
                const uint64_t q = c->t[mod_first + j].m.q;
                out[((size_t)p * nrows + j) * n + i] = v < 0 ? q - 1 : (uint64_t)v;
            }
        }
}

void orc_sample_noise(const orc_ctx *c, const uint32_t key[8], uint64_t stream, int npoly, int nrows,
                      int mod_first, uint64_t *out)
{
    const uint64_t n = c->N;
    uint64_t t[39];
    orc_noise_thresholds(t);
    for (int p = 0; p < npoly; p++)
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t r = sample_word(key, stream, (uint64_t)p, i, 0);
            int v = -19;
            for (int e = 0; e < 38; e++) v += r >= t[e];
            for (int j = 0; j < nrows; j++) {
                const uint64_t q = c->t[mod_first + j].m.q;
                out[((size_t)p * nrows + j) * n + i] = v < 0 ? q - (uint64_t)(-v) : (uint64_t)v;
            }
        }
}

/*
 * ckks_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the Microsoft SEAL 3.4.5 CKKS Evaluator algorithms that back the
 * reference's hot path (Linear_Transform_Plain /root/reference/helper.h:237-262 and friends).
 *
 * PARITY UNPINNED: the arithmetic of the path lives in Microsoft SEAL 3.4.5 (pinned only in prose at
 * /root/reference/README.md:6, consumed via find_package(SEAL) /root/reference/CMakeLists.txt:25).
 * SEAL is not vendored in /root/reference, is not installed in this image and cannot be fetched
 * (no network); the reference holds no golden vectors / KATs for the path.
 * This file therefore restates SEAL 3.4.5's published algorithms and is pinned
 * only against (a) mathematical definitions (O(N^2) negacyclic evaluation, slot-rotation semantics,
 * decrypt(op(enc x)) == op(x)), and (b) the decrypted known answers the reference prints
 * (matrix_multiplication.cpp:171-196 A*A for 1..n^2; imgs/lin_transf.jpg).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
 *
 * Layout (SEAL): ciphertext = size polys; poly = L RNS rows; row = N uint64, canonical
 * residues in [0,q_j): data[(p*L + j)*N + i]. CKKS data are always in NTT form.
 */
#ifndef CKKS_ORACLE_H
#define CKKS_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* ---- number theory (SEAL util/numth) ---- */
int orc_is_prime(uint64_t n);
/* CoeffModulus::Create(N, bit_sizes): returns 0 on success; out has nbits entries. */
int orc_coeff_modulus_create(uint64_t N, const int *bit_sizes, int nbits, uint64_t *out);
/* minimal primitive 2N-th root of unity mod q (try_minimal_primitive_root); 0 if none. */
uint64_t orc_min_primitive_root(uint64_t two_n, uint64_t q);
uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t q);
uint64_t orc_powmod(uint64_t a, uint64_t e, uint64_t q);
uint64_t orc_invmod(uint64_t a, uint64_t q);

/* ---- context: k primes, last one is the special prime P (key level); data level L uses q_0..q_{L-1} ---- */
orc_ctx *orc_ctx_create(uint64_t N, const uint64_t *primes, int k);
void orc_ctx_destroy(orc_ctx *c);
uint64_t orc_ctx_n(const orc_ctx *c);
int orc_ctx_k(const orc_ctx *c);
uint64_t orc_ctx_prime(const orc_ctx *c, int j);
uint64_t orc_ctx_psi(const orc_ctx *c, int j);

/* ---- NTT (SEAL util/smallntt, App. A.5): natural in -> bit-reversed out, out[i]=a(psi^(2*bitrev(i)+1)) ---- */
void orc_ntt_fwd(const orc_ctx *c, int j, uint64_t *a);
void orc_ntt_inv(const orc_ctx *c, int j, uint64_t *a);
/* O(N^2) definition, for pinning the fast transform at small N. */
void orc_ntt_naive(const orc_ctx *c, int j, const uint64_t *in, uint64_t *out);

/* ---- Galois (App. A.7) ---- */
uint64_t orc_galois_elt_from_step(uint64_t N, int step);
/* NAF decomposition of a rotation step as SEAL's rotate_internal applies it; returns count. */
int orc_naf_steps(uint64_t N, int step, int *out, int max_out);
void orc_galois_table(uint64_t N, uint64_t elt, uint32_t *table);
void orc_apply_galois_ntt(const orc_ctx *c, uint64_t elt, const uint64_t *in, uint64_t *out);

/* ---- Evaluator element-wise ops on [size][L][N] payloads (App. A.6) ---- */
void orc_add(const orc_ctx *c, int L, int size, const uint64_t *a, const uint64_t *b, uint64_t *out);
void orc_sub(const orc_ctx *c, int L, int size, const uint64_t *a, const uint64_t *b, uint64_t *out);
void orc_negate(const orc_ctx *c, int L, int size, const uint64_t *a, uint64_t *out);
void orc_add_plain(const orc_ctx *c, int L, int size, const uint64_t *ct, const uint64_t *pt, uint64_t *out);
void orc_multiply_plain(const orc_ctx *c, int L, int size, const uint64_t *ct, const uint64_t *pt, uint64_t *out);
/* size_a x size_b -> size_a+size_b-1 tensor product (dyadic). */
void orc_multiply(const orc_ctx *c, int L, int size_a, const uint64_t *a, int size_b, const uint64_t *b,
                  uint64_t *out);
/* 1 if every poly beyond c0 is all zero (SEAL is_transparent). */
int orc_is_transparent(const orc_ctx *c, int L, int size, const uint64_t *ct);

/* ---- key switching (App. A.8). key: [L_key][2][k][N] with L_key = k-1. ct (size 2, L rows) updated ---- */
void orc_switch_key(const orc_ctx *c, int L, uint64_t *ct, const uint64_t *target, const uint64_t *key);
/* apply_galois_inplace (CKKS, size 2): perm + switch_key. */
void orc_apply_galois(const orc_ctx *c, int L, const uint64_t *ct_in, uint64_t elt, const uint64_t *key,
                      uint64_t *ct_out);
/* relinearize size 3 -> 2 with relin key (index 0). */
void orc_relinearize(const orc_ctx *c, int L, const uint64_t *ct3, const uint64_t *key, uint64_t *ct2);
/* rescale_to_next (App. A.9): L rows -> L-1 rows; rounded=0 is SEAL 3.4.x floor variant. */
void orc_rescale(const orc_ctx *c, int L, int size, const uint64_t *in, uint64_t *out, int rounded);
/* mod_switch_to_next (CKKS): drop last row(s): L_in -> L_out rows, npoly polys. */
void orc_mod_drop(const orc_ctx *c, int L_in, int L_out, int npoly, const uint64_t *in, uint64_t *out);
/* bench unit: rotate by a directly keyed step, then multiply_plain. */
void orc_rotate_mulplain(const orc_ctx *c, int L, const uint64_t *ct_in, uint64_t elt, const uint64_t *key,
                         const uint64_t *pt, uint64_t *ct_out);

/* ---- non-hot L2 pieces needed for decrypted-value checks (App. A.11/A.12); seeded splitmix64 ---- */
void orc_fill_uniform(const orc_ctx *c, int L, int npoly, uint64_t seed, uint64_t *out);
void orc_gen_secret(const orc_ctx *c, uint64_t seed, uint64_t *sk /* [k][N] NTT */);
/* key-switch key for new secret s' ([k][N], NTT) under sk: out [k-1][2][k][N]. */
void orc_gen_kswitch_key(const orc_ctx *c, const uint64_t *sk, const uint64_t *new_sk, uint64_t seed,
                         uint64_t *out);
void orc_gen_relin_key(const orc_ctx *c, const uint64_t *sk, uint64_t seed, uint64_t *out);
void orc_gen_galois_key(const orc_ctx *c, const uint64_t *sk, uint64_t elt, uint64_t seed, uint64_t *out);
/* symmetric encryption of an NTT plaintext at data level L: ct [2][L][N]. */
void orc_encrypt_sym(const orc_ctx *c, int L, const uint64_t *sk, const uint64_t *pt, uint64_t seed,
                     uint64_t *ct);
/* decrypt any size: pt [L][N] NTT form = sum c_i s^i. */
void orc_decrypt(const orc_ctx *c, int L, int size, const uint64_t *ct, const uint64_t *sk, uint64_t *pt);
/* CKKS encode: nvals <= N/2 complex values (re,im interleaved) -> pt [L][N] NTT. */
void orc_encode(const orc_ctx *c, int L, const double *vals_ri, int nvals, double scale, uint64_t *pt);
/* CKKS decode: pt [L][N] NTT -> N/2 complex values (re,im interleaved). */
void orc_decode(const orc_ctx *c, int L, const uint64_t *pt, double scale, double *vals_ri);


/* the UNCORRECTED hoisted rotation (what rounds 1-3 shipped as a fast mode; other words than SEAL's, see ckks_oracle.c):
 * kept as the counter-example of tests/test_oracle_pinning.py -- the engine runs orc_apply_galois_hoisted_exact's identity */
void orc_apply_galois_hoisted(const orc_ctx *c, int L, const uint64_t *ct_in, uint64_t elt, const uint64_t *key,
                              uint64_t *ct_out);

/* exactly hoisted rotation (ks_mac_exact_kernel's identity): orc_apply_galois's words; returns 1 if a zero coefficient of
 * c1 made it take the regular sequence, 0 if the hoisted identity applied */
int orc_apply_galois_hoisted_exact(const orc_ctx *c, int L, const uint64_t *ct_in, uint64_t elt, const uint64_t *key,
                                   uint64_t *ct_out);
/* double-hoisted linear transform, core on ct_new (second fast mode; see ckks_oracle.c); elts[l], l = 1..d-1;
 * keys = [d-1][k-1][2][k][N]; diag = [d][k][N] key-level plaintexts; L must be k-1 */
void orc_lt_double_hoisted_core(const orc_ctx *c, int L, const uint64_t *ct_new, int d, const uint64_t *diag,
                                const uint64_t *elts, const uint64_t *keys, uint64_t *out);

/* ---- counter-mode sampling: CPU statement of csrc/hefx_sample.hip (see ckks_oracle.c for the specification) */
void orc_chacha20_block(const uint32_t key[8], uint64_t counter, uint64_t nonce, uint32_t out[16]);
void orc_noise_thresholds(uint64_t t[39]);
void orc_sample_uniform(const orc_ctx *c, const uint32_t key[8], uint64_t stream, int npoly, int nrows,
                        int mod_first, uint64_t *out);
void orc_sample_ternary(const orc_ctx *c, const uint32_t key[8], uint64_t stream, int npoly, int nrows,
                        int mod_first, uint64_t *out);
void orc_sample_noise(const orc_ctx *c, const uint32_t key[8], uint64_t stream, int npoly, int nrows,
                      int mod_first, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif

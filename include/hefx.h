/*
 * hefx.h -- C-ABI of the MI355X (gfx950) CKKS ciphertext-arithmetic engine.
 *
 * Drop-in boundary for the hot path of MarwanNour/SEAL-FYP-Logistic-Regression.  The reference has no
 * FFI layer of its own: its boundary is the C++ class API of Microsoft SEAL 3.4.5's `seal::Evaluator`
 * (`#include "seal/seal.h"`, reference file helper.h:4 (all file:line citations in this header are relative to the reference repo root); CMake target SEAL::seal,
 * CMakeLists.txt:25-41).  Every entry point below names the Evaluator member it
 * replaces and the reference call sites that reach it; include/seal/seal.h is the C++ shim that maps
 * the class API onto these calls.
 *
 * Conventions
 *  - Payload layout is SEAL's: a ciphertext of `size` polys at a level with L
 *    data primes is size*L*N uint64 words, word (p*L + j)*N + i = coefficient i of poly p mod q_j,
 *    canonical in [0,q_j), ALWAYS in NTT form.  A plaintext is L*N words.  A key-switching key is
 *    (k-1)*2*k*N words: [digit i][component c][key-level row m][N]  (k = number of primes incl. the
 *    special prime P = primes[k-1]).
 *  - All `d_*` pointers are DEVICE pointers (hefx_malloc or any hipMalloc'd memory of the same
 *    device).  Pointer ARRAYS (`const uint64_t *const *`) are HOST arrays of device pointers.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous on
 *    that stream; only hefx_download / hefx_stream_sync / hefx_check_* block.
 *  - Return value: 0 = HEFX_OK, negative = error; hefx_last_error() gives the thread-local message.
 *  - Level/scale/parms_id bookkeeping, NAF decomposition of rotation steps and SEAL's validity checks
 *    live ABOVE this ABI (in the shim); this layer is pure uint64 RNS arithmetic.
 *  - There is no CPU fallback: without a HIP device every call fails with HEFX_ERR_HIP.
 *  - Threading: a context owns its scratch, descriptor ring and internal streams, so calls on ONE context must not
 *    overlap in time on the host (the reference's drivers are single-threaded); different contexts -- e.g. one per
 *    GPU / per rank -- are independent.  Work submitted through one context is ordered on the caller's stream.
 */
#ifndef HEFX_H
#define HEFX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HEFX_OK 0
#define HEFX_ERR_INVALID (-1)     /* bad argument (maps to std::invalid_argument in the shim) */
#define HEFX_ERR_HIP (-2)         /* HIP runtime / device failure */
#define HEFX_ERR_UNSUPPORTED (-3) /* parameter set outside what the kernels are built for */
#define HEFX_ERR_TRANSPARENT (-4) /* result ciphertext is transparent (std::logic_error in the shim) */

typedef struct hefx_context hefx_context;

const char *hefx_last_error(void);
/* "gfx950" build id + version string */
const char *hefx_version(void);
/* number of visible HIP devices (0 if none / no runtime) */
int hefx_device_count(void);

/* ---- context: replaces SEALContext::Create(parms) + the Evaluator's NTT tables
 *      (linear_transformation2.cpp:229-237, helper.h:239-240).
 *      N = poly_modulus_degree in {1024..16384, 32768}; primes = coeff_modulus (last = special prime). */
int hefx_context_create(uint32_t poly_degree, const uint64_t *primes, int k, int device, hefx_context **out);
void hefx_context_destroy(hefx_context *ctx);
uint32_t hefx_poly_degree(const hefx_context *ctx);
int hefx_prime_count(const hefx_context *ctx);
uint64_t hefx_prime(const hefx_context *ctx, int j);
/* minimal primitive 2N-th root used for prime j (SEAL try_minimal_primitive_root) */
uint64_t hefx_psi(const hefx_context *ctx, int j);

/* ---- device memory + transfers (Ciphertext/Plaintext/key payload ownership stays with the caller).
 *      hefx_malloc / hefx_free are POOLED per context: a freed block is parked (up to HEFX_POOL_MB megabytes,
 *      default 65536; 0 = plain hipMalloc / hipFree) and handed out again by a later hefx_malloc of the same size,
 *      with no device synchronisation -- SEAL's MemoryPoolHandle in spirit (SURVEY 8b "Ownership").  A recycled block
 *      may still be in use by work submitted before the free; that is correct as long as this earlier work and the
 *      new owner's work are ordered on the device: one stream (what the shim and seal.py do), or streams the caller
 *      has ordered with events before calling hefx_free.  Blocks not obtained from hefx_malloc are passed to hipFree. */
int hefx_malloc(hefx_context *ctx, size_t bytes, void **d_ptr);
int hefx_free(hefx_context *ctx, void *d_ptr);
int hefx_upload(hefx_context *ctx, void *d_dst, const void *h_src, size_t bytes, void *stream);
int hefx_download(hefx_context *ctx, void *h_dst, const void *d_src, size_t bytes, void *stream); /* blocks */
int hefx_copy(hefx_context *ctx, void *d_dst, const void *d_src, size_t bytes, void *stream);
/* between two contexts, possibly on different GPUs (hipMemcpyPeerAsync; a plain device copy when both contexts share a
 * device): asynchronous on `stream` of the SOURCE context's device (null: its default stream), i.e. ordered after the
 * source context's work that produced the bytes.  What lets one host process spread independent sub-graphs of the
 * reference's loops (logistic_regression_ckks.cpp:217-229, matrix_mult_benchmark.cpp:41-43) over the GPUs of a node
 * and bring the results together (include/seal/seal.h, SEAL_SHIM_DEVICES). */
int hefx_copy_peer(hefx_context *dst_ctx, void *d_dst, hefx_context *src_ctx, const void *d_src, size_t bytes,
                   void *stream);
/* the same copy submitted on `stream` of the DESTINATION context's device (null: its default stream): ordered after the
 * destination context's earlier work -- in particular after whatever still uses a recycled destination block
 * (hefx_malloc's contract above) -- and NOT after the source's: the caller makes the source bytes complete first
 * (hefx_stream_sync on the source context, or an event) and keeps the source block alive until the copy has run.
 * This is the direction results travel home in (include/seal/seal.h flush_multi). */
int hefx_copy_peer_to(hefx_context *dst_ctx, void *d_dst, hefx_context *src_ctx, const void *d_src, size_t bytes,
                      void *dst_stream);
/* the HIP device a context lives on */
int hefx_context_device(const hefx_context *ctx);
/* free and total memory of that device in bytes (hipMemGetInfo; blocks parked in the context's pool count as used):
 * what a host side sizes its budgets with (the shim's pending-results budget is a quarter of the device) */
int hefx_device_memory(hefx_context *ctx, size_t *free_bytes, size_t *total_bytes);
/* How many key-switch chunks of this context were REDONE item by item so far: batches that rotate few distinct
 * ciphertexts many times (the d-1 rotations of Linear_Transform_Plain, helper.h:252-257) run exactly
 * hoisted -- the source is decomposed once per chunk -- which covers every input except a source whose c1 has a zero
 * coefficient in some RNS component (a transparent or hand-made ciphertext); such a chunk falls back on the device to
 * the per-item sequence, same bits either way.  Waits for the device.  A diagnostic: tests assert 0 on random inputs
 * (the fast path ran) and > 0 on planted zeros (the fallback ran). */
int hefx_ks_fallback_count(hefx_context *ctx, uint64_t *chunks);
/* Host-side counters of the key-switch front door since the context was created (no device wait): out[0] key switches
 * submitted (items of every batch, relinearisations included), out[1] of them run exactly hoisted (sharing their source's
 * decomposition), out[2] launch sequences (chunks), out[3] batched calls.  What a caller needs to price a composite --
 * e.g. how many key switches the NAF forest of one Linear_Transform_Plain with the reference's power-of-two keys
 * (linear_transformation2.cpp:239) executes after prefix sharing. */
int hefx_ks_stats(hefx_context *ctx, uint64_t out[4]);
int hefx_memset_zero(hefx_context *ctx, void *d_dst, size_t bytes, void *stream);
int hefx_stream_sync(hefx_context *ctx, void *stream);

/* ---- K1/K2: negacyclic NTT over RNS rows (SEAL util::ntt_negacyclic_harvey / inverse_...; reached from
 *      every rotate/relinearize/rescale and from encode/encrypt/decrypt).  In-place over `npoly` polys of
 *      `nrows` rows each; row r uses prime index mod_first + r. */
int hefx_ntt_forward(hefx_context *ctx, uint64_t *d_data, int npoly, int nrows, int mod_first, void *stream);
int hefx_ntt_inverse(hefx_context *ctx, uint64_t *d_data, int npoly, int nrows, int mod_first, void *stream);

/* ---- K4/K10: Evaluator::add / sub / negate / add_plain (helper.h:219,231,247,259,275,319,464,475;
 *      logistic_regression_ckks.cpp:288,341-342; polynomial.cpp:210).  `count` contiguous ciphertexts. */
int hefx_add(hefx_context *ctx, int L, int size, int count, const uint64_t *d_a, const uint64_t *d_b,
             uint64_t *d_out, void *stream);
int hefx_sub(hefx_context *ctx, int L, int size, int count, const uint64_t *d_a, const uint64_t *d_b,
             uint64_t *d_out, void *stream);
int hefx_negate(hefx_context *ctx, int L, int size, int count, const uint64_t *d_a, uint64_t *d_out,
                void *stream);
int hefx_add_plain(hefx_context *ctx, int L, int size, const uint64_t *d_ct, const uint64_t *d_pt,
                   uint64_t *d_out, void *stream);
/* n independent pairs in one launch: d_out[i] = d_a[i] +/- d_b[i] (the add / add_inplace of n dot-product chains in
 * lockstep, helper.h:464,475).  Host arrays of device pointers; d_out[i] may alias d_a[i] or d_b[i]. */
int hefx_add_batch(hefx_context *ctx, int L, int size, int n, const uint64_t *const *d_a, const uint64_t *const *d_b,
                   uint64_t *const *d_out, void *stream);
int hefx_sub_batch(hefx_context *ctx, int L, int size, int n, const uint64_t *const *d_a, const uint64_t *const *d_b,
                   uint64_t *const *d_out, void *stream);
/* Evaluator::add_many (helper.h:231,259,275,319): out = sum of n ciphertexts (one n-way reduction). */
int hefx_add_many(hefx_context *ctx, int L, int size, int n, const uint64_t *const *d_in, uint64_t *d_out,
                  void *stream);

/* ---- K3/K11: Evaluator::multiply_plain (helper.h:250,256,271,347), multiply (helper.h:222,228,432;
 *      matrix_multiplication.cpp:104,127), square (vector_ops.cpp:269; 4_ckks.cpp:114).
 *      multiply_plain records "transparent" (all polys beyond c0 zero) in a device flag; read it with
 *      hefx_check_transparent (blocks), which returns HEFX_ERR_TRANSPARENT if any call since the last
 *      check produced a transparent result. */
int hefx_multiply_plain(hefx_context *ctx, int L, int size, int count, const uint64_t *d_ct,
                        const uint64_t *d_pt, uint64_t *d_out, void *stream);
int hefx_check_transparent(hefx_context *ctx, void *stream);
/* n independent products in one launch: d_outs[i] = d_cts[i] (.) d_pts[i] (the mask products of
 * logistic_regression_ckks.cpp:229 over all rows).  Host arrays of device pointers; no output may alias its input;
 * transparency is the caller's check (the plaintexts' zero flags), as for hefx_multiply_plain_sum. */
int hefx_multiply_plain_batch(hefx_context *ctx, int L, int size, int n, const uint64_t *const *d_cts,
                              const uint64_t *const *d_pts, uint64_t *const *d_outs, void *stream);
/* Linear_Transform_CipherMatrix_PlainVector (helper.h:265-278: add_many of multiply_plain results, :271,:275) and the
 * inner sums of a baby-step/giant-step transform, in one pass: for g in [0, ceil(n/group)):
 *   d_outs[g] = sum_{i in [g*group, min(n,(g+1)*group))} d_cts[i] (.) d_pts[i]   (mod q_j per row)
 * d_cts[i]: [size][L][N], d_pts[i]: [L][N], d_outs[g]: [size][L][N], none of a group's inputs aliasing its output.
 * The canonical residues of the sums, i.e. the bits of n multiply_plain calls followed by add_many.  The pointer
 * arrays are host arrays of device pointers.  Transparency is the caller's check (the plaintexts' zero flags). */
int hefx_multiply_plain_sum(hefx_context *ctx, int L, int size, int n, int group, const uint64_t *const *d_cts,
                            const uint64_t *const *d_pts, uint64_t *const *d_outs, void *stream);

/* size 2 x size 2 -> size 3 */
int hefx_multiply(hefx_context *ctx, int L, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out3,
                  void *stream);
int hefx_square(hefx_context *ctx, int L, const uint64_t *d_a, uint64_t *d_out3, void *stream);
/* n independent size 2 x size 2 products in one launch: d_out3[i] = d_a[i] * d_b[i] (the loop of cipher_dot_product
 * over the rows of a data set, logistic_regression_ckks.cpp:217-220 -> helper.h:432).  Host arrays of device
 * pointers; d_b may repeat one ciphertext; outputs must not alias inputs. */
int hefx_multiply_batch(hefx_context *ctx, int L, int n, const uint64_t *const *d_a, const uint64_t *const *d_b,
                        uint64_t *const *d_out3, void *stream);

/* ---- K5/K6/K7: Evaluator::apply_galois_inplace = Galois permutation + key switch (one term of
 *      rotate_vector; helper.h:216,227,244,255,316,352,455,474; 5_rotation.cpp:215).
 *      d_key is the Galois key of `galois_elt` (the shim picks it: GaloisKeys index (elt-1)/2).
 *      in and out may alias (rotate_vector_inplace). */
int hefx_apply_galois(hefx_context *ctx, int L, const uint64_t *d_ct_in, uint32_t galois_elt,
                      const uint64_t *d_key, uint64_t *d_ct_out, void *stream);
/* n independent (ciphertext, element, key) triples in one launch sequence.  "Independent" is meant: no item may read
 * another item's output (an item's own input and output may be the same buffer); the engine processes the items in an
 * order of its choosing (grouped by key, so that neighbours share their key loads).  A batch that breaks the rule is
 * refused with HEFX_ERR_INVALID before anything is submitted; the check is on byte ranges, so views into one allocation
 * are covered: two outputs that overlap, an input (or, in hefx_rotate_multiply_plain_batch, a plaintext) that overlaps
 * another item's output, an input that overlaps its own output other than exactly (d_ct_in[i] == d_ct_out[i], the
 * in-place rotation).  The *_hoisted entry points are stricter: their one shared source may be no item's output.
 * Batches of more than 32 items that rotate few distinct ciphertexts run exactly hoisted (see hefx_rotate_hoisted_batch). */
int hefx_apply_galois_batch(hefx_context *ctx, int L, int n, const uint64_t *const *d_ct_in,
                            const uint32_t *galois_elts, const uint64_t *const *d_keys,
                            uint64_t *const *d_ct_out, void *stream);
/* the hot-loop body of Linear_Transform_Plain (helper.h:255-256): out_i = rotate(ct_i, elt_i) (.) pt_i with
 * a directly keyed element, multiply_plain fused into the key-switch epilogue.  This is the unit
 * BASELINE.json's metric counts.  A NULL d_pts[i] makes item i a plain rotation (no product): the rotations of one
 * dependency depth of a NAF forest -- some end in a diagonal product, some only feed deeper rotations -- go out as ONE
 * batch that way (the C++ shim's recorder does). */
int hefx_rotate_multiply_plain_batch(hefx_context *ctx, int L, int n, const uint64_t *const *d_ct_in,
                                     const uint32_t *galois_elts, const uint64_t *const *d_keys,
                                     const uint64_t *const *d_pts, uint64_t *const *d_ct_out, void *stream);

/* ---- rotate_vector_inplace + add_inplace as ONE key switch (helper.h:474-475, the body of cipher_dot_product's loop;
 *      logistic_regression_ckks.cpp:217-229 and :295-300 reach it 2000 x 7 and 8 x 1999 times per training step):
 *      ct_out[i] = apply_galois(ct_in[i], elt_i, key_i) and acc_out[i] = acc_in[i] + ct_out[i] (mod q, canonical -- the
 *      words Evaluator::add_inplace would leave), the sum taken in the mod-down epilogue that produces the rotation's
 *      words instead of a sixth launch and a second pass over the row.  acc_in[i] == acc_out[i] is the in-place sum,
 *      ct_in[i] == ct_out[i] the in-place rotation; otherwise the independence rule of hefx_apply_galois_batch holds
 *      for the rotations AND the sums (no sum may overlap another item's sum, rotation or input). */
int hefx_apply_galois_add_batch(hefx_context *ctx, int L, int n, const uint64_t *const *d_ct_in,
                                const uint32_t *galois_elts, const uint64_t *const *d_keys,
                                const uint64_t *const *d_acc_in, uint64_t *const *d_acc_out,
                                uint64_t *const *d_ct_out, void *stream);
/* the whole loop of helper.h:472-476 for n ciphertext pairs in lockstep (the eight weight chains of the LR gradient):
 *      t_0 = ct_in, t_s = apply_galois(t_(s-1), elt, key), a_s = a_(s-1) + t_s for s = 1..steps (a_0 = acc_in);
 *      ct_out = t_steps, acc_out = a_steps; the inputs are not written.  Level by level the same key switches as `steps`
 *      calls of hefx_apply_galois_add_batch (same bits); the intermediate rotations live in two engine-owned buffer sets,
 *      so the levels are two alternating launch sequences issued from one loop -- no per-level validation, allocation
 *      or host bookkeeping.  What the unvalidated levels write is checked up front, for every n: a null pointer, or any
 *      two of the 2n buffers d_ct_out[i] / d_acc_out[i] overlapping in bytes, is HEFX_ERR_INVALID before anything is
 *      submitted (d_ct_out[i] == d_ct_in[i] is fine: the inputs are read by the first level only).  Up to 32 chains run
 *      on the small-batch path (four launches per level, 38-44 us at N = 16384, L = 2), sixteen or more of them dealt
 *      over the caller's stream and the internal ones, eight per stream (HEFX_CHAIN_LANES=<k> forces k streams).
 *      HEFX_CHAIN_GRAPH=1 replays the levels as a captured two-level HIP graph instead (measured slower on MI355X /
 *      ROCm 7.2: 60 against 54 us per level at n = 8, L = 2, round 4). */
int hefx_rotate_add_chain(hefx_context *ctx, int L, int n, const uint64_t *const *d_ct_in, const uint32_t *galois_elts,
                          const uint64_t *const *d_keys, const uint64_t *const *d_acc_in, uint64_t *const *d_acc_out,
                          uint64_t *const *d_ct_out, int steps, void *stream);

/* ---- Evaluator::relinearize_inplace (helper.h:440,541; polynomial.cpp:92,187): size 3 -> 2. */
/* (input and output must not overlap in bytes -- three polynomials in, two out; checked for n = 1 as for a batch) */
int hefx_relinearize(hefx_context *ctx, int L, const uint64_t *d_ct3, const uint64_t *d_relin_key,
                     uint64_t *d_ct2, void *stream);
int hefx_relinearize_batch(hefx_context *ctx, int L, int n, const uint64_t *const *d_ct3,
                           const uint64_t *d_relin_key, uint64_t *const *d_ct2, void *stream);

/* ---- K8: Evaluator::rescale_to_next_inplace (matrix_multiplication.cpp:71-72; helper.h:441,543; polynomial.cpp:93,
 *      195,333; logistic_regression_ckks.cpp:119,189,239,321): L rows -> L-1 rows per poly, `count` contiguous cts.
 *      The division by the dropped prime q_l exists in two forms and which one SEAL 3.4.5 uses is the one item of
 *      SURVEY App. A.9 that could not be verified offline, so both are built (and both are bit-exact against the
 *      oracle's orc_rescale(rounded = 0 / 1)):
 *        HEFX_RESCALE_FLOOR  out_j = (c_j - [c_l]_(q_j)) * q_l^-1  -- App. A.9's [M]-confidence statement of 3.4.x
 *                            (BaseConverter::floor_last_coeff_modulus_ntt_inplace).
 *        HEFX_RESCALE_ROUND  out_j = (c_j - ([c_l + q_l/2]_(q_l) mod q_j - (q_l/2 mod q_j))) * q_l^-1 -- round to
 *                            nearest: 3.4.x's BaseConverter::round_last_coeff_modulus_ntt_inplace as two independent
 *                            reviews recall Evaluator::mod_switch_scale_to_next calling it, and SEAL >= 3.5's
 *                            RNSTool::divide_and_round_q_last_ntt_inplace.  The default since round 6.
 *      hefx_rescale_to_next uses the context's mode (hefx_set_rescale_mode; environment HEFX_RESCALE=floor|round
 *      presets it at hefx_context_create); hefx_rescale_to_next_mode names it per call. */
#define HEFX_RESCALE_FLOOR 0
#define HEFX_RESCALE_ROUND 1
int hefx_rescale_to_next(hefx_context *ctx, int L, int size, int count, const uint64_t *d_in, uint64_t *d_out,
                         void *stream);
int hefx_rescale_to_next_mode(hefx_context *ctx, int L, int size, int count, const uint64_t *d_in, uint64_t *d_out,
                              int mode, void *stream);
/* n independent ciphertexts through host arrays of device pointers (the rescales of n dot products advancing in
 * lockstep, helper.h:441 over logistic_regression_ckks.cpp:217); uses the context's mode; d_out[i] != d_in[i]. */
int hefx_rescale_to_next_batch(hefx_context *ctx, int L, int size, int n, const uint64_t *const *d_in,
                               uint64_t *const *d_out, void *stream);
int hefx_set_rescale_mode(hefx_context *ctx, int mode);
int hefx_get_rescale_mode(const hefx_context *ctx);
/* ---- K9: Evaluator::mod_switch_to_next / mod_switch_to for CKKS ct and pt (matrix_multiplication.cpp:112):
 *      drop trailing RNS rows, L_in -> L_out, npoly polys. */
int hefx_mod_drop(hefx_context *ctx, int L_in, int L_out, int npoly, const uint64_t *d_in, uint64_t *d_out,
                  void *stream);

/* ---- multi-GPU exchange (no reference call site; SURVEY.md 8b "hefx_allreduce_sum(ct, comm)", 8e).  The path shards
 *      by independent units (diagonals of a transform, Step-2 transforms of a matrix product, LR rows) with every rank
 *      holding the inputs and keys; the ONE exchange is the final ciphertext sum.
 *      One process per GPU, one context per process, one communicator per context: rank 0 obtains 128 opaque bytes with
 *      hefx_comm_unique_id and hands them to the other ranks out of band (file, environment, MPI, a torch store);
 *      every rank then calls hefx_comm_init(world <= 8).  RCCL is loaded at run time (dlopen): none of this touches it
 *      until hefx_comm_*; HEFX_ERR_UNSUPPORTED when librccl is absent.
 *      hefx_allreduce_sum: in place, all ranks -- all-reduce(SUM) of the uint64 words over xGMI, then every word back
 *      to [0,q_j): bit-identical to a serial add_many of the ranks' partials (helper.h:259).
 *      hefx_reduce_canonical is the local half alone, for callers that run the collective themselves
 *      (parallel.py over torch.distributed). */
int hefx_comm_unique_id(uint8_t *id128);
int hefx_comm_init(hefx_context *ctx, int world, int rank, const uint8_t *id128);
int hefx_comm_destroy(hefx_context *ctx);
int hefx_comm_world(const hefx_context *ctx); /* 0 = no communicator */
int hefx_comm_rank(const hefx_context *ctx);
int hefx_allreduce_sum(hefx_context *ctx, int L, int size, uint64_t *d_ct, void *stream);
int hefx_reduce_canonical(hefx_context *ctx, int L, int size, uint64_t *d_data, int addends, void *stream);

/* ---- Linear_Transform_Plain(ct, U_diagonals[d], gal_keys) (helper.h:237-262 = linear_transformation2.cpp:149-174)
 *      as one call: ct_new = ct + rotate(ct, -d); out = sum_l rotate(ct_new, l) * diag[l].  Rotations follow SEAL's
 *      rotate_internal (direct key when (key_elts, keys) holds it, else the NAF terms of the step); the NAF plans,
 *      their de-duplication, batching and the fusion of each plan's last key switch with multiply_plain happen
 *      behind the call.  Bit-identical to the op-by-op sequence.  HEFX_ERR_INVALID "Galois key not present" when a
 *      needed element is missing; the caller checks plaintext zero-ness (transparent result) beforehand.
 *      A forest of 96 or more key switches below ct_new (the reference's default keys from d ~ 70 on) is dealt onto TWO
 *      lanes -- the subtrees of about half the nodes each, every depth of a lane one batch on a stream of its own, the
 *      second lane in the back half of the scratch buffer -- joined before the final sum; same words (HEFX_LT_LANES=0: one
 *      lane). */
int hefx_linear_transform_plain(hefx_context *ctx, int L, const uint64_t *d_ct, int d,
                                const uint64_t *const *d_diag_pts, int nkeys, const uint32_t *key_elts,
                                const uint64_t *const *d_keys, uint64_t *d_out, void *stream);
/* `count` (1..64) Linear_Transform_Plain calls of the same dimension and key set IN LOCKSTEP: d_out[t] =
 * Linear_Transform_Plain(d_cts[t], d_diag_pts[t * d .. t * d + d)).  Independent transforms -- the sigma and tau
 * transforms of ctA and ctB in CC_Matrix_Multiplication (matrix_multiplication.cpp:22-25), the n independent products of
 * matrix_mult_benchmark.cpp -- share every launch sequence (the -d rotations, each depth of the rotation forest): the same
 * number of dependent sequences as ONE transform, each `count` times as wide.  Per input the operations and their order
 * are those of hefx_linear_transform_plain: same words.  Outputs pairwise distinct and distinct from the inputs. */
int hefx_linear_transform_plain_many(hefx_context *ctx, int L, int count, const uint64_t *const *d_cts, int d,
                                     const uint64_t *const *d_diag_pts, int nkeys, const uint32_t *key_elts,
                                     const uint64_t *const *d_keys, uint64_t *const *d_outs, void *stream);

/* ---- A FOREST of rotations in one call: node i rotates the result of node parent[i] (parent[i] < 0: the ciphertext
 *      d_ext_in[i]) by galois_elts[i] with d_keys[i] into d_out[i]; a non-NULL d_pts[i] multiplies the rotated ciphertext
 *      by that plaintext in the key-switch epilogue (d_pts itself may be NULL: no products).  parent[i] < i.  This is what
 *      Evaluator::rotate_vector's NAF chains of a linear transform add up to once shared prefixes are computed once
 *      (helper.h:252-257 with the default keys of linear_transformation2.cpp:239): the nodes run depth by depth -- one
 *      batch per depth and lane, the subtrees of a forest of 96 nodes or more on two lanes, wide one-source depths
 *      exactly hoisted -- the schedule hefx_linear_transform_plain uses behind its own planner.  Every node's words are
 *      those of hefx_apply_galois / hefx_rotate_multiply_plain_batch on the same operands.  Outputs must be pairwise
 *      disjoint and disjoint from every external input and plaintext (HEFX_ERR_INVALID before anything runs). */
int hefx_apply_galois_forest(hefx_context *ctx, int L, int n, const int32_t *parent, const uint64_t *const *d_ext_in,
                             const uint32_t *galois_elts, const uint64_t *const *d_keys, const uint64_t *const *d_pts,
                             uint64_t *const *d_out, void *stream);

/* ---- HOISTED rotations: n rotations of ONE ciphertext share its digit decomposition -- INTT and
 *      digit x modulus NTTs of c1 once, each rotation gathers them through its Galois table, multiplies with its key and
 *      mods down: (L+1)(L+2) -> 2 + 2L transforms per rotation.  EXACT since round 4, i.e. the same words as
 *      hefx_apply_galois_batch / Evaluator::rotate_vector with these keys: SEAL decomposes the rotated polynomial, whose
 *      digit is the signed permutation of the source's plus q_i on the negated coefficients; the key MAC adds that term
 *      as (q_i mod m) * NTT_m(flip mask of the element) (csrc/hefx_keyswitch.hip ks_mac_exact_kernel; the identity is
 *      pinned on the CPU in tests/test_oracle_pinning.py, and it excludes zero coefficients, which the device detects
 *      and redoes per item).  Rounds 1-3 shipped the uncorrected sum as a fast mode with other words.
 *      hefx_apply_galois_batch / hefx_rotate_multiply_plain_batch / hefx_apply_galois_add_batch take this path by
 *      themselves when a batch of more than 32 items rotates at most n/3 distinct ciphertexts, so this entry is now the
 *      same computation with a stricter contract (one source, which no output may alias) -- kept for its callers.
 *      d_pts may be NULL (no fused multiply_plain).
 *      FIRST USE of a Galois element on this path builds its flip-mask table (k rows of N words, kept for the context's
 *      life within HEFX_FLIPW_MB, default 8 GiB; beyond it the batch runs unhoisted, same words): one hipMalloc and one
 *      hipStreamSynchronize on the caller's stream -- the one place where an asynchronous entry waits on the host, and a
 *      reason not to capture the first call of a new element into a graph.  Later calls find the table. */
int hefx_rotate_hoisted_batch(hefx_context *ctx, int L, const uint64_t *d_ct_in, int n, const uint32_t *galois_elts,
                              const uint64_t *const *d_keys, const uint64_t *const *d_pts, uint64_t *const *d_ct_out,
                              void *stream);
/* Linear_Transform_Plain with the d-1 rotations of ct_new hoisted; needs a DIRECT Galois key for every step 1..d-1
 * (keygen.galois_keys(steps)); the -d rotation is a regular one.  Same words as hefx_linear_transform_plain with those
 * keys (which hoists by itself from d = 34 on); refuses key sets without the direct keys. */
int hefx_linear_transform_plain_hoisted(hefx_context *ctx, int L, const uint64_t *d_ct, int d,
                                        const uint64_t *const *d_diag_pts, int nkeys, const uint32_t *key_elts,
                                        const uint64_t *const *d_keys, uint64_t *d_out, void *stream);

/* DOUBLE hoisting: besides the shared decomposition, the d-1 products diag_l * rot_l(ct_new) are accumulated in the
 * extended basis (data primes + special prime) and modded down ONCE: per rotation only a gathered key MAC remains.
 * d_diag_pts_keylevel[l] are KEY-LEVEL plaintexts ([k][N]: encode with parms_id = key level); L must be the top data
 * level (k-1); direct Galois keys for 1..d-1.  One rounding instead of d-1: not the bits of the rotation-by-rotation
 * sum (nor SEAL's); bit-exact against the oracle's statement of this algorithm (orc_lt_double_hoisted). */
int hefx_linear_transform_plain_hoisted2(hefx_context *ctx, int L, const uint64_t *d_ct, int d,
                                         const uint64_t *const *d_diag_pts_keylevel, int nkeys,
                                         const uint32_t *key_elts, const uint64_t *const *d_keys, uint64_t *d_out,
                                         void *stream);
/* Double hoisting over a SUBSET of the diagonals (the permutation matrices of the matrix product have 2n-1 or n
 * non-zero diagonals out of n^2, matrix_multiplication.cpp:239-297): term i multiplies rotate(ct_new, steps[i]) by
 * d_diag_pts_keylevel[i]; steps[0] must be 0, every other step non-zero with a direct Galois key; d only fixes the
 * duplication rotate(ct, -d) of helper.h:244. */
int hefx_linear_transform_plain_hoisted2_sparse(hefx_context *ctx, int L, const uint64_t *d_ct, int d, int nterms,
                                                const int *steps, const uint64_t *const *d_diag_pts_keylevel,
                                                int nkeys, const uint32_t *key_elts, const uint64_t *const *d_keys,
                                                uint64_t *d_out, void *stream);

/* Baby-step / giant-step form of Linear_Transform_Plain (helper.h:237-262).  With n2 = ceil(d/n1)
 * and l = j*n1 + i:  sum_l diag_l (.) rot_l(ct_new) = sum_j rot_(j*n1)( sum_i diag'_l (.) rot_i(ct_new) ), where
 * d_shifted_diag_pts[l] encodes diag_l shifted RIGHT by j*n1 slots (done in the clear before encoding).  Needs direct
 * Galois keys for the steps 1..n1-1 and n1, 2*n1, .., (n2-1)*n1 (n1+n2-2 key switches instead of d-1); rotate(-d) may
 * use a NAF chain.  d + n1*n2 <= N/2.  hoisted_baby: the baby rotations go through hefx_rotate_hoisted_batch (same
 * words either way).  A different operation sequence than the reference's loop -- every primitive in it is one of SEAL's
 * bit for bit -- so the same plaintext result with different noise bits than Linear_Transform_Plain; the checker is the
 * same composition over the oracle. */
int hefx_linear_transform_plain_bsgs(hefx_context *ctx, int L, const uint64_t *d_ct, int d, int n1,
                                     const uint64_t *const *d_shifted_diag_pts, int nkeys, const uint32_t *key_elts,
                                     const uint64_t *const *d_keys, int hoisted_baby, uint64_t *d_out, void *stream);

/* ---- CKKSEncoder::encode(vector<double>, scale, plain) on the GPU (call sites
 *      matrix_mult_benchmark.cpp:291-323, logistic_regression_ckks.cpp:222-225,302-305, helper.h:333-343):
 *      `count` vectors of `nvalues` <= N/2 slot values each (host arrays; h_im may be NULL for real vectors) ->
 *      `count` contiguous NTT-form plaintexts of L rows at d_out.  Canonical embedding with slot i <-> root
 *      zeta^(3^i), coefficients rounded half away from zero like std::round.  Floating point: matches any other
 *      correct encoder to +-1 in a small fraction of coefficients, not bit for bit.
 *      The host arrays are copied into pinned staging memory before the call returns (they may be transient); the call
 *      does not wait for the stream -- at most for an earlier encode that still owns the staging buffer it wants. */
int hefx_ckks_encode(hefx_context *ctx, int L, const double *h_re, const double *h_im, int nvalues, int count,
                     double scale, uint64_t *d_out, void *stream);
/* the same for `count` vectors whose plaintexts are separately allocated: d_outs[i] receives vector i (host arrays as
 * above, vector i at h_re + i * nvalues).  One encode pass and one scatter launch per 256 vectors; the words
 * hefx_ckks_encode writes. */
int hefx_ckks_encode_batch(hefx_context *ctx, int L, const double *h_re, const double *h_im, int nvalues, int count,
                           double scale, uint64_t *const *d_outs, void *stream);

/* ---- randomness, Encryptor::encrypt, Decryptor::decrypt on the GPU (call sites
 *      linear_transformation2.cpp:344-350, logistic_regression_ckks.cpp:362-381, matrix_multiplication.cpp:419).
 *      Counter-mode sampling: every 64-bit word is a pure function of (key32 = 32-byte ChaCha20 key, stream id,
 *      position) -- specification in csrc/hefx_sample.hip and oracle/ckks_oracle.c (orc_sample_*), which give the
 *      same bits.  Distributions as SEAL 3.4.5: uniform mod q (rejection), ternary {-1,0,1}, clipped normal
 *      sigma 3.2 / bound 19.2 truncated toward zero.  Output [npoly][nrows][N], rows reduced mod
 *      q_(mod_first+row); ternary / noise write ONE draw per coefficient into every row, COEFFICIENT form.
 *      The caller owns the key: fresh 32 bytes from the OS per key generator / encryptor, a new stream id per call. */
int hefx_sample_uniform(hefx_context *ctx, const uint8_t *key32, uint64_t stream_id, int npoly, int nrows,
                        int mod_first, uint64_t *d_out, void *stream);
int hefx_sample_ternary(hefx_context *ctx, const uint8_t *key32, uint64_t stream_id, int npoly, int nrows,
                        int mod_first, uint64_t *d_out, void *stream);
int hefx_sample_noise(hefx_context *ctx, const uint8_t *key32, uint64_t stream_id, int npoly, int nrows,
                      int mod_first, uint64_t *d_out, void *stream);
/* KeyGenerator's key-switching keys (relin_keys, galois_keys; App. A.11) entirely on the device:
 * out[k-1][2][k][N] (SEAL's layout) = for digit i: (-(a_i*sk + e_i) + [row i] (P mod q_i)*new_sk, a_i), a_i uniform from
 * sub-stream 2*stream_id, e_i noise from 2*stream_id+1.  d_sk, d_new_sk: [k][N] NTT form (new_sk = sk^2 for the
 * relinearisation key, hefx_galois_permute(sk) for a Galois key). */
int hefx_keygen_kswitch(hefx_context *ctx, const uint64_t *d_sk, const uint64_t *d_new_sk, const uint8_t *key32,
                        uint64_t stream_id, uint64_t *d_out, void *stream);
/* out[r][w] = in[r][perm_g[w]] for `rows` rows of N words: the NTT-domain automorphism X -> X^g of plain
 * polynomials (SEAL apply_galois_ntt); in != out */
int hefx_galois_permute(hefx_context *ctx, uint32_t galois_elt, const uint64_t *d_in, int rows, uint64_t *d_out,
                        void *stream);
/* out[2][L][N] = (pk0*u + e0 + plain, pk1*u + e1), NTT form; d_pk = [2][k][N] (key-level public key), d_plain may be
 * NULL (encryption of zero); u ternary from sub-stream 4*stream_id, e0 / e1 noise from 4*stream_id+1 / +2. */
int hefx_encrypt(hefx_context *ctx, int L, const uint64_t *d_pk, const uint64_t *d_plain, const uint8_t *key32,
                 uint64_t stream_id, uint64_t *d_out, void *stream);
/* n encryptions under one public key and one sampler key, item i with stream id first_stream_id + i: the words n
 * hefx_encrypt calls with those ids produce, from five launches per 256 items instead of five per item.  d_plains may be
 * NULL, and so may any d_plains[i] (encryption of zero). */
int hefx_encrypt_batch(hefx_context *ctx, int L, int n, const uint64_t *d_pk, const uint64_t *const *d_plains,
                       const uint8_t *key32, uint64_t first_stream_id, uint64_t *const *d_outs, void *stream);
/* out[L][N] = c0 + c1*s + ... + c_(size-1)*s^(size-1), NTT form; d_sk = NTT-form secret key rows [>=L][N] */
int hefx_decrypt(hefx_context *ctx, int L, int size, const uint64_t *d_ct, const uint64_t *d_sk, uint64_t *d_out,
                 void *stream);

/* ---- CKKSEncoder::decode(plain, vector<double>&) on the GPU (linear_transformation2.cpp:396-400,
 *      logistic_regression_ckks.cpp:362-381 -- the decrypt/decode/encode/encrypt refresh of the LR loop):
 *      `count` NTT-form plaintexts of L rows -> slot values, h_re/h_im[count][N/2] (h_im may be NULL).  Inverse NTT,
 *      CRT composition (Garner, exact), centring against Q/2, division by the scale and the slot-root evaluation
 *      (two N/2-point complex FFTs) all on the device.  Floating point: agrees with a host decode to ~1e-12 relative.
 *      Blocks until the values are in the host arrays. */
int hefx_ckks_decode(hefx_context *ctx, int L, const uint64_t *d_pt, int count, double scale, double *h_re,
                     double *h_im, void *stream);

/* ---- measurement helpers (no reference counterpart; the reference times with std::chrono around the L3
 *      call, e.g. linear_transformation2.cpp:363-365).  HIP events recorded on the stream the kernels use. */
int hefx_event_create(hefx_context *ctx, void **event);
int hefx_event_destroy(hefx_context *ctx, void *event);
int hefx_event_record(hefx_context *ctx, void *event, void *stream);
int hefx_event_elapsed_ms(hefx_context *ctx, void *event_start, void *event_stop, float *ms); /* blocks */
/* Between begin and end every key-switch chunk runs serially on the caller's stream with an event between
 * its launches; end returns the summed duration per launch kind (see hefx_profile_stage_name) and the
 * number of chunks, so average launch duration = stage_ms[k] / launches. */
int hefx_profile_begin(hefx_context *ctx);
#define HEFX_PROFILE_STAGES 8 /* launch kinds hefx_profile_end reports; the last one: the (normally empty) fallback launches of
                                 an exactly hoisted chunk */
int hefx_profile_end(hefx_context *ctx, double *stage_ms /* [HEFX_PROFILE_STAGES] */, uint64_t *launches);
const char *hefx_profile_stage_name(int k);

#ifdef __cplusplus
}
#endif
#endif /* HEFX_H */

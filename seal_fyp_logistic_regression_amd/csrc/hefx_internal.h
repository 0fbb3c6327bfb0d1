// hefx_internal.h -- structures shared between the host side (hefx_capi.cpp) and the gfx950 kernels
// (hefx_kernels.hip).  Not part of the public ABI (that is include/hefx.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hefx_modarith.cuh"

namespace hefx {

// hipFuncSetAttribute (the > 64 KiB dynamic-LDS opt-in) is per device: launchers keep one of these per kernel family
// and apply the attributes the first time they run on each device of the process.
struct PerDeviceOnce {
    static constexpr int MAX_DEV = 64;
    bool done[MAX_DEV] = {};
    bool first()
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return true;  // unknown: just re-apply
        if (done[dev]) return false;
        done[dev] = true;
        return true;
    }
};

// Device-resident constant tables of one context (all pointers are device pointers).
struct DevTables {
    const ulonglong2 *tw;      // [k][N]  forward twiddles {w, floor(w*2^64/q)}, w[bitrev(i)] = psi^i
    const ulonglong2 *itw;     // [k][N]  inverse twiddles, itw[idx] = tw[idx]^-1
    const double *twf;         // [k][N]  FP64-policy forward twiddles (primes < 2^41 only), 8 bytes each
    const double *itwf;        // [k][N]  FP64-policy inverse twiddles
    const ModConst *mods;      // [k]
    const ModConstF *modsf;    // [k]     q == 0 marks a prime too wide for the FP64 policy
    const ulonglong2 *invmod;  // [k][k]  invmod[l*k+j] = {q_l^-1 mod q_j, Shoup companion} (l != j)
    const u64 *halfmod;        // [k][k]  (q_l >> 1) mod q_j
    const double2 *invmodf;    // [k][k]  FP64 policy: {q_l^-1 mod q_j, RN(that / q_j)} for q_j < 2^41
    int k;
    int logn;
};

__device__ __forceinline__ NttTables ntt_tables(const DevTables &T, int m)
{
    const size_t o = (size_t)m << T.logn;
    NttTables nt;
    nt.tw = T.tw + o;
    nt.itw = T.itw + o;
    nt.twf = T.twf + o;
    nt.itwf = T.itwf + o;
    return nt;
}

// One key-switch work item (a rotation term or a relinearisation).
struct KsItem {
    const u64 *c_in;   // source ciphertext, [2][L][N] (rotation) or [3][L][N] (relinearisation)
    const u64 *key;    // [k-1][2][k][N]
    const uint32_t *perm;  // gather table of the Galois element (out[i] = in[perm[i]]), or nullptr for relin
    const u64 *pt;     // optional plaintext [L][N] multiplied into the result (fused multiply_plain)
    u64 *c_out;        // [2][L][N]
    // optional accumulate (hefx_apply_galois_add_batch, helper.h:474-475 rotate_vector_inplace + add_inplace as one key
    // switch): acc_out = acc_in + c_out, [2][L][N] each; acc_in == acc_out is the in-place sum; both null: none
    const u64 *acc_in;
    u64 *acc_out;
    uint32_t elt;      // Galois element of a rotation (the kernels compute the gather index from it); 0 / 1: identity
    uint32_t flags;    // KS_ALIASED: c_out is the caller's input as well -- c_in points at a scratch copy of it
    // exact hoisting (ks_mac_exact_kernel): which entry of the chunk's source list this item rotates, and NTT_m of the 0/1
    // polynomial that marks the coefficients X -> X^elt negates, [k][N] (row k-1: the special prime) -- one table per
    // Galois element, cached by the context like `perm`
    uint32_t dsrc, pad_;
    const u64 *flipw;
};
constexpr uint32_t KS_ALIASED = 1;

// (global-address-space accessors gld8 / gld16 / gst16 ...: hefx_modarith.cuh)

// NTT-domain index map of the automorphism X -> X^elt: out[i] = in[galois_index(i)], with
// galois_index(i) = bitrev(((elt * (2 bitrev(i) + 1)) mod 2N - 1) / 2).  Two v_bfrev, one 24-bit multiply: cheaper
// than a table lookup and no table to keep cache-resident.  Structure the kernels rely on: positions 2j and 2j+1 map
// to a pair (2m, 2m+1) in either order (elt is odd), and more generally every aligned block of 2^s positions maps onto
// an aligned block of 2^s positions -- a gathered row touches exactly the cache lines of a straight read.
__device__ __forceinline__ uint32_t galois_index(uint32_t i, uint32_t elt, int logn)
{
    const uint32_t br = __brev(i) >> (32 - logn);
    const uint32_t raw = __umul24(elt, 2u * br + 1u) & ((2u << logn) - 1u);  // elt < 2N <= 2^16, 2 br + 1 < 2^16
    return __brev((raw - 1u) >> 1) >> (32 - logn);
}

// Items of one chunk live in a device-side descriptor ring (filled through a pinned host mirror with one async
// copy per chunk), so a chunk is not limited by the 4 KiB kernel-argument segment.
constexpr int KS_MAX_CHUNK = 1024;   // descriptor-ring slot size: the most items HEFX_CHUNK (or the size rule) may ask for
constexpr int KS_AUTO_CHUNK = 256;  // what the size rule asks for at most (N >= 16384: 256 items fill the chip in whole rounds)
constexpr int KS_RING = 32;  // (a laned linear transform submits ten batches and a pointer table per call: with 8 slots the host waited on its own call)

// Scratch layout for one chunk of key-switch items, in units of N words per item.  The Galois-permuted inputs are
// never materialised: the inverse transform of the digits, the own-prime term of the key MAC and the add-in of the
// mod-down epilogue gather them from the source ciphertext (galois_index).
struct KsScratch {
    u64 *d;    // [chunk][L][N]        digits in coefficient form
    u64 *x;    // [sub][L][L+1][N]     digit i transformed to modulus slot jj != i (jj==L: special prime); only a
               //                      SUB-chunk of items at a time, so this largest scratch array stays cache-resident
    u64 *acc;  // [chunk][2][L+1][N]   sum_i x_i * key_i, reduced
    u64 *u;    // [chunk][2][N]        INTT_P(acc_P) + P/2, coefficient form
    u64 *alias;  // [chunk][2][L][N]   copies of the inputs of in-place rotations (c_in == c_out), else unused
    // exact hoisting: q_i mod q_m for every pair of key moduli ([k][k], device), and the chunk's gate -- the source
    // decomposition stores gate_tag into *gate when a digit holds a ZERO coefficient (the one case the hoisted form does not
    // cover); gate_mode 1: the launch does nothing when the gate was hit, 2: only when it was hit (the per-item fallback), 0: no gate
    const u64 *qmod;
    uint32_t *gate, *gate_hits;  // gate_hits: chunks redone by the fallback so far (hefx_ks_fallback_count)
    uint32_t gate_tag, gate_mode;
};

constexpr int ADD_MANY_GROUP = 48;
struct PtrGroup {
    const u64 *p[ADD_MANY_GROUP];
};

// ---- launchers implemented in hefx_kernels.hip; all asynchronous on `s`; return hipError_t ----
hipError_t launch_ntt(const DevTables &T, bool inverse, u64 *data, int npoly, int nrows, int mod_first,
                      hipStream_t s);
enum EwOp { EW_ADD = 0, EW_SUB = 1, EW_NEG = 2, EW_MULPLAIN = 3, EW_ADDPLAIN = 4, EW_REDUCE = 5 };
// generic element-wise op over `count` ciphertexts of `size` polys and L rows. b may be null (NEG/REDUCE).
// For MULPLAIN/ADDPLAIN b is a plaintext [L][N] (per ciphertext stride 0).
hipError_t launch_elementwise(const DevTables &T, EwOp op, int L, int size, int count, const u64 *a,
                              const u64 *b, u64 *out, int *flag, hipStream_t s);
// pt0 != nullptr (first group only): the first addend is g.p[0] (.) pt0
hipError_t launch_add_many(const DevTables &T, int L, int size, const PtrGroup &g, int n, bool accumulate,
                           u64 *out, hipStream_t s, const u64 *pt0 = nullptr);
hipError_t launch_add_many_table(const DevTables &T, int L, int size, const u64 *const *d_ptrs, int n, int group,
                                 u64 *partial, hipStream_t s);
// table: n ciphertext pointers | n plaintext pointers | ceil(n/group) output pointers (device memory)
hipError_t launch_mulplain_sum(const DevTables &T, int L, int size, const u64 *const *d_tab, int n, int group,
                               hipStream_t s);
// table: n a-pointers | n b-pointers | n output pointers (device memory)
hipError_t launch_multiply_table(const DevTables &T, int L, const u64 *const *d_tab, int n, hipStream_t s);
hipError_t launch_multiply(const DevTables &T, int L, const u64 *a, const u64 *b, u64 *out3, hipStream_t s);
// table: n a-pointers | n b-pointers | n output pointers (device memory); element-wise add / sub of size*L rows
hipError_t launch_addsub_table(const DevTables &T, bool sub, int L, int size, const u64 *const *d_tab, int n,
                               hipStream_t s);
// profiling: an event is recorded before every launch (tagged with its stage) and one after the last
constexpr int KS_STAGES = 8;  // == HEFX_PROFILE_STAGES (hefx.h)
struct KsProf {
    hipEvent_t *ev;  // capacity `cap`
    int *stage;      // stage[i] = launch kind that follows ev[i]; -1 terminates a chunk
    int cap, used;
};
// small_items != nullptr (host copy of the n <= ks_small_max() descriptors of a non-aliasing chunk without sources): the
// descriptors are passed in the first launch's arguments; d_items is then written by that launch, not copied to
// quarter: additionally run the chunk on quarter-row workgroups (four per row, eight coefficients per thread)
// nsrc > 0: exact hoisting (ks_mac_exact_kernel) -- d_items[n .. n + nsrc) describe the chunk's distinct source
// ciphertexts; they are decomposed and extended once (scr.d / scr.x hold SOURCE rows), every item runs the gathered MAC
hipError_t launch_keyswitch_chunk(const DevTables &T, int L, int n, const KsItem *d_items, bool relin,
                                  const KsScratch &scr, int sub, bool alias, const KsItem *small_items,
                                  int quarter, hipStream_t s, KsProf *prof, int nsrc = 0);
// tables of exact hoisting: rows[e][m][.] = the flip mask (coefficient order) of the Galois element whose inverse mod 2N is
// d_ginv[e], once per modulus row m = 0..k-1; the caller transforms the rows
hipError_t launch_flip_rows(const DevTables &T, const uint32_t *d_ginv, int count, u64 *rows, hipStream_t s);
// `quarter`: which of the four transform launches run on quarter-row workgroups
constexpr int KS_Q_INTT = 1, KS_Q_NTT = 2, KS_Q_MDI = 4, KS_Q_FIN = 8, KS_Q_ALL = 15;
// ... or the whole chunk on the pair path (ks_pair_*: four launches, two transform phases; overrides the mask above)
constexpr int KS_Q_PAIR = 16;
int ks_small_max();
// double-hoisted linear transform (hefx_keyswitch.hip): see lt2_mac_kernel
hipError_t launch_lt2_decompose(const DevTables &T, int L, const KsItem *src_item, const KsItem *rot_items, int nrot,
                                const KsScratch &scr, const u64 *ct_new, u64 *partial_s, u64 *partial_c0, u64 *cbuf,
                                hipStream_t s);
hipError_t launch_lt2_moddown(const DevTables &T, int L, const KsItem *item, const KsScratch &scr, hipStream_t s);
int lt2_chunk();
// out-of-place split NTT for N = 32768 (rows do not fit one workgroup's LDS)
hipError_t launch_ntt_split15(const DevTables &T, bool inverse, const u64 *src, u64 *dst, int npoly, int nrows,
                              int mod_first, hipStream_t s);
// CKKS encode (hefx_encode.hip)
struct EncodeTables {
    const int *slot;      // [N/2]  r -> (i << 1) | conj
    const double2 *wfft;  // [N/4]  exp(-2 pi i m / (N/2))
    const double2 *pre;   // [N/2]  exp(-2 pi i r / N)
    const double2 *post;  // [N]    exp(-pi i k / N)
};
struct DecodeTables {
    u64 half[16];  // mixed-radix digits of floor(Q_L / 2), Q_L = q_0 ... q_(L-1), radix (q_0, q_1, ...)
};
hipError_t launch_decode(const DevTables &T, const EncodeTables &E, const DecodeTables &D, int L, const u64 *coef,
                         int count, double scale, double *p, double *re, double *im, hipStream_t s);
hipError_t launch_encode(const DevTables &T, const EncodeTables &E, const double *re, const double *im, int nvalues,
                         int count, double scale, int L, u64 *out, hipStream_t s);
// sampling + encrypt/decrypt arithmetic (hefx_sample.hip)
struct SampleKey {
    uint32_t w[8];  // ChaCha20 key, little-endian words of the 32 key bytes
};
struct NoiseTable {
    u64 t[39];  // value = -19 + #{i < 38 : r >= t[i]}
};
enum { SAMPLE_UNIFORM = 0, SAMPLE_TERNARY = 1, SAMPLE_NOISE = 2 };
hipError_t launch_sample(const DevTables &T, int mode, const SampleKey &key, const NoiseTable &tab, u64 stream,
                         int npoly, int nrows, int mod_first, u64 *out, hipStream_t s, u64 stream_stride = 0);
hipError_t launch_encrypt_combine(const DevTables &T, int L, const u64 *pk, const u64 *u, const u64 *e,
                                  const u64 *plain, u64 *out, hipStream_t s);
// m encryptions: u [m][L][N], e [2][m][L][N], d_tab = m plaintext pointers (null allowed) | m output pointers
hipError_t launch_encrypt_combine_table(const DevTables &T, int L, int m, const u64 *pk, const u64 *u, const u64 *e,
                                        const u64 *const *d_tab, hipStream_t s);
// dst[i][0..words) = src + i*words for i < n (device pointer table): contiguous batch results to their owners
hipError_t launch_scatter_rows(const u64 *src, const u64 *const *d_tab, int n, size_t words, hipStream_t s);
hipError_t launch_decrypt(const DevTables &T, int L, int size, const u64 *ct, const u64 *sk, u64 *out, hipStream_t s);
hipError_t launch_keygen_combine(const DevTables &T, const u64 *sk, const u64 *new_sk, const u64 *a, const u64 *e,
                                 u64 *out, hipStream_t s);
hipError_t launch_galois_permute(const DevTables &T, const uint32_t *perm, const u64 *in, int rows, u64 *out,
                                 hipStream_t s);
hipError_t warm_kernels(hipStream_t s);
hipError_t warm_keyswitch(hipStream_t s);
hipError_t warm_encode(hipStream_t s);
hipError_t warm_sample(hipStream_t s);
// in / out: `count` contiguous ciphertexts, or (tab != nullptr) a device pointer table  in[0..count) | out[0..count)
hipError_t launch_rescale(const DevTables &T, int L, int size, int count, const u64 *in, u64 *out,
                          const u64 *const *tab, u64 *scratch_d, bool rounded, hipStream_t s);

}  // namespace hefx

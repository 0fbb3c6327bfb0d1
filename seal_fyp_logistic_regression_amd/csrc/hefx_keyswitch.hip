// hefx_keyswitch.hip -- key switching (K5/K6/K7), rescale (K8) and the large-N stand-alone NTT, built on
// "split-2" NTT workgroups.
//
// Why split: a 2^14-point row needs 136 KiB of LDS, i.e. ONE 1024-thread workgroup per CU whose load / compute
// / exchange / store phases run back to back; measured, such a workgroup spends about as long waiting on its
// ~24 GB/s-per-CU memory phases as on its integer butterflies.  Every transform here is therefore run by TWO
// workgroups of N/32 threads that each own a 2^(LOGN-1)-point sub-transform (68 KiB LDS at N=16384), so two
// (or more) independent workgroups share a CU and one's memory phase overlaps the other's butterflies:
//   forward (Cooley-Tukey):   the first stage (gap N/2, one twiddle) is applied while loading -- both halves
//       load x[j], x[j+N/2] and keep X (half 0) or Y (half 1); the remaining stages are an independent
//       N/2-point transform of that half with twiddle prefix 2+h.  ~7 % redundant multiplies, no exchange.
//   inverse (Gentleman-Sande): the first stage (gap 1) pairs (2j, 2j+1); half 0 keeps the sums, half 1 the
//       twiddled differences, and from then on even and odd positions never meet again: each half is a
//       standard N/2-point inverse transform (same table, N^-1 folded as usual) whose result is the even /
//       odd coefficients.  Inverse outputs are stored DE-INTERLEAVED ([evens | odds], "EO"), which the
//       forward loaders read with full coalescing (x[j] and x[j+N/2] have the parity of the thread id).
// The two halves of a row get block ids that differ by 8, i.e. land on the same XCD (same L2) under the
// round-robin dispatch, so the second read of the row is an L2 hit.  Placement only affects speed.
#include <cstdio>
#include <cstdlib>
#include "hefx_internal.h"
#include "hefx_ntt.cuh"
#include "hefx_ntt8.cuh"

// Minimum waves per SIMD the NTT workgroups are register-allocated for (second __launch_bounds__ argument):
// 4 caps a kernel at 128 VGPRs, 2 lets it use 256.  Measured per kernel and size on MI355X (bench workload,
// us per 256- / 192-item chunk, "4" vs "2"):
//   N = 8192 : inverse digits 85 vs 52, digit NTTs 152 vs 124, mod-down finish 364 vs 285 -> 2 everywhere (at 4 the
//              256-thread workgroups spill up to 148 B/lane; at 2 three of them still share a CU)
//   N = 16384: inverse digits 138 vs 107 -> 2;  digit NTTs 455 vs 489, mod-down finish 318 vs 436 -> 4 (one
//              512-thread workgroup per CU cannot overlap its memory phases with another's butterflies)
//   N = 32768: 1024-thread workgroups own the whole register file at 128 VGPRs either way.
// -DHEFX_WAVES=n overrides every kernel (tools/build_variant.sh).
template <int LOGN>
struct KsWaves {
#ifdef HEFX_WAVES
    static constexpr int INV = HEFX_WAVES, FWD = HEFX_WAVES;
#else
    static constexpr int INV = LOGN <= 14 ? 2 : 4;  // ks_intt_digits, ks_moddown_intt, rs_intt
    static constexpr int FWD = LOGN <= 13 ? 2 : 4;  // ks_ntt_digits, ks_moddown_finish, rs_finish
#endif
    // load batches of the first (split) stage: all sixteen coefficient pairs are fetched at once in the 256-VGPR
    // builds of N <= 8192 (a few percent there; at N = 16384 the inverse kernels lose 5-15 % with one batch)
    // (N = 8192 forward: two batches since round 3 -- with the lane-contiguous loader (eo_lane) one batch took the
    // 256-thread workgroups two registers past 128 and with them the fourth workgroup per CU; two batches + eo_lane:
    // digit NTTs 94 -> 87 us, mod-down finish 102 -> 98 us per chunk, profiles/r03/ab_c2_eo_lane.txt)
    static constexpr int NB_INV = (INV <= 2 && LOGN <= 13) ? 1 : 2, NB_FWD = (FWD <= 2 && LOGN < 13) ? 1 : 2;
};

namespace hefx {

// 16-byte streaming (nontemporal) accesses: one global_load/store_dwordx4 nt per record.  As two 8-byte accesses a
// streamed row costs twice the memory instructions, and 8-byte nt / sc1 accesses run at 0.54-0.70 x the 16-byte rate
// (MI355X_MICROARCH.md, inter-workgroup visibility table).
typedef u64 u64x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ulonglong2 nt_load16(const u64 *p)
{
    const u64x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t *>(p));
    return make_ulonglong2(v.x, v.y);
}
__device__ __forceinline__ void nt_store16(u64 *p, u64 a, u64 b)
{
    u64x2_t v;
    v.x = a, v.y = b;
    __builtin_nontemporal_store(v, reinterpret_cast<u64x2_t *>(p));
}

template <int LOGN>
struct SplitCfg {
    using C = NttCfg<LOGN - 1>;  // the per-workgroup sub-transform
    static constexpr int N = 1 << LOGN;
    static constexpr int H = N / 2;
    static constexpr int T = C::T;  // threads per workgroup = N/32
    static constexpr size_t LDS_BYTES = sizeof(u64) * C::LDS_WORDS;
};

// rows -> grid: halves of row p sit at block ids that differ by 8
__host__ __device__ static inline int split_grid(int rows) { return ((rows + 7) / 8) * 16; }
__device__ static __forceinline__ void split_decode(int bid, int &p, int &h)
{
    h = (bid >> 3) & 1;
    p = ((bid >> 4) << 3) | (bid & 7);
}
// Grouped mapping for kernels in which `fan` rows read the SAME source row (a digit feeds L target moduli, a
// mod-down remainder feeds L data primes): all 2*fan half-workgroups of a source get block ids that are equal
// modulo 8, i.e. the same XCD under round-robin dispatch, and adjacent in dispatch order -- the source row is
// fetched into that XCD's L2 once and the other 2*fan-1 reads hit it.  Speed hint only.
__host__ __device__ static inline int group_grid(int groups, int fan) { return ((groups + 7) / 8) * 2 * fan * 8; }
__device__ static __forceinline__ void group_decode(int bid, int fan, int &g, int &member, int &h)
{
    const int x = bid & 7, rest = bid >> 3;
    const int slot = rest % (2 * fan), sg = rest / (2 * fan);
    g = sg * 8 + x;
    member = slot >> 1;
    h = slot & 1;
}
// index of coefficient j in a row stored de-interleaved
__device__ static __forceinline__ int eo(int j, int H) { return (j & 1) * H + (j >> 1); }
// the same for j = idx_nat(t, r) = t + T*r of a split-2 loader, written so that the address arithmetic stays 32-bit and
// affine in r (T is even: the parity is the thread's; one v_add_u32 or an immediate offset per load against a 64-bit
// add with carry per load for the general form)
template <class SC>
__device__ static __forceinline__ uint32_t eo_nat(int t, int r)
{
    static_assert(SC::C::T % 2 == 0, "split-2 loaders assume an even thread count");
    return (uint32_t)((t & 1) * SC::H + (t >> 1)) + (uint32_t)(SC::C::T / 2) * (uint32_t)r;
}
// How lazy may the words be that the digit transforms hand the key MAC for target modulus (mc, mf) at level L: 0 canonical,
// 1 below 2q, 2 below 4q?  Only the limb policy (MacL: integer-policy prime below 2^60, L <= 8 -- mac_dispatch) has
// headroom.  Its middle column takes xl*kh + xh*kl per digit (xl, kl, kh < 2^30):
//   x < 2q < 2^61: xh < 2^31, (2^60 + 2^61) * L < 2^64 for L <= 5; top column 2^61 * 5; folded sum < 2 L q^2 < q 2^64
//                  (barrett128's domain: 10 q < 2^64)
//   x < 4q < 2^62: xh < 2^32 (still one 32-bit limb), (2^60 + 2^62) * L < 2^64 for L <= 3; top column 2^62 * 3;
//                  folded sum < 4 L q^2 = 12 q^2 < q 2^64
__device__ __forceinline__ int mac_x_slack(const ModConst &mc, const ModConstF &mf, int L)
{
#ifdef HEFX_NO_LT2Q  // A/B knob (tools/build_variant.sh): canonical operands always
    return 0;
#else
    if (mf.q != 0.0 || (mc.q >> 60) != 0) return 0;
    return L <= 3 ? 2 : (L <= 5 ? 1 : 0);
#endif
}

// The coefficient column a thread of a forward split workgroup loads (ntt_fwd_core's t0): the first half of the workgroup
// takes the even columns, the second half the odd ones, so that eo_nat(eo_lane(t), r) is contiguous over the lanes of a
// wave -- with t itself neighbouring lanes alternate between the two halves of the row and every 8-byte load is split by
// the address coalescer (found on the small-batch path with tools/stamp_timeline.py, round 3).
template <class SC>
__device__ static __forceinline__ int eo_lane(int t)
{
    constexpr int T = SC::C::T;
    return ((t & (T / 2 - 1)) << 1) | (t / (T / 2));
}
// From N = 8192 on (C3 +2.7 %, C5 +3.3 %: profiles/r03/ab_eo_lane.txt; C2 +2.5 % together with two load batches, see
// KsWaves: ab_c2_eo_lane.txt).  The smaller rings keep column t.
#ifdef HEFX_NO_EO_LANE  // A/B knob (tools/build_variant.sh): column t everywhere
#define HEFX_EO_LANE(SC, t) (t)
#else
#define HEFX_EO_LANE(SC, t) (SC::N >= 8192 ? eo_lane<SC>(t) : (t))
#endif

// ------------------------------------------------------------------------------------------------
// Galois-gathered reads.  The rotated inputs perm_g(c0), perm_g(c1) are never written out: their three readers --
// the inverse transform of the digits, the own-prime term of the key MAC, the add-in of the mod-down epilogue --
// gather them from the source ciphertext through galois_index().  Pairs stay pairs (positions 2j, 2j+1 of the
// rotated row are positions 2m, 2m+1 of the source, possibly swapped), so the 16-byte record loads survive.
// ------------------------------------------------------------------------------------------------
// record j = (value[2j], value[2j+1]) of the row perm_elt(src)
__device__ __forceinline__ ulonglong2 gather_pair(const u64 *__restrict__ src, uint32_t j, uint32_t elt, int logn)
{
    const uint32_t p = galois_index(2u * j, elt, logn);
    const ulonglong2 v = gld16(src + 2 * (size_t)(p >> 1));
    return (p & 1u) ? make_ulonglong2(v.y, v.x) : v;
}
__device__ __forceinline__ uint32_t item_elt(const KsItem &it) { return it.elt ? it.elt : 1u; }
// exact hoisting's gate (KsScratch): true when this launch has nothing to do.  Workgroup-uniform; ordinary chunks carry
// gate_mode 0 and pay one scalar compare.
__device__ __forceinline__ bool ks_gated_out(const KsScratch &S)
{
    return S.gate_mode != 0 && ((*S.gate == S.gate_tag) != (S.gate_mode == 2));
}

// (0) in-place rotations only (c_in == c_out at the ABI, e.g. rotate_vector_inplace, helper.h:474): the epilogue of the
// last kernel writes c_out while it still gathers from c0, so the host points such an item's c_in at a scratch copy
// and this kernel fills it from c_out.  Items without KS_ALIASED exit at once; no launch when a chunk has none.
__global__ __launch_bounds__(256) void ks_alias_copy_kernel(DevTables T, const KsItem *__restrict__ items, int L)
{
    const KsItem it = items[blockIdx.z];
    if (!(it.flags & KS_ALIASED)) return;
    const size_t n = (size_t)1 << T.logn;
    const size_t w = (size_t)blockIdx.y * n + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;  // row = blockIdx.y < 2L
    gst16(const_cast<u64 *>(it.c_in) + w, gld16(it.c_out + w));
}

// ------------------------------------------------------------------------------------------------
// (1) digit i of item b: d[b][i] (EO) = INTT_{q_i}(perm_g(c1)[i])   (c2[i] for a relinearisation)
// noperm: the hoisted paths decompose the UNROTATED shared source (item 0).
// ------------------------------------------------------------------------------------------------
template <int LOGN>
__device__ __forceinline__ void intt_digits_body(const DevTables &T, const KsItem &it, int L, int relin, int noperm, int b,
                                                 int i, int h, const KsScratch &S, u64 *lds)
{
    using SC = SplitCfg<LOGN>;
    using C = typename SC::C;
    const int t = threadIdx.x;
    const u64 *__restrict__ src = it.c_in + ((size_t)(relin ? 2 * L : L) + i) * SC::N;
    u64 v[16];
    if (relin || noperm || item_elt(it) == 1u) {
        split_inv<LOGN, KsWaves<LOGN>::NB_INV>(v, reinterpret_cast<const ulonglong2 *>(src), lds, ntt_tables(T, i),
                                               T.mods[i], T.modsf[i], t, h);
    } else {
        const uint32_t elt = it.elt;
        split_inv_ld<LOGN, KsWaves<LOGN>::NB_INV>(v, [src, elt](int j) { return gather_pair(src, (uint32_t)j, elt, LOGN); },
                                                  lds, ntt_tables(T, i), T.mods[i], T.modsf[i], t, h);
    }
    u64 *__restrict__ dd = S.d + ((size_t)b * L + i) * SC::N + (size_t)h * SC::H;
#pragma unroll
    for (int r = 0; r < 16; ++r) dd[C::idx_nat(t, r)] = v[r];
    if (noperm && S.gate_mode) {  // the sources of an exactly hoisted chunk: any zero coefficient hits the chunk's gate
        bool z = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) z |= v[r] == 0;
        if (z) *S.gate = S.gate_tag;
    }
}

template <int LOGN>
__global__ __launch_bounds__(SplitCfg<LOGN>::T, KsWaves<LOGN>::INV) void ks_intt_digits_kernel(DevTables T, const KsItem *__restrict__ items,
                                                                              int L, int relin, int noperm, int rows,
                                                                              KsScratch S)
{
    extern __shared__ __align__(16) u64 lds[];
    int p, h;
    split_decode(blockIdx.x, p, h);
    if (p >= rows || (!noperm && ks_gated_out(S))) return;  // (noperm: the source decomposition itself is never gated)
    intt_digits_body<LOGN>(T, items[p / L], L, relin, noperm, p / L, p % L, h, S, lds);
}

// Small batches (a lone rotation of a NAF chain, the lockstep chains of a few dot products): the descriptors travel in
// the KERNEL ARGUMENTS of this first launch -- no host-to-device descriptor copy ahead of the sequence (a ~4 us blit
// plus its dependency gap on a ~90 us operation) -- and block 0 leaves them in device memory for the launches after it.
// 32 descriptors of 80 bytes = 2.5 KB of kernel arguments (the segment holds 4 KiB: static_assert below).  (8 until round 3:
// a batch of 9..32 items then paid the descriptor copy and lost the quarter-row inverse launches -- 57 us at n = 8 against
// 80 us at n = 12 at L = 2; now 59 us.)
#ifndef HEFX_SMALL_MAX
#define HEFX_SMALL_MAX 32
#endif
constexpr int KS_SMALL_MAX = HEFX_SMALL_MAX;
struct KsSmallItems {
    KsItem it[KS_SMALL_MAX];
};
// the widest argument list that carries the descriptors: (DevTables, KsSmallItems, KsItem *, 5 ints, KsScratch)
static_assert(sizeof(KsSmallItems) + sizeof(DevTables) + sizeof(KsScratch) + sizeof(void *) + 8 * sizeof(int) <= 4096,
              "small-batch descriptors no longer fit the 4 KiB kernel-argument segment: lower HEFX_SMALL_MAX");
template <int LOGN>
__global__ __launch_bounds__(SplitCfg<LOGN>::T, KsWaves<LOGN>::INV) void ks_intt_digits_small_kernel(DevTables T, KsSmallItems small,
                                                                                    KsItem *__restrict__ items_out, int n,
                                                                                    int L, int relin, int rows, KsScratch S)
{
    extern __shared__ __align__(16) u64 lds[];
    HEFX_STAMP_KERNEL(0);
    HEFX_STAMP_AT(0);
    if (blockIdx.x == 0 && (int)threadIdx.x < n) items_out[threadIdx.x] = small.it[threadIdx.x];
    int p, h;
    split_decode(blockIdx.x, p, h);
    if (p >= rows) return;
    intt_digits_body<LOGN>(T, small.it[p / L], L, relin, 0, p / L, p % L, h, S, lds);
    HEFX_STAMP_AT(15);
}

// ------------------------------------------------------------------------------------------------
// (2) digit i -> modulus slot jj != i: x[b][i][jj] = NTT_m(d[b][i] mod m)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool slot_is_f64(const DevTables &T, int L, int jj);
template <int LOGN>
__device__ __forceinline__ void ntt_digit_row(const DevTables &T, int L, int rows, int item0, int stream_x,
                                              const KsScratch &S, u64 *lds)
{
    using SC = SplitCfg<LOGN>;
    using C = typename SC::C;
    int g, jj, h;
    if (stream_x & 2) {
        // HEAVY FIRST (small chunks, round 6): an integer-policy target row costs twice the butterflies of an FP64 one, and a
        // chunk of a few hundred workgroups is one resident wave -- which workgroups end up SHARING a CU decides when the
        // launch ends (n = 8, L = 5: per-workgroup times 8 / 14 / 22.6 us min / median / max, profiles/r06/stamps_default_L5_n8.txt).
        // Dispatch order = block id: member-major, group-minor, the integer-policy members of every digit first -- the
        // heavy workgroups take CUs of their own while there are free ones, the light ones double up.
        const int x = blockIdx.x & 7, rest = blockIdx.x >> 3;
        const int nsg = (rows + 7) >> 3;
        const int sg = rest % nsg, slot = rest / nsg;
        g = sg * 8 + x;
        h = slot & 1;
        const int rank = slot >> 1, di = g % L;
        int heavy = 0;  // integer-policy targets of digit di: members 0..L-1 -> slots 0..L without di
        for (int mm = 0; mm < L; ++mm) heavy += !slot_is_f64(T, L, mm >= di ? mm + 1 : mm);
        int want = rank < heavy ? rank : rank - heavy;  // the want-th heavy (rank < heavy) or light member
        jj = 0;
        for (int mm = 0; mm < L; ++mm) {
            const bool hv = !slot_is_f64(T, L, mm >= di ? mm + 1 : mm);
            if (hv == (rank < heavy)) {
                if (want == 0) {
                    jj = mm;
                    break;
                }
                --want;
            }
        }
    } else
        group_decode(blockIdx.x, L, g, jj, h);  // g = digit (b, i); jj = one of its L target moduli
    if (g >= rows || (S.gate_mode == 2 && ks_gated_out(S))) return;  // (mode 1 gates only the launch that writes outputs)
    const int t = threadIdx.x;
    const int bl = g / L, i = g % L;  // bl: item index inside the sub-chunk
    const int b = item0 + bl;
    if (jj >= i) ++jj;  // skip the diagonal; jj == L is the special prime
    const int m = jj < L ? jj : T.k - 1;
    const ModConst mc = T.mods[m];
    const u64 qi = T.mods[i].q;
    const u64 *__restrict__ dd = S.d + ((size_t)b * L + i) * SC::N;
    u64 v[16];
    // [d_i]_m: the U64 policy needs it only when q_i > m; the FP64 policy takes any integer below 2^49 as input,
    // so a digit of a prime < 2^41 needs no reduction at all (the transform is linear and ends canonical)
    const InMode mode = {qi > mc.q, T.modsf[i].q == 0.0, false, 0};
    const int tl = HEFX_EO_LANE(SC, t);
    auto ld = [&](int r, u64 &x, u64 &y) {
        const uint32_t e = eo_nat<SC>(tl, r);
        x = dd[e];
        y = dd[e + SC::H / 2];
    };
    // MAC-operand format: canonical words for the integer-policy moduli, unfinished doubles for the FP64 ones
    const ModConstF mf = T.modsf[m];
    split_fwd<LOGN, KsWaves<LOGN>::NB_FWD, true>(v, ld, mode, lds, ntt_tables(T, m), mc, mf, t, h, tl,
                                                 mac_x_slack(mc, mf, L));
    u64 *__restrict__ xd = S.x + (((size_t)bl * L + i) * (L + 1) + jj) * SC::N + (size_t)h * SC::H;
    // stream_x: the chunk's digit x modulus products exceed the Infinity Cache, so they are written (here) and read
    // (MAC) with streaming accesses that leave the caches to the rows that are reused -- digits, twiddles, key
    if (stream_x & 1) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) nt_store16(xd + C::idx_io(t, r), v[r], v[r + 1]);  // (r, r+1) are one record
    } else {
#pragma unroll
        for (int r = 0; r < 16; r += 2)
            *reinterpret_cast<ulonglong2 *>(xd + C::idx_io(t, r)) = make_ulonglong2(v[r], v[r + 1]);
    }
}

// ------------------------------------------------------------------------------------------------
// (3) acc[b][c][jj] = sum_i x[b][i][jj] * key[i][c][m]  (lazy accumulation over the digits, one reduction at the end)
// A thread owns two adjacent coefficients of TWO consecutive items: when both use the same key (every rotation of
// a batch by the same step, every relinearisation; the host groups a batch's items by key, hefx_capi.cpp ks_run) the
// two key words are loaded once for both -- 4 loads per digit instead of 6 (measured 244 -> 218 us per 192-item chunk,
// ~4.8 TB/s of HBM traffic; sharing across four items at 96 VGPRs was no faster).  Items with different keys take the
// one-item path.
// ------------------------------------------------------------------------------------------------
// MAC arithmetic, one policy per target modulus (uniform per workgroup).  Each accumulates sum_i x_i * k_i for the two
// key polynomials (k0, k1) and two adjacent coefficients (.x, .y).  The kernel waits for HBM either way (its own time
// did not change), but the 128-bit form was a quarter of all VALU instructions of the operation (SQ_INSTS_VALU,
// round 2), issue slots it took from the transform kernels of the neighbouring chunk:
//   MacF  FP64-policy moduli (q < 2^41).  Scratch x holds the UNFINISHED transform value as a double (|x| < 2^45,
//         hefx_ntt.cuh: no canonicalisation in the digit NTTs either); x*k mod q is the exact 6-instruction FP64 modmul
//         with a result in (-0.52q, 0.52q), the L results add exactly, one canonicalisation at the end.
//   MacL  q < 2^60 and L <= 8: 30-bit limbs, x = x1*2^30 + x0, k = k1*2^30 + k0, every partial product < 2^60, so the
//         three column sums (x0k0 | x0k1 + x1k0 | x1k1) take up to 16 terms in plain v_mad_u64_u32 accumulators with no
//         carry handling; the columns are put together into 128 bits once, then one Barrett reduction.
//   MacW  anything wider: full 128-bit accumulators (the generic form).
// All three deliver the canonical residue of the same integer sum: bit-identical results.
struct MacW {
    struct Ctx {
        ModConst mc;
    };
    __device__ static __forceinline__ Ctx make(const ModConst &mc, const ModConstF &) { return Ctx{mc}; }
    typedef ulonglong2 X;
    struct K {
        ulonglong2 k0, k1;
    };
    __device__ static __forceinline__ X xin(const ulonglong2 &bits, bool, const Ctx &) { return bits; }
    __device__ static __forceinline__ K kin(const ulonglong2 &k0, const ulonglong2 &k1, const Ctx &) { return K{k0, k1}; }
    u64 a0xl = 0, a0xh = 0, a0yl = 0, a0yh = 0, a1xl = 0, a1xh = 0, a1yl = 0, a1yh = 0;
    __device__ __forceinline__ void mac(const X &x, const K &k, const Ctx &)
    {
        mac128(a0xl, a0xh, x.x, k.k0.x);
        mac128(a0yl, a0yh, x.y, k.k0.y);
        mac128(a1xl, a1xh, x.x, k.k1.x);
        mac128(a1yl, a1yh, x.y, k.k1.y);
    }
    // this += inner * (dg.x, dg.y): the diagonal product of the double-hoisted transform
    __device__ __forceinline__ void mac_diag(const MacW &in, const ulonglong2 &dg, const Ctx &c)
    {
        ulonglong2 r0, r1;
        in.result(r0, r1, c);
        mac128(a0xl, a0xh, r0.x, dg.x);
        mac128(a0yl, a0yh, r0.y, dg.y);
        mac128(a1xl, a1xh, r1.x, dg.x);
        mac128(a1yl, a1yh, r1.y, dg.y);
    }
    // LT2Q: words below 2q instead of canonical ones -- what the regular MAC leaves in the accumulator scratch: both of
    // its readers take them as they are (the inverse transform of the special-prime rows: first stage on words below 2q;
    // the mod-down epilogue: acc + 4q - f < 6q into a Shoup product), one conditional subtraction less per word
    // what a DATA-prime accumulator row holds in scratch (read by the mod-down epilogue only): integer policies their
    // result<LT2Q>; the FP64 policy its unfinished sums as doubles (MacF::result_data)
    template <bool LT2Q = false>
    __device__ __forceinline__ void result_data(ulonglong2 &r0, ulonglong2 &r1, const Ctx &c) const { result<LT2Q>(r0, r1, c); }
    template <bool LT2Q = false>
    __device__ __forceinline__ void result(ulonglong2 &r0, ulonglong2 &r1, const Ctx &c) const
    {
        r0.x = LT2Q ? barrett128_lt2q(a0xl, a0xh, c.mc) : barrett128(a0xl, a0xh, c.mc);
        r0.y = LT2Q ? barrett128_lt2q(a0yl, a0yh, c.mc) : barrett128(a0yl, a0yh, c.mc);
        r1.x = LT2Q ? barrett128_lt2q(a1xl, a1xh, c.mc) : barrett128(a1xl, a1xh, c.mc);
        r1.y = LT2Q ? barrett128_lt2q(a1yl, a1yh, c.mc) : barrett128(a1yl, a1yh, c.mc);
    }
};

struct MacL {
    typedef MacW::Ctx Ctx;
    __device__ static __forceinline__ Ctx make(const ModConst &mc, const ModConstF &) { return Ctx{mc}; }
    struct X {
        uint32_t xl, xh, yl, yh;
    };
    struct K {
        uint32_t k0xl, k0xh, k0yl, k0yh, k1xl, k1xh, k1yl, k1yh;
    };
    __device__ static __forceinline__ uint32_t lo30(u64 v) { return (uint32_t)v & 0x3FFFFFFFu; }
    __device__ static __forceinline__ uint32_t hi30(u64 v) { return (uint32_t)(v >> 30); }
    __device__ static __forceinline__ X xin(const ulonglong2 &b, bool, const Ctx &)
    {
        return X{lo30(b.x), hi30(b.x), lo30(b.y), hi30(b.y)};
    }
    __device__ static __forceinline__ K kin(const ulonglong2 &k0, const ulonglong2 &k1, const Ctx &)
    {
        return K{lo30(k0.x), hi30(k0.x), lo30(k0.y), hi30(k0.y), lo30(k1.x), hi30(k1.x), lo30(k1.y), hi30(k1.y)};
    }
    u64 c[4][3] = {};  // [a0x, a0y, a1x, a1y][column]
    __device__ static __forceinline__ void mad(u64 (&col)[3], uint32_t xl, uint32_t xh, uint32_t kl, uint32_t kh)
    {
        col[0] += (u64)xl * kl;
        col[1] += (u64)xl * kh;
        col[1] += (u64)xh * kl;
        col[2] += (u64)xh * kh;
    }
    __device__ __forceinline__ void mac(const X &x, const K &k, const Ctx &)
    {
        mad(c[0], x.xl, x.xh, k.k0xl, k.k0xh);
        mad(c[1], x.yl, x.yh, k.k0yl, k.k0yh);
        mad(c[2], x.xl, x.xh, k.k1xl, k.k1xh);
        mad(c[3], x.yl, x.yh, k.k1yl, k.k1yh);
    }
    __device__ __forceinline__ void mac_diag(const MacL &in, const ulonglong2 &dg, const Ctx &cx)
    {
        ulonglong2 r0, r1;
        in.result(r0, r1, cx);
        const uint32_t dxl = lo30(dg.x), dxh = hi30(dg.x), dyl = lo30(dg.y), dyh = hi30(dg.y);
        mad(c[0], lo30(r0.x), hi30(r0.x), dxl, dxh);
        mad(c[1], lo30(r0.y), hi30(r0.y), dyl, dyh);
        mad(c[2], lo30(r1.x), hi30(r1.x), dxl, dxh);
        mad(c[3], lo30(r1.y), hi30(r1.y), dyl, dyh);
    }
    template <bool LT2Q>
    __device__ static __forceinline__ u64 fold(const u64 (&col)[3], const ModConst &mc)
    {
        u64 lo = col[0], hi = 0, t = col[1] << 30;
        lo += t;
        hi += (col[1] >> 34) + (lo < t);
        t = col[2] << 60;
        lo += t;
        hi += (col[2] >> 4) + (lo < t);
        return LT2Q ? barrett128_lt2q(lo, hi, mc) : barrett128(lo, hi, mc);
    }
    template <bool LT2Q = false>
    __device__ __forceinline__ void result_data(ulonglong2 &r0, ulonglong2 &r1, const Ctx &cx) const { result<LT2Q>(r0, r1, cx); }
    template <bool LT2Q = false>
    __device__ __forceinline__ void result(ulonglong2 &r0, ulonglong2 &r1, const Ctx &cx) const
    {
        r0.x = fold<LT2Q>(c[0], cx.mc);
        r0.y = fold<LT2Q>(c[1], cx.mc);
        r1.x = fold<LT2Q>(c[2], cx.mc);
        r1.y = fold<LT2Q>(c[3], cx.mc);
    }
};

struct MacF {
    typedef ArithF64::Ctx Ctx;
    __device__ static __forceinline__ Ctx make(const ModConst &, const ModConstF &mf) { return ArithF64::make(mf); }
    typedef double2 X;
    struct K {
        double k0x, k0y, k1x, k1y;
    };
    // own: the digit's own prime -- canonical words of the source ciphertext instead of scratch doubles
    __device__ static __forceinline__ X xin(const ulonglong2 &b, bool own, const Ctx &)
    {
        return own ? make_double2(ArithF64::from_u64(b.x), ArithF64::from_u64(b.y))
                   : make_double2(__longlong_as_double((long long)b.x), __longlong_as_double((long long)b.y));
    }
    __device__ static __forceinline__ K kin(const ulonglong2 &k0, const ulonglong2 &k1, const Ctx &)
    {
        return K{ArithF64::from_u64(k0.x), ArithF64::from_u64(k0.y), ArithF64::from_u64(k1.x), ArithF64::from_u64(k1.y)};
    }
    double a0x = 0.0, a0y = 0.0, a1x = 0.0, a1y = 0.0;
    __device__ __forceinline__ void mac(const X &x, const K &k, const Ctx &c)
    {
        a0x += ArithF64::mm(x.x, k.k0x, c);
        a0y += ArithF64::mm(x.y, k.k0y, c);
        a1x += ArithF64::mm(x.x, k.k1x, c);
        a1y += ArithF64::mm(x.y, k.k1y, c);
    }
    // |inner sums| <= L * 0.52q < 2^45 (L <= 61): valid left operands of mm as they are
    __device__ __forceinline__ void mac_diag(const MacF &in, const ulonglong2 &dg, const Ctx &c)
    {
        const double dx = ArithF64::from_u64(dg.x), dy = ArithF64::from_u64(dg.y);
        a0x += ArithF64::mm(in.a0x, dx, c);
        a0y += ArithF64::mm(in.a0y, dy, c);
        a1x += ArithF64::mm(in.a1x, dx, c);
        a1y += ArithF64::mm(in.a1y, dy, c);
    }
    // Data-prime rows (round 5): the UNFINISHED sums as doubles -- |a| <= L * 0.52 q < 2^45 -- which the mod-down epilogue
    // subtracts its unfinished transform value from as they are (ArithF64::moddown: |acc - f| < 2^46, a valid left operand of
    // the modmul, hefx_ntt.cuh InvRecentre); no canonicalisation here (eight instructions per word) and no u64 -> f64
    // conversion there (two).  The special prime's row stays canonical (result): the inverse transform reads it.
    template <bool LT2Q = false>
    __device__ __forceinline__ void result_data(ulonglong2 &r0, ulonglong2 &r1, const Ctx &) const
    {
        r0.x = ArithF64::raw(a0x), r0.y = ArithF64::raw(a0y), r1.x = ArithF64::raw(a1x), r1.y = ArithF64::raw(a1y);
    }
    template <bool LT2Q = false>  // (canonical either way)
    __device__ __forceinline__ void result(ulonglong2 &r0, ulonglong2 &r1, const Ctx &c) const
    {
        r0.x = ArithF64::canon(a0x, c);
        r0.y = ArithF64::canon(a0y, c);
        r1.x = ArithF64::canon(a1x, c);
        r1.y = ArithF64::canon(a1y, c);
    }
};

__device__ __forceinline__ bool slot_is_f64(const DevTables &T, int L, int jj)
{
    return T.modsf[jj < L ? jj : T.k - 1].q != 0.0;
}
// the y-th integer-policy target slot among 0..L (block-uniform scalar loop)
__device__ __forceinline__ int nth_int_slot(const DevTables &T, int L, int y)
{
    int jj = 0;
    for (;; ++jj) {
        if (slot_is_f64(T, L, jj)) continue;
        if (y == 0) break;
        --y;
    }
    return jj;
}

// policy of target modulus m in a key switch over L digits (block-uniform); f(policy tag) runs the templated body
template <class F>
__device__ __forceinline__ void mac_dispatch(const DevTables &T, int m, int L, const F &f)
{
    if (T.modsf[m].q != 0.0)
        f(MacF());
    else if (L <= 8 && (T.mods[m].q >> 60) == 0)
        f(MacL());  // (for L <= 5 / L <= 3 its x operands may be below 2q / 4q instead of canonical: mac_x_slack)
    else
        f(MacW());
}

template <bool STREAM>
__device__ __forceinline__ void mac_store(u64 *acc0, u64 *acc1, size_t w, const ulonglong2 &r0, const ulonglong2 &r1)
{
    if (STREAM) {
        nt_store16(acc0 + 2 * w, r0.x, r0.y);
        nt_store16(acc1 + 2 * w, r1.x, r1.y);
    } else {
        gst16(acc0 + 2 * w, r0);
        gst16(acc1 + 2 * w, r1);
    }
}

// NI consecutive items that share `key`.  own[e] / own_elt[e]: item e's input row for digit jj's own prime and its
// Galois element (1: no rotation), used when jj < L; xrow(e, i): digit i's transformed row for this modulus in scratch x;
// acc0 / acc1: the two accumulator rows of item e.
// Shape of the loop (round 3, after reading the ISA): the own-prime term is peeled out in front, so the body over the
// other digits has no conditional loads -- the old form tested i == jj per digit, re-read the item descriptor from
// memory inside that branch and closed every conditional block with s_waitcnt vmcnt(0): 52 full drains and not one
// counted wait in the kernel -- and the operands of digit i+1 are requested before digit i is accumulated.
template <class P, int NI, bool STREAM, class XR>
__device__ __forceinline__ void mac_items(const DevTables &T, const u64 *key, int L, int jj, int m, size_t n, size_t w,
                                          const u64 *const (&own)[NI], const uint32_t (&own_elt)[NI], const XR &xrow,
                                          u64 *const (&acc0)[NI], u64 *const (&acc1)[NI])
{
    const typename P::Ctx cx = P::make(T.mods[m], T.modsf[m]);
    const int logn = T.logn;
    const size_t kpoly = (size_t)T.k * n;
    P A[NI];
    auto load_k = [&](int i, ulonglong2 &k0, ulonglong2 &k1) {
        const u64 *kb = key + ((size_t)i * 2 * T.k + m) * n;
        k0 = gld16(kb + 2 * w);
        k1 = gld16(kb + kpoly + 2 * w);
    };
    auto load_x = [&](int i, ulonglong2 (&xb)[NI]) {
#pragma unroll
        for (int e = 0; e < NI; ++e) {
            const u64 *xp = xrow(e, i) + 2 * w;
            xb[e] = STREAM ? nt_load16(xp) : gld16(xp);  // read exactly once
        }
    };
    if (jj < L) {  // the digit in NTT form modulo its own prime: the (rotated) input row itself, gathered
        ulonglong2 xb[NI], k0, k1;
#pragma unroll
        for (int e = 0; e < NI; ++e) xb[e] = gather_pair(own[e], (uint32_t)w, own_elt[e], logn);
        load_k(jj, k0, k1);
        const typename P::K k = P::kin(k0, k1, cx);
#pragma unroll
        for (int e = 0; e < NI; ++e) A[e].mac(P::xin(xb[e], true, cx), k, cx);
    }
    // the other digits, G at a time (G = 1 measured best once the accesses were global, not FLAT -- profiles/EXPERIMENTS.md): the operands of a whole group are requested up front in straight-line
    // code, so the waits inside the group are COUNTED (digit g is accumulated while digits g+1.. are still in flight).
    // A software-pipelined rolled loop does not get that: the wait-count pass merges the pending loads that cross the
    // loop's back edge into s_waitcnt vmcnt(0).  Four covers the non-own digits of L = 5 in one group.
    constexpr int G = 1;
    auto digit_at = [&](int r) { return r < jj ? r : r + 1; };  // r-th digit other than jj
    const int nd = jj < L ? L - 1 : L;
    int r0 = 0;
    for (; r0 + G <= nd; r0 += G) {
        ulonglong2 xb[G][NI], k0[G], k1[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            load_x(digit_at(r0 + g), xb[g]);
            load_k(digit_at(r0 + g), k0[g], k1[g]);
        }
        HEFX_STAGE_FENCE();  // the scheduler otherwise sinks every load next to its use (one digit in flight at a time)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const typename P::K k = P::kin(k0[g], k1[g], cx);
#pragma unroll
            for (int e = 0; e < NI; ++e) A[e].mac(P::xin(xb[g][e], false, cx), k, cx);
        }
    }
    for (; r0 < nd; ++r0) {  // remainder, one digit at a time
        ulonglong2 xb[NI], k0, k1;
        load_x(digit_at(r0), xb);
        load_k(digit_at(r0), k0, k1);
        const typename P::K k = P::kin(k0, k1, cx);
#pragma unroll
        for (int e = 0; e < NI; ++e) A[e].mac(P::xin(xb[e], false, cx), k, cx);
    }
#pragma unroll
    for (int e = 0; e < NI; ++e) {
        ulonglong2 r0, r1;
        if (jj < L) {  // a data prime's row: what the mod-down epilogue reads (result_data)
#ifdef HEFX_NO_LT2Q
            A[e].result_data(r0, r1, cx);
#else
            A[e].template result_data<true>(r0, r1, cx);
#endif
        } else {  // the special prime's row: the inverse transform's input
#ifdef HEFX_NO_LT2Q
            A[e].result(r0, r1, cx);
#else
            A[e].template result<true>(r0, r1, cx);
#endif
        }
        mac_store<STREAM>(acc0[e], acc1[e], w, r0, r1);
    }
}

// one MAC unit: target slot jj, the item pair (bl0, bl0 + 1) of the (sub-)chunk, pair index w of the row
template <bool STREAM>
__device__ __forceinline__ void mac_unit(const DevTables &T, const KsItem *__restrict__ items, int L, int relin, int item0,
                                         int count, const KsScratch &S, int jj, int bl0, size_t w)
{
    const size_t n = (size_t)1 << T.logn;
    const int m = jj < L ? jj : T.k - 1;
    const bool two = bl0 + 1 < count;
    // both descriptors once, up front (the second one of an odd tail repeats the first: loads stay unconditional)
    const KsItem it0 = items[item0 + bl0], it1 = items[item0 + bl0 + (two ? 1 : 0)];
    const int ownrow = jj < L ? jj : 0;  // jj == L has no own-prime term; the pointer is then unused
    const u64 *const own[2] = {it0.c_in + ((size_t)(relin ? 2 * L : L) + ownrow) * n,
                               it1.c_in + ((size_t)(relin ? 2 * L : L) + ownrow) * n};
    const uint32_t own_elt[2] = {relin ? 1u : item_elt(it0), relin ? 1u : item_elt(it1)};
    auto xrow2 = [&](int e, int i) { return S.x + (((size_t)(bl0 + e) * L + i) * (L + 1) + jj) * n; };
    auto accrow = [&](int bl, int c) { return S.acc + (((size_t)(item0 + bl) * 2 + c) * (L + 1) + jj) * n; };
    u64 *const acc0[2] = {accrow(bl0, 0), accrow(bl0 + 1, 0)}, *const acc1[2] = {accrow(bl0, 1), accrow(bl0 + 1, 1)};
    mac_dispatch(T, m, L, [&](auto pol) {
        using P = decltype(pol);
        if (two && it1.key == it0.key) {
            mac_items<P, 2, STREAM>(T, it0.key, L, jj, m, n, w, own, own_elt, xrow2, acc0, acc1);
        } else {
            const u64 *const o0[1] = {own[0]}, *const o1[1] = {own[1]};
            const uint32_t e0[1] = {own_elt[0]}, e1[1] = {own_elt[1]};
            u64 *const a00[1] = {acc0[0]}, *const a01[1] = {acc1[0]}, *const a10[1] = {acc0[1]}, *const a11[1] = {acc1[1]};
            mac_items<P, 1, STREAM>(T, it0.key, L, jj, m, n, w, o0, e0, [&](int, int i) { return xrow2(0, i); }, a00, a01);
            if (two)
                mac_items<P, 1, STREAM>(T, it1.key, L, jj, m, n, w, o1, e1, [&](int, int i) { return xrow2(1, i); }, a10, a11);
        }
    });
}

template <bool STREAM>
__global__ __launch_bounds__(256) void ks_mac_kernel(DevTables T, const KsItem *__restrict__ items, int L, int relin,
                                                     int item0, int count, int int_only, KsScratch S)
{
    // int_only: the FP64-policy target slots were accumulated by ks_ntt_macf_kernel; blockIdx.y counts the others
    if (S.gate_mode == 2 && ks_gated_out(S)) return;
    const int jj = int_only ? nth_int_slot(T, L, blockIdx.y) : (int)blockIdx.y;
#ifdef HEFX_STAMP
    const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (threadIdx.x == 0 && wg < 1024) hefx_stamp_buf[((size_t)2 * 1024 + wg) * 16] = wall_clock64();
#endif
    mac_unit<STREAM>(T, items, L, relin, item0, count, S, jj, 2 * blockIdx.z, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
#ifdef HEFX_STAMP
    if (HEFX_STAMP > 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && wg < 1024) hefx_stamp_buf[((size_t)2 * 1024 + wg) * 16 + 15] = wall_clock64();
#endif
}

// (2) as a launch
template <int LOGN>
__global__ __launch_bounds__(SplitCfg<LOGN>::T, KsWaves<LOGN>::FWD) void ks_ntt_digits_kernel(DevTables T, int L, int rows,
                                                                             int item0, int stream_x, KsScratch S)
{
    extern __shared__ __align__(16) u64 lds[];
    HEFX_STAMP_KERNEL(1);
    HEFX_STAMP_AT(0);
    ntt_digit_row<LOGN>(T, L, rows, item0, stream_x, S, lds);
    HEFX_STAMP_AT(15);
}

// ------------------------------------------------------------------------------------------------
// gathered MAC of the DOUBLE-hoisted transform (lt2_mac_kernel): NTT_m(a(X^g)) = perm_g(NTT_m(a)), so the digit x modulus
// products x[i][jj] of the UNPERMUTED source are read through the item's gather table.  (The single-hoisted form of rounds
// 1-3 -- this sum alone, a different lift than SEAL's -- gave way to ks_mac_exact_kernel below, which adds the missing term.)
// ------------------------------------------------------------------------------------------------
// sum over the digits of one rotation `it`, gathered through its table: the inner loop of both hoisted forms
template <class P>
__device__ __forceinline__ void mac_gathered(P &A, const typename P::Ctx &cx, const DevTables &T, const KsItem &it,
                                             const u64 *__restrict__ c1, const KsScratch &S, int L, int jj, int m,
                                             size_t n, size_t w)
{
    const uint2 pi = gld_u32x2(it.perm + 2 * w);
    for (int i = 0; i < L; ++i) {
        const u64 *__restrict__ xrow = i == jj ? c1 + (size_t)i * n : S.x + ((size_t)i * (L + 1) + jj) * n;
        ulonglong2 xb;
        xb.x = gld8(xrow + pi.x);
        xb.y = gld8(xrow + pi.y);
        const u64 *kbase = it.key + ((size_t)i * 2 * T.k + m) * n;
        const ulonglong2 k0 = gld16(kbase + 2 * w);
        const ulonglong2 k1 = gld16(kbase + (size_t)T.k * n + 2 * w);
        A.mac(P::xin(xb, i == jj, cx), P::kin(k0, k1, cx), cx);
    }
}

// ------------------------------------------------------------------------------------------------
// (3x) EXACT hoisting (round 4): the hoisted form with SEAL's bits.  SEAL decomposes the ROTATED polynomial: digit i of
// item b is t = sigma_g(T_i) with T_i = INTT_i(c1_i) in [0, q_i) and sigma_g the signed coefficient permutation taken to
// canonical residues -- a negated non-zero coefficient a becomes q_i - a.  As integers
//     t = sigma_g^Z(T_i) + q_i F_g        (sigma_g^Z: the signed permutation over Z, F_g: 1 where X -> X^g negates),
// PROVIDED no coefficient of T_i is zero (a negated zero stays 0, not q_i).  Modulo a key modulus m the first term is the
// automorphism of T_i mod m, whose transform is the evaluation-domain gather of x[i][m] = NTT_m(T_i mod m); so
//     NTT_m(t mod m) = perm_g(x[i][m]) + (q_i mod m) W_g[m],      W_g[m] = NTT_m(F_g)   (one [k][N] table per element)
// and the key MAC of the item is
//     acc[c][m] = sum_i perm_g(x[i][m]) key[i][c][m]  +  W_g[m] * sum_(i != m) (q_i mod m) key[i][c][m]     (mod m)
// -- the hoisted gather plus a second accumulation over the SAME key words with scalar operands, multiplied by one table
// row.  Everything downstream of the MAC only sees acc mod m, so the outputs are SEAL's words.  The zero proviso is
// checked where T_i is produced (intt_digits_body stores the chunk's gate tag); a chunk whose gate was hit -- a source
// with a zero c1 coefficient: probability ~ N L / q per source for a real ciphertext, certain for a transparent one -- is
// redone item by item by the ordinary kernels, which otherwise exit at once (gate_mode 2).
// The sources' rows x are computed once per chunk by (1) and (2) over the source list; per item only this MAC and the
// mod-down remain: (L+1)(L+2) -> 2 + 2L transforms, key words read once.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ks_mac_exact_kernel(DevTables T, const KsItem *__restrict__ items, int L, KsScratch S)
{
    const size_t n = (size_t)1 << T.logn;
    const int jj = blockIdx.y, b = blockIdx.z;
    const int m = jj < L ? jj : T.k - 1;
    const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // pair index within the row
    const KsItem it = items[b];
    // (diagnostic, hefx_ks_fallback_count: one thread of the launch counts a chunk whose gate was hit -- the fallback redoes it)
    if (w == 0 && jj == 0 && b == 0 && *S.gate == S.gate_tag) atomicAdd(S.gate_hits, 1u);
    const u64 *__restrict__ c1 = it.c_in + (size_t)L * n;  // the source's c1: digit i in NTT form mod its own prime
    const u64 *__restrict__ xs = S.x + (size_t)it.dsrc * L * (L + 1) * n;
    mac_dispatch(T, m, L, [&](auto pol) {
        using P = decltype(pol);
        const typename P::Ctx cx = P::make(T.mods[m], T.modsf[m]);
        P A, B;
        // positions 2w, 2w+1 gather an aligned pair in either order (galois_index): one 16-byte load and a swap
        const uint32_t p0 = gld_u32x2(it.perm + 2 * w).x;
        const ulonglong2 wv = gld16(it.flipw + (size_t)m * n + 2 * w);
        auto load = [&](int i, const u64 *__restrict__ xrow, ulonglong2 &xb, ulonglong2 &k0, ulonglong2 &k1) {
            xb = gld16(xrow + (p0 & ~1u));
            const u64 *kbase = it.key + ((size_t)i * 2 * T.k + m) * n;
            k0 = gld16(kbase + 2 * w);
            k1 = gld16(kbase + (size_t)T.k * n + 2 * w);
        };
        auto swapped = [&](const ulonglong2 &v) { return (p0 & 1u) ? make_ulonglong2(v.y, v.x) : v; };
        if (jj < L) {  // the digit's own prime: the source's c1 row itself, no correction term (q_i mod q_i = 0)
            ulonglong2 xb, k0, k1;
            load(jj, c1 + (size_t)jj * n, xb, k0, k1);
            A.mac(P::xin(swapped(xb), true, cx), P::kin(k0, k1, cx), cx);
        }
        // the other digits (operands of G digits requested up front: 1 and 2 measure the same on the direct-key transform,
        // 2.15 ms at d = 512, and 4 loses a quarter -- the kernel streams the keys at 4.2 TB/s either way)
        constexpr int G = 1;
        auto digit_at = [&](int r) { return r < jj ? r : r + 1; };
        auto term = [&](int i, const ulonglong2 &xb, const ulonglong2 &k0, const ulonglong2 &k1) {
            const typename P::K k = P::kin(k0, k1, cx);
            A.mac(P::xin(swapped(xb), false, cx), k, cx);
            const u64 c = S.qmod[(size_t)i * T.k + m];  // q_i mod m (uniform)
            B.mac(P::xin(make_ulonglong2(c, c), true, cx), k, cx);
        };
        const int nd = jj < L ? L - 1 : L;
        int r = 0;
        for (; r + G <= nd; r += G) {
            ulonglong2 xb[G], k0[G], k1[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int i = digit_at(r + g);
                load(i, xs + ((size_t)i * (L + 1) + jj) * n, xb[g], k0[g], k1[g]);
            }
            HEFX_STAGE_FENCE();
#pragma unroll
            for (int g = 0; g < G; ++g) term(digit_at(r + g), xb[g], k0[g], k1[g]);
        }
        for (; r < nd; ++r) {
            const int i = digit_at(r);
            ulonglong2 xb, k0, k1;
            load(i, xs + ((size_t)i * (L + 1) + jj) * n, xb, k0, k1);
            term(i, xb, k0, k1);
        }
        ulonglong2 r0, r1, f0, f1;
        P C;
        C.mac_diag(B, wv, cx);
        if constexpr (std::is_same<P, MacF>::value) {
            if (jj < L) {  // a data prime's FP64 row: the two unfinished sums added as doubles (MacF::result_data)
                P S2 = A;
                S2.a0x += C.a0x, S2.a0y += C.a0y, S2.a1x += C.a1x, S2.a1y += C.a1y;
                S2.result_data(r0, r1, cx);
            } else {
                A.result(r0, r1, cx);
                C.result(f0, f1, cx);
                const u64 q = T.mods[m].q;
                r0.x = addmod(r0.x, f0.x, q), r0.y = addmod(r0.y, f0.y, q);
                r1.x = addmod(r1.x, f1.x, q), r1.y = addmod(r1.y, f1.y, q);
            }
        } else {
            A.result(r0, r1, cx);
            C.result(f0, f1, cx);
            const u64 q = T.mods[m].q;
            r0.x = addmod(r0.x, f0.x, q);
            r0.y = addmod(r0.y, f0.y, q);
            r1.x = addmod(r1.x, f1.x, q);
            r1.y = addmod(r1.y, f1.y, q);
        }
        mac_store<false>(S.acc + (((size_t)b * 2 + 0) * (L + 1) + jj) * n, S.acc + (((size_t)b * 2 + 1) * (L + 1) + jj) * n,
                         w, r0, r1);
    });
}

// F_g in coefficient order, once per modulus row: rows[e][m][c] = 1 when X -> X^g negates the coefficient that lands at c,
// i.e. when c g^-1 mod 2N >= N (coefficient c of a(X^g) is +-a_s with s g = c or c + N mod 2N); transformed by the caller
__global__ __launch_bounds__(256) void flip_mask_kernel(DevTables T, const uint32_t *__restrict__ ginv, u64 *__restrict__ rows)
{
    const uint32_t n = 1u << T.logn;
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t e = blockIdx.z, m = blockIdx.y;
    rows[((size_t)e * T.k + m) * n + c] = ((c * ginv[e]) & (2 * n - 1)) >> T.logn;
}

// ------------------------------------------------------------------------------------------------
// DOUBLE HOISTING (second fast mode of the linear transform): besides sharing the digit
// decomposition, the d-1 products diag_l * rot_l(ct_new) are SUMMED IN THE EXTENDED BASIS (q_0..q_(L-1), P) and modded
// down once -- per rotation only a gathered MAC remains, no transform at all:
//   S[c][jj]  = sum_l diag_l[jj] * ( sum_i x[i][jj][perm_l] * key_l[i][c][jj] )        (all L+1 moduli)
//   C0[j]     = sum_l diag_l[j]  * c0[j][perm_l]                                        (data primes)
//   out       = (diag_0*c0 + C0, diag_0*c1) + moddown(S)
// mod-down rounds once instead of d-1 times, so the result differs from the rotation-by-rotation sum in the last
// bits of noise (like any BSGS / double-hoisted evaluation); it is bit-exact against the oracle's statement of this
// algorithm.  The diagonals must be encoded over the special prime too (key-level plaintexts); top data level only.
// Workgroups own a chunk of LT2_CHUNK rotations and write partial sums that a wide add_many reduces.
// ------------------------------------------------------------------------------------------------
constexpr int LT2_CHUNK = 8;  // <= 8: the limb policy's middle column takes two terms per rotation, sixteen in all

__global__ __launch_bounds__(256) void lt2_mac_kernel(DevTables T, const KsItem *__restrict__ items, int L, int nrot,
                                                      KsScratch S, const u64 *__restrict__ src_c1 /* [L][N] */,
                                                      u64 *__restrict__ partial /* [chunks][2][L+1][N] */)
{
    const size_t n = (size_t)1 << T.logn;
    const int jj = blockIdx.y, ch = blockIdx.z;
    const int m = jj < L ? jj : T.k - 1;
    const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int l0 = ch * LT2_CHUNK, l1 = l0 + LT2_CHUNK < nrot ? l0 + LT2_CHUNK : nrot;
    mac_dispatch(T, m, L, [&](auto pol) {
        using P = decltype(pol);
        const typename P::Ctx cx = P::make(T.mods[m], T.modsf[m]);
        P tot;
        for (int l = l0; l < l1; ++l) {
            const KsItem it = items[l];
            P A;
            mac_gathered(A, cx, T, it, src_c1, S, L, jj, m, n, w);
            tot.mac_diag(A, gld16(it.pt + (size_t)m * n + 2 * w), cx);  // key-level plaintext row m
        }
        ulonglong2 r0, r1;
        tot.result(r0, r1, cx);
        u64 *base = partial + (size_t)ch * 2 * (L + 1) * n;
        mac_store<false>(base + (size_t)jj * n, base + ((size_t)(L + 1) + jj) * n, w, r0, r1);
    });
}

// partial C0[ch][j] = sum_{l in chunk} diag_l[j] * c0[j][perm_l]
__global__ __launch_bounds__(256) void lt2_c0_kernel(DevTables T, const KsItem *__restrict__ items, int L, int nrot,
                                                     const u64 *__restrict__ c0, u64 *__restrict__ partial /* [chunks][L][N] */)
{
    const size_t n = (size_t)1 << T.logn;
    const int j = blockIdx.y, ch = blockIdx.z;
    const ModConst mc = T.mods[j];
    const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u64 xl = 0, xh = 0, yl = 0, yh = 0;
    const int l0 = ch * LT2_CHUNK, l1 = l0 + LT2_CHUNK < nrot ? l0 + LT2_CHUNK : nrot;
    const u64 *__restrict__ row = c0 + (size_t)j * n;
    for (int l = l0; l < l1; ++l) {
        const KsItem it = items[l];
        const uint2 pi = gld_u32x2(it.perm + 2 * w);
        const ulonglong2 dg = gld16(it.pt + (size_t)j * n + 2 * w);
        mac128(xl, xh, gld8(row + pi.x), dg.x);
        mac128(yl, yh, gld8(row + pi.y), dg.y);
    }
    reinterpret_cast<ulonglong2 *>(partial + ((size_t)ch * L + j) * n)[w] =
        make_ulonglong2(barrett128(xl, xh, mc), barrett128(yl, yh, mc));
}

// ------------------------------------------------------------------------------------------------
// (2+3 fused, hybrid) -- the digit x modulus products of the FP64-policy target moduli are never stored.
// One launch, two kinds of workgroup (N/16 threads, eight coefficients per thread, hefx_ntt8.cuh):
//   role F  (item b, FP64 target slot jj, half h): walks the L digits, transforms each one to modulus m in the FP64
//           policy and multiplies the UNFINISHED transform value straight into two FP64 accumulators per coefficient
//           (MacF arithmetic: the exact six-instruction modmul, |term| < 0.52 q, the L terms add exactly) that live in
//           registers for the whole loop -- 32 VGPRs, where the 128-bit accumulators of a 60-bit target need 64 and
//           spilled (round 2).  acc[b][c][jj] = canon(sum_i NTT_m([d_i]_m) * key[i][c][m]), the same integer sum as
//           ks_ntt_digits + ks_mac, hence the same bits.
//   role I  (item b, digit i, integer-policy target slot jj != i, half h): the plain digit transform, canonical words
//           to scratch x -- only these rows still travel (9 of 25 at C3); ks_mac_kernel then runs over the integer
//           target slots alone (int_only).
// All workgroups of an item share an XCD (the L digit rows they read are then served by that XCD's L2); the long
// role-F workgroups of an item come first.  x traffic per op at C3: 25 rows written + 25 read -> 9 + 9.
// ------------------------------------------------------------------------------------------------
template <int LOGN>
struct FusedCfg {
    using C = Ntt8Cfg<LOGN - 1>;
    static constexpr int N = 1 << LOGN;
    static constexpr int H = N / 2;
    static constexpr int T = C::T;  // N/16 threads
    static constexpr size_t LDS_BYTES = sizeof(u64) * C::LDS_WORDS;
};
template <int LOGN>
__global__ __launch_bounds__(FusedCfg<LOGN>::T, 4) void ks_ntt_macf_kernel(DevTables T, const KsItem *__restrict__ items, int L,
                                                                         int relin, int groups, int nf, int per_item,
                                                                         int stream_x, KsScratch S)
{
    using FC = FusedCfg<LOGN>;
    using C = typename FC::C;
    extern __shared__ __align__(16) u64 lds[];
    const int xq = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int slot = rest % per_item, b = (rest / per_item) * 8 + xq;
    if (b >= groups) return;
    const int t = threadIdx.x;
    const int h = slot & 1, role = slot >> 1;
    const size_t off = (size_t)h * FC::H;
    static_assert(C::T % 2 == 0, "EO loader assumes an even thread count");
    // coefficient j = idx_nat(t, r) = t + T r of a row stored [evens | odds]: parity is the thread's
    const uint32_t e0 = (uint32_t)((t & 1) * FC::H + (t >> 1));
    if (role < nf) {
        // ---- role F ----
        int jj = 0;
        for (int cnt = 0;; ++jj) {
            if (!slot_is_f64(T, L, jj)) continue;
            if (cnt == role) break;
            ++cnt;
        }
        const int m = jj < L ? jj : T.k - 1;
        const ModConst mc = T.mods[m];
        const ArithF64::Ctx cx = ArithF64::make(T.modsf[m]);
        const double *__restrict__ tw = T.twf + ((size_t)m << LOGN);
        const KsItem it = items[b];
        double a0[8], a1[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) a0[r] = a1[r] = 0.0;
        for (int i = 0; i < L; ++i) {
            double f[8];
            if (i == jj) {  // the digit in NTT form modulo its own prime: the (rotated) input row itself
                const u64 *__restrict__ xr = it.c_in + ((size_t)(relin ? 2 * L : L) + i) * FC::N;
                const uint32_t elt = relin ? 1u : item_elt(it);
#pragma unroll
                for (int r = 0; r < 8; r += 2) {
                    const uint32_t rec = (uint32_t)((off + C::idx_out(t, r)) >> 1);
                    const ulonglong2 v = gather_pair(xr, rec, elt, LOGN);
                    f[r] = ArithF64::from_u64(v.x);
                    f[r + 1] = ArithF64::from_u64(v.y);
                }
            } else {
                const u64 *__restrict__ dd = S.d + ((size_t)b * L + i) * FC::N;
                const InMode mode = {false, T.modsf[i].q == 0.0, false, 0};
                auto ld = [&](int r, u64 &x, u64 &y) {
                    const uint32_t e = e0 + (uint32_t)(C::T / 2) * (uint32_t)r;
                    x = dd[e];
                    y = dd[e + FC::H / 2];
                };
                split8_fwd_raw<LOGN, ArithF64>(f, ld, mode, mc, lds, tw, cx, t, h);
            }
            const u64 *__restrict__ k0 = it.key + ((size_t)i * 2 * T.k + m) * FC::N + off;
            const u64 *__restrict__ k1 = k0 + (size_t)T.k * FC::N;
            HEFX_STAGE_FENCE();  // keeps the key loads (32 VGPRs) from being hoisted above the transform
            ulonglong2 kv0[4], kv1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                kv0[r] = gld16(k0 + C::idx_out(t, 2 * r));
                kv1[r] = gld16(k1 + C::idx_out(t, 2 * r));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a0[2 * r] += ArithF64::mm(f[2 * r], ArithF64::from_u64(kv0[r].x), cx);
                a0[2 * r + 1] += ArithF64::mm(f[2 * r + 1], ArithF64::from_u64(kv0[r].y), cx);
                a1[2 * r] += ArithF64::mm(f[2 * r], ArithF64::from_u64(kv1[r].x), cx);
                a1[2 * r + 1] += ArithF64::mm(f[2 * r + 1], ArithF64::from_u64(kv1[r].y), cx);
            }
            // the next digit's first exchange reuses the LDS words this transform's last pass read
            if (i != jj) __syncthreads();
        }
        u64 *__restrict__ o0 = S.acc + (((size_t)b * 2 + 0) * (L + 1) + jj) * FC::N + off;
        u64 *__restrict__ o1 = S.acc + (((size_t)b * 2 + 1) * (L + 1) + jj) * FC::N + off;
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            // (data primes: unfinished doubles, MacF::result_data; the special prime: canonical)
            *reinterpret_cast<ulonglong2 *>(o0 + C::idx_out(t, r)) =
                jj < L ? make_ulonglong2(ArithF64::raw(a0[r]), ArithF64::raw(a0[r + 1]))
                       : make_ulonglong2(ArithF64::canon(a0[r], cx), ArithF64::canon(a0[r + 1], cx));
            *reinterpret_cast<ulonglong2 *>(o1 + C::idx_out(t, r)) =
                jj < L ? make_ulonglong2(ArithF64::raw(a1[r]), ArithF64::raw(a1[r + 1]))
                       : make_ulonglong2(ArithF64::canon(a1[r], cx), ArithF64::canon(a1[r + 1], cx));
        }
        return;
    }
    // ---- role I ----
    int rr = role - nf, jj = 0;
    for (;; ++jj) {
        if (slot_is_f64(T, L, jj)) continue;
        const int nd = jj < L ? L - 1 : L;  // digits that are transformed to this slot
        if (rr < nd) break;
        rr -= nd;
    }
    int i = rr;
    if (jj < L && i >= jj) ++i;
    const int m = jj < L ? jj : T.k - 1;
    const ModConst mc = T.mods[m];
    const u64 *__restrict__ dd = S.d + ((size_t)b * L + i) * FC::N;
    const InMode mode = {T.mods[i].q > mc.q, false, false, 0};
    auto ld = [&](int r, u64 &x, u64 &y) {
        const uint32_t e = e0 + (uint32_t)(C::T / 2) * (uint32_t)r;
        x = dd[e];
        y = dd[e + FC::H / 2];
    };
    u64 v[8];
    fwd_int_dispatch(mc, [&](auto pol) {
        using A = decltype(pol);
        const typename A::Ctx cx = A::make(mc);
        split8_fwd_raw<LOGN, A>(v, ld, mode, mc, lds, T.tw + ((size_t)m << LOGN), cx, t, h);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = A::fwd_finish(v[r], cx);
    });
    u64 *__restrict__ xd = S.x + (((size_t)b * L + i) * (L + 1) + jj) * FC::N + off;
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        const u64 w0 = v[r], w1 = v[r + 1];
        u64 *p = xd + C::idx_out(t, r);
        if (stream_x) {
            nt_store16(p, w0, w1);
        } else {
            *reinterpret_cast<ulonglong2 *>(p) = make_ulonglong2(w0, w1);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// (4) u[b][c] (EO) = (INTT_P(acc[b][c][P]) + floor(P/2)) mod P
// ------------------------------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__(SplitCfg<LOGN>::T, KsWaves<LOGN>::INV) void ks_moddown_intt_kernel(DevTables T, int L, int rows,
                                                                               KsScratch S)
{
    using SC = SplitCfg<LOGN>;
    using C = typename SC::C;
    extern __shared__ __align__(16) u64 lds[];
    HEFX_STAMP_KERNEL(3);
    HEFX_STAMP_AT(0);
    int p, h;
    split_decode(blockIdx.x, p, h);
    if (p >= rows || (S.gate_mode == 2 && ks_gated_out(S))) return;
    const int t = threadIdx.x;
    const int sp = T.k - 1;
    const ModConst mc = T.mods[sp];
    const ulonglong2 *__restrict__ src =
        reinterpret_cast<const ulonglong2 *>(S.acc + ((size_t)p * (L + 1) + L) * SC::N);  // p = b*2 + c
    u64 v[16];
    split_inv<LOGN, KsWaves<LOGN>::NB_INV>(v, src, lds, ntt_tables(T, sp), mc, T.modsf[sp], t, h);
    const u64 half = mc.q >> 1;
    u64 *__restrict__ ud = S.u + (size_t)p * SC::N + (size_t)h * SC::H;
#pragma unroll
    for (int r = 0; r < 16; ++r) ud[C::idx_nat(t, r)] = csub(v[r] + half, mc.q);
    HEFX_STAMP_AT(15);
}

// ------------------------------------------------------------------------------------------------
// (5) out[b][c][j] = (acc[b][c][j] - NTT_j((u mod q_j) - (P/2 mod q_j))) * P^-1 + add-in, optionally * pt.
// The epilogue runs in the row's arithmetic policy directly on the unfinished transform values.
// ------------------------------------------------------------------------------------------------
// P^-1 mod q_j in the row's policy.  Fetched ONCE per workgroup (md_pinv) and handed to the epilogue by value: read
// from the table inside the per-record epilogue it was re-loaded after every store to the output (the compiler cannot
// prove the table and the output distinct), each load followed by s_waitcnt vmcnt(0) -- which also waited for the
// stores just issued (found in the ISA, round 3; this is where much of the kernel's parked wave time went).
struct PinvU {
    ulonglong2 v;
};
struct PinvF {
    double2 v;
};
template <bool L16>
__device__ __forceinline__ PinvU md_pinv(ArithU64T<L16>, const DevTables &T, int sp, int j)
{
    return PinvU{T.invmod[(size_t)sp * T.k + j]};
}
__device__ __forceinline__ PinvF md_pinv(ArithF64, const DevTables &T, int sp, int j)
{
    return PinvF{T.invmodf[(size_t)sp * T.k + j]};
}
template <bool L16>
__device__ __forceinline__ u64 md_epilogue(ArithU64T<L16>, u64 f, u64 acc, u64 sadd, u64 pt, bool has_pt,
                                           const typename ArithU64T<L16>::Ctx &cx, const PinvU &pinv, const ModConst &mc)
{
    return ArithU64T<L16>::moddown(f, acc, sadd, pt, has_pt, cx, pinv.v, mc);
}
__device__ __forceinline__ u64 md_epilogue(ArithF64, double f, u64 acc, u64 sadd, u64 pt, bool has_pt,
                                           const ArithF64::Ctx &cx, const PinvF &pinv, const ModConst &)
{
    return ArithF64::moddown(f, acc, sadd, pt, has_pt, cx, pinv.v);
}

template <int LOGN, class A>
__device__ __forceinline__ void moddown_finish_body(const DevTables &T, const KsItem &it, int L, int relin,
                                                    const KsScratch &S, u64 *lds,
                                                    const typename A::TW *__restrict__ tw, const typename A::Ctx &cx,
                                                    const ModConst &mc, int b, int c, int j, int t, int h)
{
    using SC = SplitCfg<LOGN>;
    using C = typename SC::C;
    const int sp = T.k - 1;
    const u64 q = mc.q;
    const u64 half_j = T.halfmod[(size_t)sp * T.k + j];
    const auto pinv = md_pinv(A{}, T, sp, j);
    const u64 *__restrict__ ud = S.u + ((size_t)b * 2 + c) * SC::N;
    // u < P is reduced modulo q_j and (P/2 mod q_j) subtracted in the row's policy (exact for any 64-bit word)
    const InMode mode = {true, true, true, half_j, true, T.mods[sp].q < 2 * mc.q};
    const int tl = HEFX_EO_LANE(SC, t);
    auto ld = [&](int r, u64 &x, u64 &y) {
        const uint32_t e = eo_nat<SC>(tl, r);
        x = ud[e];
        y = ud[e + SC::H / 2];
    };
    // relinearisation adds (c0,c1) of the input; a rotation adds perm(c0), gathered from the source's c0
    const size_t off = (size_t)h * SC::H;
    const u64 *__restrict__ acc = S.acc + (((size_t)b * 2 + c) * (L + 1) + j) * SC::N + off;
    const u64 *__restrict__ addrow = it.c_in + ((size_t)(relin ? c : 0) * L + j) * SC::N;
    const bool has_add = relin || c == 0;
    const uint32_t elt = relin ? 1u : item_elt(it);
    const u64 *__restrict__ pt = it.pt ? it.pt + (size_t)j * SC::N + off : nullptr;
    u64 *__restrict__ dst = it.c_out + ((size_t)c * L + j) * SC::N + off;
    typename A::V f[16];
    u64 keep[16];  // the outputs, for the optional accumulate behind the epilogue (dead otherwise)
    {
        // Operands are fetched in groups of four of the thread's sixteen coefficients; in the 256-VGPR builds (N <= 8192),
        // where a second buffer is free, the next group is requested before the current one is computed and stored.
        // (Groups of 2 / 8 / 16, pipelining everywhere or nowhere, separate settings for the FP64 rows: all measured,
        // all within -1 .. -8 % of this -- profiles/EXPERIMENTS.md.)
        // Every ring size ends its forward transform in the pair layout idx_io (two adjacent words per lane, lanes
        // adjacent): 16-byte operand loads and stores, 1 KiB contiguous per instruction.
        constexpr int GS = 4, NG = 16 / GS;
        constexpr bool PIPE = KsWaves<LOGN>::FWD <= 2;
        constexpr int NBUF = PIPE ? 2 : 1;
        // registers (r, r+1), r even, are one record: two adjacent words at an even index (idx_io), so the rotated c0
        // is fetched as one gathered 16-byte pair per record -- half the index arithmetic and loads of a per-word gather
        static_assert(GS % 2 == 0, "record layout");
        split_fwd_raw<LOGN, A, decltype(ld), KsWaves<LOGN>::NB_FWD, NoHook, true>(f, ld, mode, mc, lds, tw, cx, t, h, NoHook(), tl);
        // The epilogue is SPECIALISED on (add-in?, plaintext?) by one workgroup-uniform branch around it (round 3): with
        // the two conditions tested per record inside the loop the loads sat in conditional blocks, and the compiler's
        // wait-count insertion closed every such block with s_waitcnt vmcnt(0) -- a full drain, stores included, per
        // record.  Straight-line bodies get counted waits.  The gather handles the identity too (galois_index(i, 1) = i),
        // so "rotated or not" needs no branch either.
        auto epilogue = [&](auto has_add_c, auto has_pt_c) {
            constexpr bool HA = decltype(has_add_c)::value, HP = decltype(has_pt_c)::value;
            u64 a[NBUF][GS], sadd[NBUF][GS], pp[NBUF][GS];
            auto fetch = [&](int g, int bufi) {
#pragma unroll
                for (int r = 0; r < GS; r += 2) {
                    const int idx = C::idx_io(t, GS * g + r);
                    const ulonglong2 av = gld16(acc + idx);
                    ulonglong2 sv = make_ulonglong2(0, 0), pv = make_ulonglong2(0, 0);
                    if constexpr (HA) sv = gather_pair(addrow, (uint32_t)((off + idx) >> 1), elt, LOGN);
                    if constexpr (HP) pv = gld16(pt + idx);
                    a[bufi][r] = av.x, a[bufi][r + 1] = av.y;
                    sadd[bufi][r] = sv.x, sadd[bufi][r + 1] = sv.y;
                    pp[bufi][r] = pv.x, pp[bufi][r + 1] = pv.y;
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int cur = PIPE ? (g & 1) : 0;
                if (PIPE && g + 1 < NG) fetch(g + 1, cur ^ 1);
#pragma unroll
                for (int r = 0; r < GS; r += 2) {  // one 16-byte store per record
                    ulonglong2 o;
                    o.x = md_epilogue(A{}, f[GS * g + r], a[cur][r], sadd[cur][r], pp[cur][r], HP, cx, pinv, mc);
                    o.y = md_epilogue(A{}, f[GS * g + r + 1], a[cur][r + 1], sadd[cur][r + 1], pp[cur][r + 1], HP, cx, pinv, mc);
                    gst16(dst + C::idx_io(t, GS * g + r), o);
                    if constexpr (!HP) keep[GS * g + r] = o.x, keep[GS * g + r + 1] = o.y;
                }
                if (!PIPE && g + 1 < NG) fetch(g + 1, 0);
            }
        };
        if (has_add) {
            if (pt)
                epilogue(std::true_type{}, std::true_type{});
            else
                epilogue(std::true_type{}, std::false_type{});
        } else {
            if (pt)
                epilogue(std::false_type{}, std::true_type{});
            else
                epilogue(std::false_type{}, std::false_type{});
        }
        // accumulate (hefx_apply_galois_add_batch; never together with a fused plaintext product): acc_out = acc_in + out,
        // one workgroup-uniform branch behind the epilogue, all loads of the thread requested before the first is used
        if (it.acc_out && !pt) {
            const size_t arow = ((size_t)c * L + j) * SC::N + off;
            const u64 *__restrict__ ain = it.acc_in + arow;
            u64 *__restrict__ aout = it.acc_out + arow;
            ulonglong2 av[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) av[r / 2] = gld16(ain + C::idx_io(t, r));
#pragma unroll
            for (int r = 0; r < 16; r += 2)
                gst16(aout + C::idx_io(t, r), make_ulonglong2(addmod(av[r / 2].x, keep[r], q), addmod(av[r / 2].y, keep[r + 1], q)));
        }
    }
}

template <int LOGN>
__global__ __launch_bounds__(SplitCfg<LOGN>::T, KsWaves<LOGN>::FWD) void ks_moddown_finish_kernel(DevTables T,
                                                                                 const KsItem *__restrict__ items, int L,
                                                                                 int relin, int rows, KsScratch S)
{
    extern __shared__ __align__(16) u64 lds[];
    HEFX_STAMP_KERNEL(4);
    HEFX_STAMP_AT(0);
    int g, j, h;
    group_decode(blockIdx.x, L, g, j, h);  // g = remainder polynomial (b, c); j = one of the L data primes
    if (g >= rows || ks_gated_out(S)) return;
    const int t = threadIdx.x;
    const int b = g >> 1, c = g & 1;
    const ModConst mc = T.mods[j];
    const ModConstF mf = T.modsf[j];
    const NttTables nt = ntt_tables(T, j);
    const KsItem it = items[b];
    if (mf.q != 0.0)
        moddown_finish_body<LOGN, ArithF64>(T, it, L, relin, S, lds, nt.twf, ArithF64::make(mf), mc, b, c, j, t, h);
    else
        // (the 16q butterfly here: no spill from N = 16384 on, and no gain -- 266 -> 269 us per chunk at C3,
        // profiles/r03/ab_finish_l16.txt; one of five rows per polynomial is integer-policy)
        moddown_finish_body<LOGN, ArithU64>(T, it, L, relin, S, lds, nt.tw, ArithU64::make(mc), mc, b, c, j, t, h);
    HEFX_STAMP_AT(15);
}

// ------------------------------------------------------------------------------------------------
// SMALL-BATCH path: quarter rows (hefx_ntt8.cuh) -- the same five launches with a row handled by four workgroups of
// N/32 threads, eight coefficients per thread.  For chunks that cannot fill the chip with split-2 workgroups (a lone
// rotation of a NAF chain, the eight lockstep chains of the LR gradient, logistic_regression_ckks.cpp:295-300 ->
// helper.h:472-476): there the caller waits for one launch after the other -- per launch ~1 us of table fetches, ~2 us
// until the row has arrived, four radix-8 passes of ~1.1 us on a 60-bit row (tools/stamp_timeline.py).  The scratch
// arrays have one layout, so ks_run decides PER LAUNCH between these kernels and the split-2 ones (KS_Q_* mask): quarter
// rows only while their grid still gets a CU per workgroup.  Same integers, same bits.
// ------------------------------------------------------------------------------------------------
template <int LOGN>
struct QuarterCfg {
    using C = Ntt8Cfg<LOGN - 2>;
    static constexpr int N = 1 << LOGN;
    static constexpr int Q = N / 4;   // points per workgroup
    static constexpr int T = C::T;    // threads per workgroup = N/32
    static constexpr size_t LDS_BYTES = sizeof(u64) * C::LDS_WORDS;
};
// rows -> grid: the four quarters of row p sit at block ids that differ by 8 (same XCD under round-robin dispatch)
__host__ __device__ static inline int quarter_grid(int rows) { return ((rows + 7) / 8) * 32; }
__device__ static __forceinline__ void quarter_decode(int bid, int &p, int &part)
{
    part = 3 - ((bid >> 3) & 3);  // the inverse kernels' heaviest quarter (two twiddled stages) is dispatched first
    p = ((bid >> 5) << 3) | (bid & 7);
}
// (8, fan, groups) grids of the quarter / pair kernels: the `fan` workgroups of group g share XCD g % 8 (they read one
// source row; dispatch is x-fastest).  DENSE form (a 1-D grid of groups * fan blocks; small launches whose group count is
// not a multiple of 8, round 6): block w = g * fan + slot, consecutive blocks on consecutive XCDs.  Two digits-per-item
// example: n = 2, L = 5 on the pair path is 10 groups x 20 workgroups -- grouped, XCDs 0 and 1 get 40 workgroups for their
// 32 CUs (one per CU at 190 VGPRs: a second round) and the other six 20; dense, every XCD gets 25 (55 -> 44 us per call).
__host__ static inline dim3 fan_grid(int groups, int fan)
{
    return (groups % 8 != 0 && groups * fan <= 1024) ? dim3((unsigned)(groups * fan)) : dim3(8, (unsigned)fan, (unsigned)((groups + 7) / 8));
}
__device__ static __forceinline__ void fan_decode(int fan, int &g, int &slot)
{
    if (gridDim.y == 1 && gridDim.z == 1) {
        g = (int)blockIdx.x / fan;
        slot = (int)blockIdx.x - g * fan;
    } else {
        slot = blockIdx.y;
        g = blockIdx.z * 8 + blockIdx.x;
    }
}
// EO position of coefficient j (rows in coefficient form are stored [evens | odds], as the split-2 kernels do)
template <int LOGN>
__device__ static __forceinline__ int eo_pos(int j) { return (j & 1) * (1 << (LOGN - 1)) + (j >> 1); }

template <int LOGN>
__global__ __launch_bounds__(QuarterCfg<LOGN>::T) void ks_intt_digits_q_kernel(DevTables T, KsSmallItems small,
                                                                     KsItem *__restrict__ items_out, int n, int L,
                                                                     int relin, int rows, KsScratch S)
{
    using QC = QuarterCfg<LOGN>;
    using C = typename QC::C;
    extern __shared__ __align__(16) u64 lds[];
    HEFX_STAMP_KERNEL(0);
    HEFX_STAMP_AT(0);
    if (blockIdx.x == 0 && (int)threadIdx.x < n) items_out[threadIdx.x] = small.it[threadIdx.x];
    int p, part;
    quarter_decode(blockIdx.x, p, part);
    if (p >= rows) return;
    const int t = threadIdx.x;
    const int b = p / L, i = p % L;
    const KsItem it = small.it[b];
    const u64 *__restrict__ src = it.c_in + ((size_t)(relin ? 2 * L : L) + i) * QC::N;
    const uint32_t elt = relin ? 1u : item_elt(it);
    u64 v[8];
    quarter_inv<LOGN>(v, [src, elt](int j) {
        return gather_pair(src, (uint32_t)j, elt, LOGN);
    }, lds, ntt_tables(T, i), T.mods[i], T.modsf[i], t, part);
    u64 *__restrict__ dd = S.d + ((size_t)b * L + i) * QC::N;
#pragma unroll
    for (int r = 0; r < 8; ++r) dd[eo_pos<LOGN>(4 * C::idx_nat(t, r) + part)] = v[r];
    HEFX_STAMP_AT(15);
}

// forward loader over a coefficient-form row stored EO
template <int LOGN>
struct EoQuadLoader {
    const u64 *__restrict__ row;
    int t;  // the thread's coefficient column: quarter_fwd_lane(threadIdx.x)
    __device__ __forceinline__ void operator()(int r, u64 &x0, u64 &x1, u64 &x2, u64 &x3) const
    {
        using C = typename QuarterCfg<LOGN>::C;
        const int e = eo_pos<LOGN>(C::idx_nat(t, r));
        constexpr int STEP = (1 << LOGN) / 8;  // N/4 coefficients further = N/8 EO slots further
        x0 = row[e];
        x1 = row[e + STEP];
        x2 = row[e + 2 * STEP];
        x3 = row[e + 3 * STEP];
    }
};

template <int LOGN>
__global__ __launch_bounds__(QuarterCfg<LOGN>::T) void ks_ntt_digits_q_kernel(DevTables T, int L, int rows, KsScratch S)
{
    using QC = QuarterCfg<LOGN>;
    using C = typename QC::C;
    extern __shared__ __align__(16) u64 lds[];
    HEFX_STAMP_KERNEL(1);
    HEFX_STAMP_AT(0);
    // grid (8, 4 L, groups) -> (digit g = (b, i), target slot jj, quarter): dispatched x-fastest, the 4 L workgroups of a
    // digit share an XCD (and no division sits between the kernel's entry and its first table loads)
    int slot, g;
    fan_decode(4 * L, g, slot);
    int jj = slot >> 2;
    const int part = slot & 3;
    if (g >= rows) return;
    const int t = threadIdx.x;
    const int b = g / L, i = g % L;
    if (jj >= i) ++jj;
    const int m = jj < L ? jj : T.k - 1;
    const ModConst mc = T.mods[m];
    const ModConstF mf = T.modsf[m];
    const u64 qi = T.mods[i].q;
    const bool wide_digit = T.modsf[i].q == 0.0;
    u64 v[8];
    const InMode mode = {qi > mc.q, wide_digit, false, 0};
    const EoQuadLoader<LOGN> ld{S.d + ((size_t)b * L + i) * QC::N, quarter_fwd_lane<LOGN>(t)};
    quarter_fwd<LOGN, true>(v, ld, mode, lds, ntt_tables(T, m), mc, mf, t, part, mac_x_slack(mc, mf, L));
    u64 *__restrict__ xd = S.x + (((size_t)b * L + i) * (L + 1) + jj) * QC::N + (size_t)part * QC::Q;
#pragma unroll
    for (int r = 0; r < 8; r += 2) gst16(xd + C::idx_out(t, r), make_ulonglong2(v[r], v[r + 1]));
    HEFX_STAMP_AT(15);
}

template <int LOGN>
__global__ __launch_bounds__(QuarterCfg<LOGN>::T) void ks_moddown_intt_q_kernel(DevTables T, int L, int rows, KsScratch S)
{
    using QC = QuarterCfg<LOGN>;
    using C = typename QC::C;
    extern __shared__ __align__(16) u64 lds[];
    HEFX_STAMP_KERNEL(3);
    HEFX_STAMP_AT(0);
    int p, part;
    quarter_decode(blockIdx.x, p, part);
    if (p >= rows) return;
    const int t = threadIdx.x;
    const int sp = T.k - 1;
    const ModConst mc = T.mods[sp];
    const ulonglong2 *__restrict__ src =
        reinterpret_cast<const ulonglong2 *>(S.acc + ((size_t)p * (L + 1) + L) * QC::N);  // p = b*2 + c
    u64 v[8];
    quarter_inv<LOGN>(v, [src](int j) { return src[j]; }, lds, ntt_tables(T, sp), mc, T.modsf[sp], t, part);
    const u64 half = mc.q >> 1;
    u64 *__restrict__ ud = S.u + (size_t)p * QC::N;
#pragma unroll
    for (int r = 0; r < 8; ++r) ud[eo_pos<LOGN>(4 * C::idx_nat(t, r) + part)] = csub(v[r] + half, mc.q);
    HEFX_STAMP_AT(15);
}

template <int LOGN, class A>
__device__ __forceinline__ void moddown_finish_q_body(const DevTables &T, const KsItem &it, int L, int relin,
                                                      const KsScratch &S, u64 *lds,
                                                      const typename A::TW *__restrict__ tw, const typename A::Ctx &cx,
                                                      const ModConst &mc, u64 half_j, u64 q_special,
                                                      const decltype(md_pinv(A{}, T, 0, 0)) &pinv, int b, int c, int j,
                                                      int t, int part)
{
    using QC = QuarterCfg<LOGN>;
    using C = typename QC::C;
    const InMode mode = {true, true, true, half_j, true, q_special < 2 * mc.q};
    const EoQuadLoader<LOGN> ld{S.u + ((size_t)b * 2 + c) * QC::N, quarter_fwd_lane<LOGN>(t)};
    const size_t off = (size_t)part * QC::Q;
    const u64 *__restrict__ acc = S.acc + (((size_t)b * 2 + c) * (L + 1) + j) * QC::N + off;
    const u64 *__restrict__ addrow = it.c_in + ((size_t)(relin ? c : 0) * L + j) * QC::N;
    const bool has_add = relin || c == 0;
    const uint32_t elt = relin ? 1u : item_elt(it);
    const u64 *__restrict__ pt = it.pt ? it.pt + (size_t)j * QC::N + off : nullptr;
    u64 *__restrict__ dst = it.c_out + ((size_t)c * L + j) * QC::N + off;
    // The epilogue's operands do not depend on the transform: fetched before it, they travel while it runs (registers
    // (r, r+1), r even, are one 16-byte record in every idx_out layout: one gathered pair per record).  Fetch and
    // epilogue are specialised on (add-in?, plaintext?) like the split-2 kernel's -- conditional loads cost a full
    // s_waitcnt vmcnt(0) each -- which here means the transform sits inside the specialisation: four copies of the
    // eight-coefficient core, a price the one-key-switch-at-a-time latency path is worth.
    auto body = [&](auto has_add_c, auto has_pt_c) {
        constexpr bool HA = decltype(has_add_c)::value, HP = decltype(has_pt_c)::value;
        u64 a[8], sadd[8], pp[8];
        auto fetch = [&]() {
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                const int idx = C::idx_out(t, r);
                const ulonglong2 av = gld16(acc + idx);
                ulonglong2 sv = make_ulonglong2(0, 0), pv = make_ulonglong2(0, 0);
                if constexpr (HA) sv = gather_pair(addrow, (uint32_t)((off + idx) >> 1), elt, LOGN);
                if constexpr (HP) pv = gld16(pt + idx);
                a[r] = av.x, a[r + 1] = av.y;
                sadd[r] = sv.x, sadd[r + 1] = sv.y;
                pp[r] = pv.x, pp[r + 1] = pv.y;
            }
        };
        // 1024-thread workgroups (N = 32768) are capped at 128 VGPRs: there the operands are fetched after the transform;
        // elsewhere right BEHIND the transform's own loads (ahead of them they delayed the data every wave waits for first)
        constexpr bool PRE = QC::T < 1024;
        typename A::V f[8];
        if constexpr (PRE) {
            quarter_fwd_raw<LOGN, A>(f, ld, mode, mc, lds, tw, cx, t, part, fetch);
        } else {
            quarter_fwd_raw<LOGN, A>(f, ld, mode, mc, lds, tw, cx, t, part);
            fetch();
        }
        HEFX_STAMP_AT(14);
        u64 keep[8];
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            ulonglong2 o;
            o.x = md_epilogue(A{}, f[r], a[r], sadd[r], pp[r], HP, cx, pinv, mc);
            o.y = md_epilogue(A{}, f[r + 1], a[r + 1], sadd[r + 1], pp[r + 1], HP, cx, pinv, mc);
            gst16(dst + C::idx_out(t, r), o);
            keep[r] = o.x, keep[r + 1] = o.y;
        }
        if constexpr (!HP) {  // accumulate: acc_out = acc_in + out (see moddown_finish_body)
            if (it.acc_out) {
                const size_t arow = ((size_t)c * L + j) * QC::N + off;
                const u64 *__restrict__ ain = it.acc_in + arow;
                u64 *__restrict__ aout = it.acc_out + arow;
                ulonglong2 av[4];
#pragma unroll
                for (int r = 0; r < 8; r += 2) av[r / 2] = gld16(ain + C::idx_out(t, r));
#pragma unroll
                for (int r = 0; r < 8; r += 2)
                    gst16(aout + C::idx_out(t, r),
                          make_ulonglong2(addmod(av[r / 2].x, keep[r], mc.q), addmod(av[r / 2].y, keep[r + 1], mc.q)));
            }
        }
    };
    if (has_add) {
        if (pt)
            body(std::true_type{}, std::true_type{});
        else
            body(std::true_type{}, std::false_type{});
    } else {
        if (pt)
            body(std::false_type{}, std::true_type{});
        else
            body(std::false_type{}, std::false_type{});
    }
}

template <int LOGN>
__global__ __launch_bounds__(QuarterCfg<LOGN>::T) void ks_moddown_finish_q_kernel(DevTables T, const KsItem *__restrict__ items,
                                                                        int L, int relin, int rows, KsScratch S)
{
    extern __shared__ __align__(16) u64 lds[];
    HEFX_STAMP_KERNEL(4);
    HEFX_STAMP_AT(0);
    // grid (8, 4 L, groups), dispatched x-fastest: g = remainder polynomial (b, c)
    int slot, g;
    fan_decode(4 * L, g, slot);
    const int j = slot >> 2, part = slot & 3;
    if (g >= rows) return;
    const int t = threadIdx.x;
    const int b = g >> 1, c = g & 1;
    const int sp = T.k - 1;
    const size_t spj = (size_t)sp * T.k + j;
    const ModConst mc = T.mods[j];
    const ModConstF mf = T.modsf[j];
    const KsItem it = items[b];
    const u64 half_j = T.halfmod[spj], q_special = T.mods[sp].q;
    const PinvU pinv_u{T.invmod[spj]};
    const PinvF pinv_f{T.invmodf[spj]};
    const NttTables nt = ntt_tables(T, j);
    if (mf.q != 0.0)
        moddown_finish_q_body<LOGN, ArithF64>(T, it, L, relin, S, lds, nt.twf, ArithF64::make(mf), mc, half_j, q_special,
                                              pinv_f, b, c, j, t, part);
    else if constexpr (QuarterCfg<LOGN>::T >= 1024)  // 128-VGPR cap: one integer variant
        moddown_finish_q_body<LOGN, ArithU64>(T, it, L, relin, S, lds, nt.tw, ArithU64::make(mc), mc, half_j, q_special,
                                              pinv_u, b, c, j, t, part);
    else
        fwd_int_dispatch(mc, [&](auto pol) {
            using A = decltype(pol);
            moddown_finish_q_body<LOGN, A>(T, it, L, relin, S, lds, nt.tw, A::make(mc), mc, half_j, q_special, pinv_u, b,
                                           c, j, t, part);
        });
    HEFX_STAMP_AT(15);
}

// ------------------------------------------------------------------------------------------------
// PAIR path (round 5) -- the small-batch key switch in FOUR launches with TWO transform phases instead of five launches
// with four: what one key switch at a time costs is its chain of DEPENDENT launches (each ~1 us of table fetches, ~2 us
// until its rows have arrived, the radix-8 passes, the stores, the boundary), not work -- the chip is > 90 % idle.
//
// A quarter workgroup of an inverse transform ends with the coefficients of ONE residue class modulo 4 (4 g + part), and
// every forward stage with gap >= 4 pairs coefficients of the same class: restricted to a class, stages 0 .. logN-3 of the
// size-N forward transform ARE the size-N/4 negacyclic transform with root psi^4 -- table entries tw[1 .. N/4) of the same
// table, prefix 1 (bitrev_logN(idx) = 4 bitrev_(logN-2)(idx) for idx < N/4).  So the workgroup that produced a class
// carries it straight through those stages, registers to registers, and only the LAST TWO forward stages (gaps 2 and 1:
// the four in-place positions 4g..4g+3, one from each class) need the other three quarters -- they move into the
// element-wise kernel that follows anyway.  Symmetrically the FIRST two inverse stages of the special-prime row (gaps 1
// and 2 on positions 4g..4g+3) run in the epilogue of the key MAC, which holds those four sums in registers.
//
//   (P1) ks_pair_digits   (b, i -> m, part): INTT_(q_i) quarter of perm_g(c1)[i]  ->  reduce mod m  ->  forward stages
//                         0..logN-3 mod m  ->  xpre[b][i][m][part][.]        (the digit itself is never stored; every
//                         target modulus recomputes the inverse quarter -- idle CUs, no extra latency)
//   (P2) ks_pair_mac      (b, m, g): last two forward stages on the four planes of every digit, key MAC, and for m = P the
//                         first two inverse stages  ->  acc[b][c][j]  (j < L),  upre[b][c][part][.]
//   (P3) ks_pair_moddown  (b, c, j, part): rest of INTT_P on plane `part`, + P/2, reduce mod q_j, - (P/2 mod q_j), forward
//                         stages 0..logN-3 mod q_j  ->  fpre[b][c][j][part][.]
//   (P4) ks_pair_finish   (b, c, j, g): last two forward stages, then the mod-down epilogue of ks_moddown_finish (add-in
//                         gathered from the source, optional plaintext product, optional accumulate)
// Same integers as the five-launch sequence at every stored word -- the transforms are the same butterflies in another
// order of WORKGROUPS, not of arithmetic -- hence the same bits (every small-batch parity test runs through here).
// Scratch: xpre takes the place of x (same size), upre of u, fpre lives in x's storage (x is dead once P2 has run).
// ------------------------------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__(QuarterCfg<LOGN>::T) void ks_pair_digits_kernel(DevTables T, KsSmallItems small,
                                                                   KsItem *__restrict__ items_out, int n, int L, int relin,
                                                                   int rows, KsScratch S)
{
    using QC = QuarterCfg<LOGN>;
    using C = typename QC::C;
    extern __shared__ __align__(16) u64 lds[];
    HEFX_STAMP_KERNEL(0);
    HEFX_STAMP_AT(0);
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (int)threadIdx.x < n) items_out[threadIdx.x] = small.it[threadIdx.x];
    // grid (8, 4 L, groups), dispatched x-fastest: the 4 L workgroups of digit g = (b, i) share an XCD (they all gather the
    // same source row)
    int slot, g;
    fan_decode(4 * L, g, slot);
    int jj = slot >> 2;
    const int part = slot & 3;
    if (g >= rows) return;
    const int t = threadIdx.x;
    const int b = g / L, i = g % L;
    if (jj >= i) ++jj;  // skip the diagonal; jj == L is the special prime
    const int m = jj < L ? jj : T.k - 1;
    const KsItem it = small.it[b];
    const u64 *__restrict__ src = it.c_in + ((size_t)(relin ? 2 * L : L) + i) * QC::N;
    const uint32_t elt = relin ? 1u : item_elt(it);
    u64 v[8];
    quarter_inv<LOGN>(v, [src, elt](int j) { return gather_pair(src, (uint32_t)j, elt, LOGN); }, lds, ntt_tables(T, i),
                      T.mods[i], T.modsf[i], t, part);
    HEFX_STAMP_AT(15);
    HEFX_STAMP_KERNEL(1);  // (stamp builds: the forward half reports as the old digit-NTT launch)
    HEFX_STAMP_AT(0);
    // v[r] = digit coefficient 4 idx_nat(t, r) + part, canonical modulo q_i: exactly the register layout the size-N/4
    // forward core starts from, and its first pass writes the LDS words this thread's last inverse pass read (no barrier)
    const ModConst mc = T.mods[m];
    const ModConstF mf = T.modsf[m];
    const InMode mode = {T.mods[i].q > mc.q, T.modsf[i].q == 0.0, false, 0};
    u64 *__restrict__ xd = S.x + (((size_t)b * L + i) * (L + 1) + jj) * QC::N + (size_t)part * QC::Q;
    const NttTables nt = ntt_tables(T, m);
    fwd_policy_dispatch(mc, mf, [&](auto pol) {
        using A = decltype(pol);
        const typename A::Ctx cx = make_ctx(A{}, mc, mf);
        typename A::V f[8];
        fwd_inputs8<A>(f, v, mode, cx, mc);
        ntt8_fwd_core<LOGN - 2, A>(f, reinterpret_cast<typename A::V *>(lds), fwd_tw(A{}, nt), cx, t, 1);
#pragma unroll
        for (int r = 0; r < 8; r += 2) gst16(xd + C::idx_out(t, r), make_ulonglong2(A::raw(f[r]), A::raw(f[r + 1])));
    });
    HEFX_STAMP_AT(15);
}

// the last two stages of a size-2^LOGN forward transform on in-place positions 4g..4g+3 (p0..p3): stage logN-2 pairs
// (p0,p2), (p1,p3) with tw[N/4 + g]; stage logN-1 pairs (p0,p1) with tw[N/2 + 2g] and (p2,p3) with tw[N/2 + 2g + 1]
template <int LOGN, class A>
__device__ __forceinline__ void fwd_last_two(typename A::V &p0, typename A::V &p1, typename A::V &p2, typename A::V &p3,
                                             const typename A::TW &wb, const typename A::TW &wa0,
                                             const typename A::TW &wa1, const typename A::Ctx &cx)
{
    A::ct(p0, p2, wb, cx, LOGN - 2);
    A::ct(p1, p3, wb, cx, LOGN - 2);
    A::ct(p0, p1, wa0, cx, LOGN - 1);
    A::ct(p2, p3, wa1, cx, LOGN - 1);
}

// unfinished forward values (p, q) -> the MAC policy's operand pair
template <class AF, class P>
__device__ __forceinline__ typename P::X pair_mac_operand(typename AF::V p, typename AF::V q, int slack,
                                                          const typename AF::Ctx &fx, const typename P::Ctx &cx)
{
    if constexpr (AF::IS_F64) {
        return make_double2(p, q);  // MacF multiplies the unfinished doubles as they are (|x| < 2^45)
    } else {
        ulonglong2 w;
        if (slack == 1)
            w = make_ulonglong2(AF::template mac_operand_lazy<1>(p, fx), AF::template mac_operand_lazy<1>(q, fx));
        else if (slack >= 2)
            w = make_ulonglong2(AF::template mac_operand_lazy<2>(p, fx), AF::template mac_operand_lazy<2>(q, fx));
        else
            w = make_ulonglong2(AF::mac_operand(p, fx), AF::mac_operand(q, fx));
        return P::xin(w, false, cx);
    }
}

// One thread = the four positions 4g..4g+3 of ONE key polynomial c (round 5, after the first stamps: with both polynomials
// per thread the special-prime workgroups -- 60-bit butterflies, eight Barrett reductions, two inverse radix-4 -- ran
// ~800 dependent instructions on one wave per SIMD, 5-6 us of a 44 us level).  The last two forward stages of every
// digit are recomputed by the two threads of a group; the MAC policies keep their two-polynomial shape and are fed the
// same key words in both slots -- the second slot's accumulators are never read and vanish as dead code.
template <int LOGN, class AF, class P>
__device__ __forceinline__ void pair_mac_body(const DevTables &T, const KsItem &it, int L, int relin, int b, int jj, int m,
                                              int c, uint32_t g, const KsScratch &S)
{
    constexpr size_t N = (size_t)1 << LOGN, Q = N / 4;
    const ModConst mc = T.mods[m];
    const ModConstF mf = T.modsf[m];
    const typename AF::Ctx fx = make_ctx(AF{}, mc, mf);
    const typename P::Ctx cx = P::make(mc, mf);
    const NttTables nt = ntt_tables(T, m);
    const typename AF::TW *__restrict__ tw = fwd_tw(AF{}, nt);
    const typename AF::TW wb = tw[Q + g], wa0 = tw[2 * Q + 2 * g], wa1 = tw[2 * Q + 2 * g + 1];
    const int slack = mac_x_slack(mc, mf, L);
    const size_t kpoly = (size_t)T.k * N;
    P A0, A1;  // positions (4g, 4g+1) and (4g+2, 4g+3)
    auto mac_digit = [&](int i, const typename P::X &xa, const typename P::X &xb) {
        const u64 *kb = it.key + ((size_t)i * 2 * T.k + m) * N + (size_t)c * kpoly + 4 * (size_t)g;
        const ulonglong2 ka = gld16(kb), kc = gld16(kb + 2);
        A0.mac(xa, P::kin(ka, ka, cx), cx);
        A1.mac(xb, P::kin(kc, kc, cx), cx);
    };
    if (jj < L) {  // the digit in NTT form modulo its own prime: the (rotated) input row itself, gathered
        const u64 *__restrict__ own = it.c_in + ((size_t)(relin ? 2 * L : L) + jj) * N;
        const uint32_t elt = relin ? 1u : item_elt(it);
        const ulonglong2 xa = gather_pair(own, 2 * g, elt, LOGN), xb = gather_pair(own, 2 * g + 1, elt, LOGN);
        mac_digit(jj, P::xin(xa, true, cx), P::xin(xb, true, cx));
    }
    for (int i = 0; i < L; ++i) {
        if (i == jj) continue;
        const u64 *__restrict__ xp = S.x + (((size_t)b * L + i) * (L + 1) + jj) * N + g;
        typename AF::V p0 = AF::unraw(gld8(xp)), p1 = AF::unraw(gld8(xp + Q)), p2 = AF::unraw(gld8(xp + 2 * Q)),
                       p3 = AF::unraw(gld8(xp + 3 * Q));
        fwd_last_two<LOGN, AF>(p0, p1, p2, p3, wb, wa0, wa1, fx);
        mac_digit(i, pair_mac_operand<AF, P>(p0, p1, slack, fx, cx), pair_mac_operand<AF, P>(p2, p3, slack, fx, cx));
    }
    ulonglong2 ra, rb, dead_a, dead_b;  // (slot 0 of each accumulator set; slot 1 is dead)
    if (jj < L) {
#ifdef HEFX_NO_LT2Q
        A0.result_data(ra, dead_a, cx);
        A1.result_data(rb, dead_b, cx);
#else
        A0.template result_data<true>(ra, dead_a, cx);
        A1.template result_data<true>(rb, dead_b, cx);
#endif
    } else {
#ifdef HEFX_NO_LT2Q
        A0.result(ra, dead_a, cx);
        A1.result(rb, dead_b, cx);
#else
        A0.template result<true>(ra, dead_a, cx);
        A1.template result<true>(rb, dead_b, cx);
#endif
    }
    if (jj < L) {
        u64 *a = S.acc + (((size_t)b * 2 + c) * (L + 1) + jj) * N + 4 * (size_t)g;
        gst16(a, ra);
        gst16(a + 2, rb);
        return;
    }
    // the special prime's rows: the first two stages of INTT_P here (the inverse policy of P; the MAC's words are below 2q
    // or canonical, what its first stage takes), one plane per residue class for ks_pair_moddown
    using AI = std::conditional_t<AF::IS_F64, ArithF64, ArithU64>;
    const typename AI::Ctx ix = make_ctx(AI{}, mc, mf);
    const typename AI::TW *__restrict__ itw = inv_tw(AI{}, nt);
    const typename AI::TW va0 = itw[2 * Q + 2 * g], va1 = itw[2 * Q + 2 * g + 1], vb = itw[Q + g];
    u64 *__restrict__ up = S.u + ((size_t)b * 2 + c) * N;
    const typename AI::V q0 = AI::from_u64(ra.x), q1 = AI::from_u64(ra.y), q2 = AI::from_u64(rb.x), q3 = AI::from_u64(rb.y);
    const typename AI::V s01 = AI::gs_half_sum(q0, q1, ix), d01 = AI::gs_half_diff(q0, q1, va0, ix);
    const typename AI::V s23 = AI::gs_half_sum(q2, q3, ix), d23 = AI::gs_half_diff(q2, q3, va1, ix);
    gst8(up + g, AI::raw(AI::inv_add(s01, s23, ix)));                  // part 0: (sum, sum)
    gst8(up + Q + g, AI::raw(AI::inv_add(d01, d23, ix)));              // part 1: (difference, sum)
    gst8(up + 2 * Q + g, AI::raw(AI::inv_sub_mul(s01, s23, vb, ix)));  // part 2: (sum, twiddled difference)
    gst8(up + 3 * Q + g, AI::raw(AI::inv_sub_mul(d01, d23, vb, ix)));  // part 3
}

template <int LOGN>
__global__ __launch_bounds__(256) void ks_pair_mac_kernel(DevTables T, const KsItem *__restrict__ items, int L, int relin,
                                                         KsScratch S)
{
    const int jj = blockIdx.y >> 1, c = blockIdx.y & 1, b = blockIdx.z;  // grid (N/4/256, 2 (L+1), n)
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;  // positions 4g .. 4g+3
    const int m = jj < L ? jj : T.k - 1;
#ifdef HEFX_STAMP
    const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (threadIdx.x == 0 && wg < 1024) hefx_stamp_buf[((size_t)2 * 1024 + wg) * 16] = wall_clock64();
#endif
    const KsItem it = items[b];
    const ModConst mc = T.mods[m];
    if (T.modsf[m].q != 0.0)
        pair_mac_body<LOGN, ArithF64, MacF>(T, it, L, relin, b, jj, m, c, g, S);
    else
        fwd_int_dispatch(mc, [&](auto pol) {
            using AF = decltype(pol);
            if (L <= 8 && (mc.q >> 60) == 0)  // (mac_dispatch's rule)
                pair_mac_body<LOGN, AF, MacL>(T, it, L, relin, b, jj, m, c, g, S);
            else
                pair_mac_body<LOGN, AF, MacW>(T, it, L, relin, b, jj, m, c, g, S);
        });
#ifdef HEFX_STAMP
    if (HEFX_STAMP > 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && wg < 1024) hefx_stamp_buf[((size_t)2 * 1024 + wg) * 16 + 15] = wall_clock64();
#endif
}

template <int LOGN>
__global__ __launch_bounds__(QuarterCfg<LOGN>::T) void ks_pair_moddown_kernel(DevTables T, int L, int rows, KsScratch S)
{
    using QC = QuarterCfg<LOGN>;
    using C = typename QC::C;
    extern __shared__ __align__(16) u64 lds[];
    HEFX_STAMP_KERNEL(3);
    HEFX_STAMP_AT(0);
    // grid (8, 4 L, groups): g = remainder polynomial (b, c); every data prime j recomputes the quarter of INTT_P it needs
    int slot, g;
    fan_decode(4 * L, g, slot);
    const int j = slot >> 2, part = slot & 3;
    if (g >= rows) return;
    const int t = threadIdx.x;
    const int sp = T.k - 1;
    const ModConst mcp = T.mods[sp];
    const ModConstF mfp = T.modsf[sp];
    const NttTables ntp = ntt_tables(T, sp);
    const u64 *__restrict__ plane = S.u + (size_t)g * QC::N + (size_t)part * QC::Q;
    u64 v[8];
    if (mfp.q != 0.0)
        quarter_inv_planes<LOGN, ArithF64>(v, plane, lds, ntp.itwf, ArithF64::make(mfp), t);
    else
        quarter_inv_planes<LOGN, ArithU64>(v, plane, lds, ntp.itw, ArithU64::make(mcp), t);
    HEFX_STAMP_AT(15);
    HEFX_STAMP_KERNEL(4);
    HEFX_STAMP_AT(0);
    const u64 half = mcp.q >> 1;
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = csub(v[r] + half, mcp.q);  // (INTT_P(acc_P) + floor(P/2)) mod P
    const ModConst mc = T.mods[j];
    const ModConstF mf = T.modsf[j];
    const InMode mode = {true, true, true, T.halfmod[(size_t)sp * T.k + j], true, mcp.q < 2 * mc.q};
    u64 *__restrict__ fd = S.x + ((size_t)g * L + j) * QC::N + (size_t)part * QC::Q;
    const NttTables nt = ntt_tables(T, j);
    fwd_policy_dispatch(mc, mf, [&](auto pol) {
        using A = decltype(pol);
        const typename A::Ctx cx = make_ctx(A{}, mc, mf);
        typename A::V f[8];
        fwd_inputs8<A>(f, v, mode, cx, mc);
        ntt8_fwd_core<LOGN - 2, A>(f, reinterpret_cast<typename A::V *>(lds), fwd_tw(A{}, nt), cx, t, 1);
#pragma unroll
        for (int r = 0; r < 8; r += 2) gst16(fd + C::idx_out(t, r), make_ulonglong2(A::raw(f[r]), A::raw(f[r + 1])));
    });
    HEFX_STAMP_AT(15);
}

template <int LOGN, class A>
__device__ __forceinline__ void pair_finish_body(const DevTables &T, const KsItem &it, int L, int relin, int b, int c, int j,
                                                 uint32_t g, const KsScratch &S)
{
    constexpr size_t N = (size_t)1 << LOGN, Q = N / 4;
    const int sp = T.k - 1;
    const ModConst mc = T.mods[j];
    const ModConstF mf = T.modsf[j];
    const typename A::Ctx cx = make_ctx(A{}, mc, mf);
    const auto pinv = md_pinv(A{}, T, sp, j);
    const typename A::TW *__restrict__ tw = fwd_tw(A{}, ntt_tables(T, j));
    const typename A::TW wb = tw[Q + g], wa0 = tw[2 * Q + 2 * g], wa1 = tw[2 * Q + 2 * g + 1];
    const u64 *__restrict__ fp = S.x + (((size_t)b * 2 + c) * L + j) * N + g;
    typename A::V p0 = A::unraw(gld8(fp)), p1 = A::unraw(gld8(fp + Q)), p2 = A::unraw(gld8(fp + 2 * Q)),
                  p3 = A::unraw(gld8(fp + 3 * Q));
    const size_t w4 = 4 * (size_t)g;
    const u64 *__restrict__ acc = S.acc + (((size_t)b * 2 + c) * (L + 1) + j) * N + w4;
    const ulonglong2 a01 = gld16(acc), a23 = gld16(acc + 2);
    const u64 *__restrict__ addrow = it.c_in + ((size_t)(relin ? c : 0) * L + j) * N;
    const bool has_add = relin || c == 0;
    const uint32_t elt = relin ? 1u : item_elt(it);
    const u64 *__restrict__ pt = it.pt ? it.pt + (size_t)j * N + w4 : nullptr;
    u64 *__restrict__ dst = it.c_out + ((size_t)c * L + j) * N + w4;
    // specialised on (add-in?, plaintext?) by workgroup-uniform branches, like ks_moddown_finish: no conditional loads
    auto epilogue = [&](auto has_add_c, auto has_pt_c) {
        constexpr bool HA = decltype(has_add_c)::value, HP = decltype(has_pt_c)::value;
        ulonglong2 s01 = make_ulonglong2(0, 0), s23 = s01, t01 = s01, t23 = s01;
        if constexpr (HA) {
            s01 = gather_pair(addrow, 2 * g, elt, LOGN);
            s23 = gather_pair(addrow, 2 * g + 1, elt, LOGN);
        }
        if constexpr (HP) {
            t01 = gld16(pt);
            t23 = gld16(pt + 2);
        }
        fwd_last_two<LOGN, A>(p0, p1, p2, p3, wb, wa0, wa1, cx);
        ulonglong2 o01, o23;
        o01.x = md_epilogue(A{}, p0, a01.x, s01.x, t01.x, HP, cx, pinv, mc);
        o01.y = md_epilogue(A{}, p1, a01.y, s01.y, t01.y, HP, cx, pinv, mc);
        o23.x = md_epilogue(A{}, p2, a23.x, s23.x, t23.x, HP, cx, pinv, mc);
        o23.y = md_epilogue(A{}, p3, a23.y, s23.y, t23.y, HP, cx, pinv, mc);
        gst16(dst, o01);
        gst16(dst + 2, o23);
        if constexpr (!HP) {  // accumulate (hefx_apply_galois_add_batch): acc_out = acc_in + out
            if (it.acc_out) {
                const size_t arow = ((size_t)c * L + j) * N + w4;
                const ulonglong2 i01 = gld16(it.acc_in + arow), i23 = gld16(it.acc_in + arow + 2);
                gst16(it.acc_out + arow, make_ulonglong2(addmod(i01.x, o01.x, mc.q), addmod(i01.y, o01.y, mc.q)));
                gst16(it.acc_out + arow + 2, make_ulonglong2(addmod(i23.x, o23.x, mc.q), addmod(i23.y, o23.y, mc.q)));
            }
        }
    };
    if (has_add) {
        if (pt)
            epilogue(std::true_type{}, std::true_type{});
        else
            epilogue(std::true_type{}, std::false_type{});
    } else {
        if (pt)
            epilogue(std::false_type{}, std::true_type{});
        else
            epilogue(std::false_type{}, std::false_type{});
    }
}

template <int LOGN>
__global__ __launch_bounds__(256) void ks_pair_finish_kernel(DevTables T, const KsItem *__restrict__ items, int L, int relin,
                                                            KsScratch S)
{
    const int row = blockIdx.y, b = blockIdx.z;  // row = c * L + j
    const int c = row / L, j = row % L;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
#ifdef HEFX_STAMP
    const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (threadIdx.x == 0 && wg < 1024) hefx_stamp_buf[((size_t)5 * 1024 + wg) * 16] = wall_clock64();
#endif
    const KsItem it = items[b];
    const ModConst mc = T.mods[j];
    fwd_policy_dispatch(mc, T.modsf[j], [&](auto pol) {
        using A = decltype(pol);
        pair_finish_body<LOGN, A>(T, it, L, relin, b, c, j, g, S);
    });
#ifdef HEFX_STAMP
    if (HEFX_STAMP > 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && wg < 1024) hefx_stamp_buf[((size_t)5 * 1024 + wg) * 16 + 15] = wall_clock64();
#endif
}

template <typename K>
static void set_lds(K kernel, size_t bytes)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)bytes);
}

template <int LOGN>
static hipError_t launch_keyswitch_chunk_t(const DevTables &T, int L, int n, const KsItem *batch, bool relin,
                                           const KsScratch &scr, int sub, bool alias,
                                           const KsSmallItems *small, int quarter, hipStream_t s, KsProf *prof, int nsrc)
{
    using SC = SplitCfg<LOGN>;
    const size_t lds = SC::LDS_BYTES;
    // experiment knobs: a larger LDS request as an occupancy cap per kernel (bytes; workgroups per CU = 160 KiB / request)
    auto lds_knob = [lds](const char *name) {
        const char *e = getenv(name);
        const size_t v = e ? (size_t)atol(e) : 0;
        return v > lds ? v : lds;
    };
    static const size_t lds_fin = lds_knob("HEFX_FIN_LDS"), lds_ntt = lds_knob("HEFX_NTT_LDS"),
                        lds_intt = lds_knob("HEFX_INTT_LDS"), lds_mdi = lds_knob("HEFX_MDI_LDS");
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        set_lds(ks_intt_digits_kernel<LOGN>, lds_intt);
        set_lds(ks_intt_digits_small_kernel<LOGN>, lds);
        set_lds(ks_ntt_digits_kernel<LOGN>, lds_ntt);
        set_lds(ks_moddown_intt_kernel<LOGN>, lds_mdi);
        set_lds(ks_moddown_finish_kernel<LOGN>, lds_fin);
        if (getenv("HEFX_DEBUG")) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ks_ntt_digits_kernel<LOGN>, SC::T, lds);
            fprintf(stderr, "[hefx] ks_ntt_digits_kernel<%d>: %d threads, %zu B LDS -> %d workgroups/CU\n", LOGN, SC::T,
                    lds, nb);
        }
    }
    const int rl = relin ? 1 : 0;
    // optional profiling (hefx_profile_*): an event before every launch, tagged with its stage
    auto mark = [&](int stage) {
        if (prof && prof->used < prof->cap) {
            (void)hipEventRecord(prof->ev[prof->used], s);
            prof->stage[prof->used++] = stage;
        }
    };
    if (nsrc > 0) {  // exact hoisting: sources batch[n .. n + nsrc), see ks_mac_exact_kernel
        KsScratch hs = scr, fb = scr;
        hs.gate_mode = 1;
        fb.gate_mode = 2;
        mark(1);
        hipLaunchKernelGGL((ks_intt_digits_kernel<LOGN>), dim3(split_grid(nsrc * L)), dim3(SC::T), lds_intt, s, T, batch + n, L,
                           0, 1, nsrc * L, hs);
        mark(2);
        hipLaunchKernelGGL((ks_ntt_digits_kernel<LOGN>), dim3(group_grid(nsrc * L, L)), dim3(SC::T), lds_ntt, s, T, L, nsrc * L,
                           0, 0, hs);
        mark(3);
        hipLaunchKernelGGL(ks_mac_exact_kernel, dim3(SC::N / 2 / 256, L + 1, n), dim3(256), 0, s, T, batch, L, hs);
        mark(4);
        hipLaunchKernelGGL((ks_moddown_intt_kernel<LOGN>), dim3(split_grid(n * 2)), dim3(SC::T), lds_mdi, s, T, L, n * 2, hs);
        mark(5);
        hipLaunchKernelGGL((ks_moddown_finish_kernel<LOGN>), dim3(group_grid(n * 2, L)), dim3(SC::T), lds_fin, s, T, batch, L,
                           0, n * 2, hs);
        // the fallback: the ordinary five launches, which exit at once unless a source of this chunk held a zero coefficient
        mark(7);
        hipLaunchKernelGGL((ks_intt_digits_kernel<LOGN>), dim3(split_grid(n * L)), dim3(SC::T), lds_intt, s, T, batch, L, 0, 0,
                           n * L, fb);
        hipLaunchKernelGGL((ks_ntt_digits_kernel<LOGN>), dim3(group_grid(n * L, L)), dim3(SC::T), lds_ntt, s, T, L, n * L, 0, 0,
                           fb);
        hipLaunchKernelGGL(ks_mac_kernel<false>, dim3(SC::N / 2 / 256, L + 1, (n + 1) / 2), dim3(256), 0, s, T, batch, L, 0, 0, n,
                           0, fb);
        hipLaunchKernelGGL((ks_moddown_intt_kernel<LOGN>), dim3(split_grid(n * 2)), dim3(SC::T), lds_mdi, s, T, L, n * 2, fb);
        hipLaunchKernelGGL((ks_moddown_finish_kernel<LOGN>), dim3(group_grid(n * 2, L)), dim3(SC::T), lds_fin, s, T, batch, L,
                           0, n * 2, fb);
        mark(-1);
        return hipGetLastError();
    }
    // quarter rows: a small chunk that cannot fill the chip with split-2 workgroups (unfused, no aliasing / hoisting).
    // `quarter` is a mask over the four transform launches (KS_Q_*): the scratch arrays have one layout, so each launch
    // picks its own workgroup shape -- ks_run gives quarter rows to the launches whose quarter grid still fits the chip in
    // one round (at n = 8, L = 5: the two inverse launches and not the 200 digit transforms).
    if (small && (quarter & KS_Q_PAIR) && sub >= n) {  // the pair path: four launches, two transform phases (ks_pair_*)
        static PerDeviceOnce attrp;
        const size_t ldsq = QuarterCfg<LOGN>::LDS_BYTES;
        if (attrp.first()) {
            set_lds(ks_pair_digits_kernel<LOGN>, ldsq);
            set_lds(ks_pair_moddown_kernel<LOGN>, ldsq);
        }
        constexpr int TQ = QuarterCfg<LOGN>::T;
        mark(1);
        hipLaunchKernelGGL((ks_pair_digits_kernel<LOGN>), fan_grid(n * L, 4 * L), dim3(TQ), ldsq, s, T, *small,
                           const_cast<KsItem *>(batch), n, L, rl, n * L, scr);
        mark(3);
        hipLaunchKernelGGL((ks_pair_mac_kernel<LOGN>), dim3(SC::N / 4 / 256, 2 * (L + 1), n), dim3(256), 0, s, T, batch, L, rl,
                           scr);
        mark(4);
        hipLaunchKernelGGL((ks_pair_moddown_kernel<LOGN>), fan_grid(n * 2, 4 * L), dim3(TQ), ldsq, s, T, L, n * 2,
                           scr);
        mark(5);
        hipLaunchKernelGGL((ks_pair_finish_kernel<LOGN>), dim3(SC::N / 4 / 256, 2 * L, n), dim3(256), 0, s, T, batch, L, rl,
                           scr);
        mark(-1);
        return hipGetLastError();
    }
    // heavy-first dispatch order of the digit transforms (ntt_digit_row) while the launch is a few resident waves:
    // HEFX_HEAVY_FIRST=0 switches it off, HEFX_HEAVY_MAX=<workgroups> moves the bound.  Measured at L = 5 (C4 ring,
    // profiles/r06/heavy_first_ab.txt): n = 8 80.3 -> 77.6 us, n = 12 90.9 -> 90.4, n = 16 even, n = 24 / 32 +1.5 / +2 us (several
    // rounds of workgroups: the grouped order's L2 locality wins again) -- hence up to ~2.5 workgroups per CU
    static const bool heavy_on = !(getenv("HEFX_HEAVY_FIRST") && atoi(getenv("HEFX_HEAVY_FIRST")) == 0);
    static const int heavy_max = getenv("HEFX_HEAVY_MAX") ? atoi(getenv("HEFX_HEAVY_MAX")) : 640;
    auto heavy_order = [&](int m) { return heavy_on && m * L * L * 2 <= heavy_max ? 2 : 0; };
    if (small && (quarter & KS_Q_ALL) && sub >= n) {
        static PerDeviceOnce attrq;
        const size_t ldsq = QuarterCfg<LOGN>::LDS_BYTES;
        if (attrq.first()) {
            set_lds(ks_intt_digits_q_kernel<LOGN>, ldsq);
            set_lds(ks_ntt_digits_q_kernel<LOGN>, ldsq);
            set_lds(ks_moddown_intt_q_kernel<LOGN>, ldsq);
            set_lds(ks_moddown_finish_q_kernel<LOGN>, ldsq);
        }
        constexpr int TQ = QuarterCfg<LOGN>::T;
        mark(1);
        if (quarter & KS_Q_INTT)
            hipLaunchKernelGGL((ks_intt_digits_q_kernel<LOGN>), dim3(quarter_grid(n * L)), dim3(TQ), ldsq, s, T, *small,
                               const_cast<KsItem *>(batch), n, L, rl, n * L, scr);
        else
            hipLaunchKernelGGL((ks_intt_digits_small_kernel<LOGN>), dim3(split_grid(n * L)), dim3(SC::T), lds, s, T, *small,
                               const_cast<KsItem *>(batch), n, L, rl, n * L, scr);
        mark(2);
        if (quarter & KS_Q_NTT)
            hipLaunchKernelGGL((ks_ntt_digits_q_kernel<LOGN>), fan_grid(n * L, 4 * L), dim3(TQ), ldsq, s, T, L,
                               n * L, scr);
        else
            hipLaunchKernelGGL((ks_ntt_digits_kernel<LOGN>), dim3(group_grid(n * L, L)), dim3(SC::T), lds_ntt, s, T, L,
                               n * L, 0, heavy_order(n), scr);
        mark(3);
        hipLaunchKernelGGL(ks_mac_kernel<false>, dim3(SC::N / 2 / 256, L + 1, (n + 1) / 2), dim3(256), 0, s, T, batch, L, rl, 0,
                           n, 0, scr);
        mark(4);
        if (quarter & KS_Q_MDI)
            hipLaunchKernelGGL((ks_moddown_intt_q_kernel<LOGN>), dim3(quarter_grid(n * 2)), dim3(TQ), ldsq, s, T, L, n * 2,
                               scr);
        else
            hipLaunchKernelGGL((ks_moddown_intt_kernel<LOGN>), dim3(split_grid(n * 2)), dim3(SC::T), lds_mdi, s, T, L, n * 2,
                               scr);
        mark(5);
        if (quarter & KS_Q_FIN)
            hipLaunchKernelGGL((ks_moddown_finish_q_kernel<LOGN>), fan_grid(n * 2, 4 * L), dim3(TQ), ldsq, s, T,
                               batch, L, rl, n * 2, scr);
        else
            hipLaunchKernelGGL((ks_moddown_finish_kernel<LOGN>), dim3(group_grid(n * 2, L)), dim3(SC::T), lds_fin, s, T,
                               batch, L, rl, n * 2, scr);
        mark(-1);
        return hipGetLastError();
    }
    static const int force_stream_x = getenv("HEFX_STREAM_X") ? atoi(getenv("HEFX_STREAM_X")) : -1;
    auto stream_x_of = [&](int m) {  // x of m items against the 256 MB Infinity Cache: beyond it, stream x
        return force_stream_x >= 0 ? force_stream_x : ((size_t)m * L * (L + 1) * SC::N * 8 > ((size_t)256 << 20) ? 1 : 0);
    };
    if (alias) {  // in-place rotations: their inputs move to scratch first (see ks_alias_copy_kernel)
        mark(0);
        hipLaunchKernelGGL(ks_alias_copy_kernel, dim3(SC::N / 2 / 256, 2 * L, n), dim3(256), 0, s, T, batch, L);
    }
    mark(1);
    if (small)
        hipLaunchKernelGGL((ks_intt_digits_small_kernel<LOGN>), dim3(split_grid(n * L)), dim3(SC::T), lds, s, T, *small,
                           const_cast<KsItem *>(batch), n, L, rl, n * L, scr);
    else
        hipLaunchKernelGGL((ks_intt_digits_kernel<LOGN>), dim3(split_grid(n * L)), dim3(SC::T), lds_intt, s, T, batch, L, rl, 0,
                           n * L, scr);
    if (sub < 0) {  // hybrid fused digit-NTT + MAC (LOGN <= 14): only the integer-policy targets' products travel
        if constexpr (LOGN <= 14) {
            static PerDeviceOnce fattr;
            if (fattr.first()) set_lds(ks_ntt_macf_kernel<LOGN>, FusedCfg<LOGN>::LDS_BYTES);
            // sub = -1 - (nf | special_int << 8), from ks_run: nf = FP64-policy target slots among 0..L at this level,
            // special_int = the special prime is an integer-policy slot
            const int code = -sub - 1, nf = code & 0xff;
            const bool special_int = (code >> 8) & 1;
            const int nint = L + 1 - nf;  // integer-policy target slots
            // digits transformed to an integer slot: L - 1 for a data prime, L for the special prime
            const int pairs = special_int ? (nint - 1) * (L - 1) + L : nint * (L - 1);
            const int per_item = 2 * (nf + pairs);
            const int stream_x = (size_t)n * pairs * SC::N * 8 > ((size_t)256 << 20) ? 1 : 0;
            mark(6);
            hipLaunchKernelGGL((ks_ntt_macf_kernel<LOGN>), dim3(((n + 7) / 8) * 8 * per_item), dim3(FusedCfg<LOGN>::T),
                               FusedCfg<LOGN>::LDS_BYTES, s, T, batch, L, rl, n, nf, per_item, stream_x, scr);
            if (nint > 0) {
                mark(3);
                if (stream_x)
                    hipLaunchKernelGGL(ks_mac_kernel<true>, dim3(SC::N / 2 / 256, nint, (n + 1) / 2), dim3(256), 0, s, T, batch,
                                       L, rl, 0, n, 1, scr);
                else
                    hipLaunchKernelGGL(ks_mac_kernel<false>, dim3(SC::N / 2 / 256, nint, (n + 1) / 2), dim3(256), 0, s, T, batch,
                                       L, rl, 0, n, 1, scr);
            }
        }
    } else
    // digit x modulus products only ever exist for `sub` items: K2 writes them, the MAC consumes them right away
    for (int item0 = 0; item0 < n; item0 += sub) {
        const int m = n - item0 < sub ? n - item0 : sub;
        mark(2);
        // x of this (sub-)chunk against the 256 MB Infinity Cache: beyond it, stream x (HEFX_STREAM_X=0/1 overrides)
        const int stream_x = stream_x_of(m);
        hipLaunchKernelGGL((ks_ntt_digits_kernel<LOGN>), dim3(group_grid(m * L, L)), dim3(SC::T), lds_ntt, s, T, L, m * L,
                           item0, stream_x | heavy_order(m), scr);
        mark(3);
        if (stream_x)
            hipLaunchKernelGGL(ks_mac_kernel<true>, dim3(SC::N / 2 / 256, L + 1, (m + 1) / 2), dim3(256), 0, s, T, batch,
                               L, rl, item0, m, 0, scr);
        else
            hipLaunchKernelGGL(ks_mac_kernel<false>, dim3(SC::N / 2 / 256, L + 1, (m + 1) / 2), dim3(256), 0, s, T, batch,
                               L, rl, item0, m, 0, scr);
    }
    mark(4);
    hipLaunchKernelGGL((ks_moddown_intt_kernel<LOGN>), dim3(split_grid(n * 2)), dim3(SC::T), lds_mdi, s, T, L, n * 2, scr);
    mark(5);
    hipLaunchKernelGGL((ks_moddown_finish_kernel<LOGN>), dim3(group_grid(n * 2, L)), dim3(SC::T), lds_fin, s, T, batch, L,
                       rl, n * 2, scr);
    mark(-1);
    return hipGetLastError();
}

// out[r] = (addin ? addin[r] : 0) + sum_ch partial[ch][r] for `rows` rows of N words; the modulus of row r is
// q_(r % period) for r % period < L, the special prime otherwise (period = L+1 for S, L for C0)
__global__ __launch_bounds__(256) void lt2_reduce_kernel(DevTables T, int L, int period, int rows, int chunks,
                                                         const u64 *__restrict__ partial, const u64 *addin, u64 *out)
{
    const size_t n = (size_t)1 << T.logn;
    const int r = blockIdx.y;
    const int jj = r % period;
    const u64 q = T.mods[jj < L ? jj : T.k - 1].q;
    const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    ulonglong2 acc = addin ? reinterpret_cast<const ulonglong2 *>(addin + (size_t)r * n)[w] : make_ulonglong2(0, 0);
    for (int ch = 0; ch < chunks; ++ch) {
        const ulonglong2 v = reinterpret_cast<const ulonglong2 *>(partial + ((size_t)ch * rows + r) * n)[w];
        acc.x = addmod(acc.x, v.x, q);
        acc.y = addmod(acc.y, v.y, q);
    }
    // the accumulator scratch (period L + 1) holds a data prime's FP64-policy row as doubles (MacF::result_data)
    if (period == L + 1 && jj < L && T.modsf[jj].q != 0.0)
        acc = make_ulonglong2(ArithF64::raw(ArithF64::from_u64(acc.x)), ArithF64::raw(ArithF64::from_u64(acc.y)));
    reinterpret_cast<ulonglong2 *>(out + (size_t)r * n)[w] = acc;
}

// double hoisting, device side: decompose ct_new once (item 0 of `src_item`: c_in = ct_new, no permutation), partial
// sums of the rotations, then -- after the caller has reduced the partials into scr.acc / cbuf -- the single mod-down
template <int LOGN>
static hipError_t launch_lt2_decompose_t(const DevTables &T, int L, const KsItem *src_item, const KsItem *rot_items,
                                         int nrot, const KsScratch &scr, const u64 *ct_new, u64 *partial_s,
                                         u64 *partial_c0, u64 *cbuf, hipStream_t s)
{
    using SC = SplitCfg<LOGN>;
    const size_t lds = SC::LDS_BYTES;
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        set_lds(ks_intt_digits_kernel<LOGN>, lds);
        set_lds(ks_ntt_digits_kernel<LOGN>, lds);
        set_lds(ks_moddown_intt_kernel<LOGN>, lds);
        set_lds(ks_moddown_finish_kernel<LOGN>, lds);
    }
    const int chunks = (nrot + LT2_CHUNK - 1) / LT2_CHUNK;
    hipLaunchKernelGGL((ks_intt_digits_kernel<LOGN>), dim3(split_grid(L)), dim3(SC::T), lds, s, T, src_item, L, 0, 1, L,
                       scr);
    hipLaunchKernelGGL((ks_ntt_digits_kernel<LOGN>), dim3(group_grid(L, L)), dim3(SC::T), lds, s, T, L, L, 0, 0, scr);
    hipLaunchKernelGGL(lt2_mac_kernel, dim3(SC::N / 2 / 256, L + 1, chunks), dim3(256), 0, s, T, rot_items, L, nrot, scr,
                       ct_new + (size_t)L * SC::N, partial_s);
    hipLaunchKernelGGL(lt2_c0_kernel, dim3(SC::N / 2 / 256, L, chunks), dim3(256), 0, s, T, rot_items, L, nrot, ct_new,
                       partial_c0);
    // S -> scr.acc [2][L+1][N];  C0 += partials, in place on poly 0 of cbuf
    hipLaunchKernelGGL(lt2_reduce_kernel, dim3(SC::N / 2 / 256, 2 * (L + 1)), dim3(256), 0, s, T, L, L + 1, 2 * (L + 1),
                       chunks, partial_s, (const u64 *)nullptr, scr.acc);
    hipLaunchKernelGGL(lt2_reduce_kernel, dim3(SC::N / 2 / 256, L), dim3(256), 0, s, T, L, L, L, chunks, partial_c0,
                       (const u64 *)cbuf, cbuf);
    return hipGetLastError();
}
// single mod-down of scr.acc ([2][L+1][N]) with add-in item->c_in ([2][L][N]) into item->c_out (relin-style epilogue)
template <int LOGN>
static hipError_t launch_lt2_moddown_t(const DevTables &T, int L, const KsItem *item, const KsScratch &scr, hipStream_t s)
{
    using SC = SplitCfg<LOGN>;
    const size_t lds = SC::LDS_BYTES;
    hipLaunchKernelGGL((ks_moddown_intt_kernel<LOGN>), dim3(split_grid(2)), dim3(SC::T), lds, s, T, L, 2, scr);
    hipLaunchKernelGGL((ks_moddown_finish_kernel<LOGN>), dim3(group_grid(2, L)), dim3(SC::T), lds, s, T, item, L, 1, 2,
                       scr);
    return hipGetLastError();
}

// -DHEFX_ONLY_LOGN=14: development builds that instantiate one ring size only (a fifth of the compile time)
#ifdef HEFX_ONLY_LOGN
#define HEFX_DISPATCH_SPLIT(logn, CALL) \
    if ((logn) == HEFX_ONLY_LOGN) return CALL(HEFX_ONLY_LOGN); \
    return hipErrorInvalidValue;
#else
#define HEFX_DISPATCH_SPLIT(logn, CALL)       \
    switch (logn) {                           \
        case 11: return CALL(11);             \
        case 12: return CALL(12);             \
        case 13: return CALL(13);             \
        case 14: return CALL(14);             \
        case 15: return CALL(15);             \
        default: return hipErrorInvalidValue; \
    }
#endif

hipError_t launch_lt2_decompose(const DevTables &T, int L, const KsItem *src_item, const KsItem *rot_items, int nrot,
                                const KsScratch &scr, const u64 *ct_new, u64 *partial_s, u64 *partial_c0, u64 *cbuf,
                                hipStream_t s)
{
#define CALL(LN) launch_lt2_decompose_t<LN>(T, L, src_item, rot_items, nrot, scr, ct_new, partial_s, partial_c0, cbuf, s)
    HEFX_DISPATCH_SPLIT(T.logn, CALL)
#undef CALL
}
hipError_t launch_lt2_moddown(const DevTables &T, int L, const KsItem *item, const KsScratch &scr, hipStream_t s)
{
#define CALL(LN) launch_lt2_moddown_t<LN>(T, L, item, scr, s)
    HEFX_DISPATCH_SPLIT(T.logn, CALL)
#undef CALL
}
int lt2_chunk() { return LT2_CHUNK; }

int ks_small_max() { return KS_SMALL_MAX; }
hipError_t launch_keyswitch_chunk(const DevTables &T, int L, int n, const KsItem *batch, bool relin,
                                  const KsScratch &scr, int sub, bool alias, const KsItem *small_items,
                                  int quarter, hipStream_t s, KsProf *prof, int nsrc)
{
    KsSmallItems sm;
    const KsSmallItems *small = nullptr;
    if (small_items && n <= KS_SMALL_MAX && !nsrc && !alias) {
        for (int i = 0; i < n; ++i) sm.it[i] = small_items[i];
        for (int i = n; i < KS_SMALL_MAX; ++i) sm.it[i] = KsItem{};
        small = &sm;
    }
    if (T.logn < 12) quarter = 0;  // quarter rows of N = 2048 would be half-wave workgroups
#define CALL(LN) launch_keyswitch_chunk_t<LN>(T, L, n, batch, relin, scr, sub, alias, small, quarter, s, prof, nsrc)
    HEFX_DISPATCH_SPLIT(T.logn, CALL)
#undef CALL
}

hipError_t launch_flip_rows(const DevTables &T, const uint32_t *d_ginv, int count, u64 *rows, hipStream_t s)
{
    hipLaunchKernelGGL(flip_mask_kernel, dim3((1u << T.logn) / 256, T.k, count), dim3(256), 0, s, T, d_ginv, rows);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// K8: rescale_to_next (App. A.9): per poly 1 INTT + (L-1) NTT.  Two divisions by the dropped prime q_l, selected per
// call: rounded == 0  floor, out_j = (c_j - [c_l]_(q_j)) * q_l^-1          (SEAL 3.4.x as SURVEY App. A.9 states it)
//       rounded != 0  round, out_j = (c_j - ([c_l + q_l/2]_(q_l) mod q_j - (q_l/2 mod q_j))) * q_l^-1   (SEAL >= 3.5,
//                     divide_and_round_q_last) -- the same plumbing as the key-switch mod-down by P.
// ------------------------------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__(SplitCfg<LOGN>::T, KsWaves<LOGN>::INV) void rs_intt_kernel(DevTables T, int L, int size, int rows,
                                                                       const u64 *in, const u64 *const *__restrict__ tab,
                                                                       u64 *d, int rounded)
{
    using SC = SplitCfg<LOGN>;
    using C = typename SC::C;
    extern __shared__ __align__(16) u64 lds[];
    int p, h;
    split_decode(blockIdx.x, p, h);
    if (p >= rows) return;  // p runs over count*size polys
    const int t = threadIdx.x;
    // poly p of the batch: contiguous ciphertexts, or ciphertext p / size of a pointer table
    const u64 *__restrict__ base = tab ? tab[p / size] + (size_t)(p % size) * L * SC::N : in + (size_t)p * L * SC::N;
    const ulonglong2 *__restrict__ src = reinterpret_cast<const ulonglong2 *>(base + (size_t)(L - 1) * SC::N);
    u64 v[16];
    split_inv<LOGN, KsWaves<LOGN>::NB_INV>(v, src, lds, ntt_tables(T, L - 1), T.mods[L - 1], T.modsf[L - 1], t, h);
    u64 *__restrict__ dd = d + (size_t)p * SC::N + (size_t)h * SC::H;
    const u64 ql = T.mods[L - 1].q, half = rounded ? ql >> 1 : 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) dd[C::idx_nat(t, r)] = csub(v[r] + half, ql);
}

template <int LOGN>
__global__ __launch_bounds__(SplitCfg<LOGN>::T, KsWaves<LOGN>::FWD) void rs_finish_kernel(DevTables T, int L, int size, int count,
                                                                         int rows, const u64 *in,
                                                                         const u64 *const *__restrict__ tab,
                                                                         const u64 *d, u64 *out, int rounded)
{
    using SC = SplitCfg<LOGN>;
    using C = typename SC::C;
    extern __shared__ __align__(16) u64 lds[];
    int p, h;
    split_decode(blockIdx.x, p, h);
    if (p >= rows) return;  // p = poly*(L-1) + j
    const int t = threadIdx.x;
    const int poly = p / (L - 1), j = p % (L - 1);
    const ModConst mc = T.mods[j];
    const u64 q = mc.q, ql = T.mods[L - 1].q;
    const ulonglong2 qinv = T.invmod[(size_t)(L - 1) * T.k + j];
    const u64 *__restrict__ dd = d + (size_t)poly * SC::N;
    u64 v[16];
    // rounded: subtract (q_l/2 mod q_j) from the reduced remainder, in the row's arithmetic policy
    const InMode mode = {ql > q, T.modsf[L - 1].q == 0.0, rounded != 0, T.halfmod[(size_t)(L - 1) * T.k + j]};
    // (the rescale keeps column t at N = 8192: two registers past 128 there, and no second load batch to trade them for)
    const int tl = SC::N >= 16384 ? HEFX_EO_LANE(SC, t) : t;
    auto ld = [&](int r, u64 &x, u64 &y) {
        const uint32_t e = eo_nat<SC>(tl, r);
        x = dd[e];
        y = dd[e + SC::H / 2];
    };
    split_fwd<LOGN, KsWaves<LOGN>::NB_FWD>(v, ld, mode, lds, ntt_tables(T, j), mc, T.modsf[j], t, h, tl);
    const size_t off = (size_t)h * SC::H;
    const int ci = poly / size, pi = poly % size;
    const u64 *__restrict__ src = (tab ? tab[ci] + (size_t)pi * L * SC::N : in + (size_t)poly * L * SC::N) + (size_t)j * SC::N + off;
    u64 *__restrict__ dst = (tab ? const_cast<u64 *>(tab[count + ci]) + (size_t)pi * (L - 1) * SC::N
                                 : out + (size_t)poly * (L - 1) * SC::N) + (size_t)j * SC::N + off;
    // operands in two batches of eight (all sixteen at once spilled next to the two integer forward policies)
#pragma unroll
    for (int g = 0; g < 16; g += 8) {
        u64 a[8];
#pragma unroll
        for (int r = 0; r < 8; r += 2) {  // (r, r+1) are one 16-byte record
            const ulonglong2 rec = gld16(src + C::idx_io(t, g + r));
            a[r] = rec.x, a[r + 1] = rec.y;
        }
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            ulonglong2 o;
            o.x = csub(shoup_lazy(submod(a[r], v[g + r], q), qinv.x, qinv.y, mc.nq), q);
            o.y = csub(shoup_lazy(submod(a[r + 1], v[g + r + 1], q), qinv.x, qinv.y, mc.nq), q);
            gst16(dst + C::idx_io(t, g + r), o);
        }
        HEFX_STAGE_FENCE();
    }
}

template <int LOGN>
static hipError_t launch_rescale_t(const DevTables &T, int L, int size, int count, const u64 *in, u64 *out,
                                   const u64 *const *tab, u64 *scratch_d, bool rounded, hipStream_t s)
{
    using SC = SplitCfg<LOGN>;
    const size_t lds = SC::LDS_BYTES;
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        set_lds(rs_intt_kernel<LOGN>, lds);
        set_lds(rs_finish_kernel<LOGN>, lds);
    }
    const int polys = size * count;
    hipLaunchKernelGGL((rs_intt_kernel<LOGN>), dim3(split_grid(polys)), dim3(SC::T), lds, s, T, L, size, polys, in, tab,
                       scratch_d, rounded ? 1 : 0);
    hipLaunchKernelGGL((rs_finish_kernel<LOGN>), dim3(split_grid(polys * (L - 1))), dim3(SC::T), lds, s, T, L, size,
                       count, polys * (L - 1), in, tab, scratch_d, out, rounded ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_rescale(const DevTables &T, int L, int size, int count, const u64 *in, u64 *out,
                          const u64 *const *tab, u64 *scratch_d, bool rounded, hipStream_t s)
{
#define CALL(LN) launch_rescale_t<LN>(T, L, size, count, in, out, tab, scratch_d, rounded, s)
    HEFX_DISPATCH_SPLIT(T.logn, CALL)
#undef CALL
}

// ------------------------------------------------------------------------------------------------
// Stand-alone split NTT, out of place (src -> dst, natural layouts on both sides).  Used for N = 32768, whose
// rows do not fit one workgroup's LDS; smaller N use the in-place single-workgroup kernels.
// ------------------------------------------------------------------------------------------------
template <int LOGN, bool INV>
__global__ __launch_bounds__(SplitCfg<LOGN>::T, KsWaves<LOGN>::FWD) void ntt_split_rows_kernel(DevTables T, const u64 *src, u64 *dst,
                                                                              int rows, int nrows, int mod_first)
{
    using SC = SplitCfg<LOGN>;
    using C = typename SC::C;
    extern __shared__ __align__(16) u64 lds[];
    int p, h;
    split_decode(blockIdx.x, p, h);
    if (p >= rows) return;
    const int t = threadIdx.x;
    const int m = mod_first + p % nrows;
    const ModConst mc = T.mods[m];
    const u64 *__restrict__ s = src + (size_t)p * SC::N;
    u64 *__restrict__ o = dst + (size_t)p * SC::N;
    u64 v[16];
    if (!INV) {
        auto ld = [&](int r, u64 &x, u64 &y) {
            x = s[C::idx_nat(t, r)];
            y = s[C::idx_nat(t, r) + SC::H];
        };
        const InMode mode = {false, false, false, 0};
        split_fwd<LOGN>(v, ld, mode, lds, ntt_tables(T, m), mc, T.modsf[m], t, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[(size_t)h * SC::H + C::idx_io(t, r)] = v[r];
    } else {
        split_inv<LOGN>(v, reinterpret_cast<const ulonglong2 *>(s), lds, ntt_tables(T, m), mc, T.modsf[m], t, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[2 * C::idx_nat(t, r) + h] = v[r];
    }
}

hipError_t launch_ntt_split15(const DevTables &T, bool inverse, const u64 *src, u64 *dst, int npoly, int nrows,
                              int mod_first, hipStream_t s)
{
    using SC = SplitCfg<15>;
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        set_lds(ntt_split_rows_kernel<15, false>, SC::LDS_BYTES);
        set_lds(ntt_split_rows_kernel<15, true>, SC::LDS_BYTES);
    }
    const int rows = npoly * nrows;
    if (inverse)
        hipLaunchKernelGGL((ntt_split_rows_kernel<15, true>), dim3(split_grid(rows)), dim3(SC::T), SC::LDS_BYTES, s, T,
                           src, dst, rows, nrows, mod_first);
    else
        hipLaunchKernelGGL((ntt_split_rows_kernel<15, false>), dim3(split_grid(rows)), dim3(SC::T), SC::LDS_BYTES, s,
                           T, src, dst, rows, nrows, mod_first);
    return hipGetLastError();
}

#ifdef HEFX_STAMP
}  // namespace hefx
extern "C" __attribute__((visibility("default"))) int hefx_debug_stamps(unsigned long long *out, int clear)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(hefx::hefx_stamp_buf), sizeof(hefx::hefx_stamp_buf)) != hipSuccess) return -1;
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(hefx::hefx_stamp_buf)) != hipSuccess) return -2;
        if (hipMemset(p, 0, sizeof(hefx::hefx_stamp_buf)) != hipSuccess) return -3;
    }
    return (int)(sizeof(hefx::hefx_stamp_buf) / 8);
}
namespace hefx {
#endif

// HIP loads a translation unit's code object at its first kernel launch (milliseconds); hefx_context_create pays
// that once, up front, instead of the first encode / rotation / encryption of a program.
__global__ void warm_keyswitch_kernel() {}
hipError_t warm_keyswitch(hipStream_t s)
{
    hipLaunchKernelGGL(warm_keyswitch_kernel, dim3(1), dim3(64), 0, s);
    return hipGetLastError();
}

}  // namespace hefx

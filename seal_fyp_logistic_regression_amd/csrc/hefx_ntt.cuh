// hefx_ntt.cuh -- negacyclic NTT / inverse NTT cores for one RNS row (or half row) held by ONE workgroup.
//
// Semantics (SEAL 3.4.5 util/smallntt): twiddle table tw[bitrev(i)] = psi^i with
// psi the minimal primitive 2N-th root; forward = Cooley-Tukey, natural in -> bit-reversed out,
// out[i] = a(psi^(2*bitrev(i)+1)); inverse = Gentleman-Sande with N^-1 folded into the last stage.
//
// gfx950 mapping: N/16 threads, 16 coefficients per thread in registers; four radix-2 stages per pass
// are done in registers (radix-16), passes exchange through LDS (one barrier per exchange: a thread
// only ever overwrites the LDS words it read itself).  N*8*(17/16) bytes of LDS.  Twiddles of the first
// pass are workgroup-uniform and of wide passes wave-uniform (scalar loads); later passes load a 16-byte
// twiddle record per butterfly from L2.
//
// Two arithmetic policies, chosen per RNS prime (all results are canonical residues, so both give the same
// bits as SEAL's CPU path):
//   ArithU64  any prime < 2^61: Harvey-style lazy butterflies ([0,8q) forward, [0,4q) inverse) with Shoup twiddles
//             {w, floor(w*2^64/q)} and an under-estimated quotient (no carry chain in the 64x64 high word):
//             9 integer multiplies + ~15 other instructions per butterfly, every one a 4-6 cycle issue slot.
//   ArithF64  primes < 2^41 (SEAL's 30..40-bit data primes): coefficients are kept as exact integers in
//             doubles and x*w mod q is computed EXACTLY with FMA: h = x*w, l = fma(x,w,-h) (exact product),
//             c = rint(h * (1/q)), t = fma(-c,q,h) + l, |t| < 0.52q; 6 fp64 instructions (~32 cycles measured),
//             8-byte twiddles (no per-twiddle quotient),
//             no range corrections in the forward transform (growth 0.52q per stage), one re-centering per
//             radix-16 pass in the inverse.  Exactness argument in DESIGN.md ("FP64 modmul").
#pragma once
#include <type_traits>

#include "hefx_modarith.cuh"

// -DHEFX_STAMP=1|2: development builds (tools/stamp_timeline.py) in which thread 0 of every workgroup of the small-batch
// kernels records the 100 MHz wall clock at its phase boundaries (2: after draining its outstanding memory operations, so
// that a stamp means "everything before this has arrived").  Never defined in the product build.
#ifdef HEFX_STAMP
namespace hefx {
static __device__ u64 hefx_stamp_buf[8 * 1024 * 16];
__device__ __forceinline__ int &hefx_stamp_kid()
{
    __shared__ int kid;
    return kid;
}
}  // namespace hefx
#define HEFX_STAMP_KERNEL(k) do { if (threadIdx.x == 0) hefx_stamp_kid() = (k); } while (0)
#define HEFX_STAMP_AT(id)                                                                                            \
    do {                                                                                                             \
        if (HEFX_STAMP > 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                              \
        const unsigned hefx_wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                   \
        if (threadIdx.x == 0 && hefx_wg_ < 1024)                                                                     \
            hefx_stamp_buf[((size_t)hefx_stamp_kid() * 1024 + hefx_wg_) * 16 + (id)] = wall_clock64();               \
    } while (0)
#else
#define HEFX_STAMP_KERNEL(k) do { } while (0)
#define HEFX_STAMP_AT(id) do { } while (0)
#endif

// Scheduling fence between radix-2 stages: keeps hipcc from hoisting every twiddle load of a pass (60 VGPRs)
// above the first butterfly, which pushed the kernels past 128 VGPRs and into scratch spills.
#ifndef HEFX_STAGE_FENCE
#define HEFX_STAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

namespace hefx {

// How the raw 64-bit words a forward-split loader delivers become transform inputs: optional reduction modulo the
// row's prime (the word comes from a wider prime) and optional subtraction of a constant (mod-down: - (P/2 mod q)).
struct InMode {
    bool red_int;  // integer policy: the word may exceed q
    bool red_f64;  // FP64 policy: the word may exceed 2^52 (it comes from a prime that is not an FP64 prime)
    bool has_sub;
    u64 sub;       // canonical residue to subtract
    bool below_2_61 = true;  // the words are residues of an admissible prime (< 2^61): the one-fma reduction applies
    bool lt2q = false;       // the words are below 2 q of the row's prime: one conditional subtraction reduces them
};

template <int LOGN>
struct NttCfg {
    static constexpr int N = 1 << LOGN;
    static constexpr int T = N / 16;   // threads per row
    static constexpr int FP = LOGN / 4;  // full radix-16 passes
    static constexpr int R = LOGN % 4;   // stages of the remainder pass
    static constexpr int G = 1 << R;     // contiguous coefficients per remainder group
    static constexpr int NG = 16 / G;    // groups per thread in the remainder pass
    static constexpr int PSH = 4 + R;    // pad G words every 16*G words
    static constexpr int LDS_WORDS = N + (N >> 4);
    __device__ static __forceinline__ int phys(int idx) { return idx + ((idx >> PSH) << R); }
    // phys() is AFFINE along every access pattern of the cores, which lets the sixteen LDS accesses of a pass share one
    // address register and differ in the instruction's immediate offset (the compiler cannot prove the absence of
    // carries in phys(base + S*e) and otherwise spends ~2 VALU instructions per access on it):
    //   pass with element stride S = 2^LOGS:  LOGS >= PSH -> the pad count advances by S >> PSH per element;
    //                                         LOGS == R   -> all sixteen words lie in one pad block (16*S = 2^PSH)
    //   (LOGS = LOGN - 4(p+1) is congruent to R modulo 4, so these are the only cases)
    //   remainder pass, group c of thread t = words (t + T*c)*G + e: the pad count is (t >> 4) + (T >> 4)*c
    __host__ __device__ static constexpr int pass_stride(int logs)
    {
        return logs >= PSH ? (1 << logs) + (((1 << logs) >> PSH) << R) : (1 << logs);
    }
    static constexpr int REM_STRIDE = T * G + ((T >> 4) << R);
    static_assert(T % 16 == 0, "remainder-pass addressing assumes T is a multiple of 16");
    // coefficient index of register r in the pass-0 layout (coalesced: lane-consecutive words)
    __device__ static __forceinline__ int idx_nat(int t, int r) { return t + T * r; }
    // coefficient index of register r in the layout the forward transform ends in
    __device__ static __forceinline__ int idx_out(int t, int r)
    {
        if (R == 0) return t * 16 + r;
        return (t + T * (r >> R)) * G + (r & (G - 1));
    }
    // NTT-domain index of register r at the MEMORY side of the split transforms (split_fwd* outputs; split_inv* loads
    // records the same way).  For R > 0 that is idx_out: G >= 2 adjacent words per lane, lanes adjacent -- coalesced.
    // For R == 0 idx_out would put lanes 128 bytes apart (sixteen consecutive words per thread): every global access
    // touches 64 cache lines and the operand streams of the resident workgroups evict each other from L1 between a
    // line's uses (measured at N = 8192: mod-down finish 384 -> 134 us per chunk once its accesses were coalesced).
    // There the values take one more trip through LDS into the pair layout of R = 1.
    __device__ static __forceinline__ int idx_io(int t, int r)
    {
        if (R == 0) return (t + T * (r >> 1)) * 2 + (r & 1);
        return idx_out(t, r);
    }
};

// ------------------------------------------------------------------------------------------------
// policy: 64-bit integers, Harvey/Shoup
// ------------------------------------------------------------------------------------------------
// L16 (round 3): the forward transform of a prime below 2^60 -- every prime SEAL can produce -- keeps its values in
// [0,16q) and makes the conditional subtraction (by 8q) in every OTHER stage only:
//   even stage (none):  inputs < 12q, x' = x + t < 16q, y' = x + 4q - t < 16q            (t = w*y lazy, < 4q)
//   odd stage  (by 8q): inputs < 16q -> a < 8q, x' = a + t < 12q, y' = a + 4q - t < 12q
// 16q <= 2^64 needs q <= 2^60.  Four instructions fewer in half the stages: 80.0 -> 71.7 SIMD cycles per wave-butterfly
// in isolation (tools/ubench_bfly.hip form 4, profiles/r03_butterfly_ubench.txt).  The 61-bit primes the engine also
// admits keep the [0,8q) form (L16 = false); the inverse transform has no such slack (its sums double per stage).
// Used where it measured faster in the kernels -- the digit transforms of the key switch (ks_ntt_digits: -3 % VALU
// instructions at C3, -10 % kernel time at N = 32768); the mod-down epilogue kernels got slower with a second integer
// path (registers: 111 -> 129 VGPRs at N = 8192, one wave per SIMD fewer) and keep the [0,8q) form.
template <bool L16>
struct ArithU64T {
    typedef u64 V;
    typedef ulonglong2 TW;  // {w, floor(w*2^64/q)}
    // Lazy ranges (q < 2^61, so 8q < 2^64): forward values live in [0,8q), inverse values in [0,4q).  The twiddle
    // product is shoup_lazy4 -- Shoup's multiplication with the quotient under-estimated by up to 2 (result in [0,4q)
    // instead of [0,2q)), which drops the carry chain of the exact 64x64 high word: ~24 instead of ~31 instructions per
    // butterfly, each of which costs a 4-6 cycle issue slot on gfx950.  All stored results are
    // canonical, so the bits are those of any exact arithmetic.
    struct Ctx {
        u64 q, two_q, four_q;
        u64 nq, n2q, n4q, n8q;  // 2^64 - q, - 2q, - 4q, - 8q (shoup_lazy4, csubn)
        u64 ninv, ninv_s, ilw, ilw_s;
    };
    __device__ static __forceinline__ Ctx make(const ModConst &mc)
    {
        Ctx c;
        c.q = mc.q;
        c.two_q = mc.q << 1;
        c.four_q = mc.q << 2;
        c.nq = mc.nq;
        c.n2q = mc.nq << 1;
        c.n4q = mc.nq << 2;
        c.n8q = mc.nq << 3;
        c.ninv = mc.ninv;
        c.ninv_s = mc.ninv_s;
        c.ilw = mc.ilw;
        c.ilw_s = mc.ilw_s;
        return c;
    }
    // forward butterfly of stage `stage` (0 = the first stage of the core; a compile-time constant once the cores'
    // loops are unrolled).  L16 = false: inputs/outputs in [0,8q).  L16: see the policy's header.
    __device__ static __forceinline__ void ct(V &x, V &y, const TW &w, const Ctx &c, int stage)
    {
        u64 a;
        if (L16)
            a = (stage & 1) ? csubn(x, c.n8q) : x;    // [0,8q) or [0,12q)
        else
            a = csubn(x, c.n4q);                      // [0,4q)
        u64 t = shoup_lazy4(y, w.x, w.y, c.nq);          // [0,4q), y any 64-bit word
        x = a + t;
        y = a + c.four_q - t;
    }
    // first stage of a split forward transform: keep X = x + w*y (h=0) or Y = x - w*y (h=1); x,y canonical.  The sign
    // goes into the twiddle once per workgroup -- (q - w, ~w') is the Shoup pair of -w: floor((q-w) 2^64 / q) =
    // 2^64 - 1 - floor(w 2^64 / q) because q does not divide w 2^64 -- instead of a select per element.
    __device__ static __forceinline__ TW half_twiddle(const TW &w, const Ctx &c, int h)
    {
        TW r;
        r.x = h ? c.q - w.x : w.x;
        r.y = h ? ~w.y : w.y;
        return r;
    }
    __device__ static __forceinline__ V ct_half(V x, V y, const TW &wh, const Ctx &c)
    {
        return x + shoup_lazy4(y, wh.x, wh.y, c.nq);    // < 5q
    }
    // inverse butterfly, inputs/outputs in [0,4q)
    __device__ static __forceinline__ void gs(V &x, V &y, const TW &w, const Ctx &c)
    {
        u64 s = csubn(x + y, c.n4q);
        u64 d = x + c.four_q - y;                       // (0,8q)
        x = s;
        y = shoup_lazy4(d, w.x, w.y, c.nq);
    }
    // last inverse stage: exact Shoup products (outputs in [0,2q), what inv_finish expects)
    __device__ static __forceinline__ void gs_last(V &x, V &y, const Ctx &c)
    {
        u64 s = x + y;                                  // < 8q: fine for Shoup (any 64-bit word)
        u64 d = x + c.four_q - y;
        x = shoup_lazy(s, c.ninv, c.ninv_s, c.nq);
        y = shoup_lazy(d, c.ilw, c.ilw_s, c.nq);
    }
    // first stage of a split inverse transform on a pair (a0,a1) of words below 2q: sum (h=0, < 4q) or twiddled difference
    __device__ static __forceinline__ V gs_half_sum(V a0, V a1, const Ctx &) { return a0 + a1; }
    // (a0, a1 below 2q: the key MAC's integer-policy results are not canonical, hefx_keyswitch.hip MacW / MacL result<true>)
    __device__ static __forceinline__ V gs_half_diff(V a0, V a1, const TW &w, const Ctx &c)
    {
        return shoup_lazy4(a0 + c.two_q - a1, w.x, w.y, c.nq);
    }
    template <int NV>
    __device__ static __forceinline__ void inv_pass_begin(V (&)[NV], const Ctx &) {}
    // quarter-row helpers (hefx_ntt8.cuh).  Forward: second stage on values < 5q, result < 8q.  Inverse: sum / twiddled
    // difference of two values of the inverse range [0,4q), results back in [0,4q).
    __device__ static __forceinline__ V ct_sel(V x, V y, const TW &wh, const Ctx &c)  // wh = half_twiddle(w, c, h)
    {
        // L16: x < 5q as it is; 5q + 4q < 12q, a valid input of the core's stage 0
        return (L16 ? x : csubn(x, c.n4q)) + shoup_lazy4(y, wh.x, wh.y, c.nq);
    }
    __device__ static __forceinline__ V inv_add(V x, V y, const Ctx &c) { return csubn(x + y, c.n4q); }
    __device__ static __forceinline__ V inv_sub_mul(V x, V y, const TW &w, const Ctx &c)
    {
        return shoup_lazy4(x + c.four_q - y, w.x, w.y, c.nq);
    }
    __device__ static __forceinline__ V from_u64(u64 x) { return x; }
    // a working value as a 64-bit word for scratch memory, and back (the pair kernels hand UNFINISHED values from one launch
    // to the next: hefx_keyswitch.hip, ks_pair_*)
    __device__ static __forceinline__ u64 raw(V x) { return x; }
    __device__ static __forceinline__ V unraw(u64 b) { return b; }
    static constexpr bool IS_F64 = false;
    __device__ static __forceinline__ bool fast_wide(const Ctx &) { return false; }
    // RED: 0 none, 1 Barrett (any 64-bit word), 3 the word is below 2q: one conditional subtraction -- or, when a constant
    // is subtracted anyway (mod-down: - (P/2 mod q)), no reduction at all: x + (q - sub) < 3q is a valid first-stage
    // operand (ct_half: < 3q + 4q = 7q < 8q; quarter rows: ct_sel of two such values < 11q < 12q, the 16q core's entry
    // bound), one addition instead of a conditional and a modular subtraction per word
    template <int RED>
    __device__ static __forceinline__ V input(u64 x, const InMode &m, const Ctx &c, const ModConst &mc)
    {
        if (RED == 3) {
            if (m.has_sub) return x + (c.q - m.sub);
            return csubn(x, c.nq);
        }
        if (RED) x = barrett64(x, mc.q, mc.r1);
        if (m.has_sub) x = submod(x, m.sub, c.q);
        return x;
    }
    __device__ static __forceinline__ u64 fwd_finish(V x, const Ctx &c)
    {
        if (L16) x = csubn(x, c.n8q);  // < 16q -> < 8q
        return csubn(csubn(csubn(x, c.n4q), c.n2q), c.nq);
    }
    // what the key MAC reads from scratch for a row of this policy (hefx_keyswitch.hip, MacL / MacW): canonical words
    __device__ static __forceinline__ u64 mac_operand(V x, const Ctx &c) { return fwd_finish(x, c); }
    // ... or words below 2q / below 4q where the MAC's policy has the headroom (mac_x_slack, hefx_keyswitch.hip): one or two
    // conditional subtractions less per word
    template <int SLACK>  // 1: < 2q, 2: < 4q
    __device__ static __forceinline__ u64 mac_operand_lazy(V x, const Ctx &c)
    {
        if (L16) x = csubn(x, c.n8q);
        x = csubn(x, c.n4q);
        return SLACK >= 2 ? x : csubn(x, c.n2q);
    }
    // key-switch mod-down epilogue (App. A.8) from the UNFINISHED transform value f (< 8q):
    // ((acc - f) * P^-1 + sadd) [* pt] mod q, canonical
    __device__ static __forceinline__ u64 moddown(V f, u64 acc, u64 sadd, u64 pt, bool has_pt, const Ctx &c,
                                                  const ulonglong2 &pinv, const ModConst &mc)
    {
        if (L16) f = csubn(f, c.n8q);                    // < 16q -> < 8q
        u64 z = acc + c.four_q - csubn(f, c.n4q);      // acc < 2q (MAC result<true>): < 6q (no 9q intermediate: primes may reach 2^61)
        z = shoup_lazy(z, pinv.x, pinv.y, c.nq) + sadd;   // < 3q
        if (has_pt) return mulmod(z, pt, mc);            // product < 3q*q < q*2^64: Barrett128 gives [0,q)
        return csubn(csubn(z, c.n2q), c.nq);
    }
    __device__ static __forceinline__ u64 inv_finish(V x, const Ctx &c) { return csubn(x, c.nq); }
};
typedef ArithU64T<false> ArithU64;   // any prime below 2^61; every inverse transform
typedef ArithU64T<true> ArithU64L;   // forward transforms of primes below 2^60
// f(policy tag) with the forward integer policy the modulus admits (block-uniform)
template <class F>
__device__ __forceinline__ void fwd_int_dispatch(const ModConst &mc, const F &f)
{
#ifdef HEFX_NO_L16  // A/B knob (tools/build_variant.sh): the [0,8q) butterfly everywhere
    f(ArithU64{});
#else
    if (mc.q >> 60)
        f(ArithU64{});
    else
        f(ArithU64L{});
#endif
}

// ------------------------------------------------------------------------------------------------
// policy: exact integers in doubles, FMA modmul (q < 2^41)
// ------------------------------------------------------------------------------------------------
struct ArithF64 {
    typedef double V;
    typedef double TW;  // w only: the quotient estimate is rint(RN(y*w) * RN(1/q)), no per-twiddle w/q needed
    struct Ctx {
        double q, qinv;
        double ninv, ilw;
        double c32, c40;
    };
    __device__ static __forceinline__ Ctx make(const ModConstF &mf)
    {
        Ctx c;
        c.q = mf.q;
        c.qinv = mf.qinv;
        c.ninv = mf.ninv;
        c.ilw = mf.ilw;
        c.c32 = mf.c32;
        c.c40 = mf.c40;
        return c;
    }
    // y*w mod q, exact, result in (-0.52q, 0.52q); y any integer with |y| < 2^45, w integer with |w| < 2^41:
    // h + l = y*w exactly; k = rint(h/q) up to 3*2^-53 relative error of the argument (< 0.02 absolute); h - k*q is an
    // integer below 2^41 in magnitude, hence exact in the FMA; adding l is exact for the same reason.
    __device__ static __forceinline__ double mm(double y, double w, const Ctx &c)
    {
        const double h = y * w;
        const double l = __builtin_fma(y, w, -h);
        const double k = __builtin_rint(h * c.qinv);
        const double s = __builtin_fma(-k, c.q, h);
        return s + l;
    }
    // re-centre: x - q*rint(x/q), exact, result in [-0.5q, 0.5q]
    __device__ static __forceinline__ double red(double x, const Ctx &c)
    {
        return __builtin_fma(-__builtin_rint(x * c.qinv), c.q, x);
    }
    __device__ static __forceinline__ void ct(V &x, V &y, const TW &w, const Ctx &c, int /*stage*/)
    {
        const double t = mm(y, w, c);
        const double a = x;
        x = a + t;
        y = a - t;
    }
    __device__ static __forceinline__ TW half_twiddle(const TW &w, const Ctx &, int h) { return h ? -w : w; }
    __device__ static __forceinline__ V ct_half(V x, V y, const TW &wh, const Ctx &c) { return x + mm(y, wh, c); }
    __device__ static __forceinline__ void gs(V &x, V &y, const TW &w, const Ctx &c)
    {
        const double s = x + y, d = x - y;
        x = s;
        y = mm(d, w, c);
    }
    __device__ static __forceinline__ void gs_last(V &x, V &y, const Ctx &c)
    {
        const double s = x + y, d = x - y;
        x = mm(s, c.ninv, c);
        y = mm(d, c.ilw, c);
    }
    __device__ static __forceinline__ V gs_half_sum(V a0, V a1, const Ctx &) { return a0 + a1; }
    __device__ static __forceinline__ V gs_half_diff(V a0, V a1, const TW &w, const Ctx &c)
    {
        return mm(a0 - a1, w, c);
    }
    // sums double per inverse stage: re-centre the 16 registers once per radix-16 pass (|x| <= 16*0.52q after it)
    template <int NV>
    __device__ static __forceinline__ void inv_pass_begin(V (&v)[NV], const Ctx &c)
    {
#pragma unroll
        for (int e = 0; e < NV; ++e) v[e] = red(v[e], c);
    }
    __device__ static __forceinline__ V ct_sel(V x, V y, const TW &wh, const Ctx &c) { return x + mm(y, wh, c); }
    __device__ static __forceinline__ V inv_add(V x, V y, const Ctx &) { return x + y; }
    __device__ static __forceinline__ V inv_sub_mul(V x, V y, const TW &w, const Ctx &c) { return mm(x - y, w, c); }
    // u64 <-> double for integers in [0, 2^52) by exponent splicing: one integer OR/AND on the high word plus one
    // v_add_f64, instead of the v_cvt/v_ldexp/v_trunc/v_floor sequences of a generic conversion
    __device__ static __forceinline__ V from_u64(u64 x)
    {
        return __longlong_as_double((long long)(x | 0x4330000000000000ull)) - 4503599627370496.0;
    }
    __device__ static __forceinline__ u64 raw(V x) { return (u64)__double_as_longlong(x); }
    __device__ static __forceinline__ V unraw(u64 b) { return __longlong_as_double((long long)b); }
    // any 64-bit word -> an exact integer congruent to it mod q, in (-0.52q, 0.52q + 2^32): hi*(2^32 mod q) + lo with
    // two u32 -> f64 conversions and one FP64 modmul (9 instructions against ~28 slow integer ones for Barrett)
    __device__ static __forceinline__ V reduce_wide(u64 x, const Ctx &c)
    {
        return mm(__uint2double_rn((unsigned)(x >> 32)), c.c32, c) + __uint2double_rn((unsigned)x);
    }
    // the same for a word x < 2^61 and a prime just below 2^40 (c.c40 = 2^40 mod q < 2^23, set by the host for exactly
    // those primes): x = a 2^40 + r with a < 2^21, r < 2^40, and a*c40 + r < 2^44 + 2^40 < 2^45 is an exact integer in a
    // double, congruent to x -- one shift, one conversion, one exponent splice and ONE fma (five instructions against
    // nine), inside the |y| < 2^45 the modmul's exactness argument asks of a left operand.  Callers: the mod-down
    // remainder (< P) and the digit of a 60-bit prime, both below 2^61 by construction.
    __device__ static __forceinline__ V reduce_wide40(u64 x, const Ctx &c)
    {
        const double a = __uint2double_rn((unsigned)(x >> 40));
        const double r = from_u64(x & 0xFFFFFFFFFFull);
        return __builtin_fma(a, c.c40, r);
    }
    static constexpr bool IS_F64 = true;
    // RED: 0 none, 1 generic (any 64-bit word), 2 the one-fma form (x < 2^61, c.c40 != 0)
    template <int RED>
    __device__ static __forceinline__ V input(u64 x, const InMode &m, const Ctx &c, const ModConst &)
    {
        double v = RED == 2 ? reduce_wide40(x, c) : RED == 1 ? reduce_wide(x, c) : from_u64(x);
        if (m.has_sub) v -= from_u64(m.sub);
        return v;
    }
    __device__ static __forceinline__ bool fast_wide(const Ctx &c) { return c.c40 != 0.0; }
    __device__ static __forceinline__ u64 to_u64(double r)
    {
        return (u64)__double_as_longlong(r + 4503599627370496.0) & 0x000FFFFFFFFFFFFFull;
    }
    __device__ static __forceinline__ u64 canon(double x, const Ctx &c)
    {
        double r = red(x, c);
        r = r < 0.0 ? r + c.q : r;
        return to_u64(r);
    }
    __device__ static __forceinline__ u64 fwd_finish(V x, const Ctx &c) { return canon(x, c); }
    // what the key MAC reads from scratch for a row of this policy (MacF): the UNFINISHED value itself, as a double --
    // an integer with |x| < 2^41 + (LOGN+1) * 0.52q < 2^45, a valid left operand of mm(); no canonicalisation here
    __device__ static __forceinline__ u64 mac_operand(V x, const Ctx &) { return (u64)__double_as_longlong(x); }
    template <int SLACK>
    __device__ static __forceinline__ u64 mac_operand_lazy(V x, const Ctx &c)
    {
        return mac_operand(x, c);
    }
    __device__ static __forceinline__ u64 moddown(V f, u64 acc, u64 sadd, u64 pt, bool has_pt, const Ctx &c,
                                                  const double2 &pinv)
    {
        double z = unraw(acc) - f;  // acc: the key MAC's unfinished FP64 sum as a double (MacF::result_data, |acc| < 2^45); exact: |f| < 2^45
        z = mm(z, pinv.x, c) + from_u64(sadd);
        if (has_pt) {  // the product is already in (-0.52q, 0.52q): no re-centring before the sign fix-up
            z = mm(z, from_u64(pt), c);
            return to_u64(z < 0.0 ? z + c.q : z);
        }
        return canon(z, c);
    }
    __device__ static __forceinline__ u64 inv_finish(V x, const Ctx &c) { return canon(x, c); }
};

// ---- where the FP64 inverse cores re-centre (round 5) -------------------------------------------------------------
// A Gentleman-Sande stage doubles the sum path (x' = x + y) and leaves the twiddled difference reduced, so between two
// re-centrings (|x| <= 0.5 q) the values are bounded by 0.5 q * 2^s after s stages.  The modmul stays EXACT for a left
// operand below 2^49 (DESIGN.md "FP64 modmul", extended: h + l = y*w is an error-free product for any doubles; the quotient
// estimate rint(h * RN(1/q)) errs by < 0.5 + 2^48 * 3 * 2^-53 + 2^36 / q < 0.72, so |t| < 0.75 q; h - c q is an integer below
// 2^41 and exact in the FMA; t = s + l is an integer below 2^42), hence every stage may be ENTERED with values up to
// 64 q = 0.5 q * 2^7 (the difference it feeds the modmul is then below 128 q < 2^48 for q < 2^41).  A pass of c stages entered
// at s re-centres first iff s + c - 1 > 7.  Until round 5 every pass did: four re-centrings per split-2 row at N = 16384
// where this rule asks for one (-144 of ~1300 instructions per thread on the FP64 inverse rows).
// Passes are numbered in execution order: k = 0 is the remainder pass when there is one (rstages > 0), then the full passes.
struct InvRecentre {
    static constexpr int SMAX = 7;
    __host__ __device__ static constexpr bool at(int rstages, int nfull, int cfull, int s0, int k)
    {
        int s = s0;
        bool rec = false;
        const int npass = (rstages > 0 ? 1 : 0) + nfull;
        for (int i = 0; i < npass && i <= k; ++i) {
            const int c = (rstages > 0 && i == 0) ? rstages : cfull;
            rec = s + c - 1 > SMAX;
            if (rec) s = 0;
            s += c;
        }
        return rec;
    }
};

// ---- policy plumbing for code that is templated on the arithmetic policy ------------------------------------------
__device__ __forceinline__ ArithF64::Ctx make_ctx(ArithF64, const ModConst &, const ModConstF &mf) { return ArithF64::make(mf); }
template <bool L16>
__device__ __forceinline__ typename ArithU64T<L16>::Ctx make_ctx(ArithU64T<L16>, const ModConst &mc, const ModConstF &)
{
    return ArithU64T<L16>::make(mc);
}
__device__ __forceinline__ const double *fwd_tw(ArithF64, const NttTables &nt) { return nt.twf; }
__device__ __forceinline__ const double *inv_tw(ArithF64, const NttTables &nt) { return nt.itwf; }
template <bool L16>
__device__ __forceinline__ const ulonglong2 *fwd_tw(ArithU64T<L16>, const NttTables &nt) { return nt.tw; }
template <bool L16>
__device__ __forceinline__ const ulonglong2 *inv_tw(ArithU64T<L16>, const NttTables &nt) { return nt.itw; }
// f(policy tag) with the FORWARD policy of a modulus: FP64 below 2^41, else the integer policy the prime admits
template <class F>
__device__ __forceinline__ void fwd_policy_dispatch(const ModConst &mc, const ModConstF &mf, const F &f)
{
    if (mf.q != 0.0)
        f(ArithF64{});
    else
        fwd_int_dispatch(mc, f);
}

// ---- twiddle prefetch ---------------------------------------------------------------------------
// A radix-16 pass needs 1+2+4+8 = 15 twiddle records per thread (slot (1<<u)-1+k for stage u, k < 2^u); the
// remainder pass needs NG*(G-1) <= 14.  They are loaded into registers ONE PASS AHEAD -- right after the math of
// the previous pass, i.e. before its LDS exchange and barrier -- so their L2 latency overlaps the exchange instead
// of being exposed at every stage (measured: exposed twiddle latency was ~25 % of the kernel).
template <int LOGN, class A>
__device__ __forceinline__ void load_pass_tw(typename A::TW (&w)[15], const typename A::TW *__restrict__ tw, int p,
                                             int t, int pre)
{
    const int LOGS = LOGN - 4 * (p + 1);
    int b = t >> LOGS;
    if (LOGS >= 6) b = __builtin_amdgcn_readfirstlane(b);  // wave-uniform -> scalar loads
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < (1 << u); ++k) w[(1 << u) - 1 + k] = tw[(pre << (4 * p + u)) + (b << u) + k];
}
template <int LOGN, class A>
__device__ __forceinline__ void load_rem_tw(typename A::TW (&w)[15], const typename A::TW *__restrict__ tw, int t,
                                            int pre)
{
    using C = NttCfg<LOGN>;
#pragma unroll
    for (int c = 0; c < C::NG; ++c) {
        const int g = t + C::T * c;
#pragma unroll
        for (int u = 0; u < C::R; ++u)
#pragma unroll
            for (int k = 0; k < (1 << u); ++k)
                w[c * (C::G - 1) + (1 << u) - 1 + k] = tw[(pre << (4 * C::FP + u)) + (g << u) + k];
    }
}

// v[r] holds coefficient idx_nat(t,r) on entry (U64: any value < 8q; F64: |v| <= ~2q) and the NTT value
// idx_out(t,r) on exit, NOT yet canonical (apply A::fwd_finish).
// `pre` is the twiddle-index prefix: 1 for a whole transform of size 2^LOGN; 2+h when this call is half h of a
// transform of size 2^(LOGN+1) whose first stage was applied by the caller (stage s uses tw[(pre << s) + i]).
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// `tail_hook` runs once, right after the last LDS exchange (before the final stages' arithmetic): a caller's chance to
// put loads in flight that only its epilogue needs (ks_moddown_finish).
// t0 (default: t): the thread's LOGICAL index in pass 0 -- on entry v[r] = coefficient idx_nat(t0, r).  Pass 0 combines the
// sixteen registers of a thread with workgroup-uniform twiddles and hands its results to LDS, so which column of the
// T x 16 array a thread owns there is free; loaders over [evens | odds] rows pick the one that makes lane-adjacent reads
// adjacent in memory (eo_lane, hefx_keyswitch.hip).  Every later pass works by t.
// LEAN (stand-alone row kernel, hefx_kernels.hip): the thread index the NEXT pass's twiddle addresses are formed from is
// laundered through an empty asm at the point of use.  Without it the compiler forms every pass's twiddle addresses at
// kernel entry (they depend on t only) and carries them -- 12-22 VGPRs -- through the whole transform, which under the
// 128-VGPR cap of a 1024-thread workgroup means scratch memory.
template <int LOGN, class A, class HOOK = NoHook, bool LEAN = false>
__device__ __forceinline__ void ntt_fwd_core(typename A::V (&v)[16], typename A::V *lds,
                                             const typename A::TW *__restrict__ tw, const typename A::Ctx &cx, int t,
                                             int pre, const HOOK &tail_hook = HOOK(), int t0 = -1)
{
    using C = NttCfg<LOGN>;
    static_assert(LOGN - 4 == 31 - __builtin_clz(C::T), "pass 0: one block, twiddles uniform over the workgroup");
    typename A::TW w[15];
    load_pass_tw<LOGN, A>(w, tw, 0, t, pre);
#pragma unroll
    for (int p = 0; p < C::FP; ++p) {
        const int LOGS = LOGN - 4 * (p + 1);
        const int S = 1 << LOGS;
        const int tp = (p == 0 && t0 >= 0) ? t0 : t;
        const int b = tp >> LOGS;
        const int base = b * (16 * S) + (tp & (S - 1));
        const int pb = C::phys(base), ps = C::pass_stride(LOGS);
        if (p > 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = lds[pb + ps * e];
        }
        if (C::R == 0 && p == C::FP - 1) tail_hook();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int half = 8 >> u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (e & half) continue;
                A::ct(v[e], v[e | half], w[(1 << u) - 1 + (e >> (4 - u))], cx, 4 * p + u);
            }
        }
        int tl = t;
        if (LEAN) asm volatile("" : "+v"(tl));
        if (p + 1 < C::FP)
            load_pass_tw<LOGN, A>(w, tw, p + 1, tl, pre);
        else if (C::R > 0)
            load_rem_tw<LOGN, A>(w, tw, tl, pre);
        if (p + 1 < C::FP || C::R > 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) lds[pb + ps * e] = v[e];
            __syncthreads();
        }
    }
    if (C::R > 0) {
        const int pg = C::phys(t * C::G);
#pragma unroll
        for (int c = 0; c < C::NG; ++c) {
#pragma unroll
            for (int e = 0; e < C::G; ++e) v[c * C::G + e] = lds[pg + C::REM_STRIDE * c + e];
        }
        tail_hook();
#pragma unroll
        for (int u = 0; u < C::R; ++u) {
            const int half = C::G >> (u + 1);
#pragma unroll
            for (int c = 0; c < C::NG; ++c) {
#pragma unroll
                for (int e = 0; e < C::G; ++e) {
                    if (e & half) continue;
                    A::ct(v[c * C::G + e], v[c * C::G + (e | half)],
                          w[c * (C::G - 1) + (1 << u) - 1 + (e >> (C::R - u))], cx, 4 * C::FP + u);
                }
            }
        }
    }
}

// v[r] holds NTT value idx_out(t,r) on entry (U64: [0,4q); F64: |v| < 2^45) and coefficient idx_nat(t,r) on exit,
// NOT yet canonical (apply A::inv_finish).  itw[idx] = tw[idx]^-1 (same indexing); N^-1 folded in the last stage.
// S0: the values on entry are bounded by 0.5 q * 2^S0 in magnitude (FP64 policy; InvRecentre): 2 behind the first stage of a
// split inverse (sums of two canonical words), 1 for canonical words
template <int LOGN, class A, int S0 = 2>
__device__ __forceinline__ void ntt_inv_core(typename A::V (&v)[16], typename A::V *lds,
                                             const typename A::TW *__restrict__ itw, const typename A::Ctx &cx, int t)
{
    using C = NttCfg<LOGN>;
    typename A::TW w[15];
    if (C::R > 0) {
        load_rem_tw<LOGN, A>(w, itw, t, 1);
        if (InvRecentre::at(C::R, C::FP, 4, S0, 0)) A::inv_pass_begin(v, cx);
#pragma unroll
        for (int u = C::R - 1; u >= 0; --u) {
            const int half = C::G >> (u + 1);
#pragma unroll
            for (int c = 0; c < C::NG; ++c) {
#pragma unroll
                for (int e = 0; e < C::G; ++e) {
                    if (e & half) continue;
                    A::gs(v[c * C::G + e], v[c * C::G + (e | half)],
                          w[c * (C::G - 1) + (1 << u) - 1 + (e >> (C::R - u))], cx);
                }
            }
        }
        load_pass_tw<LOGN, A>(w, itw, C::FP - 1, t, 1);  // next pass's twiddles travel during the exchange
        const int pg = C::phys(t * C::G);
#pragma unroll
        for (int c = 0; c < C::NG; ++c) {
#pragma unroll
            for (int e = 0; e < C::G; ++e) lds[pg + C::REM_STRIDE * c + e] = v[c * C::G + e];
        }
        __syncthreads();
    } else {
        load_pass_tw<LOGN, A>(w, itw, C::FP - 1, t, 1);
    }
#pragma unroll
    for (int p = C::FP - 1; p >= 0; --p) {
        const int LOGS = LOGN - 4 * (p + 1);
        const int S = 1 << LOGS;
        const int b = t >> LOGS;
        const int base = b * (16 * S) + (t & (S - 1));
        const int pb = C::phys(base), ps = C::pass_stride(LOGS);
        if (p < C::FP - 1 || C::R > 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = lds[pb + ps * e];
        }
        if (InvRecentre::at(C::R, C::FP, 4, S0, (C::R > 0 ? 1 : 0) + (C::FP - 1 - p))) A::inv_pass_begin(v, cx);
#pragma unroll
        for (int u = 3; u >= 0; --u) {
            const int half = 8 >> u;
            if (p == 0 && u == 0) {  // last stage: fold N^-1
#pragma unroll
                for (int e = 0; e < 8; ++e) A::gs_last(v[e], v[e | 8], cx);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (e & half) continue;
                    A::gs(v[e], v[e | half], w[(1 << u) - 1 + (e >> (4 - u))], cx);
                }
            }
        }
        if (p > 0) {
            load_pass_tw<LOGN, A>(w, itw, p - 1, t, 1);
#pragma unroll
            for (int e = 0; e < 16; ++e) lds[pb + ps * e] = v[e];
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// u64-in / u64-out wrappers that pick the policy for the row's prime (uniform per workgroup).
// T.modsf[m].q == 0 marks a prime that is too wide for the FP64 policy.
// ------------------------------------------------------------------------------------------------
// whole transform of size 2^LOGN; v: canonical in (idx_nat) -> canonical out (idx_out)
template <int LOGN>
__device__ __forceinline__ void ntt_fwd_row(u64 (&v)[16], u64 *lds, const NttTables &nt, const ModConst &mc,
                                            const ModConstF &mf, int t)
{
    if (mf.q != 0.0) {
        const ArithF64::Ctx cx = ArithF64::make(mf);
        double f[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) f[r] = ArithF64::from_u64(v[r]);
        ntt_fwd_core<LOGN, ArithF64>(f, reinterpret_cast<double *>(lds), nt.twf, cx, t, 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = ArithF64::fwd_finish(f[r], cx);
    } else {
        const ArithU64::Ctx cx = ArithU64::make(mc);
        ntt_fwd_core<LOGN, ArithU64>(v, lds, nt.tw, cx, t, 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = ArithU64::fwd_finish(v[r], cx);
    }
}

template <int LOGN>
__device__ __forceinline__ void ntt_inv_row(u64 (&v)[16], u64 *lds, const NttTables &nt, const ModConst &mc,
                                            const ModConstF &mf, int t)
{
    if (mf.q != 0.0) {
        const ArithF64::Ctx cx = ArithF64::make(mf);
        double f[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) f[r] = ArithF64::from_u64(v[r]);
        ntt_inv_core<LOGN, ArithF64>(f, reinterpret_cast<double *>(lds), nt.itwf, cx, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = ArithF64::inv_finish(f[r], cx);
    } else {
        const ArithU64::Ctx cx = ArithU64::make(mc);
        ntt_inv_core<LOGN, ArithU64>(v, lds, nt.itw, cx, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = ArithU64::inv_finish(v[r], cx);
    }
}

// Forward split (transform size 2^LOGN, this workgroup = half h, sub-transform 2^(LOGN-1)):
// ld(r, x, y) delivers the raw words of coefficients idx_nat(t,r) and idx_nat(t,r)+N/2, `mode` says how they become
// inputs (reduction / constant subtraction in the row's arithmetic policy); on return
// v[r] = NTT value at h*N/2 + idx_io(t,r), canonical.  Loads are issued in two batches of eight pairs.
// NB = number of load batches of the first stage: 2 (eight pairs each) under the 128-VGPR cap, 1 (all sixteen pairs in
// flight at once, one exposed memory latency instead of two) in the 256-VGPR builds.
// FAST40: also build the one-fma reduction of wide words (ArithF64::reduce_wide40) -- only where it pays for a third
// first-stage variant in the kernel's register budget: the mod-down epilogue kernel, whose every input is wide
template <int LOGN, class A, class LD, int NB = 2, class HOOK = NoHook, bool FAST40 = false>
__device__ __forceinline__ void split_fwd_raw(typename A::V (&f)[16], const LD &ld, const InMode &mode,
                                              const ModConst &mc, u64 *lds, const typename A::TW *__restrict__ tw,
                                              const typename A::Ctx &cx, int t, int h, const HOOK &tail_hook = HOOK(),
                                              int t0 = -1)
{
    const typename A::TW w1 = A::half_twiddle(tw[1], cx, h);
    constexpr int BS = 16 / NB;
    // the reduce / no-reduce decision is uniform per workgroup: one branch around the whole first stage, not one
    // select per element (which would make every row pay for the reduction)
    if (A::IS_F64 ? mode.red_f64 : mode.red_int) {
        // FAST40 kernels (mod-down epilogue): the cheap reduction the row's prime admits -- FP64: one fma (primes just
        // below 2^40); integer: ONE conditional subtraction when the words are below 2 q (two 60-bit primes)
        // (the integer form only from N = 16384 on: at N = 8192 the extra variant costs the kernel a wave per SIMD)
        constexpr bool CHEAP_OK = FAST40 && (A::IS_F64 || LOGN >= 14);
        bool fast = false;
        if constexpr (CHEAP_OK) fast = A::IS_F64 ? (mode.below_2_61 && A::fast_wide(cx)) : mode.lt2q;
        if (fast) {
            if constexpr (CHEAP_OK) {
                constexpr int CHEAP = A::IS_F64 ? 2 : 3;
#pragma unroll
                for (int g = 0; g < NB; ++g) {
                    u64 x[BS], y[BS];
#pragma unroll
                    for (int r = 0; r < BS; ++r) ld(BS * g + r, x[r], y[r]);
#pragma unroll
                    for (int r = 0; r < BS; ++r)
                        f[BS * g + r] = A::ct_half(A::template input<CHEAP>(x[r], mode, cx, mc),
                                                   A::template input<CHEAP>(y[r], mode, cx, mc), w1, cx);
                    HEFX_STAGE_FENCE();
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < NB; ++g) {
                u64 x[BS], y[BS];
#pragma unroll
                for (int r = 0; r < BS; ++r) ld(BS * g + r, x[r], y[r]);
#pragma unroll
                for (int r = 0; r < BS; ++r)
                    f[BS * g + r] = A::ct_half(A::template input<1>(x[r], mode, cx, mc),
                                               A::template input<1>(y[r], mode, cx, mc), w1, cx);
                HEFX_STAGE_FENCE();
            }
        }
    } else {
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            u64 x[BS], y[BS];
#pragma unroll
            for (int r = 0; r < BS; ++r) ld(BS * g + r, x[r], y[r]);
#pragma unroll
            for (int r = 0; r < BS; ++r)
                f[BS * g + r] = A::ct_half(A::template input<0>(x[r], mode, cx, mc),
                                           A::template input<0>(y[r], mode, cx, mc), w1, cx);
            HEFX_STAGE_FENCE();
        }
    }
    ntt_fwd_core<LOGN - 1, A, HOOK>(f, reinterpret_cast<typename A::V *>(lds), tw, cx, t, 2 + h, tail_hook, t0);
    using C = NttCfg<LOGN - 1>;
    if constexpr (C::R == 0) {  // idx_out -> idx_io through LDS (the last pass read exactly the words written here)
        typename A::V *lf = reinterpret_cast<typename A::V *>(lds);
        // R == 0: phys(i) = i + (i >> 4), written out along both patterns (affine, see NttCfg)
        const int pw = 17 * t, pr = 2 * t + (t >> 3);
#pragma unroll
        for (int r = 0; r < 16; ++r) lf[pw + r] = f[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) f[r] = lf[pr + (2 * C::T + C::T / 8) * (r >> 1) + (r & 1)];  // phys(idx_io(t, r))
    }
}

// MACOP: deliver A::mac_operand (the key MAC's input format) instead of canonical words
template <int LOGN, class A, class LD, int NB = 2, bool MACOP = false>
__device__ __forceinline__ void split_fwd_a(u64 (&v)[16], const LD &ld, const InMode &mode, const ModConst &mc,
                                            u64 *lds, const typename A::TW *__restrict__ tw,
                                            const typename A::Ctx &cx, int t, int h, int t0 = -1, int mac_slack = 0)
{
    typename A::V f[16];
    split_fwd_raw<LOGN, A, LD, NB>(f, ld, mode, mc, lds, tw, cx, t, h, NoHook(), t0);
    if (MACOP && !A::IS_F64 && mac_slack == 1) {  // workgroup-uniform: one branch around the sixteen words
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = A::template mac_operand_lazy<1>(f[r], cx);
    } else if (MACOP && !A::IS_F64 && mac_slack >= 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = A::template mac_operand_lazy<2>(f[r], cx);
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = MACOP ? A::mac_operand(f[r], cx) : A::fwd_finish(f[r], cx);
    }
}

template <int LOGN, int NB = 2, bool MACOP = false, class LD>
__device__ __forceinline__ void split_fwd(u64 (&v)[16], const LD &ld, const InMode &mode, u64 *lds,
                                          const NttTables &nt, const ModConst &mc, const ModConstF &mf, int t, int h,
                                          int t0 = -1, int mac_slack = 0)
{
    if (mf.q != 0.0)
        split_fwd_a<LOGN, ArithF64, LD, NB, MACOP>(v, ld, mode, mc, lds, nt.twf, ArithF64::make(mf), t, h, t0);
    else if constexpr (MACOP)  // the digit transforms of the key switch: the lighter L16 butterfly where the prime admits it
        fwd_int_dispatch(mc, [&](auto pol) {
            using A = decltype(pol);
            split_fwd_a<LOGN, A, LD, NB, MACOP>(v, ld, mode, mc, lds, nt.tw, A::make(mc), t, h, t0, mac_slack);
        });
    else
        split_fwd_a<LOGN, ArithU64, LD, NB, MACOP>(v, ld, mode, mc, lds, nt.tw, ArithU64::make(mc), t, h, t0);
}

// Inverse split: a0[r], a1[r] = canonical NTT values at positions 2j, 2j+1 with j = idx_out(t,r) of the
// sub-transform (fetched as lane-adjacent records and exchanged through LDS when R == 0); on return v[r] = coefficient 2*idx_nat(t,r) + h, canonical.
// `ldp(j)` delivers the row's (value[2j], value[2j+1]) record -- a plain 16-byte load, or a Galois-gathered one
// (hefx_keyswitch.hip).  The first stage is done in two batches of eight pairs (+ eight twiddles for the odd half) with
// a scheduling fence between them: issuing all 16 pair loads and 16 twiddle loads at once needs ~250 VGPRs in the FP64
// policy and spilled heavily at the 128 cap.
template <int LOGN, class A, int NB = 2, class LDP>
__device__ __forceinline__ void split_inv_a(u64 (&v)[16], const LDP &ldp, u64 *lds,
                                            const typename A::TW *__restrict__ itw, const typename A::Ctx &cx, int t,
                                            int h)
{
    using C = NttCfg<LOGN - 1>;
    constexpr int BS = 16 / NB;
    typename A::V f[16];
    // the first stage is element-wise on records, so for R == 0 it runs on lane-adjacent records (idx_nat; see idx_io)
    // and its results reach the layout the core starts in (idx_out) through LDS
    // ... and for R >= 2 as well, where a lane would walk four or eight consecutive records (64 / 128 bytes) through its
    // loads: N = 32768 +1.3 % end to end, the two inverse kernels -10 % and -4 % (profiles/r03/ab_inverse_record_order.txt).
    // R == 1 (N = 16384: two records, 32 bytes per lane) keeps idx_out: the transposition costs what it saves there.
    constexpr bool NATREC = C::R != 1;
    auto rec = [&](int r) { return NATREC ? C::idx_nat(t, r) : C::idx_out(t, r); };
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        ulonglong2 pr[BS];
#pragma unroll
        for (int r = 0; r < BS; ++r) pr[r] = ldp(rec(BS * g + r));
        if (h == 0) {
#pragma unroll
            for (int r = 0; r < BS; ++r)
                f[BS * g + r] = A::gs_half_sum(A::from_u64(pr[r].x), A::from_u64(pr[r].y), cx);
        } else {
            typename A::TW w[BS];
#pragma unroll
            for (int r = 0; r < BS; ++r) w[r] = itw[C::N + rec(BS * g + r)];  // itw[N/2 + j]
#pragma unroll
            for (int r = 0; r < BS; ++r)
                f[BS * g + r] = A::gs_half_diff(A::from_u64(pr[r].x), A::from_u64(pr[r].y), w[r], cx);
        }
        HEFX_STAGE_FENCE();
    }
    if constexpr (C::R == 0) {
        typename A::V *lf = reinterpret_cast<typename A::V *>(lds);
        const int pw = t + (t >> 4), pr = 17 * t;  // R == 0: phys(i) = i + (i >> 4)
#pragma unroll
        for (int r = 0; r < 16; ++r) lf[pw + (C::T + C::T / 16) * r] = f[r];  // phys(idx_nat(t, r))
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) f[r] = lf[pr + r];  // phys(16t + r): the words the core's first pass owns
    } else if constexpr (NATREC) {
        typename A::V *lf = reinterpret_cast<typename A::V *>(lds);
#pragma unroll
        for (int r = 0; r < 16; ++r) lf[C::phys(C::idx_nat(t, r))] = f[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) f[r] = lf[C::phys(C::idx_out(t, r))];  // the words the remainder pass rewrites
    }
    ntt_inv_core<LOGN - 1, A>(f, reinterpret_cast<typename A::V *>(lds), itw, cx, t);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = A::inv_finish(f[r], cx);
}

template <int LOGN, int NB = 2, class LDP>
__device__ __forceinline__ void split_inv_ld(u64 (&v)[16], const LDP &ldp, u64 *lds, const NttTables &nt,
                                             const ModConst &mc, const ModConstF &mf, int t, int h)
{
    if (mf.q != 0.0)
        split_inv_a<LOGN, ArithF64, NB>(v, ldp, lds, nt.itwf, ArithF64::make(mf), t, h);
    else
        split_inv_a<LOGN, ArithU64, NB>(v, ldp, lds, nt.itw, ArithU64::make(mc), t, h);
}

// `pairs` points at the row viewed as (value[2j], value[2j+1]) records
template <int LOGN, int NB = 2>
__device__ __forceinline__ void split_inv(u64 (&v)[16], const ulonglong2 *__restrict__ pairs, u64 *lds,
                                          const NttTables &nt, const ModConst &mc, const ModConstF &mf, int t, int h)
{
    split_inv_ld<LOGN, NB>(v, [pairs](int j) { return gld16(pairs + j); }, lds, nt, mc, mf, t, h);
}

}  // namespace hefx

// hefx_modarith.cuh -- 64-bit modular arithmetic for gfx950 lanes.
//
// gfx950 has no native 64x64 multiply: everything below lowers to v_mad_u64_u32 / v_mul_lo_u32 /
// v_mul_hi_u32 chains.  All public results of the library are canonical residues in [0,q), so any
// exact reduction reproduces SEAL 3.4.5's bits; laziness is internal only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hefx {

typedef unsigned long long u64;

// Per-modulus constants, one 64-byte record per RNS prime (uniform per workgroup -> scalar loads).
struct ModConst {
    u64 q;
    u64 r0, r1;        // floor(2^128/q) = r1*2^64 + r0   (Barrett)
    u64 ninv, ninv_s;  // N^-1 mod q and its Shoup companion floor(ninv*2^64/q)
    u64 ilw, ilw_s;    // (psi^-1 twiddle of the last inverse stage) * N^-1, and Shoup companion
    u64 nq;            // 2^64 - q (shoup_lazy*, csubn: additions instead of subtractions)
};

// Per-modulus constants of the FP64 policy (valid only for q < 2^41).
struct ModConstF {
    double q, qinv;          // q and RN(1/q)
    double ninv, ninv_r;     // N^-1 mod q and RN(ninv/q)
    double ilw, ilw_r;       // (last inverse-stage twiddle * N^-1) mod q and RN(ilw/q)
    double c32;              // 2^32 mod q (FP64 reduction of a 64-bit word: hi*c32 + lo)
    double c40;              // 2^40 mod q when that is below 2^23 (q just under 2^40, SEAL's 40-bit primes), else 0:
                             // a word x < 2^61 then reduces as (x >> 40)*c40 + (x mod 2^40) < 2^45 in ONE fma
};

// Twiddle tables of one modulus (integer and FP64 policies).
struct NttTables {
    const ulonglong2 *tw, *itw;  // [N] of this modulus
    const double *twf, *itwf;    // [N] of this modulus (FP64 policy)
};

// Pointers that a kernel READS FROM DEVICE MEMORY -- the fields of an item descriptor, the entries of a pointer table
// -- are GENERIC pointers to the compiler (only pointer kernel arguments, also inside by-value argument structs, are
// known to be global), and every access through a generic pointer is a FLAT instruction: it counts in vmcnt AND lgkmcnt
// and completes out of order, so the only wait the compiler can place after one is the full drain
// s_waitcnt vmcnt(0) lgkmcnt(0) -- no load of a later record stays in flight across the use of an earlier one, and a
// store is waited for like a load.  Found in the ISA in round 3: 64 flat loads in ks_intt_digits, 84 in ks_mac (not one
// counted wait in the whole kernel), 62 flat loads and ALL 59 stores of ks_moddown_finish.  The accessors below make
// the access through a pointer in the global address space (the cast has to sit AT the access: laundering the pointer
// through address space 1 and back is folded away; and the 16-byte forms go through a built-in vector type, because a
// class-type load is a copy constructor on a generic reference).  Every buffer the engine is handed is device memory
// (hipMalloc / hefx_malloc), so the claim is true by the ABI's contract.
#define HEFX_GLOBAL_AS __attribute__((address_space(1)))
typedef u64 hefx_u64x2 __attribute__((ext_vector_type(2)));
typedef uint32_t hefx_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u64 gld8(const u64 *p) { return *(const HEFX_GLOBAL_AS u64 *)p; }
__device__ __forceinline__ void gst8(u64 *p, u64 v) { *(HEFX_GLOBAL_AS u64 *)p = v; }
__device__ __forceinline__ ulonglong2 gld16(const void *p)
{
    const hefx_u64x2 v = *(const HEFX_GLOBAL_AS hefx_u64x2 *)p;
    return make_ulonglong2(v.x, v.y);
}
__device__ __forceinline__ void gst16(void *p, const ulonglong2 &v)
{
    hefx_u64x2 t;
    t.x = v.x, t.y = v.y;
    *(HEFX_GLOBAL_AS hefx_u64x2 *)p = t;
}
__device__ __forceinline__ uint2 gld_u32x2(const void *p)
{
    const hefx_u32x2 v = *(const HEFX_GLOBAL_AS hefx_u32x2 *)p;
    return make_uint2(v.x, v.y);
}

__device__ __forceinline__ u64 mulhi64(u64 a, u64 b) { return __umul64hi(a, b); }


// floor(x*ws / 2^64) UNDER-estimated by at most 2: only the three partial products that reach bit 64, no carry chain
// between them.  With x = x1 2^32 + x0, ws = w1 2^32 + w0 the exact value is
//   x1 w1 + floor(((x0 w1 + x1 w0) 2^32 + x0 w0) / 2^64) = x1 w1 + hi32(x0 w1) + hi32(x1 w0) + e,
// e = floor(((lo32(x0 w1) + lo32(x1 w0)) 2^32 + x0 w0) / 2^64) in {0,1,2}.  Five instructions (two v_mul_hi_u32, a
// 33-bit add, one v_mad_u64_u32) against eleven for the exact high word; the sum cannot wrap (it is <= the exact value).
__device__ __forceinline__ u64 mulhi64_under2(u64 x, u64 ws)
{
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), w0 = (uint32_t)ws, w1 = (uint32_t)(ws >> 32);
    return (u64)x1 * w1 + ((u64)__umulhi(x0, w1) + (u64)__umulhi(x1, w0));
}
// x*w - h*q mod 2^64 as x*w + h*nq with nq = 2^64 - q, every partial product a 64-bit multiply-add that also does the
// addition (and, through the negated modulus, the subtraction): six v_mad_u64_u32 and one 32-bit add where the
// compiler's form of x*w - h*q takes ten instructions (two v_mad_u64_u32, four v_mul_lo_u32, two three-way adds and a
// borrow chain).  The four cross products only matter through the low word of their sum; left to itself the optimiser
// sees that and goes back to 32-bit multiplies plus adds, hence the empty asm (a value barrier, no instruction) and the
// one-instruction asm for the 32-bit add (written in C it becomes a 64-bit add of u << 32).  Same value, bit for bit.
__device__ __forceinline__ u64 mul_sub_lo64(u64 x, u64 w, u64 h, u64 nq)
{
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint32_t h0 = (uint32_t)h, h1 = (uint32_t)(h >> 32);
    uint32_t n0 = (uint32_t)nq, n1 = (uint32_t)(nq >> 32);
    // nq is loop-invariant: its zero-extended halves get hoisted out of the (unrolled) transform as 64-bit values, and
    // instruction selection -- which only sees one basic block -- then multiplies by the zero high words too (ten
    // multiply-adds and seven moves per butterfly instead of seven and none).  Pinning the halves here keeps the
    // extension next to its use.
    asm("" : "+v"(n0), "+v"(n1));
    u64 u = (u64)x0 * w1;
    u = (u64)x1 * w0 + u;
    u = (u64)h0 * n1 + u;
    u = (u64)h1 * n0 + u;
    asm("" : "+v"(u));
    u64 a = (u64)x0 * w0;
    a = (u64)h0 * n0 + a;
    uint32_t ahi;
    asm("v_add_u32 %0, %1, %2" : "=v"(ahi) : "v"((uint32_t)(a >> 32)), "v"((uint32_t)u));
    return (u64)(uint32_t)a | ((u64)ahi << 32);
}
// x*w mod q in [0,2q), ws = floor(w*2^64/q), nq = 2^64 - q; valid for ANY 64-bit x (Harvey/Shoup).
__device__ __forceinline__ u64 shoup_lazy(u64 x, u64 w, u64 ws, u64 nq) { return mul_sub_lo64(x, w, mulhi64(x, ws), nq); }
// x*w mod q in [0,4q) for ANY 64-bit x: Shoup with the under-estimated quotient (each missing unit adds one q);
// nq = 2^64 - q
__device__ __forceinline__ u64 shoup_lazy4(u64 x, u64 w, u64 ws, u64 nq)
{
    return mul_sub_lo64(x, w, mulhi64_under2(x, ws), nq);
}
// x in [0,2m) -> [0,m) given nm = 2^64 - m as a LOADED value (ModConst::nq and its multiples; computed in the kernel the
// optimiser turns x + (0 - m) back into a subtraction): one 64-bit add (v_lshl_add_u64), one compare, two selects
// -- no borrow chain through VCC (x - m needs v_sub_co / v_subb_co and their wait states)
__device__ __forceinline__ u64 csubn(u64 x, u64 nm)
{
    const u64 d = x + nm;
    return d > x ? x : d;
}

// x in [0,2m) -> [0,m).  Written on the borrow of x - m (sub, subb, two selects) rather than compare-then-subtract.
__device__ __forceinline__ u64 csub(u64 x, u64 q)
{
    u64 d;
    return __builtin_usubll_overflow(x, q, &d) ? x : d;
}

__device__ __forceinline__ u64 addmod(u64 a, u64 b, u64 q) { return csub(a + b, q); }
__device__ __forceinline__ u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
__device__ __forceinline__ u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }

// any 64-bit x -> [0,q); r1 = floor(2^64/q)
__device__ __forceinline__ u64 barrett64(u64 x, u64 q, u64 r1) { return csub(x - mulhi64(x, r1) * q, q); }

// (hi:lo) < q*2^64 -> [0,2q): the quotient estimate is short by at most one
__device__ __forceinline__ u64 barrett128_lt2q(u64 lo, u64 hi, const ModConst &m)
{
    u64 carry = mulhi64(lo, m.r0);
    u64 t_lo = lo * m.r1, t_hi = mulhi64(lo, m.r1);
    u64 tmp1 = t_lo + carry;
    u64 tmp3 = t_hi + (tmp1 < carry);
    u64 u_lo = hi * m.r0, u_hi = mulhi64(hi, m.r0);
    u64 s = tmp1 + u_lo;
    u64 carry2 = u_hi + (s < u_lo);
    u64 qhat = hi * m.r1 + tmp3 + carry2;
    return lo - qhat * m.q;
}
// (hi:lo) < q*2^64 -> [0,q)
__device__ __forceinline__ u64 barrett128(u64 lo, u64 hi, const ModConst &m) { return csub(barrett128_lt2q(lo, hi, m), m.q); }

__device__ __forceinline__ u64 mulmod(u64 a, u64 b, const ModConst &m)
{
    return barrett128(a * b, mulhi64(a, b), m);
}

// 128-bit accumulate acc += a*b
__device__ __forceinline__ void mac128(u64 &lo, u64 &hi, u64 a, u64 b)
{
    u64 pl = a * b, ph = mulhi64(a, b);
    lo += pl;
    hi += ph + (lo < pl);
}

}  // namespace hefx

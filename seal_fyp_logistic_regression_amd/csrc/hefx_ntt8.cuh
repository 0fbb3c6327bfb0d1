// hefx_ntt8.cuh -- NTT cores with EIGHT coefficients per thread (radix-8 passes), used by the fused
// "digit NTT + key MAC" kernel (two FP64 accumulators per coefficient = 32 VGPRs next to the transform's working set,
// where sixteen coefficients per thread would need 64) and by the quarter-row kernels of the small-batch path.
// Same conventions as hefx_ntt.cuh (policies ArithU64 / ArithF64, twiddle prefix `pre`, one-pass-ahead twiddle
// prefetch, one barrier per LDS exchange, padding G words every 8*G words -> conflict-free b64 accesses).
#pragma once
#include "hefx_ntt.cuh"

namespace hefx {

template <int LOGN>
struct Ntt8Cfg {
    static constexpr int N = 1 << LOGN;
    static constexpr int T = N / 8;      // threads per (sub-)transform
    static constexpr int FP = LOGN / 3;  // full radix-8 passes
    static constexpr int R = LOGN % 3;   // stages of the remainder pass
    static constexpr int G = 1 << R;
    static constexpr int NG = 8 / G;
    static constexpr int PSH = 3 + R;
    static constexpr int LDS_WORDS = N + (N >> 3);
    __device__ static __forceinline__ int phys(int idx) { return idx + ((idx >> PSH) << R); }
    // affine forms of phys() along the cores' access patterns (same argument as NttCfg::pass_stride: the element
    // stride 2^logs is congruent to R modulo 3, so either logs >= PSH or the eight words share one pad block)
    __host__ __device__ static constexpr int pass_stride(int logs)
    {
        return logs >= PSH ? (1 << logs) + (((1 << logs) >> PSH) << R) : (1 << logs);
    }
    static constexpr int REM_STRIDE = T * G + ((T >> 3) << R);
    static_assert(T % 8 == 0, "remainder-pass addressing assumes T is a multiple of 8");
    __device__ static __forceinline__ int idx_nat(int t, int r) { return t + T * r; }
    __device__ static __forceinline__ int idx_out(int t, int r)
    {
        if (R == 0) return t * 8 + r;
        return (t + T * (r >> R)) * G + (r & (G - 1));
    }
};

template <int LOGN, class A>
__device__ __forceinline__ void load_pass_tw8(typename A::TW (&w)[7], const typename A::TW *__restrict__ tw, int p,
                                              int t, int pre)
{
    const int LOGS = LOGN - 3 * (p + 1);
    int b = t >> LOGS;
    if (LOGS >= 6) b = __builtin_amdgcn_readfirstlane(b);
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int k = 0; k < (1 << u); ++k) w[(1 << u) - 1 + k] = tw[(pre << (3 * p + u)) + (b << u) + k];
}
template <int LOGN, class A>
__device__ __forceinline__ void load_rem_tw8(typename A::TW (&w)[7], const typename A::TW *__restrict__ tw, int t,
                                             int pre)
{
    using C = Ntt8Cfg<LOGN>;
#pragma unroll
    for (int c = 0; c < C::NG; ++c) {
        const int g = t + C::T * c;
#pragma unroll
        for (int u = 0; u < C::R; ++u)
#pragma unroll
            for (int k = 0; k < (1 << u); ++k)
                w[c * (C::G - 1) + (1 << u) - 1 + k] = tw[(pre << (3 * C::FP + u)) + (g << u) + k];
    }
}

// v[r] = coefficient idx_nat(t,r) on entry, NTT value idx_out(t,r) on exit (not yet canonical: A::fwd_finish).
// `w` holds the twiddles of pass 0 on entry (load_pass_tw8(w, tw, 0, t, pre), issued by the caller as early as it likes:
// the quarter-row kernels fetch them together with their data).
// t0: the thread's LOGICAL index in pass 0 -- on entry v[r] = coefficient idx_nat(t0, r).  Pass 0 combines the eight
// registers of a thread with workgroup-uniform twiddles and hands its results to LDS, so WHICH column t0 of the
// T x 8 array a thread owns there is free: the loaders pick the permutation that makes their global reads lane-contiguous
// (quarter_fwd_lane); every later pass works by t.
template <int LOGN, class A>
__device__ __forceinline__ void ntt8_fwd_core_w(typename A::V (&v)[8], typename A::TW (&w)[7], typename A::V *lds,
                                                const typename A::TW *__restrict__ tw, const typename A::Ctx &cx, int t,
                                                int pre, int t0)
{
    using C = Ntt8Cfg<LOGN>;
    static_assert(LOGN - 3 == 31 - __builtin_clz(C::T), "pass 0: one block, twiddles uniform over the workgroup");
#pragma unroll
    for (int p = 0; p < C::FP; ++p) {
        const int LOGS = LOGN - 3 * (p + 1);
        const int S = 1 << LOGS;
        const int tp = p == 0 ? t0 : t;
        const int b = tp >> LOGS;
        const int base = b * (8 * S) + (tp & (S - 1));
        const int pb = C::phys(base), ps = C::pass_stride(LOGS);
        if (p > 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = lds[pb + ps * e];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int half = 4 >> u;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e & half) continue;
                A::ct(v[e], v[e | half], w[(1 << u) - 1 + (e >> (3 - u))], cx, 3 * p + u);
            }
        }
        HEFX_STAGE_FENCE();
        HEFX_STAMP_AT(8 + p);
        if (p + 1 < C::FP)
            load_pass_tw8<LOGN, A>(w, tw, p + 1, t, pre);
        else if (C::R > 0)
            load_rem_tw8<LOGN, A>(w, tw, t, pre);
        if (p + 1 < C::FP || C::R > 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) lds[pb + ps * e] = v[e];
            __syncthreads();
            HEFX_STAMP_AT(3 + p);
        }
    }
    if (C::R > 0) {
        const int pg = C::phys(t * C::G);
#pragma unroll
        for (int c = 0; c < C::NG; ++c) {
#pragma unroll
            for (int e = 0; e < C::G; ++e) v[c * C::G + e] = lds[pg + C::REM_STRIDE * c + e];
        }
#pragma unroll
        for (int u = 0; u < C::R; ++u) {
            const int half = C::G >> (u + 1);
#pragma unroll
            for (int c = 0; c < C::NG; ++c) {
#pragma unroll
                for (int e = 0; e < C::G; ++e) {
                    if (e & half) continue;
                    A::ct(v[c * C::G + e], v[c * C::G + (e | half)],
                          w[c * (C::G - 1) + (1 << u) - 1 + (e >> (C::R - u))], cx, 3 * C::FP + u);
                }
            }
        }
    }
}

template <int LOGN, class A>
__device__ __forceinline__ void ntt8_fwd_core(typename A::V (&v)[8], typename A::V *lds,
                                              const typename A::TW *__restrict__ tw, const typename A::Ctx &cx, int t,
                                              int pre)
{
    typename A::TW w[7];
    load_pass_tw8<LOGN, A>(w, tw, 0, t, pre);
    ntt8_fwd_core_w<LOGN, A>(v, w, lds, tw, cx, t, pre, t);
}

// Half h of a forward transform of size 2^LOGN: ld(r, x, y) delivers the raw words of coefficients idx_nat(t,r) and
// idx_nat(t,r) + N/2, `mode` says how they become inputs (reduction in the row's policy); on return
// f[r] = UNFINISHED NTT value at h*N/2 + idx_out(t,r) (A::fwd_finish / A::mac_operand make it a stored word).
template <int LOGN, class A, class LD>
__device__ __forceinline__ void split8_fwd_raw(typename A::V (&f)[8], const LD &ld, const InMode &mode,
                                               const ModConst &mc, u64 *lds, const typename A::TW *__restrict__ tw,
                                               const typename A::Ctx &cx, int t, int h)
{
    const typename A::TW w1 = A::half_twiddle(tw[1], cx, h);
    auto stage = [&](auto red) {
        constexpr int RED = decltype(red)::value;
        u64 x[8], y[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) ld(r, x[r], y[r]);
#pragma unroll
        for (int r = 0; r < 8; ++r)
            f[r] = A::ct_half(A::template input<RED>(x[r], mode, cx, mc), A::template input<RED>(y[r], mode, cx, mc), w1, cx);
        HEFX_STAGE_FENCE();
    };
    if (A::IS_F64 ? mode.red_f64 : mode.red_int)
        stage(std::integral_constant<int, 1>{});
    else
        stage(std::integral_constant<int, 0>{});
    ntt8_fwd_core<LOGN - 1, A>(f, reinterpret_cast<typename A::V *>(lds), tw, cx, t, 2 + h);
}

// Inverse core with eight coefficients per thread: v[r] = NTT value idx_out(t,r) on entry (U64: [0,4q); F64: |v| < 2^45),
// coefficient idx_nat(t,r) on exit, not yet canonical (A::inv_finish).  Mirror of ntt8_fwd_core; N^-1 folded into the
// last stage exactly as in ntt_inv_core.
// the twiddles the inverse core starts with: the remainder pass's, or (R == 0) those of the last full pass
template <int LOGN, class A>
__device__ __forceinline__ void load_inv_first_tw8(typename A::TW (&w)[7], const typename A::TW *__restrict__ itw, int t)
{
    using C = Ntt8Cfg<LOGN>;
    if (C::R > 0)
        load_rem_tw8<LOGN, A>(w, itw, t, 1);
    else
        load_pass_tw8<LOGN, A>(w, itw, C::FP - 1, t, 1);
}

// `w` = load_inv_first_tw8 on entry
// S0 (InvRecentre, hefx_ntt.cuh): on entry |v| <= 0.5 q * 2^S0 -- 3 behind the two stages the quarter loaders apply
// (sums of four canonical words)
template <int LOGN, class A, int S0 = 3>
__device__ __forceinline__ void ntt8_inv_core_w(typename A::V (&v)[8], typename A::TW (&w)[7], typename A::V *lds,
                                                const typename A::TW *__restrict__ itw, const typename A::Ctx &cx, int t)
{
    using C = Ntt8Cfg<LOGN>;
    if (C::R > 0) {
        if (InvRecentre::at(C::R, C::FP, 3, S0, 0)) A::inv_pass_begin(v, cx);
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
        load_pass_tw8<LOGN, A>(w, itw, C::FP - 1, t, 1);
        const int pg = C::phys(t * C::G);
#pragma unroll
        for (int c = 0; c < C::NG; ++c) {
#pragma unroll
            for (int e = 0; e < C::G; ++e) lds[pg + C::REM_STRIDE * c + e] = v[c * C::G + e];
        }
        __syncthreads();
        HEFX_STAMP_AT(3);
    }
#pragma unroll
    for (int p = C::FP - 1; p >= 0; --p) {
        const int LOGS = LOGN - 3 * (p + 1);
        const int S = 1 << LOGS;
        const int b = t >> LOGS;
        const int base = b * (8 * S) + (t & (S - 1));
        const int pb = C::phys(base), ps = C::pass_stride(LOGS);
        if (p < C::FP - 1 || C::R > 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = lds[pb + ps * e];
        }
        if (InvRecentre::at(C::R, C::FP, 3, S0, (C::R > 0 ? 1 : 0) + (C::FP - 1 - p))) A::inv_pass_begin(v, cx);
#pragma unroll
        for (int u = 2; u >= 0; --u) {
            const int half = 4 >> u;
            if (p == 0 && u == 0) {  // last stage: fold N^-1
#pragma unroll
                for (int e = 0; e < 4; ++e) A::gs_last(v[e], v[e | 4], cx);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (e & half) continue;
                    A::gs(v[e], v[e | half], w[(1 << u) - 1 + (e >> (3 - u))], cx);
                }
            }
        }
        if (p > 0) {
            load_pass_tw8<LOGN, A>(w, itw, p - 1, t, 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) lds[pb + ps * e] = v[e];
            __syncthreads();
            HEFX_STAMP_AT(4 + (C::FP - 1 - p));
        }
    }
}

template <int LOGN, class A>
__device__ __forceinline__ void ntt8_inv_core(typename A::V (&v)[8], typename A::V *lds,
                                              const typename A::TW *__restrict__ itw, const typename A::Ctx &cx, int t)
{
    typename A::TW w[7];
    load_inv_first_tw8<LOGN, A>(w, itw, t);
    ntt8_inv_core_w<LOGN, A>(v, w, lds, itw, cx, t);
}

// ------------------------------------------------------------------------------------------------
// QUARTER rows for small batches: a row of 2^LOGN points is handled by FOUR workgroups of N/32 threads, eight
// coefficients per thread -- the same two waves per SIMD as the split-2 workgroups (one wave alone cannot keep a SIMD
// issuing: dependent VALU instructions need a second wave to fill the pipeline, which is why quarter rows with sixteen
// coefficients per thread were slower), but ~2300 instead of ~3600 instructions per thread on a 60-bit row and twice as
// many CUs per row.  The first TWO forward stages (gaps N/2, N/4; twiddles tw[1], tw[2 + h0]) are applied while
// loading -- every quarter reads the whole row and keeps one of four outputs per position -- and the first two
// inverse stages (gaps 1, 2) combine the four values of positions 4g..4g+3.  Used when a chunk cannot fill the chip:
// a lone rotation of a NAF chain, the lockstep chains of a few dot products.
// ------------------------------------------------------------------------------------------------
// The coefficient column a thread loads (ntt8_fwd_core_w's t0): rows in coefficient form are stored [evens | odds], so
// the first half of the workgroup takes the even columns and the second half the odd ones -- lane-adjacent reads are then
// adjacent in memory.  (With t0 = t neighbouring lanes alternated between the two halves of the row: the 32 loads of a
// thread took ~28 cycles each in the address coalescer, the last wave of a workgroup had its data ~1.6 us after the
// first, and every wave waited at the first barrier: tools/stamp_timeline.py.)
template <int LOGN>
__device__ __forceinline__ int quarter_fwd_lane(int t)
{
    constexpr int T = Ntt8Cfg<LOGN - 2>::T;
    return ((t & (T / 2 - 1)) << 1) | (t / (T / 2));
}

// ld(r, x0, x1, x2, x3): raw words of the coefficients idx_nat(t0,r) + m*N/4, m = 0..3, t0 = quarter_fwd_lane(t); on return
// f[r] = unfinished NTT value at qd*N/4 + idx_out(t,r), qd = 2*h0 + h1
// after_loads(): called once, right after the data loads are requested (the mod-down finish puts its epilogue operands
// in flight there)
template <int LOGN, class A, class LD, class HOOK = NoHook>
__device__ __forceinline__ void quarter_fwd_raw(typename A::V (&f)[8], const LD &ld, const InMode &mode,
                                                const ModConst &mc, u64 *lds, const typename A::TW *__restrict__ tw,
                                                const typename A::Ctx &cx, int t, int qd, const HOOK &after_loads = HOOK())
{
    const int h0 = qd >> 1, h1 = qd & 1;
    const typename A::TW w1 = A::half_twiddle(tw[1], cx, h0), w2 = A::half_twiddle(tw[2 + h0], cx, h1);
    HEFX_STAMP_AT(1);
    // the first radix-8 pass's twiddles travel with the data (one kernel at a time is latency, not registers -- except in
    // the 1024-thread workgroups of N = 32768, capped at 128 VGPRs: there they are fetched after the first two stages)
    constexpr bool PREW = Ntt8Cfg<LOGN - 2>::T < 1024;
    typename A::TW wp[7];
    if constexpr (PREW) load_pass_tw8<LOGN - 2, A>(wp, tw, 0, t, 4 + qd);
    auto stage = [&](auto red) {
        constexpr int RED = decltype(red)::value;
        u64 x[8][4];
#pragma unroll
        for (int r = 0; r < 8; ++r) ld(r, x[r][0], x[r][1], x[r][2], x[r][3]);
        after_loads();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            // stage 0 (gap N/2): half h0 of (x0, x2) and of (x1, x3); stage 1 (gap N/4): quarter h1 of that half
            const typename A::V a = A::ct_half(A::template input<RED>(x[r][0], mode, cx, mc),
                                               A::template input<RED>(x[r][2], mode, cx, mc), w1, cx);
            const typename A::V b = A::ct_half(A::template input<RED>(x[r][1], mode, cx, mc),
                                               A::template input<RED>(x[r][3], mode, cx, mc), w1, cx);
            f[r] = A::ct_sel(a, b, w2, cx);
        }
        HEFX_STAGE_FENCE();
        HEFX_STAMP_AT(2);
    };
    if (A::IS_F64 ? mode.red_f64 : mode.red_int) {
        if (A::IS_F64 && mode.below_2_61 && A::fast_wide(cx))
            stage(std::integral_constant<int, 2>{});
        else if (PREW && !A::IS_F64 && mode.lt2q)  // 32 Barrett reductions per thread were ~2 us of the 60-bit row's mod-down finish
            stage(std::integral_constant<int, 3>{});
        else
            stage(std::integral_constant<int, 1>{});
    } else {
        stage(std::integral_constant<int, 0>{});
    }
    if constexpr (!PREW) load_pass_tw8<LOGN - 2, A>(wp, tw, 0, t, 4 + qd);
    ntt8_fwd_core_w<LOGN - 2, A>(f, wp, reinterpret_cast<typename A::V *>(lds), tw, cx, t, 4 + qd,
                                 quarter_fwd_lane<LOGN>(t));
}

template <int LOGN, bool MACOP = false, class LD>
__device__ __forceinline__ void quarter_fwd(u64 (&v)[8], const LD &ld, const InMode &mode, u64 *lds, const NttTables &nt,
                                            const ModConst &mc, const ModConstF &mf, int t, int qd, int mac_slack = 0)
{
    if (mf.q != 0.0) {
        const ArithF64::Ctx cx = ArithF64::make(mf);
        double f[8];
        quarter_fwd_raw<LOGN, ArithF64, LD>(f, ld, mode, mc, lds, nt.twf, cx, t, qd);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = MACOP ? ArithF64::mac_operand(f[r], cx) : ArithF64::fwd_finish(f[r], cx);
    } else {
        fwd_int_dispatch(mc, [&](auto pol) {  // the 16q butterfly where the prime admits it
            using A = decltype(pol);
            const typename A::Ctx cx = A::make(mc);
            quarter_fwd_raw<LOGN, A, LD>(v, ld, mode, mc, lds, nt.tw, cx, t, qd);
            if (MACOP && mac_slack == 1) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = A::template mac_operand_lazy<1>(v[r], cx);
            } else if (MACOP && mac_slack >= 2) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = A::template mac_operand_lazy<2>(v[r], cx);
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = MACOP ? A::mac_operand(v[r], cx) : A::fwd_finish(v[r], cx);
            }
        });
    }
}

// ldp(j) = (value[2j], value[2j+1]) of the row (canonical NTT values); part = position mod 4 of the kept value; on return
// v[r] = coefficient 4*idx_nat(t,r) + part, canonical.
// Stages A and B are element-wise on the groups of four positions 4j..4j+3, so when the core starts on eight CONSECUTIVE
// groups per thread (R == 0: N = 4096, 16384) they run on lane-adjacent groups instead (idx_nat) and their results reach the
// core's layout through LDS, as in split_inv_a: with idx_out every lane of a 16-byte load sat in its own cache line, sixteen
// loads (forty with the twiddles of part 3) walked 256 contiguous bytes per lane, and the first phase of the inverse
// kernels took 8.3-8.9 us of their 14-15 (tools/stamp_timeline.py).  All operands of a batch of groups -- records, stage
// twiddles, the core's first twiddles -- are requested before the first is used.
template <int LOGN, class A, class LDP>
__device__ __forceinline__ void quarter_inv_a(u64 (&v)[8], const LDP &ldp, u64 *lds,
                                              const typename A::TW *__restrict__ itw, const typename A::Ctx &cx, int t,
                                              int part)
{
    using C = Ntt8Cfg<LOGN - 2>;
    constexpr int N = 1 << LOGN;
    constexpr int NB = C::T >= 1024 ? 2 : 1, BS = 8 / NB;  // 1024-thread workgroups: 128 VGPRs, two batches
    const int b0 = part & 1, b1 = part >> 1;
    typename A::V f[8];
    typename A::TW w[7];
    auto rec = [&](int r) { return C::R == 0 ? C::idx_nat(t, r) : C::idx_out(t, r); };
    HEFX_STAMP_AT(1);
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        ulonglong2 p0[BS], p1[BS];
        typename A::TW wa0[BS], wa1[BS], wb[BS];
#pragma unroll
        for (int r = 0; r < BS; ++r) {
            const int j = rec(BS * g + r);  // the four positions 4j .. 4j+3
            p0[r] = ldp(2 * j);
            p1[r] = ldp(2 * j + 1);
        }
        if (b0) {
#pragma unroll
            for (int r = 0; r < BS; ++r) {
                const int j = rec(BS * g + r);
                wa0[r] = itw[N / 2 + 2 * j];
                wa1[r] = itw[N / 2 + 2 * j + 1];
            }
        }
        if (b1) {
#pragma unroll
            for (int r = 0; r < BS; ++r) wb[r] = itw[N / 4 + rec(BS * g + r)];
        }
        if (g == NB - 1) load_inv_first_tw8<LOGN - 2, A>(w, itw, t);
#pragma unroll
        for (int r = 0; r < BS; ++r) {
            typename A::V e0, e1;  // stage A (gap 1): positions 4j + b0 and 4j + 2 + b0
            if (b0 == 0) {
                e0 = A::gs_half_sum(A::from_u64(p0[r].x), A::from_u64(p0[r].y), cx);
                e1 = A::gs_half_sum(A::from_u64(p1[r].x), A::from_u64(p1[r].y), cx);
            } else {
                e0 = A::gs_half_diff(A::from_u64(p0[r].x), A::from_u64(p0[r].y), wa0[r], cx);
                e1 = A::gs_half_diff(A::from_u64(p1[r].x), A::from_u64(p1[r].y), wa1[r], cx);
            }
            // stage B (gap 2): sum (b1 = 0) or twiddled difference (b1 = 1)
            f[BS * g + r] = b1 == 0 ? A::inv_add(e0, e1, cx) : A::inv_sub_mul(e0, e1, wb[r], cx);
        }
        HEFX_STAGE_FENCE();
    }
    HEFX_STAMP_AT(2);
    if constexpr (C::R == 0) {
        typename A::V *lf = reinterpret_cast<typename A::V *>(lds);
        const int pw = t + (t >> 3), pr = 9 * t;  // R == 0: phys(i) = i + (i >> 3)
#pragma unroll
        for (int r = 0; r < 8; ++r) lf[pw + (C::T + C::T / 8) * r] = f[r];  // phys(idx_nat(t, r))
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) f[r] = lf[pr + r];  // phys(8t + r): the words the core's first pass owns (and rewrites)
    }
    ntt8_inv_core_w<LOGN - 2, A>(f, w, reinterpret_cast<typename A::V *>(lds), itw, cx, t);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = A::inv_finish(f[r], cx);
}

// The REST of a quarter inverse transform whose first two stages (gaps 1 and 2) ran elsewhere -- in the epilogue of the
// pair path's key MAC (ks_pair_mac_kernel), which holds the four values of positions 4j..4j+3 in registers anyway:
// plane[j] = what stage B left for group j in this quarter's residue class, as a raw working value (A::raw: inverse lazy
// range / unfinished double).  One 8-byte word per coefficient instead of the two 16-byte records and three twiddles of
// quarter_inv_a.  On return v[r] = coefficient 4*idx_nat(t,r) + part, canonical.
template <int LOGN, class A>
__device__ __forceinline__ void quarter_inv_planes(u64 (&v)[8], const u64 *__restrict__ plane, u64 *lds,
                                                   const typename A::TW *__restrict__ itw, const typename A::Ctx &cx, int t)
{
    using C = Ntt8Cfg<LOGN - 2>;
    typename A::V f[8];
    typename A::TW w[7];
    if constexpr (C::R == 0) {  // lane-adjacent words, transposed into the core's layout through LDS (as quarter_inv_a)
#pragma unroll
        for (int r = 0; r < 8; ++r) f[r] = A::unraw(gld8(plane + C::idx_nat(t, r)));
    } else {  // G >= 2 adjacent words per lane: (r, r+1) are one 16-byte record
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            const ulonglong2 p = gld16(plane + C::idx_out(t, r));
            f[r] = A::unraw(p.x), f[r + 1] = A::unraw(p.y);
        }
    }
    load_inv_first_tw8<LOGN - 2, A>(w, itw, t);
    if constexpr (C::R == 0) {
        typename A::V *lf = reinterpret_cast<typename A::V *>(lds);
        const int pw = t + (t >> 3), pr = 9 * t;
#pragma unroll
        for (int r = 0; r < 8; ++r) lf[pw + (C::T + C::T / 8) * r] = f[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) f[r] = lf[pr + r];
    }
    ntt8_inv_core_w<LOGN - 2, A>(f, w, reinterpret_cast<typename A::V *>(lds), itw, cx, t);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = A::inv_finish(f[r], cx);
}

// Canonical (or merely 64-bit) words -> inputs of a forward core in policy A, eight at a time; the reduction the words
// need is workgroup-uniform (InMode), so ONE branch selects the variant for all eight (as quarter_fwd_raw does)
template <class A>
__device__ __forceinline__ void fwd_inputs8(typename A::V (&f)[8], const u64 (&v)[8], const InMode &mode,
                                            const typename A::Ctx &cx, const ModConst &mc)
{
    auto go = [&](auto red) {
        constexpr int RED = decltype(red)::value;
#pragma unroll
        for (int r = 0; r < 8; ++r) f[r] = A::template input<RED>(v[r], mode, cx, mc);
    };
    if constexpr (A::IS_F64) {
        if (!mode.red_f64)
            go(std::integral_constant<int, 0>{});
        else if (mode.below_2_61 && A::fast_wide(cx))
            go(std::integral_constant<int, 2>{});
        else
            go(std::integral_constant<int, 1>{});
    } else {
        if (!mode.red_int)
            go(std::integral_constant<int, 0>{});
        else if (mode.lt2q)
            go(std::integral_constant<int, 3>{});
        else
            go(std::integral_constant<int, 1>{});
    }
}

template <int LOGN, class LDP>
__device__ __forceinline__ void quarter_inv(u64 (&v)[8], const LDP &ldp, u64 *lds, const NttTables &nt,
                                            const ModConst &mc, const ModConstF &mf, int t, int part)
{
    if (mf.q != 0.0)
        quarter_inv_a<LOGN, ArithF64>(v, ldp, lds, nt.itwf, ArithF64::make(mf), t, part);
    else
        quarter_inv_a<LOGN, ArithU64>(v, ldp, lds, nt.itw, ArithU64::make(mc), t, part);
}

}  // namespace hefx

// hefx_ntt8.cuh -- forward NTT core with EIGHT coefficients per thread (radix-8 passes), used by the fused
// "digit NTT + key MAC" kernel: with 8 coefficients per thread the two 128-bit accumulators per coefficient
// (64 VGPRs) fit next to the transform's working set, which they do not with 16.
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
template <int LOGN, class A>
__device__ __forceinline__ void ntt8_fwd_core(typename A::V (&v)[8], typename A::V *lds,
                                              const typename A::TW *__restrict__ tw, const typename A::Ctx &cx, int t,
                                              int pre)
{
    using C = Ntt8Cfg<LOGN>;
    typename A::TW w[7];
    load_pass_tw8<LOGN, A>(w, tw, 0, t, pre);
#pragma unroll
    for (int p = 0; p < C::FP; ++p) {
        const int LOGS = LOGN - 3 * (p + 1);
        const int S = 1 << LOGS;
        const int b = t >> LOGS;
        const int base = b * (8 * S) + (t & (S - 1));
        if (p > 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = lds[C::phys(base + S * e)];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int half = 4 >> u;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e & half) continue;
                A::ct(v[e], v[e | half], w[(1 << u) - 1 + (e >> (3 - u))], cx);
            }
        }
        HEFX_STAGE_FENCE();
        if (p + 1 < C::FP)
            load_pass_tw8<LOGN, A>(w, tw, p + 1, t, pre);
        else if (C::R > 0)
            load_rem_tw8<LOGN, A>(w, tw, t, pre);
        if (p + 1 < C::FP || C::R > 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) lds[C::phys(base + S * e)] = v[e];
            __syncthreads();
        }
    }
    if (C::R > 0) {
#pragma unroll
        for (int c = 0; c < C::NG; ++c) {
            const int g = t + C::T * c;
#pragma unroll
            for (int e = 0; e < C::G; ++e) v[c * C::G + e] = lds[C::phys(g * C::G + e)];
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
                          w[c * (C::G - 1) + (1 << u) - 1 + (e >> (C::R - u))], cx);
                }
            }
        }
    }
}

// Half h of a forward transform of size 2^LOGN for one digit: ld(r, x, y) delivers coefficients idx_nat(t,r) and
// +N/2 (reduced as the policy requires); out[r] = canonical NTT value at h*N/2 + idx_out(t,r).
template <int LOGN, class A, class LD>
__device__ __forceinline__ void split8_fwd_a(u64 (&out)[8], const LD &ld, u64 *lds,
                                             const typename A::TW *__restrict__ tw, const typename A::Ctx &cx, int t,
                                             int h)
{
    typename A::V f[8];
    const typename A::TW w1 = tw[1];
    u64 x[8], y[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) ld(r, x[r], y[r]);
#pragma unroll
    for (int r = 0; r < 8; ++r) f[r] = A::ct_half(A::from_u64(x[r]), A::from_u64(y[r]), w1, cx, h);
    HEFX_STAGE_FENCE();
    ntt8_fwd_core<LOGN - 1, A>(f, reinterpret_cast<typename A::V *>(lds), tw, cx, t, 2 + h);
#pragma unroll
    for (int r = 0; r < 8; ++r) out[r] = A::fwd_finish(f[r], cx);
}

}  // namespace hefx

// hefx_ntt.cuh -- negacyclic NTT / inverse NTT cores for one RNS row held by ONE workgroup.
//
// Semantics (SEAL 3.4.5 util/smallntt): twiddle table tw[bitrev(i)] = psi^i with
// psi the minimal primitive 2N-th root; forward = Cooley-Tukey, natural in -> bit-reversed out,
// out[i] = a(psi^(2*bitrev(i)+1)); inverse = Gentleman-Sande with N^-1 folded into the last stage.
//
// gfx950 mapping: N/16 threads, 16 coefficients per thread in registers; four radix-2 stages per pass
// are done in registers (radix-16), passes exchange through LDS (one barrier per exchange: a thread
// only ever overwrites the LDS words it read itself).  N*8*(17/16) bytes of LDS: 136 KiB at N=16384
// (one row per CU), 68 KiB at N=8192 (two rows per CU).  Twiddles of the first pass are workgroup-
// uniform and of the second pass wave-uniform (scalar loads); later passes load {w, w_shoup} as one
// 16-byte vector load per butterfly from L2.  Butterflies are Harvey lazy ([0,4q) forward, [0,2q)
// inverse) with Shoup twiddles; outputs are canonical.
#pragma once
#include "hefx_modarith.cuh"

namespace hefx {

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
    // coefficient index of register r in the pass-0 layout (coalesced: lane-consecutive words)
    __device__ static __forceinline__ int idx_nat(int t, int r) { return t + T * r; }
    // coefficient index of register r in the layout the forward transform ends in
    __device__ static __forceinline__ int idx_out(int t, int r)
    {
        if (R == 0) return t * 16 + r;
        return (t + T * (r >> R)) * G + (r & (G - 1));
    }
};

// forward butterfly, inputs/outputs in [0,4q)
__device__ __forceinline__ void ct_bfly(u64 &x, u64 &y, u64 w, u64 ws, u64 q, u64 two_q)
{
    u64 a = csub(x, two_q);
    u64 t = shoup_lazy(y, w, ws, q);
    x = a + t;
    y = a + two_q - t;
}

// inverse butterfly, inputs/outputs in [0,2q)
__device__ __forceinline__ void gs_bfly(u64 &x, u64 &y, u64 w, u64 ws, u64 q, u64 two_q)
{
    u64 s = csub(x + y, two_q);
    u64 d = x + two_q - y;
    x = s;
    y = shoup_lazy(d, w, ws, q);
}

// v[r] holds coefficient idx_nat(t,r) (any value < 4q) on entry and NTT value idx_out(t,r) in [0,q) on exit.
template <int LOGN>
__device__ __forceinline__ void ntt_fwd_core(u64 (&v)[16], u64 *lds, const ulonglong2 *__restrict__ tw, u64 q,
                                             int t)
{
    using C = NttCfg<LOGN>;
    const u64 two_q = q << 1;
#pragma unroll
    for (int p = 0; p < C::FP; ++p) {
        const int LOGS = LOGN - 4 * (p + 1);
        const int S = 1 << LOGS;
        int b = t >> LOGS;
        const int base = b * (16 * S) + (t & (S - 1));
        if (p > 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = lds[C::phys(base + S * e)];
        }
        if (LOGS >= 6) b = __builtin_amdgcn_readfirstlane(b);  // wave-uniform -> scalar twiddle loads
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int half = 8 >> u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (e & half) continue;
                const ulonglong2 w = tw[(1 << (4 * p + u)) + (b << u) + (e >> (4 - u))];
                ct_bfly(v[e], v[e | half], w.x, w.y, q, two_q);
            }
        }
        if (p + 1 < C::FP || C::R > 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) lds[C::phys(base + S * e)] = v[e];
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
                const int g = t + C::T * c;
#pragma unroll
                for (int e = 0; e < C::G; ++e) {
                    if (e & half) continue;
                    const ulonglong2 w = tw[(1 << (4 * C::FP + u)) + (g << u) + (e >> (C::R - u))];
                    ct_bfly(v[c * C::G + e], v[c * C::G + (e | half)], w.x, w.y, q, two_q);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = csub(csub(v[e], two_q), q);
}

// v[r] holds NTT value idx_out(t,r) in [0,2q) on entry and coefficient idx_nat(t,r) in [0,q) on exit.
// itw[idx] = tw[idx]^-1 (same indexing); mc.ilw = itw[1]*N^-1, mc.ninv = N^-1 (folded last stage).
template <int LOGN>
__device__ __forceinline__ void ntt_inv_core(u64 (&v)[16], u64 *lds, const ulonglong2 *__restrict__ itw,
                                             const ModConst &mc, int t)
{
    using C = NttCfg<LOGN>;
    const u64 q = mc.q, two_q = q << 1;
    if (C::R > 0) {
#pragma unroll
        for (int u = C::R - 1; u >= 0; --u) {
            const int half = C::G >> (u + 1);
#pragma unroll
            for (int c = 0; c < C::NG; ++c) {
                const int g = t + C::T * c;
#pragma unroll
                for (int e = 0; e < C::G; ++e) {
                    if (e & half) continue;
                    const ulonglong2 w = itw[(1 << (4 * C::FP + u)) + (g << u) + (e >> (C::R - u))];
                    gs_bfly(v[c * C::G + e], v[c * C::G + (e | half)], w.x, w.y, q, two_q);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < C::NG; ++c) {
            const int g = t + C::T * c;
#pragma unroll
            for (int e = 0; e < C::G; ++e) lds[C::phys(g * C::G + e)] = v[c * C::G + e];
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = C::FP - 1; p >= 0; --p) {
        const int LOGS = LOGN - 4 * (p + 1);
        const int S = 1 << LOGS;
        int b = t >> LOGS;
        const int base = b * (16 * S) + (t & (S - 1));
        if (p < C::FP - 1 || C::R > 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = lds[C::phys(base + S * e)];
        }
        if (LOGS >= 6) b = __builtin_amdgcn_readfirstlane(b);
#pragma unroll
        for (int u = 3; u >= 0; --u) {
            const int half = 8 >> u;
            if (p == 0 && u == 0) {  // last stage: fold N^-1
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    u64 s = csub(v[e] + v[e | 8], two_q);
                    u64 d = v[e] + two_q - v[e | 8];
                    v[e] = shoup_lazy(s, mc.ninv, mc.ninv_s, q);
                    v[e | 8] = shoup_lazy(d, mc.ilw, mc.ilw_s, q);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (e & half) continue;
                    const ulonglong2 w = itw[(1 << (4 * p + u)) + (b << u) + (e >> (4 - u))];
                    gs_bfly(v[e], v[e | half], w.x, w.y, q, two_q);
                }
            }
        }
        if (p > 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) lds[C::phys(base + S * e)] = v[e];
            __syncthreads();
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = csub(v[e], q);
}

}  // namespace hefx

// hefx_encode.hip -- CKKSEncoder::encode on the GPU (App. A.12): canonical embedding with
// slot i <-> root zeta^(3^i), zeta = exp(2 pi i / 2N).  Reference call sites: matrix_mult_benchmark.cpp:291-323,
// logistic_regression_ckks.cpp:222-225,302-305 (one-hot masks inside the LR hot loop), helper.h:333-343.
//
// Math.  The plaintext polynomial p (real, degree < N) has p(zeta^(2r+1)) = A_r for all r < N, where A_r = v_i at
// r = (3^i - 1)/2 and A_(N-1-r) = conj(A_r).  Pairing r with N-1-r:
//     p_k = (2/N) * Re( zeta^(-k) * sum_{r < N/2} A_r * exp(-2 pi i r k / N) ),
// and splitting k = 2k' + kappa turns the inner sum into TWO complex FFTs of N/2 points (kappa = 0, 1) of
// A_r * exp(-2 pi i r kappa / N).  One workgroup per (vector, kappa): N/16 threads, 8 complex points per thread,
// radix-8 decimation-in-frequency passes through LDS (128 KiB of double2 at N = 16384), output scattered from
// bit-reversed order.  Then p_k * scale is rounded half away from zero (std::round, as SEAL does), reduced into
// every RNS row and the rows go through the regular forward NTT.
//
// Parity: floating point -- not bit-exact with any other FFT; |coefficient difference| <= 1 against the CPU
// oracle on a small fraction of coefficients (tested), decode(encode(v)) == v to ~1e-9 at scale 2^40.
#include "hefx_internal.h"

namespace hefx {

__device__ __forceinline__ double2 cmul(double2 a, double2 b)
{
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

template <int LM>  // M = 2^LM = N/2 complex points
struct FftCfg {
    static constexpr int M = 1 << LM;
    static constexpr int T = M / 8;
    static constexpr int FP = LM / 3;
    static constexpr int R = LM % 3;
};

// one radix-2 DIF butterfly of block length 2*gap on positions (idx, idx+gap)
__device__ __forceinline__ void dif(double2 &u, double2 &v, const double2 *__restrict__ w, int idx, int gap, int M)
{
    const double2 s = make_double2(u.x + v.x, u.y + v.y);
    const double2 d = make_double2(u.x - v.x, u.y - v.y);
    const int j = idx & (gap - 1);
    u = s;
    v = cmul(d, w[j * (M / (2 * gap))]);
}

// SPLIT (N = 32768: 16384 complex points do not fit LDS): the first DIF stage is applied while loading -- half h of
// an FFT of FM = 2*M points keeps a0 + a1 (h = 0: even frequencies) or (a0 - a1) * w_FM^r (h = 1: odd frequencies) --
// and the workgroup runs the remaining M-point transform; blockIdx.x = 2*kappa + h.
template <int LM, bool SPLIT>
__global__ __launch_bounds__(FftCfg<LM>::T) void ckks_encode_kernel(DevTables T, EncodeTables E, const double *re,
                                                                   const double *im, int nvalues, double scale, int L,
                                                                   u64 *out /* [count][L][N] coefficient form */)
{
    using C = FftCfg<LM>;
    extern __shared__ __align__(16) double2 fl[];
    const int t = threadIdx.x, vec = blockIdx.y;
    const int kappa = SPLIT ? blockIdx.x >> 1 : blockIdx.x, hh = SPLIT ? blockIdx.x & 1 : 0;
    constexpr int FM = SPLIT ? 2 * C::M : C::M;  // points of the whole FFT = N/2 = size the twiddle table is built for
    const int N = 2 * FM;
    const double *vre = re + (size_t)vec * nvalues;
    const double *vim = im ? im + (size_t)vec * nvalues : nullptr;
    auto load = [&](int r) {  // A_r, pre-twisted for kappa = 1
        const int s = E.slot[r];
        double2 a = make_double2(0.0, 0.0);
        if (s >= 0 && (s >> 1) < nvalues) {
            a.x = vre[s >> 1];
            a.y = vim ? vim[s >> 1] : 0.0;
            if (s & 1) a.y = -a.y;
        }
        return kappa ? cmul(a, E.pre[r]) : a;
    };
    double2 v[8];
    // natural order, r = t + T*e
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int r = t + C::T * e;
        if (SPLIT) {
            const double2 a0 = load(r), a1 = load(r + C::M);
            v[e] = hh ? cmul(make_double2(a0.x - a1.x, a0.y - a1.y), E.wfft[r])
                      : make_double2(a0.x + a1.x, a0.y + a1.y);
        } else {
            v[e] = load(r);
        }
    }
    // radix-8 DIF passes: pass p works on stride S = M / 8^(p+1); gaps 4S, 2S, S
#pragma unroll
    for (int p = 0; p < C::FP; ++p) {
        const int LOGS = LM - 3 * (p + 1);
        const int S = 1 << LOGS;
        const int b = t >> LOGS;
        const int base = b * (8 * S) + (t & (S - 1));
        if (p > 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fl[base + S * e];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int half = 4 >> u;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (e & half) continue;
                dif(v[e], v[e | half], E.wfft, base + S * e, half * S, FM);
            }
        }
        if (p + 1 < C::FP || C::R > 0) {
            __syncthreads();  // everyone has read its inputs of this pass
#pragma unroll
            for (int e = 0; e < 8; ++e) fl[base + S * e] = v[e];
            __syncthreads();
        }
    }
    if (C::R > 0) {  // remaining R stages on groups of G = 2^R contiguous points
        constexpr int G = 1 << C::R, NG = 8 / G;
#pragma unroll
        for (int c = 0; c < NG; ++c)
#pragma unroll
            for (int e = 0; e < G; ++e) v[c * G + e] = fl[(t + C::T * c) * G + e];
#pragma unroll
        for (int u = 0; u < C::R; ++u) {
            const int half = G >> (u + 1);
#pragma unroll
            for (int c = 0; c < NG; ++c)
#pragma unroll
                for (int e = 0; e < G; ++e) {
                    if (e & half) continue;
                    dif(v[c * G + e], v[c * G + (e | half)], E.wfft, (t + C::T * c) * G + e, half, FM);
                }
        }
    }
    // position pos holds X[bitrev(pos)]: k' = bitrev(pos), k = 2k' + kappa
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int pos;
        if (C::R == 0) {
            pos = t * 8 + e;
        } else {
            constexpr int G = 1 << C::R;
            pos = (t + C::T * (e / G)) * G + (e % G);
        }
        int kp = (int)(__brev((unsigned)pos) >> (32 - LM));
        if (SPLIT) kp = 2 * kp + hh;
        const int k = 2 * kp + kappa;
        const double2 z = cmul(v[e], E.post[k]);
        const double co = __builtin_round(z.x * (2.0 / (double)N) * scale);
        const bool neg = co < 0.0;
        const u64 mag = (u64)__builtin_fabs(co);
        for (int j = 0; j < L; ++j) {
            const ModConst mc = T.mods[j];
            const u64 r = barrett64(mag, mc.q, mc.r1);
            out[((size_t)vec * L + j) * N + k] = neg ? (r ? mc.q - r : 0) : r;
        }
    }
}

hipError_t launch_encode(const DevTables &T, const EncodeTables &E, const double *re, const double *im, int nvalues,
                         int count, double scale, int L, u64 *out, hipStream_t s)
{
    const int lm = T.logn - 1;
#define LAUNCH(LMV, SPL)                                                                                         \
    {                                                                                                            \
        const size_t lds = sizeof(double2) * (size_t)FftCfg<LMV>::M;                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ckks_encode_kernel<LMV, SPL>),                  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
        hipLaunchKernelGGL((ckks_encode_kernel<LMV, SPL>), dim3(SPL ? 4 : 2, count), dim3(FftCfg<LMV>::T), lds, s, T, \
                           E, re, im, nvalues, scale, L, out);                                                   \
    }
    switch (lm) {
        case 9: LAUNCH(9, false) break;
        case 10: LAUNCH(10, false) break;
        case 11: LAUNCH(11, false) break;
        case 12: LAUNCH(12, false) break;
        case 13: LAUNCH(13, false) break;
        case 14: LAUNCH(13, true) break;  // N = 32768: two half-size workgroups per (vector, kappa)
        default: return hipErrorInvalidValue;
    }
#undef LAUNCH
    return hipGetLastError();
}

}  // namespace hefx

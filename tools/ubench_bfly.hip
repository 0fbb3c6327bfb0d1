// Micro-benchmark: the integer Cooley-Tukey butterfly of hefx_ntt.cuh (ArithU64::ct) in its two forms, on gfx950.
//   form 0  what the compiler makes of  t = y*w - hi(y*w')*q;  a = csub(x, 4q) on the borrow;  x' = a + t;  y' = a + 4q - t
//   form 1  the tree's form: t = y*w + h*(2^64 - q) as six v_mad_u64_u32 and one 32-bit add, a = x + (2^64 - 4q) / compare /
//           two selects (hefx_modarith.cuh: mul_sub_lo64, csubn)
// Every thread runs radix-16 passes over sixteen registers with one (uniform) twiddle per round; 4 waves per SIMD on every
// CU, like the transform kernels.  Prints nanoseconds and SIMD cycles (at 2.4 GHz) per wave-butterfly.
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_bfly.hip -o gpurun_out/ubench_bfly && gpurun_out/ubench_bfly
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;

__device__ __forceinline__ u64 under2(u64 x, u64 ws)
{
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), w0 = (uint32_t)ws, w1 = (uint32_t)(ws >> 32);
    return (u64)x1 * w1 + ((u64)__umulhi(x0, w1) + (u64)__umulhi(x1, w0));
}
// the same under-estimate with the two cross high words taken from full 64-bit multiply-adds instead of v_mul_hi_u32
// (quarter-rate on gfx950: 8.5 cycles per wave-instruction against 5.65 for v_mad_u64_u32, profiles/r01_valu_issue_rates.txt)
__device__ __forceinline__ u64 under2m(u64 x, u64 ws)
{
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), w0 = (uint32_t)ws, w1 = (uint32_t)(ws >> 32);
    u64 t = (u64)x0 * w1, u = (u64)x1 * w0;
    asm("" : "+v"(t), "+v"(u));  // keep the full products: the optimiser would go back to v_mul_hi_u32
    return (u64)x1 * w1 + ((t >> 32) + (u >> 32));
}
__device__ __forceinline__ u64 csub(u64 x, u64 m)
{
    u64 d;
    return __builtin_usubll_overflow(x, m, &d) ? x : d;
}
__device__ __forceinline__ u64 csubn(u64 x, u64 nm)
{
    const u64 d = x + nm;
    return d > x ? x : d;
}
__device__ __forceinline__ u64 mul_sub_lo64(u64 x, u64 w, u64 h, u64 nq)
{
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint32_t h0 = (uint32_t)h, h1 = (uint32_t)(h >> 32);
    uint32_t n0 = (uint32_t)nq, n1 = (uint32_t)(nq >> 32);
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

// Pseudo-Mersenne product for q = 2^60 - delta (SEAL's 60-bit primes: delta = 2^60 - q < 2^23 for every set of
// tests/golden/appendix_b.json): y < 2^63, w < q  ->  y*w mod q in [0, 2q), from the twiddle alone (no Shoup companion).
//   P = y*w < 2^123 by four multiply-adds (c = y1*w0 + b cannot wrap: y1 < 2^31, b < 2^60 + 2^32);
//   h = P >> 60, fold: h*delta + (P mod 2^60) = t1*2^32 + lo32(t0), h2 = that >> 60, r = (that mod 2^60) + h2*delta < 2^60 + 2^49.
__device__ __forceinline__ u64 pm_mul(u64 y, u64 w, uint32_t delta)
{
    const uint32_t y0 = (uint32_t)y, y1 = (uint32_t)(y >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    u64 a = (u64)y0 * w0;
    asm("" : "+v"(a));  // keep the full product: otherwise v_mul_lo + v_mul_hi (quarter rate) instead of one multiply-add
    u64 b = (u64)y0 * w1 + (a >> 32);
    u64 c = (u64)y1 * w0 + b;
    asm("" : "+v"(c));
    const u64 hi = (u64)y1 * w1 + (c >> 32);
    const uint32_t plo_hi = (uint32_t)c;
    const uint32_t h_lo = __builtin_amdgcn_alignbit((uint32_t)hi, plo_hi, 28);
    const uint32_t h_hi = __builtin_amdgcn_alignbit((uint32_t)(hi >> 32), (uint32_t)hi, 28);
    const u64 plo60 = ((u64)(plo_hi & 0x0FFFFFFFu) << 32) | (uint32_t)a;
    u64 t0 = (u64)h_lo * delta + plo60;
    asm("" : "+v"(t0));
    const u64 t1 = (u64)h_hi * delta + (t0 >> 32);
    const uint32_t h2 = __builtin_amdgcn_alignbit((uint32_t)(t1 >> 32), (uint32_t)t1, 28);
    const u64 low60 = ((u64)((uint32_t)t1 & 0x0FFFFFFFu) << 32) | (uint32_t)t0;
    return (u64)h2 * delta + low60;
}

template <int FORM>
__global__ __launch_bounds__(256) void k(u64 *p, const ulonglong2 *tw, u64 q, u64 nq, int rounds)
{
    u64 v[16];
    u64 *mine = p + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    for (int i = 0; i < 16; ++i) v[i] = mine[i];
    const u64 q4 = 4 * q, nq4 = nq << 2, nq8 = nq << 3;
    // forms 3, 4: values in [0,16q) (q < 2^60): the conditional subtraction (by 8q) only in every other stage --
    // stage without: in < 12q, out < 16q; stage with: in < 16q -> < 8q, out < 12q
    constexpr bool LAZY16 = FORM >= 3, MADQ = FORM == 2 || FORM == 3;
    for (int r = 0; r < rounds; ++r) {
        const ulonglong2 w = tw[r & 63];
#pragma unroll
        for (int s = 8; s >= 1; s >>= 1)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (!(i & s)) {
                    u64 a, t;
                    if (FORM == 5) {
                        // values in [0,8q) = below 2^63: a = csub(x, 4q) in every other stage (in < 8q -> < 4q, out < 6q;
                        // next stage in < 6q, out < 8q); the product is below 2q, so y' = a + 2q - t
                        a = (s == 8 || s == 2) ? v[i] : csubn(v[i], nq4);
                        t = pm_mul(v[i + s], w.x, (uint32_t)nq);
                        v[i] = a + t;
                        v[i + s] = a + 2 * q - t;
                        continue;
                    }
                    if (FORM == 0) {
                        a = csub(v[i], q4);
                        t = v[i + s] * w.x - under2(v[i + s], w.y) * q;
                    } else {
                        if (LAZY16)
                            a = (s == 8 || s == 2) ? v[i] : csubn(v[i], nq8);
                        else
                            a = csubn(v[i], nq4);
                        t = mul_sub_lo64(v[i + s], w.x, MADQ ? under2m(v[i + s], w.y) : under2(v[i + s], w.y), nq);
                    }
                    v[i] = a + t;
                    v[i + s] = a + q4 - t;
                }
    }
    for (int i = 0; i < 16; ++i) mine[i] = v[i];
}

template <int FORM>
static double run(u64 *d, const ulonglong2 *tw, u64 q, int rounds, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, d, tw, q, 0 - q, rounds);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, d, tw, q, 0 - q, rounds);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main()
{
    const u64 q = 0xffffffffffd8001ull;
    const int blocks = 256 * 4, rounds = 512;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    u64 *d;
    ulonglong2 *tw;
    hipMalloc(&d, (size_t)blocks * 256 * 16 * 8);
    hipMalloc(&tw, 64 * sizeof(ulonglong2));
    std::vector<u64> h((size_t)blocks * 256 * 16);
    u64 s = 88172645463325252ull;
    for (auto &x : h) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        x = s % q;
    }
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    std::vector<ulonglong2> ht(64);
    for (auto &t : ht) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        t.x = s % q;
        t.y = (u64)(((unsigned __int128)t.x << 64) / q);
    }
    hipMemcpy(tw, ht.data(), 64 * sizeof(ulonglong2), hipMemcpyHostToDevice);
    const double wave_bf = (double)blocks * 4 * rounds * 32 / (256.0 * 4);  // wave-butterflies per SIMD
    // every form computes the same residues: check them against form 1 on a short run from the same input
    auto residues = [&](auto runner) {
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        runner();
        hipDeviceSynchronize();
        std::vector<u64> o(4096);
        hipMemcpy(o.data(), d, o.size() * 8, hipMemcpyDeviceToHost);
        for (auto &x : o) x %= q;
        return o;
    };
    const auto r1 = residues([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, tw, q, 0 - q, 7); });
    const auto r2 = residues([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, tw, q, 0 - q, 7); });
    const auto r3 = residues([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, tw, q, 0 - q, 7); });
    const auto r4 = residues([&] { hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, tw, q, 0 - q, 7); });
    const auto r5 = residues([&] { hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, d, tw, q, 0 - q, 7); });
    printf("residues equal to form 1: form 2 %d, form 3 %d, form 4 %d, form 5 (pseudo-Mersenne) %d\n", r1 == r2, r1 == r3, r1 == r4,
           r1 == r5);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    const double m0 = run<0>(d, tw, q, rounds, blocks), m1 = run<1>(d, tw, q, rounds, blocks);
    for (int f = 2; f <= 4; ++f) {
        const double m = f == 2 ? run<2>(d, tw, q, rounds, blocks) : f == 3 ? run<3>(d, tw, q, rounds, blocks) : run<4>(d, tw, q, rounds, blocks);
        printf("form %d (%s%s): %.3f ms -> %.1f ns, %.1f SIMD cycles @2.4 GHz per wave-butterfly\n", f,
               f == 2 || f == 3 ? "quotient by multiply-adds" : "v_mul_hi quotient", f >= 3 ? ", 16q lazy range" : "", m,
               m * 1e6 / wave_bf, m * 1e-3 * 2.4e9 / wave_bf);
    }
    {
        const double m = run<5>(d, tw, q, rounds, blocks);
        printf("form 5 (pseudo-Mersenne fold, 8-byte twiddle, 8q lazy range): %.3f ms -> %.1f ns, %.1f SIMD cycles @2.4 GHz per wave-butterfly\n",
               m, m * 1e6 / wave_bf, m * 1e-3 * 2.4e9 / wave_bf);
    }
    printf("compiler's form : %.3f ms -> %.1f ns, %.1f SIMD cycles @2.4 GHz per wave-butterfly\n", m0, m0 * 1e6 / wave_bf,
           m0 * 1e-3 * 2.4e9 / wave_bf);
    printf("multiply-add form: %.3f ms -> %.1f ns, %.1f SIMD cycles @2.4 GHz per wave-butterfly\n", m1, m1 * 1e6 / wave_bf,
           m1 * 1e-3 * 2.4e9 / wave_bf);
    return 0;
}

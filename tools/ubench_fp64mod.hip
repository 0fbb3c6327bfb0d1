// Micro-benchmark: exact modmul for primes < 2^42 with FP64 FMA vs integer Shoup (gfx950), plus exactness check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef unsigned long long u64;
#define ITERS 2048
__device__ __forceinline__ double modmul_f64(double y, double w, double winv, double q)
{
    double h = y * w;
    double l = fma(y, w, -h);
    double c = __builtin_rint(y * winv);
    double s = fma(-c, q, h);
    return s + l;
}
__device__ __forceinline__ double modmul_f64_magic(double y, double w, double winv, double q)
{
    const double M = 6755399441055744.0;  // 1.5 * 2^52
    double h = y * w;
    double l = fma(y, w, -h);
    double c = fma(y, winv, M) - M;
    double s = fma(-c, q, h);
    return s + l;
}
template <int OP>
__global__ void k(double *out, double q, double w, double winv)
{
    double a0 = threadIdx.x + 1.0, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    for (int i = 0; i < ITERS; ++i) {
        if (OP == 0) {
            a0 = modmul_f64(a0, w, winv, q); a1 = modmul_f64(a1, w, winv, q);
            a2 = modmul_f64(a2, w, winv, q); a3 = modmul_f64(a3, w, winv, q);
        } else if (OP == 1) {
            a0 = modmul_f64_magic(a0, w, winv, q); a1 = modmul_f64_magic(a1, w, winv, q);
            a2 = modmul_f64_magic(a2, w, winv, q); a3 = modmul_f64_magic(a3, w, winv, q);
        } else {  // CT butterfly pair: t = y*w; x' = x + t; y' = x - t
            double t0 = modmul_f64(a1, w, winv, q); double n0 = a0 + t0; a1 = a0 - t0; a0 = n0;
            double t1 = modmul_f64(a3, w, winv, q); double n1 = a2 + t1; a3 = a2 - t1; a2 = n1;
            if ((i & 7) == 7) { a0 -= q * __builtin_rint(a0 / q); a2 -= q * __builtin_rint(a2 / q); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int OP>
void run(const char *name, int groups)
{
    double *d;
    hipMalloc(&d, 1024 * 256 * 8);
    const double q = 1099511480321.0;  // 0xffffb20001
    const double w = 48411826.0, winv = w / q;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, q, w, winv);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, q, w, winv);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    double per_simd = 4.0 * ITERS * groups;  // 4 waves per SIMD
    printf("%-24s %.3f ms -> %.1f cycles per modmul(or butterfly) per wave per SIMD @2.4GHz\n", name, ms,
           ms * 1e-3 * 2.4e9 / per_simd);
    hipFree(d);
}
// exactness: compare fp64 modmul with 128-bit integer arithmetic on random inputs
__global__ void check(const long long *ys, const u64 *ws, u64 q, int n, int *bad)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double qd = (double)q, w = (double)ws[i], winv = w / qd, y = (double)ys[i];
    double t = modmul_f64(y, w, winv, qd);
    double t2 = modmul_f64_magic(y, w, winv, qd);
    __int128 prod = (__int128)ys[i] * (__int128)ws[i];
    long long tt = (long long)t;
    __int128 diff = prod - (__int128)tt;
    bool ok = (diff % (__int128)q == 0) && fabs(t) < 0.53 * qd && t == t2 && (double)tt == t;
    if (!ok) {
        int n0 = atomicAdd(bad, 1);
        if (n0 < 6)
            printf("bad i=%d y=%lld w=%llu t=%.1f t2=%.1f cong=%d bound=%d same=%d int=%d\n", i, ys[i], ws[i], t, t2,
                   (int)(diff % (__int128)q == 0), (int)(fabs(t) < 0.53 * qd), (int)(t == t2), (int)((double)tt == t));
    }
}
int main()
{
    run<0>("fp64 modmul (rndne)", 4);
    run<1>("fp64 modmul (magic)", 4);
    run<2>("fp64 CT butterfly", 2);
    const int n = 1 << 22;
    const u64 q = 0xffffe80001ull;
    long long *ys; u64 *ws; int *bad;
    hipMallocManaged(&ys, n * 8); hipMallocManaged(&ws, n * 8); hipMallocManaged(&bad, 4);
    *bad = 0;
    u64 s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        long long y = (long long)(s % (u64)(1ull << 46)) - (1ll << 45);
        if (i < 1000) y = (i & 1) ? (1ll << 45) - i : -(1ll << 45) + i;
        ys[i] = y;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        ws[i] = (i < 64) ? q - 1 - i : s % q;
    }
    hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, 0, ys, ws, q, n, bad);
    hipDeviceSynchronize();
    printf("exactness: %d bad of %d\n", *bad, n);
    return 0;
}

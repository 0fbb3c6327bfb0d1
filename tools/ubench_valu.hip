// Micro-benchmark: issue rate of the integer instructions the 64-bit modmul is built from (gfx950).
// Usage: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o gpurun_out/ubench && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
#define ITERS 4096
template <int OP>
__global__ void k(u64 *out, u64 seed)
{
    u64 a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    u64 b = seed * 11 + 5;
    uint32_t x0 = (uint32_t)a0, x1 = (uint32_t)a1, x2 = (uint32_t)a2, x3 = (uint32_t)a3, y = (uint32_t)b | 1;
    double d0 = (double)a0, d1 = (double)a1, d2 = (double)a2, d3 = (double)a3, e = 1.0000001;
    for (int i = 0; i < ITERS; ++i) {
        if (OP == 0) {  // v_mad_u64_u32
            a0 = (u64)(uint32_t)a0 * y + a0; a1 = (u64)(uint32_t)a1 * y + a1;
            a2 = (u64)(uint32_t)a2 * y + a2; a3 = (u64)(uint32_t)a3 * y + a3;
        } else if (OP == 1) {  // v_mul_lo_u32
            x0 = x0 * y + 1; x1 = x1 * y + 1; x2 = x2 * y + 1; x3 = x3 * y + 1;
        } else if (OP == 2) {  // v_mul_hi_u32
            x0 = __umulhi(x0, y) + 7; x1 = __umulhi(x1, y) + 7; x2 = __umulhi(x2, y) + 7; x3 = __umulhi(x3, y) + 7;
        } else if (OP == 3) {  // 64-bit add
            a0 += b; a1 += a0; a2 += a1; a3 += a2;
        } else if (OP == 4) {  // v_fma_f64
            d0 = d0 * e + d1; d1 = d1 * e + d2; d2 = d2 * e + d3; d3 = d3 * e + d0;
        } else if (OP == 5) {  // full Shoup lazy modmul
            u64 q = 0xffffffffffd8001ull;
            a0 = a0 * b - __umul64hi(a0, seed) * q; a1 = a1 * b - __umul64hi(a1, seed) * q;
            a2 = a2 * b - __umul64hi(a2, seed) * q; a3 = a3 * b - __umul64hi(a3, seed) * q;
        } else if (OP == 6) {  // v_mul_u32_u24
            x0 = __umul24(x0, y) + 1; x1 = __umul24(x1, y) + 1; x2 = __umul24(x2, y) + 1; x3 = __umul24(x3, y) + 1;
        } else if (OP == 7) {  // 32-bit add
            x0 += y; x1 += x0; x2 += x1; x3 += x2;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + x0 + x1 + x2 + x3 + (u64)(d0 + d1 + d2 + d3);
}
template <int OP>
void run(const char *name, int per_iter)
{
    u64 *d;
    hipMalloc(&d, 256 * 16 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;  // 4 blocks of 256 threads per CU -> 4 waves / SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345ull + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    double wave_instr = (double)blocks * 4 /*waves/block*/ * ITERS * per_iter;
    double per_simd = wave_instr / (256.0 * 4);
    // 4 waves per SIMD each issue ITERS*per_iter instructions of this kind (plus loop overhead)
    printf("%-16s %.3f ms  -> %.2f cycles per wave-instruction per SIMD @2.4GHz\n", name, ms,
           ms * 1e-3 * 2.4e9 / per_simd);
    hipFree(d);
}
int main()
{
    run<0>("mad_u64_u32", 4);
    run<1>("mul_lo_u32", 4);
    run<2>("mul_hi_u32", 4);
    run<3>("add_u64", 4);
    run<4>("fma_f64", 4);
    run<5>("shoup_modmul", 4);
    run<6>("mul_u32_u24", 4);
    run<7>("add_u32", 4);
    return 0;
}

// Engine clock seen by a lone workgroup vs a full-chip launch: a chain of N dependent full-rate VALU instructions takes
// 4*N cycles per wave (wave64 on a 16-lane SIMD), so wall time / (4*N) is the clock period.
// usage: hipcc --offload-arch=gfx950 -O3 tools/ubench_clock.hip -o gpurun_out/ubench_clock && gpurun_out/ubench_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS (1 << 16)
__global__ void chain(uint32_t *out, uint32_t seed)
{
    uint32_t x = seed + threadIdx.x;
#pragma unroll 16
    for (int i = 0; i < ITERS; ++i) x = x * 3 + (x >> 1);  // v_lshrrev + v_mad_u32_u24-free: 2-3 dependent ops
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
static double run(int blocks, int threads, int reps)
{
    uint32_t *d;
    hipMalloc(&d, (size_t)blocks * threads * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(chain, dim3(blocks), dim3(threads), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(chain, dim3(blocks), dim3(threads), 0, 0, d, (uint32_t)r);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipFree(d);
    return ms / reps;
}
int main()
{
    const double one = run(1, 64, 20), few = run(16, 512, 20), full = run(256 * 8, 256, 20);
    printf("1 wave alone      : %.3f ms per launch\n", one);
    printf("16 x 512 threads  : %.3f ms per launch (2 waves / SIMD on 16 CUs)\n", few);
    printf("2048 x 256 threads: %.3f ms per launch (8 waves / SIMD on every CU)\n", full);
    printf("ratio few/one = %.2f (2 waves share a SIMD: 2.0 if the lone wave issues back to back, ~1.0 if it cannot),"
           " full/one = %.2f (8.0 at equal clocks)\n", few / one, full / one);
    return 0;
}

// ubench_icache.hip -- what does a launch pay for instruction fetch when its code is not in the CUs' instruction caches?
// The small-batch key switch is a chain of four DIFFERENT fully unrolled kernels (30-50 KB of code each), one workgroup
// per CU: by the time a kernel comes round again the other three have passed through the 64 KB instruction cache.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_icache tools/ubench_icache.hip && tools/ubench_icache
// big<ID, STEPS>: straight-line code of STEPS dependent-free FP64 multiply-adds with distinct literals (cannot be rolled
// back into a loop); four instantiations = four code ranges.  Timed with events over a stream of launches:
//   same    A A A A ...          (code warm after the first launch)
//   rotate  A B C D A B C D ...  (each launch finds the cache filled by the others)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

template <int ID, int STEPS>
__global__ void big(double *p)
{
    double x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = p[threadIdx.x + 64 * j];
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        // eight independent chains: the body is issue-bound, not latency-bound, like a radix pass
        x[i & 7] = x[i & 7] * (1.0 + (double)(i * 4 + ID) * 1e-7) + (0.25 + (double)(i + ID * 7919) * 1e-6);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) p[(size_t)blockIdx.x * 512 + threadIdx.x + 64 * j] = x[j];
}

template <int STEPS>
static void run(hipStream_t st, double *buf, int grid, int threads, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float same = 0, rot = 0;
    for (int w = 0; w < 2; ++w) {
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) big<0, STEPS><<<grid, threads, 0, st>>>(buf);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&same, e0, e1));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; r += 4) {
            big<0, STEPS><<<grid, threads, 0, st>>>(buf);
            big<1, STEPS><<<grid, threads, 0, st>>>(buf);
            big<2, STEPS><<<grid, threads, 0, st>>>(buf);
            big<3, STEPS><<<grid, threads, 0, st>>>(buf);
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&rot, e0, e1));
    }
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(big<0, STEPS>)));
    std::printf("steps %5d, grid %3d x %3d threads: same kernel %.2f us per launch, four kernels in rotation %.2f us  (+%.2f)\n", STEPS, grid,
                threads, same * 1000.0 / reps, rot * 1000.0 / reps, (rot - same) * 1000.0 / reps);
    (void)fa;
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    double *buf;
    CK(hipMalloc(&buf, (size_t)1024 * 512 * 8));
    CK(hipMemset(buf, 0, (size_t)1024 * 512 * 8));
    for (int grid : {16, 64, 256}) {
        for (int threads : {64, 512}) {
            run<250>(st, buf, grid, threads, 400);
            run<1000>(st, buf, grid, threads, 400);
            run<2000>(st, buf, grid, threads, 400);
            run<4000>(st, buf, grid, threads, 400);
        }
    }
    return 0;
}

// ubench_gridbar.hip -- what does a device-wide barrier inside one persistent kernel cost on gfx950, against the
// launch boundary it would replace?  (the cooperative form of the small-batch key switch was
// rejected on an estimate; this is the measurement.)
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gridbar tools/ubench_gridbar.hip
//   tools/ubench_gridbar
//
// Three things are timed with HIP events on one stream:
//   (a) K dependent launches of a kernel that moves the same bytes per workgroup as (b) does per phase
//   (b) ONE launch of a persistent kernel that runs K phases separated by a barrier over all G workgroups
//       (one agent-scope release add + acquire spin by thread 0, __syncthreads around it), every phase reading
//       what the neighbouring workgroup wrote in the previous phase -- so visibility across XCDs is checked
//   (c) the same with the barrier over groups of `gs` workgroups only (an item's workgroups), counters 256 B apart
// Spins are bounded: a barrier that is not reached within 2^22 polls raises a flag and every workgroup leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

constexpr int WORDS = 1024;  // u64 words one workgroup writes per phase (8 KB)

__device__ __forceinline__ void phase_body(unsigned long long *buf, int G, int g, int phase, int *bad)
{
    // read the neighbour's words of the previous phase, write mine for this one
    const int nb = (g + 1) % G;
    unsigned long long *mine = buf + ((size_t)(phase & 1) * G + g) * WORDS;
    const unsigned long long *theirs = buf + ((size_t)((phase + 1) & 1) * G + nb) * WORDS;
    for (int i = threadIdx.x; i < WORDS; i += blockDim.x) {
        unsigned long long v = 0;
        if (phase > 0) {
            v = theirs[i];
            if (v != (unsigned long long)(phase - 1) * 1000003ull + (unsigned long long)nb * 4099ull + i) atomicAdd(bad, 1);
        }
        mine[i] = (unsigned long long)phase * 1000003ull + (unsigned long long)g * 4099ull + i;
    }
}

__global__ void step_kernel(unsigned long long *buf, int phase, int *bad) { phase_body(buf, gridDim.x, blockIdx.x, phase, bad); }

__device__ __forceinline__ bool barrier(unsigned *ctr, unsigned target, int *hung)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int polls = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++polls > (1 << 22)) {
                atomicExch(hung, 1);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return __hip_atomic_load(hung, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

// relaxed polling, one acquire fence at the end (cheaper when the invalidate is what costs)
__device__ __forceinline__ bool barrier_relaxed(unsigned *ctr, unsigned target, int *hung)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int polls = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++polls > (1 << 22)) {
                atomicExch(hung, 1);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);  // system scope by default in clang; see the agent form below
    }
    __syncthreads();
    return __hip_atomic_load(hung, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

__device__ __forceinline__ bool barrier_relaxed_agent(unsigned *ctr, unsigned target, int *hung)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int polls = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++polls > (1 << 22)) {
                atomicExch(hung, 1);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return __hip_atomic_load(hung, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

// no cache maintenance at all: the exchanged words themselves are written and read at agent scope (sc1 stores and
// loads, coherent across the XCDs' L2s), the counter is relaxed, and each thread waits for its own stores first
__device__ __forceinline__ bool barrier_bare(unsigned *ctr, unsigned target, int *hung)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int polls = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++polls > (1 << 22)) {
                atomicExch(hung, 1);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    return __hip_atomic_load(hung, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
}

template <int MODE>
__global__ void persistent_kernel(unsigned long long *buf, int K, int gs, unsigned *ctrs, int *bad, int *hung)
{
    const int G = gridDim.x, g = blockIdx.x;
    unsigned *ctr = ctrs + (size_t)(g / gs) * 64;  // 256 B apart
    const int members = (g / gs + 1) * gs <= G ? gs : G - (g / gs) * gs;
    for (int phase = 0; phase < K; ++phase) {
        // with group barriers the neighbour must be inside the group
        const int base = (g / gs) * gs;
        {
            const int nb = base + (g - base + 1) % members;
            unsigned long long *mine = buf + ((size_t)(phase & 1) * G + g) * WORDS;
            const unsigned long long *theirs = buf + ((size_t)((phase + 1) & 1) * G + nb) * WORDS;
            for (int i = threadIdx.x; i < WORDS; i += blockDim.x) {
                if (phase > 0) {
                    const unsigned long long v =
                        MODE == 3 ? __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : theirs[i];
                    if (v != (unsigned long long)(phase - 1) * 1000003ull + (unsigned long long)nb * 4099ull + i) atomicAdd(bad, 1);
                }
                const unsigned long long mv = (unsigned long long)phase * 1000003ull + (unsigned long long)g * 4099ull + i;
                if (MODE == 3) __hip_atomic_store(mine + i, mv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else mine[i] = mv;
            }
        }
        const unsigned target = (unsigned)(phase + 1) * (unsigned)members;
        bool ok;
        if (MODE == 0) ok = barrier(ctr, target, hung);
        else if (MODE == 1) ok = barrier_relaxed(ctr, target, hung);
        else if (MODE == 2) ok = barrier_relaxed_agent(ctr, target, hung);
        else ok = barrier_bare(ctr, target, hung);
        if (!ok) return;
    }
}

int main(int argc, char **argv)
{
    const int K = argc > 1 ? std::atoi(argv[1]) : 400;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    std::printf("device %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int Gs[] = {8, 16, 32, 64, 128, 256};
    const int threads = 512;
    unsigned long long *buf;
    unsigned *ctrs;
    int *flags;
    CK(hipMalloc(&buf, (size_t)2 * 256 * WORDS * 8));
    CK(hipMalloc(&ctrs, 256 * 64 * sizeof(unsigned)));
    CK(hipMalloc(&flags, 2 * sizeof(int)));
    for (int G : Gs) {
        if (G > prop.multiProcessorCount) continue;  // one workgroup per CU at most: all co-resident
        int h[2];
        // (a) dependent launches
        CK(hipMemsetAsync(flags, 0, 8, st));
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(e0, st));
            for (int p = 0; p < K; ++p) step_kernel<<<G, threads, 0, st>>>(buf, p, flags);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
        }
        float ms_a;
        CK(hipEventElapsedTime(&ms_a, e0, e1));
        CK(hipMemcpy(h, flags, 8, hipMemcpyDeviceToHost));
        std::printf("G=%3d  launches: %.2f us per phase (bad %d)\n", G, ms_a * 1000.0 / K, h[0]);
        // (b)/(c)
        for (int mode = 0; mode < 4; ++mode) {
            for (int gs : {G, 8, 4}) {
                if (gs > G) continue;
                if (gs != G && G < 16) continue;
                float ms = 0;
                for (int w = 0; w < 2; ++w) {
                    CK(hipMemsetAsync(flags, 0, 8, st));
                    CK(hipMemsetAsync(ctrs, 0, 256 * 64 * sizeof(unsigned), st));
                    CK(hipEventRecord(e0, st));
                    if (mode == 0) persistent_kernel<0><<<G, threads, 0, st>>>(buf, K, gs, ctrs, flags, flags + 1);
                    else if (mode == 1) persistent_kernel<1><<<G, threads, 0, st>>>(buf, K, gs, ctrs, flags, flags + 1);
                    else if (mode == 2) persistent_kernel<2><<<G, threads, 0, st>>>(buf, K, gs, ctrs, flags, flags + 1);
                    else persistent_kernel<3><<<G, threads, 0, st>>>(buf, K, gs, ctrs, flags, flags + 1);
                    CK(hipEventRecord(e1, st));
                    CK(hipStreamSynchronize(st));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                }
                CK(hipMemcpy(h, flags, 8, hipMemcpyDeviceToHost));
                std::printf("G=%3d  persistent mode %d, barrier over %3d workgroups: %.2f us per phase (bad %d, hung %d)\n", G, mode,
                            gs, ms * 1000.0 / K, h[0], h[1]);
                if (h[1]) {
                    std::printf("barrier not reached: stopping\n");
                    return 2;
                }
            }
        }
    }
    return 0;
}

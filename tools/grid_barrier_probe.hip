// grid_barrier_probe.hip -- diagnostic: what a device-wide barrier inside ONE persistent kernel costs on gfx950
// (256 co-resident workgroups, one per CU), to compare with the ~1.45 us gap between two dependent launches.
// Every workgroup's wavefront 0 arrives with one atomic and polls; the other wavefronts wait at a workgroup barrier.
// A poll budget bounds every spin loop: the kernel always terminates.
//   hipcc -O3 --offload-arch=gfx950 tools/grid_barrier_probe.hip -o tools/grid_barrier_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                         \
    do                                                                   \
    {                                                                    \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess)                                            \
        {                                                                \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                                \
        }                                                                \
    } while (0)

// counters[c * 32]: one counter per 128-byte line; `groups` counters are used (workgroup b arrives at b % groups)
template <int GROUPS>
__global__ void __launch_bounds__(512) probe(unsigned* counters, int iters, int busy_ticks, unsigned long long* out, unsigned* failed, int sleepy)
{
    const unsigned nb = gridDim.x;
    const unsigned per_group = nb / GROUPS;  // (nb is a multiple of GROUPS)
    __shared__ int go;
    unsigned long long t_begin = 0;
    for (int it = 0; it < iters; ++it)
    {
        if (it == 8 && threadIdx.x == 0) t_begin = __builtin_amdgcn_s_memrealtime();
        // "work": idle for busy_ticks of the 100 MHz clock
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < busy_ticks) __builtin_amdgcn_s_sleep(2);
        __syncthreads();
        if (threadIdx.x == 0)
        {
            __hip_atomic_fetch_add(&counters[(blockIdx.x % GROUPS) * 32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it + 1) * per_group;
            int budget = 1 << 20;
            bool done = false;
            while (!done && --budget > 0)
            {
                done = true;
#pragma unroll
                for (int g = 0; g < GROUPS; ++g)  // relaxed polls (no cache invalidation per poll), one acquire at the end
                    if (__hip_atomic_load(&counters[g * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) done = false;
                if (!done && sleepy) __builtin_amdgcn_s_sleep(4);
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            if (budget <= 0) atomicAdd(failed, 1u);
            go = it;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = __builtin_amdgcn_s_memrealtime() - t_begin;
}

int main()
{
    unsigned* counters;
    unsigned long long* out;
    unsigned* failed;
    CHECK(hipMalloc(&counters, 8 * 32 * sizeof(unsigned)));
    CHECK(hipMalloc(&out, 8));
    CHECK(hipMalloc(&failed, 4));
    const int iters = 2008;
    for (int sleepy : {0, 1})
    for (int threads : {512})
        for (int busy : {0, 300})
            for (int groups : {1, 8})
            {
                CHECK(hipMemset(counters, 0, 8 * 32 * sizeof(unsigned)));
                CHECK(hipMemset(failed, 0, 4));
                if (groups == 1)
                    hipLaunchKernelGGL(probe<1>, dim3(256), dim3(threads), 0, 0, counters, iters, busy, out, failed, sleepy);
                else
                    hipLaunchKernelGGL(probe<8>, dim3(256), dim3(threads), 0, 0, counters, iters, busy, out, failed, sleepy);
                CHECK(hipDeviceSynchronize());
                unsigned long long ticks = 0;
                unsigned f = 0;
                CHECK(hipMemcpy(&ticks, out, 8, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));
                std::printf("%s %3d threads per workgroup, %.1f us of work, %d arrival counter(s): %.2f us per iteration (barrier ~%.2f us)%s\n", sleepy ? "sleep" : "spin ", threads,
                            busy * 0.01, groups, ticks * 0.01 / (iters - 8), ticks * 0.01 / (iters - 8) - busy * 0.01, f ? "  POLL BUDGET EXHAUSTED" : "");
            }
    return 0;
}

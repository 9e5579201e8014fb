// hostwrite_probe.hip -- diagnostic: does a trickle of stores into pinned host memory (a few tens of KB per launch,
// spread over all workgroups) lengthen a chain of dependent ~5 us launches?  Build:
//   hipcc -O3 --offload-arch=gfx950 tools/hostwrite_probe.hip -o tools/hostwrite_probe.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
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

typedef double v2d __attribute__((ext_vector_type(2)));

// 256 workgroups x 512 threads; every workgroup reads/writes a little device memory and idles until ~`busy_ticks`
// of the 100 MHz clock have passed; its last wavefront copies `per_block` bytes from `src` (device) to `dst` (host)
__global__ void __launch_bounds__(512) probe(const v2d* in, v2d* out, const v2d* src, v2d* dst, int per_block_pieces, int busy_ticks, int launch, int through)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    v2d v = in[t];
    if ((threadIdx.x >> 6) == 7)
    {
        const int lane = threadIdx.x & 63;
        for (int k = lane; k < per_block_pieces; k += 64)
        {
            const size_t idx = ((size_t)launch * gridDim.x + blockIdx.x) * (size_t)per_block_pieces + k;
            const v2d x = src[idx];
            if (through)
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + idx), "v"(x) : "memory");
            else
                dst[idx] = x;
        }
    }
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < busy_ticks) __builtin_amdgcn_s_sleep(8);
    out[t] = v + 1.0;
}

int main()
{
    const int blocks = 256, threads = 512, launches = 400;
    v2d *a, *b, *src, *dst;
    const size_t total_pieces = (size_t)launches * blocks * 64 * 4;  // up to 4 KB per block per launch
    CHECK(hipMalloc(&a, (size_t)blocks * threads * 16));
    CHECK(hipMalloc(&b, (size_t)blocks * threads * 16));
    CHECK(hipMalloc(&src, total_pieces * 16));
    CHECK(hipHostMalloc(&dst, total_pieces * 16, hipHostMallocDefault));
    CHECK(hipMemset(a, 0, (size_t)blocks * threads * 16));
    CHECK(hipMemset(b, 0, (size_t)blocks * threads * 16));
    CHECK(hipMemset(src, 1, total_pieces * 16));
    std::memset(dst, 0, total_pieces * 16);
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int through : {0, 1})
    for (int busy : {450})
        for (int pieces : {0, 4, 16, 64, 256})
        {
            hipGraph_t g;
            CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            for (int l = 0; l < launches; ++l)
                hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, st, (l & 1) ? b : a, (l & 1) ? a : b, src, dst, pieces, busy, l, through);
            CHECK(hipStreamEndCapture(st, &g));
            hipGraphExec_t ex;
            CHECK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
            CHECK(hipGraphLaunch(ex, st));
            CHECK(hipStreamSynchronize(st));
            CHECK(hipEventRecord(e0, st));
            for (int r = 0; r < 4; ++r) CHECK(hipGraphLaunch(ex, st));
            CHECK(hipEventRecord(e1, st));
            CHECK(hipStreamSynchronize(st));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            // verify the host copy of the last replay
            size_t bad = 0;
            const unsigned char* d8 = reinterpret_cast<const unsigned char*>(dst);
            for (size_t i = 0; i < (size_t)launches * blocks * pieces * 16; ++i) bad += d8[i] != 1;
            std::printf("%s busy %.1f us, %5d B per workgroup (%7.1f KB per launch) to pinned host: %.2f us per launch%s\n", through ? "write-through" : "plain        ", busy * 0.01, pieces * 16,
                        pieces * 16.0 * blocks / 1024, ms * 1e3 / (4.0 * launches), bad ? "  HOST COPY WRONG" : "");
            std::memset(dst, 0, total_pieces * 16);
            CHECK(hipGraphExecDestroy(ex));
            CHECK(hipGraphDestroy(g));
        }
    return 0;
}

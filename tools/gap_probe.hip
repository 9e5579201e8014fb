// gap_probe.hip -- diagnostic: what a dependent kernel launch costs on gfx950 as a function of how many bytes the
// previous kernel wrote (and how).  Replays a hipGraph of N identical dependent launches and prints the average
// time per launch.  Build: hipcc -O3 --offload-arch=gfx950 tools/gap_probe.hip -o gpurun_out/gap_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
    do                                                                            \
    {                                                                             \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess)                                                     \
        {                                                                         \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));          \
            std::exit(1);                                                         \
        }                                                                         \
    } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

// `blocks` workgroups of 256 threads stream-write `pieces` 16-byte pieces in total (coalesced, grid-stride), after
// reading the same amount of the other buffer when `rd`; every workgroup records its start and end on the 100 MHz clock
template <int KIND>
__global__ void __launch_bounds__(256) probe(const v2d* in, v2d* out, int pieces, int rd, unsigned long long* stamps)
{
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int nt = gridDim.x * blockDim.x;
    v2d acc = {1.0, 2.0};
    if (rd)
        for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < pieces; t += nt) acc += in[t];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < pieces; t += nt)
    {
        if (KIND == 0)
            out[t] = acc;
        else if (KIND == 1)
            __builtin_nontemporal_store(acc, out + t);
        else
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(out + t), "v"(acc) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0)
    {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

int main()
{
    const int threads = 1 << 19;  // 8 MB of 16-byte pieces
    v2d *a, *b;
    CHECK(hipMalloc(&a, (size_t)threads * 16));
    CHECK(hipMalloc(&b, (size_t)threads * 16));
    CHECK(hipMemset(a, 0, (size_t)threads * 16));
    CHECK(hipMemset(b, 0, (size_t)threads * 16));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int launches = 512;
    const int blocks = 1024;
    unsigned long long* d_stamps;
    CHECK(hipMalloc(&d_stamps, 2 * blocks * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(2 * blocks);
    const char* names[3] = {"plain", "nontemporal", "sc0 sc1"};
    for (int rd = 0; rd < 2; ++rd)
        for (int kind = 0; kind < 3; ++kind)
            for (int wkb : {0, 256, 512, 1024, 2048, 4096, 8192})
            {
                const int pieces = wkb * 1024 / 16;
                hipGraph_t g;
                CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
                for (int l = 0; l < launches; ++l)
                {
                    v2d* in = (l & 1) ? b : a;
                    v2d* out = (l & 1) ? a : b;
                    if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, st, in, out, pieces, rd, d_stamps);
                    if (kind == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, st, in, out, pieces, rd, d_stamps);
                    if (kind == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, st, in, out, pieces, rd, d_stamps);
                }
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
                CHECK(hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost));
                unsigned long long lo = ~0ULL, hi = 0;
                for (int k = 0; k < blocks; ++k)
                {
                    if (h[2 * k] < lo) lo = h[2 * k];
                    if (h[2 * k + 1] > hi) hi = h[2 * k + 1];
                }
                const double per = ms * 1e3 / (4.0 * launches), span = (hi - lo) * 0.01;
                std::printf("%-12s %s writes %4d KB: %.2f us per launch, kernel span %.2f us, gap %.2f us\n", names[kind], rd ? "reads+" : "      ", wkb, per,
                            span, per - span);
                CHECK(hipGraphExecDestroy(ex));
                CHECK(hipGraphDestroy(g));
            }
    return 0;
}

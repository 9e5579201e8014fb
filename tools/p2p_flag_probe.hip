// p2p_flag_probe.hip -- diagnostic: what ONE ensemble step costs in memory hand-offs if the step launches are replaced by
// a persistent kernel whose workgroups synchronise point to point (experiment 3a): a workgroup publishes the
// rows it updated with write-through stores and an epoch flag, and a consumer polls only the flags of the workgroups
// that own the walkers it needs -- no device-wide barrier.
//
// Geometry of the C2 full-step launch: 256 workgroups (one per CU) x 4 updating wavefronts x (8 red + 8 black) walkers,
// rows of 256 bytes.  Per step a wavefront needs 24 rows owned by others (8 partners of its red walkers, 8 red partners
// of its black walkers and those partners' own 8 partners): random walkers, i.e. about 22 distinct workgroups of 256.
// One iteration here = publish (64 rows per workgroup, 16-byte `sc1` stores, drained, then one `sc1` epoch flag per
// workgroup) -> every wavefront polls the epoch flags of the owners of its 24 random rows (one lane per flag, relaxed
// `sc1` loads) -> gathers those rows with `sc1` loads.  No arithmetic: this is the floor the hand-offs alone set, to
// be compared with the 1.45 us dependent-launch boundary plus the two cold round trips of the launch-per-step scheme
// (about 4.2 us per step in all without the calculator).  Position buffers rotate over four copies so that a fast
// workgroup never overwrites rows a slow one still reads (a skew beyond two steps is counted, and would need a check
// on the real thing).  Every spin loop has a poll budget: the kernel always terminates.
//   hipcc -O3 --offload-arch=gfx950 tools/p2p_flag_probe.hip -o tools/p2p_flag_probe.bin
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

constexpr int kBlocks = 256, kWaves = 4, kRowsPerWave = 16, kRowDoubles = 32, kNeed = 24, kCopies = 4;
constexpr int kWalkers = kBlocks * kWaves * kRowsPerWave;  // 16 384

__device__ __forceinline__ void store16_sc1(double* p, double a, double b)
{
    typedef double v2dd __attribute__((ext_vector_type(2)));
    const v2dd v = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
// six 16-byte `sc1` loads in flight, ONE statement with its own wait: an asm load without a wait hands the compiler
// registers the hardware has not written yet (it may reuse them, e.g. for an address, before the data lands)
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void load6x16_sc1(const double* const (&p)[6], v2d (&v)[6])
{
    asm volatile(
        "global_load_dwordx4 %0, %6, off sc0 sc1\n\t"
        "global_load_dwordx4 %1, %7, off sc0 sc1\n\t"
        "global_load_dwordx4 %2, %8, off sc0 sc1\n\t"
        "global_load_dwordx4 %3, %9, off sc0 sc1\n\t"
        "global_load_dwordx4 %4, %10, off sc0 sc1\n\t"
        "global_load_dwordx4 %5, %11, off sc0 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5])
        : "memory");
}
__device__ __forceinline__ unsigned load_flag_sc1(const unsigned* p)
{
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned hash32(unsigned x)
{
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

// pos: [kCopies][kWalkers][kRowDoubles]; flags: [kBlocks] words on lines of their own (32 words apart)
// mode 0: flags + gathers as described; mode 1: publish + flags only (no row gather); mode 2: no publish payload (flags only)
__global__ void __launch_bounds__(64 * kWaves) probe(double* pos, unsigned* flags, int iters, int mode, unsigned long long* out, unsigned* failed,
                                                      unsigned* skewed, double* sink)
{
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int wave = blockIdx.x * kWaves + wib;
    const int sub = lane & 15, grp = lane >> 4;  // 16 lanes x 16 bytes = one 256-byte row; 4 rows per wave instruction
    unsigned long long t_begin = 0;
    double acc = 0.0;
    for (int it = 1; it <= iters; ++it)
    {
        if (it == 9 && threadIdx.x == 0) t_begin = __builtin_amdgcn_s_memrealtime();
        double* out_buf = pos + (size_t)(it % kCopies) * kWalkers * kRowDoubles;
        const double* in_buf = pos + (size_t)((it + kCopies - 1) % kCopies) * kWalkers * kRowDoubles;
        // ---- wait for the owners of the 24 rows this wavefront needs from step it-1, then gather them ----
        if (it > 1)
        {
            const unsigned want = (unsigned)(it - 1);
            const unsigned r = hash32((unsigned)wave * 131u + (unsigned)(lane % kNeed) * 7919u + (unsigned)it * 2654435761u) % (unsigned)kWalkers;
            const unsigned owner = r / (kWaves * kRowsPerWave);
            bool gave_up = false;
            if (lane < kNeed)
            {
                int budget = 1 << 16;
                unsigned seen = load_flag_sc1(flags + owner * 32);
                while (seen < want && --budget > 0) seen = load_flag_sc1(flags + owner * 32);
                if (budget <= 0) atomicAdd(failed, 1u);
                if (seen > want + 1) atomicAdd(skewed, 1u);  // the owner is more than two steps ahead: its rows of it-1 are about to go
                gave_up = budget <= 0;
            }
            if (__any(gave_up)) break;  // (then everybody who waits for this workgroup gives up too: the kernel drains)
            __builtin_amdgcn_wave_barrier();
            if (mode == 0)
            {
                // 24 rows, 4 per wave instruction: the row index of need k sits in lane k
                const double* ptr[6];
                v2d got[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) ptr[k] = in_buf + (size_t)__shfl(r, 4 * k + grp) * kRowDoubles + 2 * sub;
                load6x16_sc1(ptr, got);
#pragma unroll
                for (int k = 0; k < 6; ++k) acc += got[k].x + got[k].y;
            }
        }
        // ---- publish this wavefront's 16 rows of step it (write-through), drain, then the workgroup's epoch flag ----
        if (mode != 2)
        {
#pragma unroll
            for (int k = 0; k < kRowsPerWave; k += 4)
                store16_sc1(out_buf + (size_t)(wave * kRowsPerWave + k + grp) * kRowDoubles + 2 * sub, acc + it, (double)lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every storing wavefront has drained
        if (threadIdx.x == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(flags + blockIdx.x * 32), "v"((unsigned)it) : "memory");
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = __builtin_amdgcn_s_memrealtime() - t_begin;
    if (acc == 12345.678) sink[0] = acc;
}

int main()
{
    double* pos;
    unsigned *flags, *failed, *skewed;
    unsigned long long* out;
    double* sink;
    CHECK(hipMalloc(&pos, sizeof(double) * (size_t)kCopies * kWalkers * kRowDoubles));
    CHECK(hipMalloc(&flags, kBlocks * 32 * sizeof(unsigned)));
    CHECK(hipMalloc(&out, 8));
    CHECK(hipMalloc(&failed, 4));
    CHECK(hipMalloc(&skewed, 4));
    CHECK(hipMalloc(&sink, 8));
    const int iters = 2008;
    const char* what[3] = {"publish 16 KB per workgroup + epoch flags + gather of 24 rows per wavefront", "publish + epoch flags, no gather",
                           "epoch flags only (no payload)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode)
        {
            CHECK(hipMemset(pos, 0, sizeof(double) * (size_t)kCopies * kWalkers * kRowDoubles));
            CHECK(hipMemset(flags, 0, kBlocks * 32 * sizeof(unsigned)));
            CHECK(hipMemset(failed, 0, 4));
            CHECK(hipMemset(skewed, 0, 4));
            hipLaunchKernelGGL(probe, dim3(kBlocks), dim3(64 * kWaves), 0, 0, pos, flags, iters, mode, out, failed, skewed, sink);
            CHECK(hipDeviceSynchronize());
            unsigned long long ticks = 0;
            unsigned f = 0, sk = 0;
            CHECK(hipMemcpy(&ticks, out, 8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(&sk, skewed, 4, hipMemcpyDeviceToHost));
            std::printf("%-80s %.2f us per step (workgroup 0, %d steps)%s; polls that found the owner > 2 steps ahead: %u\n", what[mode],
                        ticks * 0.01 / (iters - 8), iters - 8, f ? "  POLL BUDGET EXHAUSTED" : "", sk);
        }
    return 0;
}

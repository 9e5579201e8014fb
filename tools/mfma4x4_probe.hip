// Probe (diagnostic, not part of the library): v_mfma_f64_4x4x4_4b_f64 on gfx950 -- operand/result layout found with
// one-hot operands, rounding order against the ascending-k fma chain, and issue cost of a dependent chain beside
// v_mfma_f64_16x16x4_f64.  Question behind it: could the half-empty black tile of the full-step kernel
// (8 rows of 16) go through 4x4 blocks instead?
//   hipcc --offload-arch=gfx950 -O2 tools/mfma4x4_probe.hip -o tools/mfma4x4_probe.bin && tools/mfma4x4_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

// out[p * 64 + q] = mask of result lanes that are non-zero when A is one-hot in lane p and B one-hot in lane q
__global__ void onehot(unsigned long long* out)
{
    const int l = threadIdx.x;
    for (int p = 0; p < 64; ++p)
        for (int q = 0; q < 64; ++q)
        {
            const double a = l == p ? 1.0 : 0.0, b = l == q ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (l == 0) out[p * 64 + q] = m;
        }
}

__global__ void apply(const double* A, const double* B, const double* C, double* D)
{
    const int l = threadIdx.x;
    D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], C[l], 0, 0, 0);
}

// clocks per instruction of a chain of n dependent MFMAs (one wavefront on an otherwise idle chip)
__global__ void chain_4x4(double* sink, long long* clocks, int n)
{
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3, d = 0.0;
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d, 0, 0, 0);
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    sink[threadIdx.x] = d;
    if (threadIdx.x == 0)
    {
        clocks[0] = t1 - t0;
        clocks[1] = w1 - w0;
    }
}
__global__ void chain_4x4_two(double* sink, long long* clocks, int n)  // two independent chains interleaved
{
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3, d = 0.0, e = 0.0;
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i)
    {
        d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d, 0, 0, 0);
        e = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, e, 0, 0, 0);
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    sink[threadIdx.x] = d + e;
    if (threadIdx.x == 0)
    {
        clocks[0] = t1 - t0;
        clocks[1] = w1 - w0;
    }
}
__global__ void chain_16x16(double* sink, long long* clocks, int n)
{
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    double4_t d = {0, 0, 0, 0};
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    sink[threadIdx.x] = d[0] + d[1] + d[2] + d[3];
    if (threadIdx.x == 0)
    {
        clocks[0] = t1 - t0;
        clocks[1] = w1 - w0;
    }
}
__global__ void chain_16x16_two(double* sink, long long* clocks, int n)
{
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    double4_t d = {0, 0, 0, 0}, e = {0, 0, 0, 0};
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i)
    {
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
        e = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, e, 0, 0, 0);
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    sink[threadIdx.x] = d[0] + e[0] + d[1] + e[1] + d[2] + e[2] + d[3] + e[3];
    if (threadIdx.x == 0)
    {
        clocks[0] = t1 - t0;
        clocks[1] = w1 - w0;
    }
}

#define CK(x)                                                                      \
    do                                                                             \
    {                                                                              \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess)                                                      \
        {                                                                          \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                              \
        }                                                                          \
    } while (0)

int main()
{
    // ---- layout ----
    unsigned long long* d_mask;
    CK(hipMalloc(&d_mask, 4096 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(onehot, dim3(1), dim3(64), 0, 0, d_mask);
    std::vector<unsigned long long> mask(4096);
    CK(hipMemcpy(mask.data(), d_mask, 4096 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    // A lane p meets B lane q where they share block and k; the result lanes hit are (block, i of p, j of q)
    int partners_min = 64, partners_max = 0, outs_min = 64, outs_max = 0;
    for (int p = 0; p < 64; ++p)
    {
        int partners = 0;
        for (int q = 0; q < 64; ++q)
            if (mask[p * 64 + q])
            {
                ++partners;
                const int outs = __builtin_popcountll(mask[p * 64 + q]);
                outs_min = outs < outs_min ? outs : outs_min;
                outs_max = outs > outs_max ? outs : outs_max;
            }
        partners_min = partners < partners_min ? partners : partners_min;
        partners_max = partners > partners_max ? partners : partners_max;
    }
    printf("one-hot: every A lane meets %d..%d B lanes, each meeting lights %d..%d result lane(s)\n", partners_min, partners_max, outs_min, outs_max);
    printf("A lane p -> B lanes q it meets (result lane):\n");
    for (int p = 0; p < 64; p += 1)
    {
        if (p % 16 >= 6 && p % 16 != 15) continue;  // (a readable excerpt: lanes 0..5 and 15 of every sixteen)
        printf("  A %2d:", p);
        for (int q = 0; q < 64; ++q)
            if (mask[p * 64 + q]) printf(" B%d(->%d)", q, __builtin_ctzll(mask[p * 64 + q]));
        printf("\n");
    }
    // derive (block, i, k) of A lanes and (block, k, j) of B lanes under the hypothesis lane = 16*block + 4*k + i (A), 16*block + 4*k + j (B), result 16*block + 4*j?+i?
    int hyp_ok[2] = {1, 1};  // result lane = 16 b + 4 i + j   |   16 b + 4 j + i   ... with A lane = 16 b + 4 k + i?  try both A/B index orders too
    const char* names[4] = {"A: 16b+4k+i, B: 16b+4k+j", "A: 16b+4i+k, B: 16b+4j+k", "A: 16b+4k+i, B: 16b+4j+k", "A: 16b+4i+k, B: 16b+4k+j"};
    for (int h = 0; h < 4; ++h)
        for (int ro = 0; ro < 2; ++ro)
        {
            bool ok = true;
            for (int p = 0; p < 64 && ok; ++p)
                for (int q = 0; q < 64 && ok; ++q)
                {
                    const int bp = p / 16, bq = q / 16;
                    const int pi = (h == 0 || h == 2) ? p % 4 : (p % 16) / 4, pk = (h == 0 || h == 2) ? (p % 16) / 4 : p % 4;
                    const int qj = (h == 0 || h == 3) ? q % 4 : (q % 16) / 4, qk = (h == 0 || h == 3) ? (q % 16) / 4 : q % 4;
                    unsigned long long want = 0;
                    if (bp == bq && pk == qk) want = 1ULL << (16 * bp + (ro == 0 ? 4 * pi + qj : 4 * qj + pi));
                    ok = mask[p * 64 + q] == want;
                }
            if (ok) printf("layout: %s, result lane = 16b + %s\n", names[h], ro == 0 ? "4i + j" : "4j + i");
            (void)hyp_ok;
        }

    // ---- rounding order: ascending-k fma chain from C? (uses the layout printed above only through the one-hot table) ----
    std::vector<double> A(64), B(64), C(64), D(64);
    srand(11);
    for (int l = 0; l < 64; ++l)
    {
        A[l] = (rand() / (double)RAND_MAX - 0.5) * 3.7;
        B[l] = (rand() / (double)RAND_MAX - 0.5) * 2.9;
        C[l] = (rand() / (double)RAND_MAX - 0.5) * 5.1;
    }
    double *dA, *dB, *dC, *dD;
    CK(hipMalloc(&dA, 512));
    CK(hipMalloc(&dB, 512));
    CK(hipMalloc(&dC, 512));
    CK(hipMalloc(&dD, 512));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, C.data(), 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(apply, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    CK(hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost));
    // for every result lane collect its four (A lane, B lane) pairs from the one-hot table, ordered by A lane
    int asc = 0, desc = 0, neither = 0;
    for (int r = 0; r < 64; ++r)
    {
        int pa[4], qb[4], n = 0;
        for (int p = 0; p < 64; ++p)
            for (int q = 0; q < 64; ++q)
                if (mask[p * 64 + q] & (1ULL << r))
                {
                    if (n < 4)
                    {
                        pa[n] = p;
                        qb[n] = q;
                    }
                    ++n;
                }
        if (n != 4)
        {
            printf("result lane %d has %d contributions\n", r, n);
            continue;
        }
        double up = C[r], down = C[r];
        for (int k = 0; k < 4; ++k) up = fma(A[pa[k]], B[qb[k]], up);
        for (int k = 3; k >= 0; --k) down = fma(A[pa[k]], B[qb[k]], down);
        if (std::memcmp(&up, &D[r], 8) == 0)
            ++asc;
        else if (std::memcmp(&down, &D[r], 8) == 0)
            ++desc;
        else
            ++neither;
    }
    printf("rounding: %d result lanes equal the fma chain in ascending A-lane order from C, %d the descending chain, %d neither\n", asc, desc, neither);

    // ---- issue cost ----
    double* sink;
    long long* dclk;
    CK(hipMalloc(&sink, 512));
    CK(hipMalloc(&dclk, 16));
    const int n = 4096;
    long long c[2];
#define TIME(K, label, per)                                                                              \
    hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, sink, dclk, n);                                       \
    hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, sink, dclk, n);                                       \
    CK(hipMemcpy(c, dclk, 16, hipMemcpyDeviceToHost));                                                   \
    printf("%-44s %.1f clock64 ticks, %.1f ns per instruction\n", label, (double)c[0] / (n * per), (double)c[1] * 10.0 / (n * per));
    TIME(chain_4x4, "4x4x4_4b, one dependent chain:", 1);
    TIME(chain_4x4_two, "4x4x4_4b, two chains interleaved:", 2);
    TIME(chain_16x16, "16x16x4, one dependent chain:", 1);
    TIME(chain_16x16_two, "16x16x4, two chains interleaved:", 2);
    printf("(clock64 = s_memtime; ns from wall_clock64 = s_memrealtime at 100 MHz)\n");
    return 0;
}

// Probe (diagnostic, not part of the library): operand/result layout and rounding behaviour of
// v_mfma_f32_16x16x4_f32 (and v_mfma_f32_32x32x2_f32 is not probed) on gfx950.  Checks whether D = A*B + C equals, bit for bit, the chain
// fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,c)))) in ascending k (what a host Calculator's loop computes).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_f32_probe.hip -o tools/mfma_f32_probe.bin && tools/mfma_f32_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float real4_t __attribute__((ext_vector_type(4)));

__global__ void probe(const float* A /*16x4 row-major [m][k]*/, const float* B /*4x16 [k][n]*/, const float* C /*16x16 [m][n]*/,
                      float* D /*[lane][4]*/)
{
    const int l = threadIdx.x;
    // assumed layout: A: lane l holds A[m = l%16][k = l/16]; B: lane l holds B[k = l/16][n = l%16]
    // C/D: lane l register r holds [m = 4*r + l/16][n = l%16]
    const float a = A[(l % 16) * 4 + l / 16];
    const float b = B[(l / 16) * 16 + l % 16];
    real4_t c;
    // (the fp32 instruction's C/D layout: lane l register r holds [m = 4*(l/16) + r][n = l%16] -- not the fp64 one's [4r + l/16])
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * (l / 16) + r) * 16 + l % 16];
    real4_t d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = d[r];
}

int main()
{
    std::vector<float> A(64), B(64), C(256), D(256);
    srand(7);
    auto rnd = []() { return (rand() / (float)RAND_MAX - 0.5) * 4.0; };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    for (auto& v : C) v = rnd() * 8;
    float *dA, *dB, *dC, *dD;
    (void)hipMalloc(&dA, 64 * 4); (void)hipMalloc(&dB, 64 * 4); (void)hipMalloc(&dC, 256 * 4); (void)hipMalloc(&dD, 256 * 4);
    (void)hipMemcpy(dA, A.data(), 64 * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), 64 * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, C.data(), 256 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    (void)hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
    int asc = 0, desc = 0, close = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r)
        {
            const int m = 4 * (l / 16) + r, n = l % 16;
            float up = C[m * 16 + n], down = C[m * 16 + n], exact = C[m * 16 + n];
            for (int k = 0; k < 4; ++k) up = std::fmaf(A[m * 4 + k], B[k * 16 + n], up);
            for (int k = 3; k >= 0; --k) down = std::fmaf(A[m * 4 + k], B[k * 16 + n], down);
            for (int k = 0; k < 4; ++k) exact += A[m * 4 + k] * B[k * 16 + n];
            const float got = D[l * 4 + r];
            asc += got == up;
            desc += got == down;
            close += std::fabs(got - exact) <= 1e-5f * (1 + std::fabs(exact));
        }
    printf("of 256 outputs: equal to ascending-k fma chain %d, descending %d, numerically right (layout ok) %d\n", asc, desc, close);
    return 0;
}

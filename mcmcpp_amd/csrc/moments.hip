// moments.hip -- first and second moments of stored chain steps on the GPU: the device side of
// Analysis::CovarianceMatrix (reference: MCMCpp/Analysis/CovarianceMatrix.h:154-257).
//
// The reference walks the chain sample by sample and Kahan-sums x_i and x_i*x_j (D*(D+1)/2 products per sample).
// That is a rank-N update S = X^T X of a D x D matrix -- GEMM-shaped work, so it runs on the matrix cores:
//   * stored steps ([n][W][D], host memory) are uploaded in chunks; a wavefront takes samples four at a time, every
//     lane holding element (sample l/16, parameter 16 t + l%16) of the 4 x D slab -- which is at once the A operand
//     (16 parameters x 4 samples) and the B operand (4 samples x 16 parameters) of v_mfma_f64_16x16x4_f64
//     (layouts measured in tools/mfma_probe.hip), so one 16-byte-coalesced load feeds both sides;
//   * the lower triangle of 16 x 16 tiles is accumulated in registers, fp64 throughout (fp32 samples are widened on
//     load: their products are exact in fp64);
//   * every wavefront adds its tiles into its own slot of a partial buffer (no atomics), and the slots are summed in a
//     fixed order at the end: results do not depend on scheduling.
// Parity bar (tests/test_moments.py): the reference's order of operations cannot be kept by a parallel sum, so
// covariance and correlation agree with the oracle's restatement of the reference within a stated tolerance
// (fp64: 1e-10 of sqrt(var_i var_j); the device sum is the more accurate of the two for fp32 chains).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/mcmcpp_hip.h"

namespace
{
typedef double mfma_f64x4 __attribute__((ext_vector_type(4)));

constexpr int kMomentWavesPerBlock = 4;
constexpr int kMaxTiles = 4;  // matrix-core path: D <= 64

// partial layout per wavefront slot: [Dp][Dp] products (lower triangle of tiles filled) followed by [Dp] sums; Dp = 16 T
template <class T, int TILES>
__global__ void __launch_bounds__(64 * kMomentWavesPerBlock)
moments_mfma_kernel(const T* samples, long long n_samples, int dims, double* partial)
{
    constexpr int DP = 16 * TILES;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kMomentWavesPerBlock + (threadIdx.x >> 6);
    const int waves = gridDim.x * kMomentWavesPerBlock;
    const int krow = lane >> 4, col = lane & 15;
    // samples in groups of four; the groups are dealt round-robin to the wavefronts
    const long long groups = (n_samples + 3) / 4;
    mfma_f64x4 acc[TILES][TILES];
#pragma unroll
    for (int a = 0; a < TILES; ++a)
#pragma unroll
        for (int b = 0; b < TILES; ++b) acc[a][b] = mfma_f64x4{0, 0, 0, 0};
    double sum[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) sum[t] = 0.0;
    // four groups per round: their loads are in flight together (a wavefront's chain of loads is what bounds this kernel)
    constexpr int U = 4;
    for (long long g0 = wave; g0 < groups; g0 += (long long)waves * U)
    {
        double x[U][TILES];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const long long s = 4 * (g0 + (long long)u * waves) + krow;  // (beyond the last group: s >= n_samples, zeros)
#pragma unroll
            for (int t = 0; t < TILES; ++t)
            {
                const int p = 16 * t + col;
                x[u][t] = (s < n_samples && p < dims) ? (double)samples[(size_t)s * dims + p] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
#pragma unroll
            for (int t = 0; t < TILES; ++t) sum[t] += x[u][t];
#pragma unroll
            for (int a = 0; a < TILES; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][a], x[u][b], acc[a][b], 0, 0, 0);
        }
    }
    double* slot = partial + (size_t)wave * (size_t)(DP * DP + DP);
    // C/D layout: lane l, register r -> [m = 4r + l/16][n = l%16]
#pragma unroll
    for (int a = 0; a < TILES; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) slot[(size_t)(16 * a + 4 * r + krow) * DP + 16 * b + col] += acc[a][b][r];
    // parameter sums: the four lane groups hold disjoint samples
#pragma unroll
    for (int t = 0; t < TILES; ++t)
    {
        double v = sum[t];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (krow == 0) slot[(size_t)DP * DP + 16 * t + col] += v;
    }
}

// any D: one workgroup per slot, samples staged through LDS four at a time, every thread owns pairs (i, j <= i)
template <class T>
__global__ void __launch_bounds__(256) moments_generic_kernel(const T* samples, long long n_samples, int dims, int dp, double* partial)
{
    extern __shared__ double tile[];  // [4][dims]
    double* slot = partial + (size_t)blockIdx.x * ((size_t)dp * dp + dp);
    const long long per = (n_samples + gridDim.x - 1) / gridDim.x;
    const long long lo = per * blockIdx.x, hi = (lo + per < n_samples) ? lo + per : n_samples;
    const long long pairs = (long long)dims * (dims + 1) / 2;
    for (long long s0 = lo; s0 < hi; s0 += 4)
    {
        const int cnt = (int)((hi - s0 < 4) ? hi - s0 : 4);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt * dims; k += blockDim.x) tile[k] = (double)samples[(size_t)s0 * dims + k];
        __syncthreads();
        for (long long pr = threadIdx.x; pr < pairs; pr += blockDim.x)
        {
            // pair index -> (i, j <= i)
            int i = (int)((sqrt(8.0 * (double)pr + 1.0) - 1.0) * 0.5);
            while ((long long)i * (i + 1) / 2 > pr) --i;
            while ((long long)(i + 1) * (i + 2) / 2 <= pr) ++i;
            const int j = (int)(pr - (long long)i * (i + 1) / 2);
            double acc = 0.0;
            for (int k = 0; k < cnt; ++k) acc += tile[k * dims + i] * tile[k * dims + j];
            slot[(size_t)i * dp + j] += acc;
        }
        for (int p = threadIdx.x; p < dims; p += blockDim.x)
        {
            double acc = 0.0;
            for (int k = 0; k < cnt; ++k) acc += tile[k * dims + p];
            slot[(size_t)dp * dp + p] += acc;
        }
    }
}

// sums the slots in a fixed order (pairwise tree over the slot index): out[e] for every element of a slot
__global__ void __launch_bounds__(256) moments_reduce_kernel(const double* partial, int slots, long long slot_elems, double* out)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= slot_elems) return;
    // sequential Kahan sum over slots: a fixed order, a few thousand terms at most
    double s = 0.0, c = 0.0;
    for (int k = 0; k < slots; ++k)
    {
        const double y = partial[(size_t)k * slot_elems + e] - c;
        const double t = s + y;
        c = (t - s) - y;
        s = t;
    }
    out[e] = s;
}
}  // namespace

struct mcmcpp_hip_moments
{
    std::string error;
    int dtype = 0, device = 0, W = 0, D = 0, dp = 0, slots = 0;
    bool matrix_core = false;
    long long points = 0;       // samples accumulated so far
    size_t slot_elems = 0;
    double *d_partial = nullptr, *d_total = nullptr;
    void* d_chunk = nullptr;
    size_t chunk_bytes = 0;
    hipStream_t stream = nullptr;
};

namespace
{
thread_local std::string g_moments_error;

int fail(mcmcpp_hip_moments* m, int code, const char* what, hipError_t e = hipSuccess)
{
    std::string msg = what;
    if (e != hipSuccess) msg += std::string(": ") + hipGetErrorString(e);
    if (m)
        m->error = msg;
    else
        g_moments_error = msg;
    return code;
}

#define MOM_TRY(expr)                                                  \
    do                                                                 \
    {                                                                  \
        hipError_t e_ = (expr);                                        \
        if (e_ != hipSuccess) return fail(m, MCMCPP_HIP_E_HIP, #expr, e_); \
    } while (0)

template <class T>
int accumulate(mcmcpp_hip_moments* m, const T* dev_samples, long long n_samples)
{
    if (m->matrix_core)
    {
        const unsigned grid = (unsigned)(m->slots / kMomentWavesPerBlock);
        switch (m->dp / 16)
        {
        case 1: hipLaunchKernelGGL((moments_mfma_kernel<T, 1>), dim3(grid), dim3(64 * kMomentWavesPerBlock), 0, m->stream, dev_samples, n_samples, m->D, m->d_partial); break;
        case 2: hipLaunchKernelGGL((moments_mfma_kernel<T, 2>), dim3(grid), dim3(64 * kMomentWavesPerBlock), 0, m->stream, dev_samples, n_samples, m->D, m->d_partial); break;
        case 3: hipLaunchKernelGGL((moments_mfma_kernel<T, 3>), dim3(grid), dim3(64 * kMomentWavesPerBlock), 0, m->stream, dev_samples, n_samples, m->D, m->d_partial); break;
        default: hipLaunchKernelGGL((moments_mfma_kernel<T, 4>), dim3(grid), dim3(64 * kMomentWavesPerBlock), 0, m->stream, dev_samples, n_samples, m->D, m->d_partial); break;
        }
    }
    else
        hipLaunchKernelGGL((moments_generic_kernel<T>), dim3((unsigned)m->slots), dim3(256), sizeof(double) * 4 * (size_t)m->D, m->stream, dev_samples,
                           n_samples, m->D, m->dp, m->d_partial);
    MOM_TRY(hipGetLastError());
    return MCMCPP_HIP_OK;
}
}  // namespace

extern "C"
{
const char* mcmcpp_hip_moments_last_error(const mcmcpp_hip_moments* m) { return m ? m->error.c_str() : g_moments_error.c_str(); }

int mcmcpp_hip_moments_create(int32_t dtype, int32_t device, int32_t num_walkers, int32_t num_params, mcmcpp_hip_moments** out)
{
    mcmcpp_hip_moments* m = nullptr;
    if (!out) return fail(m, MCMCPP_HIP_E_ARG, "moments_create: out is NULL");
    *out = nullptr;
    if ((dtype != MCMCPP_HIP_F64 && dtype != MCMCPP_HIP_F32) || num_walkers < 1 || num_params < 1 || num_params > 1024)
        return fail(m, MCMCPP_HIP_E_ARG, "moments_create: dtype must be F64/F32, num_walkers >= 1, 1 <= num_params <= 1024");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(m, MCMCPP_HIP_E_NO_DEVICE, "no HIP device visible to this process");
    if (device >= ndev) return fail(m, MCMCPP_HIP_E_NO_DEVICE, "moments_create: device out of range");
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return fail(m, MCMCPP_HIP_E_HIP, "hipGetDevice failed");
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess)
        return fail(m, MCMCPP_HIP_E_HIP, "moments_create: cannot select the device");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(m, MCMCPP_HIP_E_NO_DEVICE, "this library is built for gfx950 only");
    m = new (std::nothrow) mcmcpp_hip_moments();
    if (!m) return MCMCPP_HIP_E_NOMEM;
    m->dtype = dtype;
    m->device = device;
    m->W = num_walkers;
    m->D = num_params;
    m->matrix_core = num_params <= 16 * kMaxTiles;
    m->dp = m->matrix_core ? 16 * ((num_params + 15) / 16) : num_params;
    // slots: one per wavefront (matrix cores: 4 wavefronts per CU) / one per workgroup, bounded by 256 MiB of partials
    m->slot_elems = (size_t)m->dp * m->dp + m->dp;
    m->slots = m->matrix_core ? prop.multiProcessorCount * kMomentWavesPerBlock * 2 : prop.multiProcessorCount;  // two workgroups per CU
    while (m->slots > kMomentWavesPerBlock && (size_t)m->slots * m->slot_elems * sizeof(double) > ((size_t)256 << 20)) m->slots /= 2;
    m->slots -= m->slots % kMomentWavesPerBlock;
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&m->d_partial, sizeof(double) * (size_t)m->slots * m->slot_elems) != hipSuccess ||
        hipMalloc(&m->d_total, sizeof(double) * m->slot_elems) != hipSuccess ||
        hipMemset(m->d_partial, 0, sizeof(double) * (size_t)m->slots * m->slot_elems) != hipSuccess)
    {
        g_moments_error = "moments_create: cannot allocate device memory";
        mcmcpp_hip_moments_destroy(m);
        return MCMCPP_HIP_E_NOMEM;
    }
    *out = m;
    return MCMCPP_HIP_OK;
}

void mcmcpp_hip_moments_destroy(mcmcpp_hip_moments* m)
{
    if (!m) return;
    hipSetDevice(m->device);
    if (m->stream) hipStreamSynchronize(m->stream);
    if (m->d_partial) hipFree(m->d_partial);
    if (m->d_total) hipFree(m->d_total);
    if (m->d_chunk) hipFree(m->d_chunk);
    if (m->stream) hipStreamDestroy(m->stream);
    delete m;
}

int mcmcpp_hip_moments_reset(mcmcpp_hip_moments* m)
{
    if (!m) return MCMCPP_HIP_E_ARG;
    MOM_TRY(hipSetDevice(m->device));
    MOM_TRY(hipMemsetAsync(m->d_partial, 0, sizeof(double) * (size_t)m->slots * m->slot_elems, m->stream));
    MOM_TRY(hipStreamSynchronize(m->stream));
    m->points = 0;
    return MCMCPP_HIP_OK;
}

int mcmcpp_hip_moments_add_steps(mcmcpp_hip_moments* m, const void* steps, int64_t n_steps, int64_t step_stride)
{
    if (!m) return MCMCPP_HIP_E_ARG;
    if (n_steps < 0 || step_stride < 1 || (n_steps > 0 && !steps)) return fail(m, MCMCPP_HIP_E_ARG, "moments_add_steps: bad arguments");
    if (n_steps == 0) return MCMCPP_HIP_OK;
    MOM_TRY(hipSetDevice(m->device));
    const size_t esize = m->dtype == MCMCPP_HIP_F64 ? 8 : 4;
    const size_t step_bytes = esize * (size_t)m->W * m->D;
    // chunks of up to 64 MiB (at least one step)
    int64_t per_chunk = (int64_t)(((size_t)64 << 20) / step_bytes);
    if (per_chunk < 1) per_chunk = 1;
    if (per_chunk > n_steps) per_chunk = n_steps;
    if (m->chunk_bytes < step_bytes * (size_t)per_chunk)
    {
        MOM_TRY(hipStreamSynchronize(m->stream));
        if (m->d_chunk) hipFree(m->d_chunk);
        m->d_chunk = nullptr;
        m->chunk_bytes = 0;
        if (hipMalloc(&m->d_chunk, step_bytes * (size_t)per_chunk) != hipSuccess) return fail(m, MCMCPP_HIP_E_NOMEM, "moments_add_steps: cannot allocate the upload buffer");
        m->chunk_bytes = step_bytes * (size_t)per_chunk;
    }
    for (int64_t first = 0; first < n_steps; first += per_chunk)
    {
        const int64_t now = (n_steps - first < per_chunk) ? n_steps - first : per_chunk;
        const char* src = (const char*)steps + step_bytes * (size_t)(first * step_stride);
        MOM_TRY(hipStreamSynchronize(m->stream));  // the previous chunk's kernel has read the buffer
        if (step_stride == 1)
            MOM_TRY(hipMemcpyAsync(m->d_chunk, src, step_bytes * (size_t)now, hipMemcpyHostToDevice, m->stream));
        else
            for (int64_t k = 0; k < now; ++k)
                MOM_TRY(hipMemcpyAsync((char*)m->d_chunk + step_bytes * (size_t)k, src + step_bytes * (size_t)(k * step_stride), step_bytes,
                                       hipMemcpyHostToDevice, m->stream));
        const long long n_samples = (long long)now * m->W;
        const int rc = m->dtype == MCMCPP_HIP_F64 ? accumulate<double>(m, (const double*)m->d_chunk, n_samples)
                                                  : accumulate<float>(m, (const float*)m->d_chunk, n_samples);
        if (rc) return rc;
        m->points += n_samples;
    }
    MOM_TRY(hipStreamSynchronize(m->stream));
    return MCMCPP_HIP_OK;
}

int mcmcpp_hip_moments_add_device_steps(mcmcpp_hip_moments* m, const void* device_steps, int64_t n_steps)
{
    if (!m) return MCMCPP_HIP_E_ARG;
    if (n_steps < 0 || (n_steps > 0 && !device_steps)) return fail(m, MCMCPP_HIP_E_ARG, "moments_add_device_steps: bad arguments");
    if (n_steps == 0) return MCMCPP_HIP_OK;
    MOM_TRY(hipSetDevice(m->device));
    const long long n_samples = (long long)n_steps * m->W;
    const int rc = m->dtype == MCMCPP_HIP_F64 ? accumulate<double>(m, (const double*)device_steps, n_samples)
                                              : accumulate<float>(m, (const float*)device_steps, n_samples);
    if (rc) return rc;
    m->points += n_samples;
    MOM_TRY(hipStreamSynchronize(m->stream));
    return MCMCPP_HIP_OK;
}

int mcmcpp_hip_moments_finish(mcmcpp_hip_moments* m, int64_t* num_points, void* mean, void* cov, void* corr)
{
    if (!m) return MCMCPP_HIP_E_ARG;
    if (m->points < 1) return fail(m, MCMCPP_HIP_E_STATE, "moments_finish: no samples have been added");
    MOM_TRY(hipSetDevice(m->device));
    hipLaunchKernelGGL(moments_reduce_kernel, dim3((unsigned)((m->slot_elems + 255) / 256)), dim3(256), 0, m->stream, m->d_partial, m->slots,
                       (long long)m->slot_elems, m->d_total);
    MOM_TRY(hipGetLastError());
    std::vector<double> tot(m->slot_elems);
    MOM_TRY(hipMemcpyAsync(tot.data(), m->d_total, sizeof(double) * m->slot_elems, hipMemcpyDeviceToHost, m->stream));
    MOM_TRY(hipStreamSynchronize(m->stream));
    if (num_points) *num_points = m->points;
    // CovarianceMatrix::finalizeMatrix (CovarianceMatrix.h:178-224), in fp64; narrowed to the chain's type at the end
    const int D = m->D, dp = m->dp;
    const double count = (double)m->points;
    std::vector<double> avg(D), cv((size_t)D * D), cr((size_t)D * D);
    for (int i = 0; i < D; ++i) avg[i] = tot[(size_t)dp * dp + i] / count;
    for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j)
        {
            const double t = tot[(size_t)i * dp + j] / count - avg[i] * avg[j];
            cv[(size_t)i * D + j] = t;
            cv[(size_t)j * D + i] = t;
        }
    for (int i = 0; i < D; ++i)
    {
        const double sv = std::sqrt(cv[(size_t)i * D + i]);
        for (int j = 0; j < D; ++j) cr[(size_t)i * D + j] = cv[(size_t)i * D + j] / sv;
    }
    for (int j = 0; j < D; ++j)
    {
        const double sv = std::sqrt(cv[(size_t)j * D + j]);
        for (int i = 0; i < D; ++i) cr[(size_t)i * D + j] /= sv;
    }
    auto put = [&](void* dst, const std::vector<double>& src) {
        if (!dst) return;
        if (m->dtype == MCMCPP_HIP_F64)
            std::memcpy(dst, src.data(), sizeof(double) * src.size());
        else
            for (size_t k = 0; k < src.size(); ++k) ((float*)dst)[k] = (float)src[k];
    };
    put(mean, avg);
    put(cov, cv);
    put(corr, cr);
    return MCMCPP_HIP_OK;
}
}

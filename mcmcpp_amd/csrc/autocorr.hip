// autocorr.hip -- integrated autocorrelation times of the parameters in a stored chain on the GPU: the device side of
// Analysis::AutoCorrCalc (reference: MCMCpp/Analysis/AutoCorrCalc.h:151-207 and Analysis/Detail/AutoCov.h:146-322;
// SURVEY.md 8f row f2).
//
// The reference takes one (walker, parameter) series at a time on one core: subtract the series' Kahan average, radix-2
// FFT of the zero-extended series (next power of two >= n, so the autocovariance is the circular one), |.|^2, inverse
// FFT, divide by lag 0; the W functions of a parameter are Kahan-summed, divided by W, and the windowed sum of that
// average is the autocorrelation time.  Here:
//   * the stored steps ([n][W][D], host memory) are uploaded once (288 GB of HBM: a whole chain fits);
//   * one thread per series forms the Kahan average in the reference's order (coalesced across series);
//   * one workgroup per series runs the two transforms in LDS (split real/imaginary arrays, 256 threads sharing the
//     butterflies, two stages per pass), W * D workgroups in flight; series too long for 64 KB of LDS use a scratch array in
//     global memory with the same code;
//   * the sum over walkers runs one thread per (parameter, lag), walkers in the reference's order;
//   * the windowed sum is sequential by nature: one thread per parameter, fed from LDS tiles.
// Every butterfly, product and division is the reference's own expression (contraction off), the twiddle factors are
// computed on the host with the same libm calls, and sums keep the reference's order: results are BIT-IDENTICAL to
// the oracle's restatement (oracle/stretch_oracle_typed.inc, autocorr_times with emulate_defect = 0), which is itself
// pinned bit for bit to the reference's Detail::AutoCov and, with its defect emulation on, to the whole class.
// Two deliberate differences from the reference, both documented in INTEGRATION.md: transferWalker's accumulation
// onto stale (and initially uninitialised) scratch memory is not reproduced, and a walker subset is an evenly spaced
// one instead of a draw from std::random_device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mcmcpp_hip.h"

namespace
{
constexpr int kAcThreads = 256;
constexpr int kXcds = 8;  // MI355X: 8 XCDs x 32 CUs, workgroups dispatched to them round robin
constexpr size_t kAcLdsBytes = 64 << 10;

__device__ __forceinline__ unsigned ac_bit_reverse(unsigned x, int lg) { return __builtin_bitreverse32(x) >> (32 - lg); }

// avg[sel * D + p] = Kahan sum over the n steps of walker walker_idx[sel], parameter p, divided by n (AutoCorrCalc.h:248-276)
template <class T>
__global__ void __launch_bounds__(kAcThreads) ac_chain_average_kernel(const T* steps, int n, long long step_elems, const int* walker_idx, int use, int D, T* avg)
{
    const long long j = (long long)blockIdx.x * kAcThreads + threadIdx.x;
    if (j >= (long long)use * D) return;
    const int sel = (int)(j / D), p = (int)(j % D);
    const T* src = steps + (size_t)walker_idx[sel] * D + p;
    T sum = T(0), comp = T(0);
    for (int t = 0; t < n; ++t)
    {
        const T value = src[(size_t)t * step_elems] - comp;
        const T temp = sum + value;
        comp = (temp - sum) - value;
        sum = temp;
    }
    avg[j] = sum / (T)n;
}

// The iterative Cooley-Tukey transform over bit-reversed input (AutoCov.h:166-216); sign -1 forward, +1 inverse.
// One butterfly of stage s: (x[i0], x[i1]) <- (x[i0] + w x[i1], x[i0] - w x[i1]), w = tw[j << (lg - s)], i1 = i0 + 2^(s-1).
// Two stages at a time: a thread takes the four elements i, i+m, i+2m, i+3m (m = 2^(s-1)) that stages s and s+1 combine
// among themselves and runs the reference's four butterflies on them in registers, in the reference's order of
// operations -- same roundings, half the LDS traffic and half the workgroup barriers.  A last single stage when lg is odd.
template <class T>
__device__ __forceinline__ void ac_butterfly(T wr, T wi, T& ar, T& ai, T& br, T& bi)
{
    const T t1r = wr * br - wi * bi, t1i = wr * bi + wi * br;
    const T t2r = ar, t2i = ai;
    ar = t2r + t1r;
    ai = t2i + t1i;
    br = t2r - t1r;
    bi = t2i - t1i;
}

template <class T>
__device__ __forceinline__ void ac_fft_pass(T* re, T* im, int lg, const T* __restrict__ tw, T sign)
{
    const int quarter = 1 << (lg > 1 ? lg - 2 : 0);
    int s = 1;
    for (; s + 1 <= lg; s += 2)
    {
        const int m = 1 << (s - 1);
        for (int b = threadIdx.x; b < quarter; b += kAcThreads)
        {
            const int j = b & (m - 1);
            const int i = ((b >> (s - 1)) << (s + 1)) + j;
            // stage s: (i, i+m) and (i+2m, i+3m) with twiddle index j << (lg - s)
            const int t_a = j << (lg - s);
            const T war = tw[2 * t_a], wai = sign * tw[2 * t_a + 1];
            T x0r = re[i], x0i = im[i], x1r = re[i + m], x1i = im[i + m];
            T x2r = re[i + 2 * m], x2i = im[i + 2 * m], x3r = re[i + 3 * m], x3i = im[i + 3 * m];
            ac_butterfly(war, wai, x0r, x0i, x1r, x1i);
            ac_butterfly(war, wai, x2r, x2i, x3r, x3i);
            // stage s+1: (i, i+2m) with index j << (lg - s - 1), (i+m, i+3m) with index (j + m) << (lg - s - 1)
            const int t_b = j << (lg - s - 1), t_c = (j + m) << (lg - s - 1);
            const T wbr = tw[2 * t_b], wbi = sign * tw[2 * t_b + 1];
            const T wcr = tw[2 * t_c], wci = sign * tw[2 * t_c + 1];
            ac_butterfly(wbr, wbi, x0r, x0i, x2r, x2i);
            ac_butterfly(wcr, wci, x1r, x1i, x3r, x3i);
            re[i] = x0r;
            im[i] = x0i;
            re[i + m] = x1r;
            im[i + m] = x1i;
            re[i + 2 * m] = x2r;
            im[i + 2 * m] = x2i;
            re[i + 3 * m] = x3r;
            im[i + 3 * m] = x3i;
        }
        __syncthreads();
    }
    if (s == lg)
    {
        const int half = 1 << (lg - 1);
        const int m2 = 1 << (s - 1);
        for (int b = threadIdx.x; b < half; b += kAcThreads)
        {
            const int j = b & (m2 - 1);
            const int i0 = ((b >> (s - 1)) << s) + j, i1 = i0 + m2;
            const int ti = j << (lg - s);
            T ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
            ac_butterfly(tw[2 * ti], sign * tw[2 * ti + 1], ar, ai, br, bi);
            re[i0] = ar;
            im[i0] = ai;
            re[i1] = br;
            im[i1] = bi;
        }
        __syncthreads();
    }
}

// acov[series][0..n): normalised autocovariance function of series (first_sel + series / D, series % D)
template <class T, bool LDS>
__global__ void __launch_bounds__(kAcThreads)
ac_autocov_kernel(const T* steps, int n, long long step_elems, const int* walker_idx, int first_sel, int D, const T* avg, const T* __restrict__ tw, int lg, T* scratch, T* acov,
                  int series_count)
{
    extern __shared__ __align__(16) unsigned char ac_smem[];
    const int fft = 1 << lg;
    // Workgroups go to the eight XCDs round robin, and each XCD has an L2 of its own.  A series is read with a stride of
    // a whole stored step (the chain is step-major), eight bytes out of every line, and the other words of those lines
    // belong to the NEIGHBOURING series: so neighbouring series go to the same XCD (XCD x takes series [x, x + 1) * count / 8
    // in order), where the workgroups resident together share the lines in L2 instead of each fetching them from memory.
    const int per_xcd = (series_count + kXcds - 1) / kXcds;
    const int series = (int)(blockIdx.x % kXcds) * per_xcd + (int)(blockIdx.x / kXcds);
    if (series >= series_count) return;  // (the grid is whole rounds over the XCDs)
    const int sel = first_sel + series / D, p = series % D;
    T* re = LDS ? reinterpret_cast<T*>(ac_smem) : scratch + (size_t)series * 2 * fft;
    T* im = re + fft;
    const T* src = steps + (size_t)walker_idx[sel] * D + p;
    const T mean = avg[(size_t)sel * D + p];
    for (int i = threadIdx.x; i < fft; i += kAcThreads)  // AutoCov.h:241-252
    {
        const unsigned r = ac_bit_reverse((unsigned)i, lg);
        re[r] = i < n ? src[(size_t)i * step_elems] - mean : T(0);
        im[r] = T(0);
    }
    __syncthreads();
    ac_fft_pass<T>(re, im, lg, tw, T(-1));
    for (int i = threadIdx.x; i < fft; i += kAcThreads)  // |.|^2 into bit-reversed order, in place (AutoCov.h:187-195)
    {
        const int r = (int)ac_bit_reverse((unsigned)i, lg);
        if (i < r)
        {
            const T vi = re[i] * re[i] + im[i] * im[i];
            const T vr = re[r] * re[r] + im[r] * im[r];
            re[r] = vi;
            re[i] = vr;
            im[r] = T(0);
            im[i] = T(0);
        }
        else if (i == r)
        {
            re[i] = re[i] * re[i] + im[i] * im[i];
            im[i] = T(0);
        }
    }
    __syncthreads();
    ac_fft_pass<T>(re, im, lg, tw, T(1));
    const T norm = re[0];
    T* out = acov + (size_t)series * n;
    for (int i = threadIdx.x; i < n; i += kAcThreads) out[i] = re[i] / norm;  // AutoCov.h:157-163
}

// sum[p][i], comp[p][i] += the chunk's functions, walkers in order (AutoCorrCalc.h:209-221, the reference's own
// compensation formula)
template <class T>
__global__ void __launch_bounds__(kAcThreads) ac_accumulate_kernel(const T* acov, int n, int D, int chunk_walkers, T* sum, T* comp)
{
    const long long idx = (long long)blockIdx.x * kAcThreads + threadIdx.x;
    if (idx >= (long long)D * n) return;
    const int p = (int)(idx / n), i = (int)(idx % n);
    T s = sum[idx], c = comp[idx];
    // eight loads in flight per thread: the additions have to follow one another, the loads do not
    constexpr int kBatch = 8;
    const size_t walker_stride = (size_t)D * n;
    const T* src = acov + (size_t)p * n + i;
    int w = 0;
    for (; w + kBatch <= chunk_walkers; w += kBatch)
    {
        T v[kBatch];
#pragma unroll
        for (int b = 0; b < kBatch; ++b) v[b] = src[(size_t)(w + b) * walker_stride];
#pragma unroll
        for (int b = 0; b < kBatch; ++b)
        {
            const T value = v[b] + c;
            const T temp = s + value;
            c = (temp - s) - value;
            s = temp;
        }
    }
    for (; w < chunk_walkers; ++w)
    {
        const T value = src[(size_t)w * walker_stride] + c;
        const T temp = s + value;
        c = (temp - s) - value;
        s = temp;
    }
    sum[idx] = s;
    comp[idx] = c;
}

// one workgroup per parameter: sum /= use (AutoCorrCalc.h:223-231), then the windowed sum (AutoCorrCalc.h:185-206)
template <class T>
__global__ void __launch_bounds__(kAcThreads) ac_window_kernel(T* sum, int n, int use, int window_scaling, T* times)
{
    __shared__ T tile[kAcThreads];
    __shared__ int done;
    T* f = sum + (size_t)blockIdx.x * n;
    const T div = (T)use;
    if (threadIdx.x == 0) done = 0;
    T acs = T(0), comp = T(0);
    const T factor = (T)window_scaling;
    for (int base = 0; base < n; base += kAcThreads)
    {
        const int i = base + threadIdx.x;
        T v = T(0);
        if (i < n)
        {
            v = f[i] / div;
            f[i] = v;
        }
        __syncthreads();  // (previous tile consumed; `done` of the previous tile visible)
        tile[threadIdx.x] = v;
        __syncthreads();
        if (threadIdx.x == 0 && !done)
        {
            if (base == 0) acs = -tile[0];
            const int m = n - base < kAcThreads ? n - base : kAcThreads;
            for (int k = 0; k < m; ++k)
            {
                const T value = (T(2) * tile[k]) - comp;
                const T temp = acs + value;
                comp = (temp - acs) - value;
                acs = temp;
                if ((T)(base + k) > factor * acs)
                {
                    done = 1;
                    break;
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) times[blockIdx.x] = done ? acs : -acs;
}

// cos and sin of one angle: g++ -O2, which the reference and the oracle are built with, folds the pair into ONE sincos
// call, and glibc's sincos is not bit-identical to its separate cos and sin for every argument (first difference in
// the 4096-point table, entry 1955).  The table follows the reference as built: sincos.
inline void ac_sincos(double a, double* s, double* c) { ::sincos(a, s, c); }
inline void ac_sincos(float a, float* s, float* c) { ::sincosf(a, s, c); }

thread_local std::string g_ac_error;

int ac_fail(int code, const std::string& msg)
{
    g_ac_error = msg;
    return code;
}

struct DeviceBuffers
{
    std::vector<void*> ptrs;
    hipStream_t stream = nullptr;
    ~DeviceBuffers()
    {
        if (stream) (void)hipStreamSynchronize(stream);
        for (void* p : ptrs) (void)hipFree(p);
        if (stream) (void)hipStreamDestroy(stream);
    }
    template <class U>
    bool alloc(U** out, size_t count)
    {
        void* p = nullptr;
        if (hipMalloc(&p, sizeof(U) * (count ? count : 1)) != hipSuccess) return false;
        ptrs.push_back(p);
        *out = static_cast<U*>(p);
        return true;
    }
};

#define AC_TRY(expr)                                                                                          \
    do                                                                                                        \
    {                                                                                                         \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return ac_fail(MCMCPP_HIP_E_HIP, std::string(#expr ": ") + hipGetErrorString(e_)); \
    } while (0)

template <class T>
// steps: n_steps host pointers, or nullptr when device_steps holds the whole [n_steps][W][D] chain in device memory
int autocorr_times(const void* const* steps, const T* device_steps, int64_t n_steps, int W, int D, int use, int window_scaling, T* times, T* functions)
{
    const int n = (int)n_steps;
    const size_t step_elems = (size_t)W * D;
    const int lg = (int)std::ceil(std::log2((double)n));  // AutoCov.h:318-322
    const int fft = 1 << lg;
    const bool lds = sizeof(T) * 2 * (size_t)fft <= kAcLdsBytes;

    // host side: twiddle factors (AutoCov.h:304-315; the rational two pi of AutoCov.h:29-30,135) and the walker list
    std::vector<T> tw((size_t)fft);  // fft/2 (cos, sin) pairs
    const T two_pi = T(2) * (static_cast<T>(3141592653589793239ULL) / static_cast<T>(1000000000000000000ULL));
    for (int i = 0; i < fft / 2; ++i)
    {
        const T angle = two_pi * static_cast<T>(i) / static_cast<T>(fft);
        ac_sincos(angle, &tw[2 * (size_t)i + 1], &tw[2 * (size_t)i]);
    }
    std::vector<int> walker_idx((size_t)use);
    for (int i = 0; i < use; ++i) walker_idx[(size_t)i] = (int)(((int64_t)i * W) / use);  // all of them, or evenly spaced

    DeviceBuffers dev;
    AC_TRY(hipStreamCreateWithFlags(&dev.stream, hipStreamNonBlocking));
    // walkers per pass over the transforms: bounded by 1 GiB of functions and 2 GiB of global scratch
    size_t per_walker = sizeof(T) * (size_t)D * n;
    size_t per_walker_scratch = lds ? 0 : sizeof(T) * (size_t)D * 2 * fft;
    int chunk = use;
    while (chunk > 1 && ((size_t)chunk * per_walker > ((size_t)1 << 30) || (size_t)chunk * per_walker_scratch > ((size_t)2 << 30))) chunk = (chunk + 1) / 2;

    T *d_avg, *d_tw, *d_acov, *d_sum, *d_comp, *d_times, *d_scratch = nullptr;
    T* d_upload = nullptr;
    int* d_idx;
    if ((steps && !dev.alloc(&d_upload, (size_t)n * step_elems)) || !dev.alloc(&d_avg, (size_t)use * D) || !dev.alloc(&d_tw, (size_t)fft) ||
        !dev.alloc(&d_acov, (size_t)chunk * D * n) || !dev.alloc(&d_sum, (size_t)D * n) || !dev.alloc(&d_comp, (size_t)D * n) ||
        !dev.alloc(&d_times, (size_t)D) || !dev.alloc(&d_idx, (size_t)use) || (!lds && !dev.alloc(&d_scratch, (size_t)chunk * D * 2 * fft)))
        return ac_fail(MCMCPP_HIP_E_NOMEM, "autocorr_times: cannot allocate device memory for the chain and the work arrays");

    const T* d_steps = steps ? d_upload : device_steps;
    // upload: steps that follow each other in host memory go in one copy
    for (int64_t s = 0; steps && s < n_steps;)
    {
        int64_t e = s + 1;
        while (e < n_steps && static_cast<const T*>(steps[e]) == static_cast<const T*>(steps[e - 1]) + step_elems) ++e;
        AC_TRY(hipMemcpyAsync(d_upload + (size_t)s * step_elems, steps[s], sizeof(T) * (size_t)(e - s) * step_elems, hipMemcpyHostToDevice, dev.stream));
        s = e;
    }
    AC_TRY(hipMemcpyAsync(d_tw, tw.data(), sizeof(T) * tw.size(), hipMemcpyHostToDevice, dev.stream));
    AC_TRY(hipMemcpyAsync(d_idx, walker_idx.data(), sizeof(int) * walker_idx.size(), hipMemcpyHostToDevice, dev.stream));
    AC_TRY(hipMemsetAsync(d_sum, 0, sizeof(T) * (size_t)D * n, dev.stream));
    AC_TRY(hipMemsetAsync(d_comp, 0, sizeof(T) * (size_t)D * n, dev.stream));

    const long long series_total = (long long)use * D;
    hipLaunchKernelGGL(ac_chain_average_kernel<T>, dim3((unsigned)((series_total + kAcThreads - 1) / kAcThreads)), dim3(kAcThreads), 0, dev.stream, d_steps, n,
                       (long long)step_elems, d_idx, use, D, d_avg);
    for (int first = 0; first < use; first += chunk)
    {
        const int cw = use - first < chunk ? use - first : chunk;
        const unsigned ac_grid = (unsigned)(((cw * D + kXcds - 1) / kXcds) * kXcds);  // (whole rounds over the XCDs)
        if (lds)
            hipLaunchKernelGGL((ac_autocov_kernel<T, true>), dim3(ac_grid), dim3(kAcThreads), sizeof(T) * 2 * (size_t)fft, dev.stream, d_steps, n,
                               (long long)step_elems, d_idx, first, D, d_avg, d_tw, lg, d_scratch, d_acov, cw * D);
        else
            hipLaunchKernelGGL((ac_autocov_kernel<T, false>), dim3(ac_grid), dim3(kAcThreads), 0, dev.stream, d_steps, n, (long long)step_elems, d_idx,
                               first, D, d_avg, d_tw, lg, d_scratch, d_acov, cw * D);
        hipLaunchKernelGGL(ac_accumulate_kernel<T>, dim3((unsigned)(((long long)D * n + kAcThreads - 1) / kAcThreads)), dim3(kAcThreads), 0, dev.stream, d_acov, n,
                           D, cw, d_sum, d_comp);
    }
    hipLaunchKernelGGL(ac_window_kernel<T>, dim3((unsigned)D), dim3(kAcThreads), 0, dev.stream, d_sum, n, use, window_scaling, d_times);
    AC_TRY(hipGetLastError());
    AC_TRY(hipMemcpyAsync(times, d_times, sizeof(T) * (size_t)D, hipMemcpyDeviceToHost, dev.stream));
    if (functions) AC_TRY(hipMemcpyAsync(functions, d_sum, sizeof(T) * (size_t)D * n, hipMemcpyDeviceToHost, dev.stream));
    AC_TRY(hipStreamSynchronize(dev.stream));
    return MCMCPP_HIP_OK;
}
}  // namespace

extern "C"
{
const char* mcmcpp_hip_autocorr_last_error(void) { return g_ac_error.c_str(); }

static int autocorr_entry(int32_t dtype, int32_t device, const void* const* steps, const void* device_steps, int64_t n_steps, int32_t num_walkers,
                          int32_t num_params, int32_t walkers_to_use, int32_t window_scaling, void* times, void* functions)
{
    if ((!steps && !device_steps) || !times) return ac_fail(MCMCPP_HIP_E_ARG, "autocorr_times: steps and times must not be NULL");
    if ((dtype != MCMCPP_HIP_F64 && dtype != MCMCPP_HIP_F32) || num_walkers < 1 || num_params < 1)
        return ac_fail(MCMCPP_HIP_E_ARG, "autocorr_times: dtype must be F64/F32, num_walkers >= 1, num_params >= 1");
    if (n_steps < 2 || n_steps > (int64_t(1) << 24)) return ac_fail(MCMCPP_HIP_E_ARG, "autocorr_times: 2 <= n_steps <= 2^24");
    if (walkers_to_use < 0 || walkers_to_use > num_walkers) return ac_fail(MCMCPP_HIP_E_ARG, "autocorr_times: 0 <= walkers_to_use <= num_walkers");
    if ((int64_t)num_params * n_steps >= (int64_t(1) << 31) || (int64_t)num_walkers * num_params >= (int64_t(1) << 31))
        return ac_fail(MCMCPP_HIP_E_ARG, "autocorr_times: num_params * n_steps and num_walkers * num_params must be below 2^31");
    for (int64_t s = 0; steps && s < n_steps; ++s)
        if (!steps[s]) return ac_fail(MCMCPP_HIP_E_ARG, "autocorr_times: a step pointer is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return ac_fail(MCMCPP_HIP_E_NO_DEVICE, "no HIP device visible to this process");
    if (device >= ndev) return ac_fail(MCMCPP_HIP_E_NO_DEVICE, "autocorr_times: device out of range");
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return ac_fail(MCMCPP_HIP_E_HIP, "hipGetDevice failed");
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess)
        return ac_fail(MCMCPP_HIP_E_HIP, "autocorr_times: cannot select the device");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ac_fail(MCMCPP_HIP_E_NO_DEVICE, "this library is built for gfx950 only");
    const int use = walkers_to_use == 0 ? num_walkers : walkers_to_use;
    if (dtype == MCMCPP_HIP_F64)
        return autocorr_times<double>(steps, static_cast<const double*>(device_steps), n_steps, num_walkers, num_params, use, window_scaling,
                                      static_cast<double*>(times), static_cast<double*>(functions));
    return autocorr_times<float>(steps, static_cast<const float*>(device_steps), n_steps, num_walkers, num_params, use, window_scaling,
                                 static_cast<float*>(times), static_cast<float*>(functions));
}

int mcmcpp_hip_autocorr_times(int32_t dtype, int32_t device, const void* const* steps, int64_t n_steps, int32_t num_walkers, int32_t num_params,
                              int32_t walkers_to_use, int32_t window_scaling, void* times, void* functions)
{
    if (!steps) return ac_fail(MCMCPP_HIP_E_ARG, "autocorr_times: steps must not be NULL");
    return autocorr_entry(dtype, device, steps, nullptr, n_steps, num_walkers, num_params, walkers_to_use, window_scaling, times, functions);
}

int mcmcpp_hip_autocorr_times_device(int32_t dtype, int32_t device, const void* device_steps, int64_t n_steps, int32_t num_walkers, int32_t num_params,
                                     int32_t walkers_to_use, int32_t window_scaling, void* times, void* functions)
{
    if (!device_steps) return ac_fail(MCMCPP_HIP_E_ARG, "autocorr_times_device: device_steps must not be NULL");
    return autocorr_entry(dtype, device, nullptr, device_steps, n_steps, num_walkers, num_params, walkers_to_use, window_scaling, times, functions);
}
}

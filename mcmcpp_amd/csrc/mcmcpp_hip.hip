// mcmcpp_hip.hip -- host side of libmcmcpp_hip.so: the C ABI of include/mcmcpp_hip.h on top of the
// gfx950 kernels in stretch_kernel.hpp.
//
// Reference roles replaced (paths relative to /root/reference):
//   EnsembleSampler ctor / setInitialWalkerPos / runMCMC / reset / counters  MCMCpp/EnsembleSampler.h:199-360
//   ParallelEnsembleSampler's thread pool + red/black controller             MCMCpp/Threading/*.h
//     -> one kernel launch per ensemble step (small ensembles: full_step_kernel.hpp) or per half-step (large ones)
//        on one HIP stream, replayed from a hipGraph; the stream and step counters travel in device memory
//        (StepCtl) so a replay needs no host-side updates
//   Walker[] (heap row per walker)  MCMCpp/Walker/Walker.h:142-149 -> pos[W][D], logp[W], n_accept[W] in HBM
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcmcpp_hip.h"
#define MCMCPP_DEFINE_REDUCE_KERNEL
#include "launch_table.hpp"
#include "sampler_base.hpp"

using namespace mcmcpp;

namespace
{
thread_local std::string g_create_error;



// memcpy of a large block split over a few threads (the un-overlapped tail of a run's chain download: a single core
// moves ~12 GB/s into pageable memory)
void parallel_memcpy(char* dst, const char* src, size_t bytes)
{
    const size_t kMinPiece = 512u << 10;
    int pieces = (int)(bytes / kMinPiece);
    if (pieces > 4) pieces = 4;
    if (pieces < 2)
    {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t piece = ((bytes / (size_t)pieces) + 63) & ~(size_t)63;
    std::thread helpers[3];
    for (int k = 1; k < pieces; ++k)
    {
        const size_t off = piece * (size_t)k;
        const size_t len = (k == pieces - 1) ? bytes - off : piece;
        helpers[k - 1] = std::thread([=]() { std::memcpy(dst + off, src + off, len); });
    }
    std::memcpy(dst, src, piece);
    for (int k = 1; k < pieces; ++k) helpers[k - 1].join();
}

long env_long(const char* name, long fallback)
{
    const char* v = std::getenv(name);
    return (v && *v) ? std::strtol(v, nullptr, 10) : fallback;
}

struct RegisteredCalc
{
    const void* f64;
    const void* f32;
    int params_len;
};
std::mutex g_registry_mutex;
std::map<int, RegisteredCalc> g_registry;

bool registered_calc(int calc_id, RegisteredCalc* out)
{
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    std::map<int, RegisteredCalc>::const_iterator it = g_registry.find(calc_id);
    if (it == g_registry.end()) return false;
    if (out) *out = it->second;
    return true;
}

template <class T>
const LaunchTable<T>* table_for(int calc_id);
template <>
const LaunchTable<double>* table_for<double>(int calc_id)
{
    switch (calc_id)
    {
    case MCMCPP_HIP_CALC_ISO_GAUSSIAN: return launch_table_f64_iso();
    case MCMCPP_HIP_CALC_DENSE_GAUSSIAN: return launch_table_f64_dense();
    case MCMCPP_HIP_CALC_ROSENBROCK: return launch_table_f64_rosenbrock();
    case MCMCPP_HIP_CALC_SKEWED_GAUSSIAN_2D: return launch_table_f64_skewed();
    default:
    {
        RegisteredCalc r;
        return registered_calc(calc_id, &r) ? static_cast<const LaunchTable<double>*>(r.f64) : nullptr;
    }
    }
}
template <>
const LaunchTable<float>* table_for<float>(int calc_id)
{
    switch (calc_id)
    {
    case MCMCPP_HIP_CALC_ISO_GAUSSIAN: return launch_table_f32_iso();
    case MCMCPP_HIP_CALC_DENSE_GAUSSIAN: return launch_table_f32_dense();
    case MCMCPP_HIP_CALC_ROSENBROCK: return launch_table_f32_rosenbrock();
    case MCMCPP_HIP_CALC_SKEWED_GAUSSIAN_2D: return launch_table_f32_skewed();
    default:
    {
        RegisteredCalc r;
        return registered_calc(calc_id, &r) ? static_cast<const LaunchTable<float>*>(r.f32) : nullptr;
    }
    }
}
}  // namespace

namespace mcmcpp
{
const void* launch_table_lookup(int dtype, int calc_id)
{
    return dtype == MCMCPP_HIP_F64 ? static_cast<const void*>(table_for<double>(calc_id)) : static_cast<const void*>(table_for<float>(calc_id));
}
void launch_fill_draws(const HalfStepArgs<double>& a, U128 base, const U128* red_base, hipStream_t stream)
{
    const unsigned grid = (unsigned)((3 * (long)a.shard_count + 255) / 256);
    hipLaunchKernelGGL(fill_draws_kernel<double>, dim3(grid), dim3(256), 0, stream, a, base, red_base ? *red_base : base, red_base ? 1 : 0);
}
void launch_fill_draws(const HalfStepArgs<float>& a, U128 base, const U128* red_base, hipStream_t stream)
{
    const unsigned grid = (unsigned)((3 * (long)a.shard_count + 255) / 256);
    hipLaunchKernelGGL(fill_draws_kernel<float>, dim3(grid), dim3(256), 0, stream, a, base, red_base ? *red_base : base, red_base ? 1 : 0);
}
void launch_accepted_reduce(const uint32_t* partials, int partial_slots, int partial_waves, int count,
                            const StepCtl* ctl_after, const RunInfo* run, hipStream_t stream)
{
    hipLaunchKernelGGL(accepted_reduce_kernel, dim3((unsigned)count), dim3(256), 0, stream, partials, partial_slots,
                       partial_waves, count, ctl_after, run);
}
}  // namespace mcmcpp

namespace
{
template <class T>
class Sampler final : public mcmcpp_hip_sampler
{
public:
    Sampler() {}
    ~Sampler() override { release(); }

    int init(const mcmcpp_hip_config& c)
    {
        cfg = c;
        W = c.num_walkers;
        D = c.num_params;
        n = W / 2;
        table = table_for<T>(c.calc_id);
        if (!table) return fail(MCMCPP_HIP_E_ARG, "calc_id %d has no kernels for this element type", c.calc_id);
        if (table->abi != kLaunchTableAbi || table->elem_size != sizeof(T))
            return fail(MCMCPP_HIP_E_ARG, "calc_id %d: the plug-in was built against other headers (table abi %08x)", c.calc_id, table->abi);

        // lane mapping: LPW lanes x EPL elements cover the walker's D-vector padded to a power of two
        const int base = Vec16<T>::N;
        const int n2 = pow2_at_least(D > base ? D : base);
        lpw = n2 / base < 64 ? n2 / base : 64;
        epl = n2 / lpw;
        const int lpw_log = ilog2(lpw), epl_shift = ilog2(epl / base);
        if (epl_shift >= kMaxEplShift || !table->half_step[lpw_log][epl_shift])
            return fail(MCMCPP_HIP_E_UNSUPPORTED, "no kernel for D=%d with this calculator (LPW=%d EPL=%d)", D, lpw, epl);
        half_fn = table->half_step[lpw_log][epl_shift];
        calc_fn = table->calc[lpw_log][epl_shift];
        vec_ok = (D % base == 0) ? 1 : 0;

        shard_begin = c.shard_begin;
        shard_count = c.shard_count > 0 ? c.shard_count : n;
        if (shard_begin < 0 || shard_begin + shard_count > n) return fail(MCMCPP_HIP_E_ARG, "shard out of range");

        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
            return fail(MCMCPP_HIP_E_NO_DEVICE, "no HIP device visible to this process");
        if (c.device >= ndev) return fail(MCMCPP_HIP_E_NO_DEVICE, "device %d out of range (%d visible)", c.device, ndev);
        if (c.device >= 0)
            device = c.device;
        else
            HIP_TRY(hipGetDevice(&device));
        HIP_TRY(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(MCMCPP_HIP_E_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device,
                        prop.gcnArchName);
        num_cus = prop.multiProcessorCount;

        // Walkers per wavefront: fill the chip first (about two wavefronts per SIMD), then up to 8 per wavefront (still
        // served by the draw wavefront) and 16 for the largest ensembles.  Measured, 32 dims fp64 (tools/sweep_passes.txt):
        // 65 536 walkers 5.2 / 5.8 / 4.4e9 walker-steps/s with 4 / 8 / 16 walkers per wavefront, 262 144: 7.4e9 with 8,
        // 1 M: 6.7 / 8.0 / 7.8 / 7.4e9 with 8 / 16 / 32 / 64.
        const int wpp = 64 / lpw;
        long forced = env_long("MCMCPP_HIP_PASSES", 0);
        if (forced > 0)
            passes = (int)forced;
        else
        {
            const long target_waves = (long)num_cus * 4 * env_long("MCMCPP_HIP_WAVES_PER_SIMD", 2);
            const int per_wave_cap = shard_count > 196608 ? 16 : 8;
            passes = 1;
            while (passes * 2 <= lpw && wpp * passes * 2 <= per_wave_cap && (long)shard_count / ((long)wpp * passes * 2) >= target_waves) passes *= 2;
        }
        if (passes < 1) passes = 1;
        if (passes > lpw) passes = lpw;
        // Matrix-core variants of the half-step kernel (dense calculators, fp64, even D in 18..32): the wavefront's
        // walkers are rows of one MFMA tile -- 8 walkers (2 passes) until the chip is full, 16 (4 passes) beyond.
        const long mc_min = env_long("MCMCPP_HIP_MATRIX_CORE_MIN_WALKERS", 0);
        if (table->half_step_mc[0][lpw_log][epl_shift] && (D % 2 == 0) && mc_min >= 0 && shard_count >= mc_min)
        {
            const int big = shard_count >= env_long("MCMCPP_HIP_MATRIX_CORE_4PASS_WALKERS", 32768) ? 1 : 0;
            half_fn = table->half_step_mc[big][lpw_log][epl_shift];
            passes = big ? 4 : 2;
        }

        // One launch per ensemble step (full_step_kernel.hpp) while the ensemble is small enough that a half-step
        // launch is bounded by its launch boundary and latencies rather than by HBM; needs the whole ensemble here.
        full_fn = nullptr;
        if (shard_count == n && shard_begin == 0 && env_long("MCMCPP_HIP_FULL_STEP", 1) != 0 &&
            W <= env_long("MCMCPP_HIP_FULL_STEP_MAX_WALKERS", 32768))
        {
            full_fn = table->full_step[lpw_log][epl_shift];
            full_wpb = kWavesPerBlock * (64 / lpw);
            if (table->full_step_mc[lpw_log][epl_shift] && (D % 2 == 0) && mc_min >= 0 && shard_count >= mc_min &&
                c.calc_id == MCMCPP_HIP_CALC_DENSE_GAUSSIAN)  // (it reads the padded matrix this file prepares)
            {
                full_fn = table->full_step_mc[lpw_log][epl_shift];
                full_wpb = kWavesPerBlock * 8;
            }
        }

        if (c.flags & MCMCPP_HIP_FLAG_CALLER_STREAM)
        {
            stream = (hipStream_t)c.hip_stream;  // may be the null (legacy default) stream
            own_stream = false;
            stream_valid = true;
        }
        else
        {
            HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            own_stream = true;
            stream_valid = true;
        }
        for (int k = 0; k < 4; ++k)
        {
            HIP_TRY(hipEventCreate(&ev_t0[k]));
            HIP_TRY(hipEventCreate(&ev_t1[k]));
        }


        {
            // upper bound of everything carved below (each piece rounded up to 256 bytes)
            const size_t graph_len = (size_t)(c.graph_steps == 0 ? default_graph_steps() : (c.graph_steps > 0 ? c.graph_steps : 1));
            const size_t waves_bound = (size_t)n + 64;  // no kernel uses more wavefronts per colour than walkers
            size_t need = 2 * sizeof(T) * (size_t)W * D                 // pos, pos_alt
                          + sizeof(T) * (size_t)W * 2 + sizeof(uint32_t) * (size_t)W + tables_total_bytes(n, true)
                          + sizeof(T) * ((size_t)(c.calc_params_len > 0 ? c.calc_params_len : 0) + 32 * 32)
                          + sizeof(uint32_t) * graph_len * 2 * waves_bound + 64 * 1024;
            HIP_TRY(hipMalloc(&arena, need));
            arena_bytes = need;
            arena_used = 0;
        }
        if (c.device_positions)
        {
            if (((uintptr_t)c.device_positions & 15u) != 0) return fail(MCMCPP_HIP_E_ARG, "device_positions must be 16-byte aligned");
            d_pos = (T*)c.device_positions;
            own_pos = false;
        }
        else
        {
            if (int rc = carve(&d_pos, sizeof(T) * (size_t)W * D)) return rc;
            own_pos = true;
        }
        // log-posteriors [2][W] (the second half is the full-step kernels' other buffer) and, right behind them, the
        // accepted counters [W]: one piece, so that kernels short of preloaded arguments can derive both addresses
        if (int rc = carve(&d_logp, sizeof(T) * (size_t)W * 2 + sizeof(uint32_t) * (size_t)W)) return rc;
        d_nacc = reinterpret_cast<uint32_t*>(d_logp + 2 * (size_t)W);
        if (full_fn)
            if (int rc = carve(&d_pos_alt, sizeof(T) * (size_t)W * D)) return rc;
        if (int rc = carve(&d_ctl, sizeof(StepCtl) * 2)) return rc;
        if (int rc = carve(&d_run, sizeof(RunInfo))) return rc;
        if (int rc = carve(&d_diag, sizeof(Diag))) return rc;
        // the draw records (two buffers: see HalfStepArgs::draws) and, right behind them, the jump tables: one piece
        // whose layout follows from n alone (JumpTables), so that kernels reach the tables from the record pointer
        static_assert(sizeof(DrawRec<T>) == 32, "the table offsets assume 32-byte records");
        have_task_table = (size_t)3 * n * sizeof(Affine128) <= ((size_t)env_long("MCMCPP_HIP_TASK_TABLE_MB", 16) << 20);
        {
            char* piece = nullptr;
            if (int rc = carve(&piece, tables_total_bytes(n, have_task_table))) return rc;
            d_draws = reinterpret_cast<DrawRec<T>*>(piece);
            d_task_jump = have_task_table ? reinterpret_cast<Affine128*>(piece + tables_offset_task(n)) : nullptr;
            d_jump_hi = reinterpret_cast<Affine128*>(piece + tables_offset_hi(n, have_task_table));
            d_jump_lo = reinterpret_cast<Affine128*>(piece + tables_offset_lo(n, have_task_table));
        }
        HIP_TRY(hipMemset(d_draws, 0, sizeof(DrawRec<T>) * (size_t)W * 2));
#ifdef MCMCPP_STAMPS
        HIP_TRY(hipMalloc(&d_stamps, kStampWords * sizeof(unsigned long long)));  // [8 stamps][2 alternating launches][start, end of 4096 workgroups | end of their draw wavefronts]
        HIP_TRY(hipMemset(d_stamps, 0, kStampWords * sizeof(unsigned long long)));
#endif
        HIP_TRY(hipMemset(d_nacc, 0, sizeof(uint32_t) * (size_t)W));
        HIP_TRY(hipMemset(d_diag, 0, sizeof(Diag)));
        HIP_TRY(hipMemset(d_run, 0, sizeof(RunInfo)));
        HIP_TRY(hipMemset(d_ctl, 0, sizeof(StepCtl) * 2));
        HIP_TRY(hipHostMalloc(&h_pinned, 512, hipHostMallocDefault));

        // calculator parameters (the dense Gaussian's matrix goes over transposed: see DenseGaussianFn)
        if (c.calc_params_len > 0)
        {
            std::vector<T> prm((const T*)c.calc_params, (const T*)c.calc_params + c.calc_params_len);
            if (c.calc_id == MCMCPP_HIP_CALC_DENSE_GAUSSIAN)
            {
                const T* p = (const T*)c.calc_params;
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j) prm[(size_t)j * D + i] = p[(size_t)i * D + j];
            }
            if (int rc = carve(&d_params, sizeof(T) * prm.size())) return rc;
            HIP_TRY(hipMemcpy(d_params, prm.data(), sizeof(T) * prm.size(), hipMemcpyHostToDevice));
            if (c.calc_id == MCMCPP_HIP_CALC_DENSE_GAUSSIAN && D <= 32)
            {
                // the matrix-core kernels read P^T zero-padded to 32 x 32 straight into registers
                std::vector<T> pad((size_t)32 * 32, (T)0);
                for (int k = 0; k < D; ++k)
                    for (int i = 0; i < D; ++i) pad[(size_t)k * 32 + i] = prm[(size_t)k * D + i];
                if (int rc = carve(&d_params_padded, sizeof(T) * pad.size())) return rc;
                HIP_TRY(hipMemcpy(d_params_padded, pad.data(), sizeof(T) * pad.size(), hipMemcpyHostToDevice));
            }
        }

        // pcg64 stream (MultiSampler.h:54) and its jump tables
        pcg_seed(c.seed, c.stream, &state0, &inc);
        {
            std::vector<Affine128> lo(256), hi((size_t)(n + 255) / 256);
            const Affine128 step3 = pcg_jump(inc, 3);
            lo[0].mult = make_u128(0, 1);
            lo[0].plus = make_u128(0, 0);
            for (int k = 1; k < 256; ++k) lo[k] = compose(step3, lo[k - 1]);
            const Affine128 step768 = pcg_jump(inc, 768);
            hi[0] = lo[0];
            for (size_t m = 1; m < hi.size(); ++m) hi[m] = compose(step768, hi[m - 1]);
            HIP_TRY(hipMemcpy(d_jump_lo, lo.data(), sizeof(Affine128) * lo.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(d_jump_hi, hi.data(), sizeof(Affine128) * hi.size(), hipMemcpyHostToDevice));
        }
        half_jump = pcg_jump(inc, (unsigned __int128)3 * (unsigned)n);
        if (have_task_table)
        {
            std::vector<Affine128> tj((size_t)3 * n);
            const Affine128 step1 = pcg_jump(inc, 1);
            tj[0] = step1;
            for (size_t t = 1; t < tj.size(); ++t) tj[t] = compose(step1, tj[t - 1]);
            HIP_TRY(hipMemcpy(d_task_jump, tj.data(), sizeof(Affine128) * tj.size(), hipMemcpyHostToDevice));
        }

        graph_steps = c.graph_steps == 0 ? (int)default_graph_steps() : c.graph_steps;
        partial_slots = graph_steps >= 1 ? graph_steps : 1;
        partial_waves = (int)(full_fn ? full_grid_blocks() : grid_blocks()) * kWavesPerBlock;
        if ((int)grid_blocks() * kWavesPerBlock > partial_waves) partial_waves = (int)grid_blocks() * kWavesPerBlock;
        if (int rc = carve(&d_partials, sizeof(uint32_t) * (size_t)partial_slots * 2 * (size_t)partial_waves)) return rc;
        HIP_TRY(hipMemset(d_partials, 0, sizeof(uint32_t) * (size_t)partial_slots * 2 * (size_t)partial_waves));
        chain_subchunk_bytes = (size_t)env_long("MCMCPP_HIP_CHAIN_SUBCHUNK_MB", 32) << 20;
        return MCMCPP_HIP_OK;
    }

    // Ensemble steps per hipGraph replay.  A boundary between two replays costs a few microseconds of the launch
    // sequence: ensembles small enough for one launch per step (5.6 us each) take 300 per replay -- with the bench's
    // slicing interval of 100 that is three stored steps per replay, as many as the forwarding ring allows; measured 1.5 %
    // over 128 -- larger ones, whose launches are long and whose per-step counters grow with the walker count, 128.
    long default_graph_steps() const { return env_long("MCMCPP_HIP_GRAPH_STEPS", W <= 32768 ? 300 : 128); }

    // Everything a step launch touches lives in ONE device allocation, carved here (one allocation, one free; tried as
    // a way to make the cold first accesses of a launch cheaper through fewer address translations: no measurable
    // difference, 6.05 us per launch either way).
    template <class P>
    int carve(P** out, size_t bytes)
    {
        const size_t off = (arena_used + 255) & ~(size_t)255;
        if (off + bytes > arena_bytes) return fail(MCMCPP_HIP_E_NOMEM, "internal: device arena too small (%zu + %zu > %zu)", off, bytes, arena_bytes);
        *out = reinterpret_cast<P*>(static_cast<char*>(arena) + off);
        arena_used = off + bytes;
        return MCMCPP_HIP_OK;
    }

    int set_state(const void* pos, const void* logp) override
    {
        if (!pos || !logp) return fail(MCMCPP_HIP_E_ARG, "set_state: null pointer");
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipMemcpyAsync(d_pos, pos, sizeof(T) * (size_t)W * D, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(d_logp, logp, sizeof(T) * (size_t)W, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(d_nacc, 0, sizeof(uint32_t) * (size_t)W, stream));
        HIP_TRY(hipMemsetAsync(d_diag, 0, sizeof(Diag), stream));
        half_steps = 0;
        steps_since_reset = 0;
        records_valid = false;
        int rc = write_ctl(0);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(stream));
        have_state = true;
        return MCMCPP_HIP_OK;
    }

    // EnsembleSampler::runMCMC.  Stored steps stream out while the sampler keeps stepping: a run is cut into
    // sub-chunks of stored steps; after the launches of sub-chunk c the same stream copies its device chain
    // half into pinned staging, and while the GPU works on sub-chunk c+1 the host thread copies sub-chunk c's
    // staging into the caller's (pageable) memory.  One stream on purpose: with a second active stream every
    // half-step launch of this latency-bound kernel was measured 1.5 us slower (5.8 -> 7.3 us).
    // Everything is ordered by events; nothing is allocated on the way.
    int run(int64_t n_saved, int32_t interval, void* chain_out, uint32_t* accepted_per_step) override
    {
        if (!have_state) return fail(MCMCPP_HIP_E_STATE, "run: set_state has not been called");
        if (n_saved < 0 || interval < 1) return fail(MCMCPP_HIP_E_ARG, "run: n_saved >= 0 and interval >= 1 required");
        if (shard_count != n) return fail(MCMCPP_HIP_E_UNSUPPORTED, "run: a sharded handle is driven with half_step_async");
        if (half_steps & 1) return fail(MCMCPP_HIP_E_STATE, "run: an ensemble step is half done (half_step_async)");
        HIP_TRY(hipSetDevice(device));
        const int64_t total = n_saved * (int64_t)interval;
        last_ms = 0.0;
        last_launches = 0;
        if (total == 0) return MCMCPP_HIP_OK;
        const bool dbg = env_long("MCMCPP_HIP_DEBUG_TIMING", 0) != 0;
        const auto tp0 = std::chrono::steady_clock::now();

        const size_t step_bytes = sizeof(T) * (size_t)W * D;
        int64_t sub_saved = n_saved;  // stored steps per sub-chunk
        if (chain_out)
        {
            sub_saved = (int64_t)(chain_subchunk_bytes / step_bytes);
            const int64_t eighth = (n_saved + 7) / 8;  // keep the last (un-overlappable) host copy short
            if (sub_saved > eighth) sub_saved = eighth;
            if (sub_saved < 1) sub_saved = 1;
        }
        // Full-step kernels forward stored steps to pinned host memory themselves (trickle_stored_step): a ring of
        // `ring` slots on the device with a twin in pinned host memory, no copy engine, no gap in the launch sequence.
        const bool trickle = full_fn && chain_out && step_bytes % 16 == 0 && env_long("MCMCPP_HIP_TRICKLE", 1) != 0;
        int64_t ring = 0, chunk_steps = 0;
        if (trickle)
        {
            ring = 4;
            while (ring < 64 && (size_t)(2 * ring) * step_bytes <= 2 * chain_subchunk_bytes) ring *= 2;
            // the host enqueues one chunk ahead of the one it waits for: stored steps of two chunks are in flight
            int64_t per_chunk = (graph_steps > 0 ? graph_steps : 64) / (int64_t)interval;
            if (per_chunk > (ring - 2) / 2) per_chunk = (ring - 2) / 2;
            if (per_chunk < 1) per_chunk = 1;
            chunk_steps = per_chunk * interval;
        }
        int rc = ensure_run_buffers(accepted_per_step ? (size_t)total : 0, (chain_out && !trickle) ? step_bytes * (size_t)sub_saved : 0,
                                    trickle ? step_bytes * (size_t)ring : 0);
        if (rc) return rc;
        if (accepted_per_step) HIP_TRY(hipMemsetAsync(d_acc, 0, sizeof(uint32_t) * (size_t)total, stream));
        rc = write_ctl(0);  // step_in_run = 0, stream position from the host-side half-step count
        if (rc) return rc;
        records_valid = false;  // (until this call has finished: an error on the way leaves them unknown)
        run_info_idle = false;
        if (full_fn)
        {
            hipLaunchKernelGGL(mark_rows_moved_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, stream, d_nacc, W, kRowMovedBit);
            HIP_TRY(hipGetLastError());
        }
        enq_step = half_steps >> 1;
        run_step = 0;
        args_red = make_args(0);
        args_blk = make_args(1);
        rc = ensure_graphs();
        if (rc) return rc;

        const auto tp1 = std::chrono::steady_clock::now();
        double launch_ms = 0.0;  // GPU time of the step launches alone (downloads excluded)
        if (trickle)
        {
            rc = run_trickle(n_saved, interval, (char*)chain_out, accepted_per_step != nullptr, step_bytes, ring, chunk_steps, &launch_ms);
            sub_saved = n_saved + 1;  // (the sub-chunk loop below has nothing to do)
        }
        const int64_t n_sub = trickle ? 0 : (n_saved + sub_saved - 1) / sub_saved;
        int64_t pending_first = -1, pending_count = 0;  // sub-chunk whose staging still has to reach chain_out
        int pending_buf = 0;
        for (int64_t c = 0; c < n_sub && rc == MCMCPP_HIP_OK; ++c)
        {
            const int buf = (int)(c & 1);
            const int64_t first = c * sub_saved;
            const int64_t now = (n_saved - first < sub_saved) ? n_saved - first : sub_saved;
            RunInfo* ri = reinterpret_cast<RunInfo*>((char*)h_pinned + 256 + 64 * (c % 4));
            static_assert(sizeof(RunInfo) <= 64, "the pinned upload slots are 64 bytes apart");
            ri->chain = chain_out ? d_chain[buf] : nullptr;
            ri->accepted_per_step = accepted_per_step ? d_acc : nullptr;
            ri->interval = interval;
            ri->chain_slot_base = -first;
            ri->stage = nullptr;
            ri->slot_mask = -1;
            ri->slice_bytes = 0;
            ri->step_bytes = (int64_t)step_bytes;
            HIP_TRY(hipMemcpyAsync(d_run, ri, sizeof(RunInfo), hipMemcpyHostToDevice, stream));
            // the events of slot c%4 were last used by sub-chunk c-4, which has long been waited for
            HIP_TRY(hipEventRecord(ev_t0[c & 3], stream));
            rc = enqueue_steps(now * interval);
            if (rc) break;
            HIP_TRY(hipEventRecord(ev_t1[c & 3], stream));
            if (c >= 3)
            {
                float ms = 0.f;
                HIP_TRY(hipEventSynchronize(ev_t1[(c - 3) & 3]));
                HIP_TRY(hipEventElapsedTime(&ms, ev_t0[(c - 3) & 3], ev_t1[(c - 3) & 3]));
                launch_ms += ms;
            }
            if (chain_out)
            {
                // the staging buffer is free: its previous content (sub-chunk c-2) was copied out below
                if (copy_stream)
                {
                    // the download runs beside the next sub-chunk's launches (which fill the other device half)
                    HIP_TRY(hipEventRecord(ev_filled[buf], stream));
                    HIP_TRY(hipStreamWaitEvent(copy_stream, ev_filled[buf], 0));
                    HIP_TRY(hipMemcpyAsync(h_stage[buf], d_chain[buf], step_bytes * (size_t)now, hipMemcpyDeviceToHost, copy_stream));
                    HIP_TRY(hipEventRecord(ev_copied[buf], copy_stream));
                }
                else
                {
                    HIP_TRY(hipMemcpyAsync(h_stage[buf], d_chain[buf], step_bytes * (size_t)now, hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipEventRecord(ev_copied[buf], stream));
                }
                if (pending_first >= 0)
                {
                    HIP_TRY(hipEventSynchronize(ev_copied[pending_buf]));
                    std::memcpy((char*)chain_out + step_bytes * (size_t)pending_first, h_stage[pending_buf], step_bytes * (size_t)pending_count);
                }
                pending_first = first;
                pending_count = now;
                pending_buf = buf;
            }
        }
        const auto tp2 = std::chrono::steady_clock::now();
        if (rc == MCMCPP_HIP_OK)
        {
            if (pending_first >= 0)
            {
                HIP_TRY(hipEventSynchronize(ev_copied[pending_buf]));
                std::memcpy((char*)chain_out + step_bytes * (size_t)pending_first, h_stage[pending_buf], step_bytes * (size_t)pending_count);
            }
            if (full_fn && (run_step & 1))
            {
                // an odd number of full steps leaves the ensemble in the second buffer: bring it (and the control
                // record) home, so that everything outside run() only ever knows the first
                HIP_TRY(hipMemcpyAsync(d_pos, d_pos_alt, sizeof(T) * (size_t)W * D, hipMemcpyDeviceToDevice, stream));
                HIP_TRY(hipMemcpyAsync(d_logp, d_logp + W, sizeof(T) * (size_t)W, hipMemcpyDeviceToDevice, stream));
                HIP_TRY(hipMemcpyAsync(d_ctl, d_ctl + 1, sizeof(StepCtl), hipMemcpyDeviceToDevice, stream));
            }
            HIP_TRY(hipStreamSynchronize(stream));
            for (int64_t c = (n_sub > 3 ? n_sub - 3 : 0); c < n_sub; ++c)
            {
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, ev_t0[c & 3], ev_t1[c & 3]));
                launch_ms += ms;
            }
            last_ms = launch_ms;
            last_launches = full_fn ? total : 2 * total;
            half_steps += 2 * (uint64_t)total;
            steps_since_reset += (uint64_t)total;
            // the last launch left the records of the next ensemble step behind (full-step launches: with partner2)
            records_valid = true;
            records_step = half_steps >> 1;
            records_partner2 = full_fn != nullptr;
            if (accepted_per_step)
                HIP_TRY(hipMemcpy(accepted_per_step, d_acc, sizeof(uint32_t) * (size_t)total, hipMemcpyDeviceToHost));
        }
        const auto tp3 = std::chrono::steady_clock::now();
        // (the device-side RunInfo still points to run-scoped buffers; half_step_async replaces it before it launches)
        if (dbg)
        {
            const auto tp4 = std::chrono::steady_clock::now();
            auto us = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) {
                return std::chrono::duration<double, std::micro>(y - x).count();
            };
            std::fprintf(stderr, "[mcmcpp_hip] run: setup %.0f us, enqueue %.0f us, drain %.0f us, idle-info %.0f us, gpu launches %.0f us\n",
                         us(tp0, tp1), us(tp1, tp2), us(tp2, tp3), us(tp3, tp4), last_ms * 1e3);
        }
        return rc;
    }

    // The chain path of the full-step kernels.  Stored step k is complete in the pinned ring when ensemble step
    // (k + 2) * interval - 1 has finished (every launch forwards 1/interval of the previous stored step), and its
    // ring slot is overwritten from step (k + ring + 1) * interval on: the host enqueues chunks of steps, stays one
    // chunk ahead of the one it waits for, copies out whatever has become complete and never lets the launches
    // run into a slot it has not copied yet.  The run's last stored step has no launches behind it: it is copied
    // from the device ring at the end.
    int run_trickle(int64_t n_saved, int32_t interval, char* chain_out, bool want_accepted, size_t step_bytes, int64_t ring,
                    int64_t chunk_steps, double* launch_ms)
    {
        const int64_t total = n_saved * (int64_t)interval;
        RunInfo* ri = reinterpret_cast<RunInfo*>((char*)h_pinned + 256);
        ri->chain = d_ring;
        ri->accepted_per_step = want_accepted ? d_acc : nullptr;
        ri->interval = interval;
        ri->chain_slot_base = 0;
        ri->stage = h_ring;
        ri->slot_mask = ring - 1;
        ri->slice_bytes = (int64_t)(((step_bytes + (size_t)interval - 1) / (size_t)interval + 15) / 16 * 16);
        ri->step_bytes = (int64_t)step_bytes;
        HIP_TRY(hipMemcpyAsync(d_run, ri, sizeof(RunInfo), hipMemcpyHostToDevice, stream));

        int64_t enq = 0, copied = 0;       // ensemble steps enqueued; stored steps handed to the caller
        int64_t chunk_end[4] = {0, 0, 0, 0};
        int64_t next_chunk = 0, oldest = 0;  // chunks enqueued / chunks whose completion has been processed
        auto process_oldest = [&]() -> int {
            const int e = (int)(oldest & 3);
            float ms = 0.f;
            HIP_TRY(hipEventSynchronize(ev_t1[e]));
            HIP_TRY(hipEventElapsedTime(&ms, ev_t0[e], ev_t1[e]));
            *launch_ms += ms;
            const int64_t complete = chunk_end[e] / interval - 1;  // stored steps fully forwarded by now
            for (; copied < complete; ++copied)
            {
                char* dst = chain_out + step_bytes * (size_t)copied;
                const char* src = (char*)h_ring + step_bytes * (size_t)(copied & (ring - 1));
                if (enq == total)
                    parallel_memcpy(dst, src, step_bytes);  // nothing left to overlap with: be quick
                else
                    std::memcpy(dst, src, step_bytes);
            }
            ++oldest;
            return MCMCPP_HIP_OK;
        };
        while (enq < total)
        {
            const int64_t now = (total - enq < chunk_steps) ? total - enq : chunk_steps;
            // at most two chunks in flight, and no launch may forward into a ring slot that is still to be copied out
            while (next_chunk > oldest && (next_chunk - oldest >= 2 || enq + now > (copied + ring + 1) * (int64_t)interval))
            {
                const int rc = process_oldest();
                if (rc) return rc;
            }
            const int e = (int)(next_chunk & 3);
            HIP_TRY(hipEventRecord(ev_t0[e], stream));
            const int rc = enqueue_steps(now);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ev_t1[e], stream));
            enq += now;
            chunk_end[e] = enq;
            ++next_chunk;
        }
        // What the launches do not forward: the run's last stored step.  Its download is queued now, behind the last
        // launch, so that it runs while the host still copies out the steps before it.
        {
            const size_t off = step_bytes * (size_t)((n_saved - 1) & (ring - 1));
            HIP_TRY(hipMemcpyAsync((char*)h_ring + off, (char*)d_ring + off, step_bytes, hipMemcpyDeviceToHost, stream));
        }
        while (next_chunk > oldest)
        {
            const int rc = process_oldest();
            if (rc) return rc;
        }
        HIP_TRY(hipStreamSynchronize(stream));
        for (; copied < n_saved; ++copied)  // (exactly one: every earlier one has been forwarded and copied above)
        {
            const size_t off = step_bytes * (size_t)(copied & (ring - 1));
            parallel_memcpy(chain_out + step_bytes * (size_t)copied, (char*)h_ring + off, step_bytes);
        }
        return MCMCPP_HIP_OK;
    }

    int get_state(void* pos, void* logp, uint32_t* n_accept) override
    {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamSynchronize(stream));
        if (pos) HIP_TRY(hipMemcpy(pos, d_pos, sizeof(T) * (size_t)W * D, hipMemcpyDeviceToHost));
        if (logp) HIP_TRY(hipMemcpy(logp, d_logp, sizeof(T) * (size_t)W, hipMemcpyDeviceToHost));
        if (n_accept)
        {
            HIP_TRY(hipMemcpy(n_accept, d_nacc, sizeof(uint32_t) * (size_t)W, hipMemcpyDeviceToHost));
            for (int w = 0; w < W; ++w) n_accept[w] &= ~kRowMovedBit;  // (the top bit is the full-step kernels' bookkeeping)
        }
        return MCMCPP_HIP_OK;
    }

    int seek(uint64_t steps_done) override
    {
        if (!have_state) return fail(MCMCPP_HIP_E_STATE, "seek: set_state has not been called");
        if (steps_done > (~0ULL >> 2)) return fail(MCMCPP_HIP_E_ARG, "seek: step count out of range");
        HIP_TRY(hipSetDevice(device));
        half_steps = 2 * steps_done;
        records_valid = false;
        return write_ctl(0);  // repositions the stream and re-primes the draw records of the next two half-steps
    }

    int reset_counters() override
    {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipMemsetAsync(d_nacc, 0, sizeof(uint32_t) * (size_t)W, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        steps_since_reset = 0;
        return MCMCPP_HIP_OK;
    }

    int get_counters(uint64_t* accepted, uint64_t* steps, uint64_t* ties, uint64_t* redraws) override
    {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamSynchronize(stream));
        if (accepted)
        {
            std::vector<uint32_t> a((size_t)W);
            HIP_TRY(hipMemcpy(a.data(), d_nacc, sizeof(uint32_t) * (size_t)W, hipMemcpyDeviceToHost));
            uint64_t s = 0;
            for (int c = 0; c < 2; ++c)
                for (int i = 0; i < shard_count; ++i) s += a[(size_t)c * n + shard_begin + i] & ~kRowMovedBit;
            *accepted = s;
        }
        if (steps) *steps = steps_since_reset;
        if (ties || redraws)
        {
            Diag d;
            HIP_TRY(hipMemcpy(&d, d_diag, sizeof(Diag), hipMemcpyDeviceToHost));
            if (ties) *ties = d.near_ties;
            if (redraws) *redraws = d.redraws;
        }
        return MCMCPP_HIP_OK;
    }

    int calc_logp(const void* pos, int64_t count, void* out) override
    {
        if (count < 0 || (count > 0 && (!pos || !out))) return fail(MCMCPP_HIP_E_ARG, "calc_logp: bad arguments");
        if (count == 0) return MCMCPP_HIP_OK;
        HIP_TRY(hipSetDevice(device));
        struct Scratch  // freed on every way out
        {
            T *rows = nullptr, *out = nullptr;
            ~Scratch()
            {
                if (rows) (void)hipFree(rows);
                if (out) (void)hipFree(out);
            }
        } scratch;
        HIP_TRY(hipMalloc(&scratch.rows, sizeof(T) * (size_t)count * D));
        HIP_TRY(hipMalloc(&scratch.out, sizeof(T) * (size_t)count));
        T *dp = scratch.rows, *dout = scratch.out;
        HIP_TRY(hipMemcpyAsync(dp, pos, sizeof(T) * (size_t)count * D, hipMemcpyHostToDevice, stream));
        const long long per_block = (long long)(64 / lpw) * kWavesPerBlock;
        const unsigned grid = (unsigned)((count + per_block - 1) / per_block);
        calc_fn(dp, dout, d_params, count, D, vec_ok, grid, stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out, dout, sizeof(T) * (size_t)count, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return MCMCPP_HIP_OK;
    }

    int last_run_timing(double* ms, int64_t* launches) override
    {
        if (ms) *ms = last_ms;
        if (launches) *launches = last_launches;
        return MCMCPP_HIP_OK;
    }

    int half_step_async(int32_t color, int64_t save_slot) override
    {
        if (!have_state) return fail(MCMCPP_HIP_E_STATE, "half_step_async: set_state has not been called");
        if (color != (int)(half_steps & 1)) return fail(MCMCPP_HIP_E_ARG, "half_step_async: colour %d out of order", color);
        if (save_slot >= 0 && (!bound_chain || save_slot >= bound_slots))
            return fail(MCMCPP_HIP_E_ARG, "half_step_async: save_slot outside the bound device chain");
        HIP_TRY(hipSetDevice(device));
        if (!run_info_idle)
        {
            const int rc = upload_idle_run_info();
            if (rc) return rc;
        }
        HalfStepArgs<T> a = make_args(color, (int)((half_steps >> 1) & 1));
        a.use_ctl_save = 0;
        a.partials = nullptr;
        a.direct_save_slot = save_slot;
        half_fn(a, grid_blocks(), stream);
        HIP_TRY(hipGetLastError());
        half_steps += 1;
        if (color == 1) steps_since_reset += 1;
        // the two launches of a step leave the next step's records of this handle's shard behind (without partner2)
        records_valid = color == 1 && shard_count == n;
        records_step = half_steps >> 1;
        records_partner2 = false;
        return MCMCPP_HIP_OK;
    }

    int bind_device_chain(void* chain, int64_t slots) override
    {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamSynchronize(stream));
        bound_chain = chain;
        bound_slots = chain ? slots : 0;
        return upload_idle_run_info();
    }

    void* device_positions() override { return d_pos; }

    int shard_span(int32_t color, int64_t* off, int64_t* cnt) override
    {
        if (color != 0 && color != 1) return fail(MCMCPP_HIP_E_ARG, "shard_span: colour must be 0 or 1");
        if (off) *off = ((int64_t)(color ? n : 0) + shard_begin) * D;
        if (cnt) *cnt = (int64_t)shard_count * D;
        return MCMCPP_HIP_OK;
    }

    int debug_stamps(unsigned long long* out8) override
    {
        if (!d_stamps) return fail(MCMCPP_HIP_E_UNSUPPORTED, "not a diagnostic build");
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(out8, d_stamps, kStampWords * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        return MCMCPP_HIP_OK;
    }

    int synchronize() override
    {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamSynchronize(stream));
        return MCMCPP_HIP_OK;
    }

private:
    int hip_rc(hipError_t e, const char* what)
    {
        if (e == hipSuccess) return MCMCPP_HIP_OK;
        return fail(MCMCPP_HIP_E_HIP, "%s failed: %s", what, hipGetErrorString(e));
    }

    // RunInfo used outside run(): the chain bound for half_step_async (if any), no per-step counters
    int upload_idle_run_info()
    {
        RunInfo ri;
        ri.chain = bound_chain;
        ri.accepted_per_step = nullptr;
        ri.interval = 1;
        ri.chain_slot_base = 0;
        ri.stage = nullptr;
        ri.slot_mask = -1;
        ri.slice_bytes = 0;
        ri.step_bytes = 0;
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(d_run, &ri, sizeof ri, hipMemcpyHostToDevice));
        run_info_idle = true;
        return MCMCPP_HIP_OK;
    }

    unsigned grid_blocks() const
    {
        const long per_wave = (long)(64 / lpw) * passes;
        const long waves = (shard_count + per_wave - 1) / per_wave;
        return (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    }

    unsigned full_grid_blocks() const { return (unsigned)((n + full_wpb - 1) / full_wpb); }

    HalfStepArgs<T> make_args(int color, int parity) const
    {
        HalfStepArgs<T> a = make_args(color);
        a.draw_parity = parity;
        return a;
    }

    HalfStepArgs<T> make_args(int color) const
    {
        HalfStepArgs<T> a;
        std::memset(&a, 0, sizeof a);
        a.pos = d_pos;
        a.logp = d_logp;
        a.n_accept = d_nacc;
        a.ctl_in = d_ctl + color;
        a.ctl_out = d_ctl + (1 - color);
        a.run = d_run;
        a.diag = d_diag;
        a.jump_lo = d_jump_lo;
        a.jump_hi = d_jump_hi;
        a.task_jump = d_task_jump;
        a.calc_params = d_params;
        a.draws = d_draws;
        a.half_jump = half_jump;
        for (int k = 0; k < 3; ++k) a.draw_jump[k] = pcg_jump(inc, (unsigned)k + 1);
        a.inc = inc;
        a.redraw_threshold = (uint64_t)(0 - (uint64_t)n) % (uint64_t)n;
        // GwDistribution<T,2,1> (MCMCpp/Utility/GwDistribution.h:45-55)
        const T alpha = (T)(cfg.gw_alpha_num > 0 ? cfg.gw_alpha_num : 2) / (T)(cfg.gw_alpha_den > 0 ? cfg.gw_alpha_den : 1);
        const T sqrt_a = std::sqrt(alpha);
        const T inv_sqrt_a = (T)1 / sqrt_a;
        a.gw_term1 = sqrt_a - inv_sqrt_a;
        a.gw_inv_sqrt = inv_sqrt_a;
        a.dims_minus_one = (T)(D - 1);
        a.tie_eps = sizeof(T) == 8 ? (T)1e-12 : (T)6e-7;
        a.n = n;
        a.dims = D;
        a.color = color;
        a.shard_begin = shard_begin;
        a.shard_count = shard_count;
        a.passes = passes;
        a.vec_ok = vec_ok;
        a.n_is_pow2 = (n & (n - 1)) == 0;
        a.partials = d_partials;
        a.partial_slots = partial_slots;
        a.partial_waves = partial_waves;
        a.direct_save_slot = -1;
        a.use_ctl_save = 1;
        a.stamps = d_stamps;
        a.draw_parity = 0;
        a.pos_alt = d_pos_alt;
        a.logp_alt = d_logp + W;
        a.pos_parity = 0;
        a.calc_params_padded = d_params_padded;
        // a fifth wavefront per workgroup computes the next draws when that is at most two rounds of 64 draws
        a.draw_wave = (3 * kWavesPerBlock * (64 / lpw) * passes <= 128 && env_long("MCMCPP_HIP_NO_DRAW_WAVE", 0) == 0) ? 1 : 0;
        return a;
    }

    // device StepCtl[0] <- {stream position of half-step `half_steps`, counters}; half_steps must be even
    int write_ctl(uint64_t step_in_run)
    {
        StepCtl* c = (StepCtl*)((char*)h_pinned + 128);
        const Affine128 j = pcg_jump(inc, (unsigned __int128)3 * (unsigned)n * (unsigned __int128)half_steps);
        c->state = apply(j, state0);
        const U128 state1 = apply(half_jump, c->state);
        c->state2 = apply(half_jump, state1);
        c->half_step = half_steps;
        c->step_in_run = step_in_run;
        c->chain_slot = 0;
        c->save_phase = 0;
        c->partial_slot = 0;
        HIP_TRY(hipMemcpyAsync(d_ctl + (half_steps & 1), c, sizeof(StepCtl), hipMemcpyHostToDevice, stream));
        // the draw records of the next red and the next black half-step (afterwards the launches keep them going):
        // unless the launches of the previous call left exactly these behind
        if (!(records_valid && records_step == (half_steps >> 1) && (!full_fn || records_partner2)))
        {
            const int parity = (int)((half_steps >> 1) & 1);  // the buffer the coming ensemble step reads
            launch_fill_draws(make_args(0, parity), c->state, nullptr, stream);
            launch_fill_draws(make_args(1, parity), state1, full_fn ? &c->state : nullptr, stream);
            HIP_TRY(hipGetLastError());
            records_valid = true;
            records_step = half_steps >> 1;
            records_partner2 = full_fn != nullptr;
        }
        HIP_TRY(hipStreamSynchronize(stream));
        return MCMCPP_HIP_OK;
    }

    // pos_parity: which position buffer a full-step launch reads (ensemble steps enqueued in this run() & 1)
    void enqueue_step(int parity, int pos_parity)
    {
        if (full_fn)
        {
            HalfStepArgs<T>& a = args_red;
            a.draw_parity = parity;
            a.pos_parity = pos_parity;
            a.ctl_in = d_ctl + pos_parity;
            a.ctl_out = d_ctl + (1 - pos_parity);
            a.partial_waves = partial_waves;
            // the next draws by four extra wavefronts (two per colour) when that is one round of 64 draws each
            a.draw_wave = (3 * (kFullDrawWaves == 4 ? (full_wpb + 1) / 2 : full_wpb) <= 64 && env_long("MCMCPP_HIP_NO_DRAW_WAVE", 0) == 0) ? 1 : 0;
            full_fn(a, full_grid_blocks(), stream);
            return;
        }
        args_red.draw_parity = parity;
        args_blk.draw_parity = parity;
        half_fn(args_red, grid_blocks(), stream);
        half_fn(args_blk, grid_blocks(), stream);
    }

    // hipGraph of `steps` ensemble steps followed by the accepted-count reduction (cached per step count:
    // graph_steps for the bulk, one graph per distinct remainder)
    // (the record-buffer parity of every node is frozen into the graph, hence one graph per starting parity)
    int graph_for(int steps, int start_parity, int pos_parity, hipGraphExec_t* out)
    {
        const size_t key = (size_t)steps * 4 + (size_t)start_parity * 2 + (size_t)pos_parity;
        if (graph_cache.size() <= key) graph_cache.resize(key + 1, nullptr);
        if (!graph_cache[key])
        {
            hipGraph_t g = nullptr;
            HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
            for (int s = 0; s < steps; ++s) enqueue_step((start_parity + s) & 1, (pos_parity + s) & 1);
            launch_accepted_reduce(d_partials, partial_slots, partial_waves, steps, ctl_after(pos_parity + steps), d_run, stream);
            HIP_TRY(hipStreamEndCapture(stream, &g));
            hipGraphExec_t ex = nullptr;
            HIP_TRY(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
            HIP_TRY(hipGraphDestroy(g));
            graph_cache[key] = ex;
        }
        *out = graph_cache[key];
        return MCMCPP_HIP_OK;
    }

    int ensure_graphs()
    {
        if (graph_steps < 1) return MCMCPP_HIP_OK;
        hipGraphExec_t ex;
        return graph_for(graph_steps, (int)(enq_step & 1), 0, &ex);
    }

    // the control record the last launch of a step sequence leaves behind: the half-step pair always ends in [0],
    // full-step launches alternate with the position buffers
    const StepCtl* ctl_after(int64_t pos_parity_after) const { return d_ctl + (full_fn ? (pos_parity_after & 1) : 0); }

    // enqueue `steps` ensemble steps on the launch stream (graph replays, or plain launches when graphs are off)
    int enqueue_steps(int64_t steps)
    {
        // enq_step: ensemble steps enqueued since set_state/seek (its low bit selects the record buffer)
        int64_t left = steps;
        if (graph_steps >= 1)
        {
            hipGraphExec_t ex = nullptr;
            while (left >= graph_steps)
            {
                int rc = graph_for(graph_steps, (int)(enq_step & 1), (int)(run_step & 1), &ex);
                if (rc) return rc;
                HIP_TRY(hipGraphLaunch(ex, stream));
                left -= graph_steps;
                enq_step += (uint64_t)graph_steps;
                run_step += (uint64_t)graph_steps;
            }
            if (left > 0)
            {
                int rc = graph_for((int)left, (int)(enq_step & 1), (int)(run_step & 1), &ex);  // one replay for the remainder
                if (rc) return rc;
                HIP_TRY(hipGraphLaunch(ex, stream));
                enq_step += (uint64_t)left;
                run_step += (uint64_t)left;
            }
        }
        else
        {
            for (; left > 0; --left)
            {
                enqueue_step((int)(enq_step & 1), (int)(run_step & 1));
                launch_accepted_reduce(d_partials, partial_slots, partial_waves, 1, ctl_after((int64_t)run_step + 1), d_run, stream);
                enq_step += 1;
                run_step += 1;
            }
            HIP_TRY(hipGetLastError());
        }
        return MCMCPP_HIP_OK;
    }

    // persistent per-run buffers, grown on demand: per-step accepted counters, the two halves of the device
    // chain and their pinned staging twins
    int ensure_run_buffers(size_t acc_entries, size_t half_bytes, size_t ring_bytes)
    {
        if (ring_bytes > ring_capacity)
        {
            HIP_TRY(hipStreamSynchronize(stream));
            if (d_ring) hipFree(d_ring);
            if (h_ring) hipHostFree(h_ring);
            d_ring = nullptr;
            h_ring = nullptr;
            ring_capacity = 0;
            if (hipMalloc(&d_ring, ring_bytes) != hipSuccess) return fail(MCMCPP_HIP_E_NOMEM, "run: cannot allocate %zu bytes of device chain", ring_bytes);
            if (hipHostMalloc(&h_ring, ring_bytes, hipHostMallocDefault) != hipSuccess)
                return fail(MCMCPP_HIP_E_NOMEM, "run: cannot allocate %zu bytes of pinned staging", ring_bytes);
            ring_capacity = ring_bytes;
        }
        if (acc_entries > acc_capacity)
        {
            if (d_acc) hipFree(d_acc);
            d_acc = nullptr;
            acc_capacity = 0;
            if (hipMalloc(&d_acc, sizeof(uint32_t) * acc_entries) != hipSuccess)
                return fail(MCMCPP_HIP_E_NOMEM, "run: cannot allocate %zu accepted counters", acc_entries);
            acc_capacity = acc_entries;
        }
        if (half_bytes > 0 && ev_copied[0] == nullptr)
        {
            for (int k = 0; k < 2; ++k) HIP_TRY(hipEventCreateWithFlags(&ev_copied[k], hipEventDisableTiming));
            for (int k = 0; k < 2; ++k) HIP_TRY(hipEventCreateWithFlags(&ev_filled[k], hipEventDisableTiming));
            if (env_long("MCMCPP_HIP_COPY_STREAM", 0) != 0) HIP_TRY(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        }
        if (half_bytes > chain_half_capacity)
        {
            HIP_TRY(hipStreamSynchronize(stream));
            for (int k = 0; k < 2; ++k)
            {
                if (d_chain[k]) hipFree(d_chain[k]);
                if (h_stage[k]) hipHostFree(h_stage[k]);
                d_chain[k] = nullptr;
                h_stage[k] = nullptr;
            }
            chain_half_capacity = 0;
            for (int k = 0; k < 2; ++k)
            {
                void* dp = nullptr;
                if (hipMalloc(&dp, half_bytes) != hipSuccess)
                    return fail(MCMCPP_HIP_E_NOMEM, "run: cannot allocate %zu bytes of device chain", half_bytes);
                d_chain[k] = (T*)dp;
                if (hipHostMalloc(&h_stage[k], half_bytes, hipHostMallocDefault) != hipSuccess)
                    return fail(MCMCPP_HIP_E_NOMEM, "run: cannot allocate %zu bytes of pinned staging", half_bytes);
            }
            chain_half_capacity = half_bytes;
        }
        return MCMCPP_HIP_OK;
    }

    void release()
    {
        if (device >= 0) hipSetDevice(device);
        if (stream_valid) hipStreamSynchronize(stream);  // half_step_async work may still be in flight
        for (hipGraphExec_t ex : graph_cache)
            if (ex) hipGraphExecDestroy(ex);
        if (arena) hipFree(arena);  // positions, log-posteriors, counters, records, tables, parameters, partial counts
        if (d_acc) hipFree(d_acc);
        for (int k = 0; k < 2; ++k)
        {
            if (d_chain[k]) hipFree(d_chain[k]);
            if (h_stage[k]) hipHostFree(h_stage[k]);
            if (ev_copied[k]) hipEventDestroy(ev_copied[k]);
            if (ev_filled[k]) hipEventDestroy(ev_filled[k]);
        }
        if (d_ring) hipFree(d_ring);
        if (h_ring) hipHostFree(h_ring);

        if (h_pinned) hipHostFree(h_pinned);
        for (int k = 0; k < 4; ++k)
        {
            if (ev_t0[k]) hipEventDestroy(ev_t0[k]);
            if (ev_t1[k]) hipEventDestroy(ev_t1[k]);
        }
        if (copy_stream)
        {
            hipStreamSynchronize(copy_stream);
            hipStreamDestroy(copy_stream);
        }
        if (own_stream && stream) hipStreamDestroy(stream);
    }

    mcmcpp_hip_config cfg;
    const LaunchTable<T>* table = nullptr;
    typename LaunchTable<T>::HalfStepFn half_fn = nullptr;
    typename LaunchTable<T>::HalfStepFn full_fn = nullptr;  // non-null: run() steps with one launch per ensemble step
    int full_wpb = 1;                                       // walkers of each colour per full-step workgroup
    T* d_pos_alt = nullptr;
    uint64_t run_step = 0;                                  // ensemble steps enqueued in the current run()
    typename LaunchTable<T>::CalcFn calc_fn = nullptr;
    int W = 0, D = 0, n = 0, lpw = 1, epl = 1, passes = 1, vec_ok = 0, num_cus = 256;
    int shard_begin = 0, shard_count = 0, device = -1, graph_steps = 32;
    size_t chain_subchunk_bytes = 0, chain_half_capacity = 0, acc_capacity = 0;
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_filled[2] = {nullptr, nullptr};
    void* arena = nullptr;  // one device allocation holding everything a step launch touches (see carve)
    size_t arena_bytes = 0, arena_used = 0;
    void *d_ring = nullptr, *h_ring = nullptr;  // full-step chain path: device ring of stored steps and its pinned host twin
    size_t ring_capacity = 0;
    hipStream_t copy_stream = nullptr;  // experiment (MCMCPP_HIP_COPY_STREAM=1): chain downloads beside the launches
    T* d_chain[2] = {nullptr, nullptr};
    void* h_stage[2] = {nullptr, nullptr};
    uint32_t* d_acc = nullptr;
    hipStream_t stream = nullptr;
    bool own_stream = false, own_pos = false, have_state = false, stream_valid = false;
    hipEvent_t ev_t0[4] = {nullptr, nullptr, nullptr, nullptr}, ev_t1[4] = {nullptr, nullptr, nullptr, nullptr};
    T *d_pos = nullptr, *d_logp = nullptr, *d_params = nullptr, *d_params_padded = nullptr;
    uint32_t* d_nacc = nullptr;
    StepCtl* d_ctl = nullptr;
    RunInfo* d_run = nullptr;
    Diag* d_diag = nullptr;
    DrawRec<T>* d_draws = nullptr;
    static constexpr size_t kStampWords = 8 + 2 * 3 * 4096 + 8;
    unsigned long long* d_stamps = nullptr;  // diagnostic build only
    uint32_t* d_partials = nullptr;
    int partial_slots = 1, partial_waves = 0;
    Affine128 *d_jump_lo = nullptr, *d_jump_hi = nullptr, *d_task_jump = nullptr;  // behind d_draws: see JumpTables
    bool have_task_table = false;
    // which ensemble step the draw records on the device belong to, if known, and whether the black ones carry partner2
    bool records_valid = false, records_partner2 = false, run_info_idle = false;
    uint64_t records_step = 0;
    void* h_pinned = nullptr;
    U128 state0, inc;
    Affine128 half_jump;
    HalfStepArgs<T> args_red, args_blk;
    std::vector<hipGraphExec_t> graph_cache;  // [steps] -> instantiated graph
    uint64_t half_steps = 0, steps_since_reset = 0, enq_step = 0;
    double last_ms = 0.0;
    int64_t last_launches = 0;
    void* bound_chain = nullptr;
    int64_t bound_slots = 0;
};

int check_config(const mcmcpp_hip_config* c, std::string& err)
{
    char buf[256];
#define BAD(...)                              \
    do                                        \
    {                                         \
        snprintf(buf, sizeof buf, __VA_ARGS__); \
        err = buf;                            \
        return MCMCPP_HIP_E_ARG;              \
    } while (0)
    if (!c) BAD("config is NULL");
    if (c->struct_size != sizeof(mcmcpp_hip_config)) BAD("struct_size %u != %zu (ABI mismatch)", c->struct_size, sizeof(mcmcpp_hip_config));
    if (c->dtype != MCMCPP_HIP_F64 && c->dtype != MCMCPP_HIP_F32) BAD("dtype must be MCMCPP_HIP_F64 or MCMCPP_HIP_F32");
    if (c->num_params < 1 || c->num_params > 1024) BAD("num_params must be in 1..1024");
    // EnsembleSampler.h:207-208
    if (c->num_walkers < 2 || (c->num_walkers & 1)) BAD("num_walkers must be even");
    if (c->num_walkers <= 2 * c->num_params) BAD("num_walkers must exceed 2*num_params");
    switch (c->calc_id)
    {
    case MCMCPP_HIP_CALC_ISO_GAUSSIAN:
        if (c->calc_params_len != 0) BAD("IsoGaussian takes no parameters");
        break;
    case MCMCPP_HIP_CALC_DENSE_GAUSSIAN:
        if (!c->calc_params || c->calc_params_len != c->num_params * c->num_params) BAD("DenseGaussian needs D*D parameters");
        break;
    case MCMCPP_HIP_CALC_ROSENBROCK:
        if (!c->calc_params || c->calc_params_len != 3) BAD("Rosenbrock needs 3 parameters (a, b, c)");
        break;
    case MCMCPP_HIP_CALC_SKEWED_GAUSSIAN_2D:
        if (!c->calc_params || c->calc_params_len != 1 || c->num_params != 2) BAD("SkewedGaussian2D needs D == 2 and 1 parameter");
        break;
    default:
    {
        RegisteredCalc r;
        if (c->calc_id < MCMCPP_HIP_CALC_USER_BASE || !registered_calc(c->calc_id, &r)) BAD("unknown calc_id %d", c->calc_id);
        if (r.params_len >= 0 && c->calc_params_len != r.params_len) BAD("calculator %d takes %d parameters", c->calc_id, r.params_len);
        if (c->calc_params_len < 0 || (c->calc_params_len > 0 && !c->calc_params)) BAD("calc_params missing");
    }
    }
    if (c->shard_begin < 0 || c->shard_count < 0) BAD("negative shard bounds");
    if (c->mover != MCMCPP_HIP_MOVER_STRETCH && c->mover != MCMCPP_HIP_MOVER_DIFFERENTIAL_EVOLUTION) BAD("unknown mover %u", c->mover);
    if (c->mover == MCMCPP_HIP_MOVER_DIFFERENTIAL_EVOLUTION && (c->shard_count != 0 || c->device_positions))
        BAD("the differential-evolution mover runs one whole ensemble per handle (no shards, no caller-owned position buffer)");
    if (c->gw_alpha_num < 0 || c->gw_alpha_den < 0 || ((c->gw_alpha_num == 0) != (c->gw_alpha_den == 0)))
        BAD("gw_alpha_num/gw_alpha_den must both be positive (or both 0 for the default 2/1)");
    if (c->gw_alpha_num > 0 && c->gw_alpha_num <= c->gw_alpha_den) BAD("the stretch scale alpha must exceed 1");
#undef BAD
    return MCMCPP_HIP_OK;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------
extern "C"
{
int mcmcpp_hip_abi_version(void) { return MCMCPP_HIP_ABI_VERSION; }

#ifdef MCMCPP_STAMPS
// diagnostic build only: shader-clock stamps of the last half-step launch (not part of the ABI)
int mcmcpp_hip_debug_stamps(mcmcpp_hip_sampler* h, unsigned long long* out8)
{
    if (!h || !out8) return MCMCPP_HIP_E_ARG;
    return h->debug_stamps(out8);
}
#endif

int mcmcpp_hip_register_calculator(int32_t calc_id, const void* table_f64, const void* table_f32, int32_t params_len)
{
    if (calc_id < MCMCPP_HIP_CALC_USER_BASE || (!table_f64 && !table_f32) || params_len < -1)
    {
        g_create_error = "register_calculator: calc_id must be >= MCMCPP_HIP_CALC_USER_BASE with at least one table";
        return MCMCPP_HIP_E_ARG;
    }
    const uint32_t a64 = table_f64 ? static_cast<const LaunchTable<double>*>(table_f64)->abi : kLaunchTableAbi;
    const uint32_t a32 = table_f32 ? static_cast<const LaunchTable<float>*>(table_f32)->abi : kLaunchTableAbi;
    if (a64 != kLaunchTableAbi || a32 != kLaunchTableAbi)
    {
        g_create_error = "register_calculator: the plug-in was built against other headers";
        return MCMCPP_HIP_E_ARG;
    }
    RegisteredCalc r;
    r.f64 = table_f64;
    r.f32 = table_f32;
    r.params_len = params_len;
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    g_registry[calc_id] = r;
    return MCMCPP_HIP_OK;
}

int mcmcpp_hip_create(const mcmcpp_hip_config* cfg, mcmcpp_hip_sampler** out)
{
    if (!out)
    {
        g_create_error = "out is NULL";
        return MCMCPP_HIP_E_ARG;
    }
    *out = nullptr;
    int rc = check_config(cfg, g_create_error);
    if (rc) return rc;
    mcmcpp_hip_sampler* h = nullptr;
    int irc;
    if (cfg->mover == MCMCPP_HIP_MOVER_DIFFERENTIAL_EVOLUTION)
    {
        h = mcmcpp::make_de_sampler(*cfg, &irc);
        if (!h) return MCMCPP_HIP_E_NOMEM;
    }
    else if (cfg->dtype == MCMCPP_HIP_F64)
    {
        Sampler<double>* s = new (std::nothrow) Sampler<double>();
        if (!s) return MCMCPP_HIP_E_NOMEM;
        irc = s->init(*cfg);
        h = s;
    }
    else
    {
        Sampler<float>* s = new (std::nothrow) Sampler<float>();
        if (!s) return MCMCPP_HIP_E_NOMEM;
        irc = s->init(*cfg);
        h = s;
    }
    if (irc)
    {
        g_create_error = h->error;
        delete h;
        return irc;
    }
    *out = h;
    return MCMCPP_HIP_OK;
}

void mcmcpp_hip_destroy(mcmcpp_hip_sampler* h) { delete h; }

const char* mcmcpp_hip_last_error(const mcmcpp_hip_sampler* h) { return h ? h->error.c_str() : g_create_error.c_str(); }

#define NEED_H \
    if (!h) return MCMCPP_HIP_E_ARG

int mcmcpp_hip_set_state(mcmcpp_hip_sampler* h, const void* positions, const void* logp)
{
    NEED_H;
    return h->set_state(positions, logp);
}
int mcmcpp_hip_run(mcmcpp_hip_sampler* h, int64_t n_saved, int32_t interval, void* chain_out, uint32_t* accepted_per_step)
{
    NEED_H;
    return h->run(n_saved, interval, chain_out, accepted_per_step);
}
int mcmcpp_hip_get_state(mcmcpp_hip_sampler* h, void* positions, void* logp, uint32_t* n_accept)
{
    NEED_H;
    return h->get_state(positions, logp, n_accept);
}
int mcmcpp_hip_seek(mcmcpp_hip_sampler* h, uint64_t ensemble_steps_done)
{
    NEED_H;
    return h->seek(ensemble_steps_done);
}
int mcmcpp_hip_reset_counters(mcmcpp_hip_sampler* h)
{
    NEED_H;
    return h->reset_counters();
}
int mcmcpp_hip_get_counters(mcmcpp_hip_sampler* h, uint64_t* accepted, uint64_t* ensemble_steps, uint64_t* near_ties,
                            uint64_t* redraws)
{
    NEED_H;
    return h->get_counters(accepted, ensemble_steps, near_ties, redraws);
}
int mcmcpp_hip_calc_logp(mcmcpp_hip_sampler* h, const void* positions, int64_t count, void* logp_out)
{
    NEED_H;
    return h->calc_logp(positions, count, logp_out);
}
int mcmcpp_hip_last_run_timing(mcmcpp_hip_sampler* h, double* gpu_ms, int64_t* step_launches)
{
    NEED_H;
    return h->last_run_timing(gpu_ms, step_launches);
}
int mcmcpp_hip_half_step_async(mcmcpp_hip_sampler* h, int32_t color, int64_t save_slot)
{
    NEED_H;
    return h->half_step_async(color, save_slot);
}
int mcmcpp_hip_bind_device_chain(mcmcpp_hip_sampler* h, void* device_chain, int64_t slots)
{
    NEED_H;
    return h->bind_device_chain(device_chain, slots);
}
void* mcmcpp_hip_device_positions(mcmcpp_hip_sampler* h) { return h ? h->device_positions() : nullptr; }
int mcmcpp_hip_shard_span(mcmcpp_hip_sampler* h, int32_t color, int64_t* offset_elems, int64_t* count_elems)
{
    NEED_H;
    return h->shard_span(color, offset_elems, count_elems);
}
int mcmcpp_hip_synchronize(mcmcpp_hip_sampler* h)
{
    NEED_H;
    return h->synchronize();
}
}

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
#include "exchange_kernels.hpp"
#include "rccl_dyn.hpp"
#include "sampler_base.hpp"

using namespace mcmcpp;

#define NCCL_TRY(expr)                                                                                          \
    do                                                                                                          \
    {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                               \
        if (r_ != ncclSuccess) return fail(MCMCPP_HIP_E_COMM, "%s failed: %s", #expr, rccl->GetErrorString(r_)); \
    } while (0)

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

// The library's tuning and diagnostic knobs (environment variables, DESIGN.md section 9 lists them).  Read ONCE, when a
// handle is created; nothing on the launch path touches the environment.  Negative "unset" values mean "library default".
struct Knobs
{
    long passes;                  // MCMCPP_HIP_PASSES                   walkers-per-wavefront rounds of the half-step kernels (0: chosen from the size)
    long waves_per_simd;          // MCMCPP_HIP_WAVES_PER_SIMD           wavefronts per SIMD to reach before a wavefront takes more walkers (2)
    long matrix_core_min_walkers; // MCMCPP_HIP_MATRIX_CORE_MIN_WALKERS  smallest shard stepped by the matrix-core kernels (0; -1: never)
    long matrix_core_4pass;       // MCMCPP_HIP_MATRIX_CORE_4PASS_WALKERS from this many updates per launch on: 16 walkers per wavefront (18432)
    long matrix_core_late;        // MCMCPP_HIP_MATRIX_CORE_LATE_DRAWS    from this many updates per launch on: the 16-walker wavefronts make their next draws behind the accept, four to a SIMD (49152; -1: never)
    long full_step;               // MCMCPP_HIP_FULL_STEP                1: one launch per ensemble step for small ensembles (1)
    long full_step_max_walkers;   // MCMCPP_HIP_FULL_STEP_MAX_WALKERS    largest ensemble stepped that way (-1: 32768; 32767 where the matrix-core
                                  //                                     half-step kernel is the alternative)
    long task_table_mb;           // MCMCPP_HIP_TASK_TABLE_MB            size limit of the one-entry-per-draw jump table (16)
    long chain_subchunk_mb;       // MCMCPP_HIP_CHAIN_SUBCHUNK_MB        device chain staging per sub-chunk / ring budget (32)
    long graph_steps;             // MCMCPP_HIP_GRAPH_STEPS              ensemble steps per hipGraph replay (-1 here: 300 up to 32768 walkers, else 128)
    long debug_timing;            // MCMCPP_HIP_DEBUG_TIMING             1: run() prints its host-side phases to stderr
    long trickle;                 // MCMCPP_HIP_TRICKLE                  1: stored steps forwarded to pinned memory by the launches (1)
    long no_draw_wave;            // MCMCPP_HIP_NO_DRAW_WAVE             1: no extra draw wavefronts (0)
    long batch_draws;             // MCMCPP_HIP_BATCH_DRAWS              ensemble steps whose draw records one launch makes ahead of the matrix-core full-step launches; 0: the launches make them themselves; -1: as many as a graph replays (-1)
    long fill_branch;             // MCMCPP_HIP_FILL_BRANCH              > 0: the records of the NEXT graph replay are made beside this replay's step launches, in that many pieces on a parallel branch of the graph (0: one launch in line at the head of each replay)
    long copy_stream;             // MCMCPP_HIP_COPY_STREAM              1: chain downloads on a second stream (0)
    long pinned_direct;           // MCMCPP_HIP_PINNED_DIRECT            1: stored steps forwarded straight into a pinned chain_out (1)
    long comm_full_step;          // MCMCPP_HIP_COMM_FULL_STEP           split ensembles: 1 = one exchange per ensemble step (1), 0 = one per half-step
    long comm_compact;            // MCMCPP_HIP_COMM_COMPACT             split ensembles of more than one rank: 1 = exchange only the rows that moved (1), 0 = all-gather the slices
    long comm_compact_cap;        // MCMCPP_HIP_COMM_COMPACT_CAP         slots of an exchange block (0: learned from the run; a bound that is too small costs
                                  //                                     a repeated chunk, never a wrong chain)
    long comm_compact_chunk;      // MCMCPP_HIP_COMM_COMPACT_CHUNK       ensemble steps between two looks at the overflow flag (256)
    long force_multi_chain_kernels; // MCMCPP_HIP_FORCE_MC                experiments: single ensembles stepped by the several-chains instantiations (0)
    static Knobs from_environment()
    {
        Knobs k;
        k.passes = env_long("MCMCPP_HIP_PASSES", 0);
        k.waves_per_simd = env_long("MCMCPP_HIP_WAVES_PER_SIMD", 2);
        k.matrix_core_min_walkers = env_long("MCMCPP_HIP_MATRIX_CORE_MIN_WALKERS", 0);
        // (measured after the round-3 rework, us per launch with 8 / 16 walkers per wavefront: 16 384 updates 5.04 / 5.23,
        //  20 480: 7.37 / 6.57, 24 576: 7.54 / 6.63, 28 672: 9.97 / 6.72, 32 768: 10.02 / 6.78 -- profiles/r03_mc_p2_p4.txt)
        k.matrix_core_4pass = env_long("MCMCPP_HIP_MATRIX_CORE_4PASS_WALKERS", 18432);
        k.matrix_core_late = env_long("MCMCPP_HIP_MATRIX_CORE_LATE_DRAWS", 49152);
        k.full_step = env_long("MCMCPP_HIP_FULL_STEP", 1);
        k.full_step_max_walkers = env_long("MCMCPP_HIP_FULL_STEP_MAX_WALKERS", -1);
        k.task_table_mb = env_long("MCMCPP_HIP_TASK_TABLE_MB", 16);
        k.chain_subchunk_mb = env_long("MCMCPP_HIP_CHAIN_SUBCHUNK_MB", 32);
        k.graph_steps = env_long("MCMCPP_HIP_GRAPH_STEPS", -1);
        k.debug_timing = env_long("MCMCPP_HIP_DEBUG_TIMING", 0);
        k.trickle = env_long("MCMCPP_HIP_TRICKLE", 1);
        k.no_draw_wave = env_long("MCMCPP_HIP_NO_DRAW_WAVE", 0);
        k.batch_draws = env_long("MCMCPP_HIP_BATCH_DRAWS", -1);
        k.fill_branch = env_long("MCMCPP_HIP_FILL_BRANCH", 0);
        k.copy_stream = env_long("MCMCPP_HIP_COPY_STREAM", 0);
        k.pinned_direct = env_long("MCMCPP_HIP_PINNED_DIRECT", 1);
        k.comm_full_step = env_long("MCMCPP_HIP_COMM_FULL_STEP", 1);
        k.force_multi_chain_kernels = env_long("MCMCPP_HIP_FORCE_MC", 0);
        k.comm_compact = env_long("MCMCPP_HIP_COMM_COMPACT", 1);
        k.comm_compact_cap = env_long("MCMCPP_HIP_COMM_COMPACT_CAP", 0);
        k.comm_compact_chunk = env_long("MCMCPP_HIP_COMM_COMPACT_CHUNK", 256);
        if (k.comm_compact_chunk < 1) k.comm_compact_chunk = 1;
        return k;
    }
};

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
void launch_fill_draws_batch(const HalfStepArgs<double>& a, const StepCtl* ctl, const Affine128* step_jump, DrawRec<double>* out, int steps, hipStream_t stream)
{
    const unsigned grid = (unsigned)((a.shard_count + 63) / 64);
    hipLaunchKernelGGL(fill_draws_batch_kernel<double>, dim3(grid, (unsigned)(2 * steps)), dim3(192), 0, stream, a, ctl, step_jump, out);
}
void launch_fill_draws_batch(const HalfStepArgs<float>& a, const StepCtl* ctl, const Affine128* step_jump, DrawRec<float>* out, int steps, hipStream_t stream)
{
    const unsigned grid = (unsigned)((a.shard_count + 63) / 64);
    hipLaunchKernelGGL(fill_draws_batch_kernel<float>, dim3(grid, (unsigned)(2 * steps)), dim3(192), 0, stream, a, ctl, step_jump, out);
}
void launch_accepted_reduce(const uint32_t* partials, int partial_slots, int partial_waves, int count,
                            const StepCtl* ctl_after, const RunInfo* run, hipStream_t stream, int chains, StepCtl* ctl_keep)
{
    hipLaunchKernelGGL(accepted_reduce_kernel, dim3((unsigned)count, (unsigned)(chains > 1 ? chains : 1)), dim3(256), 0, stream, partials, partial_slots,
                       partial_waves, count, ctl_after, run, ctl_keep);
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
        knobs = Knobs::from_environment();
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
        if (c.comm_world >= 1)
        {
            // a rank of a split ensemble owns the comm_rank-th of comm_world equal slices of each half
            if (c.comm_rank < 0 || c.comm_rank >= c.comm_world) return fail(MCMCPP_HIP_E_ARG, "comm_rank %d outside 0..%d", c.comm_rank, c.comm_world - 1);
            if (n % c.comm_world) return fail(MCMCPP_HIP_E_ARG, "W/2 = %d does not divide by comm_world = %d", n, c.comm_world);
            const int per = n / c.comm_world;
            if (c.shard_count == 0)
            {
                shard_begin = c.comm_rank * per;
                shard_count = per;
            }
            else if (shard_begin != c.comm_rank * per || shard_count != per)
                return fail(MCMCPP_HIP_E_ARG, "the shard of rank %d of %d must be [%d, +%d)", c.comm_rank, c.comm_world, c.comm_rank * per, per);
            if (!c.comm && !c.comm_id) return fail(MCMCPP_HIP_E_ARG, "comm_world >= 1 needs comm_id or comm");
        }
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
        if (c.comm_world >= 1)
        {
            std::string why;
            rccl = Rccl::get(&why);
            if (!rccl) return fail(MCMCPP_HIP_E_COMM, "%s", why.c_str());
            if (c.comm)
            {
                comm = static_cast<ncclComm_t>(c.comm);
                int cnt = -1, rk = -1;
                NCCL_TRY(rccl->CommCount(comm, &cnt));
                NCCL_TRY(rccl->CommUserRank(comm, &rk));
                if (cnt != c.comm_world || rk != c.comm_rank)
                    return fail(MCMCPP_HIP_E_ARG, "the communicator is rank %d of %d, the config says %d of %d", rk, cnt, c.comm_rank, c.comm_world);
            }
            else
            {
                ncclUniqueId id;
                static_assert(sizeof(id) == MCMCPP_HIP_COMM_ID_BYTES, "mcmcpp_hip.h states the size of an RCCL id");
                std::memcpy(&id, c.comm_id, sizeof id);
                NCCL_TRY(rccl->CommInitRank(&comm, c.comm_world, id, c.comm_rank));
                own_comm = true;
            }
        }

        // Walkers per wavefront: fill the chip first (about two wavefronts per SIMD), then up to 8 per wavefront (still
        // served by the draw wavefront) and 16 for the largest ensembles.  Measured, 32 dims fp64 (tools/sweep_passes.txt):
        // 65 536 walkers 5.2 / 5.8 / 4.4e9 walker-steps/s with 4 / 8 / 16 walkers per wavefront, 262 144: 7.4e9 with 8,
        // 1 M: 6.7 / 8.0 / 7.8 / 7.4e9 with 8 / 16 / 32 / 64.
        // Independent ensembles stepped by the same launches (BASELINE config 4 on one GPU): chain k is seeded with
        // seed + k on the same stream, so all chains share the jump tables; see ChainGeometry for the layout.  What a
        // launch has to fill the chip with is the walkers of all chains together.
        K = c.num_chains > 1 ? c.num_chains : 1;
        if (K > kMaxChains) return fail(MCMCPP_HIP_E_ARG, "num_chains %d exceeds %d", K, kMaxChains);
        const long launch_walkers = (long)shard_count * K;  // walkers of one colour a launch updates
        const int wpp = 64 / lpw;
        long forced = knobs.passes;
        if (forced > 0)
            passes = (int)forced;
        else
        {
            const long target_waves = (long)num_cus * 4 * knobs.waves_per_simd;
            const int per_wave_cap = launch_walkers > 196608 ? 16 : 8;
            passes = 1;
            while (passes * 2 <= lpw && wpp * passes * 2 <= per_wave_cap && launch_walkers / ((long)wpp * passes * 2) >= target_waves) passes *= 2;
        }
        if (passes < 1) passes = 1;
        if (passes > lpw) passes = lpw;
        step_lpw = lpw;  // lanes per walker of the step kernels in use
        // Matrix-core variants of the half-step kernel (dense calculators, fp64, even D in 18..32): the wavefront's
        // walkers are rows of one MFMA tile -- 8 walkers (2 passes) until the chip is full, 16 (4 passes) beyond.
        const long mc_min = knobs.matrix_core_min_walkers;
        int mc_level = -1;  // matrix-core half-step kernel in use: 0 = 8 walkers per wavefront, 1 = 16, 2 = 16 with late draws
        if (table->half_step_mc[0][lpw_log][epl_shift] && (D % 2 == 0) && mc_min >= 0 && shard_count >= mc_min &&
            c.calc_id == MCMCPP_HIP_CALC_DENSE_GAUSSIAN &&     // (they read the padded matrix this file prepares)
            (size_t)W * (size_t)D * sizeof(T) < (1ull << 32))  // (and address a chain's arrays with 32-bit byte offsets)
        {
            mc_level = launch_walkers >= knobs.matrix_core_4pass ? 1 : 0;
            // (about as many updates as one round of wavefront slots holds at three wavefronts per SIMD, or more: four per SIMD,
            //  draws behind the accept -- profiles/r03_mc_threshold.txt)
            if (mc_level == 1 && knobs.matrix_core_late >= 0 && launch_walkers >= knobs.matrix_core_late && table->half_step_mc[2][lpw_log][epl_shift]) mc_level = 2;
            half_fn = table->half_step_mc[mc_level][lpw_log][epl_shift];
            passes = mc_level == 0 ? 2 : 4;
            step_lpw = 16;  // (the matrix-core kernels map a walker to 16 lanes x 2 elements in either element type)
        }

        // One launch per ensemble step (full_step_kernel.hpp) while the ensemble is small enough that a half-step
        // launch is bounded by its launch boundary and latencies rather than by HBM; needs the whole ensemble here.
        // A rank of a split ensemble takes the same kernels for its slice (they repeat red updates owned by other ranks,
        // so the ranks exchange rows once per ensemble step), by the size of what it updates.
        full_fn = nullptr;
        const bool whole = shard_count == n && shard_begin == 0;
        // (measured, 32 dims fp64, us per ensemble step full / half: isotropic 32 768 walkers 7 % in favour of full steps;
        //  dense with the matrix-core half-step kernel 10.75 / 10.06 at 32 768, 5.47 / 7.28 at 16 384: profiles/r03_mc_probe_e.txt)
        const bool mc_half = mc_level >= 0;
        const long full_step_max = knobs.full_step_max_walkers >= 0 ? knobs.full_step_max_walkers : (mc_half ? 32767 : 32768);
        if ((c.comm_world >= 1 ? (knobs.comm_full_step != 0 && knobs.full_step != 0) : (whole && knobs.full_step != 0)) &&
            2 * launch_walkers <= full_step_max)
        {
            full_fn = table->full_step[lpw_log][epl_shift];
            full_wpb = kWavesPerBlock * (64 / lpw);
            if (table->full_step_mc[lpw_log][epl_shift] && (D % 2 == 0) && mc_min >= 0 && shard_count >= mc_min &&
                c.calc_id == MCMCPP_HIP_CALC_DENSE_GAUSSIAN &&  // (it reads the padded matrix this file prepares)
                W < (1 << 24) && (size_t)W * (size_t)D * sizeof(T) < (1ull << 32))  // (and addresses rows by 24-bit products, 32-bit offsets)
            {
                full_fn = table->full_step_mc[lpw_log][epl_shift];
                full_wpb = kWavesPerBlock * 8;
            }
        }

        if (c.comm_world > 1 && knobs.comm_compact != 0)
        {
            const uint32_t cap_full = (uint32_t)((full_fn ? 2 : 1) * shard_count);
            void* p = nullptr;
            HIP_TRY(hipMalloc(&p, xblock_bytes<T>(cap_full, D) * (size_t)c.comm_world));
            d_xblocks = (char*)p;
            HIP_TRY(hipMalloc(&p, sizeof(uint32_t) * (size_t)W));
            d_seen = (uint32_t*)p;
            HIP_TRY(hipMalloc(&p, sizeof(XStats)));
            d_xstats = (XStats*)p;
            HIP_TRY(hipMalloc(&p, sizeof(T) * (size_t)W * D + (sizeof(T) + sizeof(uint32_t)) * (size_t)W + sizeof(Diag)));
            d_snap = (char*)p;
        }
        if (K > 1 && (!whole || c.comm_world >= 1 || c.device_positions))
            return fail(MCMCPP_HIP_E_ARG, "num_chains > 1: whole ensembles on one device only (no shards, communicator or caller-owned positions)");
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
            size_t need = (size_t)K * 2 * sizeof(T) * (size_t)W * D    // pos, pos_alt
                          + (size_t)K * logp_chain_stride_bytes<T>(n) + (size_t)K * kCtlChainStride + tables_total_bytes(n, true, K)
                          + sizeof(T) * ((size_t)(c.calc_params_len > 0 ? c.calc_params_len : 0) + 32 * 32)
                          + (size_t)K * sizeof(uint32_t) * graph_len * 2 * waves_bound + 64 * 1024;
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
            if (int rc = carve(&d_pos, sizeof(T) * (size_t)W * D * K)) return rc;
            own_pos = true;
        }
        // log-posteriors [2][W] (the second half is the full-step kernels' other buffer) and, right behind them, the
        // accepted counters [W]: one piece, so that kernels short of preloaded arguments can derive both addresses
        static_assert(sizeof(StepCtl) == 64 && sizeof(RunInfo) == 64 && kRunBehindCtlBytes + (int)sizeof(RunInfo) <= kCtlChainStride, "ChainGeometry");
        if (int rc = carve(&d_logp, logp_chain_stride_bytes<T>(n) * (size_t)K)) return rc;
        d_nacc = reinterpret_cast<uint32_t*>(d_logp + 2 * (size_t)W);
        if (full_fn)
            if (int rc = carve(&d_pos_alt, sizeof(T) * (size_t)W * D * K)) return rc;
        {
            // the two control records and, kRunBehindCtlBytes behind the first, the run record: one piece (a kernel short
            // of preloaded arguments derives the run record's address)
            static_assert(2 * sizeof(StepCtl) <= (size_t)kRunBehindCtlBytes, "the run record follows the control records");
            char* piece = nullptr;
            if (int rc = carve(&piece, (size_t)kCtlChainStride * (size_t)K)) return rc;
            d_ctl = reinterpret_cast<StepCtl*>(piece);
            d_run = reinterpret_cast<RunInfo*>(piece + kRunBehindCtlBytes);
        }
        if (int rc = carve(&d_diag, sizeof(Diag))) return rc;
        if (int rc = carve(&d_status, 8 * sizeof(uint64_t))) return rc;  // split ensembles: the status word the ranks agree on
        // the draw records (two buffers: see HalfStepArgs::draws) and, right behind them, the jump tables: one piece
        // whose layout follows from n alone (JumpTables), so that kernels reach the tables from the record pointer
        static_assert(sizeof(DrawRec<T>) == 32, "the table offsets assume 32-byte records");
        have_task_table = (size_t)3 * n * sizeof(Affine128) <= ((size_t)knobs.task_table_mb << 20);
        {
            char* piece = nullptr;
            if (int rc = carve(&piece, tables_total_bytes(n, have_task_table, K))) return rc;
            d_draws = reinterpret_cast<DrawRec<T>*>(piece);
            d_task_jump = have_task_table ? reinterpret_cast<Affine128*>(piece + tables_offset_task(n, K)) : nullptr;
            d_jump_hi = reinterpret_cast<Affine128*>(piece + tables_offset_hi(n, have_task_table, K));
            d_jump_lo = reinterpret_cast<Affine128*>(piece + tables_offset_lo(n, have_task_table, K));
        }
        HIP_TRY(hipMemset(d_draws, 0, sizeof(DrawRec<T>) * (size_t)W * 2 * K));
#if defined(MCMCPP_EXP_PUSH) && !defined(MCMCPP_STAMPS)
        // experiment 3i (a variant build, never the library that ships): the scratch "inboxes" the pushed rows are written to
        HIP_TRY(hipMalloc(&d_stamps, sizeof(T) * (size_t)3 * (size_t)n * (size_t)D + 4096));
#endif
#ifdef MCMCPP_STAMPS
        HIP_TRY(hipMalloc(&d_stamps, kStampWords * sizeof(unsigned long long)));  // [8 stamps][2 alternating launches][start, end of 4096 workgroups | end of their draw wavefronts]
        HIP_TRY(hipMemset(d_stamps, 0, kStampWords * sizeof(unsigned long long)));
#endif
        HIP_TRY(hipMemset(d_logp, 0, logp_chain_stride_bytes<T>(n) * (size_t)K));
        HIP_TRY(hipMemset(d_diag, 0, sizeof(Diag)));
        HIP_TRY(hipMemset(d_ctl, 0, (size_t)kCtlChainStride * (size_t)K));
        // pinned scratch of the host: [0, 512) as before (control record at 128, four run records from 256 on);
        // per-chain control records from 1024, per-chain run records from 1024 + 64 * kMaxChains on
        HIP_TRY(hipHostMalloc(&h_pinned, 1024 + 64 * kMaxChains + 4 * 64 * kMaxChains, hipHostMallocDefault));

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
        for (int k = 0; k < K; ++k)
        {
            U128 inc_k;
            pcg_seed(c.seed + (uint64_t)k, c.stream, &state0_of[k], &inc_k);  // (same stream: the same increment)
        }
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
        // HIP cannot capture on the legacy default stream (hipErrorStreamCaptureUnsupported): a caller that hands over
        // NULL / hipStreamLegacy gets plain launches instead of graph replays
        if (!own_stream && (stream == nullptr || stream == hipStreamLegacy)) graph_steps = -1;
        // Draw records made ahead of the step launches, a batch of steps per launch (fill_draws_batch_kernel): for the
        // matrix-core full-step kernel of one whole ensemble on one device.  step_jump[j]: the draws of j ensemble steps.
        batch_draws = 0;
        // (only for handles that step by graph replays: with plain launches -- the caller's legacy default stream -- every
        //  step would drag a fill launch of its own along, and a replay's worth of record memory would sit unused)
        if (full_fn && full_fn == table->full_step_mc[lpw_log][epl_shift] && K == 1 && c.comm_world < 1 && whole && knobs.batch_draws != 0 && knobs.no_draw_wave == 0 &&
            graph_steps >= 1)
        {
            // (by default as many steps as a graph replays: one fill launch per replay)
            const long want = knobs.batch_draws > 0 ? knobs.batch_draws : (long)graph_steps;
            batch_draws = (int)(want > 512 ? 512 : want);
            HIP_TRY(hipMalloc(&d_draws_batch, sizeof(DrawRec<T>) * (size_t)batch_draws * 2 * (size_t)n));
            HIP_TRY(hipMemset(d_draws_batch, 0, sizeof(DrawRec<T>) * (size_t)batch_draws * 2 * (size_t)n));  // (partner indices a kernel may follow)
            // Records of the NEXT replay made beside this replay's launches (fill_pieces > 0): piece k, forked off the launch
            // sequence behind the last step that read its record sets, refills them for the same steps of the next replay
            // from the control record the previous replay left in d_ctl_keep (step_jump[batch_draws + j]).
            if (knobs.fill_branch > 0 && batch_draws == graph_steps && batch_draws >= 2)
            {
                fill_pieces = (int)(knobs.fill_branch < batch_draws ? knobs.fill_branch : batch_draws);
                HIP_TRY(hipStreamCreateWithFlags(&fill_stream, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&ev_fill_fork, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&ev_fill_join, hipEventDisableTiming));
                HIP_TRY(hipMalloc(&d_ctl_keep, sizeof(StepCtl)));
                HIP_TRY(hipMemset(d_ctl_keep, 0, sizeof(StepCtl)));
            }
            std::vector<Affine128> sj((size_t)batch_draws * (fill_pieces > 0 ? 2 : 1));
            for (size_t j = 0; j < sj.size(); ++j) sj[j] = pcg_jump(inc, (unsigned __int128)6 * (unsigned)n * (unsigned __int128)j);
            HIP_TRY(hipMalloc(&d_step_jump, sizeof(Affine128) * sj.size()));
            HIP_TRY(hipMemcpy(d_step_jump, sj.data(), sizeof(Affine128) * sj.size(), hipMemcpyHostToDevice));
        }
        partial_slots = graph_steps >= 1 ? graph_steps : 1;
        partial_waves = (int)(full_fn ? full_grid_blocks() : grid_blocks()) * kWavesPerBlock;
        if ((int)grid_blocks() * kWavesPerBlock > partial_waves) partial_waves = (int)grid_blocks() * kWavesPerBlock;
        if (int rc = carve(&d_partials, sizeof(uint32_t) * (size_t)partial_slots * 2 * (size_t)partial_waves * K)) return rc;
        HIP_TRY(hipMemset(d_partials, 0, sizeof(uint32_t) * (size_t)partial_slots * 2 * (size_t)partial_waves * K));
        chain_subchunk_bytes = (size_t)knobs.chain_subchunk_mb << 20;
        return MCMCPP_HIP_OK;
    }

    // Ensemble steps per hipGraph replay.  A boundary between two replays costs a few microseconds of the launch
    // sequence: ensembles small enough for one launch per step (5.6 us each) take 300 per replay -- with the bench's
    // slicing interval of 100 that is three stored steps per replay, as many as the forwarding ring allows; measured 1.5 %
    // over 128 -- larger ones, whose launches are long and whose per-step counters grow with the walker count, 128.
    long default_graph_steps() const { return (knobs.graph_steps >= 0 ? knobs.graph_steps : (W <= 32768 ? 300 : 128)); }

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
        HIP_TRY(hipMemcpyAsync(d_pos, pos, sizeof(T) * (size_t)W * D * K, hipMemcpyHostToDevice, stream));
        for (int k = 0; k < K; ++k)
        {
            HIP_TRY(hipMemcpyAsync(logp_of(k), (const T*)logp + (size_t)k * W, sizeof(T) * (size_t)W, hipMemcpyHostToDevice, stream));
            HIP_TRY(hipMemsetAsync(nacc_of(k), 0, sizeof(uint32_t) * (size_t)W, stream));
        }
        HIP_TRY(hipMemsetAsync(d_diag, 0, sizeof(Diag), stream));
        half_steps = 0;
        steps_since_reset = 0;
        records_valid = false;
        records_ahead_of = kNoStep;
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
        run_touched_device = false;
        const int rc = comm ? run_split(n_saved, interval, chain_out, accepted_per_step) : run_whole(n_saved, interval, chain_out, accepted_per_step);
        if (rc != MCMCPP_HIP_OK && run_touched_device) abandon_state();
        return rc;
    }

    // staging -> the caller's memory: stored steps [first, first + count) of every chain (chain k's steps are n_saved steps
    // apart in the caller's array, sub_saved steps apart in the staging buffer)
    void hand_out_subchunk(char* chain_out, const char* stage, size_t step_bytes, int64_t sub_saved, int64_t n_saved, int64_t first, int64_t count) const
    {
        for (int k = 0; k < K; ++k)
            std::memcpy(chain_out + step_bytes * ((size_t)n_saved * (size_t)k + (size_t)first), stage + step_bytes * (size_t)sub_saved * (size_t)k,
                        step_bytes * (size_t)count);
    }

    // A failure after the first launch of a run leaves walkers, control records and draw records ahead of the host's
    // counters (and possibly the live ensemble in the second buffer, or the stream in capture mode): nothing on the
    // device can be trusted any more.  The handle then insists on a new set_state, as the DE sampler does.
    void abandon_state()
    {
        const std::string keep = error;
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
        {
            hipGraph_t g = nullptr;
            (void)hipStreamEndCapture(stream, &g);
            if (g) (void)hipGraphDestroy(g);
        }
        (void)hipStreamSynchronize(stream);
        (void)hipGetLastError();
        have_state = false;
        records_valid = false;
        records_ahead_of = kNoStep;
        error = keep + " (the walker state on the device is no longer consistent: call set_state again)";
    }

    int run_whole(int64_t n_saved, int32_t interval, void* chain_out, uint32_t* accepted_per_step)
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
        const bool dbg = knobs.debug_timing != 0;
        const auto tp0 = std::chrono::steady_clock::now();

        const size_t step_bytes = sizeof(T) * (size_t)W * D;
        int64_t sub_saved = n_saved;  // stored steps per sub-chunk
        if (chain_out)
        {
            sub_saved = (int64_t)(chain_subchunk_bytes / (step_bytes * (size_t)K));
            const int64_t eighth = (n_saved + 7) / 8;  // keep the last (un-overlappable) host copy short
            if (sub_saved > eighth) sub_saved = eighth;
            if (sub_saved < 1) sub_saved = 1;
        }
        // Full-step kernels forward stored steps to pinned host memory themselves (trickle_stored_step): a ring of
        // `ring` slots on the device with a twin in pinned host memory, no copy engine, no gap in the launch sequence.
        const bool trickle = full_fn && chain_out && step_bytes % 16 == 0 && knobs.trickle != 0;
        // chain_out in pinned host memory (mcmcpp_hip_host_alloc: the facade's Chain blocks): the launches forward stored
        // steps straight into their final place -- no pinned twin of the device ring, no host copy
        void* direct_stage = (trickle && knobs.pinned_direct != 0) ? device_view_of_pinned(chain_out, step_bytes * (size_t)n_saved * K) : nullptr;
        int64_t ring = 0, chunk_steps = 0;
        if (trickle)
        {
            ring = 4;
            while (ring < 64 && (size_t)(2 * ring) * step_bytes <= 2 * chain_subchunk_bytes) ring *= 2;
            // the host enqueues one chunk ahead of the one it waits for: stored steps of two chunks are in flight
            int64_t per_chunk = (graph_steps > 0 ? graph_steps : 64) / (int64_t)interval;
            if (!direct_stage && per_chunk > (ring - 2) / 2) per_chunk = (ring - 2) / 2;
            if (per_chunk < 1) per_chunk = 1;
            chunk_steps = per_chunk * interval;
        }
        int rc = ensure_run_buffers(accepted_per_step ? (size_t)total * K : 0, (chain_out && !trickle) ? step_bytes * (size_t)sub_saved * K : 0,
                                    trickle ? step_bytes * (size_t)ring * K : 0, direct_stage == nullptr);
        if (rc) return rc;
        if (accepted_per_step) HIP_TRY(hipMemsetAsync(d_acc, 0, sizeof(uint32_t) * (size_t)total * K, stream));
        run_touched_device = true;  // from here on an error leaves the device ahead of the host's bookkeeping
        rc = write_ctl(0);  // step_in_run = 0, stream position from the host-side half-step count
        if (rc) return rc;
        records_valid = false;  // (until this call has finished: an error on the way leaves them unknown)
        run_info_idle = false;
        if (full_fn)
        {
            for (int k = 0; k < K; ++k)
                hipLaunchKernelGGL(mark_rows_moved_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, stream, nacc_of(k), W, kRowMovedBit);
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
            rc = run_trickle(n_saved, interval, (char*)chain_out, accepted_per_step != nullptr, step_bytes, ring, chunk_steps, &launch_ms, (char*)direct_stage);
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
            for (int k = 0; k < K; ++k)
            {
                // (several chains: one record each; the upload slots rotate per sub-chunk as for one chain)
                RunInfo* ri = K > 1 ? reinterpret_cast<RunInfo*>((char*)h_pinned + 1024 + 64 * kMaxChains + 64 * k + 64 * kMaxChains * (c % 4))
                                    : reinterpret_cast<RunInfo*>((char*)h_pinned + 256 + 64 * (c % 4));
                static_assert(sizeof(RunInfo) <= 64, "the pinned upload slots are 64 bytes apart");
                // chain k's stored steps of this sub-chunk: the k-th run of sub_saved steps of the device half
                ri->chain = chain_out ? (void*)((char*)d_chain[buf] + step_bytes * (size_t)sub_saved * (size_t)k) : nullptr;
                ri->accepted_per_step = accepted_per_step ? d_acc + (size_t)k * (size_t)total : nullptr;
                ri->interval = interval;
                ri->chain_slot_base = -first;
                ri->stage = nullptr;
                ri->slot_mask = -1;
                ri->slice_bytes = 0;
                ri->step_bytes = (int64_t)step_bytes;
                HIP_TRY(hipMemcpyAsync(run_of(k), ri, sizeof(RunInfo), hipMemcpyHostToDevice, stream));
            }
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
                // (the whole device half in one copy: with several chains, chain k's steps sit sub_saved steps apart)
                const size_t half_used = K > 1 ? step_bytes * (size_t)sub_saved * (size_t)(K - 1) + step_bytes * (size_t)now : step_bytes * (size_t)now;
                if (copy_stream)
                {
                    // the download runs beside the next sub-chunk's launches (which fill the other device half)
                    HIP_TRY(hipEventRecord(ev_filled[buf], stream));
                    HIP_TRY(hipStreamWaitEvent(copy_stream, ev_filled[buf], 0));
                    HIP_TRY(hipMemcpyAsync(h_stage[buf], d_chain[buf], half_used, hipMemcpyDeviceToHost, copy_stream));
                    HIP_TRY(hipEventRecord(ev_copied[buf], copy_stream));
                }
                else
                {
                    HIP_TRY(hipMemcpyAsync(h_stage[buf], d_chain[buf], half_used, hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipEventRecord(ev_copied[buf], stream));
                }
                if (pending_first >= 0)
                {
                    HIP_TRY(hipEventSynchronize(ev_copied[pending_buf]));
                    hand_out_subchunk((char*)chain_out, (const char*)h_stage[pending_buf], step_bytes, sub_saved, n_saved, pending_first, pending_count);
                    publish_stored(pending_first + pending_count);
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
                hand_out_subchunk((char*)chain_out, (const char*)h_stage[pending_buf], step_bytes, sub_saved, n_saved, pending_first, pending_count);
                publish_stored(pending_first + pending_count);
            }
            if (full_fn && (run_step & 1))
            {
                // an odd number of full steps leaves the ensemble in the second buffer: bring it (and the control
                // record) home, so that everything outside run() only ever knows the first
                HIP_TRY(hipMemcpyAsync(d_pos, d_pos_alt, sizeof(T) * (size_t)W * D * K, hipMemcpyDeviceToDevice, stream));
                for (int k = 0; k < K; ++k)
                {
                    HIP_TRY(hipMemcpyAsync(logp_of(k), logp_of(k) + W, sizeof(T) * (size_t)W, hipMemcpyDeviceToDevice, stream));
                    HIP_TRY(hipMemcpyAsync(ctl_of(k), ctl_of(k) + 1, sizeof(StepCtl), hipMemcpyDeviceToDevice, stream));
                }
            }
            HIP_TRY(hipStreamSynchronize(stream));
            for (int64_t c = (n_sub > 3 ? n_sub - 3 : 0); c < n_sub; ++c)
            {
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, ev_t0[c & 3], ev_t1[c & 3]));
                launch_ms += ms;
            }
            last_ms = launch_ms;
            last_launches = full_fn ? total : 2 * total;  // (a launch steps all chains)
            half_steps += 2 * (uint64_t)total;
            steps_since_reset += (uint64_t)total;
            // the last launch left the records of the next ensemble step behind (full-step launches: with partner2) --
            // unless the records were made ahead in batches, which leaves the two-buffer records alone
            records_valid = batch_draws == 0;
            records_step = half_steps >> 1;
            records_partner2 = full_fn != nullptr;
            if (accepted_per_step)
                HIP_TRY(hipMemcpy(accepted_per_step, d_acc, sizeof(uint32_t) * (size_t)total * K, hipMemcpyDeviceToHost));
        }
        const auto tp3 = std::chrono::steady_clock::now();
        host_enqueue_ms = std::chrono::duration<double, std::milli>(tp2 - tp1).count();
        host_wall_ms = std::chrono::duration<double, std::milli>(tp3 - tp0).count();
        exchange_us_per_step = 0.0;
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

    // ---- one ensemble split over the ranks of an RCCL communicator (BASELINE config 5) -----------------
    // Stands in for ParallelEnsembleSampler::runMCMC with threadCount workers (ParallelEnsembleSampler.h:285-291) and the
    // controller's mid-step / end-step barriers (Threading/RedBlkCtrlerSpinLock.h:240-322).  Every rank holds the full
    // replica of the positions and updates its slice; launches AND exchanges are enqueued on the launch stream by this
    // host thread, nothing waits for the device until the end (stored steps excepted, a staging buffer at a time).
    //   full-step kernels (slices of up to full_step_max_walkers / 2 walkers per colour): ONE exchange per ensemble
    //     step.  A black walker's group repeats the red update of its partner wherever that one lives, so a rank needs
    //     nothing from the others inside a step; afterwards the updated rows and log-posteriors of both colours are
    //     all-gathered (in place: every rank's slice sits where the gather puts it) into the buffer the step wrote.  The
    //     repeated updates read the red draw records of ALL walkers, which fill_draws_kernel makes per step (the
    //     records of a rank's own walkers are also left behind by its draw wavefronts: same bits).
    //   half-step kernels (larger slices): the reference's scheme, one exchange of the updated colour per half-step.
    // The random stream is addressed by the global walker index, so the trajectory does not depend on the number of ranks.
    int exchange_rows(T* pos_buf, T* logp_buf, int first_color, int colors)
    {
        NCCL_TRY(rccl->GroupStart());
        for (int c = first_color; c < first_color + colors; ++c)
        {
            T* half = pos_buf + (size_t)c * n * D;
            NCCL_TRY(rccl->AllGather(half + (size_t)shard_begin * D, half, (size_t)shard_count * D, RcclType<T>::value, comm, stream));
            if (logp_buf)
            {
                T* lh = logp_buf + (size_t)c * n;
                NCCL_TRY(rccl->AllGather(lh + shard_begin, lh, (size_t)shard_count, RcclType<T>::value, comm, stream));
            }
        }
        NCCL_TRY(rccl->GroupEnd());
        return MCMCPP_HIP_OK;
    }

    // Rank-local preparation of a split run: argument and state checks, staging and counter buffers.  Whatever fails here
    // fails on this rank only -- the caller agrees on it with the other ranks (agree_on_status) before the first launch.
    int prepare_split(int64_t n_saved, int32_t interval, void* chain_out, uint32_t* accepted_per_step, int64_t* stage_slots_out)
    {
        if (!have_state) return fail(MCMCPP_HIP_E_STATE, "run: set_state has not been called");
        if (n_saved < 0 || interval < 1) return fail(MCMCPP_HIP_E_ARG, "run: n_saved >= 0 and interval >= 1 required");
        if (half_steps & 1) return fail(MCMCPP_HIP_E_STATE, "run: an ensemble step is half done (half_step_async)");
        HIP_TRY(hipSetDevice(device));
        const int64_t total = n_saved * (int64_t)interval;
        const size_t step_bytes = sizeof(T) * (size_t)W * D;
        // stored steps: device -> pinned staging on the launch stream, handed to the caller a staging buffer at a time
        // (the same number on every rank, whether it stores or not: the chunks of a run end where the staging buffer of
        //  the ranks that do store is full, and every rank must cut its run into the same chunks)
        int64_t stage_slots = (int64_t)(((size_t)256 << 20) / step_bytes);
        if (stage_slots < 1) stage_slots = 1;
        if (stage_slots > n_saved) stage_slots = n_saved;
        if (chain_out && total > 0)
        {
            if (step_bytes * (size_t)stage_slots > split_stage_capacity)
            {
                HIP_TRY(hipStreamSynchronize(stream));
                if (h_split_stage) hipHostFree(h_split_stage);
                h_split_stage = nullptr;
                split_stage_capacity = 0;
                if (hipHostMalloc(&h_split_stage, step_bytes * (size_t)stage_slots, hipHostMallocDefault) != hipSuccess)
                {
                    (void)hipGetLastError();
                    return fail(MCMCPP_HIP_E_NOMEM, "run: cannot allocate %zu bytes of pinned staging", step_bytes * (size_t)stage_slots);
                }
                split_stage_capacity = step_bytes * (size_t)stage_slots;
            }
        }
        *stage_slots_out = stage_slots;
        // (the per-step accepted counts are always kept on the device and all-reduced at the end of a split run, whether this
        //  rank's caller wants them or not: a collective must not depend on one rank's arguments)
        (void)accepted_per_step;
        const int rc = ensure_run_buffers(total > 0 ? (size_t)total : 0, 0, 0);
        if (rc) return rc;
        if (ev_x.empty())
        {
            ev_x.assign(2 * kMaxExchangeSamples, nullptr);
            for (hipEvent_t& e : ev_x) HIP_TRY(hipEventCreate(&e));
        }
        return MCMCPP_HIP_OK;
    }

    // Every rank learns the worst status among the ranks (and that all were asked for the same number of steps) before any
    // of them launches or exchanges anything: a rank that failed its preparation would otherwise leave the others waiting
    // in their first all-gather for good.  One small all-reduce and one stream synchronisation per run.
    int agree_on_status(int local_rc, int64_t total, int32_t interval, bool stores, bool* any_rank_stores)
    {
        uint64_t* hs = reinterpret_cast<uint64_t*>(h_pinned);  // [0, 64): words out, [64, 128): words back
        hs[0] = (uint64_t)local_rc;
        hs[1] = (uint64_t)total;
        hs[2] = ~(uint64_t)total;
        hs[3] = (uint64_t)(uint32_t)interval;
        hs[4] = ~(uint64_t)(uint32_t)interval;
        hs[5] = stores ? 1u : 0u;  // (stored steps are handed out a staging buffer at a time: where the chunks of the run end)
        const std::string mine = error;
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipMemcpyAsync(d_status, hs, 6 * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        NCCL_TRY(rccl->AllReduce(d_status, d_status, 6, ncclUint64, ncclMax, comm, stream));
        HIP_TRY(hipMemcpyAsync(hs + 8, d_status, 6 * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (local_rc != MCMCPP_HIP_OK)
        {
            error = mine;
            return local_rc;
        }
        if (hs[8] != 0) return fail((int)hs[8], "run: the preparation of another rank of the split ensemble failed (code %d); nothing was launched", (int)hs[8]);
        if (hs[9] != ~hs[10] || hs[11] != ~hs[12])
            return fail(MCMCPP_HIP_E_ARG, "run: the ranks of the split ensemble were asked for different numbers of steps or intervals; nothing was launched");
        *any_rank_stores = hs[13] != 0;
        return MCMCPP_HIP_OK;
    }

    static constexpr int kMaxExchangeSamples = 32;

    // ---- the exchange of moved rows only (exchange_kernels.hpp) ----------------------------------------------------------
    // pack -> one all-gather of G equal blocks of `cap` slots -> scatter into the replica (both position buffers when
    // `other_pos` is given).  Colours [color0, color0 + colors) of this rank's slice.
    int exchange_compact(T* cur_pos, T* other_pos, T* cur_logp, T* other_logp, int color0, int colors, uint32_t cap)
    {
        const size_t bb = xblock_bytes<T>(cap, D);
        char* own = d_xblocks + bb * (size_t)cfg.comm_rank;
        const int pack_waves = (colors * shard_count + kPackWalkersPerWave - 1) / kPackWalkersPerWave;
        hipLaunchKernelGGL(exchange_pack_kernel<T>, dim3((unsigned)((pack_waves + kPackWavesPerBlock - 1) / kPackWavesPerBlock)), dim3(64 * kPackWavesPerBlock), 0, stream, (const T*)cur_pos, (const T*)cur_logp,
                           (const uint32_t*)d_nacc, d_seen, own, cap, n, D, shard_begin, shard_count, color0, colors);
        HIP_TRY(hipGetLastError());
        NCCL_TRY(rccl->AllGather(own, d_xblocks, bb, ncclInt8, comm, stream));
        const bool vec = ((size_t)D * sizeof(T)) % 16 == 0;
        const int pieces = vec ? (int)((size_t)D * sizeof(T) / 16) : D;
        int lpr = 1;
        while (lpr < pieces && lpr < 64) lpr <<= 1;
        const unsigned rows_per_block = 256u / (unsigned)lpr;
        hipLaunchKernelGGL(exchange_scatter_kernel<T>, dim3((cap + rows_per_block - 1) / rows_per_block, (unsigned)(cfg.comm_world - 1)), dim3(256), 0, stream,
                           d_xblocks, bb, cap, cfg.comm_world, cfg.comm_rank, D, cur_pos, other_pos, cur_logp, other_logp, d_xstats);
        HIP_TRY(hipGetLastError());
        return MCMCPP_HIP_OK;
    }

    // (seen counters of the own slice <- accepted counters; statistics and the own block's count <- 0)
    int exchange_reset(uint32_t cap)
    {
        hipLaunchKernelGGL(exchange_sync_seen_kernel, dim3((unsigned)((2 * shard_count + 255) / 256)), dim3(256), 0, stream, (const uint32_t*)d_nacc, d_seen, n,
                           shard_begin, shard_count);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemsetAsync(d_xstats, 0, sizeof(XStats), stream));
        HIP_TRY(hipMemsetAsync(d_xblocks + xblock_bytes<T>(cap, D) * (size_t)cfg.comm_rank, 0, sizeof(XBlockHeader), stream));
        return MCMCPP_HIP_OK;
    }

    // One ensemble over the ranks of a communicator.  The run proceeds in CHUNKS of steps; the host waits for the device
    // only at the end of a chunk, and only when it has a reason to: stored steps to hand out (a staging buffer's worth), or
    // -- exchanging moved rows only -- the overflow flag to look at.  A chunk whose blocks overflowed is rolled back to the
    // snapshot taken in front of it and repeated with blocks that hold a whole slice; the slot bound of the following
    // chunks is what the last one needed, plus an eighth.
    int run_split(int64_t n_saved, int32_t interval, void* chain_out, uint32_t* accepted_per_step)
    {
        last_ms = 0.0;
        last_launches = 0;
        host_enqueue_ms = host_wall_ms = exchange_us_per_step = 0.0;
        xchg_bytes_per_step = 0.0;
        xchg_rollbacks = 0;
        const auto tp0 = std::chrono::steady_clock::now();
        int64_t stage_slots = 0;
        const int prep = prepare_split(n_saved, interval, chain_out, accepted_per_step, &stage_slots);
        const int64_t total = (n_saved > 0 && interval > 0) ? n_saved * (int64_t)interval : 0;
        bool any_rank_stores = false;
        int rc = agree_on_status(prep, total, interval, chain_out != nullptr, &any_rank_stores);
        if (rc) return rc;
        if (total == 0) return MCMCPP_HIP_OK;
        const size_t step_bytes = sizeof(T) * (size_t)W * D;
        constexpr int kMaxSamples = kMaxExchangeSamples;
        HIP_TRY(hipMemsetAsync(d_acc, 0, sizeof(uint32_t) * (size_t)total, stream));
        run_touched_device = true;
        rc = write_ctl(0);
        if (rc) return rc;
        records_valid = false;
        run_info_idle = false;
        {
            RunInfo* ri = reinterpret_cast<RunInfo*>((char*)h_pinned + 256);
            ri->chain = nullptr;  // (stored steps are copied from the replica after the exchange)
            ri->accepted_per_step = d_acc;
            ri->interval = interval;
            ri->chain_slot_base = 0;
            ri->stage = nullptr;
            ri->slot_mask = -1;
            ri->slice_bytes = 0;
            ri->step_bytes = (int64_t)step_bytes;
            HIP_TRY(hipMemcpyAsync(d_run, ri, sizeof(RunInfo), hipMemcpyHostToDevice, stream));
        }
        if (full_fn)
        {
            hipLaunchKernelGGL(mark_rows_moved_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, stream, d_nacc, W, kRowMovedBit);
            HIP_TRY(hipGetLastError());
        }
        const uint64_t half_steps0 = half_steps;  // (the member moves on when the run has succeeded)
        enq_step = half_steps0 >> 1;
        run_step = 0;
        args_red = make_args(0);
        args_blk = make_args(1);
        // engine state in front of the red half-step of the coming ensemble step (what write_ctl put into the control record)
        U128 red_base = apply(pcg_jump(inc, (unsigned __int128)3 * (unsigned)n * (unsigned __int128)half_steps0), state0);
        HalfStepArgs<T> fill_red = make_args(0);
        fill_red.shard_begin = 0;
        fill_red.shard_count = n;

        const bool compact = d_xblocks != nullptr;
        const uint32_t cap_full = (uint32_t)((full_fn ? 2 : 1) * shard_count);  // a block that holds every walker of an exchange
        uint32_t cap = cap_full;
        if (compact)
        {
            if (knobs.comm_compact_cap > 0)
                cap = (uint32_t)(knobs.comm_compact_cap < (long)cap_full ? knobs.comm_compact_cap : (long)cap_full);
            else if (xcap_learned > 0)
                cap = xcap_learned < cap_full ? xcap_learned : cap_full;
            if (full_fn)
            {
                // a remote walker's row must be current in BOTH buffers (the scatter keeps it so from here on)
                HIP_TRY(hipMemcpyAsync(d_pos_alt, d_pos, step_bytes, hipMemcpyDeviceToDevice, stream));
                HIP_TRY(hipMemcpyAsync(d_logp + W, d_logp, sizeof(T) * (size_t)W, hipMemcpyDeviceToDevice, stream));
            }
        }

        const int64_t sample_stride = total > kMaxSamples ? total / kMaxSamples : 1;
        int samples = 0;
        int64_t handed = 0;  // stored steps handed to the caller
        double xbytes = 0.0;  // bytes this rank received in the exchanges of the steps that count
        bool learning = compact && knobs.comm_compact_cap <= 0 && xcap_learned == 0;  // first chunk: short, whole-slice blocks
        HIP_TRY(hipEventRecord(ev_t0[0], stream));
        const auto tp1 = std::chrono::steady_clock::now();
        int64_t s0 = 0;  // first step of the chunk in hand
        while (s0 < total)
        {
            // ---- how far this chunk goes
            int64_t len = total - s0;
            if (compact)
            {
                const int64_t want = learning ? 16 : knobs.comm_compact_chunk;
                if (len > want) len = want;
            }
            if (any_rank_stores)
            {
                const int64_t fits = (s0 / interval + stage_slots) * (int64_t)interval - s0;  // steps until the staging buffer is full
                if (len > fits) len = fits;
            }
            if (compact)
            {
                // snapshot: everything a repeated chunk must find as this one found it
                T* cur = (full_fn && (run_step & 1)) ? d_pos_alt : d_pos;
                T* curl = (full_fn && (run_step & 1)) ? d_logp + W : d_logp;
                HIP_TRY(hipMemcpyAsync(d_snap, cur, step_bytes, hipMemcpyDeviceToDevice, stream));
                HIP_TRY(hipMemcpyAsync(d_snap + step_bytes, curl, sizeof(T) * (size_t)W, hipMemcpyDeviceToDevice, stream));
                HIP_TRY(hipMemcpyAsync(d_snap + step_bytes + sizeof(T) * (size_t)W, d_nacc, sizeof(uint32_t) * (size_t)W, hipMemcpyDeviceToDevice, stream));
                HIP_TRY(hipMemcpyAsync(d_snap + step_bytes + (sizeof(T) + sizeof(uint32_t)) * (size_t)W, d_diag, sizeof(Diag), hipMemcpyDeviceToDevice, stream));
                if ((rc = exchange_reset(cap))) return rc;
            }
            int unreduced = 0;
            int64_t staged = handed;
            const int samples_before = samples;
            for (int64_t s = s0; s < s0 + len; ++s)
            {
                const int parity = (int)(enq_step & 1), pos_parity = (int)(run_step & 1);
                const bool sample = samples < kMaxSamples && s % sample_stride == 0;
                const bool last_of_chunk = s + 1 == s0 + len;
                T* cur_pos = d_pos;  // the replica that holds the ensemble after this step
                if (full_fn)
                {
                    fill_red.draw_parity = parity;
                    launch_fill_draws(fill_red, red_base, nullptr, stream);
                    enqueue_step(parity, pos_parity);
                    // (the per-wavefront accepted counts are summed once per partial_slots steps, and at the end of a chunk)
                    if (++unreduced == partial_slots || last_of_chunk)
                    {
                        launch_accepted_reduce(d_partials, partial_slots, partial_waves, unreduced, ctl_after((int64_t)run_step + 1), d_run, stream, K);
                        unreduced = 0;
                    }
                    HIP_TRY(hipGetLastError());
                    cur_pos = pos_parity ? d_pos : d_pos_alt;
                    T* other_pos = pos_parity ? d_pos_alt : d_pos;
                    T* cur_logp = pos_parity ? d_logp : d_logp + W;
                    T* other_logp = pos_parity ? d_logp + W : d_logp;
                    if (sample) HIP_TRY(hipEventRecord(ev_x[2 * samples], stream));
                    rc = compact ? exchange_compact(cur_pos, other_pos, cur_logp, other_logp, 0, 2, cap) : exchange_rows(cur_pos, cur_logp, 0, 2);
                    if (rc) return rc;
                    if (sample) HIP_TRY(hipEventRecord(ev_x[2 * samples + 1], stream));
                    red_base = apply(half_jump, apply(half_jump, red_base));
                }
                else
                {
                    args_red.draw_parity = parity;
                    args_blk.draw_parity = parity;
                    half_fn(args_red, grid_blocks_for(args_red.shard_count), stream);
                    HIP_TRY(hipGetLastError());
                    if (sample) HIP_TRY(hipEventRecord(ev_x[2 * samples], stream));
                    rc = compact ? exchange_compact(d_pos, nullptr, d_logp, nullptr, 0, 1, cap) : exchange_rows(d_pos, nullptr, 0, 1);
                    if (rc) return rc;
                    if (sample) HIP_TRY(hipEventRecord(ev_x[2 * samples + 1], stream));
                    half_fn(args_blk, grid_blocks_for(args_blk.shard_count), stream);
                    if (++unreduced == partial_slots || last_of_chunk)
                    {
                        launch_accepted_reduce(d_partials, partial_slots, partial_waves, unreduced, ctl_after((int64_t)run_step + 1), d_run, stream, K);
                        unreduced = 0;
                    }
                    HIP_TRY(hipGetLastError());
                    rc = compact ? exchange_compact(d_pos, nullptr, d_logp, nullptr, 1, 1, cap) : exchange_rows(d_pos, nullptr, 1, 1);
                    if (rc) return rc;
                }
                if (sample) ++samples;
                enq_step += 1;
                run_step += 1;
                if (chain_out && (s + 1) % interval == 0)
                {
                    HIP_TRY(hipMemcpyAsync((char*)h_split_stage + step_bytes * (size_t)(staged - handed), cur_pos, step_bytes, hipMemcpyDeviceToHost, stream));
                    ++staged;
                }
            }
            // ---- end of the chunk
            const bool more = s0 + len < total;
            if (compact)
            {
                XStats* hx = reinterpret_cast<XStats*>((char*)h_pinned + 112);  // (behind the status words agree_on_status reads back)
                HIP_TRY(hipMemcpyAsync(hx, d_xstats, sizeof(XStats), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                if (hx->overflow)
                {
                    // Some block of some exchange of this chunk was too small: whatever the chunk computed rests on a
                    // replica that missed rows.  Back to the snapshot -- into both buffers, so that the repeated chunk may
                    // start at buffer 0 like a run does -- and once more with blocks nothing can overflow.  Every rank
                    // reads the same gathered headers, so every rank takes this branch together.
                    ++xchg_rollbacks;
                    HIP_TRY(hipMemcpyAsync(d_pos, d_snap, step_bytes, hipMemcpyDeviceToDevice, stream));
                    HIP_TRY(hipMemcpyAsync(d_logp, d_snap + step_bytes, sizeof(T) * (size_t)W, hipMemcpyDeviceToDevice, stream));
                    if (full_fn)
                    {
                        HIP_TRY(hipMemcpyAsync(d_pos_alt, d_snap, step_bytes, hipMemcpyDeviceToDevice, stream));
                        HIP_TRY(hipMemcpyAsync(d_logp + W, d_snap + step_bytes, sizeof(T) * (size_t)W, hipMemcpyDeviceToDevice, stream));
                    }
                    HIP_TRY(hipMemcpyAsync(d_nacc, d_snap + step_bytes + sizeof(T) * (size_t)W, sizeof(uint32_t) * (size_t)W, hipMemcpyDeviceToDevice, stream));
                    HIP_TRY(hipMemcpyAsync(d_diag, d_snap + step_bytes + (sizeof(T) + sizeof(uint32_t)) * (size_t)W, sizeof(Diag), hipMemcpyDeviceToDevice, stream));
                    half_steps = half_steps0 + 2 * (uint64_t)s0;
                    records_valid = false;
                    rc = write_ctl((uint64_t)s0, interval);
                    half_steps = half_steps0;
                    if (rc) return rc;
                    enq_step = (half_steps0 >> 1) + (uint64_t)s0;
                    run_step = 0;
                    red_base = apply(pcg_jump(inc, (unsigned __int128)3 * (unsigned)n * ((unsigned __int128)half_steps0 + 2 * (unsigned __int128)s0)), state0);
                    samples = samples_before;
                    cap = cap_full;
                    continue;  // (the same chunk again)
                }
                xbytes += (double)len * (double)(full_fn ? 1 : 2) * (double)(cfg.comm_world - 1) * (double)xblock_bytes<T>(cap, D);
                if (knobs.comm_compact_cap <= 0)
                {
                    // the next chunk's bound: what this one needed, plus an eighth and a little
                    uint64_t want = (uint64_t)hx->max_count + hx->max_count / 8 + 64;
                    want = (want + 63) & ~(uint64_t)63;
                    cap = want < cap_full ? (uint32_t)want : cap_full;
                    xcap_learned = cap;
                }
                else
                    cap = (uint32_t)(knobs.comm_compact_cap < (long)cap_full ? knobs.comm_compact_cap : (long)cap_full);
                learning = false;
            }
            else
                xbytes += (double)len * (double)(cfg.comm_world - 1) * (double)shard_count * 2.0 * (double)((size_t)D + (full_fn ? 1 : 0)) * sizeof(T);
            if (chain_out && staged > handed && (staged - handed == stage_slots || !more || compact))
            {
                HIP_TRY(hipStreamSynchronize(stream));
                std::memcpy((char*)chain_out + step_bytes * (size_t)handed, h_split_stage, step_bytes * (size_t)(staged - handed));
                handed = staged;
                publish_stored(handed);
            }
            s0 += len;
        }
        const auto tp2 = std::chrono::steady_clock::now();
        if (full_fn && (run_step & 1))
        {
            HIP_TRY(hipMemcpyAsync(d_pos, d_pos_alt, sizeof(T) * (size_t)W * D, hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemcpyAsync(d_logp, d_logp + W, sizeof(T) * (size_t)W, hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemcpyAsync(d_ctl, d_ctl + 1, sizeof(StepCtl), hipMemcpyDeviceToDevice, stream));
        }
        HIP_TRY(hipEventRecord(ev_t1[0], stream));
        // every rank ends the run with the whole ensemble's log-posteriors and accepted counters (get_state is then the
        // same on all ranks), and with the ensemble-wide accepted counts per step
        NCCL_TRY(rccl->GroupStart());
        for (int c = 0; c < 2; ++c)
        {
            if (!full_fn && !compact)  // (the exchanges of the other schemes carry the log-posteriors along)
                NCCL_TRY(rccl->AllGather(d_logp + (size_t)c * n + shard_begin, d_logp + (size_t)c * n, (size_t)shard_count, RcclType<T>::value, comm, stream));
            NCCL_TRY(rccl->AllGather(d_nacc + (size_t)c * n + shard_begin, d_nacc + (size_t)c * n, (size_t)shard_count, ncclUint32, comm, stream));
        }
        NCCL_TRY(rccl->AllReduce(d_acc, d_acc, (size_t)total, ncclUint32, ncclSum, comm, stream));
        NCCL_TRY(rccl->GroupEnd());
        HIP_TRY(hipStreamSynchronize(stream));
        {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ev_t0[0], ev_t1[0]));
            last_ms = ms;
            double sum = 0.0;
            for (int k = 0; k < samples; ++k)
            {
                HIP_TRY(hipEventElapsedTime(&ms, ev_x[2 * k], ev_x[2 * k + 1]));
                sum += ms;
            }
            // (half-step scheme: the sampled exchange is the red one, the black one moves as much)
            exchange_us_per_step = samples ? sum / samples * 1e3 * (full_fn ? 1.0 : 2.0) : 0.0;
        }
        xchg_bytes_per_step = xbytes / (double)total;
        xchg_cap_slots = compact ? (int64_t)cap : 0;
        last_launches = full_fn ? total : 2 * total;
        half_steps += 2 * (uint64_t)total;
        steps_since_reset += (uint64_t)total;
        records_valid = full_fn == nullptr;  // (full-step scheme: the next run re-primes, the red records of other ranks' walkers are per step anyway)
        records_step = half_steps >> 1;
        records_partner2 = false;
        if (accepted_per_step) HIP_TRY(hipMemcpy(accepted_per_step, d_acc, sizeof(uint32_t) * (size_t)total, hipMemcpyDeviceToHost));
        const auto tp3 = std::chrono::steady_clock::now();
        host_enqueue_ms = std::chrono::duration<double, std::milli>(tp2 - tp1).count();
        host_wall_ms = std::chrono::duration<double, std::milli>(tp3 - tp0).count();
        return MCMCPP_HIP_OK;
    }

    // The chain path of the full-step kernels.  Stored step k is complete in the pinned ring when ensemble step
    // (k + 2) * interval - 1 has finished (every launch forwards 1/interval of the previous stored step), and its
    // ring slot is overwritten from step (k + ring + 1) * interval on: the host enqueues chunks of steps, stays one
    // chunk ahead of the one it waits for, copies out whatever has become complete and never lets the launches
    // run into a slot it has not copied yet.  The run's last stored step has no launches behind it: it is copied
    // from the device ring at the end.
    int run_trickle(int64_t n_saved, int32_t interval, char* chain_out, bool want_accepted, size_t step_bytes, int64_t ring,
                    int64_t chunk_steps, double* launch_ms, char* direct_stage)
    {
        const int64_t total = n_saved * (int64_t)interval;
        const bool direct = direct_stage != nullptr;  // the launches forward into chain_out itself (pinned memory)
        // chain k: its own ring of stored steps on the device (and, unless the launches forward into chain_out itself,
        // its own twin in pinned memory); in the caller's memory chain k is the k-th run of n_saved steps
        const size_t ring_bytes = step_bytes * (size_t)ring, out_bytes = step_bytes * (size_t)n_saved;
        for (int k = 0; k < K; ++k)
        {
            RunInfo* ri = reinterpret_cast<RunInfo*>((char*)h_pinned + 1024 + 64 * kMaxChains + 64 * k);
            ri->chain = (char*)d_ring + ring_bytes * (size_t)k;
            ri->accepted_per_step = want_accepted ? d_acc + (size_t)k * (size_t)total : nullptr;
            ri->interval = interval;
            ri->chain_slot_base = 0;
            ri->stage = direct ? (void*)(direct_stage + out_bytes * (size_t)k) : (void*)((char*)h_ring + ring_bytes * (size_t)k);
            ri->slot_mask = ring - 1;
            ri->slice_bytes = (int64_t)(((step_bytes + (size_t)interval - 1) / (size_t)interval + 15) / 16 * 16) | (direct ? 1 : 0);
            ri->step_bytes = (int64_t)step_bytes;
            HIP_TRY(hipMemcpyAsync(run_of(k), ri, sizeof(RunInfo), hipMemcpyHostToDevice, stream));
        }

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
            if (direct)
            {
                if (complete > copied) copied = complete;  // (they are where they belong already)
            }
            else
                for (; copied < complete; ++copied)
                    for (int k = 0; k < K; ++k)
                    {
                        char* dst = chain_out + out_bytes * (size_t)k + step_bytes * (size_t)copied;
                        const char* src = (char*)h_ring + ring_bytes * (size_t)k + step_bytes * (size_t)(copied & (ring - 1));
                        if (enq == total)
                            parallel_memcpy(dst, src, step_bytes);  // nothing left to overlap with: be quick
                        else
                            std::memcpy(dst, src, step_bytes);
                    }
            publish_stored(copied);
            ++oldest;
            return MCMCPP_HIP_OK;
        };
        while (enq < total)
        {
            int64_t now = (total - enq < chunk_steps) ? total - enq : chunk_steps;
            // The stored steps that become complete with the LAST chunk are copied out with nothing left to overlap: the
            // run ends with a chunk of one interval, so that this tail is one stored step instead of a chunk's worth
            // (11.80 -> 11.55 ms per 2 000 steps at C2)
            if (!direct && now == total - enq && now > (int64_t)interval) now -= interval;
            // at most two chunks in flight, and no launch may forward into a ring slot that is still to be copied out
            // (forwarding into the final place needs no such care: a device slot is reused ring + 1 stored steps after it
            //  was written, its forwarding is over one stored step after)
            while (next_chunk > oldest && (next_chunk - oldest >= 2 || (!direct && enq + now > (copied + ring + 1) * (int64_t)interval)))
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
        for (int k = 0; k < K; ++k)
        {
            const size_t off = ring_bytes * (size_t)k + step_bytes * (size_t)((n_saved - 1) & (ring - 1));
            char* dst = direct ? chain_out + out_bytes * (size_t)k + step_bytes * (size_t)(n_saved - 1) : (char*)h_ring + off;
            HIP_TRY(hipMemcpyAsync(dst, (char*)d_ring + off, step_bytes, hipMemcpyDeviceToHost, stream));
        }
        while (next_chunk > oldest)
        {
            const int rc = process_oldest();
            if (rc) return rc;
        }
        HIP_TRY(hipStreamSynchronize(stream));
        if (direct)
            copied = n_saved;
        else
            for (; copied < n_saved; ++copied)  // (exactly one: every earlier one has been forwarded and copied above)
                for (int k = 0; k < K; ++k)
                {
                    const size_t off = ring_bytes * (size_t)k + step_bytes * (size_t)(copied & (ring - 1));
                    parallel_memcpy(chain_out + out_bytes * (size_t)k + step_bytes * (size_t)copied, (char*)h_ring + off, step_bytes);
                }
        publish_stored(copied);
        return MCMCPP_HIP_OK;
    }

    // Device-visible address of [p, p + bytes) when that range is pinned host memory (hipHostMalloc / mcmcpp_hip_host_alloc),
    // nullptr for pageable memory.
    void* device_view_of_pinned(void* p, size_t bytes)
    {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess || at.type != hipMemoryTypeHost || at.devicePointer == nullptr)
        {
            (void)hipGetLastError();  // (pageable memory is reported as an error)
            return nullptr;
        }
        hipPointerAttribute_t last;
        if (hipPointerGetAttributes(&last, (char*)p + bytes - 1) != hipSuccess || last.type != hipMemoryTypeHost)
        {
            (void)hipGetLastError();
            return nullptr;
        }
        return at.devicePointer;
    }

    int get_state(void* pos, void* logp, uint32_t* n_accept) override
    {
        if (!have_state) return fail(MCMCPP_HIP_E_STATE, "get_state: no walker state (set_state has not been called, or a run failed half way)");
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamSynchronize(stream));
        if (pos) HIP_TRY(hipMemcpy(pos, d_pos, sizeof(T) * (size_t)W * D * K, hipMemcpyDeviceToHost));
        for (int k = 0; k < K; ++k)
        {
            if (logp) HIP_TRY(hipMemcpy((T*)logp + (size_t)k * W, logp_of(k), sizeof(T) * (size_t)W, hipMemcpyDeviceToHost));
            if (n_accept) HIP_TRY(hipMemcpy(n_accept + (size_t)k * W, nacc_of(k), sizeof(uint32_t) * (size_t)W, hipMemcpyDeviceToHost));
        }
        if (n_accept)
            for (size_t w = 0; w < (size_t)W * K; ++w) n_accept[w] &= ~kRowMovedBit;  // (the top bit is the full-step kernels' bookkeeping)
        return MCMCPP_HIP_OK;
    }

    int seek(uint64_t steps_done) override
    {
        if (!have_state) return fail(MCMCPP_HIP_E_STATE, "seek: set_state has not been called");
        if (steps_done > (~0ULL >> 2)) return fail(MCMCPP_HIP_E_ARG, "seek: step count out of range");
        HIP_TRY(hipSetDevice(device));
        half_steps = 2 * steps_done;
        records_valid = false;
        records_ahead_of = kNoStep;
        return write_ctl(0);  // repositions the stream and re-primes the draw records of the next two half-steps
    }

    int reset_counters() override
    {
        HIP_TRY(hipSetDevice(device));
        for (int k = 0; k < K; ++k) HIP_TRY(hipMemsetAsync(nacc_of(k), 0, sizeof(uint32_t) * (size_t)W, stream));
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
            uint64_t s = 0;
            for (int k = 0; k < K; ++k)
            {
                HIP_TRY(hipMemcpy(a.data(), nacc_of(k), sizeof(uint32_t) * (size_t)W, hipMemcpyDeviceToHost));
                for (int c = 0; c < 2; ++c)
                    for (int i = 0; i < shard_count; ++i) s += a[(size_t)c * n + shard_begin + i] & ~kRowMovedBit;
            }
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
        if (K > 1) return fail(MCMCPP_HIP_E_UNSUPPORTED, "half_step_async: not with several chains per handle");
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

    // chain k's arrays (ChainGeometry; k = 0: the arrays themselves)
    T* logp_of(int k) const { return reinterpret_cast<T*>(reinterpret_cast<char*>(d_logp) + logp_chain_stride_bytes<T>(n) * (size_t)k); }
    uint32_t* nacc_of(int k) const { return reinterpret_cast<uint32_t*>(logp_of(k) + 2 * (size_t)W); }
    StepCtl* ctl_of(int k) const { return reinterpret_cast<StepCtl*>(reinterpret_cast<char*>(d_ctl) + (size_t)kCtlChainStride * (size_t)k); }
    RunInfo* run_of(int k) const { return reinterpret_cast<RunInfo*>(reinterpret_cast<char*>(d_run) + (size_t)kCtlChainStride * (size_t)k); }

    unsigned grid_blocks() const { return grid_blocks_for(shard_count); }
    unsigned grid_blocks_for(int count) const
    {
        const long per_wave = (long)(64 / step_lpw) * passes;
        const long waves = (count + per_wave - 1) / per_wave;
        return (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    }

    unsigned full_grid_blocks() const { return full_grid_blocks_for(shard_count); }
    unsigned full_grid_blocks_for(int count) const { return (unsigned)((count + full_wpb - 1) / full_wpb); }

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
        a.chains = (K == 1 && knobs.force_multi_chain_kernels != 0) ? -1 : K;
        // a fifth wavefront per workgroup computes the next draws when that is at most two rounds of 64 draws
        a.draw_wave = (3 * kWavesPerBlock * (64 / step_lpw) * passes <= 128 && knobs.no_draw_wave == 0) ? 1 : 0;
        return a;
    }

    // device StepCtl[0] <- {stream position of half-step `half_steps`, counters}; half_steps must be even
    // step_in_run > 0 (a chunk of a split run being repeated): the counters the kernels keep instead of dividing follow
    int write_ctl(uint64_t step_in_run, int32_t interval = 1)
    {
        const Affine128 j = pcg_jump(inc, (unsigned __int128)3 * (unsigned)n * (unsigned __int128)half_steps);
        // the draw records of the next red and the next black half-step (afterwards the launches keep them going):
        // unless the launches of the previous call left exactly these behind
        const bool refill = !(records_valid && records_step == (half_steps >> 1) && (!full_fn || records_partner2));
        for (int k = 0; k < K; ++k)
        {
            StepCtl* c = K > 1 ? (StepCtl*)((char*)h_pinned + 1024 + 64 * k) : (StepCtl*)((char*)h_pinned + 128);
            c->state = apply(j, state0_of[k]);
            const U128 state1 = apply(half_jump, c->state);
            c->state2 = apply(half_jump, state1);
            c->half_step = half_steps;
            c->step_in_run = step_in_run;
            c->chain_slot = (long long)(step_in_run / (uint64_t)interval);
            c->save_phase = (uint32_t)(step_in_run % (uint64_t)interval);
            c->partial_slot = (uint32_t)(step_in_run % (uint64_t)partial_slots);
            HIP_TRY(hipMemcpyAsync(ctl_of(k) + (half_steps & 1), c, sizeof(StepCtl), hipMemcpyHostToDevice, stream));
            if (d_ctl_keep) HIP_TRY(hipMemcpyAsync(d_ctl_keep, c, sizeof(StepCtl), hipMemcpyHostToDevice, stream));
            if (refill)
            {
                const int parity = (int)((half_steps >> 1) & 1);  // the buffer the coming ensemble step reads
                HalfStepArgs<T> fr = make_args(0, parity), fb = make_args(1, parity);
                fr.draws = fb.draws = d_draws + (size_t)k * 4 * (size_t)n;  // (the chain's own records; the tables are shared)
                launch_fill_draws(fr, c->state, nullptr, stream);
                launch_fill_draws(fb, state1, full_fn ? &c->state : nullptr, stream);
                HIP_TRY(hipGetLastError());
            }
        }
        if (refill)
        {
            records_valid = true;
            records_step = half_steps >> 1;
            records_partner2 = full_fn != nullptr;
        }
        HIP_TRY(hipStreamSynchronize(stream));
        return MCMCPP_HIP_OK;
    }

    // pos_parity: which position buffer a full-step launch reads (ensemble steps enqueued in this run() & 1)
    // batch_slot >= 0: the step's draw records are record set batch_slot of d_draws_batch, made ahead (fill_batch)
    void enqueue_step(int parity, int pos_parity, int batch_slot = -1)
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
            a.draw_wave = (3 * (kFullDrawWaves == 4 ? (full_wpb + 1) / 2 : full_wpb) <= 64 && knobs.no_draw_wave == 0) ? 1 : 0;
            if (batch_slot >= 0)
            {
                a.draw_wave = 2;
                a.draw_parity = 0;
                a.draws = d_draws_batch + (size_t)batch_slot * 2 * (size_t)n;
            }
            full_fn(a, full_grid_blocks_for(a.shard_count), stream);
            a.draws = d_draws;
            return;
        }
        args_red.draw_parity = parity;
        args_blk.draw_parity = parity;
        half_fn(args_red, grid_blocks_for(args_red.shard_count), stream);
        half_fn(args_blk, grid_blocks_for(args_blk.shard_count), stream);
    }

    // the draw records of `count` ensemble steps from the one that reads position buffer pos_parity on (its control
    // record is what the launch before it left behind, or write_ctl)
    void fill_batch(int pos_parity, int count)
    {
        HalfStepArgs<T> a = args_red;
        a.draws = d_draws;
        launch_fill_draws_batch(a, d_ctl + pos_parity, d_step_jump, d_draws_batch, count, stream);
    }

    // record sets [first, first + count) for the same steps of the NEXT replay, on the parallel branch (stream capture:
    // the fork is the event, the join comes before the accepted-count reduction)
    int fork_fill_of_next_replay(int first, int count)
    {
        HalfStepArgs<T> a = args_red;
        a.draws = d_draws;
        HIP_TRY(hipEventRecord(ev_fill_fork, stream));
        HIP_TRY(hipStreamWaitEvent(fill_stream, ev_fill_fork, 0));
        launch_fill_draws_batch(a, d_ctl_keep, d_step_jump + batch_draws + first, d_draws_batch + (size_t)first * 2 * (size_t)n, count, fill_stream);
        return MCMCPP_HIP_OK;
    }

    // `steps` ensemble steps from record-buffer parity start_parity / position-buffer parity pos_parity on
    // records_ahead: the batch records of these steps are in place (the replay before made them on its branch);
    // fill_next: make the next replay's on a branch of this one
    int enqueue_step_sequence(int steps, int start_parity, int pos_parity, bool records_ahead = false, bool fill_next = false)
    {
        int piece_first = 0, piece = 0;
        for (int s = 0; s < steps; ++s)
        {
            if (batch_draws > 0)
            {
                if (s % batch_draws == 0 && !(s == 0 && records_ahead)) fill_batch((pos_parity + s) & 1, steps - s < batch_draws ? steps - s : batch_draws);
                enqueue_step(0, (pos_parity + s) & 1, s % batch_draws);
                if (fill_next && (s + 1 == (int)(((int64_t)steps * (piece + 1)) / fill_pieces)))
                {
                    if (int rc = fork_fill_of_next_replay(piece_first, s + 1 - piece_first)) return rc;
                    piece_first = s + 1;
                    ++piece;
                }
            }
            else
                enqueue_step((start_parity + s) & 1, (pos_parity + s) & 1);
        }
        if (fill_next)
        {
            HIP_TRY(hipEventRecord(ev_fill_join, fill_stream));
            HIP_TRY(hipStreamWaitEvent(stream, ev_fill_join, 0));
        }
        return MCMCPP_HIP_OK;
    }

    // hipGraph of `steps` ensemble steps followed by the accepted-count reduction (cached per step count:
    // graph_steps for the bulk, one graph per distinct remainder)
    // (the record-buffer parity of every node is frozen into the graph, hence one graph per starting parity)
    // (with records made a replay ahead: a replay of graph_steps steps always makes the next replay's on its branch, and
    //  one graph per "the records of my own steps are in place already")
    int graph_for(int steps, int start_parity, int pos_parity, hipGraphExec_t* out, bool records_ahead = false)
    {
        const size_t key = ((size_t)steps * 2 + (records_ahead ? 1 : 0)) * 4 + (size_t)start_parity * 2 + (size_t)pos_parity;
        if (graph_cache.size() <= key) graph_cache.resize(key + 1, nullptr);
        if (!graph_cache[key])
        {
            hipGraph_t g = nullptr;
            HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
            if (int rc = enqueue_step_sequence(steps, start_parity, pos_parity, records_ahead, fill_pieces > 0 && steps == graph_steps)) return rc;
            launch_accepted_reduce(d_partials, partial_slots, partial_waves, steps, ctl_after(pos_parity + steps), d_run, stream, K, d_ctl_keep);
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
        return graph_for(graph_steps, (int)(enq_step & 1), 0, &ex, fill_pieces > 0 && records_ahead_of == enq_step);
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
                int rc = graph_for(graph_steps, (int)(enq_step & 1), (int)(run_step & 1), &ex, fill_pieces > 0 && records_ahead_of == enq_step);
                if (rc) return rc;
                HIP_TRY(hipGraphLaunch(ex, stream));
                left -= graph_steps;
                enq_step += (uint64_t)graph_steps;
                run_step += (uint64_t)graph_steps;
                records_ahead_of = fill_pieces > 0 ? enq_step : kNoStep;
            }
            if (left > 0)
            {
                // one replay for the remainder (it may use records made ahead, it makes none: the record sets would no
                // longer line up with the steps of a whole replay)
                int rc = graph_for((int)left, (int)(enq_step & 1), (int)(run_step & 1), &ex, fill_pieces > 0 && records_ahead_of == enq_step);
                if (rc) return rc;
                HIP_TRY(hipGraphLaunch(ex, stream));
                enq_step += (uint64_t)left;
                run_step += (uint64_t)left;
                records_ahead_of = kNoStep;
            }
        }
        else
        {
            for (; left > 0; --left)
            {
                (void)enqueue_step_sequence(1, (int)(enq_step & 1), (int)(run_step & 1));
                launch_accepted_reduce(d_partials, partial_slots, partial_waves, 1, ctl_after((int64_t)run_step + 1), d_run, stream, K);
                enq_step += 1;
                run_step += 1;
            }
            HIP_TRY(hipGetLastError());
        }
        return MCMCPP_HIP_OK;
    }

    // persistent per-run buffers, grown on demand: per-step accepted counters, the two halves of the device
    // chain and their pinned staging twins
    int ensure_run_buffers(size_t acc_entries, size_t half_bytes, size_t ring_bytes, bool need_host_ring = true)
    {
        if (ring_bytes > ring_capacity)
        {
            HIP_TRY(hipStreamSynchronize(stream));
            if (d_ring) hipFree(d_ring);
            d_ring = nullptr;
            ring_capacity = 0;
            if (hipMalloc(&d_ring, ring_bytes) != hipSuccess) return fail(MCMCPP_HIP_E_NOMEM, "run: cannot allocate %zu bytes of device chain", ring_bytes);
            ring_capacity = ring_bytes;
        }
        if (need_host_ring && ring_bytes > host_ring_capacity)
        {
            HIP_TRY(hipStreamSynchronize(stream));
            if (h_ring) hipHostFree(h_ring);
            h_ring = nullptr;
            host_ring_capacity = 0;
            if (hipHostMalloc(&h_ring, ring_bytes, hipHostMallocDefault) != hipSuccess)
                return fail(MCMCPP_HIP_E_NOMEM, "run: cannot allocate %zu bytes of pinned staging", ring_bytes);
            host_ring_capacity = ring_bytes;
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
            if (knobs.copy_stream != 0) HIP_TRY(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
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
        if (d_draws_batch) hipFree(d_draws_batch);
        if (d_step_jump) hipFree(d_step_jump);
        if (d_ctl_keep) hipFree(d_ctl_keep);
        if (fill_stream) hipStreamDestroy(fill_stream);
        if (ev_fill_fork) hipEventDestroy(ev_fill_fork);
        if (ev_fill_join) hipEventDestroy(ev_fill_join);
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
        if (d_xblocks) hipFree(d_xblocks);
        if (d_seen) hipFree(d_seen);
        if (d_xstats) hipFree(d_xstats);
        if (d_snap) hipFree(d_snap);
        if (own_comm && comm && rccl) (void)rccl->CommDestroy(comm);
        if (h_split_stage) hipHostFree(h_split_stage);
        for (hipEvent_t e : ev_x)
            if (e) hipEventDestroy(e);
        if (own_stream && stream) hipStreamDestroy(stream);
    }

    mcmcpp_hip_config cfg;
    Knobs knobs;
    const Rccl* rccl = nullptr;  // split ensembles only
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    // split ensembles of more than one rank, exchanging moved rows only (exchange_kernels.hpp)
    char* d_xblocks = nullptr;      // [comm_world][block]: this rank's block and, behind the all-gather, everybody's
    uint32_t* d_seen = nullptr;     // [W]: a walker's accepted counter as of the last exchange (own slice)
    XStats* d_xstats = nullptr;
    char* d_snap = nullptr;         // positions, log-posteriors, counters, diagnostics in front of the chunk in hand
    uint32_t xcap_learned = 0;      // the slot bound the last run ended with
    void* h_split_stage = nullptr;  // split ensembles: pinned staging of stored steps
    size_t split_stage_capacity = 0;
    std::vector<hipEvent_t> ev_x;   // split ensembles: events around a sample of exchanges
    const LaunchTable<T>* table = nullptr;
    typename LaunchTable<T>::HalfStepFn half_fn = nullptr;
    typename LaunchTable<T>::HalfStepFn full_fn = nullptr;  // non-null: run() steps with one launch per ensemble step
    int full_wpb = 1;                                       // walkers of each colour per full-step workgroup
    T* d_pos_alt = nullptr;
    uint64_t run_step = 0;                                  // ensemble steps enqueued in the current run()
    typename LaunchTable<T>::CalcFn calc_fn = nullptr;
    int W = 0, D = 0, n = 0, lpw = 1, epl = 1, step_lpw = 1, passes = 1, vec_ok = 0, num_cus = 256;
    int shard_begin = 0, shard_count = 0, device = -1, graph_steps = 32;
    size_t chain_subchunk_bytes = 0, chain_half_capacity = 0, acc_capacity = 0;
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_filled[2] = {nullptr, nullptr};
    void* arena = nullptr;  // one device allocation holding everything a step launch touches (see carve)
    size_t arena_bytes = 0, arena_used = 0;
    void *d_ring = nullptr, *h_ring = nullptr;  // full-step chain path: device ring of stored steps and its pinned host twin
    size_t ring_capacity = 0, host_ring_capacity = 0;
    hipStream_t copy_stream = nullptr;  // experiment (MCMCPP_HIP_COPY_STREAM=1): chain downloads beside the launches
    T* d_chain[2] = {nullptr, nullptr};
    void* h_stage[2] = {nullptr, nullptr};
    uint32_t* d_acc = nullptr;
    hipStream_t stream = nullptr;
    bool own_stream = false, own_pos = false, have_state = false, stream_valid = false, run_touched_device = false;
    hipEvent_t ev_t0[4] = {nullptr, nullptr, nullptr, nullptr}, ev_t1[4] = {nullptr, nullptr, nullptr, nullptr};
    T *d_pos = nullptr, *d_logp = nullptr, *d_params = nullptr, *d_params_padded = nullptr;
    uint32_t* d_nacc = nullptr;
    StepCtl* d_ctl = nullptr;
    RunInfo* d_run = nullptr;
    Diag* d_diag = nullptr;
    uint64_t* d_status = nullptr;
    DrawRec<T>* d_draws = nullptr;
    DrawRec<T>* d_draws_batch = nullptr;  // [batch_draws][2][n]: records made ahead of the matrix-core full-step launches
    Affine128* d_step_jump = nullptr;     // [batch_draws]
    int batch_draws = 0;                  // 0: the step launches make their own next records
    int fill_pieces = 0;                  // > 0: a whole replay makes the next replay's records on a parallel branch, in that many pieces
    hipStream_t fill_stream = nullptr;    // (capture only: the branch)
    hipEvent_t ev_fill_fork = nullptr, ev_fill_join = nullptr;
    StepCtl* d_ctl_keep = nullptr;        // the control record at the head of the coming replay (write_ctl, accepted_reduce_kernel)
    static constexpr uint64_t kNoStep = ~0ULL;
    uint64_t records_ahead_of = kNoStep;  // the ensemble step (since set_state) whose records sit in record set 0, made ahead by a branch
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
    U128 state0_of[kMaxChains];  // per chain (seed + k)
    int K = 1;                   // independent ensembles stepped by one launch
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
    if (c->comm_world < 0) BAD("comm_world must not be negative");
    if (c->num_chains < 0) BAD("num_chains must not be negative");
    if (c->num_chains > 1 && c->mover != MCMCPP_HIP_MOVER_STRETCH) BAD("several chains per handle: StretchMove only");
    if (c->mover == MCMCPP_HIP_MOVER_DIFFERENTIAL_EVOLUTION && (c->shard_count != 0 || c->device_positions || c->comm_world >= 1))
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

void mcmcpp_hip_destroy(mcmcpp_hip_sampler* h)
{
    if (!h) return;
    // an asynchronous run still uses the arena, streams, graphs and the communicator the destructor releases: let it end first
    if (h->async_worker.joinable()) h->async_worker.join();
    h->async_active = false;
    delete h;
}

const char* mcmcpp_hip_last_error(const mcmcpp_hip_sampler* h)
{
    if (!h) return g_create_error.c_str();
    return h->refused ? h->refused : h->error.c_str();
}

#define NEED_H \
    if (!h) return MCMCPP_HIP_E_ARG
// between run_async and run_wait the handle belongs to its worker thread: only wait_stored (and destroy, which joins) may be called
#define NOT_WHILE_ASYNC(name) \
    if (h->async_active) return h->refuse("mcmcpp_hip_" name ": an asynchronous run is in progress (only wait_stored may be called before run_wait)"); \
    h->refused = nullptr

int mcmcpp_hip_set_state(mcmcpp_hip_sampler* h, const void* positions, const void* logp)
{
    NEED_H;
    NOT_WHILE_ASYNC("set_state");
    return h->set_state(positions, logp);
}
int mcmcpp_hip_run(mcmcpp_hip_sampler* h, int64_t n_saved, int32_t interval, void* chain_out, uint32_t* accepted_per_step)
{
    NEED_H;
    NOT_WHILE_ASYNC("run");
    return h->run(n_saved, interval, chain_out, accepted_per_step);
}
int mcmcpp_hip_run_async(mcmcpp_hip_sampler* h, int64_t n_saved, int32_t interval, void* chain_out, uint32_t* accepted_per_step)
{
    NEED_H;
    if (h->async_active) return h->refuse("mcmcpp_hip_run_async: the previous asynchronous run has not been waited for (mcmcpp_hip_run_wait)");
    h->refused = nullptr;
    {
        std::lock_guard<std::mutex> lock(h->async_mutex);
        h->async_stored = 0;
        h->async_done = false;
        h->async_rc = MCMCPP_HIP_OK;
    }
    h->async_active = true;
    try
    {
        h->async_worker = std::thread([=]() {
            const int rc = h->run(n_saved, interval, chain_out, accepted_per_step);
            {
                std::lock_guard<std::mutex> lock(h->async_mutex);
                h->async_rc = rc;
                h->async_done = true;
                if (rc == MCMCPP_HIP_OK && chain_out) h->async_stored = n_saved;  // (whatever the sampler announced on the way)
            }
            h->async_cv.notify_all();
        });
    }
    catch (...)
    {
        h->async_active = false;
        h->async_done = true;
        return h->fail(MCMCPP_HIP_E_NOMEM, "run_async: cannot start the worker thread");
    }
    return MCMCPP_HIP_OK;
}
int mcmcpp_hip_wait_stored(mcmcpp_hip_sampler* h, int64_t count)
{
    NEED_H;
    if (!h->async_active) return h->fail(MCMCPP_HIP_E_STATE, "wait_stored: no asynchronous run in progress");
    std::unique_lock<std::mutex> lock(h->async_mutex);
    h->async_cv.wait(lock, [&]() { return h->async_stored >= count || h->async_done; });
    if (h->async_stored >= count) return MCMCPP_HIP_OK;
    return h->async_rc != MCMCPP_HIP_OK ? h->async_rc : MCMCPP_HIP_E_ARG;  // the run ended without storing that many steps
}
int mcmcpp_hip_run_wait(mcmcpp_hip_sampler* h)
{
    NEED_H;
    if (!h->async_active) return h->fail(MCMCPP_HIP_E_STATE, "run_wait: no asynchronous run in progress");
    if (h->async_worker.joinable()) h->async_worker.join();
    h->async_active = false;
    h->refused = nullptr;
    return h->async_rc;
}
void* mcmcpp_hip_host_alloc(uint64_t bytes)
{
    void* p = nullptr;
    // (portable: usable from every device of the process, whichever one happens to be current on the calling thread --
    //  the facade's Chain obtains blocks on a helper thread that has never selected a device)
    if (hipHostMalloc(&p, bytes ? (size_t)bytes : 64, hipHostMallocPortable) != hipSuccess)
    {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void mcmcpp_hip_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}
int mcmcpp_hip_get_state(mcmcpp_hip_sampler* h, void* positions, void* logp, uint32_t* n_accept)
{
    NEED_H;
    NOT_WHILE_ASYNC("get_state");
    return h->get_state(positions, logp, n_accept);
}
int mcmcpp_hip_seek(mcmcpp_hip_sampler* h, uint64_t ensemble_steps_done)
{
    NEED_H;
    NOT_WHILE_ASYNC("seek");
    return h->seek(ensemble_steps_done);
}
int mcmcpp_hip_reset_counters(mcmcpp_hip_sampler* h)
{
    NEED_H;
    NOT_WHILE_ASYNC("reset_counters");
    return h->reset_counters();
}
int mcmcpp_hip_get_counters(mcmcpp_hip_sampler* h, uint64_t* accepted, uint64_t* ensemble_steps, uint64_t* near_ties,
                            uint64_t* redraws)
{
    NEED_H;
    NOT_WHILE_ASYNC("get_counters");
    return h->get_counters(accepted, ensemble_steps, near_ties, redraws);
}
int mcmcpp_hip_calc_logp(mcmcpp_hip_sampler* h, const void* positions, int64_t count, void* logp_out)
{
    NEED_H;
    NOT_WHILE_ASYNC("calc_logp");
    return h->calc_logp(positions, count, logp_out);
}
int mcmcpp_hip_last_run_timing(mcmcpp_hip_sampler* h, double* gpu_ms, int64_t* step_launches)
{
    NEED_H;
    return h->last_run_timing(gpu_ms, step_launches);
}
int mcmcpp_hip_last_run_host_timing(mcmcpp_hip_sampler* h, double* enqueue_ms, double* wall_ms, double* exchange_us_per_step)
{
    NEED_H;
    if (enqueue_ms) *enqueue_ms = h->host_enqueue_ms;
    if (wall_ms) *wall_ms = h->host_wall_ms;
    if (exchange_us_per_step) *exchange_us_per_step = h->exchange_us_per_step;
    return MCMCPP_HIP_OK;
}
int mcmcpp_hip_last_run_exchange(mcmcpp_hip_sampler* h, double* bytes_per_step, int64_t* repeated_chunks, int64_t* block_slots)
{
    NEED_H;
    if (bytes_per_step) *bytes_per_step = h->xchg_bytes_per_step;
    if (repeated_chunks) *repeated_chunks = h->xchg_rollbacks;
    if (block_slots) *block_slots = h->xchg_cap_slots;
    return MCMCPP_HIP_OK;
}
int mcmcpp_hip_comm_unique_id(void* id_out)
{
    if (!id_out)
    {
        g_create_error = "comm_unique_id: id_out is NULL";
        return MCMCPP_HIP_E_ARG;
    }
    std::string why;
    const Rccl* r = Rccl::get(&why);
    if (!r)
    {
        g_create_error = why;
        return MCMCPP_HIP_E_COMM;
    }
    ncclUniqueId id;
    const ncclResult_t rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess)
    {
        g_create_error = std::string("ncclGetUniqueId failed: ") + r->GetErrorString(rc);
        return MCMCPP_HIP_E_COMM;
    }
    std::memcpy(id_out, &id, sizeof id);
    return MCMCPP_HIP_OK;
}
int mcmcpp_hip_half_step_async(mcmcpp_hip_sampler* h, int32_t color, int64_t save_slot)
{
    NEED_H;
    NOT_WHILE_ASYNC("half_step_async");
    return h->half_step_async(color, save_slot);
}
int mcmcpp_hip_bind_device_chain(mcmcpp_hip_sampler* h, void* device_chain, int64_t slots)
{
    NEED_H;
    NOT_WHILE_ASYNC("bind_device_chain");
    return h->bind_device_chain(device_chain, slots);
}
void* mcmcpp_hip_device_positions(mcmcpp_hip_sampler* h) { return h ? h->device_positions() : nullptr; }
int mcmcpp_hip_shard_span(mcmcpp_hip_sampler* h, int32_t color, int64_t* offset_elems, int64_t* count_elems)
{
    NEED_H;
    NOT_WHILE_ASYNC("shard_span");
    return h->shard_span(color, offset_elems, count_elems);
}
int mcmcpp_hip_synchronize(mcmcpp_hip_sampler* h)
{
    NEED_H;
    NOT_WHILE_ASYNC("synchronize");
    return h->synchronize();
}
}

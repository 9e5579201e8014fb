// stretch_kernel.hpp -- the stretch-move half-step kernel for gfx950 (MI355X).
//
// One launch = one half of EnsembleSampler::performStep (MCMCpp/EnsembleSampler.h:345-354; threaded
// twin Threading/RedBlkUpdater.h:125-147): every walker of one colour performs
// StretchMove::updateWalker (MCMCpp/Movers/StretchMove.h:100-123) against the other colour as it stands.
// The kernel boundary between the red and the black launch is the reference's mid-step barrier
// (Threading/RedBlkCtrlerSpinLock.h:240-254).
//
// Mapping.  A walker's D-vector is spread over LPW lanes x EPL elements (16 bytes per lane at the
// base EPL), so a wavefront reads and writes whole walker rows with fully coalesced 16-byte accesses
// (walker-major, parameter-contiguous rows, as the reference lays them out).  A wavefront owns
// NW = (64/LPW)*passes consecutive walkers:
//   phase A (one walker per lane, lanes 0..NW-1): jump the pcg64 stream to the walker's three draws,
//            partner index, stretch factor z, (D-1) ln z, ln U, current log-posterior -> wave-private LDS
//   phase B (LPW lanes per walker, `passes` rounds of 64/LPW walkers): gather the partner row, form the
//            proposal, evaluate the Calculator functor (cross-lane tree reduction), Metropolis accept
//            in place, optional chain store, per-step accepted count
// No MFMA: the work is element-wise plus a per-walker reduction.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "calculators.hpp"
#include "fast_log.hpp"
#include "pcg128.hpp"

namespace mcmcpp
{

// Device-resident record carrying the random stream and step counters from one half-step launch to the
// next (two records, ping-pong: the red launch reads [0] and writes [1], the black launch the reverse),
// so that graph replays need no host-side argument updates.
struct StepCtl
{
    U128 state;            // engine state before the first draw of this half-step
    uint64_t half_step;    // half-steps executed since set_state
    uint64_t step_in_run;  // ensemble step index inside the current run() call
    long long chain_slot;  // stored steps written so far in this run (slot of the next one)
    uint32_t save_phase;   // step_in_run mod interval, kept as a counter (no division in the kernel)
    uint32_t partial_slot; // step_in_run mod partial_slots, likewise
};

// Per-run constants, written by the host before the first launch of a run.
struct RunInfo
{
    void* chain;                 // device chain buffer [slots][W][D] or nullptr
    uint32_t* accepted_per_step; // device counters [steps of this run] or nullptr
    int64_t interval;            // store the last step of every `interval`
    int64_t chain_slot_base;     // slot of the first stored step of this run
};

struct Diag
{
    unsigned long long near_ties;
    unsigned long long redraws;
};

template <class T>
struct HalfStepArgs
{
    T* pos;                  // [W][D]
    T* logp;                 // [W]
    uint32_t* n_accept;      // [W]
    const StepCtl* ctl_in;
    StepCtl* ctl_out;
    const RunInfo* run;
    Diag* diag;
    const Affine128* jump_lo; // [256]   map of 3*k draws
    const Affine128* jump_hi; // [ceil(n/256)] map of 3*256*m draws
    const Affine128* task_jump; // [3n] map of t+1 draws (base state -> state behind draw t), or nullptr for large n
    const T* calc_params;
    Affine128 half_jump;      // map of 3*n draws: this half-step's base state -> the next one's
    Affine128 draw_jump[3];   // maps of 1, 2, 3 draws: a walker's base state -> the state behind draw k
    U128 inc;                 // pcg stream increment
    uint64_t redraw_threshold; // (2^64 - n) mod n (pcg bounded_rand)
    T gw_term1, gw_inv_sqrt;  // GwDistribution<T,2,1> constants (MCMCpp/Utility/GwDistribution.h:45-55)
    T dims_minus_one;         // (T)(D-1)  (StretchMove.h:110)
    T tie_eps;
    int n;                    // walkers per half
    int dims;                 // D
    int color;                // 0 red = walkers [0,n), 1 black = [n,2n)
    int shard_begin;          // first walker (index inside the half) updated by this launch
    int shard_count;          // number of walkers updated by this launch
    int passes;               // rounds of 64/LPW walkers per wavefront
    int vec_ok;               // rows are 16-byte aligned multiples: use 128-bit accesses
    int n_is_pow2;
    uint32_t* partials;         // [partial_slots][2][partial_waves] per-wavefront accepted counts, or nullptr
    int partial_slots;          // ensemble steps between two runs of accepted_reduce_kernel
    int partial_waves;          // wavefronts of one half-step launch
    unsigned long long* stamps; // diagnostic build only (MCMCPP_STAMPS): 8 shader-clock stamps of wavefront 0
    long long direct_save_slot; // >= 0: store into run->chain at this slot regardless of interval (sharded driver)
    int use_ctl_save;           // 1: saving follows RunInfo.interval / StepCtl.step_in_run
};

template <class T, int EPL>
__device__ __forceinline__ void load_slice(const T* row, int i0, int D, bool vec_ok, bool active, T (&out)[EPL])
{
    constexpr int VN = Vec16<T>::N;
    typedef typename Vec16<T>::type V;
#pragma unroll
    for (int e = 0; e < EPL; ++e) out[e] = (T)0;
    if (!active) return;
    if (vec_ok)
    {
#pragma unroll
        for (int v = 0; v < EPL / VN; ++v)
        {
            if (i0 + v * VN < D)
            {
                const V x = *reinterpret_cast<const V*>(row + i0 + v * VN);
                const T* xs = reinterpret_cast<const T*>(&x);
#pragma unroll
                for (int k = 0; k < VN; ++k) out[v * VN + k] = xs[k];
            }
        }
    }
    else
    {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
            if (i0 + e < D) out[e] = row[i0 + e];
    }
}

template <class T, int EPL>
__device__ __forceinline__ void store_slice(T* row, int i0, int D, bool vec_ok, const T (&val)[EPL])
{
    constexpr int VN = Vec16<T>::N;
    typedef typename Vec16<T>::type V;
    if (vec_ok)
    {
#pragma unroll
        for (int v = 0; v < EPL / VN; ++v)
        {
            if (i0 + v * VN < D)
            {
                V x;
                T* xs = reinterpret_cast<T*>(&x);
#pragma unroll
                for (int k = 0; k < VN; ++k) xs[k] = val[v * VN + k];
                *reinterpret_cast<V*>(row + i0 + v * VN) = x;
            }
        }
    }
    else
    {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
            if (i0 + e < D) row[i0 + e] = val[e];
    }
}

__device__ __forceinline__ double dev_log(double x) { return fast_log(x); }
__device__ __forceinline__ float dev_log(float x) { return logf(x); }
__device__ __forceinline__ double dev_abs(double x) { return fabs(x); }
__device__ __forceinline__ float dev_abs(float x) { return fabsf(x); }

// wave-private LDS record of phase A
template <class T>
struct PhaseA
{
    T z[64];
    T zs[64];
    T ln_u[64];
    uint32_t partner[64];
};

constexpr int kWavesPerBlock = 4;

// Diagnostic build only (make STAMPS=1 -> libmcmcpp_hip_stamps.so): wavefront 0 of workgroup 0 drains its
// memory counters and records the shader clock at a few points; the product build compiles none of it.
#ifdef MCMCPP_STAMPS
#define MCMCPP_STAMP(k)                                                                     \
    do                                                                                      \
    {                                                                                       \
        if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x < 64)                     \
        {                                                                                   \
            unsigned long long t_;                                                          \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            if (threadIdx.x == 0) stamp_val[k] = t_;                                        \
        }                                                                                   \
    } while (0)
#else
#define MCMCPP_STAMP(k) \
    do                  \
    {                   \
    } while (0)
#endif

// dynamic LDS of one workgroup: [phase-A records][proposal stage (if the calculator wants it)][calculator tables]
template <class T, class Calc, int EPL>
struct LdsLayout
{
    __host__ __device__ static constexpr size_t stage_offset() { return kWavesPerBlock * sizeof(PhaseA<T>); }
    __host__ __device__ static constexpr size_t block_offset()
    {
        return stage_offset() + (Calc::kNeedsStage ? (size_t)kWavesPerBlock * 64 * EPL * sizeof(T) : 0);
    }
    __host__ static size_t bytes(int dims) { return block_offset() + Calc::block_scratch_elems(dims) * sizeof(T); }
};
static_assert(sizeof(PhaseA<double>) % 16 == 0 && sizeof(PhaseA<float>) % 16 == 0, "LDS pieces must stay 16-byte aligned");

template <class T, class Calc, int EPL, int LPW>
__global__ void __launch_bounds__(64 * kWavesPerBlock) stretch_half_step_kernel(const HalfStepArgs<T> a)
{
    static_assert((LPW & (LPW - 1)) == 0 && LPW >= 1 && LPW <= 64, "LPW must be a power of two <= 64");
    static_assert(EPL % Vec16<T>::N == 0, "EPL must be a whole number of 16-byte vectors");
    constexpr int WPP = 64 / LPW;  // walkers per pass

    // LDS carve-up (all dynamic, 16-byte aligned pieces): phase-A records | proposal stage | calculator tables
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PhaseA<T>* sh_a = reinterpret_cast<PhaseA<T>*>(smem);
    T* sh_stage = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::stage_offset());
    T* sh_block = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::block_offset());

#ifdef MCMCPP_STAMPS
    unsigned long long stamp_val[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    MCMCPP_STAMP(0);
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int wave = blockIdx.x * kWavesPerBlock + wib;
    const int nw = WPP * a.passes;
    const int first = wave * nw;  // first walker of this wavefront, relative to the shard
    const bool wave_active = first < a.shard_count;
    const int half_base = a.color ? a.n : 0;
    const int other_base = a.color ? 0 : a.n;
    const int sub = lane & (LPW - 1);
    const int grp = lane / LPW;
    const int i0 = sub * EPL;
    const bool vec_ok = a.vec_ok != 0;
    const int last_li = a.shard_count - 1;

    // ---- every load that does not depend on the random draws is issued up front, so that the launch pays
    //      one memory round trip for all of them: control record, run constants, jump-table entries,
    //      current log-posterior and counter, the first pass's own rows and the calculator's tables
    const StepCtl ctl = *a.ctl_in;  // wave-uniform
    const RunInfo run = *a.run;

    GroupCtx<T, EPL, LPW> ctx;
    ctx.sub = sub;
    ctx.dims = a.dims;
    ctx.lane = lane;
    ctx.stage = Calc::kNeedsStage ? &sh_stage[wib * 64 * EPL] : nullptr;
    ctx.vec_ok = vec_ok;

    // phase-A work is spread over the wavefront's lanes: task t = 3*slot + k computes draw k of walker `slot`
    const int tasks = 3 * nw;
    const int slot_a = lane / 3, k_a = lane - 3 * slot_a;
    const int i_a = a.shard_begin + (wave_active ? min(first + slot_a, last_li) : 0);
    // small ensembles jump with one table entry per draw (one 128-bit multiply-add on the critical path),
    // large ones compose a two-level table (256-walker blocks x position inside the block) with the draw offset
    const bool direct_jump = a.task_jump != nullptr;
    Affine128 j_hi, j_lo;
    if (direct_jump)
        j_hi = a.task_jump[3 * i_a + k_a];
    else
    {
        j_hi = a.jump_hi[i_a >> 8];
        j_lo = a.jump_lo[i_a & 255];
    }

    T own[EPL];
    T lp_old;
    uint32_t nacc_old = 0;
    {
        const int li0 = first + grp;
        const bool act0 = wave_active && li0 < a.shard_count;
        const int w0 = half_base + a.shard_begin + (act0 ? li0 : 0);
        load_slice<T, EPL>(a.pos + (size_t)w0 * a.dims, i0, a.dims, vec_ok, act0, own);
        lp_old = a.logp[w0];
        if (sub == 0) nacc_old = a.n_accept[w0];
    }
    const bool has_block_scratch = Calc::block_scratch_elems(a.dims) != 0;
    Calc::block_init(sh_block, a.calc_params, a.dims, vec_ok, (int)threadIdx.x, 64 * kWavesPerBlock);
    if (has_block_scratch) __syncthreads();
    ctx.block_scratch = has_block_scratch ? sh_block : nullptr;
    // the calculator's per-lane registers (read from the LDS copy: they land while phase A computes)
    typename Calc::template Regs<EPL, LPW> cregs;
    Calc::template preload<EPL, LPW>(ctx, a.calc_params, cregs);
    MCMCPP_STAMP(1);  // every up-front load has landed

    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        // hand the stream and the counters to the next half-step
        StepCtl nx = ctl;
        nx.state = apply(a.half_jump, ctl.state);
        nx.half_step = ctl.half_step + 1;
        if (a.color)
        {
            // the ensemble step ends with the black half: advance the per-step counters
            const bool saved = ctl.save_phase + 1u == (uint32_t)run.interval;
            nx.step_in_run = ctl.step_in_run + 1;
            nx.save_phase = saved ? 0u : ctl.save_phase + 1u;
            nx.chain_slot = ctl.chain_slot + (saved ? 1 : 0);
            nx.partial_slot = (ctl.partial_slot + 1u == (uint32_t)a.partial_slots) ? 0u : ctl.partial_slot + 1u;
        }
        *a.ctl_out = nx;
    }
    if (!wave_active) return;

    // does this ensemble step go to the chain?  (EnsembleSampler.h:298-306: interval-1 unsaved, 1 saved)
    long long save_slot = -1;
    if (a.direct_save_slot >= 0)
        save_slot = a.direct_save_slot;
    else if (a.use_ctl_save && run.chain != nullptr && ctl.save_phase + 1u == (uint32_t)run.interval)
        save_slot = run.chain_slot_base + ctl.chain_slot;

    PhaseA<T>& pa = sh_a[wib];

    // ---------------- phase A: the three random draws of every walker of this wavefront ---------------
    // Draw k of the walker at position i of the half is draw 3*i + k of this half-step (the order in which
    // the reference's loop consumes its engine, EnsembleSampler.h:345-354 / StretchMove.h:102,104,113): the
    // lane jumps the half-step's base state there directly.  k = 0: partner = engine(n); k = 1: z = Gw(u)
    // and (D-1) ln z; k = 2: ln U = -(-log(1-u)/1).  One lane per draw keeps the dependent chain short.
    for (int t = lane; t < tasks; t += 64)
    {
        const int slot = (t == lane) ? slot_a : t / 3;
        const int k = (t == lane) ? k_a : t - 3 * slot;
        if (first + slot < a.shard_count)
        {
            U128 s;
            const int i = a.shard_begin + first + slot;
            if (direct_jump)
                s = apply(t == lane ? j_hi : a.task_jump[3 * i + k], ctl.state);
            else
            {
                if (t == lane)
                    s = apply(j_lo, apply(j_hi, ctl.state));
                else
                    s = apply(a.jump_lo[i & 255], apply(a.jump_hi[i >> 8], ctl.state));
                const Affine128 dj = k == 0 ? a.draw_jump[0] : (k == 1 ? a.draw_jump[1] : a.draw_jump[2]);
                s = apply(dj, s);
            }
            const uint64_t r = pcg_output(s);
            if (k == 0)
            {
                if (r < a.redraw_threshold) atomicAdd(&a.diag->redraws, 1ULL);
                pa.partner[slot] = a.n_is_pow2 ? (uint32_t)(r & (uint64_t)(a.n - 1)) : (uint32_t)(r % (uint64_t)a.n);
            }
            else
            {
                const T u = canonical(r, T());
                const T tmp = a.gw_term1 * u + a.gw_inv_sqrt;
                const T z = tmp * tmp;                 // GwDistribution.h:58
                const T arg = (k == 1) ? z : (T)1 - u;  // one logarithm serves both kinds of lane
                const T lg = dev_log(arg);
                if (k == 1)
                {
                    pa.z[slot] = z;
                    pa.zs[slot] = lg * a.dims_minus_one;  // StretchMove.h:110
                }
                else
                    pa.ln_u[slot] = lg;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    MCMCPP_STAMP(2);  // phase A done

    // ---------------- phase B: LPW lanes per walker ------------------------------------------------
    unsigned accepted_here = 0;
    for (int q = 0; q < a.passes; ++q)
    {
        const int slot = q * WPP + grp;  // walker slot inside the wavefront
        const int li = first + slot;
        const bool active = li < a.shard_count;
        const int w = half_base + a.shard_begin + (active ? li : 0);
        const uint32_t p = active ? pa.partner[slot] : 0u;
        const T z = pa.z[slot];

        T* row = a.pos + (size_t)w * a.dims;
        const T* prow = a.pos + (size_t)(other_base + (int)p) * a.dims;
        T par[EPL], prop[EPL];
        load_slice<T, EPL>(prow, i0, a.dims, vec_ok, active, par);

        // the next pass's own rows and counters travel while this pass computes
        T own_next[EPL];
        T lp_next = (T)0;
        uint32_t nacc_next = 0;
        if (q + 1 < a.passes)
        {
            const int lin = li + WPP;
            const bool actn = lin < a.shard_count;
            const int wn = half_base + a.shard_begin + (actn ? lin : 0);
            load_slice<T, EPL>(a.pos + (size_t)wn * a.dims, i0, a.dims, vec_ok, actn, own_next);
            lp_next = a.logp[wn];
            if (sub == 0) nacc_next = a.n_accept[wn];
        }

        if (q == 0) MCMCPP_STAMP(3);  // partner rows landed
        // StretchMove.h:105-108  proposal = sel + z*(cur - sel); padded cells stay +0
#pragma unroll
        for (int e = 0; e < EPL; ++e)
        {
            const T d = own[e] - par[e];
            const T zd = z * d;
            prop[e] = par[e] + zd;
        }
        const T lp_new = Calc::template eval<EPL, LPW>(ctx, a.calc_params, cregs, prop);
        if (q == 0) MCMCPP_STAMP(4);  // calculator done

        // StretchMove.h:112-113  accept iff lnU < (probScaling + newProb) - oldProb
        const T zs = pa.zs[slot], ln_u = pa.ln_u[slot];
        const T delta = zs + lp_new - lp_old;
        const bool accept = active && (ln_u < delta);
        if (active && sub == 0)
        {
            const T margin = dev_abs(ln_u - delta);
            const T scale = dev_abs(ln_u) + dev_abs(zs) + dev_abs(lp_new) + dev_abs(lp_old);
            if (margin <= a.tie_eps * scale) atomicAdd(&a.diag->near_ties, 1ULL);
        }
        if (accept)
        {
            // Walker::jumpToNewPointSwap (Walker/Walker.h:172-179)
            store_slice<T, EPL>(row, i0, a.dims, vec_ok, prop);
            if (sub == 0)
            {
                a.logp[w] = lp_new;
                a.n_accept[w] = nacc_old + 1u;
            }
        }
        if (save_slot >= 0 && active)
        {
            // Walker -> Chain::storeWalker (Chain/ChainBlock.h:125-131): cell = slot*W*D + walker*D + p
            T* crow = reinterpret_cast<T*>(run.chain) + ((size_t)save_slot * (size_t)(2 * a.n) + (size_t)w) * a.dims;
            if (accept)
                store_slice<T, EPL>(crow, i0, a.dims, vec_ok, prop);
            else
                store_slice<T, EPL>(crow, i0, a.dims, vec_ok, own);
        }
        accepted_here += (unsigned)__popcll(__ballot(accept && sub == 0));
#pragma unroll
        for (int e = 0; e < EPL; ++e) own[e] = own_next[e];
        lp_old = lp_next;
        nacc_old = nacc_next;
    }
    MCMCPP_STAMP(5);  // all stores of this wavefront acknowledged
#ifdef MCMCPP_STAMPS
    if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
        for (int k = 0; k < 8; ++k) a.stamps[k] = stamp_val[k];
#endif
    // per-wavefront accepted count of this half-step; summed per ensemble step by accepted_reduce_kernel
    // (one plain store per wavefront: thousands of same-address atomics would serialise for ~12 ns each)
    if (a.partials != nullptr && run.accepted_per_step != nullptr && lane == 0)
        a.partials[((size_t)ctl.partial_slot * 2 + (size_t)a.color) * (size_t)a.partial_waves + (size_t)wave] = accepted_here;
}

// Sums the per-wavefront accepted counts of the last `count` ensemble steps into RunInfo.accepted_per_step.
// Runs once after every graph replay (one workgroup per step); `ctl_after` is the control record the
// last black launch left behind, so step_in_run is the number of steps finished in this run.
#ifdef MCMCPP_DEFINE_REDUCE_KERNEL  // one definition, in mcmcpp_hip.hip
__global__ void __launch_bounds__(256)
accepted_reduce_kernel(const uint32_t* partials, int partial_slots, int partial_waves, int count, const StepCtl* ctl_after,
                       const RunInfo* run_ptr)
{
    __shared__ unsigned sums[4];
    const RunInfo run = *run_ptr;
    if (run.accepted_per_step == nullptr) return;
    const uint64_t done = ctl_after->step_in_run;
    const uint64_t step = done - (uint64_t)count + blockIdx.x;
    const uint32_t* src = partials + (size_t)(step % (uint64_t)partial_slots) * 2 * (size_t)partial_waves;
    unsigned s = 0;
    for (int k = threadIdx.x; k < 2 * partial_waves; k += blockDim.x) s += src[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sums[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) run.accepted_per_step[step] = sums[0] + sums[1] + sums[2] + sums[3];
}
#endif

// Calculator evaluated on arbitrary rows (mcmcpp_hip_calc_logp): same functor, same lane mapping.
template <class T, class Calc, int EPL, int LPW>
__global__ void __launch_bounds__(64 * kWavesPerBlock)
calc_logp_kernel(const T* pos, T* out, const T* calc_params, long long count, int dims, int vec_ok)
{
    constexpr int WPP = 64 / LPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* sh_stage = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::stage_offset());
    T* sh_block = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::block_offset());
    const bool has_block_scratch = Calc::block_scratch_elems(dims) != 0;
    Calc::block_init(sh_block, calc_params, dims, vec_ok != 0, (int)threadIdx.x, 64 * kWavesPerBlock);
    if (has_block_scratch) __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const long long wave = (long long)blockIdx.x * kWavesPerBlock + wib;
    const int sub = lane & (LPW - 1);
    const long long w = wave * WPP + lane / LPW;
    const bool active = w < count;
    GroupCtx<T, EPL, LPW> ctx;
    ctx.sub = sub;
    ctx.dims = dims;
    ctx.lane = lane;
    ctx.stage = Calc::kNeedsStage ? &sh_stage[wib * 64 * EPL] : nullptr;
    ctx.block_scratch = has_block_scratch ? sh_block : nullptr;
    ctx.vec_ok = vec_ok != 0;
    T x[EPL];
    load_slice<T, EPL>(pos + (size_t)(active ? w : 0) * dims, sub * EPL, dims, vec_ok != 0, active, x);
    typename Calc::template Regs<EPL, LPW> cregs;
    Calc::template preload<EPL, LPW>(ctx, calc_params, cregs);
    const T lp = Calc::template eval<EPL, LPW>(ctx, calc_params, cregs, x);
    if (active && sub == 0) out[w] = lp;
}

}  // namespace mcmcpp

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
#include "pcg128.hpp"

namespace mcmcpp
{

// Device-resident record carrying the random stream and step counters from one half-step launch to the
// next (two records, ping-pong: the red launch reads [0] and writes [1], the black launch the reverse),
// so that graph replays need no host-side argument updates.
struct StepCtl
{
    U128 state;           // engine state before the first draw of this half-step
    uint64_t half_step;   // half-steps executed since set_state
    uint64_t step_in_run; // ensemble step index inside the current run() call
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
    const T* calc_params;
    Affine128 half_jump;      // map of 3*n draws: this half-step's base state -> the next one's
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
    long long direct_save_slot; // >= 0: store into run->chain at this slot regardless of interval (sharded driver)
    int use_ctl_save;           // 1: saving follows RunInfo.interval / StepCtl.step_in_run
};

template <class T>
struct Vec16;
template <>
struct Vec16<double>
{
    typedef double2 type;
    static constexpr int N = 2;
};
template <>
struct Vec16<float>
{
    typedef float4 type;
    static constexpr int N = 4;
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

__device__ __forceinline__ double dev_log(double x) { return log(x); }
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
    T lp_old[64];
    uint32_t partner[64];
};

constexpr int kWavesPerBlock = 4;

template <class T, class Calc, int EPL, int LPW>
__global__ void __launch_bounds__(64 * kWavesPerBlock) stretch_half_step_kernel(const HalfStepArgs<T> a)
{
    static_assert((LPW & (LPW - 1)) == 0 && LPW >= 1 && LPW <= 64, "LPW must be a power of two <= 64");
    static_assert(EPL % Vec16<T>::N == 0, "EPL must be a whole number of 16-byte vectors");
    constexpr int WPP = 64 / LPW;  // walkers per pass

    __shared__ PhaseA<T> sh_a[kWavesPerBlock];
    __shared__ T sh_stage[Calc::kNeedsStage ? kWavesPerBlock * 64 * EPL : 1];

    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int wave = blockIdx.x * kWavesPerBlock + wib;
    const int nw = WPP * a.passes;
    const int first = wave * nw;  // first walker of this wavefront, relative to the shard

    // control record of this half-step (wave-uniform)
    const StepCtl ctl = *a.ctl_in;
    const RunInfo run = *a.run;

    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        // hand the stream and the counters to the next half-step
        StepCtl nx;
        nx.state = apply(a.half_jump, ctl.state);
        nx.half_step = ctl.half_step + 1;
        nx.step_in_run = ctl.step_in_run + (a.color ? 1 : 0);
        *a.ctl_out = nx;
    }
    if (first >= a.shard_count) return;

    // does this ensemble step go to the chain?  (EnsembleSampler.h:298-306: interval-1 unsaved, 1 saved)
    long long save_slot = -1;
    if (a.direct_save_slot >= 0)
        save_slot = a.direct_save_slot;
    else if (a.use_ctl_save && run.chain != nullptr)
    {
        const long long s1 = (long long)ctl.step_in_run + 1;
        if (s1 % run.interval == 0) save_slot = run.chain_slot_base + s1 / run.interval - 1;
    }

    PhaseA<T>& pa = sh_a[wib];
    const int half_base = a.color ? a.n : 0;
    const int other_base = a.color ? 0 : a.n;

    // ---------------- phase A: one walker per lane -------------------------------------------------
    {
        const int li = first + lane;
        if (lane < nw && li < a.shard_count)
        {
            const int i = a.shard_begin + li;  // index inside the half == position in the reference's loop
            // engine state before this walker's first draw: base state advanced by 3*i draws
            U128 s = apply(a.jump_hi[i >> 8], ctl.state);
            s = apply(a.jump_lo[i & 255], s);
            s = pcg_step(s, a.inc);
            const uint64_t r0 = pcg_output(s);  // StretchMove.h:102  partner = engine(n)
            s = pcg_step(s, a.inc);
            const uint64_t r1 = pcg_output(s);  // StretchMove.h:104  z = Gw(uniform)
            s = pcg_step(s, a.inc);
            const uint64_t r2 = pcg_output(s);  // StretchMove.h:113  -Exp(1)
            if (r0 < a.redraw_threshold) atomicAdd(&a.diag->redraws, 1ULL);
            const uint32_t p = a.n_is_pow2 ? (uint32_t)(r0 & (uint64_t)(a.n - 1)) : (uint32_t)(r0 % (uint64_t)a.n);
            const T u1 = canonical(r1, T());
            const T tmp = a.gw_term1 * u1 + a.gw_inv_sqrt;
            const T z = tmp * tmp;
            const T u2 = canonical(r2, T());
            pa.partner[lane] = p;
            pa.z[lane] = z;
            pa.zs[lane] = dev_log(z) * a.dims_minus_one;
            pa.ln_u[lane] = dev_log((T)1 - u2);  // -(-log(1-u)/1)
            pa.lp_old[lane] = a.logp[half_base + i];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---------------- phase B: LPW lanes per walker ------------------------------------------------
    const int sub = lane & (LPW - 1);
    const int grp = lane / LPW;
    const int i0 = sub * EPL;
    const bool vec_ok = a.vec_ok != 0;
    GroupCtx<T, EPL, LPW> ctx;
    ctx.sub = sub;
    ctx.dims = a.dims;
    ctx.lane = lane;
    ctx.stage = Calc::kNeedsStage ? &sh_stage[wib * 64 * EPL] : nullptr;

    unsigned accepted_here = 0;
    for (int q = 0; q < a.passes; ++q)
    {
        const int slot = q * WPP + grp;  // walker slot inside the wavefront
        const int li = first + slot;
        const bool active = li < a.shard_count;
        const int i = a.shard_begin + (active ? li : 0);
        const uint32_t p = active ? pa.partner[slot] : 0u;
        const T z = pa.z[slot];

        T* row = a.pos + (size_t)(half_base + i) * a.dims;
        const T* prow = a.pos + (size_t)(other_base + (int)p) * a.dims;
        T own[EPL], par[EPL], prop[EPL];
        load_slice<T, EPL>(row, i0, a.dims, vec_ok, active, own);
        load_slice<T, EPL>(prow, i0, a.dims, vec_ok, active, par);
        // StretchMove.h:105-108  proposal = sel + z*(cur - sel); padded cells stay +0
#pragma unroll
        for (int e = 0; e < EPL; ++e)
        {
            const T d = own[e] - par[e];
            const T zd = z * d;
            prop[e] = par[e] + zd;
        }
        const T lp_new = Calc::template eval<EPL, LPW>(ctx, a.calc_params, prop);

        // StretchMove.h:112-113  accept iff lnU < (probScaling + newProb) - oldProb
        const T zs = pa.zs[slot], ln_u = pa.ln_u[slot], lp_old = pa.lp_old[slot];
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
                a.logp[half_base + i] = lp_new;
                a.n_accept[half_base + i] += 1u;
            }
        }
        if (save_slot >= 0 && active)
        {
            // Walker -> Chain::storeWalker (Chain/ChainBlock.h:125-131): cell = slot*W*D + walker*D + p
            T* crow = reinterpret_cast<T*>(run.chain) + ((size_t)save_slot * (size_t)(2 * a.n) + (size_t)(half_base + i)) * a.dims;
            if (accept)
                store_slice<T, EPL>(crow, i0, a.dims, vec_ok, prop);
            else
                store_slice<T, EPL>(crow, i0, a.dims, vec_ok, own);
        }
        accepted_here += (unsigned)__popcll(__ballot(accept && sub == 0));
    }
    if (run.accepted_per_step != nullptr && lane == 0 && accepted_here != 0)
        atomicAdd(run.accepted_per_step + ctl.step_in_run, accepted_here);
}

// Calculator evaluated on arbitrary rows (mcmcpp_hip_calc_logp): same functor, same lane mapping.
template <class T, class Calc, int EPL, int LPW>
__global__ void __launch_bounds__(64 * kWavesPerBlock)
calc_logp_kernel(const T* pos, T* out, const T* calc_params, long long count, int dims, int vec_ok)
{
    constexpr int WPP = 64 / LPW;
    __shared__ T sh_stage[Calc::kNeedsStage ? kWavesPerBlock * 64 * EPL : 1];
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
    T x[EPL];
    load_slice<T, EPL>(pos + (size_t)(active ? w : 0) * dims, sub * EPL, dims, vec_ok != 0, active, x);
    const T lp = Calc::template eval<EPL, LPW>(ctx, calc_params, x);
    if (active && sub == 0) out[w] = lp;
}

}  // namespace mcmcpp

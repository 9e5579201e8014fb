// diffevo_kernel.hpp -- Mover::DifferentialEvolution on gfx950.
//
// Reference: MCMCpp/Movers/DifferentialEvolution.h:80-112.  One update draws, from the sampler's single pcg64 stream,
//   ind1 = engine(n); ind2 = engine(n) until it differs from ind1; D uniform jitters; (Calculator); one exponential
// i.e. D + 3 draws plus every draw thrown away (by pcg's bounded_rand below its threshold, by the ind2 loop).  The
// next walker's first draw follows this walker's last, so a walker's place in the stream depends on how many draws
// all walkers before it threw away -- the reason the reference's loop is sequential.  Whether an update starting
// at a given stream position throws draws away depends on the stream alone, not on the walkers, so the places are
// found ahead of the update, in parallel, in two launches per half-step:
//   1. de_plan_kernel (diffevo.hip): a wavefront per walker k looks at the draws at (D+3)k + r for every shift
//      r = 0..kDeMaxShift the walkers before it may have caused (lane j makes draw j), and reports the walker as a
//      candidate when any of those starts would throw a draw away (about kDeMaxShift + 1 of the n walkers), with its
//      table extra[r];
//   2. de_update_kernel (here): every workgroup sorts the few candidates and walks them in order, r += extra[r] -- the
//      only sequential part, a few dozen steps from LDS (workgroup 0 also hands the stream on to the next half-step);
//      then every walker finds its shift by a search in that short list, jumps to its place through three table
//      look-ups, replays its integer draws, and its lanes draw their own jitters.
// More than kDeMaxShift thrown-away draws in one half-step (expected: about one, whatever n is) raise a sticky error
// flag the host turns into a failed run: never a silently different chain.
//
// Everything else is the stretch kernels' machinery: the calculator functor with its lane mapping (LPW lanes x EPL
// elements), rows updated in place (a half only reads the other half), optional store into the device chain.
#pragma once

#include "stretch_kernel.hpp"

namespace mcmcpp
{
constexpr int kDeMaxShift = 31;   // largest number of thrown-away draws inside one half-step that is followed exactly
constexpr int kDeWindow = 32;     // raw draws one update's integer part may consume (2 + up to 30 thrown away: even two
                                  // walkers per half, where every second ind2 collides, overrun once in 1e9 updates)
constexpr int kDeOverrun = 255;   // DeCand::extra value of a start whose update would not fit that window
constexpr int kDeRaw = kDeMaxShift + 1 + kDeWindow;
constexpr int kDeMaxCand = 128;   // candidates per half-step the lists hold (typical: kDeMaxShift + 1)
constexpr int kDeAccSlots = 64;   // counters an ensemble step's accepted proposals are spread over (same-address atomics serialise)

enum : uint32_t
{
    kDeErrShift = 1u,   // more than kDeMaxShift draws thrown away in one half-step
    kDeErrCand = 2u,    // more than kDeMaxCand candidates
    kDeErrWindow = 4u,  // one update threw away more than kDeWindow - 2 draws
};

// one per half-step parity
struct alignas(64) DeCtl
{
    U128 state;                  // engine state in front of this half-step's first draw
    unsigned long long extra_total;  // draws thrown away before this half-step
    uint32_t cand_count;         // filled by this half-step's plan kernel; cleared when the previous update kernel hands over
    uint32_t error;              // kDeErr* bits, sticky
    uint32_t pad[8];
};

struct DeCand
{
    uint32_t k;
    uint8_t extra[kDeMaxShift + 1];  // draws thrown away by an update of walker k that starts r draws late
};
static_assert(sizeof(DeCand) == 4 * (1 + (kDeMaxShift + 1) / 4), "the update kernel stages DeCand word by word");

// resolved list entry (LDS), sorted by k: walkers behind k start shift_after draws late
struct DePlan
{
    uint32_t k;
    uint32_t shift_after;
};

template <class T>
struct DeArgs
{
    T* pos;                 // [W][D]
    T* logp;                // [W]
    uint32_t* n_accept;     // [W]
    const T* calc_params;
    const DeCtl* ctl;       // this half-step's
    DeCtl* ctl_next;        // the next half-step's: written by workgroup 0
    const DeCand* cand;     // [kDeMaxCand] candidates of this half-step, unordered
    Affine128 half_jump;    // (D+3)*n draws
    const Affine128* jump_hi;     // [ceil(n/256)]  (D+3)*256*m draws
    const Affine128* jump_lo;     // [256]          (D+3)*j draws
    const Affine128* jump_small;  // [D + kDeRaw + 1]  j draws
    Diag* diag;
    T* chain;               // device chain, or nullptr
    uint32_t* accepted;     // accepted proposals of this ensemble step ([kDeAccSlots] counters, summed by the host), or nullptr
    long long save_slot;    // >= 0: store the rows into chain[save_slot]
    uint64_t threshold;     // (2^64 - n) mod n
    U128 inc;               // pcg stream increment
    T gamma, jitter_low, jitter_width, tie_eps;
    int n, dims, color, vec_ok;
};

__device__ __forceinline__ uint32_t de_bounded(uint64_t v, int n, bool pow2) { return pow2 ? (uint32_t)(v & (uint64_t)(n - 1)) : (uint32_t)(v % (uint64_t)n); }

template <class T, class Calc, int EPL, int LPW>
__global__ void __launch_bounds__(64 * kWavesPerBlock) de_update_kernel(const DeArgs<T> a)
{
    constexpr int WPP = 64 / LPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* sh_stage = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::stage_offset());
    T* sh_block = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::block_offset());
    __shared__ DePlan sh_plan[kDeMaxCand];
    __shared__ __attribute__((aligned(16))) DeCand sh_cand[kDeMaxCand];  // as the planning left them
    __shared__ DeCand sh_sorted[kDeMaxCand];                             // in walker order: the walk never leaves LDS
    __shared__ Affine128 sh_small[kDeMaxShift + 1];  // the jump of `shift` draws: read right behind the search, from LDS
    constexpr int kThreads = 64 * kWavesPerBlock;
    const int dims = a.dims, n = a.n;
    const bool vec_ok = a.vec_ok != 0;
    const bool has_block_scratch = Calc::block_scratch_elems(dims) != 0;
    typename Calc::Prefetch calc_pf;
    Calc::block_prefetch(calc_pf, a.calc_params, dims, vec_ok, (int)threadIdx.x, kThreads);
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int sub = lane & (LPW - 1);
    const int k = (blockIdx.x * kWavesPerBlock + wib) * WPP + lane / LPW;  // walker inside the half
    const bool active = k < n;
    const int kk = active ? k : 0;
    const int half_base = a.color ? n : 0, other_base = a.color ? 0 : n;
    const int w = half_base + kk;
    const int i0 = sub * EPL;

    // First round trip, everything that needs no other load's result: the control record, the whole candidate list (its
    // length is in the control record: entries beyond it are stale and never looked at), and what the walker index
    // alone addresses -- own row, log-posterior, counter, the table jumps.  They travel while the workgroup sorts and
    // walks the candidates below.
    // own row, log-posterior, counter: addressed by the walker index alone
    T own[EPL];
    load_slice<T, EPL>(a.pos + (size_t)w * dims, i0, dims, vec_ok, active, own);
    const T lp_old = a.logp[w];
    const uint32_t nacc_old = a.n_accept[w];
    const Affine128 j_hi = a.jump_hi[kk >> 8], j_lo = a.jump_lo[kk & 255];
    const Affine128 j_uni = a.jump_small[i0 < dims ? i0 : dims], j_exp = a.jump_small[dims];
    const DeCtl ctl = *a.ctl;
    // the first kDeFirstCand candidates (all there are, 99.7 % of the time) in one 16-byte load per thread, issued without
    // waiting for the count in the control record; entries beyond the count are stale and never looked at
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    constexpr int kDeFirstCand = 48;
    constexpr int kFirstPieces = kDeFirstCand * (int)sizeof(DeCand) / 16;
    static_assert(kDeFirstCand * sizeof(DeCand) % 16 == 0 && kFirstPieces <= kThreads, "one 16-byte piece per thread");
    if ((int)threadIdx.x < kFirstPieces) reinterpret_cast<v4u*>(sh_cand)[threadIdx.x] = reinterpret_cast<const v4u*>(a.cand)[threadIdx.x];
    const int plan_count = (int)(ctl.cand_count < (uint32_t)kDeMaxCand ? ctl.cand_count : (uint32_t)kDeMaxCand);
    if (threadIdx.x <= kDeMaxShift) sh_small[threadIdx.x] = a.jump_small[threadIdx.x];
    for (int t = kDeFirstCand * 9 + (int)threadIdx.x; t < plan_count * 9; t += kThreads)  // (a DeCand is nine words)
        reinterpret_cast<uint32_t*>(sh_cand)[t] = reinterpret_cast<const uint32_t*>(a.cand)[t];
    Calc::block_commit(calc_pf, sh_block, a.calc_params, dims, vec_ok, (int)threadIdx.x, kThreads);
    __syncthreads();
    for (int j = threadIdx.x; j < plan_count; j += kThreads)
    {
        int rank = 0;
        const uint32_t mine = sh_cand[j].k;
        for (int i = 0; i < plan_count; ++i) rank += sh_cand[i].k < mine ? 1 : 0;  // (walker indices are distinct)
        sh_sorted[rank] = sh_cand[j];
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        // the sequential part: the candidates in walker order, each starting as late as those before it made it
        int r = 0;
        uint32_t err = ctl.error;
        for (int j = 0; j < plan_count; ++j)
        {
            int own = (int)sh_sorted[j].extra[r];
            if (own == kDeOverrun)
            {
                err |= kDeErrWindow;
                own = 0;
            }
            r += own;
            if (r > kDeMaxShift)
            {
                err |= kDeErrShift;
                r = kDeMaxShift;
            }
            sh_plan[j].k = sh_sorted[j].k;
            sh_plan[j].shift_after = (uint32_t)r;
        }
        if (blockIdx.x == 0)
        {
            // hand the stream on (the next half-step's record is not in use: its last reader was the previous update)
            DeCtl nx;
            nx.state = apply(sh_small[r], apply(a.half_jump, ctl.state));
            nx.extra_total = ctl.extra_total + (unsigned long long)r;
            nx.cand_count = 0;
            nx.error = err;
            for (int q = 0; q < 8; ++q) nx.pad[q] = 0;
            *a.ctl_next = nx;
        }
    }
    __syncthreads();

    GroupCtx<T, EPL, LPW> ctx;
    ctx.sub = sub;
    ctx.dims = dims;
    ctx.lane = lane;
    ctx.stage = Calc::kNeedsStage ? &sh_stage[wib * 64 * EPL] : nullptr;
    ctx.block_scratch = has_block_scratch ? sh_block : nullptr;
    ctx.vec_ok = vec_ok;

    // this walker's place in the stream: the last list entry in front of it says how late it starts
    int lo = 0, hi = plan_count;  // first entry with k' >= k
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if ((int)sh_plan[mid].k < kk)
            lo = mid + 1;
        else
            hi = mid;
    }
    const int shift = lo > 0 ? (int)sh_plan[lo - 1].shift_after : 0;
    U128 s = apply(sh_small[shift], apply(j_lo, apply(j_hi, ctl.state)));

    // ind1, ind2 (DifferentialEvolution.h:83-87), thrown-away draws included; the plan bounds the loops
    const bool pow2 = (n & (n - 1)) == 0;
    uint64_t v;
    int budget = kDeWindow;
    do
    {
        s = pcg_step(s, a.inc);
        v = pcg_output(s);
    } while (v < a.threshold && --budget > 0);
    const uint32_t ind1 = de_bounded(v, n, pow2);
    uint32_t ind2;
    do
    {
        do
        {
            s = pcg_step(s, a.inc);
            v = pcg_output(s);
        } while (v < a.threshold && --budget > 0);
        ind2 = de_bounded(v, n, pow2);
    } while (ind2 == ind1 && --budget > 0);

    T w1[EPL], w2[EPL];
    load_slice<T, EPL>(a.pos + (size_t)(other_base + (int)ind1) * dims, i0, dims, vec_ok, active, w1);
    load_slice<T, EPL>(a.pos + (size_t)(other_base + (int)ind2) * dims, i0, dims, vec_ok, active, w2);

    typename Calc::template Regs<EPL, LPW> cregs;
    Calc::template preload<EPL, LPW>(ctx, a.calc_params, cregs);

    // the jitters of this lane's elements (draws i0 .. i0+EPL-1 behind the integer draws) and the exponential
    // (draw D behind them): MultiSampler.h:66,80
    U128 su = apply(j_uni, s);
    U128 se = pcg_step(apply(j_exp, s), a.inc);
    T prop[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e)
    {
        su = pcg_step(su, a.inc);
        const T u = canonical(pcg_output(su), T());
        const T jitter = a.jitter_low + (u * a.jitter_width);
        const T d = w1[e] - w2[e];
        const T gd = a.gamma * d;
        const T moved = own[e] + gd;
        const T p = moved + jitter;
        prop[e] = (active && i0 + e < dims) ? p : (T)0;  // padded cells stay +0
    }
    const T neg_exp = dev_log((T)1 - canonical(pcg_output(se), T()));  // -(-log(1 - u)/1)

    const T lp_new = Calc::template eval<EPL, LPW>(ctx, a.calc_params, cregs, prop);
    const T delta = lp_new - lp_old;
    const bool accept = active && (delta > neg_exp);  // DifferentialEvolution.h:100
    if (active && sub == 0)
    {
        const T margin = dev_abs(neg_exp - delta);
        const T scale = dev_abs(neg_exp) + dev_abs(lp_new) + dev_abs(lp_old);
        if (margin <= a.tie_eps * scale) atomicAdd(&a.diag->near_ties, 1ULL);
    }
    if (accept)
    {
        store_slice<T, EPL>(a.pos + (size_t)w * dims, i0, dims, vec_ok, prop);
        if (sub == 0)
        {
            a.logp[w] = lp_new;
            a.n_accept[w] = nacc_old + 1u;
        }
    }
    if (a.save_slot >= 0 && active)
    {
        T* crow = a.chain + ((size_t)a.save_slot * (size_t)(2 * n) + (size_t)w) * dims;
        if (accept)
            store_slice<T, EPL>(crow, i0, dims, vec_ok, prop);
        else
            store_slice<T, EPL>(crow, i0, dims, vec_ok, own);
    }
    const unsigned acc = (unsigned)__popcll(__ballot(accept && sub == 0));
    if (a.accepted != nullptr && lane == 0 && acc != 0) atomicAdd(a.accepted + ((blockIdx.x * kWavesPerBlock + wib) & (kDeAccSlots - 1)), acc);
}

}  // namespace mcmcpp

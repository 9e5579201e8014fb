// diffevo_kernel.hpp -- Mover::DifferentialEvolution on gfx950.
//
// Reference: MCMCpp/Movers/DifferentialEvolution.h:80-112.  One update draws, from the sampler's single pcg64 stream,
//   ind1 = engine(n); ind2 = engine(n) until it differs from ind1; D uniform jitters; (Calculator); one exponential
// i.e. D + 3 draws plus every draw thrown away (by pcg's bounded_rand below its threshold, by the ind2 loop).  The
// next walker's first draw follows this walker's last, so a walker's place in the stream depends on how many draws
// all walkers before it threw away -- the reason the reference's loop is sequential.  Whether an update starting
// at a given stream position throws draws away depends on the stream alone, never on the walkers, so everything that
// concerns the stream is PLANNED AHEAD of the updates, by extra workgroups of the very launch that updates an earlier
// half-step.  One launch per half-step h (de_step_kernel), three kinds of workgroup:
//   update   half-step h: like the stretch half-step kernel -- first round trip: the walker's 32-byte record (its two
//            partners, the engine state behind its integer draws, the logarithm of its accept draw), own row,
//            log-posterior, counter; second round trip: the two partner rows; in its shadow the lanes draw their own
//            jitters; calculator, accept in place, optional chain store;
//   records  half-step h + 1: every planning workgroup first sorts the few candidates of that half-step (found one
//            launch earlier) and walks them in order, r += extra[r] -- the only sequential part, a few dozen steps from
//            LDS; then one lane per walker finds its shift by a search in that short list, jumps to its place through
//            three table look-ups, replays its integer draws and leaves the walker's record behind;
//   find     half-step h + 2 (its base state follows from the walk above): a wavefront per walker k looks at the draws
//            at (D+3)k + r for every shift r = 0..kDeMaxShift the walkers before it may have caused (lane j makes draw
//            j), and reports the walker as a candidate when any of those starts would throw a draw away (about
//            kDeMaxShift + 1 of the n walkers), with its table extra[r].
// The launches of a run are replayed from a hipGraph; the stream position, the error flags and the per-run counters
// travel in device memory (DeCtl ring, DeStepCtl).  More than kDeMaxShift thrown-away draws in one half-step (expected:
// about one, whatever n is) raise a sticky error flag the host turns into a failed run: never a silently different chain.
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

// one per half-step, a ring of four (index = half-step & 3): launch h reads the records of h + 1 and h + 2 and writes that of h + 3
struct alignas(64) DeCtl
{
    U128 state;                      // engine state in front of this half-step's first draw
    unsigned long long extra_total;  // draws thrown away before this half-step
    uint32_t pad[10];
};
static_assert(sizeof(DeCtl) == 64, "one line per record");

// candidate counters of the half-steps (index = half-step & 3), the sticky error flags, all in one line
struct alignas(64) DeShared
{
    uint32_t cand_count[4];
    uint32_t error;       // kDeErr* bits
    uint32_t finished[4]; // finder workgroups done with the candidates of half-step (index & 3): the last one resolves them
    uint32_t pad[7];
};

// resolved list entry, sorted by k: walkers behind k start shift_after draws late
struct DePlan
{
    uint32_t k;
    uint32_t shift_after;
};
// The candidates of one half-step, resolved: sorted, walked (r += extra[r]).  Made by the finder workgroup that finishes
// last, read by the planners of the next launch (two buffers, index = half-step & 1).
struct alignas(64) DeResolved
{
    uint32_t count;
    uint32_t total;  // draws thrown away in the whole half-step
    uint32_t pad[14];
    DePlan plan[kDeMaxCand];
};

struct DeCand
{
    uint32_t k;
    uint8_t extra[kDeMaxShift + 1];  // draws thrown away by an update of walker k that starts r draws late
};
static_assert(sizeof(DeCand) == 4 * (1 + (kDeMaxShift + 1) / 4), "the planner stages DeCand word by word");

// The stream part of one update, made one half-step ahead (DifferentialEvolution.h:83-87,100)
template <class T>
struct alignas(16) DeRec
{
    U128 s;           // engine state behind the integer draws: the D jitter draws and the accept draw follow
    T neg_exp;        // -(-log(1 - u)/1) of the accept draw
    uint32_t ind1, ind2;
};
static_assert(sizeof(DeRec<double>) == 32 && sizeof(DeRec<float>) == 32, "32-byte records");

// per-run constants (uploaded by the host before the launches of a piece of the run)
struct alignas(64) DeRunInfo
{
    void* chain;         // device chain [slots][W][D], or nullptr
    uint32_t* accepted;  // [steps of this piece][kDeAccSlots] counters, or nullptr
    long long interval;
    long long pad[5];
};
// per-step counters handed from ensemble step to ensemble step (two records, index = step & 1)
struct alignas(64) DeStepCtl
{
    long long step_in_piece;
    long long chain_slot;
    uint32_t save_phase;
    uint32_t pad[11];
};

template <class T>
struct DeArgs
{
    T* pos;                 // [W][D]
    T* logp;                // [W]
    uint32_t* n_accept;     // [W]
    const T* calc_params;
    DeCtl* ctl;             // [4] ring
    DeShared* shared;
    DeCand* cand;           // [2][kDeMaxCand]: candidates of half-step h live in buffer h & 1
    DeResolved* resolved;   // [2]: the same resolved
    DeRec<T>* recs;         // [2][n]: records of colour c in buffer c
    const DeRunInfo* run;
    DeStepCtl* step_ctl;    // [2]
    Affine128 half_jump;    // (D+3)*n draws
    const Affine128* jump_hi;     // [ceil(n/256)]  (D+3)*256*m draws
    const Affine128* jump_lo;     // [256]          (D+3)*j draws
    const Affine128* jump_small;  // [D + kDeRaw + 1]  j draws
    Diag* diag;
    uint64_t threshold;     // (2^64 - n) mod n
    U128 inc;               // pcg stream increment
    T gamma, jitter_low, jitter_width, tie_eps;
    int n, dims, vec_ok;
    int half_step_mod4;     // half-step h & 3 of the update this launch performs (colour = h & 1, ensemble step parity = (h >> 1) & 1)
    int update_blocks;      // workgroups [0, update_blocks) update half-step h (0: a planning-only launch)
    int record_blocks;      // the next record_blocks workgroups make the records of half-step h + 1 (0: none, priming); the rest
                            // find the candidates of half-step h + 2
};

__device__ __forceinline__ uint32_t de_bounded(uint64_t v, int n, bool pow2) { return pow2 ? (uint32_t)(v & (uint64_t)(n - 1)) : (uint32_t)(v % (uint64_t)n); }

// ---- find: candidates of the half-step whose base state is `state`, appended to cand[0..] through *count ----------------
// Would an update of walker k that starts r draws late (r = 0..kDeMaxShift) throw draws away?  A wavefront takes kDeFindBatch
// walkers at a time: first one lane per walker jumps to the walker's base state (two table look-ups), then walker after
// walker lane j makes raw draw j behind it (one more jump), neighbouring lanes compare, and for the rare walker where the
// answer is yes for some r, lane r walks the update that starts at draw r and the table extra[r] goes to the candidate list.
constexpr int kDeFindBatch = 16;
template <class T>
__device__ __forceinline__ void de_find(const DeArgs<T>& a, U128 state, DeCand* cand, uint32_t* count, uint64_t (*sh_raw)[64], U128 (*sh_base)[kDeFindBatch],
                                        int first_batch, int batches_stride)
{
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int n = a.n;
    const bool pow2 = (n & (n - 1)) == 0;
    const uint64_t threshold = a.threshold;
    const Affine128 j_draw = a.jump_small[(lane < kDeRaw ? lane : kDeRaw - 1) + 1];
    for (int k0 = (first_batch + wib) * kDeFindBatch; k0 < n; k0 += batches_stride * kDeFindBatch)
    {
        if (lane < kDeFindBatch)
        {
            const int k = k0 + lane < n ? k0 + lane : n - 1;
            sh_base[wib][lane] = apply(a.jump_lo[k & 255], apply(a.jump_hi[k >> 8], state));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int batch = n - k0 < kDeFindBatch ? n - k0 : kDeFindBatch;
        for (int g = 0; g < batch; ++g)
        {
            const int k = k0 + g;
            const uint64_t raw = pcg_output(apply(j_draw, sh_base[wib][g]));
            const uint64_t nxt = __shfl_down(raw, 1);
            // clean(j): draws j and j+1 are both kept and name different walkers: an update starting at j throws nothing away
            const bool bad = lane <= kDeMaxShift && (raw < threshold || nxt < threshold || de_bounded(raw, n, pow2) == de_bounded(nxt, n, pow2));
            if (__ballot(bad) == 0) continue;
            uint32_t slot = 0;
            if (lane == 0) slot = atomicAdd(count, 1u);
            slot = __shfl(slot, 0);
            if (slot >= (uint32_t)kDeMaxCand)
            {
                if (lane == 0) atomicOr(&a.shared->error, kDeErrCand);
                continue;
            }
            // (one wavefront: its LDS accesses execute in order; the fences keep the compiler from moving them)
            sh_raw[wib][lane] = raw;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane <= kDeMaxShift)
            {
                // DifferentialEvolution.h:83-87 from draw `lane` on
                const uint64_t* rw = sh_raw[wib];
                int at = lane;
                const int end = lane + kDeWindow;
                uint64_t v;
                do v = rw[at++];
                while (v < threshold && at < end);
                const uint32_t ind1 = de_bounded(v, n, pow2);
                uint32_t ind2 = ind1;
                bool overrun = v < threshold;
                do
                {
                    if (at >= end)
                    {
                        overrun = true;
                        break;
                    }
                    do v = rw[at++];
                    while (v < threshold && at < end);
                    if (v < threshold) overrun = true;
                    ind2 = de_bounded(v, n, pow2);
                } while (ind2 == ind1);
                // (an overrun is an error only if the walk over the candidates comes through this start)
                cand[slot].extra[lane] = overrun ? (uint8_t)kDeOverrun : (uint8_t)(at - lane - 2);
            }
            if (lane == 0) cand[slot].k = (uint32_t)k;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <class T, class Calc, int EPL, int LPW>
__global__ void __launch_bounds__(64 * kWavesPerBlock) de_step_kernel(const DeArgs<T> a)
{
    constexpr int WPP = 64 / LPW;
    constexpr int kThreads = 64 * kWavesPerBlock;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int dims = a.dims, n = a.n;
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int h4 = a.half_step_mod4;

    if ((int)blockIdx.x >= a.update_blocks)
    {
        // =================================== planning workgroups ==========================================================
        __shared__ DePlan sh_plan[kDeMaxCand];
        __shared__ __attribute__((aligned(16))) DeCand sh_cand[kDeMaxCand];  // as the finders left them
        __shared__ DeCand sh_sorted[kDeMaxCand];                             // in walker order: the walk never leaves LDS
        __shared__ uint64_t sh_raw[kWavesPerBlock][64];
        __shared__ U128 sh_base[kWavesPerBlock][kDeFindBatch];
        __shared__ int sh_last;
        const int pb = (int)blockIdx.x - a.update_blocks;  // planner index
        const int planners = (int)gridDim.x - a.update_blocks;
        const int h1 = (h4 + 1) & 3, h2 = (h4 + 2) & 3, h3 = (h4 + 3) & 3;
        if (pb < a.record_blocks)
        {
            // ---- records of half-step h + 1: its candidates were found and resolved one launch ago ----
            const DeCtl* ctl1 = a.ctl + h1;
            const DeResolved* res = a.resolved + (h1 & 1);
            const U128 state1 = ctl1->state;
            const int plan_count = (int)res->count;
            for (int j = threadIdx.x; j < plan_count; j += kThreads) sh_plan[j] = res->plan[j];
            const int k = pb * kThreads + (int)threadIdx.x;
            const int kc = k < n ? k : n - 1;
            const Affine128 j_hi = a.jump_hi[kc >> 8], j_lo = a.jump_lo[kc & 255], j_exp = a.jump_small[dims];
            __syncthreads();
            if (k >= n) return;
            // this walker's place in the stream: the last list entry in front of it says how late it starts
            int lo = 0, hi = plan_count;  // first entry with k' >= k
            while (lo < hi)
            {
                const int mid = (lo + hi) >> 1;
                if ((int)sh_plan[mid].k < k)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            const int shift = lo > 0 ? (int)sh_plan[lo - 1].shift_after : 0;
            U128 s = apply(a.jump_small[shift], apply(j_lo, apply(j_hi, state1)));
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
            // the exponential (draw D behind the integer draws and the D jitters): MultiSampler.h:80
            const U128 se = pcg_step(apply(j_exp, s), a.inc);
            DeRec<T>* out = a.recs + (size_t)(h1 & 1) * n + k;
            out->s = s;
            out->neg_exp = dev_log((T)1 - canonical(pcg_output(se), T()));  // -(-log(1 - u)/1)
            out->ind1 = ind1;
            out->ind2 = ind2;
            return;
        }
        // ---- candidates of half-step h + 2 (its base state was left in the ring by the launch before this one) ----
        const int fb = pb - a.record_blocks, finders = planners - a.record_blocks;
        const DeCtl ctl2 = a.ctl[h2];
        DeCand* cand2 = a.cand + (size_t)(h2 & 1) * kDeMaxCand;
        de_find<T>(a, ctl2.state, cand2, &a.shared->cand_count[h2], sh_raw, sh_base, fb * kWavesPerBlock, finders * kWavesPerBlock);
        // The finder workgroup that finishes last resolves the list: sorted by walker, then walked in order, each candidate
        // starting as late as those before it made it (r += extra[r], the only sequential part: a few dozen steps from
        // LDS), and hands the stream on.  Hand-off: every storing wavefront drains its stores, workgroup barrier, agent
        // release, ticket; the last arriver acquires before it reads.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
        {
            __atomic_thread_fence(__ATOMIC_RELEASE);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t ticket = __hip_atomic_fetch_add(&a.shared->finished[h2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_last = ticket + 1u == (uint32_t)finders ? 1 : 0;
            if (sh_last)
            {
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        if (!sh_last) return;
        const uint32_t count2 = __hip_atomic_load(&a.shared->cand_count[h2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int plan_count = (int)(count2 < (uint32_t)kDeMaxCand ? count2 : (uint32_t)kDeMaxCand);
        for (int t = (int)threadIdx.x; t < plan_count * 9; t += kThreads)  // (a DeCand is nine words)
            reinterpret_cast<uint32_t*>(sh_cand)[t] = reinterpret_cast<const uint32_t*>(cand2)[t];
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
            int r = 0;
            uint32_t err = 0;
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
            DeResolved* res = a.resolved + (h2 & 1);
            res->count = (uint32_t)plan_count;
            res->total = (uint32_t)r;
            // the record of half-step h + 3 (its last reader was two launches ago)
            DeCtl* nx = a.ctl + h3;
            nx->state = apply(a.jump_small[r], apply(a.half_jump, ctl2.state));
            nx->extra_total = ctl2.extra_total + (unsigned long long)r;
            // this half-step's counters are free for the half-step four later
            a.shared->cand_count[h2] = 0;
            a.shared->finished[h2] = 0;
            if (err) atomicOr(&a.shared->error, err);
        }
        __syncthreads();
        {
            DeResolved* res = a.resolved + (h2 & 1);
            for (int j = threadIdx.x; j < plan_count; j += kThreads) res->plan[j] = sh_plan[j];
        }
        return;
    }

    // ======================================= update workgroups: half-step h ===============================================
    T* sh_stage = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::stage_offset());
    T* sh_block = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::block_offset());
    const bool vec_ok = a.vec_ok != 0;
    const bool has_block_scratch = Calc::block_scratch_elems(dims) != 0;
    typename Calc::Prefetch calc_pf;
    Calc::block_prefetch(calc_pf, a.calc_params, dims, vec_ok, (int)threadIdx.x, kThreads);
    const int color = h4 & 1;
    const int sub = lane & (LPW - 1);
    const int k = (blockIdx.x * kWavesPerBlock + wib) * WPP + lane / LPW;  // walker inside the half
    const bool active = k < n;
    const int kk = active ? k : 0;
    const int half_base = color ? n : 0, other_base = color ? 0 : n;
    const int w = half_base + kk;
    const int i0 = sub * EPL;

    // first round trip: what the walker index alone addresses
    const DeRec<T> rec = a.recs[(size_t)color * n + kk];
    T own[EPL];
    load_slice<T, EPL>(a.pos + (size_t)w * dims, i0, dims, vec_ok, active, own);
    const T lp_old = a.logp[w];
    const uint32_t nacc_old = a.n_accept[w];
    const Affine128 j_uni = a.jump_small[i0 < dims ? i0 : dims];
    // (field by field: a whole-record copy drags the padding through registers and scratch)
    const DeStepCtl* scp = a.step_ctl + ((h4 >> 1) & 1);
    const long long sc_step = scp->step_in_piece, sc_slot = scp->chain_slot;
    const uint32_t sc_phase = scp->save_phase;
    void* const run_chain = a.run->chain;
    uint32_t* const run_accepted = a.run->accepted;
    const long long run_interval = a.run->interval;

    // second round trip: the two partner rows
    T w1[EPL], w2[EPL];
    load_slice<T, EPL>(a.pos + (size_t)(other_base + (int)rec.ind1) * dims, i0, dims, vec_ok, active, w1);
    load_slice<T, EPL>(a.pos + (size_t)(other_base + (int)rec.ind2) * dims, i0, dims, vec_ok, active, w2);

    Calc::block_commit(calc_pf, sh_block, a.calc_params, dims, vec_ok, (int)threadIdx.x, kThreads);
    if (has_block_scratch) __syncthreads();
    GroupCtx<T, EPL, LPW> ctx;
    ctx.sub = sub;
    ctx.dims = dims;
    ctx.lane = lane;
    ctx.stage = Calc::kNeedsStage ? &sh_stage[wib * 64 * EPL] : nullptr;
    ctx.block_scratch = has_block_scratch ? sh_block : nullptr;
    ctx.vec_ok = vec_ok;
    typename Calc::template Regs<EPL, LPW> cregs;
    Calc::template preload<EPL, LPW>(ctx, a.calc_params, cregs);

    // the ensemble step ends with the black half: one lane of the grid advances the per-step counters
    const bool saved_step = sc_phase + 1u == (uint32_t)run_interval;
    if (color == 1 && blockIdx.x == 0 && threadIdx.x == 0)
    {
        DeStepCtl* nx = a.step_ctl + (((h4 >> 1) + 1) & 1);
        nx->step_in_piece = sc_step + 1;
        nx->save_phase = saved_step ? 0u : sc_phase + 1u;
        nx->chain_slot = sc_slot + (saved_step ? 1 : 0);
    }
    const long long save_slot = (run_chain != nullptr && saved_step) ? sc_slot : -1;

    // the jitters of this lane's elements (draws i0 .. i0+EPL-1 behind the integer draws): MultiSampler.h:66
    U128 su = apply(j_uni, rec.s);
    T jit[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e)
    {
        su = pcg_step(su, a.inc);
        const T u = canonical(pcg_output(su), T());
        jit[e] = a.jitter_low + (u * a.jitter_width);
    }
    T prop[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e)
    {
        const T d = w1[e] - w2[e];
        const T gd = a.gamma * d;
        const T moved = own[e] + gd;
        const T p = moved + jit[e];
        prop[e] = (active && i0 + e < dims) ? p : (T)0;  // padded cells stay +0
    }
    const T neg_exp = rec.neg_exp;

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
    if (save_slot >= 0 && active)
    {
        T* crow = reinterpret_cast<T*>(run_chain) + ((size_t)save_slot * (size_t)(2 * n) + (size_t)w) * dims;
        if (accept)
            store_slice<T, EPL>(crow, i0, dims, vec_ok, prop);
        else
            store_slice<T, EPL>(crow, i0, dims, vec_ok, own);
    }
    const unsigned acc = (unsigned)__popcll(__ballot(accept && sub == 0));
    if (run_accepted != nullptr && lane == 0 && acc != 0)
        atomicAdd(run_accepted + (size_t)sc_step * kDeAccSlots + ((blockIdx.x * kWavesPerBlock + wib) & (kDeAccSlots - 1)), acc);
}

}  // namespace mcmcpp

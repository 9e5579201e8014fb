// diffevo_kernel.hpp -- Mover::DifferentialEvolution on gfx950.
//
// Reference: MCMCpp/Movers/DifferentialEvolution.h:80-112.  One update draws, from the sampler's single pcg64 stream,
//   ind1 = engine(n); ind2 = engine(n) until it differs from ind1; D uniform jitters; (Calculator); one exponential
// i.e. D + 3 draws plus every draw thrown away (by pcg's bounded_rand below its threshold, by the ind2 loop).  The
// next walker's first draw follows this walker's last, so a walker's place in the stream depends on how many draws
// all walkers before it threw away -- the reason the reference's loop is sequential.  But whether an update that
// STARTS at a given stream position throws draws away is a property of that position alone (of the two draws there:
// one below the threshold, or both naming the same walker), never of the walkers: call such a position bad, and E(p)
// the number of draws an update starting there throws away.  So the stream is PLANNED AHEAD of the updates, by extra
// workgroups of the very launch that updates an earlier half-step.  One launch per half-step h (de_step_kernel),
// three kinds of workgroup:
//   update   half-step h: like the stretch half-step kernel -- first round trip: the walker's 32-byte record (its two
//            partners, the engine state behind its integer draws, the logarithm of its accept draw), own row,
//            log-posterior, counter; second round trip: the two partner rows; in its shadow the lanes draw their own
//            jitters; calculator, accept in place, optional chain store;
//   records  half-step h + 1: every planning workgroup first RESOLVES that half-step from the list of its bad
//            positions (made one launch earlier): only a bad position p with p mod (D+3) <= 31 can be a walker's start
//            at all (walker k starts at (D+3)k + r with r, the draws thrown away so far in the half-step, at most
//            kDeMaxShift); those few are sorted, and the walk "the next start that is bad for the current r" is a
//            handful of wave ballots (about one event per half-step, whatever n is); then one lane per walker finds its
//            shift by a search in the event list, jumps to its place through three table look-ups, replays its integer
//            draws and leaves the walker's record behind;
//   scan     half-step h + 2 (its base state follows from the resolve above): every stream position the half-step can
//            reach is drawn ONCE -- a lane jumps to eight consecutive positions' start (two table look-ups) and steps
//            through them -- and the bad ones (about D + 3 of them) go to that half-step's list with their E.
// The launches of a run are replayed from a hipGraph; the stream position, the error flags and the per-run counters
// travel in device memory (DeCtl ring, DeStepCtl).  More than kDeMaxShift thrown-away draws in one half-step (expected:
// about one) raise a sticky error flag the host turns into a failed run: never a silently different chain.
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
constexpr int kDeOverrun = 255;   // E of a start whose update would not fit that window
constexpr int kDeMaxEvents = 128; // bad positions that can be a walker's start, per half-step (typical: about 32)
constexpr int kDeScanRun = 2;     // consecutive stream positions one scanning lane steps through (default of DeArgs::scan_run)
constexpr int kDeAccSlots = 64;   // counters an ensemble step's accepted proposals are spread over (same-address atomics serialise)

enum : uint32_t
{
    kDeErrShift = 1u,   // more than kDeMaxShift draws thrown away in one half-step
    kDeErrCand = 2u,    // more bad positions than the lists hold
    kDeErrWindow = 4u,  // one update threw away more than kDeWindow - 2 draws
};

// one per half-step, a ring of four (index = half-step & 3): launch h reads the record of h + 1 and writes that of h + 2
struct alignas(64) DeCtl
{
    U128 state;                      // engine state in front of this half-step's first draw
    unsigned long long extra_total;  // draws thrown away before this half-step
    uint32_t pad[10];
};
static_assert(sizeof(DeCtl) == 64, "one line per record");

// bad-position counters of the half-steps (index = half-step & 3), the sticky error flags, all in one line
struct alignas(64) DeShared
{
    uint32_t bad_count[4];
    uint32_t error;  // kDeErr* bits
    uint32_t pad[11];
};

// a bad stream position of a half-step: p draws behind the half-step's first, an update starting there throws away e draws
struct DeBad
{
    uint32_t p;
    uint32_t e;
};

// resolved list entry (LDS), in walker order: walkers behind k start shift_after draws late
struct DePlan
{
    uint32_t k;
    uint32_t shift_after;
};

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
    DeBad* bad;             // [2][bad_capacity]: bad positions of half-step h live in buffer h & 1
    DeRec<T>* recs;         // [2][n]: records of colour c in buffer c
    const DeRunInfo* run;
    DeStepCtl* step_ctl;    // [2]
    Affine128 half_jump;    // (D+3)*n draws
    const Affine128* jump_hi;     // [ceil(n/256)]  (D+3)*256*m draws
    const Affine128* jump_lo;     // [256]          (D+3)*j draws
    const Affine128* jump_small;  // [max(D, kDeMaxShift) + 2]  j draws
    const Affine128* scan_hi;     // [ceil(scan lanes / 256)]  scan_run*256*m draws
    const Affine128* scan_lo;     // [256]                     scan_run*j draws
    Diag* diag;
    uint64_t threshold;     // (2^64 - n) mod n
    U128 inc;               // pcg stream increment
    T gamma, jitter_low, jitter_width, tie_eps;
    int n, dims, vec_ok;
    int bad_capacity;       // entries of one bad-position list
    int scan_positions;     // stream positions a half-step can reach: (D+3)*n + kDeMaxShift + 1
    int scan_run;           // consecutive positions one scanning lane steps through (the scan tables are built for it)
    int half_step_mod4;     // half-step h & 3 of the update this launch performs (colour = h & 1, ensemble step parity = (h >> 1) & 1)
    int update_blocks;      // workgroups [0, update_blocks) update half-step h (0: a planning-only launch)
    unsigned long long* debug_times;  // diagnostics (MCMCPP_HIP_DE_DEBUG=3): [grid][2] start / end of every workgroup, 100 MHz clock
    int record_blocks;      // the next record_blocks workgroups make the records of half-step h + 1 (0: none, priming); the rest
                            // scan the positions of half-step h + 2 (priming without records: of half-step h + 1, unresolved)
};

__device__ __forceinline__ uint32_t de_bounded(uint64_t v, int n, bool pow2) { return pow2 ? (uint32_t)(v & (uint64_t)(n - 1)) : (uint32_t)(v % (uint64_t)n); }

// ---- scan: every stream position of the half-step whose base state is `state`, drawn once; the bad ones listed ----------
template <class T>
__device__ __forceinline__ void de_scan(const DeArgs<T>& a, U128 state, DeBad* bad, uint32_t* count, int first_lane, int lanes_stride)
{
    const int n = a.n;
    const bool pow2 = (n & (n - 1)) == 0;
    const uint64_t threshold = a.threshold;
    const int positions = a.scan_positions, run = a.scan_run;
    for (int t = first_lane + (int)threadIdx.x; t * run < positions; t += lanes_stride)
    {
        // the state behind run * t draws, then position after position
        U128 s = apply(a.scan_lo[t & 255], apply(a.scan_hi[t >> 8], state));
        s = pcg_step(s, a.inc);
        uint64_t raw = pcg_output(s);
#pragma unroll 1
        for (int i = 0; i < run; ++i)
        {
            const int p = t * run + i;
            const U128 s_next = pcg_step(s, a.inc);
            const uint64_t nxt = pcg_output(s_next);
            // bad: a draw below the threshold, or both draws naming the same walker -- an update starting here throws draws away
            const bool is_bad = p < positions && (raw < threshold || nxt < threshold || de_bounded(raw, n, pow2) == de_bounded(nxt, n, pow2));
            if (is_bad)
            {
                // DifferentialEvolution.h:83-87 from this position on (rare: about D + 3 positions of a half-step)
                U128 w = s;
                uint64_t v = raw;
                int used = 1;
                while (v < threshold && used < kDeWindow)
                {
                    w = pcg_step(w, a.inc);
                    v = pcg_output(w);
                    ++used;
                }
                const uint32_t ind1 = de_bounded(v, n, pow2);
                bool overrun = v < threshold;
                uint32_t ind2 = ind1;
                while (!overrun && ind2 == ind1)
                {
                    do
                    {
                        if (used >= kDeWindow)
                        {
                            overrun = true;
                            break;
                        }
                        w = pcg_step(w, a.inc);
                        v = pcg_output(w);
                        ++used;
                    } while (v < threshold);
                    if (!overrun) ind2 = de_bounded(v, n, pow2);
                }
                const uint32_t slot = atomicAdd(count, 1u);
                if (slot < (uint32_t)a.bad_capacity)
                {
                    DeBad b;
                    b.p = (uint32_t)p;
                    b.e = overrun ? (uint32_t)kDeOverrun : (uint32_t)(used - 2);
                    bad[slot] = b;
                }
                else
                    atomicOr(&a.shared->error, kDeErrCand);
            }
            s = s_next;
            raw = nxt;
        }
    }
}

// hot_bits of de_step_kernel: dims | half_step_mod4 << 12 | vec_ok << 14
__host__ __device__ inline uint32_t de_hot_bits(int dims, int half_step_mod4, int vec_ok) { return (uint32_t)dims | ((uint32_t)half_step_mod4 << 12) | ((uint32_t)vec_ok << 14); }

// The hot_* arguments are what an updating wavefront needs before its second round trip; they travel in the 16 dwords
// the command processor preloads into SGPRs (see HotBits in stretch_kernel.hpp), everything else in `a`, whose cold
// kernarg lines an updating wavefront touches only behind its second trip's loads.
template <class T, class Calc, int EPL, int LPW>
__global__ void __launch_bounds__(64 * kWavesPerBlock)
de_step_kernel(T* hot_pos, T* hot_logp, uint32_t* hot_n_accept, const DeRec<T>* hot_recs, const Affine128* hot_jump_small, DeStepCtl* hot_step_ctl, int hot_n,
               uint32_t hot_bits, int hot_planner_blocks, const DeArgs<T> a)
{
    constexpr int WPP = 64 / LPW;
    constexpr int kThreads = 64 * kWavesPerBlock;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int dims = (int)(hot_bits & 0xFFFu), n = hot_n;
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int h4 = (int)((hot_bits >> 12) & 3u);

    struct StampWorkgroup  // (diagnostics: nothing unless a buffer is given)
    {
        unsigned long long* slot;
        __device__ StampWorkgroup(unsigned long long* base) : slot(base ? base + 2 * (size_t)blockIdx.x : nullptr)
        {
            if (slot && threadIdx.x == 0) slot[0] = __builtin_amdgcn_s_memrealtime();
        }
        __device__ ~StampWorkgroup()
        {
            if (slot && threadIdx.x == 0) slot[1] = __builtin_amdgcn_s_memrealtime();
        }
    } stamp(a.debug_times);
    // (the planners come first in the grid: theirs is the longer dependent chain)
    if ((int)blockIdx.x < hot_planner_blocks)
    {
        // =================================== planning workgroups ==========================================================
        __shared__ DePlan sh_plan[kDeMaxEvents];
        __shared__ DeBad sh_rel[kDeMaxEvents];     // the bad positions that can be a walker's start, as found
        __shared__ DeBad sh_sorted[kDeMaxEvents];  // by position
        __shared__ int sh_rel_count, sh_events, sh_total;
        const int pb = (int)blockIdx.x;  // planner index
        const int planners = hot_planner_blocks;
        const int h1 = (h4 + 1) & 3, h2 = (h4 + 2) & 3;
        const unsigned per = (unsigned)dims + 3u;
        const DeCtl ctl1 = a.ctl[h1];
        if (a.record_blocks == 0)
        {
            // priming: nothing to resolve yet -- the positions of half-step h + 1, straight from its record in the ring
            de_scan<T>(a, ctl1.state, a.bad + (size_t)(h1 & 1) * a.bad_capacity, &a.shared->bad_count[h1], pb * kThreads, planners * kThreads);
            return;
        }
        // ---- resolve half-step h + 1 from its bad positions (listed one launch ago) ----
        const DeBad* bad1 = a.bad + (size_t)(h1 & 1) * a.bad_capacity;
        const uint32_t listed = a.shared->bad_count[h1];
        const int bad_count = (int)(listed < (uint32_t)a.bad_capacity ? listed : (uint32_t)a.bad_capacity);
        if (threadIdx.x == 0) sh_rel_count = 0;
        __syncthreads();
        for (int j = threadIdx.x; j < bad_count; j += kThreads)
        {
            const DeBad b = bad1[j];
            // walker k starts at per * k + r with r <= kDeMaxShift: a position whose residue mod per is larger cannot be a start
            if (b.p % per <= (unsigned)kDeMaxShift)
            {
                const int slot = atomicAdd(&sh_rel_count, 1);
                if (slot < kDeMaxEvents) sh_rel[slot] = b;
            }
        }
        __syncthreads();
        int rel = sh_rel_count;
        uint32_t err = 0;
        if (rel > kDeMaxEvents)
        {
            err |= kDeErrCand;
            rel = kDeMaxEvents;
        }
        for (int j = threadIdx.x; j < rel; j += kThreads)
        {
            int rank = 0;
            const uint32_t mine = sh_rel[j].p;
            for (int i = 0; i < rel; ++i) rank += sh_rel[i].p < mine ? 1 : 0;  // (positions are distinct)
            sh_sorted[rank] = sh_rel[j];
        }
        __syncthreads();
        if (wib == 0)
        {
            // The walk, in stream order: with r draws thrown away so far, walker k starts at per * k + r; the next event is
            // the first listed position that IS the start of a walker behind the last event's.  One ballot per event;
            // lane l holds entries l and l + 64.
            static_assert(kDeMaxEvents == 128, "two entries per lane");
            const DeBad e0 = lane < rel ? sh_sorted[lane] : DeBad{0u, 0u};
            const DeBad e1 = lane + 64 < rel ? sh_sorted[lane + 64] : DeBad{0u, 0u};
            int r = 0, events = 0, last_k = -1;  // (a position inside the draws of the walker whose start was the last event is no start)
            while (true)
            {
                // is this entry the start of a walker k in (last_k, n) when r draws have been thrown away?
                auto starts = [&](const DeBad& e, bool have) -> bool {
                    if (!have || e.p < (uint32_t)r) return false;
                    const uint32_t d = e.p - (uint32_t)r;
                    const uint32_t q = d / per;
                    return d - q * per == 0u && (int)q > last_k && (int)q < n;
                };
                const unsigned long long m0 = __ballot(starts(e0, lane < rel));
                const unsigned long long m1 = __ballot(starts(e1, lane + 64 < rel));
                int at;
                if (m0)
                    at = __ffsll((long long)m0) - 1;
                else if (m1)
                    at = 64 + __ffsll((long long)m1) - 1;
                else
                    break;
                const uint32_t ep = at < 64 ? __shfl(e0.p, at) : __shfl(e1.p, at - 64);
                uint32_t own = at < 64 ? __shfl(e0.e, at) : __shfl(e1.e, at - 64);
                if (own == (uint32_t)kDeOverrun)
                {
                    err |= kDeErrWindow;
                    own = 0;
                }
                const int ek = (int)((ep - (uint32_t)r) / per);  // (r: still the shift this walker starts with)
                r += (int)own;
                if (r > kDeMaxShift)
                {
                    err |= kDeErrShift;
                    r = kDeMaxShift;
                }
                if (lane == 0)
                {
                    sh_plan[events].k = (uint32_t)ek;
                    sh_plan[events].shift_after = (uint32_t)r;
                }
                ++events;
                last_k = ek;
            }
            if (lane == 0)
            {
                sh_events = events;
                sh_total = r;
                if (pb == 0)
                {
                    // hand the stream on: the record of half-step h + 2 (its last reader was the launch before this one),
                    // and clear the bad-position counter of half-step h + 3 (scanned by the next launch)
                    DeCtl* nx = a.ctl + h2;
                    nx->state = apply(a.jump_small[r], apply(a.half_jump, ctl1.state));
                    nx->extra_total = ctl1.extra_total + (unsigned long long)r;
                    a.shared->bad_count[(h4 + 3) & 3] = 0;
                    if (err) atomicOr(&a.shared->error, err);
                }
            }
        }
        __syncthreads();
        if (pb < a.record_blocks)
        {
            // ---- records of half-step h + 1: one lane per walker ----
            const int k = pb * kThreads + (int)threadIdx.x;
            if (k >= n) return;
            const int plan_count = sh_events;
            // this walker's place in the stream: the last event in front of it says how late it starts
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
            U128 s = apply(a.jump_small[shift], apply(a.jump_lo[k & 255], apply(a.jump_hi[k >> 8], ctl1.state)));
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
            const U128 se = pcg_step(apply(a.jump_small[dims], s), a.inc);
            DeRec<T>* out = a.recs + (size_t)(h1 & 1) * n + k;
            out->s = s;
            out->neg_exp = dev_log((T)1 - canonical(pcg_output(se), T()));  // -(-log(1 - u)/1)
            out->ind1 = ind1;
            out->ind2 = ind2;
            return;
        }
        // ---- the positions of half-step h + 2: its base state follows from the resolve ----
        const U128 state2 = apply(a.jump_small[sh_total], apply(a.half_jump, ctl1.state));
        const int scanners = planners - a.record_blocks;
        de_scan<T>(a, state2, a.bad + (size_t)(h2 & 1) * a.bad_capacity, &a.shared->bad_count[h2], (pb - a.record_blocks) * kThreads, scanners * kThreads);
        return;
    }

    // ======================================= update workgroups: half-step h ===============================================
    T* sh_stage = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::stage_offset());
    T* sh_block = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::block_offset());
    const bool vec_ok = ((hot_bits >> 14) & 1u) != 0;
    const bool has_block_scratch = Calc::block_scratch_elems(dims) != 0;
    const int color = h4 & 1;
    const int sub = lane & (LPW - 1);
    const int ub = (int)blockIdx.x - hot_planner_blocks;                    // updating workgroup
    const int k = (ub * kWavesPerBlock + wib) * WPP + lane / LPW;           // walker inside the half
    const bool active = k < n;
    const int kk = active ? k : 0;
    const int half_base = color ? n : 0, other_base = color ? 0 : n;
    const int w = half_base + kk;
    const int i0 = sub * EPL;

    // first round trip, from preloaded arguments only: what the walker index alone addresses
    const DeRec<T> rec = hot_recs[(size_t)color * n + kk];
    T own[EPL];
    load_slice<T, EPL>(hot_pos + (size_t)w * dims, i0, dims, vec_ok, active, own);
    const T lp_old = hot_logp[w];
    const uint32_t nacc_old = hot_n_accept[w];
    const Affine128 j_uni = hot_jump_small[i0 < dims ? i0 : dims];
    // the per-step counters and, right behind the two records of them, the run's constants (one allocation: DeStepCtl[2],
    // DeRunInfo), through a preloaded pointer: one more load of the first round trip instead of a round trip of its own
    // (field by field: a whole-record copy drags the padding through registers and scratch)
    const DeStepCtl* scp = hot_step_ctl + ((h4 >> 1) & 1);
    const DeRunInfo* rip = reinterpret_cast<const DeRunInfo*>(hot_step_ctl + 2);
    const long long sc_step = scp->step_in_piece, sc_slot = scp->chain_slot;
    const uint32_t sc_phase = scp->save_phase;
    void* const run_chain = rip->chain;
    uint32_t* const run_accepted = rip->accepted;
    const long long run_interval = rip->interval;

    // second round trip: the two partner rows
    T w1[EPL], w2[EPL];
    load_slice<T, EPL>(hot_pos + (size_t)(other_base + (int)rec.ind1) * dims, i0, dims, vec_ok, active, w1);
    load_slice<T, EPL>(hot_pos + (size_t)(other_base + (int)rec.ind2) * dims, i0, dims, vec_ok, active, w2);
    typename Calc::Prefetch calc_pf;
    Calc::block_prefetch(calc_pf, a.calc_params, dims, vec_ok, (int)threadIdx.x, kThreads);

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
    if (color == 1 && ub == 0 && threadIdx.x == 0)
    {
        DeStepCtl* nx = hot_step_ctl + (((h4 >> 1) + 1) & 1);
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
        store_slice<T, EPL>(hot_pos + (size_t)w * dims, i0, dims, vec_ok, prop);
        if (sub == 0)
        {
            hot_logp[w] = lp_new;
            hot_n_accept[w] = nacc_old + 1u;
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
        atomicAdd(run_accepted + (size_t)sc_step * kDeAccSlots + ((ub * kWavesPerBlock + wib) & (kDeAccSlots - 1)), acc);
}

}  // namespace mcmcpp

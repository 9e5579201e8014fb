// diffevo_kernel.hpp -- Mover::DifferentialEvolution on gfx950.
//
// Reference: MCMCpp/Movers/DifferentialEvolution.h:80-112.  One update draws, from the sampler's single pcg64 stream,
//   ind1 = engine(n); ind2 = engine(n) until it differs from ind1; D uniform jitters; (Calculator); one exponential
// i.e. D + 3 draws plus every draw thrown away (by pcg's bounded_rand below its threshold, by the ind2 loop).  The
// next walker's first draw follows this walker's last, so a walker's place in the stream depends on how many draws
// all walkers before it threw away -- the reason the reference's loop is sequential.  But whether an update that
// STARTS at a given stream position throws draws away is a property of that position alone (of the two draws there:
// one below the threshold, or both naming the same walker), never of the walkers: call such a position bad, and E(p)
// the number of draws an update starting there throws away.  Update m of the stream (m counts updates across
// half-steps) starts at position (D+3) m + c, c the draws thrown away before it; c changes only where a start is bad
// (an EVENT: about one per half-step, whatever n is).  So the stream is PLANNED, a batch of up to kDeBatchMax
// half-steps at a time, by three roles that depend on nothing but the stream:
//   scan     every stream position the batch can reach is drawn ONCE -- a lane jumps to its run of positions through two
//            table look-ups and steps through it -- and the bad ones go, with their E, to one of kDeSegments lists (a
//            counter per list on a line of its own: same-line atomics serialise);
//   resolve  one workgroup sorts the bad positions by residue mod (D+3) in LDS and walks the events: "the first bad
//            position that IS the start of an update behind the last event, given c" is one minimum over a wavefront
//            among the handful of positions of one residue; it leaves the event list and the stream head behind the batch;
//   records  one lane per update: its shift from a search in the event list, a jump to its place (three table
//            look-ups), its integer draws replayed, and the update's 32-byte record -- the two partners, the engine
//            state behind the integer draws, the logarithm of its accept draw.
// The updates themselves are one launch per half-step (de_update_kernel), the stretch half-step kernel's structure:
// first round trip the record, own row, log-posterior, counter; second round trip the two partner rows; in its shadow
// the lanes draw their own jitters; calculator, accept in place, optional chain store, the wavefront's accepted count.
// The planning runs as ONE launch of its own between the update launches of two batches:
//   de_boundary_kernel   in front of batch b: the resolve of batch b + 1 (one workgroup), the records of batch b (from the
//                        resolve made one boundary earlier) and the scan of batch b + 2, all independent of each other --
//                        the scan does not wait for the resolve to say where batch b + 2 begins: it starts from where it
//                        would begin had batch b + 1 thrown no draws away (DeHead::provisional) and looks kDeShiftMax + 1
//                        positions further; the resolve of b + 2 subtracts what b + 1 did throw away.
// Measured against two alternatives that hide the planning instead of adding it: on a second stream beside the
// update launches (the cross-queue dependencies and the chip-filling scan cost more than they hide), and as extra
// workgroups of the update launches themselves (the 128-bit integer arithmetic of scan and jitters then competes for the
// same vector ALUs: every launch stretches by as much as the planning takes on its own).
// A run's launches are replayed from a hipGraph; the stream head, the sticky error flags and the per-run counters travel
// in device memory.  A batch that throws away more than kDeShiftMax draws, lists more bad positions than the resolver
// holds, or contains an update that throws away more than kDeWindow - 2 raises an error flag the host turns into a
// failed run: never a silently different chain.
#pragma once

#include "stretch_kernel.hpp"

namespace mcmcpp
{
constexpr int kDeBatchMax = 64;     // half-steps one plan covers at most
constexpr int kDeShiftMax = 4095;   // thrown-away draws inside one batch that are followed exactly (expected: about one per half-step)
constexpr int kDeWindow = 32;       // raw draws one update's integer part may consume (2 + up to 30 thrown away: even two
                                    // walkers per half, where every second ind2 collides, overrun once in 1e9 updates)
constexpr int kDeOverrun = 255;     // E of a start whose update would not fit that window
constexpr int kDeSegments = 64;     // bad-position lists of a batch (by position)
constexpr int kDeCountStride = 32;  // uint32 between two list counters: one 128-byte line each
constexpr int kDeMaxBad = 8192;     // bad positions of a batch the resolver holds (about (D + 3) per half-step)
constexpr int kDeMaxEvents = 1024;  // events of a batch
constexpr int kDeScanRun = 8;       // consecutive stream positions one scanning lane steps through (default)
constexpr int kDePlanThreads = 256;

enum : uint32_t
{
    kDeErrShift = 1u,   // more than kDeShiftMax draws thrown away in one batch
    kDeErrCand = 2u,    // more bad positions or events than the lists hold
    kDeErrWindow = 4u,  // one update threw away more than kDeWindow - 2 draws
};

// the stream between batches
struct alignas(64) DeHead
{
    U128 state;                      // engine state in front of the next batch to be resolved
    unsigned long long extra_total;  // draws thrown away so far
    uint32_t error;                  // kDeErr* bits, sticky
    uint32_t last_shift;             // draws thrown away inside the batch resolved last
    // Where a batch is scanned from before the batch in front of it has been resolved: [b & 1] = the state in front of
    // batch b had batch b - 1 thrown no draws away; its true first draw lies last_shift (of b - 1) draws further on.
    U128 provisional[2];
};
static_assert(sizeof(DeHead) == 64, "one line");

// a bad stream position of a batch: p draws behind the batch's first, an update starting there throws away e draws
struct DeBad
{
    uint32_t p;
    uint32_t e;
};

// an event: update m of the batch started at a bad position; the updates behind it start shift_after draws late
struct DePlan
{
    uint32_t m;
    uint32_t shift_after;
};

// what the resolver leaves for the record lanes
struct alignas(64) DeBatch
{
    U128 base;  // engine state in front of the batch's first draw
    unsigned long long extra_base;  // draws thrown away before the batch
    uint32_t events;
    uint32_t pad[9];
    DePlan plan[kDeMaxEvents];
};

// The stream part of one update (DifferentialEvolution.h:83-87,100)
template <class T>
struct alignas(16) DeRec
{
    U128 s;           // engine state behind the integer draws: the D jitter draws and the accept draw follow
    T neg_exp;        // -(-log(1 - u)/1) of the accept draw
    uint32_t ind1, ind2;
};
static_assert(sizeof(DeRec<double>) == 32 && sizeof(DeRec<float>) == 32, "32-byte records");

// Per-run record: uploaded by the host in front of a piece of the run, advanced by de_advance_kernel behind every
// graph replay.  The wavefronts' accepted counts of a replay ([step in replay][colour][wavefront]) follow it directly
// (one allocation: the update kernel reaches both through one preloaded pointer).
struct alignas(64) DeRunInfo
{
    void* chain;         // device chain [slots][W][D], or nullptr
    uint32_t* accepted;  // [steps of this piece], or nullptr
    long long slot0;     // chain slot the next stored step goes to
    long long step0;     // ensemble steps of this piece behind us
    uint32_t interval;   // EnsembleSampler.h:284-310: interval - 1 unsaved steps, one saved
    uint32_t phase0;     // unsaved steps since the last saved one
    uint32_t pad[6];
};
static_assert(sizeof(DeRunInfo) == 64, "one line");

struct DePlanArgs
{
    DeHead* head;
    DeBatch* batch;               // the record the resolve writes
    DeBad* bad;                   // [kDeSegments][bad_capacity]: the lists the scan fills
    uint32_t* counts;             // [kDeSegments] at stride kDeCountStride
    const DeBad* resolve_bad;     // the lists the resolve reads (the other of two sets: a scan runs beside it)
    uint32_t* resolve_counts;
    int scan_parity;              // the scan's batch & 1 (which provisional state it starts from)
    const Affine128* scan_hi;     // [ceil(scan lanes / 256)]  scan_run*256*j draws
    const Affine128* scan_lo;     // [256]                     scan_run*j draws
    const Affine128* jump_hi;     // [ceil(updates / 256)]     (D+3)*256*j draws
    const Affine128* jump_lo;     // [256]                     (D+3)*j draws
    const Affine128* jump_small;  // [kDeShiftMax + D + 2]     j draws
    Affine128 batch_jump;         // (D+3) * updates draws
    U128 inc;                     // pcg stream increment
    uint64_t threshold;           // (2^64 - n) mod n
    int n, dims;
    int updates;                  // n * half-steps of this batch
    int positions;                // stream positions an update of the batch can start at: (D+3)(updates - 1) + kDeShiftMax + 1
    int scan_positions;           // positions the scan looks at, from its provisional start: positions + kDeShiftMax + 1
    int scan_run;                 // consecutive positions one scanning lane steps through (the scan tables are built for it)
    int seg_len;                  // positions per list
    int bad_capacity;             // entries of one list
};

template <class T>
struct DeArgs
{
    const T* calc_params;
    Diag* diag;
    U128 inc;
    T gamma, jitter_low, jitter_width, tie_eps;
    int partial_waves;  // updating wavefronts of a launch
};

// hot_bits of de_update_kernel: dims | colour << 12 | vec_ok << 14
__host__ __device__ inline uint32_t de_hot_bits(int dims, int color, int vec_ok, int step = 0)
{
    return (uint32_t)dims | ((uint32_t)color << 12) | ((uint32_t)vec_ok << 14) | ((uint32_t)step << 16);
}

// The run record and the two lines of the launch description (kernarg) that hold DeArgs, in ONE batch of scalar loads with
// one wait -- called behind the first round trip's vector loads, where the wait is free.  Left to the compiler these are cold
// scalar misses issued where a field is first needed: behind the partner gather, with a wait that the matrix loads and the
// jitter arithmetic then sit behind (see load_records_and_warm_args in stretch_kernel.hpp; the kernels' argument lists are
// 64 bytes of preloaded hot arguments followed by the DeArgs).
template <class T>
__device__ __forceinline__ DeRunInfo de_load_run_and_warm_args(const DeRunInfo* run_ptr)
{
    static_assert(sizeof(DeRunInfo) == 64 && 64 + sizeof(DeArgs<T>) <= 0xc0, "one s_load_dwordx16; adjust the lines touched below");
    typedef unsigned v16u __attribute__((ext_vector_type(16)));
    const unsigned long long k = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    const unsigned long long v = (unsigned long long)run_ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    const unsigned long long run_addr = ((unsigned long long)hi << 32) | lo;
    v16u r;
    unsigned t1, t2;
    asm volatile(
        "s_load_dwordx16 %0, %3, 0x0\n\t"
        "s_load_dword %1, %4, 0x40\n\t"
        "s_load_dword %2, %4, 0x80\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(r), "=&s"(t1), "=&s"(t2)
        : "s"(run_addr), "s"(k)
        : "memory");
    return __builtin_bit_cast(DeRunInfo, r);
}

// ---- the update of one half-step ----------------------------------------------------------------------------------------
// The hot_* arguments are what an updating wavefront needs before its second round trip; they travel in the 16 dwords
// the command processor preloads into SGPRs (see HotBits in stretch_kernel.hpp), everything else in `a`.
// hot_recs: the n records of THIS half-step; hot_step: ensemble step inside the graph replay (0 for plain launches).
template <class T, class Calc, int EPL, int LPW>
__global__ void __launch_bounds__(64 * kWavesPerBlock)
de_update_kernel(T* hot_pos, T* hot_logp, uint32_t* hot_n_accept, const DeRec<T>* hot_recs, const Affine128* hot_jump_small, DeRunInfo* hot_run, int hot_n,
                 uint32_t hot_bits, int hot_step, const DeArgs<T> a)
{
    constexpr int WPP = 64 / LPW;
    constexpr int kThreads = 64 * kWavesPerBlock;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int dims = (int)(hot_bits & 0xFFFu), n = hot_n;
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int color = (int)((hot_bits >> 12) & 1u);
    T* sh_stage = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::stage_offset());
    T* sh_block = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::block_offset());
    const bool vec_ok = ((hot_bits >> 14) & 1u) != 0;
    const bool has_block_scratch = Calc::block_scratch_elems(dims) != 0;
    const int sub = lane & (LPW - 1);
    const int wave = (int)blockIdx.x * kWavesPerBlock + wib;  // updating wavefront
    const int k = wave * WPP + lane / LPW;                    // walker inside the half
    const bool active = k < n;
    const int kk = active ? k : 0;
    const int half_base = color ? n : 0, other_base = color ? 0 : n;
    const int w = half_base + kk;
    const int i0 = sub * EPL;

    // first round trip, from preloaded arguments only: what the walker index alone addresses
    const DeRec<T> rec = hot_recs[kk];
    T own[EPL];
    load_slice<T, EPL>(hot_pos + (size_t)w * dims, i0, dims, vec_ok, active, own);
    const T lp_old = hot_logp[w];
    const uint32_t nacc_old = hot_n_accept[w];
    // (the jump to this lane's FIRST jitter, one draw beyond its first element's index: no step in front of it)
    const Affine128 j_uni = hot_jump_small[i0 < dims ? i0 + 1 : dims];
    // the run record, field by field (needed behind the calculator: the loads have the whole launch to land)
    void* const run_chain = assume_global(hot_run->chain);
    const long long run_slot0 = hot_run->slot0;
    const uint32_t run_interval = hot_run->interval, run_phase0 = hot_run->phase0;

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

    // is this ensemble step stored, and where (EnsembleSampler.h:284-310)
    const uint32_t since = run_phase0 + (uint32_t)hot_step;  // unsaved steps in front of this one, counted from the last saved step of the previous replay
    const uint32_t whole = since / run_interval;
    const bool saved_step = since - whole * run_interval + 1u == run_interval;
    long long save_slot = (run_chain != nullptr && saved_step) ? run_slot0 + (long long)whole : -1;
    // (materialised HERE, where every load has landed anyway: left to the compiler, the first look at the run record comes
    //  behind the accept stores, and the only wait that covers a load in front of a data-dependent number of stores is
    //  vmcnt(0) -- every updating wavefront would sit out the acknowledgement of its own stores)
    asm volatile("" : "+v"(save_slot));

    // the jitters of this lane's elements (draws i0 .. i0+EPL-1 behind the integer draws): MultiSampler.h:66
    U128 su = apply(j_uni, rec.s);
    T jit[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e)
    {
        if (e > 0) su = pcg_step(su, a.inc);
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
        if (margin <= a.tie_eps * scale) count_near_tie(a.diag);
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
    // the wavefront's accepted proposals: a plain store (thousands of atomics on a few lines serialise), summed per
    // ensemble step by de_accepted_kernel behind the replay
    const unsigned acc = (unsigned)__popcll(__ballot(accept && sub == 0));
    if (lane == 0)
    {
        uint32_t* partials = reinterpret_cast<uint32_t*>(hot_run + 1);
        partials[((size_t)hot_step * 2 + (size_t)color) * (size_t)a.partial_waves + (size_t)wave] = acc;
    }
}

// ---- the same update for the dense Gaussian on the matrix cores (fp64, 17..32 even dimensions) ----------------------------
// As stretch_half_step_mfma_kernel: a wavefront's 4 P walkers are the rows of one 16-row tile -- walker (pass q, lane
// group g) is row 4 q + g; lane (g, sub) holds elements 2 sub, 2 sub + 1 of its P walkers -- and Y = X * P^T runs on
// v_mfma_f64_16x16x4_f64 (mc_eval: bit for bit the host calculator's fma chain).  Everything else is de_update_kernel.
template <class T, class Calc, int EPL, int LPW, int P>
__global__ void __launch_bounds__(64 * kWavesPerBlock)
de_update_mfma_kernel(T* hot_pos, T* hot_logp, uint32_t* hot_n_accept, const DeRec<T>* hot_recs, const Affine128* hot_jump_small, DeRunInfo* hot_run, int hot_n,
                      uint32_t hot_bits, const T* hot_matrix, const DeArgs<T> a)
{
    static_assert(EPL == 2 && LPW == 16, "matrix-core path: 16 lanes x 2 elements per walker, 16 < D <= 32");
    constexpr int NW = 4 * P;  // walkers per wavefront
    // LDS: proposal rows, NW x kMcXS per wavefront.  The wavefront's share of P^T (zero-padded to 32 x 32 by the host) comes
    // straight from memory into registers through a preloaded pointer, as in the stretch kernels: no LDS copy of the
    // matrix, no workgroup barrier.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kStageRows = sizeof(T) == 8 ? NW : 16;  // rows of the tile that hold walkers (mc_row)
    T* sh_x = reinterpret_cast<T*>(smem) + (threadIdx.x >> 6) * (kStageRows * kMcXS);
    const int dims = (int)(hot_bits & 0xFFFu), n = hot_n;
    const int hot_step = (int)(hot_bits >> 16);  // ensemble step inside the graph replay
    const int lane = threadIdx.x & 63;
    const int color = (int)((hot_bits >> 12) & 1u);
    const int wave = (int)blockIdx.x * kWavesPerBlock + (int)(threadIdx.x >> 6);
    const int first = wave * NW;
    const bool wave_active = first < n;
    const int half_base = color ? n : 0, other_base = color ? 0 : n;
    const int sub = lane & 15, grp = lane >> 4, i0 = sub * 2;
    // (the host only selects this kernel for even D: rows are whole 16-byte pieces, every access is branch-free)
    typedef typename Vec2<T>::type V2;
    const bool col_ok = i0 < dims;
    const int i0c = col_ok ? i0 : 0;

    // first round trip, from preloaded arguments only
    DeRec<T> rec[P];
    T own[P][2], lp_old[P];
    uint32_t nacc_old[P];
    bool active[P];
    int w[P];
#pragma unroll
    for (int q = 0; q < P; ++q)
    {
        const int k = first + 4 * q + grp;
        active[q] = wave_active && k < n;
        const int kk = active[q] ? k : 0;
        w[q] = half_base + kk;
        rec[q] = hot_recs[kk];
        const V2 v = *reinterpret_cast<const V2*>(hot_pos + (size_t)w[q] * dims + i0c);
        own[q][0] = (active[q] && col_ok) ? v.x : (T)0;
        own[q][1] = (active[q] && col_ok) ? v.y : (T)0;
        lp_old[q] = hot_logp[w[q]];
        nacc_old[q] = hot_n_accept[w[q]];
    }
    // the jump from the record's stream position to this lane's FIRST jitter (jump_small[j]: j draws; the table has more than
    // D + 1 entries): a jump and one step per pass instead of a jump and two steps -- a 128-bit multiplication in three
    // fewer; the update launch of a large ensemble is bound by its quarter-rate integer multiplies
    const Affine128 j_uni = hot_jump_small[i0 < dims ? i0 + 1 : dims];
    // The run record and the launch description's cold lines: in one batch of scalar loads behind the vector loads above for
    // the 8-walker wavefronts of small, latency-bound launches (C2's half-step 4.95 -> 4.66 us: left to the compiler the
    // kernarg miss sits behind the partner gather, in front of the matrix loads and the jitter arithmetic); field by field,
    // where first needed, for the 16-walker wavefronts of large ones, where the early wait costs more than it saves
    // (262 144 walkers: 30.3 against 30.9 us; profiles/r03_de_warm.txt).
    DeRunInfo run_rec;
    if constexpr (P == 2)
        run_rec = de_load_run_and_warm_args<T>(hot_run);
    else
    {
        run_rec.chain = hot_run->chain;
        run_rec.slot0 = hot_run->slot0;
        run_rec.interval = hot_run->interval;
        run_rec.phase0 = hot_run->phase0;
    }
    void* const run_chain = assume_global(run_rec.chain);
    const long long run_slot0 = run_rec.slot0;
    const uint32_t run_interval = run_rec.interval, run_phase0 = run_rec.phase0;

    // second round trip: the two partner rows of every pass
    T w1[P][2], w2[P][2];
#pragma unroll
    for (int q = 0; q < P; ++q)
    {
        const V2 v1 = *reinterpret_cast<const V2*>(hot_pos + (size_t)(other_base + (active[q] ? (int)rec[q].ind1 : 0)) * dims + i0c);
        const V2 v2 = *reinterpret_cast<const V2*>(hot_pos + (size_t)(other_base + (active[q] ? (int)rec[q].ind2 : 0)) * dims + i0c);
        w1[q][0] = v1.x, w1[q][1] = v1.y, w2[q][0] = v2.x, w2[q][1] = v2.y;
    }
    // behind the gather (so that it does not compete with the records the gather waits for): the wavefront's share of P^T
    asm volatile("" ::: "memory");
    McB<T> matB;
    mc_load_b(hot_matrix, sub, grp, matB);
    if (!wave_active) return;
    // in the gather's shadow: the lanes draw their jitters

    const uint32_t since = run_phase0 + (uint32_t)hot_step;
    const uint32_t whole = since / run_interval;
    const bool saved_step = since - whole * run_interval + 1u == run_interval;
    long long save_slot = (run_chain != nullptr && saved_step) ? run_slot0 + (long long)whole : -1;
    asm volatile("" : "+v"(save_slot));  // (see de_update_kernel)

    T prop[P][2];
#pragma unroll
    for (int q = 0; q < P; ++q)
    {
        // the jitters of this lane's elements (draws i0, i0 + 1 behind the integer draws): MultiSampler.h:66
        U128 su = apply(j_uni, rec[q].s);
#pragma unroll
        for (int e = 0; e < 2; ++e)
        {
            if (e > 0) su = pcg_step(su, a.inc);
            const T u = canonical(pcg_output(su), T());
            const T jit = a.jitter_low + (u * a.jitter_width);
            const T d = w1[q][e] - w2[q][e];
            const T gd = a.gamma * d;
            const T moved = own[q][e] + gd;
            const T p = moved + jit;
            prop[q][e] = (active[q] && col_ok) ? p : (T)0;  // padded cells stay +0
        }
    }
    T lp_new[P];
    mc_eval<P>(matB, sh_x, sub, grp, dims, prop, lp_new);

    unsigned acc = 0;
#pragma unroll
    for (int q = 0; q < P; ++q)
    {
        const T neg_exp = rec[q].neg_exp;
        const T delta = lp_new[q] - lp_old[q];
        const bool accept = active[q] && (delta > neg_exp);  // DifferentialEvolution.h:100
        if (active[q] && sub == 0)
        {
            const T margin = dev_abs(neg_exp - delta);
            const T scale = dev_abs(neg_exp) + dev_abs(lp_new[q]) + dev_abs(lp_old[q]);
            if (margin <= a.tie_eps * scale) count_near_tie(a.diag);
        }
        if (accept)
        {
            if (col_ok) *reinterpret_cast<V2*>(hot_pos + (size_t)w[q] * dims + i0) = Vec2<T>::make(prop[q][0], prop[q][1]);
            if (sub == 0)
            {
                hot_logp[w[q]] = lp_new[q];
                hot_n_accept[w[q]] = nacc_old[q] + 1u;
            }
        }
        if (save_slot >= 0 && active[q] && col_ok)
        {
            T* crow = reinterpret_cast<T*>(run_chain) + ((size_t)save_slot * (size_t)(2 * n) + (size_t)w[q]) * dims;
            *reinterpret_cast<V2*>(crow + i0) = accept ? Vec2<T>::make(prop[q][0], prop[q][1]) : Vec2<T>::make(own[q][0], own[q][1]);
        }
        acc += (unsigned)__popcll(__ballot(accept && sub == 0));
    }
    if (lane == 0)
    {
        uint32_t* partials = reinterpret_cast<uint32_t*>(hot_run + 1);
        partials[((size_t)hot_step * 2 + (size_t)color) * (size_t)a.partial_waves + (size_t)wave] = acc;
    }
}

}  // namespace mcmcpp

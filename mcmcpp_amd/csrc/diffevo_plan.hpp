// diffevo_plan.hpp -- the stream planning of Mover::DifferentialEvolution (scheme: diffevo_kernel.hpp): scan, resolve,
// records, and the two small launches behind a replay.  They depend on the random stream alone, not on the calculator:
// compiled once, in diffevo.hip.
#pragma once

#include "diffevo_kernel.hpp"

namespace mcmcpp
{
__device__ __forceinline__ uint32_t de_bounded(uint64_t v, int n, bool pow2) { return pow2 ? (uint32_t)(v & (uint64_t)(n - 1)) : (uint32_t)(v % (uint64_t)n); }

// ---- scan: every stream position of the batch, drawn once; the bad ones listed ------------------------------------------
// t: the scanning lane (of the whole batch)
// the low 32 bits of the XSL-RR output of state s (pcg_output), through one funnel shift instead of a 64-bit rotate
__device__ __forceinline__ uint32_t de_output_low(U128 s)
{
    const uint64_t x = s.hi ^ s.lo;
    const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
    const uint32_t rot = (uint32_t)(s.hi >> 58);
    return (rot & 32u) ? __builtin_amdgcn_alignbit(xl, xh, rot & 31u) : __builtin_amdgcn_alignbit(xh, xl, rot & 31u);
}

__device__ __forceinline__ void de_scan_lane(const DePlanArgs& a, int t)
{
    const int n = a.n;
    const bool pow2 = (n & (n - 1)) == 0;
    const uint64_t threshold = a.threshold;
    // (the loads first: the bounds come from the kernarg segment, which is cold)
    const Affine128 f_lo = a.scan_lo[t & 255], f_hi = a.scan_hi[t >> 8];
    const U128 head_state = a.head->provisional[a.scan_parity];
    const int positions = a.scan_positions, run = a.scan_run;
    if ((long long)t * run >= positions) return;
    // a bad position: DifferentialEvolution.h:83-87 from this position on (rare: one position in n), listed with its E
    auto list_bad = [&](int p, U128 w, uint64_t v) {
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
        const int seg = p / a.seg_len;
        const uint32_t slot = atomicAdd(a.counts + (size_t)seg * kDeCountStride, 1u);
        if (slot < (uint32_t)a.bad_capacity)
        {
            DeBad b;
            b.p = (uint32_t)p;
            b.e = overrun ? (uint32_t)kDeOverrun : (uint32_t)(used - 2);
            a.bad[(size_t)seg * a.bad_capacity + slot] = b;
        }
        else
            atomicOr(&a.head->error, kDeErrCand);
    };
    // the state behind run * t draws, then position after position
    U128 s = apply(f_lo, apply(f_hi, head_state));
    s = pcg_step(s, a.inc);
    if (pow2)
    {
        // A power-of-two half (the threshold is 0: no draw is thrown away for its own sake): a position is bad when the
        // two draws there name the same walker, i.e. agree in the low bits of their outputs -- which one funnel shift
        // per draw gives (n <= 2^30), no 64-bit rotate, no 64-bit compares.
        const uint32_t mask = (uint32_t)(n - 1);
        uint32_t ind = de_output_low(s) & mask;
#pragma unroll 1
        for (int i = 0; i < run; ++i)
        {
            const int p = t * run + i;
            const U128 s_next = pcg_step(s, a.inc);
            const uint32_t ind_next = de_output_low(s_next) & mask;
            if (p < positions && ind == ind_next) list_bad(p, s, pcg_output(s));
            s = s_next;
            ind = ind_next;
        }
        return;
    }
    uint64_t raw = pcg_output(s);
    uint32_t ind = de_bounded(raw, n, pow2);
#pragma unroll 1
    for (int i = 0; i < run; ++i)
    {
        const int p = t * run + i;
        const U128 s_next = pcg_step(s, a.inc);
        const uint64_t nxt = pcg_output(s_next);
        const uint32_t ind_next = de_bounded(nxt, n, pow2);
        // bad: a draw below the threshold, or both draws naming the same walker -- an update starting here throws draws away
        if (p < positions && (raw < threshold || nxt < threshold || ind == ind_next)) list_bad(p, s, raw);
        s = s_next;
        raw = nxt;
        ind = ind_next;
    }
}

// minimum over the wavefront, in every lane (DPP inside the rows of 16, then the four rows through scalar registers)
__device__ __forceinline__ uint32_t de_wave_min(uint32_t v)
{
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    v = o < v ? o : v;
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, false);  // row_mirror
    v = o < v ? o : v;
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
    return a < b ? a : b;
}

// ---- resolve: the events of the batch, in stream order; the stream head behind the batch -------------------------------
// A bad position p = per * q + res is the start of update q - c / per when c draws have been thrown away and
// res == c mod per: the positions are kept by residue (a counting sort in LDS), so that an event looks at one residue's
// handful of positions only.  One workgroup of `threads` lanes sorts, its first wavefront walks.
// lds: capacity + 2 * per + 2 + kDeSegments uint32.
__host__ __device__ inline size_t de_resolve_lds_bytes(int capacity, int per) { return sizeof(uint32_t) * ((size_t)capacity + 2 * (size_t)per + 2 + kDeSegments); }

__device__ __forceinline__ void de_resolve_block(const DePlanArgs& a, uint32_t* lds, int capacity, int threads)
{
    const uint32_t per = (uint32_t)a.dims + 3u;
    uint32_t* const sh_key = lds;                  // [capacity]  q << 8 | E, residue after residue
    uint32_t* const sh_start = sh_key + capacity;  // [per + 1]   where each residue's keys begin
    uint32_t* const sh_fill = sh_start + per + 1;  // [per]
    uint32_t* const sh_off = sh_fill + per;        // [kDeSegments + 1]
    static_assert(kDeSegments == 64, "one list per lane of the first wavefront");
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const DeHead head = *a.head;
    const uint32_t behind = head.last_shift;
    uint32_t err = 0;

    if (wave == 0)
    {
        // the lists end to end: an inclusive scan of their lengths over the wavefront
        const uint32_t listed = a.resolve_counts[(size_t)lane * kDeCountStride];
        uint32_t sum = listed < (uint32_t)a.bad_capacity ? listed : (uint32_t)a.bad_capacity;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1)
        {
            const uint32_t up = __shfl_up(sum, off);
            if (lane >= off) sum += up;
        }
        sh_off[lane + 1] = sum;
        if (lane == 0) sh_off[0] = 0;
        a.resolve_counts[(size_t)lane * kDeCountStride] = 0;  // for the scan that fills these lists next
    }
    for (uint32_t r = (uint32_t)tid; r < per; r += (uint32_t)threads) sh_fill[r] = 0;
    __syncthreads();
    int total = (int)sh_off[kDeSegments];
    if (total > capacity)
    {
        err |= kDeErrCand;  // (positions beyond what the LDS holds are not looked at: a failed run)
        total = capacity;
    }
    auto entry = [&](int idx) -> DeBad {
        int lo = 0, hi = kDeSegments - 1;  // the list idx falls into: the last one whose offset is <= idx
        while (lo < hi)
        {
            const int mid = (lo + hi + 1) >> 1;
            if (sh_off[mid] <= (uint32_t)idx)
                lo = mid;
            else
                hi = mid - 1;
        }
        // the lists were made from the batch's provisional start: the true one lies `behind` draws further on
        DeBad b = a.resolve_bad[(size_t)lo * a.bad_capacity + ((uint32_t)idx - sh_off[lo])];
        b.p = (b.p >= behind && b.p - behind < (uint32_t)a.positions) ? b.p - behind : ~0u;  // (~0: not a position of this batch)
        return b;
    };
    // Eight positions per lane at a time (their loads leave together), kept in registers between the two passes of the
    // counting sort when that is all there is (the usual case).
    DeBad held[8];
    const bool once = total <= 8 * threads;
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (tid + u * threads < total) held[u] = entry(tid + u * threads);
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (tid + u * threads < total && held[u].p != ~0u) atomicAdd(&sh_fill[held[u].p % per], 1u);
    for (int idx = tid + 8 * threads; idx < total; idx += threads)
    {
        const DeBad b = entry(idx);
        if (b.p != ~0u) atomicAdd(&sh_fill[b.p % per], 1u);
    }
    __syncthreads();
    if (wave == 0)
    {
        uint32_t carry = 0;
        for (uint32_t r0 = 0; r0 < per; r0 += 64)
        {
            const uint32_t r = r0 + (uint32_t)lane;
            const uint32_t mine = r < per ? sh_fill[r] : 0u;
            uint32_t sum = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1)
            {
                const uint32_t up = __shfl_up(sum, off);
                if (lane >= off) sum += up;
            }
            if (r < per)
            {
                sh_start[r] = carry + sum - mine;
                sh_fill[r] = 0;
            }
            carry += (uint32_t)__builtin_amdgcn_readlane((int)sum, 63);
        }
        if (lane == 0) sh_start[per] = carry;
    }
    __syncthreads();
    auto place = [&](const DeBad& b) {
        if (b.p == ~0u) return;
        const uint32_t q = b.p / per, res = b.p - q * per;
        sh_key[sh_start[res] + atomicAdd(&sh_fill[res], 1u)] = (q << 8) | (b.e & 0xFFu);
    };
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (tid + u * threads < total) place(held[u]);
    if (!once)
        for (int idx = tid + 8 * threads; idx < total; idx += threads) place(entry(idx));
    __syncthreads();
    if (wave != 0) return;

    // The walk, by one wavefront: with c draws thrown away so far, update m starts at per * m + c; the next event is the
    // first bad position that IS the start of an update behind the last event's.  One minimum over the wavefront per
    // event (key: update index, then E; the updates of a batch number fewer than 2^23).
    DeBatch* const out = a.batch;
    uint32_t c = 0, cq = 0, cr = 0, events = 0;  // c = per * cq + cr
    int m_last = -1;  // (a position inside the draws of the update whose start was the last event is no start)
    while (true)
    {
        uint32_t best = ~0u;
        const int begin = (int)sh_start[cr], end = (int)sh_start[cr + 1];
        for (int i = begin + lane; i < end; i += 64)
        {
            const uint32_t key = sh_key[i];
            const int m = (int)(key >> 8) - (int)cq;
            if (m > m_last && m < a.updates)
            {
                const uint32_t mine = ((uint32_t)m << 8) | (key & 0xFFu);
                best = mine < best ? mine : best;
            }
        }
        best = de_wave_min(best);
        if (best == ~0u) break;
        uint32_t own = best & 0xFFu;
        const uint32_t m = best >> 8;
        if (own == (uint32_t)kDeOverrun)
        {
            err |= kDeErrWindow;
            own = 0;
        }
        if (c + own > (uint32_t)kDeShiftMax)
        {
            err |= kDeErrShift;
            own = (uint32_t)kDeShiftMax - c;
        }
        c += own;
        cr += own;
        while (cr >= per)
        {
            cr -= per;
            ++cq;
        }
        if (events < (uint32_t)kDeMaxEvents)
        {
            if (lane == 0)
            {
                out->plan[events].m = m;
                out->plan[events].shift_after = c;
            }
            ++events;
        }
        else
            err |= kDeErrCand;
        m_last = (int)m;
    }
    if (lane == 0)
    {
        out->base = head.state;
        out->extra_base = head.extra_total;
        out->events = events;
        // hand the stream on: where the next batch begins, and where the one behind it is scanned from
        const U128 next = apply(a.jump_small[c], apply(a.batch_jump, head.state));
        a.head->state = next;
        a.head->extra_total = head.extra_total + (unsigned long long)c;
        a.head->last_shift = c;
        a.head->provisional[a.scan_parity ^ 1] = apply(a.batch_jump, next);  // (the batch behind the one scanned beside this resolve)
        if (err) atomicOr(&a.head->error, err);
    }
}

// ---- records: one lane per update of the batch --------------------------------------------------------------------------
// One workgroup of `threads` lanes makes the records of updates [first, first + threads) of the batch.
// sh_plan: LDS for the batch's events (kDeMaxEvents entries).
template <class T>
__device__ __forceinline__ void de_records_block(const DePlanArgs& a, const DeBatch* batch, DeRec<T>* recs, DePlan* sh_plan, int first, int threads)
{
    const int n = a.n, dims = a.dims;
    const int m = first + (int)threadIdx.x;
    const int mm = m < a.updates ? m : a.updates - 1;
    // first round trip: the batch record's head and its first events (one per lane, whether they exist or not: waiting
    // for the count first would be a round trip of its own), this update's two table entries, the jump to the accept draw
    const DePlan early = batch->plan[threadIdx.x < (unsigned)kDeMaxEvents ? threadIdx.x : 0];
    const Affine128 j_hi = a.jump_hi[mm >> 8], j_lo = a.jump_lo[mm & 255], j_exp = a.jump_small[dims];
    const int plan_count = (int)batch->events;
    const U128 base = batch->base;
    if ((int)threadIdx.x < plan_count) sh_plan[threadIdx.x] = early;
    for (int j = (int)threadIdx.x + threads; j < plan_count; j += threads) sh_plan[j] = batch->plan[j];
    __syncthreads();
    if (m >= a.updates) return;
    // this update's place in the stream: the last event in front of it says how late it starts
    int lo = 0;  // first entry with m' >= m
    if (plan_count <= 64)
    {
        // (the usual case: a count over all events -- independent LDS reads -- instead of a chain of dependent ones)
        for (int j = 0; j < plan_count; ++j) lo += (int)sh_plan[j].m < m ? 1 : 0;
    }
    else
    {
        int hi = plan_count;
        while (lo < hi)
        {
            const int mid = (lo + hi) >> 1;
            if ((int)sh_plan[mid].m < m)
                lo = mid + 1;
            else
                hi = mid;
        }
    }
    const int shift = lo > 0 ? (int)sh_plan[lo - 1].shift_after : 0;
    const Affine128 j_shift = a.jump_small[shift];  // (second round trip, beside the two multiplications in front of it)
    U128 s = apply(j_shift, apply(j_lo, apply(j_hi, base)));
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
    DeRec<T> out;
    out.s = s;
    out.neg_exp = dev_log((T)1 - canonical(pcg_output(se), T()));  // -(-log(1 - u)/1)
    out.ind1 = ind1;
    out.ind2 = ind2;
    recs[m] = out;
}

// ---- the launches ------------------------------------------------------------------------------------------------------------
// In front of batch b (all three parts optional, for priming): workgroup 0 resolves batch b + 1 into a.batch; the next
// record_blocks workgroups make the records of batch b from rec_batch; the rest scan batch b + 2.
template <class T>
__global__ void __launch_bounds__(kDePlanThreads) de_boundary_kernel(const DePlanArgs a, int resolve, int capacity, int record_blocks, const DeBatch* rec_batch, DeRec<T>* recs)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int block = (int)blockIdx.x;
    if (resolve != 0)
    {
        if (block == 0)
        {
            de_resolve_block(a, reinterpret_cast<uint32_t*>(smem), capacity, kDePlanThreads);
            return;
        }
        --block;
    }
    if (block < record_blocks)
    {
        de_records_block<T>(a, rec_batch, recs, reinterpret_cast<DePlan*>(smem), block * kDePlanThreads, kDePlanThreads);
        return;
    }
    de_scan_lane(a, (block - record_blocks) * kDePlanThreads + (int)threadIdx.x);
}

// ---- behind a replay of `steps` ensemble steps: accepted proposals per step, then the run record moves on ---------------
__global__ void __launch_bounds__(256) de_accepted_kernel(const DeRunInfo* run, int partial_waves)
{
    __shared__ uint32_t sh_sum[256 / 64];
    uint32_t* const accepted = run->accepted;
    if (accepted == nullptr) return;
    const uint32_t* partials = reinterpret_cast<const uint32_t*>(run + 1) + (size_t)blockIdx.x * 2 * (size_t)partial_waves;
    uint32_t sum = 0;
    for (int j = (int)threadIdx.x; j < 2 * partial_waves; j += 256) sum += partials[j];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63) == 0) sh_sum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) accepted[run->step0 + (long long)blockIdx.x] = sh_sum[0] + sh_sum[1] + sh_sum[2] + sh_sum[3];
}

__global__ void de_advance_kernel(DeRunInfo* run, int steps)
{
    const uint32_t since = run->phase0 + (uint32_t)steps;
    const uint32_t whole = since / run->interval;
    run->phase0 = since - whole * run->interval;
    run->slot0 += (long long)whole;
    run->step0 += (long long)steps;
}

}  // namespace mcmcpp

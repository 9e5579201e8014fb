// diffevo_plan.hpp -- the planning launches of Mover::DifferentialEvolution (scheme: diffevo_kernel.hpp): scan, resolve,
// records, and the two small launches behind a replay.  They depend on the random stream alone, not on the calculator:
// compiled once, in diffevo.hip.
#pragma once

#include "diffevo_kernel.hpp"

namespace mcmcpp
{
__device__ __forceinline__ uint32_t de_bounded(uint64_t v, int n, bool pow2) { return pow2 ? (uint32_t)(v & (uint64_t)(n - 1)) : (uint32_t)(v % (uint64_t)n); }

// ---- scan: every stream position of the batch, drawn once; the bad ones listed ------------------------------------------
__global__ void __launch_bounds__(kDePlanThreads) de_scan_kernel(const DePlanArgs a)
{
    const int n = a.n;
    const bool pow2 = (n & (n - 1)) == 0;
    const uint64_t threshold = a.threshold;
    const int positions = a.positions, run = a.scan_run;
    const int t = (int)blockIdx.x * kDePlanThreads + (int)threadIdx.x;
    if ((long long)t * run >= positions) return;
    // the state behind run * t draws, then position after position
    U128 s = apply(a.scan_lo[t & 255], apply(a.scan_hi[t >> 8], a.head->state));
    s = pcg_step(s, a.inc);
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
        const bool is_bad = p < positions && (raw < threshold || nxt < threshold || ind == ind_next);
        if (is_bad)
        {
            // DifferentialEvolution.h:83-87 from this position on (rare: one position in n)
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
        }
        s = s_next;
        raw = nxt;
        ind = ind_next;
    }
}

// ---- resolve: the events of the batch, in stream order; the stream head behind the batch -------------------------------
__global__ void __launch_bounds__(kDePlanThreads) de_resolve_kernel(const DePlanArgs a)
{
    // a bad position p = per * q + res is the start of update q - c / per when c draws have been thrown away and
    // res == c mod per
    __shared__ uint32_t sh_q[kDeMaxBad];
    __shared__ uint16_t sh_res[kDeMaxBad];
    __shared__ uint8_t sh_e[kDeMaxBad];
    __shared__ uint32_t sh_off[kDeSegments + 1];
    __shared__ unsigned long long sh_min[kDePlanThreads / 64];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per = (uint32_t)a.dims + 3u;
    const DeHead head = *a.head;
    uint32_t err = 0;

    if (tid < kDeSegments)
    {
        const uint32_t listed = a.counts[(size_t)tid * kDeCountStride];
        sh_off[tid + 1] = listed < (uint32_t)a.bad_capacity ? listed : (uint32_t)a.bad_capacity;
    }
    __syncthreads();
    if (tid == 0)
    {
        uint32_t sum = 0;
        sh_off[0] = 0;
        for (int g = 0; g < kDeSegments; ++g)
        {
            sum += sh_off[g + 1];
            sh_off[g + 1] = sum;
        }
    }
    __syncthreads();
    int total = (int)sh_off[kDeSegments];
    if (total > kDeMaxBad)
    {
        err |= kDeErrCand;
        total = kDeMaxBad;
    }
    for (int idx = tid; idx < total; idx += kDePlanThreads)
    {
        int lo = 0, hi = kDeSegments - 1;  // the list idx falls into: the last one whose offset is <= idx
        while (lo < hi)
        {
            const int mid = (lo + hi + 1) >> 1;
            if (sh_off[mid] <= (uint32_t)idx)
                lo = mid;
            else
                hi = mid - 1;
        }
        const DeBad b = a.bad[(size_t)lo * a.bad_capacity + ((uint32_t)idx - sh_off[lo])];
        const uint32_t q = b.p / per;
        sh_q[idx] = q;
        sh_res[idx] = (uint16_t)(b.p - q * per);
        sh_e[idx] = (uint8_t)b.e;
    }
    __syncthreads();

    // The walk, in stream order: with c draws thrown away so far, update m starts at per * m + c; the next event is the
    // first bad position that IS the start of an update behind the last event's.  One minimum per event.
    uint32_t c = 0, events = 0;
    long long m_last = -1;  // (a position inside the draws of the update whose start was the last event is no start)
    while (true)
    {
        const uint32_t cq = c / per, cr = c - cq * per;
        unsigned long long best = ~0ULL;
        for (int idx = tid; idx < total; idx += kDePlanThreads)
        {
            if ((uint32_t)sh_res[idx] != cr) continue;
            const long long m = (long long)sh_q[idx] - (long long)cq;
            if (m > m_last && m < (long long)a.updates)
            {
                const unsigned long long key = ((unsigned long long)m << 8) | (unsigned long long)sh_e[idx];
                best = key < best ? key : best;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
        {
            const unsigned long long other = __shfl_xor(best, off);
            best = other < best ? other : best;
        }
        __syncthreads();  // (the previous round's sh_min has been read by everyone)
        if (lane == 0) sh_min[wave] = best;
        __syncthreads();
        best = sh_min[0];
#pragma unroll
        for (int wv = 1; wv < kDePlanThreads / 64; ++wv) best = sh_min[wv] < best ? sh_min[wv] : best;
        if (best == ~0ULL) break;
        uint32_t own = (uint32_t)(best & 0xFFu);
        const uint32_t m = (uint32_t)(best >> 8);
        if (own == (uint32_t)kDeOverrun)
        {
            err |= kDeErrWindow;
            own = 0;
        }
        c += own;
        if (c > (uint32_t)kDeShiftMax)
        {
            err |= kDeErrShift;
            c = (uint32_t)kDeShiftMax;
        }
        if (events < (uint32_t)kDeMaxEvents)
        {
            if (tid == 0)
            {
                a.batch->plan[events].m = m;
                a.batch->plan[events].shift_after = c;
            }
            ++events;
        }
        else
            err |= kDeErrCand;
        m_last = (long long)m;
    }
    if (tid == 0)
    {
        a.batch->base = head.state;
        a.batch->events = events;
        // hand the stream on
        a.head->state = apply(a.jump_small[c], apply(a.batch_jump, head.state));
        a.head->extra_total = head.extra_total + (unsigned long long)c;
        if (err) atomicOr(&a.head->error, err);
    }
    if (tid < kDeSegments) a.counts[(size_t)tid * kDeCountStride] = 0;  // for the next batch's scan
}

// ---- records: one lane per update of the batch --------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(kDePlanThreads) de_records_kernel(const DePlanArgs a, DeRec<T>* recs)
{
    __shared__ DePlan sh_plan[kDeMaxEvents];
    const int n = a.n, dims = a.dims;
    const int plan_count = (int)a.batch->events;
    const U128 base = a.batch->base;
    for (int j = (int)threadIdx.x; j < plan_count; j += kDePlanThreads) sh_plan[j] = a.batch->plan[j];
    __syncthreads();
    const int m = (int)blockIdx.x * kDePlanThreads + (int)threadIdx.x;
    if (m >= a.updates) return;
    // this update's place in the stream: the last event in front of it says how late it starts
    int lo = 0, hi = plan_count;  // first entry with m' >= m
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if ((int)sh_plan[mid].m < m)
            lo = mid + 1;
        else
            hi = mid;
    }
    const int shift = lo > 0 ? (int)sh_plan[lo - 1].shift_after : 0;
    U128 s = apply(a.jump_small[shift], apply(a.jump_lo[m & 255], apply(a.jump_hi[m >> 8], base)));
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
    DeRec<T> out;
    out.s = s;
    out.neg_exp = dev_log((T)1 - canonical(pcg_output(se), T()));  // -(-log(1 - u)/1)
    out.ind1 = ind1;
    out.ind2 = ind2;
    recs[m] = out;
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

// full_step_kernel.hpp -- one launch per ENSEMBLE step (both halves of EnsembleSampler::performStep,
// MCMCpp/EnsembleSampler.h:345-354) for ensembles small enough that the launch boundary and memory latency, not
// HBM bandwidth, are what a half-step launch costs.
//
// The black half needs the red half's results, which is why the reference (and stretch_half_step_kernel) put a
// barrier between them.  But one black update reads exactly ONE red walker -- its partner j -- and red j's
// update reads exactly one black walker's OLD row.  So a black walker's lane group first repeats its partner's
// red update (same draw record, same rows, same arithmetic, hence the same bits as the red owner's own result)
// and then performs its own update against it: no barrier, one launch, three calculator evaluations per
// red/black pair instead of two.  That only works if nobody overwrites a row another group still needs, so the
// positions and log-posteriors ping-pong between two buffers: a step reads buffer `in` (all rows as they stood
// before the step) and writes buffer `out`.  Writes are what such a launch pays most for (1.2 of 7 us at 16384 x 32
// when every row is rewritten), so a walker that stays put is not copied if `out` already holds its row, i.e. if it
// did not move in the previous step either: the top bit of its accepted counter remembers "moved in the last step"
// (kRowMovedBit), which cuts the row traffic to p + (1-p)p of the rows (41 % at an acceptance rate p of 0.23).
//
// A wavefront owns WPP red walkers and the WPP black walkers of the same index.  Its dependent chain is two
// memory round trips and two calculator evaluations:
//   trip 1   the draw records, rows, log-posteriors and counters of its own red and black walkers
//   trip 2   everything the records point to: the red walker's partner row; the black walker's red partner x
//            (record, row, log-posterior) AND x's own partner row -- the black record carries that index
//            (DrawRec::partner2: the draws do not depend on the walkers, so whoever makes the black record can
//            repeat x's partner draw)
//   then     x's update repeated (together with the red owner's own), the black update against its result.
// Every scalar miss of an updating wavefront (control record, run record, the lines of the launch description) is
// taken in ONE batch behind the second trip's loads (load_records_and_warm_args).  Draw records one step ahead
// (four extra wavefronts per workgroup, two per colour), counters, chain stores: as in stretch_kernel.hpp; stored
// steps reach pinned host memory through the launches themselves (trickle_stored_step).
#pragma once

#include "stretch_kernel.hpp"

namespace mcmcpp
{

constexpr uint32_t kRowMovedBit = 0x80000000u;  // in n_accept[w]: the walker's row in the other position buffer is out of date
#ifndef MCMCPP_FULL_DRAW_WAVES
#define MCMCPP_FULL_DRAW_WAVES 4
#endif
constexpr int kFullDrawWaves = MCMCPP_FULL_DRAW_WAVES;  // extra wavefronts of a full-step workgroup: 4 = next red draws (2), next black draws (2); 2 = one per colour

// Hands the random stream and the step counters to the next full-step launch (one lane of the whole grid).
template <class T>
__device__ __forceinline__ void hand_over_full(const HalfStepArgs<T>& a, const StepCtl& ctl, const RunInfo& run, StepCtl* ctl_out)
{
    StepCtl nx = ctl;
    nx.state = apply(a.half_jump, apply(a.half_jump, ctl.state));
    nx.state2 = apply(a.half_jump, apply(a.half_jump, ctl.state2));
    nx.half_step = ctl.half_step + 2;
    const bool saved = ctl.save_phase + 1u == (uint32_t)run.interval;
    nx.step_in_run = ctl.step_in_run + 1;
    nx.save_phase = saved ? 0u : ctl.save_phase + 1u;
    nx.chain_slot = ctl.chain_slot + (saved ? 1 : 0);
    nx.partial_slot = (ctl.partial_slot + 1u == (uint32_t)a.partial_slots) ? 0u : ctl.partial_slot + 1u;
    *ctl_out = nx;  // (plain stores: written through, this and the partial counts below cost 0.2 us per launch)
}

// hot_bits of the full-step kernels: HotBits::pack(...) | pos_parity << 26
__host__ __device__ inline uint32_t full_step_bits(uint32_t half_bits, int pos_parity) { return half_bits | ((uint32_t)pos_parity << 26); }

// the workgroup's extra wavefronts: the draws of the NEXT ensemble step of every walker this workgroup updates
// (`wpb` of each colour, starting at walker blockIdx.x * wpb); extra wavefronts 0, 1 make the red records of the first
// and second half of those walkers, 2 and 3 the black ones
template <class T>
__device__ __forceinline__ void full_step_draw_wave(const HalfStepArgs<T>& a, const JumpTables& tab, const StepCtl* ctl_ptr, const RunInfo* run_ptr,
                                                    bool block_barrier, DrawRec<T>* dn_red, int n, int sh_begin, int sh_count, int wpb, int which, int lane,
                                                    int run_behind_ctl = -1, int ctl_chain = 0)
{
    const int black = kFullDrawWaves == 4 ? which >> 1 : which, half = kFullDrawWaves == 4 ? (which & 1) : 0;
    const int h0 = kFullDrawWaves == 4 ? (wpb + 1) / 2 : wpb;  // walkers of the first half
    DrawRec<T>* const dst = black ? dn_red + n : dn_red;
    // (the first of them also forwards this launch's slice of the last stored step: trickle_stored_step)
    draw_wave_body<T, 1>(a, tab, ctl_ptr, block_barrier, dst, dst, 1, sh_begin, sh_count, blockIdx.x * wpb + half * h0, half ? wpb - h0 : h0, lane, black != 0,
                         which == 0 ? run_ptr : nullptr, which == 0 ? run_behind_ctl : -1, ctl_chain);
}

// the one extra wavefront of a launch whose draw records were made ahead of it: the stored steps' forwarding only
template <class T>
__device__ __forceinline__ void full_step_trickle_wave(const StepCtl* ctl_ptr, int run_behind_ctl, int ctl_chain, int lane)
{
    StepCtl ctl;
    RunInfo run;
    const unsigned ctl_off = (unsigned)ctl_chain * (unsigned)kCtlChainStride;
    load_records_and_warm_args<T>(ctl_ptr, ctl_off, ctl_ptr, ctl_off + (unsigned)kRunBehindCtlBytes - (unsigned)run_behind_ctl * (unsigned)sizeof(StepCtl), ctl, run);
    trickle_stored_step(run, ctl, lane);
}

// Both full-step kernels update walkers [sh_begin, sh_begin + sh_count) of EACH colour (the whole halves, or the slice
// of one rank of a split ensemble: a black walker's group repeats the update of its red partner wherever that one
// lives, so the ranks exchange rows once per ensemble step instead of once per half-step).  The bounds travel in the
// 16 preloaded dwords; what they displaced is derived: the accepted counters lie right behind the two log-posterior
// buffers ([2][W] elements, then [W] counters: one allocation), and the run record kRunBehindCtlBytes behind the
// first control record.
// (kRunBehindCtlBytes, stretch_kernel.hpp)

template <class T, class Calc, int EPL, int LPW, bool MC = false>
__global__ void __launch_bounds__(64 * (kWavesPerBlock + kFullDrawWaves))
stretch_full_step_kernel(DrawRec<T>* hot_draws, T* hot_pos_a, T* hot_pos_b, T* hot_logp_a, const RunInfo* hot_run, int hot_sh_begin, int hot_sh_count, int hot_n,
                         uint32_t hot_bits, const StepCtl* hot_ctl_in, const HalfStepArgs<T> rest)
{
    const HalfStepArgs<T>& a = rest;
    constexpr int WPP = 64 / LPW;  // walkers of each colour per wavefront
    const int h_n = hot_n;
    // chain blockIdx.y of (hot_bits >> 28) + 1 (ChainGeometry): every per-chain array at its fixed stride
    const int chain = MC ? (int)blockIdx.y : 0, chains = MC ? (int)(hot_bits >> 28) + 1 : 1;  // (MC: see stretch_half_step_kernel)
    const void* const draws_chain0 = hot_draws;
    if (MC && chain != 0)  // (a branch on purpose: chain 0 -- every single-ensemble launch -- skips the 64-bit products)
    {
        hot_draws += (size_t)chain * 4 * (size_t)h_n;
        hot_pos_a += (size_t)chain * 2 * (size_t)h_n * (size_t)(hot_bits & 0xFFFu);
        hot_pos_b += (size_t)chain * 2 * (size_t)h_n * (size_t)(hot_bits & 0xFFFu);
        hot_logp_a = reinterpret_cast<T*>(reinterpret_cast<char*>(hot_logp_a) + (size_t)chain * logp_chain_stride_bytes<T>(h_n));
        // (the control record's address is not among the preloaded arguments: its chain offset travels as a load offset)
        hot_run = reinterpret_cast<const RunInfo*>(reinterpret_cast<const char*>(hot_run) + (size_t)chain * kCtlChainStride);
    }
    const int h_dims = (int)(hot_bits & 0xFFFu);
    const bool vec_ok = ((hot_bits >> 21) & 1u) != 0;
    const int h_use_ctl_save = (int)((hot_bits >> 23) & 1u);
    const int h_parity = (int)((hot_bits >> 24) & 1u);
    const bool h_draw_wave = ((hot_bits >> 25) & 1u) != 0;
    const bool h_flip = ((hot_bits >> 26) & 1u) != 0;
    const DrawRec<T>* const dr_red = hot_draws + (size_t)h_parity * 2 * (size_t)h_n;
    const DrawRec<T>* const dr_blk = dr_red + h_n;
    DrawRec<T>* const dn_red = hot_draws + (size_t)(1 - h_parity) * 2 * (size_t)h_n;
    DrawRec<T>* const dn_blk = dn_red + h_n;
    const T* const pin = h_flip ? hot_pos_b : hot_pos_a;
    T* const pout = h_flip ? hot_pos_a : hot_pos_b;
    T* const hot_logp_b = hot_logp_a + 2 * (size_t)h_n;  // (the two log-posterior buffers are one allocation: [2][W])
    const T* const lin = h_flip ? hot_logp_b : hot_logp_a;
    T* const lout = h_flip ? hot_logp_a : hot_logp_b;
    uint32_t* const h_n_accept = reinterpret_cast<uint32_t*>(hot_logp_a + 4 * (size_t)h_n);
    const int sh_begin = hot_sh_begin, sh_count = hot_sh_count;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* sh_stage = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::stage_offset());
    T* sh_block = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::block_offset());

#ifdef MCMCPP_STAMPS
    unsigned long long stamp_val[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    stamp_val[6] = __builtin_amdgcn_s_memrealtime();
#endif
    MCMCPP_STAMP(0);
    MCMCPP_STAMP_BLOCK(0);
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    if (wib >= kWavesPerBlock)
    {
        full_step_draw_wave<T>(a, jump_tables_behind(draws_chain0, h_n, ((hot_bits >> 27) & 1u) != 0, chains), hot_ctl_in, hot_run, Calc::block_scratch_elems(h_dims) != 0,
                               dn_red, h_n, sh_begin, sh_count, kWavesPerBlock * WPP, wib - kWavesPerBlock, lane, -1, chain);
        return;
    }
    const int wave = blockIdx.x * kWavesPerBlock + wib;
    const int first = wave * WPP;  // (relative to the shard)
    const bool wave_active = first < sh_count;
    const int sub = lane & (LPW - 1);
    const int grp = lane / LPW;
    const int i0 = sub * EPL;
    const int li = first + grp;
    const bool active = li < sh_count;
    const int ir = sh_begin + (active ? li : 0);  // this group's red walker, and (n + ir) its black walker

    GroupCtx<T, EPL, LPW> ctx;
    ctx.sub = sub;
    ctx.dims = h_dims;
    ctx.lane = lane;
    ctx.stage = Calc::kNeedsStage ? &sh_stage[wib * 64 * EPL] : nullptr;
    ctx.vec_ok = vec_ok;

    // ---- first round trip: records, rows, log-posteriors and counters of the group's own red and black walker ----
    const DrawRec<T> rec_r = dr_red[ir];
    const DrawRec<T> rec_b = dr_blk[ir];
    T own_r[EPL], own_b[EPL];
    if (vec_ok)  // (one branch around both rows: see load_slice)
    {
        load_slice_as<T, EPL, true>(pin + (size_t)ir * h_dims, i0, h_dims, active, own_r);
        load_slice_as<T, EPL, true>(pin + (size_t)(h_n + ir) * h_dims, i0, h_dims, active, own_b);
    }
    else
    {
        load_slice_as<T, EPL, false>(pin + (size_t)ir * h_dims, i0, h_dims, active, own_r);
        load_slice_as<T, EPL, false>(pin + (size_t)(h_n + ir) * h_dims, i0, h_dims, active, own_b);
    }
    const T lp_r = lin[ir];
    const T lp_b = lin[h_n + ir];
    uint32_t nacc_r = h_n_accept[ir];
    uint32_t nacc_b = h_n_accept[h_n + ir];
    typename Calc::Prefetch calc_pf;
    Calc::block_prefetch(calc_pf, a.calc_params, h_dims, vec_ok, (int)threadIdx.x, 64 * kWavesPerBlock);

    // ---- second round trip: everything the two records point to ----
    T par_r[EPL], own_x[EPL], par_x[EPL];
    const int jx = (int)rec_b.partner;
    const DrawRec<T> rec_x = dr_red[jx];
    const T lp_x = lin[jx];
    if (vec_ok)  // (one branch around the three rows: see load_slice)
    {
        load_slice_as<T, EPL, true>(pin + (size_t)(h_n + (int)rec_r.partner) * h_dims, i0, h_dims, active, par_r);
        load_slice_as<T, EPL, true>(pin + (size_t)jx * h_dims, i0, h_dims, active, own_x);
        load_slice_as<T, EPL, true>(pin + (size_t)(h_n + (int)rec_b.partner2) * h_dims, i0, h_dims, active, par_x);
    }
    else
    {
        load_slice_as<T, EPL, false>(pin + (size_t)(h_n + (int)rec_r.partner) * h_dims, i0, h_dims, active, par_r);
        load_slice_as<T, EPL, false>(pin + (size_t)jx * h_dims, i0, h_dims, active, own_x);
        load_slice_as<T, EPL, false>(pin + (size_t)(h_n + (int)rec_b.partner2) * h_dims, i0, h_dims, active, par_x);
    }
    MCMCPP_STAMP(1);  // first round trip landed, second issued
    // Every scalar miss of this wavefront is taken here, in one batch whose wait overlaps the second round trip: the
    // control and run records (preloaded pointers) and all lines of the launch description (cold misses of the order of
    // a microsecond each; issued one by one where first needed they delayed the calculator by about that much).
    StepCtl ctl;  // wave-uniform
    RunInfo run;
    load_records_and_warm_args<T>(hot_ctl_in, (unsigned)chain * (unsigned)kCtlChainStride, hot_run, 0u, ctl, run);
    const StepCtl* const ctl_mine = reinterpret_cast<const StepCtl*>(reinterpret_cast<const char*>(hot_ctl_in) + (size_t)chain * kCtlChainStride);

    // ---- in its shadow: the calculator's tables, the hand-over to the next launch ----
    const bool has_block_scratch = Calc::block_scratch_elems(h_dims) != 0;
    Calc::block_commit(calc_pf, sh_block, a.calc_params, h_dims, vec_ok, (int)threadIdx.x, 64 * kWavesPerBlock);
    if (has_block_scratch) __syncthreads();
    ctx.block_scratch = has_block_scratch ? sh_block : nullptr;
    typename Calc::template Regs<EPL, LPW> cregs;
    Calc::template preload<EPL, LPW>(ctx, a.calc_params, cregs);
    if (blockIdx.x == 0 && threadIdx.x == 0) hand_over_full<T>(a, ctl, run, const_cast<StepCtl*>(ctl_mine) + (h_flip ? -1 : 1));
    if (!wave_active) return;

    long long save_slot = -1;
    if (h_use_ctl_save && run.chain != nullptr && ctl.save_phase + 1u == (uint32_t)run.interval) save_slot = (run.chain_slot_base + ctl.chain_slot) & run.slot_mask;

    if (!h_draw_wave)
    {
        // no extra wavefronts (many walkers per wavefront): the stored-step forwarding and the next step's draws of this
        // wavefront's walkers happen here
        if (wib == 0) trickle_stored_step(run, ctl, lane);
        const bool direct = a.task_jump != nullptr;
        const U128 base_b = apply(a.half_jump, ctl.state2);
        for (int t = lane; t < 6 * WPP; t += 64)
        {
            const int c = t >= 3 * WPP ? 1 : 0;
            const int tt = t - 3 * WPP * c;
            const int slot = tt / 3, k = tt - 3 * slot;
            if (first + slot >= sh_count) continue;
            const int i = sh_begin + first + slot;
            Affine128 j_a, j_b;
            if (direct)
                j_a = a.task_jump[3 * i + k];
            else
            {
                j_a = a.jump_hi[i >> 8];
                j_b = a.jump_lo[i & 255];
            }
            compute_draw<T>(a, c ? base_b : ctl.state2, j_a, j_b, direct, k, (c ? dn_blk : dn_red) + i, c != 0, ctl.state2);
        }
    }
    MCMCPP_STAMP(2);

    // One StretchMove::updateWalker (StretchMove.h:100-123) in two parts: the proposal, then calculator + accept test.
    auto propose = [&](const T (&own)[EPL], const T (&par)[EPL], const DrawRec<T>& rec, T (&prop)[EPL]) {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
        {
            const T d = own[e] - par[e];
            const T zd = rec.z * d;
            prop[e] = par[e] + zd;
        }
    };
    // `fin` receives the walker's row after the update; returns accept, lp_out = its log-posterior afterwards.
    auto finish = [&](const T (&own)[EPL], const T (&prop)[EPL], const DrawRec<T>& rec, T lp_old, bool count_ties, T (&fin)[EPL], T& lp_out) -> bool {
        const T lp_new = Calc::template eval<EPL, LPW>(ctx, a.calc_params, cregs, prop);
        const T delta = rec.zs + lp_new - lp_old;
        const bool accept = rec.ln_u < delta;
        if (count_ties && active && sub == 0)
        {
            const T margin = dev_abs(rec.ln_u - delta);
            const T scale = dev_abs(rec.ln_u) + dev_abs(rec.zs) + dev_abs(lp_new) + dev_abs(lp_old);
            if (margin <= a.tie_eps * scale) count_near_tie(a.diag);
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) fin[e] = accept ? prop[e] : own[e];
        lp_out = accept ? lp_new : lp_old;
        return accept;
    };
    // Walker::jumpToNewPointSwap / stayAtCurrentPoint (Walker/Walker.h:162-179): `out` receives the row unless it holds it already
    auto commit = [&](int w, const T (&fin)[EPL], T lp_fin, bool accept, uint32_t nacc_old) {
        if (!active) return;
        if (accept || (nacc_old & kRowMovedBit) != 0u)
        {
            store_slice<T, EPL, true>(pout + (size_t)w * h_dims, i0, h_dims, vec_ok, fin);
            if (sub == 0)
            {
                store_through(lout + w, lp_fin);
                store_through(h_n_accept + w, accept ? ((nacc_old + 1u) | kRowMovedBit) : (nacc_old & ~kRowMovedBit));
            }
        }
        if (save_slot >= 0)
        {
            T* crow = reinterpret_cast<T*>(run.chain) + ((size_t)save_slot * (size_t)(2 * h_n) + (size_t)w) * h_dims;
            store_slice<T, EPL>(crow, i0, h_dims, vec_ok, fin);
        }
    };

    // ---- the black walker first (the longer chain): its red partner's update repeated, then its own against the result ----
    // All three proposals' inputs are consumed before the first (conditional) store: a wait for loads placed behind
    // such a store would also wait for the store.
    T prop_x[EPL], prop_r[EPL];
    propose(own_x, par_x, rec_x, prop_x);
    propose(own_r, par_r, rec_r, prop_r);
#pragma unroll
    for (int e = 0; e < EPL; ++e) asm volatile("" : "+v"(prop_x[e]), "+v"(prop_r[e]));
    T new_x[EPL], fin[EPL];
    T lp_x_new, lp_fin;
    finish(own_x, prop_x, rec_x, lp_x, false, new_x, lp_x_new);
    MCMCPP_STAMP(3);  // second round trip landed, partner's update repeated
    T prop_b[EPL];
    propose(own_b, new_x, rec_b, prop_b);
    // ---- the group's red walker: its row goes out now and drains while the black update computes ----
    const bool acc_r = finish(own_r, prop_r, rec_r, lp_r, true, fin, lp_fin) && active;
    commit(ir, fin, lp_fin, acc_r, nacc_r);
    MCMCPP_STAMP(4);  // red update done, its stores issued
    const bool acc_b = finish(own_b, prop_b, rec_b, lp_b, true, fin, lp_fin) && active;
    commit(h_n + ir, fin, lp_fin, acc_b, nacc_b);

    MCMCPP_STAMP(5);
    MCMCPP_STAMP_BLOCK(1);
#ifdef MCMCPP_STAMPS
    if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
    {
        stamp_val[7] = __builtin_amdgcn_s_memrealtime();
        for (int k = 0; k < 8; ++k) a.stamps[k] = stamp_val[k];
    }
#endif
    const unsigned acc_red = (unsigned)__popcll(__ballot(acc_r && sub == 0));
    const unsigned acc_blk = (unsigned)__popcll(__ballot(acc_b && sub == 0));
    if (a.partials != nullptr && run.accepted_per_step != nullptr && lane == 0)
    {
        uint32_t* p = a.partials + ((size_t)chain * (size_t)a.partial_slots + (size_t)ctl.partial_slot) * 2 * (size_t)a.partial_waves;
        p[wave] = acc_red;
        p[(size_t)a.partial_waves + wave] = acc_blk;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Matrix-core variant (dense D x D product, fp64, even D in 18..32; see stretch_half_step_mfma_kernel for the
// operand layout and why v_mfma_f64_16x16x4_f64 reproduces the host's fma chain).  A wavefront owns 8 red and 8
// black walkers (two passes of four 16-lane groups each).  The wavefront's share of P^T (zero-padded to 32 x 32 by
// the host) comes straight from memory into registers with the first round trip: no LDS copy, no workgroup
// barrier, so the draw wavefronts start at once.  First ONE full MFMA tile evaluates the 8 repeated partner
// updates (rows 0..7) together with the 8 red owners' updates (rows 8..15), then a half tile the 8 black updates.
// ---------------------------------------------------------------------------------------------------------
// 16 bytes of a walker row of the `out` buffer (written through: see store_through)
__device__ __forceinline__ void store_row_piece(double* p, double x0, double x1)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d v = {x0, x1};
    store_through16(p, v);
}
// (fp32: 8 bytes of a row)
__device__ __forceinline__ void store_row_piece(float* p, float x0, float x1)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = {x0, x1};
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// The same stores addressed as (uniform base, 32-bit byte offset): one register per address, no 64-bit arithmetic
__device__ __forceinline__ void store_row_piece_at(char* base, uint32_t off, double x0, double x1)
{
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d v = {x0, x1};
    asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void store_row_piece_at(char* base, uint32_t off, float x0, float x1)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = {x0, x1};
    asm volatile("global_store_dwordx2 %0, %1, %2 sc0 sc1" ::"v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void store_through_at(char* base, uint32_t off, uint32_t v) { asm volatile("global_store_dword %0, %1, %2 sc0 sc1" ::"v"(off), "v"(v), "s"(base) : "memory"); }
__device__ __forceinline__ void store_through_at(char* base, uint32_t off, float v) { asm volatile("global_store_dword %0, %1, %2 sc0 sc1" ::"v"(off), "v"(v), "s"(base) : "memory"); }
__device__ __forceinline__ void store_through_at(char* base, uint32_t off, double v) { asm volatile("global_store_dwordx2 %0, %1, %2 sc0 sc1" ::"v"(off), "v"(v), "s"(base) : "memory"); }

template <class T, class Calc, int EPL, int LPW, bool MC = false>
__global__ void __launch_bounds__(64 * (kWavesPerBlock + kFullDrawWaves))
stretch_full_step_mfma_kernel(DrawRec<T>* hot_draws, T* hot_pos_a, T* hot_pos_b, T* hot_logp_a, const T* hot_matrix, int hot_sh_begin, int hot_sh_count, int hot_n,
                              uint32_t hot_bits, const StepCtl* hot_ctl_in, const HalfStepArgs<T> rest)
{
    static_assert(EPL == 2 && LPW == 16, "matrix-core path: 16 lanes x 2 elements per walker, 16 < D <= 32");
    constexpr int NW = 8;  // walkers of each colour per wavefront
    // LDS: per updating wavefront one 16-row staging area of proposals and one for the second tile (8 rows in fp64, whose
    // walkers sit in rows 4q + g; 16 in fp32: rows 4g + q, see mc_row)
    constexpr int kSecondTileRows = sizeof(T) == 8 ? NW : 2 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* sh_x = reinterpret_cast<T*>(smem) + (threadIdx.x >> 6) * ((2 * NW + kSecondTileRows) * kMcXS);

    const HalfStepArgs<T>& a = rest;
#ifdef MCMCPP_STAMPS
    unsigned long long stamp_val[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    stamp_val[6] = __builtin_amdgcn_s_memrealtime();
#endif
    MCMCPP_STAMP(0);
    MCMCPP_STAMP_BLOCK(0);
    const int h_n = hot_n;
    const int h_dims = (int)(hot_bits & 0xFFFu);
    // chain blockIdx.y of (hot_bits >> 28) + 1 (ChainGeometry): every per-chain array at its fixed stride
    const int chain = MC ? (int)blockIdx.y : 0, chains = MC ? (int)(hot_bits >> 28) + 1 : 1;  // (MC: see stretch_half_step_kernel)
    const void* const draws_chain0 = hot_draws;
    if (MC && chain != 0)  // (a branch on purpose: chain 0 -- every single-ensemble launch -- skips the 64-bit products)
    {
        hot_draws += (size_t)chain * 4 * (size_t)h_n;
        hot_pos_a += (size_t)chain * 2 * (size_t)h_n * (size_t)h_dims;
        hot_pos_b += (size_t)chain * 2 * (size_t)h_n * (size_t)h_dims;
        hot_logp_a = reinterpret_cast<T*>(reinterpret_cast<char*>(hot_logp_a) + (size_t)chain * logp_chain_stride_bytes<T>(h_n));
    }
    const int h_use_ctl_save = (int)((hot_bits >> 23) & 1u);
    const int h_parity = (int)((hot_bits >> 24) & 1u);
    const bool h_flip = ((hot_bits >> 26) & 1u) != 0;
    const DrawRec<T>* const dr_red = hot_draws + (size_t)h_parity * 2 * (size_t)h_n;
    const DrawRec<T>* const dr_blk = dr_red + h_n;
    DrawRec<T>* const dn_red = hot_draws + (size_t)(1 - h_parity) * 2 * (size_t)h_n;
    const T* const pin = h_flip ? hot_pos_b : hot_pos_a;
    T* const pout = h_flip ? hot_pos_a : hot_pos_b;
    T* const hot_logp_b = hot_logp_a + 2 * (size_t)h_n;  // (the two log-posterior buffers are one allocation: [2][W])
    const T* const lin = h_flip ? hot_logp_b : hot_logp_a;
    T* const lout = h_flip ? hot_logp_a : hot_logp_b;
    // (log-posteriors [2][W] and accepted counters [W] are one allocation: the counters' address is derived)
    uint32_t* const h_n_accept = reinterpret_cast<uint32_t*>(hot_logp_a + 4 * (size_t)h_n);
    const int sh_begin = hot_sh_begin, sh_count = hot_sh_count;

    const int lane = threadIdx.x & 63;
    if ((threadIdx.x >> 6) >= kWavesPerBlock)
    {
        if ((hot_bits >> 20) & 1u)
        {
            // HalfStepArgs::draw_wave == 2: this step's records were made ahead of the launches (fill_draws_batch_kernel);
            // the one extra wavefront forwards this launch's slice of the last stored step and that is all
            if ((int)(threadIdx.x >> 6) == kWavesPerBlock) full_step_trickle_wave<T>(hot_ctl_in, h_flip ? 1 : 0, chain, lane);
            return;
        }
        // (the run record is read at an offset from the control record's address, where it is needed)
        full_step_draw_wave<T>(a, jump_tables_behind(draws_chain0, h_n, ((hot_bits >> 27) & 1u) != 0, chains), hot_ctl_in, nullptr, false, dn_red, h_n, sh_begin,
                               sh_count, kWavesPerBlock * NW, (int)(threadIdx.x >> 6) - kWavesPerBlock, lane, h_flip ? 1 : 0, chain);
        return;
    }
    // the updating wavefronts go ahead of the draw wavefronts they share SIMDs with (5.646 -> 5.626 us per launch at C2)
    __builtin_amdgcn_s_setprio(3);
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int first = wave * NW;  // (relative to the shard)
    if (first >= sh_count) return;  // (no workgroup barrier in this kernel)
    const int sub = lane & 15, grp = lane >> 4, i0 = sub * 2;
    typedef typename Vec2<T>::type V2;
    const bool col_ok = i0 < h_dims;  // (even D only: rows are whole 16-byte pieces)
    const int i0c = col_ok ? i0 : 0;
    // Unconditional 16-byte loads from always-valid addresses, no masking: lanes beyond D re-read the row's first
    // piece and idle groups re-read walker 0.  Their (finite) values only ever meet the zero padding of P^T (an fma
    // with +-0 leaves the chain untouched) or results that are never stored, and a load nobody masks is a load the
    // compiler has no reason to sink into a branch (which would serialise the round trip).
    // Addresses: (uniform array base, 32-bit byte offset) -- one register and one full-rate 24-bit multiply-add per row
    // address instead of a 64-bit product and a 64-bit add; a dozen of them sit between the first round trip's arrival and
    // the second's departure.  (The host takes this kernel only below 2^24 walkers and 4 GiB of rows.)
    const char* const pin_b = reinterpret_cast<const char*>(pin);
    const char* const lin_b = reinterpret_cast<const char*>(lin);
    const char* const cnt_b = reinterpret_cast<const char*>(h_n_accept);
    const char* const drr_b = reinterpret_cast<const char*>(dr_red);
    const char* const drb_b = reinterpret_cast<const char*>(dr_blk);
    char* const pout_b = reinterpret_cast<char*>(pout);
    char* const lout_b = reinterpret_cast<char*>(lout);
    const uint32_t un = (uint32_t)h_n;
    auto row_off = [&](uint32_t w, int col) -> uint32_t { return (__umul24(w, (uint32_t)h_dims) + (uint32_t)col) * (uint32_t)sizeof(T); };
    auto load_row = [&](uint32_t w, T (&out)[2]) {
        const V2 v = *reinterpret_cast<const V2*>(pin_b + row_off(w, i0c));
        out[0] = v.x;
        out[1] = v.y;
    };
    auto load_rec = [&](const char* base, uint32_t i) -> DrawRec<T> { return *reinterpret_cast<const DrawRec<T>*>(base + i * (uint32_t)sizeof(DrawRec<T>)); };
    auto load_lp = [&](uint32_t w) -> T { return *reinterpret_cast<const T*>(lin_b + w * (uint32_t)sizeof(T)); };

    // ---- first round trip ----
    bool active[2];
    uint32_t ir[2];
    DrawRec<T> rec_r[2], rec_b[2];
    T own_r[2][2], own_b[2][2], lp_r[2], lp_b[2];
    uint32_t nacc_r[2], nacc_b[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
    {
        const int li = first + 4 * q + grp;
        active[q] = li < sh_count;
        ir[q] = (uint32_t)(sh_begin + (active[q] ? li : 0));
        rec_r[q] = load_rec(drr_b, ir[q]);
        rec_b[q] = load_rec(drb_b, ir[q]);
        load_row(ir[q], own_r[q]);
        load_row(un + ir[q], own_b[q]);
        lp_r[q] = load_lp(ir[q]);
        lp_b[q] = load_lp(un + ir[q]);
        nacc_r[q] = *reinterpret_cast<const uint32_t*>(cnt_b + ir[q] * 4u);
        nacc_b[q] = *reinterpret_cast<const uint32_t*>(cnt_b + (un + ir[q]) * 4u);
    }

    // ---- second round trip: everything the records point to ----
    T par_r[2][2], own_x[2][2], par_x[2][2], lp_x[2];
    DrawRec<T> rec_x[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
    {
#if defined(MCMCPP_EXP_PUSH) && (MCMCPP_EXP_PUSH & 1)
        // EXPERIMENT 3i (timing only, THE CHAIN IS WRONG): what a push scheme would read -- the rows the records point to
        // at addresses that follow from the walker's own index (an inbox of three rows per red/black pair), so that they
        // go out with the first round trip instead of behind it
        const uint32_t bx = (3u * ir[q]) & (un - 1u);
        load_row(un + bx, par_r[q]);
        const uint32_t jx = (bx + 1u) & (un - 1u);
        rec_x[q] = load_rec(drr_b, jx);
        load_row(jx, own_x[q]);
        lp_x[q] = load_lp(jx);
        load_row(un + ((bx + 2u) & (un - 1u)), par_x[q]);
#else
        load_row(un + rec_r[q].partner, par_r[q]);
        const uint32_t jx = rec_b[q].partner;
        rec_x[q] = load_rec(drr_b, jx);
        load_row(jx, own_x[q]);
        lp_x[q] = load_lp(jx);
        load_row(un + rec_b[q].partner2, par_x[q]);
#endif
    }
    // The wavefront's share of P^T: 8 x 16 bytes per lane, the same 8 KiB for every wavefront (L2 hits), through a preloaded
    // pointer (no kernarg miss in front).  Behind the second trip's loads on purpose: issued with the first trip they
    // compete with every wavefront's record loads, which the second trip waits for (5.76 -> 5.60 us per launch).
    asm volatile("" ::: "memory");
    McB<T> matB;
    mc_load_b(hot_matrix, sub, grp, matB);
    MCMCPP_STAMP(1);
    // every scalar miss of this wavefront in one batch whose wait overlaps the second round trip (see the plain kernel)
    StepCtl ctl;
    RunInfo run;
    // (the chain's own records, the run record kRunBehindCtlBytes behind the first control record: offsets of the batch's loads)
    load_records_and_warm_args<T>(hot_ctl_in, (unsigned)chain * (unsigned)kCtlChainStride, hot_ctl_in,
                                  (unsigned)chain * (unsigned)kCtlChainStride + (unsigned)kRunBehindCtlBytes - (h_flip ? (unsigned)sizeof(StepCtl) : 0u), ctl, run);
    const StepCtl* const ctl_mine = reinterpret_cast<const StepCtl*>(reinterpret_cast<const char*>(hot_ctl_in) + (size_t)chain * kCtlChainStride);
    if (blockIdx.x == 0 && threadIdx.x == 0) hand_over_full<T>(a, ctl, run, const_cast<StepCtl*>(ctl_mine) + (h_flip ? -1 : 1));
    long long save_slot = -1;
    if (h_use_ctl_save && run.chain != nullptr && ctl.save_phase + 1u == (uint32_t)run.interval) save_slot = (run.chain_slot_base + ctl.chain_slot) & run.slot_mask;

    // accept test of walker slot q; fin = row afterwards
    auto decide = [&](int q, const T (&own)[2], const T (&prop)[2], const DrawRec<T>& rec, T lp_old, T lp_new, bool count_ties, T (&fin)[2], T& lp_fin) -> bool {
        const T delta = rec.zs + lp_new - lp_old;
        const bool accept = rec.ln_u < delta;
        if (count_ties && active[q] && sub == 0)
        {
            const T margin = dev_abs(rec.ln_u - delta);
            const T scale = dev_abs(rec.ln_u) + dev_abs(rec.zs) + dev_abs(lp_new) + dev_abs(lp_old);
            if (margin <= a.tie_eps * scale) count_near_tie(a.diag);
        }
        fin[0] = accept ? prop[0] : own[0];
        fin[1] = accept ? prop[1] : own[1];
        lp_fin = accept ? lp_new : lp_old;
        return accept;
    };
    auto commit = [&](int q, uint32_t w, const T (&fin)[2], T lp_fin, bool accept, uint32_t nacc_old) {
        if (!active[q]) return;
        if (accept || (nacc_old & kRowMovedBit) != 0u)  // (otherwise `out` holds this row already)
        {
            if (col_ok) store_row_piece_at(pout_b, row_off(w, i0), fin[0], fin[1]);
            if (sub == 0)
            {
                store_through_at(lout_b, w * (uint32_t)sizeof(T), lp_fin);
                store_through_at(const_cast<char*>(cnt_b), w * 4u, accept ? ((nacc_old + 1u) | kRowMovedBit) : (nacc_old & ~kRowMovedBit));
            }
        }
        if (save_slot >= 0 && col_ok)
        {
            T* crow = reinterpret_cast<T*>(run.chain) + ((size_t)save_slot * (size_t)(2 * h_n) + (size_t)w) * h_dims;
            *reinterpret_cast<V2*>(crow + i0) = Vec2<T>::make(fin[0], fin[1]);
        }
    };

    // ---- one full tile: rows 0..7 the black walkers' red partners (repeated updates), rows 8..15 the red owners ----
    T prop4[4][2], lp4[4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e)
        {
            const T dx = own_x[q][e] - par_x[q][e];
            const T zdx = rec_x[q].z * dx;
            prop4[q][e] = par_x[q][e] + zdx;
            const T dr = own_r[q][e] - par_r[q][e];
            const T zdr = rec_r[q].z * dr;
            prop4[2 + q][e] = par_r[q][e] + zdr;
        }
    mc_eval<4>(matB, sh_x, sub, grp, h_dims, prop4, lp4);
    MCMCPP_STAMP(2);  // second round trip landed, first tile done

    // ---- the black walkers against their partners' results ----
    T prop_b[2][2], lp_b_new[2], new_x[2], lp_dummy;
#pragma unroll
    for (int q = 0; q < 2; ++q)
    {
        decide(q, own_x[q], prop4[q], rec_x[q], lp_x[q], lp4[q], false, new_x, lp_dummy);
#pragma unroll
        for (int e = 0; e < 2; ++e)
        {
            const T d = own_b[q][e] - new_x[e];
            const T zd = rec_b[q].z * d;
            prop_b[q][e] = new_x[e] + zd;
        }
    }
    // the red owners' rows go out now (every load has long been consumed) and drain while the black tile computes
    unsigned acc_red = 0, acc_blk = 0;
    T fin[2], lp_fin;
#pragma unroll
    for (int q = 0; q < 2; ++q)
    {
        const bool accr = decide(q, own_r[q], prop4[2 + q], rec_r[q], lp_r[q], lp4[2 + q], true, fin, lp_fin) && active[q];
        commit(q, ir[q], fin, lp_fin, accr, nacc_r[q]);
#if defined(MCMCPP_EXP_PUSH) && (MCMCPP_EXP_PUSH & 2)
        // EXPERIMENT 3i: the red walker's final row pushed to the one consumer it has on average (a scattered inbox slot)
        if (a.stamps != nullptr && col_ok) store_row_piece(reinterpret_cast<T*>(a.stamps) + (size_t)(3 * rec_r[q].partner + 1) * h_dims + i0, fin[0], fin[1]);
#endif
        acc_red += (unsigned)__popcll(__ballot(accr && sub == 0));
    }
    MCMCPP_STAMP(3);  // red rows decided, their stores issued
    mc_eval<2>(matB, sh_x + 2 * NW * kMcXS, sub, grp, h_dims, prop_b, lp_b_new);
    MCMCPP_STAMP(4);  // second tile done
#pragma unroll
    for (int q = 0; q < 2; ++q)
    {
        const bool accb = decide(q, own_b[q], prop_b[q], rec_b[q], lp_b[q], lp_b_new[q], true, fin, lp_fin) && active[q];
        commit(q, un + ir[q], fin, lp_fin, accb, nacc_b[q]);
#if defined(MCMCPP_EXP_PUSH) && (MCMCPP_EXP_PUSH & 2)
        // EXPERIMENT 3i: the black walker's final row pushed to the two consumers it has on average
        if (a.stamps != nullptr && col_ok)
        {
            store_row_piece(reinterpret_cast<T*>(a.stamps) + (size_t)(3 * rec_b[q].partner) * h_dims + i0, fin[0], fin[1]);
            store_row_piece(reinterpret_cast<T*>(a.stamps) + (size_t)(3 * rec_b[q].partner2 + 2) * h_dims + i0, fin[0], fin[1]);
        }
#endif
        acc_blk += (unsigned)__popcll(__ballot(accb && sub == 0));
    }
    MCMCPP_STAMP(5);
    MCMCPP_STAMP_BLOCK(1);
#ifdef MCMCPP_STAMPS
    if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
    {
        stamp_val[7] = __builtin_amdgcn_s_memrealtime();
        for (int k = 0; k < 8; ++k) a.stamps[k] = stamp_val[k];
    }
#endif
    if (a.partials != nullptr && run.accepted_per_step != nullptr && lane == 0)
    {
        uint32_t* p = a.partials + ((size_t)chain * (size_t)a.partial_slots + (size_t)ctl.partial_slot) * 2 * (size_t)a.partial_waves;
        p[wave] = acc_red;
        p[(size_t)a.partial_waves + wave] = acc_blk;
    }
}

}  // namespace mcmcpp

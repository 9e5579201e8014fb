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
//   update   (LPW lanes per walker, `passes` rounds of 64/LPW walkers): read the walker's draw record,
//            gather the partner row, form the proposal, evaluate the Calculator functor (cross-lane tree
//            reduction), Metropolis accept in place, optional chain store, per-step accepted count
//   draws    (one lane per draw): the random draws do not depend on the walkers, so the three draws each of
//            these walkers needs at its NEXT update (two half-steps ahead) are computed now and left in a
//            32-byte record per walker, in the other of two record buffers.  A fifth wavefront per workgroup
//            does nothing else (for wide lane groups; otherwise the updating wavefronts do it in the shadow of
//            their partner-row gather).  The launch's dependent chain is then two memory round trips plus the
//            calculator.
// No MFMA here: the work is element-wise plus a per-walker reduction (calculators with a dense product have a
// matrix-core variant further down).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

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
    U128 state2;           // the same two half-steps later (the next update of this colour)
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
    // full-step kernels: the device chain is a ring of slot_mask + 1 slots and every launch forwards a slice of the
    // most recent stored step to its twin in pinned host memory (trickle_stored_step); nullptr: no forwarding
    void* stage;
    int64_t slot_mask;           // stored step k lives in slot k & slot_mask (all ones: slots are not reused)
    int64_t slice_bytes;         // bytes of a stored step forwarded per launch (a multiple of 16) | bit 0: stage slots are not reused
    int64_t step_bytes;          // W * D * sizeof(T) (a multiple of 16 when stage != nullptr)
};

// The random part of one stretch-move update, computed one update ahead (StretchMove.h:102,104,110,113).
template <class T>
struct alignas(16) DrawRec
{
    T z;               // stretch factor
    T zs;              // (D-1) ln z
    T ln_u;            // ln U of the accept test
    uint32_t partner;  // index inside the complementary half
    uint32_t partner2; // black records made for the full-step kernels only: the partner (a black walker) of this walker's
                       // red partner in the same ensemble step, so that both rows can be fetched in one round trip
};

// What a half-step update uses of its record.  The kernels keep THIS across their pass loop, not the 32-byte record: a
// copied DrawRec<float> (20 bytes of fields, 12 of padding, partner2 unused here) is not taken apart into registers -- its
// unused half stays a private array, which the back end "promotes" to LDS, and a kernel with such an array reads the
// workgroup size from the dispatch packet (host memory) at the start of every wavefront: 13-28 us per fp32 half-step
// launch where fp64 took 3-10 (round 3; profiles/r03_fp32_half_step_anomaly.txt).
template <class T>
struct DrawUse
{
    T z, zs, ln_u;
    uint32_t partner;
    __device__ __forceinline__ DrawUse() {}
    __device__ __forceinline__ DrawUse(const DrawRec<T>& r) : z(r.z), zs(r.zs), ln_u(r.ln_u), partner(r.partner) {}
};

struct Diag
{
    unsigned long long near_ties;
    unsigned long long redraws;
};

// An accept decision within a few ulp of a tie (include/mcmcpp_hip.h: near_ties).
__device__ __forceinline__ void count_near_tie(Diag* diag)
{
#ifndef MCMCPP_EXP_NO_TIE_ATOMIC
    atomicAdd(&diag->near_ties, 1ULL);
#else
    (void)diag;
#endif
}

// The jump tables live right behind the draw records, at offsets that follow from the number of walkers per colour
// alone, so that a kernel can reach them from its preloaded record pointer without touching the kernarg segment
// (a draw wavefront's first loads would otherwise wait for a cold scalar miss):
//   [records: 4 n x 32 B][task_jump: 3 n entries, if built][jump_hi: ceil(n / 256) entries][jump_lo: 256 entries]
// every piece rounded up to 256 bytes.
struct JumpTables
{
    const Affine128* task;  // nullptr when the table was not built (very large ensembles)
    const Affine128* hi;
    const Affine128* lo;
};
__host__ __device__ inline size_t round_up_256(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
// (`chains` independent ensembles stepped by one launch share the tables: same stream increment, see ChainGeometry)
__host__ __device__ inline size_t tables_offset_task(int n, int chains = 1) { return round_up_256((size_t)4 * (size_t)n * 32 * (size_t)chains); }
__host__ __device__ inline size_t tables_offset_hi(int n, bool direct, int chains = 1)
{
    return tables_offset_task(n, chains) + (direct ? round_up_256((size_t)3 * (size_t)n * sizeof(Affine128)) : 0);
}
__host__ __device__ inline size_t tables_offset_lo(int n, bool direct, int chains = 1)
{
    return tables_offset_hi(n, direct, chains) + round_up_256((size_t)((n + 255) / 256) * sizeof(Affine128));
}
__host__ __device__ inline size_t tables_total_bytes(int n, bool direct, int chains = 1) { return tables_offset_lo(n, direct, chains) + 256 * sizeof(Affine128); }
__device__ __forceinline__ JumpTables jump_tables_behind(const void* draws_base, int n, bool direct, int chains = 1)
{
    const char* b = static_cast<const char*>(draws_base);
    JumpTables t;
    t.task = direct ? reinterpret_cast<const Affine128*>(b + tables_offset_task(n, chains)) : nullptr;
    t.hi = reinterpret_cast<const Affine128*>(b + tables_offset_hi(n, direct, chains));
    t.lo = reinterpret_cast<const Affine128*>(b + tables_offset_lo(n, direct, chains));
    return t;
}

// Several independent ensembles ("chains": BASELINE config 4 on one GPU) stepped by ONE launch: workgroup row
// blockIdx.y is chain blockIdx.y.  Every per-chain array is the single-chain array repeated with a fixed stride that
// follows from n and D alone, so the kernels need no further arguments (the chain count travels in the hot bits):
//   positions (both buffers)   [chains][2n][D]
//   log-posteriors + counters  [chains]{[2][2n] T, [2n] u32}
//   control + run records      [chains]{StepCtl[2], pad to kRunBehindCtlBytes, RunInfo, pad to kCtlChainStride}
//   draw records               [chains][2][2][n], then the shared jump tables
//   partial accepted counts    [chains][partial_slots][2][partial_waves]
// The chains differ in their seed (seed + chain: same stream increment, hence the same jump tables).
constexpr int kCtlChainStride = 512;
constexpr int kRunBehindCtlBytes = 256;
constexpr int kMaxChains = 16;  // (four hot bits)
template <class T>
__host__ __device__ inline size_t logp_chain_stride_bytes(int n) { return (size_t)4 * (size_t)n * sizeof(T) + (size_t)2 * (size_t)n * sizeof(uint32_t); }

// The launch description.  It travels by value in the kernarg segment (behind 64 bytes of preloaded hot arguments)
// and is read with scalar loads where a field is first used; after a launch boundary every 64-byte line of it is a
// cold miss of several hundred ns, so the fields are grouped by who needs them first: a draw wavefront touches the
// first two lines only (the next two as well with the two-level jump tables).
template <class T>
struct alignas(64) HalfStepArgs
{
    // ---- line 0: the random stream (draw wavefronts, fill_draws_kernel) ----
    const Affine128* task_jump; // [3n] map of t+1 draws (base state -> state behind draw t), or nullptr for large n
    const Affine128* jump_hi;   // [ceil(n/256)] map of 3*256*m draws
    const Affine128* jump_lo;   // [256]   map of 3*k draws
    Diag* diag;
    Affine128 half_jump;        // map of 3*n draws: this half-step's base state -> the next one's
    // ---- line 1: constants of the draws and of the accept test ----
    uint64_t redraw_threshold;  // (2^64 - n) mod n (pcg bounded_rand)
    T gw_term1, gw_inv_sqrt;    // GwDistribution<T,2,1> constants (MCMCpp/Utility/GwDistribution.h:45-55)
    T dims_minus_one;           // (T)(D-1)  (StretchMove.h:110)
    T tie_eps;
    int n;                      // walkers per half
    int n_is_pow2;
    int dims;                   // D
    int color;                  // 0 red = walkers [0,n), 1 black = [n,2n)
    int shard_begin;            // first walker (index inside the half) updated by this launch
    int shard_count;            // number of walkers updated by this launch
    // ---- lines 2-3: two-level jump tables only; diagnostics ----
    alignas(64) Affine128 draw_jump[3];  // maps of 1, 2, 3 draws: a walker's base state -> the state behind draw k
    U128 inc;                   // pcg stream increment
    unsigned long long* stamps; // diagnostic build only (MCMCPP_STAMPS): shader-clock stamps
    // ---- the updating wavefronts ----
    alignas(64) T* pos;         // [W][D]
    T* logp;                    // [W]
    uint32_t* n_accept;         // [W]
    const StepCtl* ctl_in;
    StepCtl* ctl_out;
    const RunInfo* run;
    const T* calc_params;
    DrawRec<T>* draws;          // [2 buffers][2 colours][n] records of the coming updates: the update of ensemble step s
                                // reads buffer s&1 and the draws for step s+1 are written to the other one
    uint32_t* partials;         // [partial_slots][2][partial_waves] per-wavefront accepted counts, or nullptr
    // full-step kernels only (full_step_kernel.hpp): the second position / log-posterior buffer
    T* pos_alt;                 // [W][D]
    T* logp_alt;                // [W]
    const T* calc_params_padded; // matrix-core kernels: P^T zero-padded to 32 x 32 (row stride 32)
    long long direct_save_slot; // >= 0: store into run->chain at this slot regardless of interval (sharded driver)
    int passes;                 // rounds of 64/LPW walkers per wavefront
    int vec_ok;                 // rows are 16-byte aligned multiples: use 128-bit accesses
    int partial_slots;          // ensemble steps between two runs of accepted_reduce_kernel
    int partial_waves;          // wavefronts of one half-step launch
    int use_ctl_save;           // 1: saving follows RunInfo.interval / StepCtl.step_in_run
    int draw_parity;            // ensemble step & 1: which record buffer this launch reads
    int draw_wave;              // 1: the workgroup carries extra wavefronts that compute the next draws
    int pos_parity;             // full-step kernels: 0: read pos/logp, write pos_alt/logp_alt; 1: the reverse
    int chains;                 // independent ensembles stepped by this launch (grid.y), 1..kMaxChains
};

// This lane's EPL elements of a walker row; cells beyond D (and everything when !active) are +0.  Branch-free on
// purpose: the address is clamped to a valid one, the load is unconditional and unwanted data is masked off bitwise.
// A load inside a branch is waited for inside that branch, which serialises what should be one round trip of many
// loads in flight (measured in the full-step kernel: five dependent round trips instead of two).  `row` must be a
// valid row address also for inactive lanes (callers clamp the walker index).
template <class T, int EPL, bool VEC>
__device__ __forceinline__ void load_slice_as(const T* row, int i0, int D, bool active, T (&out)[EPL])
{
    constexpr int VN = Vec16<T>::N;
    typedef T VX __attribute__((ext_vector_type(VN)));
    typedef typename std::conditional<sizeof(T) == 8, unsigned long long, unsigned>::type Bits;
    if (VEC)
    {
#pragma unroll
        for (int v = 0; v < EPL / VN; ++v)
        {
            const int col = i0 + v * VN;
            const bool in = active && col < D;  // (D is a multiple of VN here: a piece lies wholly inside or outside)
            const VX x = *reinterpret_cast<const VX*>(row + (in ? col : 0));
            const Bits mask = in ? ~(Bits)0 : (Bits)0;
#pragma unroll
            for (int k = 0; k < VN; ++k) out[v * VN + k] = __builtin_bit_cast(T, (Bits)(__builtin_bit_cast(Bits, (T)x[k]) & mask));
        }
    }
    else
    {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
        {
            const bool in = active && i0 + e < D;
            const T x = row[in ? i0 + e : 0];
            const Bits mask = in ? ~(Bits)0 : (Bits)0;
            out[e] = __builtin_bit_cast(T, (Bits)(__builtin_bit_cast(Bits, x) & mask));
        }
    }
}
// The half-step kernels' row load (one row per round trip): lanes that have nothing to fetch skip the access.
// (The branch-free variant above is slower here -- 8.3 against 7.6 us per launch at 65 536 walkers -- and only pays
// where several rows must travel in one round trip.)
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

// Write-through stores (sc0 sc1: system scope).  What a launch writes with plain stores sits dirty in L2 until the
// end-of-kernel write-back, which lengthens the gap to the next launch; written through, the data drains while the
// kernel still runs.  Measured on the full-step kernel at 16384 x 32 (us per launch): plain 6.70, non-temporal 6.52,
// write-through 6.10, no row stores at all 5.77.
__device__ __forceinline__ void store_through(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_through(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_through(double* p, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
template <class V16>
__device__ __forceinline__ void store_through16(void* p, V16 v)
{
    static_assert(sizeof(V16) == 16, "a 16-byte vector");
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// STREAM: write-through stores (see store_through), for the full-step kernels' `out` buffer.
template <class T, int EPL, bool STREAM = false>
__device__ __forceinline__ void store_slice(T* row, int i0, int D, bool vec_ok, const T (&val)[EPL])
{
    constexpr int VN = Vec16<T>::N;
    typedef T VX __attribute__((ext_vector_type(VN)));
    if (vec_ok)
    {
#pragma unroll
        for (int v = 0; v < EPL / VN; ++v)
        {
            if (i0 + v * VN < D)
            {
                VX x;
#pragma unroll
                for (int k = 0; k < VN; ++k) x[k] = val[v * VN + k];
                if (STREAM)
                    store_through16(row + i0 + v * VN, x);
                else
                    *reinterpret_cast<VX*>(row + i0 + v * VN) = x;
            }
        }
    }
    else
    {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
            if (i0 + e < D)
            {
                if (STREAM)
                    store_through(row + i0 + e, val[e]);
                else
                    row[i0 + e] = val[e];
            }
    }
}

// A pointer that was loaded from memory (a run record's chain or counter address) is a generic pointer to the compiler:
// accesses through it become flat_* instructions, which tick both memory counters, and in front of which the compiler
// waits for every outstanding global operation (seen in the differential-evolution update kernel: a wait for the row stores' acknowledgement at
// the end of every updating wavefront).  Device memory is all such a pointer ever holds: say so.
template <class P>
__device__ __forceinline__ P* assume_global(P* p)
{
    typedef __attribute__((address_space(1))) unsigned char GlobalByte;
    // (through the global address space and back: the compiler's address-space inference follows the cast)
    return reinterpret_cast<P*>((unsigned char*)reinterpret_cast<GlobalByte*>(reinterpret_cast<unsigned long long>(p)));
}

__device__ __forceinline__ double dev_log(double x) { return fast_log(x); }
__device__ __forceinline__ float dev_log(float x) { return logf(x); }
__device__ __forceinline__ double dev_abs(double x) { return fabs(x); }
__device__ __forceinline__ float dev_abs(float x) { return fabsf(x); }

#ifndef MCMCPP_WAVES_PER_BLOCK  // (experiment builds: make VARIANT=... EXTRA=-DMCMCPP_WAVES_PER_BLOCK=2)
#define MCMCPP_WAVES_PER_BLOCK 4
#endif
constexpr int kWavesPerBlock = MCMCPP_WAVES_PER_BLOCK;

// Diagnostic build only (make STAMPS=1 -> libmcmcpp_hip_stamps.so): wavefront 0 of workgroup 0 drains its
// memory counters and records the shader clock at a few points; the product build compiles none of it.
#ifdef MCMCPP_STAMPS
#ifdef MCMCPP_STAMPS_DRAIN  // attribute waits to the segment that issued the memory operations
#define MCMCPP_STAMP_DRAIN "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
#else  // leave the kernel's own overlap intact: only read the clock
#define MCMCPP_STAMP_DRAIN ""
#endif
#define MCMCPP_STAMP(k)                                                                     \
    do                                                                                      \
    {                                                                                       \
        if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x < 64)                     \
        {                                                                                   \
            unsigned long long t_;                                                          \
            asm volatile(MCMCPP_STAMP_DRAIN "s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            if (threadIdx.x == 0) stamp_val[k] = t_;                                        \
        }                                                                                   \
    } while (0)
// start / end of every workgroup's first wavefront on the chip-wide 100 MHz clock
#define MCMCPP_STAMP_BLOCK(which)                                                                               \
    do                                                                                                          \
    {                                                                                                           \
        if (a.stamps != nullptr && threadIdx.x == 0 && blockIdx.x < 4096)                                       \
            a.stamps[8 + (a.pos_parity | a.color) * 3 * 4096 + 2 * blockIdx.x + (which)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define MCMCPP_STAMP(k) \
    do                  \
    {                   \
    } while (0)
#define MCMCPP_STAMP_BLOCK(which) \
    do                            \
    {                             \
    } while (0)
#endif

// dynamic LDS of one workgroup: [proposal stage (if the calculator wants it)][calculator tables]
template <class T, class Calc, int EPL>
struct LdsLayout
{
    __host__ __device__ static constexpr size_t stage_offset() { return 0; }
    __host__ __device__ static constexpr size_t block_offset()
    {
        return stage_offset() + (Calc::kNeedsStage ? (size_t)kWavesPerBlock * 64 * EPL * sizeof(T) : 0);
    }
    __host__ static size_t bytes(int dims) { return block_offset() + Calc::block_scratch_elems(dims) * sizeof(T); }
};

// The launch description sits in the kernarg segment and is read with scalar loads wherever a field is first used;
// after a launch boundary every one of its 64-byte lines is a cold miss (several hundred ns), and the compiler
// places those loads lazily, one dependent miss after the other along a wavefront's path.  This touches all lines
// at once and waits for them, so that every later field access hits the scalar cache: call it where a wait is
// free (behind the first vector loads of an updating wavefront; first thing in a draw wavefront).
// (The kernels' argument lists are 64 bytes of preloaded hot arguments followed by the HalfStepArgs: the struct
// occupies bytes 64 .. 64 + sizeof of the kernarg segment; the hidden arguments follow it.)
// The control record, the run record and every line of the launch description in ONE batch of scalar loads with one
// wait.  Inline assembly on purpose: as plain loads the compiler issues each where it is first needed -- three
// dependent cold misses in a row along a wavefront's path -- and is free to hoist a wait in front of vector loads
// that should have gone out first (it did: in front of the second round trip).
// The records are read at base + offset with the offsets in SGPRs (the s_load's own offset operand): a base that comes
// from the kernarg segment -- the control record's address is not among the preloaded arguments -- then needs no
// arithmetic outside this batch, so nothing can drag the wait for it in front of vector loads issued earlier.  (Pointer
// arithmetic on such a base, even pinned with an empty asm, was scheduled ahead of the matrix loads: 5.63 -> 6.01 us.)
template <class T>
__device__ __forceinline__ void load_records_and_warm_args(const StepCtl* ctl_ptr, unsigned ctl_off, const void* run_ptr, unsigned run_off, StepCtl& ctl,
                                                           RunInfo& run)
{
    static_assert(64 + sizeof(HalfStepArgs<T>) > 0x180 && 64 + sizeof(HalfStepArgs<T>) <= 0x200, "adjust the lines touched below");
    static_assert(sizeof(StepCtl) == 64 && sizeof(RunInfo) == 64, "one s_load_dwordx16 each");
    typedef unsigned v16u __attribute__((ext_vector_type(16)));
    const unsigned long long k = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    // (wave-uniform by construction; made so for the compiler too: the pointers may have been chosen by wavefront index)
    auto uniform = [](const void* p) {
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    };
    const unsigned long long ctl_addr = uniform(ctl_ptr), run_addr = uniform(run_ptr);
    v16u c, r;
    unsigned t1, t2, t3, t4, t5, t6, t7;
    const unsigned off_c = __builtin_amdgcn_readfirstlane(ctl_off), off_r = __builtin_amdgcn_readfirstlane(run_off);
    asm volatile(
        "s_load_dwordx16 %0, %9, %12\n\t"
        "s_load_dwordx16 %1, %10, %13\n\t"
        "s_load_dword %2, %11, 0x40\n\t"
        "s_load_dword %3, %11, 0x80\n\t"
        "s_load_dword %4, %11, 0xc0\n\t"
        "s_load_dword %5, %11, 0x100\n\t"
        "s_load_dword %6, %11, 0x140\n\t"
        "s_load_dword %7, %11, 0x180\n\t"
        "s_load_dword %8, %11, 0x1c0\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(c), "=&s"(r), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7)
        : "s"(ctl_addr), "s"(run_addr), "s"(k), "s"(off_c), "s"(off_r)
        : "memory");
    ctl = __builtin_bit_cast(StepCtl, c);
    run = __builtin_bit_cast(RunInfo, r);
}

template <class T>
__device__ __forceinline__ void load_records_and_warm_args(const StepCtl* ctl_ptr, const RunInfo* run_ptr, StepCtl& ctl, RunInfo& run)
{
    load_records_and_warm_args<T>(ctl_ptr, 0u, run_ptr, 0u, ctl, run);
}

// One random draw of one walker: task k of the walker at position i of the half (draw 3*i + k of the
// half-step whose base engine state is `base`), written into the walker's record.  k = 0: partner =
// engine(n) (StretchMove.h:102); k = 1: z = Gw(u) and (D-1) ln z (StretchMove.h:104,110); k = 2:
// ln U = -(-log(1-u)/1) (StretchMove.h:113).  One lane per draw keeps the dependent chain short.
// In pieces, so that a caller with several draws per lane can overlap their memory accesses:

// the raw 64-bit output of the task
template <class T>
__device__ __forceinline__ uint64_t draw_raw(const HalfStepArgs<T>& a, U128 base, const Affine128& jump_a, const Affine128& jump_b, bool direct, int k)
{
    U128 s;
    if (direct)
        s = apply(jump_a, base);
    else
    {
        s = apply(jump_b, apply(jump_a, base));
        const Affine128 dj = k == 0 ? a.draw_jump[0] : (k == 1 ? a.draw_jump[1] : a.draw_jump[2]);
        s = apply(dj, s);
    }
    return pcg_output(s);
}
template <class T>
__device__ __forceinline__ uint32_t draw_partner(const HalfStepArgs<T>& a, uint64_t r)
{
    return a.n_is_pow2 ? (uint32_t)(r & (uint64_t)(a.n - 1)) : (uint32_t)(r % (uint64_t)a.n);
}
// the record fields of task k from its raw output
template <class T>
__device__ __forceinline__ void draw_store(const HalfStepArgs<T>& a, int k, uint64_t r, DrawRec<T>* rec)
{
    if (k == 0)
    {
        if (r < a.redraw_threshold) atomicAdd(&a.diag->redraws, 1ULL);
        rec->partner = draw_partner<T>(a, r);
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
            rec->z = z;
            rec->zs = lg * a.dims_minus_one;
        }
        else
            rec->ln_u = lg;  // (plain stores: written through, the scattered record fields cost more than they save)
    }
}
// Black records made for the full-step kernels: the partner draw of this walker's red partner j (draw 3*j + 0
// of the red half-step of the same ensemble step, base state `red_base`), kept as partner2.
template <class T>
__device__ __forceinline__ void draw_partner2_jump(const HalfStepArgs<T>& a, uint32_t j, bool direct, Affine128& g_a, Affine128& g_b)
{
    g_a = *(direct ? a.task_jump + 3 * (size_t)j : a.jump_hi + (j >> 8));  // branch-free: one pointer, one load
    g_b = a.jump_lo[direct ? 0u : (j & 255u)];
}
template <class T>
__device__ __forceinline__ uint32_t draw_partner2(const HalfStepArgs<T>& a, U128 red_base, const Affine128& g_a, const Affine128& g_b, bool direct)
{
    return draw_partner<T>(a, draw_raw<T>(a, red_base, g_a, g_b, direct, 0));
}

template <class T>
__device__ __forceinline__ void compute_draw(const HalfStepArgs<T>& a, U128 base, const Affine128& jump_a,
                                             const Affine128& jump_b, bool direct, int k, DrawRec<T>* rec, bool with_partner2 = false,
                                             U128 red_base = U128())
{
    const uint64_t r = draw_raw<T>(a, base, jump_a, jump_b, direct, k);
    draw_store<T>(a, k, r, rec);
    if (with_partner2 && k == 0)
    {
        Affine128 g_a, g_b;
        draw_partner2_jump<T>(a, draw_partner<T>(a, r), direct, g_a, g_b);
        rec->partner2 = draw_partner2<T>(a, red_base, g_a, g_b, direct);
    }
}

// The run record lies kRunBehindCtlBytes behind the first of the two control records (one allocation): a kernel short
// of preloaded arguments reads it at an offset from the control record's address (load_records_and_warm_args).
// Hot scalars of a launch, packed so that the arguments every wavefront needs before its first memory
// access fit the 16 dwords the command processor preloads into SGPRs (-amdgpu-kernarg-preload-count=16);
// everything else stays in the by-value HalfStepArgs and is fetched from the kernarg segment on demand.
struct HotBits
{
    // (bit 26: full-step kernels' position-buffer parity; bit 27: the one-entry-per-draw jump table exists;
    //  bits 28-31: chains - 1)
    static __host__ __device__ uint32_t pack(int dims, int passes, int color, int vec_ok, int n_is_pow2, int use_ctl_save,
                                             int draw_parity, int draw_wave, int direct_jump)
    {
        return (uint32_t)dims | ((uint32_t)passes << 12) | ((uint32_t)color << 20) | ((uint32_t)vec_ok << 21) |
               ((uint32_t)n_is_pow2 << 22) | ((uint32_t)use_ctl_save << 23) | ((uint32_t)draw_parity << 24) |
               ((uint32_t)draw_wave << 25) | ((uint32_t)direct_jump << 27);
    }
};

// Hands the random stream and the step counters to the next half-step launch (one lane of the whole grid).
template <class T>
__device__ __forceinline__ void hand_over(const HalfStepArgs<T>& a, const StepCtl& ctl, const RunInfo& run, int color, StepCtl* ctl_out)
{
    StepCtl nx = ctl;
    nx.state = apply(a.half_jump, ctl.state);
    nx.state2 = apply(a.half_jump, ctl.state2);
    nx.half_step = ctl.half_step + 1;
    if (color)
    {
        // the ensemble step ends with the black half: advance the per-step counters
        const bool saved = ctl.save_phase + 1u == (uint32_t)run.interval;
        nx.step_in_run = ctl.step_in_run + 1;
        nx.save_phase = saved ? 0u : ctl.save_phase + 1u;
        nx.chain_slot = ctl.chain_slot + (saved ? 1 : 0);
        nx.partial_slot = (ctl.partial_slot + 1u == (uint32_t)a.partial_slots) ? 0u : ctl.partial_slot + 1u;
    }
    *ctl_out = nx;
}

// Stored steps reach the host without a copy engine and without a gap in the launch sequence: every launch forwards
// 1/interval of the most recent stored step from the device chain ring to its twin in pinned host memory (one
// 16-byte piece per lane, spread over the first draw wavefront of every workgroup, written through so that the
// PCIe traffic leaves while the kernel is still busy; measured: up to ~64 KB per launch are free, a separate copy
// of the 4 MiB step costs 75 us of the launch stream).  Stored step k is complete in host memory when ensemble step
// (k + 2) * interval - 1 has finished; the host copies the run's last stored step itself.
__device__ __forceinline__ void trickle_stored_step(const RunInfo& run, const StepCtl& ctl, int lane)
{
    const long long prev = run.chain_slot_base + ctl.chain_slot - 1;  // the stored step before the one this interval ends with
    if (run.stage == nullptr || prev < 0) return;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const size_t slot_off = (size_t)(prev & run.slot_mask) * (size_t)run.step_bytes;
    // bit 0 of slice_bytes: `stage` is the stored steps' FINAL place (a pinned Chain block: slot = stored step index),
    // not a ring that mirrors the device's
    const size_t slice = (size_t)(run.slice_bytes & ~(int64_t)15);
    const size_t begin = (size_t)ctl.save_phase * slice;
    size_t end = begin + slice;
    if (end > (size_t)run.step_bytes) end = (size_t)run.step_bytes;
    const char* src = static_cast<const char*>(run.chain) + slot_off;
    char* dst = static_cast<char*>(run.stage) + ((run.slice_bytes & 1) ? (size_t)prev * (size_t)run.step_bytes : slot_off);
    for (size_t off = begin + ((size_t)blockIdx.x * 64 + (size_t)lane) * 16; off < end; off += (size_t)gridDim.x * 64 * 16)
        store_through16(dst + off, *reinterpret_cast<const v4u*>(src + off));
}

// Body of the workgroup's extra wavefront (when HalfStepArgs::draw_wave): the next draws of all walkers the
// workgroup updates -- `group_walkers` walkers of each of `colours` colours (1: a half-step launch, base state
// ctl.state2; 2: a full-step launch, red then black, the black base one half-step further), starting at walker
// `group_first` of the shard.  It runs beside the updating wavefronts, so the draw arithmetic is on nobody's
// critical path; its own chain is kept short by fetching every round's jump entries before anything else (and
// before the workgroup barrier the updating wavefronts need for their tables, when there is one).
template <class T, int MAXR>
__device__ __forceinline__ void draw_wave_body(const HalfStepArgs<T>& a, const JumpTables& tab, const StepCtl* ctl_ptr, bool block_barrier,
                                               DrawRec<T>* write0, DrawRec<T>* write1, int colours, int shard_begin, int shard_count,
                                               int group_first, int group_walkers, int lane, bool black_only = false,
                                               const RunInfo* trickle_run = nullptr, int run_behind_ctl = -1, int ctl_chain = 0)
{
#ifdef MCMCPP_STAMPS
    unsigned long long dstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MCMCPP_DSTAMP(k, drain)                                                                             \
    do                                                                                                      \
    {                                                                                                       \
        if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                              \
        dstamp[k] = __builtin_amdgcn_s_memrealtime();                                                       \
    } while (0)
#else
#define MCMCPP_DSTAMP(k, drain) \
    do                          \
    {                           \
    } while (0)
#endif
    MCMCPP_DSTAMP(0, false);
    const bool direct = tab.task != nullptr;
    const int per_colour = 3 * group_walkers;
    const int tasks = colours * per_colour;
    const int last = shard_begin + shard_count - 1;
    Affine128 j_a[MAXR], j_b[MAXR];
    int wi[MAXR], kk[MAXR], cc[MAXR];
    bool ok[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r)
    {
        const int t = lane + 64 * r;
        cc[r] = t >= per_colour ? 1 : 0;
        const int tt = t - cc[r] * per_colour;
        const int slot = tt / 3;
        kk[r] = tt - 3 * slot;
        ok[r] = t < tasks && group_first + slot < shard_count;
        wi[r] = min(shard_begin + group_first + slot, last);  // always a valid table index: the loads are unconditional
        // (table addresses from preloaded arguments: these loads go out before any scalar miss is waited for)
        j_a[r] = *(direct ? tab.task + (3 * wi[r] + kk[r]) : tab.hi + (wi[r] >> 8));  // branch-free: one pointer, one load
        j_b[r] = tab.lo[direct ? 0 : (wi[r] & 255)];
    }
    // the control record and the lines of the launch description this wavefront needs, in one batch of scalar loads
    StepCtl ctl;
    {
        RunInfo run;  // (only wanted when this wavefront also forwards stored steps; otherwise the control record twice)
        // (the chain's own records, ChainGeometry: ctl_chain control-record strides further; the run record either given or,
        //  run_behind_ctl >= 0, kRunBehindCtlBytes behind the first control record -- all as offsets of the batch's loads)
        const unsigned ctl_off = (unsigned)ctl_chain * (unsigned)kCtlChainStride;
        const bool wants_run = trickle_run != nullptr || run_behind_ctl >= 0;
        if (run_behind_ctl >= 0)
            load_records_and_warm_args<T>(ctl_ptr, ctl_off, ctl_ptr, ctl_off + (unsigned)kRunBehindCtlBytes - (unsigned)run_behind_ctl * (unsigned)sizeof(StepCtl), ctl, run);
        else
            load_records_and_warm_args<T>(ctl_ptr, ctl_off, trickle_run != nullptr ? static_cast<const void*>(trickle_run) : static_cast<const void*>(ctl_ptr),
                                          trickle_run != nullptr ? 0u : ctl_off, ctl, run);
        if (wants_run) trickle_stored_step(run, ctl, lane);
    }
    MCMCPP_DSTAMP(1, true);
    if (block_barrier) __syncthreads();  // keep the workgroup barrier count whole
    MCMCPP_DSTAMP(2, false);
    // black_only: the full-step kernels' second draw wavefront (black records of the next step, with partner2)
    const bool with_p2 = colours > 1 || black_only;
    const U128 base1 = with_p2 ? apply(a.half_jump, ctl.state2) : ctl.state2;
    uint64_t raw[MAXR];
    bool p2[MAXR];
    // raw outputs of all rounds; then the partner2 gathers of all rounds go out together, ahead of every store
#pragma unroll
    for (int r = 0; r < MAXR; ++r)
    {
        const bool blk = black_only || cc[r] != 0;
        raw[r] = draw_raw<T>(a, blk ? base1 : ctl.state2, j_a[r], j_b[r], direct, kk[r]);
        p2[r] = ok[r] && with_p2 && blk && kk[r] == 0;
    }
    // (the jump entries of this round are spent: their registers take the gathered ones; every lane fetches, lanes
    //  without a partner draw entry 0)
#pragma unroll
    for (int r = 0; r < MAXR; ++r)
    {
        const uint32_t jj = p2[r] ? draw_partner<T>(a, raw[r]) : 0u;
        j_a[r] = *(direct ? tab.task + 3 * (size_t)jj : tab.hi + (jj >> 8));  // branch-free: one pointer, one load
        j_b[r] = tab.lo[direct ? 0u : (jj & 255u)];
    }
    MCMCPP_DSTAMP(5, false);  // raw outputs computed, gathers issued
#pragma unroll
    for (int r = 0; r < MAXR; ++r)
        if (ok[r]) draw_store<T>(a, kk[r], raw[r], (cc[r] ? write1 : write0) + wi[r]);
    MCMCPP_DSTAMP(6, false);  // record fields computed and stored
    MCMCPP_DSTAMP(7, true);   // gathers landed
#pragma unroll
    for (int r = 0; r < MAXR; ++r)
        if (p2[r]) ((cc[r] ? write1 : write0) + wi[r])->partner2 = draw_partner2<T>(a, ctl.state2, j_a[r], j_b[r], direct);
    MCMCPP_DSTAMP(3, false);
#ifdef MCMCPP_STAMPS
    if (a.stamps != nullptr && lane == 0 && blockIdx.x < 4096)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (blockIdx.x == 0)
        {
            dstamp[4] = __builtin_amdgcn_s_memrealtime();
            for (int k = 0; k < 8; ++k) a.stamps[8 + 2 * 3 * 4096 + k] = dstamp[k];
        }
        a.stamps[8 + (a.pos_parity | a.color) * 3 * 4096 + 2 * 4096 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// DW: built with the extra draw wavefront (HalfStepArgs::draw_wave) or without it.  Two instantiations on purpose: the
// draw wavefront's code raises the kernel's register count (85 against 64 VGPRs for the isotropic target), which
// costs the variant without it -- the one large, bandwidth-bound ensembles take -- occupancy (measured at 1 M
// walkers: 73.7 against 68.8 us per launch).
template <class T, class Calc, int EPL, int LPW, bool DW, bool MC = false>
__global__ void __launch_bounds__(64 * (kWavesPerBlock + (DW ? 1 : 0)))
stretch_half_step_kernel(DrawRec<T>* hot_draws, T* hot_pos, T* hot_logp, uint32_t* hot_n_accept, int hot_n, uint32_t hot_bits,
                         int hot_shard_begin, int hot_shard_count, const StepCtl* hot_ctl_in, const HalfStepArgs<T> rest)
{
    // `a` is the launch description in the kernarg segment; the h_* locals are the preloaded copies of the
    // fields every wavefront needs before its first memory access (same values, no kernarg fetch)
    const HalfStepArgs<T>& a = rest;
    // chain blockIdx.y of (hot_bits >> 28) + 1 (ChainGeometry): every per-chain array at its fixed stride; a branch on
    // purpose -- chain 0, i.e. every single-ensemble launch, skips the 64-bit products
        // (MC: built for several chains per launch; the single-ensemble instantiation carries none of it -- measured 2 % of a
    //  65 536-walker launch otherwise)
    const int chain = MC ? (int)blockIdx.y : 0, chains = MC ? (int)(hot_bits >> 28) + 1 : 1;
    const void* const draws_chain0 = hot_draws;
    if (MC && chain != 0)
    {
        hot_draws += (size_t)chain * 4 * (size_t)hot_n;
        hot_pos += (size_t)chain * 2 * (size_t)hot_n * (size_t)(hot_bits & 0xFFFu);
        hot_logp = reinterpret_cast<T*>(reinterpret_cast<char*>(hot_logp) + (size_t)chain * logp_chain_stride_bytes<T>(hot_n));
        hot_n_accept = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(hot_n_accept) + (size_t)chain * logp_chain_stride_bytes<T>(hot_n));
    }
    const int h_color = (int)((hot_bits >> 20) & 1u);
    const int h_parity = (int)((hot_bits >> 24) & 1u);
    constexpr bool h_draw_wave = DW;  // (hot_bits bit 25 says the same)
    // draw records: this launch reads buffer `parity`, the draws of the colour's next update go to the other one
    const DrawRec<T>* const h_draws = hot_draws + ((size_t)h_parity * 2 + (size_t)h_color) * (size_t)hot_n;
    DrawRec<T>* const h_draws_next = hot_draws + ((size_t)(1 - h_parity) * 2 + (size_t)h_color) * (size_t)hot_n;
    T* const h_pos = hot_pos;
    T* const h_logp = hot_logp;
    uint32_t* const h_n_accept = hot_n_accept;
    const int h_n = hot_n;
    const int h_dims = (int)(hot_bits & 0xFFFu);
    const int h_passes = (int)((hot_bits >> 12) & 0xFFu);
    const int h_vec_ok = (int)((hot_bits >> 21) & 1u);
    const int h_use_ctl_save = (int)((hot_bits >> 23) & 1u);
    const int h_shard_begin = hot_shard_begin;
    const int h_shard_count = hot_shard_count;
    static_assert((LPW & (LPW - 1)) == 0 && LPW >= 1 && LPW <= 64, "LPW must be a power of two <= 64");
    static_assert(EPL % Vec16<T>::N == 0, "EPL must be a whole number of 16-byte vectors");
    static_assert(sizeof(DrawRec<T>) % 16 == 0, "draw records are read with 16-byte loads");
    constexpr int WPP = 64 / LPW;  // walkers per pass

    // LDS carve-up (all dynamic, 16-byte aligned pieces): proposal stage | calculator tables
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* sh_stage = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::stage_offset());
    T* sh_block = reinterpret_cast<T*>(smem + LdsLayout<T, Calc, EPL>::block_offset());

#ifdef MCMCPP_STAMPS
    unsigned long long stamp_val[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    MCMCPP_STAMP(0);
#ifdef MCMCPP_STAMPS
    stamp_val[6] = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int nw = WPP * h_passes;
    if constexpr (DW)
    {
        if (wib == kWavesPerBlock)
        {
            // the workgroup's extra wavefront: next draws of every walker this workgroup updates
            draw_wave_body<T, 2>(a, jump_tables_behind(draws_chain0, hot_n, ((hot_bits >> 27) & 1u) != 0, chains), hot_ctl_in, Calc::block_scratch_elems(h_dims) != 0,
                                 h_draws_next, h_draws_next, 1, h_shard_begin, h_shard_count, blockIdx.x * kWavesPerBlock * nw, kWavesPerBlock * nw, lane, false,
                                 nullptr, -1, chain);
            return;
        }
    }
    const int wave = blockIdx.x * kWavesPerBlock + wib;
    const int first = wave * nw;  // first walker of this wavefront, relative to the shard
    const bool wave_active = first < h_shard_count;
    const int half_base = h_color ? h_n : 0;
    const int other_base = h_color ? 0 : h_n;
    const int sub = lane & (LPW - 1);
    const int grp = lane / LPW;
    const int i0 = sub * EPL;
    const bool vec_ok = h_vec_ok != 0;
    const int last_li = h_shard_count - 1;

    // ---- first round trip: everything that is addressed by the walker index alone --------------------------
    // Issued first, from preloaded arguments only (no kernarg fetch in front of them): the draw record (whose
    // partner index the second round trip waits for), own rows, log-posterior, counter.
    GroupCtx<T, EPL, LPW> ctx;
    ctx.sub = sub;
    ctx.dims = h_dims;
    ctx.lane = lane;
    ctx.stage = Calc::kNeedsStage ? &sh_stage[wib * 64 * EPL] : nullptr;
    ctx.vec_ok = vec_ok;

    T own[EPL];
    T lp_old;
    uint32_t nacc_old = 0;
    DrawUse<T> rec;
    {
        const int li0 = first + grp;
        const bool act0 = wave_active && li0 < h_shard_count;
        const int w0 = half_base + h_shard_begin + (act0 ? li0 : 0);
        rec = DrawUse<T>(h_draws[w0 - half_base]);  // the LPW lanes of a walker read the same bytes: one transaction
        load_slice<T, EPL>(h_pos + (size_t)w0 * h_dims, i0, h_dims, vec_ok, act0, own);
        lp_old = h_logp[w0];
        nacc_old = h_n_accept[w0];  // every lane of the group reads the same word: no divergent branch, no wait
    }
    typename Calc::Prefetch calc_pf;
    Calc::block_prefetch(calc_pf, a.calc_params, h_dims, vec_ok, (int)threadIdx.x, 64 * kWavesPerBlock);

    const StepCtl* ctl_mine = hot_ctl_in;
    const RunInfo* run_mine = a.run;
    if (MC && chain != 0)  // (the chain's own records, ChainGeometry)
    {
        ctl_mine = reinterpret_cast<const StepCtl*>(reinterpret_cast<const char*>(ctl_mine) + (size_t)chain * kCtlChainStride);
        run_mine = reinterpret_cast<const RunInfo*>(reinterpret_cast<const char*>(run_mine) + (size_t)chain * kCtlChainStride);
    }
    const StepCtl ctl = *ctl_mine;  // wave-uniform
    const RunInfo run = *run_mine;

    // the draws of the NEXT update of this wavefront's walkers: task t = 3*slot + k is draw k of walker `slot`
    const int tasks = 3 * nw;
    const int slot_a = lane / 3, k_a = lane - 3 * slot_a;
    const int i_a = h_shard_begin + (wave_active ? min(first + slot_a, last_li) : 0);
    // small ensembles jump with one table entry per draw (one 128-bit multiply-add), large ones compose a
    // two-level table (256-walker blocks x position inside the block) with the draw offset
    const bool direct_jump = a.task_jump != nullptr;
    Affine128 j_a, j_b;
    if (direct_jump)
        j_a = a.task_jump[3 * i_a + k_a];
    else
    {
        j_a = a.jump_hi[i_a >> 8];
        j_b = a.jump_lo[i_a & 255];
    }

    // ---- second round trip: the partner rows of the first pass (needs only rec.partner) --------------------
    T par[EPL];
    load_slice<T, EPL>(h_pos + (size_t)(other_base + (int)rec.partner) * h_dims, i0, h_dims, vec_ok,
                       wave_active && first + grp < h_shard_count, par);
    MCMCPP_STAMP(1);  // records landed, partner gather issued (diagnostic build: also landed)

    // ---- in its shadow: the calculator's tables, the hand-over to the next launch, the next draws -----------
    const bool has_block_scratch = Calc::block_scratch_elems(h_dims) != 0;
    Calc::block_commit(calc_pf, sh_block, a.calc_params, h_dims, vec_ok, (int)threadIdx.x, 64 * kWavesPerBlock);
    if (has_block_scratch) __syncthreads();
    ctx.block_scratch = has_block_scratch ? sh_block : nullptr;
    typename Calc::template Regs<EPL, LPW> cregs;
    Calc::template preload<EPL, LPW>(ctx, a.calc_params, cregs);

    // one lane of the grid hands the stream and the counters to the next launch (in the shadow of its gather wait;
    // in the extra wavefront it would lengthen the last wavefront to finish: measured)
    if (blockIdx.x == 0 && threadIdx.x == 0)
        hand_over<T>(a, ctl, run, h_color, reinterpret_cast<StepCtl*>(reinterpret_cast<char*>(a.ctl_out) + (size_t)chain * kCtlChainStride));
    if (!wave_active) return;

    // does this ensemble step go to the chain?  (EnsembleSampler.h:298-306: interval-1 unsaved, 1 saved)
    long long save_slot = -1;
    if (a.direct_save_slot >= 0)
        save_slot = a.direct_save_slot;
    else if (h_use_ctl_save && run.chain != nullptr && ctl.save_phase + 1u == (uint32_t)run.interval)
        save_slot = run.chain_slot_base + ctl.chain_slot;

    // Draws of this wavefront's walkers for their next update (half-step + 2, base state ctl.state2).  With one
    // pass per wavefront the records being replaced were all read above, so the draws are computed here, in
    // the shadow of the partner gather; with several passes the later passes still have to read theirs, so the
    // draws wait until the update loop is done.
    auto next_draws = [&]() {
        for (int t = lane; t < tasks; t += 64)
        {
            const int slot = (t == lane) ? slot_a : t / 3;
            const int k = (t == lane) ? k_a : t - 3 * slot;
            if (first + slot < h_shard_count)
            {
                const int i = h_shard_begin + first + slot;
                if (t != lane)
                {
                    if (direct_jump)
                        j_a = a.task_jump[3 * i + k];
                    else
                    {
                        j_a = a.jump_hi[i >> 8];
                        j_b = a.jump_lo[i & 255];
                    }
                }
                compute_draw<T>(a, ctl.state2, j_a, j_b, direct_jump, k, h_draws_next + i);
            }
        }
    };
    MCMCPP_STAMP(2);

    // ---------------- the update: LPW lanes per walker ------------------------------------------------------
    unsigned accepted_here = 0;
    for (int q = 0; q < h_passes; ++q)
    {
        const int li = first + q * WPP + grp;
        const bool active = li < h_shard_count;
        const int w = half_base + h_shard_begin + (active ? li : 0);
        T* row = h_pos + (size_t)w * h_dims;

        // the next pass's record, own rows and counters travel while this pass computes
        T own_next[EPL];
        T lp_next = (T)0;
        uint32_t nacc_next = 0;
        DrawUse<T> rec_next = rec;
        if (q + 1 < h_passes)
        {
            const int lin = li + WPP;
            const bool actn = lin < h_shard_count;
            const int wn = half_base + h_shard_begin + (actn ? lin : 0);
            rec_next = DrawUse<T>(h_draws[wn - half_base]);
            load_slice<T, EPL>(h_pos + (size_t)wn * h_dims, i0, h_dims, vec_ok, actn, own_next);
            lp_next = h_logp[wn];
            nacc_next = h_n_accept[wn];
        }
        if (q > 0)
            load_slice<T, EPL>(h_pos + (size_t)(other_base + (int)rec.partner) * h_dims, i0, h_dims, vec_ok, active, par);
        if (q == 0) MCMCPP_STAMP(3);  // partner rows landed

        // StretchMove.h:105-108  proposal = sel + z*(cur - sel); padded cells stay +0
        T prop[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e)
        {
            const T d = own[e] - par[e];
            const T zd = rec.z * d;
            prop[e] = par[e] + zd;
        }
        const T lp_new = Calc::template eval<EPL, LPW>(ctx, a.calc_params, cregs, prop);
        if (q == 0) MCMCPP_STAMP(4);  // calculator done

        // StretchMove.h:112-113  accept iff lnU < (probScaling + newProb) - oldProb
        const T zs = rec.zs, ln_u = rec.ln_u;
        const T delta = zs + lp_new - lp_old;
        const bool accept = active && (ln_u < delta);
        if (active && sub == 0)
        {
            const T margin = dev_abs(ln_u - delta);
            const T scale = dev_abs(ln_u) + dev_abs(zs) + dev_abs(lp_new) + dev_abs(lp_old);
            if (margin <= a.tie_eps * scale) count_near_tie(a.diag);
        }
        if (accept)
        {
            // Walker::jumpToNewPointSwap (Walker/Walker.h:172-179)
            store_slice<T, EPL>(row, i0, h_dims, vec_ok, prop);  // (written through: no measurable difference here)
            if (sub == 0)
            {
                h_logp[w] = lp_new;
                asm volatile("" : "+v"(nacc_old));  // keep the +1 here: computed at the load it would drain every load
                h_n_accept[w] = nacc_old + 1u;
            }
        }
        if (save_slot >= 0 && active)
        {
            // Walker -> Chain::storeWalker (Chain/ChainBlock.h:125-131): cell = slot*W*D + walker*D + p
            T* crow = reinterpret_cast<T*>(run.chain) + ((size_t)save_slot * (size_t)(2 * h_n) + (size_t)w) * h_dims;
            if (accept)
                store_slice<T, EPL>(crow, i0, h_dims, vec_ok, prop);
            else
                store_slice<T, EPL>(crow, i0, h_dims, vec_ok, own);
        }
        accepted_here += (unsigned)__popcll(__ballot(accept && sub == 0));
#pragma unroll
        for (int e = 0; e < EPL; ++e) own[e] = own_next[e];
        lp_old = lp_next;
        nacc_old = nacc_next;
        rec = rec_next;
    }
    // Without a draw wavefront (many walkers per wavefront: large, bandwidth-bound ensembles, or very few
    // dimensions) the next draws are made here, when the update loop's registers are free: made in the shadow of the
    // gather they lift the kernel from 64 to 85 VGPRs and cost occupancy (7 % at 1 M walkers).  The record buffers
    // alternate, so the order against the update loop is free as far as correctness goes.
    if (!h_draw_wave) next_draws();
    MCMCPP_STAMP(5);  // all stores of this wavefront acknowledged
#ifdef MCMCPP_STAMPS
    if (a.stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
    {
        stamp_val[7] = __builtin_amdgcn_s_memrealtime();  // 100 MHz constant clock, with [6] taken at entry
        for (int k = 0; k < 8; ++k) a.stamps[k] = stamp_val[k];
    }
#endif
    // per-wavefront accepted count of this half-step; summed per ensemble step by accepted_reduce_kernel
    // (one plain store per wavefront: thousands of same-address atomics would serialise for ~12 ns each)
    if (a.partials != nullptr && run.accepted_per_step != nullptr && lane == 0)
        a.partials[(((size_t)chain * (size_t)a.partial_slots + (size_t)ctl.partial_slot) * 2 + (size_t)h_color) * (size_t)a.partial_waves + (size_t)wave] = accepted_here;
}

// Fills the draw records of one colour for the half-step whose base engine state is `base` (one thread per
// draw).  Run by the host for the first two half-steps after set_state / at the start of run(); from then on
// every half-step launch leaves the records of its colour's next update behind.
template <class T>
__global__ void __launch_bounds__(256) fill_draws_kernel(const HalfStepArgs<T> a, U128 base, U128 red_base, int with_partner2)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * a.shard_count) return;
    const int slot = t / 3, k = t - 3 * slot;
    const int i = a.shard_begin + slot;
    const bool direct = a.task_jump != nullptr;
    Affine128 j_a, j_b;
    if (direct)
        j_a = a.task_jump[3 * i + k];
    else
    {
        j_a = a.jump_hi[i >> 8];
        j_b = a.jump_lo[i & 255];
    }
    compute_draw<T>(a, base, j_a, j_b, direct, k, a.draws + ((size_t)a.draw_parity * 2 + (size_t)a.color) * (size_t)a.n + i,
                    with_partner2 != 0, red_base);
}

// The records of whole ensemble steps, made AHEAD of the launches that use them (full-step launches with
// HalfStepArgs::draw_wave == 2): blockIdx.y = 2 * j + colour for step j behind the one whose control record is `ctl`
// (step_jump[j]: 6 n j draws).  The draws do not depend on the walkers; made a graph replay's worth at a time by a launch
// of their own they cost a step launch 0.1 us less than made by its four extra wavefronts (C2: 5.65 -> 5.52 us per
// ensemble step, the fill launch's 0.18 us per step included -- DESIGN.md, experiment 3e).
// Workgroups of three wavefronts: wavefront k makes draw k (partner | z and (D-1) ln z | ln U) of 64 walkers, so that a
// wavefront runs ONE of the three kinds of arithmetic instead of all three in turn (draws 3 i + k in lane order, as the
// half-step priming kernel above has them, diverge three ways).
template <class T>
__global__ void __launch_bounds__(192) fill_draws_batch_kernel(const HalfStepArgs<T> a, const StepCtl* ctl, const Affine128* step_jump, DrawRec<T>* out)
{
    const int k = (int)threadIdx.x >> 6;
    const int slot = (int)blockIdx.x * 64 + ((int)threadIdx.x & 63);
    if (slot >= a.shard_count) return;
    const int j = (int)blockIdx.y >> 1, colour = (int)blockIdx.y & 1;
    const int i = a.shard_begin + slot;
    const bool direct = a.task_jump != nullptr;
    Affine128 j_a, j_b;
    if (direct)
        j_a = a.task_jump[3 * i + k];
    else
    {
        j_a = a.jump_hi[i >> 8];
        j_b = a.jump_lo[i & 255];
    }
    const U128 red_base = apply(step_jump[j], ctl->state);
    const U128 base = colour ? apply(a.half_jump, red_base) : red_base;
    compute_draw<T>(a, base, j_a, j_b, direct, k, out + ((size_t)j * 2 + (size_t)colour) * (size_t)a.n + i, colour != 0, red_base);
}

// ---------------------------------------------------------------------------------------------------------
// Matrix-core variant for calculators whose log-posterior contains a dense D x D product (the correlated
// Gaussian): a wavefront updates 16 walkers at once and evaluates  Y = X * P^T  ([16 walkers x 32] x [32 x 32])
// with v_mfma_f64_16x16x4_f64.  On gfx950 that instruction is, bit for bit, the fma chain in ascending k
// starting from the C operand (tools/mfma_probe.hip), i.e. exactly the host Calculator's
// `acc = fma(P[i][j], x[j], acc)` loop, so parity is unchanged.  Everything else (draw records one update
// ahead, two memory round trips, in-place accept, chain store, accepted partials) is as in the kernel above.
// Operand layout (measured): A lane l -> [m = l%16][k = l/16]; B lane l -> [k = l/16][n = l%16];
// C/D lane l, register r -> [m = 4r + l/16][n = l%16].  Walker slot (pass q, lane group g) is matrix row
// 4q + g, so its result comes back in register q of the very lanes that apply the accept.
// ---------------------------------------------------------------------------------------------------------
typedef double mfma_f64x4 __attribute__((ext_vector_type(4)));
typedef float mfma_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMcXS = 34;  // row stride of staged proposals (elements): even (whole 2-element pieces), = 2 mod 16 to spread LDS banks

// two consecutive elements of a row (what a lane of the matrix-core kernels holds of a walker: 16 lanes x 2 elements)
template <class T>
struct Vec2;
template <>
struct Vec2<double>
{
    typedef double2 type;
    __device__ __forceinline__ static double2 make(double a, double b) { return make_double2(a, b); }
};
template <>
struct Vec2<float>
{
    typedef float2 type;
    __device__ __forceinline__ static float2 make(float a, float b) { return make_float2(a, b); }
};

// Row of the 16-row tile that walker (pass q, lane group g) occupies: chosen so that its result comes back in register q
// of the lanes of group g, which apply the accept.  The two instructions lay their results out differently
// (tools/mfma_probe.hip, tools/mfma_f32_probe.hip): v_mfma_f64_16x16x4_f64 lane l, register r -> [m = 4r + l/16][n = l%16];
// v_mfma_f32_16x16x4_f32 lane l, register r -> [m = 4(l/16) + r][n = l%16].  Both are, bit for bit, the fma chain in
// ascending k starting from the C operand -- the host Calculator's loop.
template <class T>
__device__ __forceinline__ int mc_row(int q, int g)
{
    return sizeof(T) == 8 ? 4 * q + g : 4 * g + q;
}

// The wavefront's share of P^T, held in registers for the whole launch: for k-step ks, B[k = 4ks + lane/16][n] with
// tile column n of y0 = matrix column 2n and of y1 = matrix column 2n + 1, so that the product comes back in the
// very lanes (and element order) that hold the proposal: lane (grp, sub) holds x[2 sub], x[2 sub + 1].
template <class T>
struct McB
{
    typename Vec2<T>::type b[8];
};
template <class T>
__device__ __forceinline__ void mc_load_b(const T* sh_pt, int sub, int grp, McB<T>& B)
{
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) B.b[ks] = *reinterpret_cast<const typename Vec2<T>::type*>(sh_pt + (4 * ks + grp) * 32 + 2 * sub);
}

__device__ __forceinline__ mfma_f64x4 mc_mfma(double a, double b, mfma_f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ mfma_f32x4 mc_mfma(float a, float b, mfma_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// log-posteriors -1/2 x^T P x of the 4P proposals prop[q] (q < P; row mc_row(q, grp) of the tile), one MFMA tile:
// Y = X * P^T in 8 k-steps, k ascending (the host's fma order), then t_j = x_j * y_j and the canonical tree
// (in-lane pair, then the 16-lane butterfly).  `sx` is a wave-private staging area of 16 x kMcXS elements (4P x kMcXS
// for fp64, whose rows 4q + g stay below 4P).
template <int P, class T>
__device__ __forceinline__ void mc_eval(const McB<T>& B, T* sx, int sub, int grp, int dims, const T (&prop)[P][2], T (&lp)[P])
{
    typedef typename std::conditional<sizeof(T) == 8, mfma_f64x4, mfma_f32x4>::type Acc;
#pragma unroll
    for (int q = 0; q < P; ++q) *reinterpret_cast<typename Vec2<T>::type*>(sx + mc_row<T>(q, grp) * kMcXS + 2 * sub) = Vec2<T>::make(prop[q][0], prop[q][1]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // A[m = lane%16][k = 4ks + lane/16]; rows without a walker read as +0 (fp64: rows >= 4P; fp32: registers >= P of a group)
    const bool row_used = sizeof(T) == 8 ? sub < 4 * P : (sub & 3) < P;
    T xa[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) xa[ks] = row_used ? sx[sub * kMcXS + 4 * ks + grp] : (T)0;
    Acc y0 = {0, 0, 0, 0}, y1 = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
    {
        y0 = mc_mfma(xa[ks], B.b[ks].x, y0);
        y1 = mc_mfma(xa[ks], B.b[ks].y, y1);
    }
    const bool in = 2 * sub < dims;  // D is even: both of the lane's elements or neither
#pragma unroll
    for (int r = 0; r < P; ++r)
    {
        const T t0 = prop[r][0] * y0[r], t1 = prop[r][1] * y1[r];
        T t = in ? t0 + t1 : (T)0;
        t = t + dpp_move<0xB1>(t);
        t = t + dpp_move<0x4E>(t);
        t = t + dpp_move<0x141>(t);
        t = t + dpp_move<0x140>(t);
        lp[r] = (T)-0.5 * t;
    }
}

// P = passes per wavefront (2 or 4): the wavefront's 4 P walkers are rows 0 .. 4P-1 of the 16-row tile.
// LATE (P = 4, no draw wavefront): the walkers' next draws are made BEHIND the accept instead of in the shadow of the partner
// gather, which takes the jump entries and the 128-bit arithmetic out of the stretch where everything else is live -- 118
// registers instead of 152 in fp64, four wavefronts per SIMD instead of three.  A launch is a whole number of rounds of the
// chip's wavefront slots and a round is one latency chain (about 5 us) whatever it carries: 65 536 updates are 4 096
// wavefronts of 16 walkers -- one round at four per SIMD, one and a third at three (measured as two: 10.7 against 12.9 us);
// below 49 152 updates per launch the shorter wavefront wins (6.85 against 7.29 us at 32 768): the host picks by the size
// of the launch (profiles/r03_mc_occupancy.txt, r03_mc_threshold.txt).
template <class T, class Calc, int EPL, int LPW, int P, bool DW, bool MC = false, bool LATE = false>
__global__ void __launch_bounds__(64 * (kWavesPerBlock + (DW ? 1 : 0)), (LATE && sizeof(T) == 8) ? 4 : 1)
stretch_half_step_mfma_kernel(DrawRec<T>* hot_draws, T* hot_pos, T* hot_logp, uint32_t* hot_n_accept, int hot_n, uint32_t hot_bits,
                              int hot_shard_begin, int hot_shard_count, const StepCtl* hot_ctl_in, const T* hot_matrix, const HalfStepArgs<T> rest)
{
    static_assert(EPL == 2 && LPW == 16, "matrix-core path: 16 lanes x 2 elements per walker, 16 < D <= 32");
    static_assert(P == 2 || P == 4, "two or four passes");
    static_assert(!LATE || (P == 4 && !DW), "late draws: the 16-walker wavefront without a draw wavefront");
    constexpr int NW = 4 * P;   // walkers per wavefront
    constexpr int XS = kMcXS;
    constexpr int kStageRows = sizeof(T) == 8 ? NW : 16;  // rows of the tile that hold walkers (mc_row)
    // LDS: proposal rows, kStageRows x XS per wavefront, then 32 x 32 elements for P^T (see kMatrixViaLds below).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* sh_x = reinterpret_cast<T*>(smem) + (threadIdx.x >> 6) * (kStageRows * XS);

    const HalfStepArgs<T>& a = rest;
#ifdef MCMCPP_STAMPS
    unsigned long long stamp_val[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    stamp_val[6] = __builtin_amdgcn_s_memrealtime();
#endif
    MCMCPP_STAMP(0);
    MCMCPP_STAMP_BLOCK(0);
    // chain blockIdx.y of (hot_bits >> 28) + 1 (ChainGeometry), as in the kernel above
        // (MC: built for several chains per launch; the single-ensemble instantiation carries none of it -- measured 2 % of a
    //  65 536-walker launch otherwise)
    const int chain = MC ? (int)blockIdx.y : 0, chains = MC ? (int)(hot_bits >> 28) + 1 : 1;
    const void* const draws_chain0 = hot_draws;
    if (MC && chain != 0)
    {
        hot_draws += (size_t)chain * 4 * (size_t)hot_n;
        hot_pos += (size_t)chain * 2 * (size_t)hot_n * (size_t)(hot_bits & 0xFFFu);
        hot_logp = reinterpret_cast<T*>(reinterpret_cast<char*>(hot_logp) + (size_t)chain * logp_chain_stride_bytes<T>(hot_n));
        hot_n_accept = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(hot_n_accept) + (size_t)chain * logp_chain_stride_bytes<T>(hot_n));
    }
    const int h_color = (int)((hot_bits >> 20) & 1u);
    const int h_parity = (int)((hot_bits >> 24) & 1u);
    const DrawRec<T>* const h_draws = hot_draws + ((size_t)h_parity * 2 + (size_t)h_color) * (size_t)hot_n;
    DrawRec<T>* const h_draws_next = hot_draws + ((size_t)(1 - h_parity) * 2 + (size_t)h_color) * (size_t)hot_n;
    T* const h_pos = hot_pos;
    T* const h_logp = hot_logp;
    uint32_t* const h_n_accept = hot_n_accept;
    const int h_n = hot_n;
    const int h_dims = (int)(hot_bits & 0xFFFu);
    const int h_use_ctl_save = (int)((hot_bits >> 23) & 1u);
    const int h_shard_begin = hot_shard_begin;
    const int h_shard_count = hot_shard_count;

    const int lane = threadIdx.x & 63;
    if constexpr (DW)
    {
        if ((threadIdx.x >> 6) == kWavesPerBlock)
        {
            // the workgroup's extra wavefront: next draws of every walker this workgroup updates
            draw_wave_body<T, 2>(a, jump_tables_behind(draws_chain0, hot_n, ((hot_bits >> 27) & 1u) != 0, chains), hot_ctl_in,
                                 false /* no workgroup barrier in this kernel */, h_draws_next, h_draws_next, 1, h_shard_begin,
                                 h_shard_count, blockIdx.x * kWavesPerBlock * NW, kWavesPerBlock * NW, lane, false, nullptr, -1, chain);
            return;
        }
    }
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int first = wave * NW;
    const bool wave_active = first < h_shard_count;
    const int half_base = h_color ? h_n : 0;
    const int other_base = h_color ? 0 : h_n;
    const int sub = lane & 15, grp = lane >> 4, i0 = sub * 2;
    // (the host only selects this kernel for even D: rows are whole 2-element pieces, every access is branch-free)
    typedef typename Vec2<T>::type V2;
    const bool col_ok = i0 < h_dims;
    const int i0c = col_ok ? i0 : 0;
    const int last_li = h_shard_count - 1;

    // ---- first round trip: draw records, own rows, log-posteriors, counters of all passes; the matrix ------------
    // Addresses are the (uniform) array base plus a 32-bit byte offset per lane -- one register instead of two for each of
    // the 5 P addresses a wavefront keeps until its stores (the host selects this kernel only for arrays below 4 GiB).
    const char* const pos_b = reinterpret_cast<const char*>(h_pos);
    const char* const logp_b = reinterpret_cast<const char*>(h_logp);
    const char* const nacc_b = reinterpret_cast<const char*>(h_n_accept);
    const char* const draws_b = reinterpret_cast<const char*>(h_draws);
    DrawRec<T> rec[P];
    T own[P][2];
    T lp_old[P];
    bool active[P];
    uint32_t w[P];
#pragma unroll
    for (int q = 0; q < P; ++q)
    {
        const int li = first + 4 * q + grp;
        active[q] = wave_active && li < h_shard_count;
        const uint32_t in_half = (uint32_t)(h_shard_begin + (active[q] ? li : 0));
        w[q] = (uint32_t)half_base + in_half;
        rec[q] = *reinterpret_cast<const DrawRec<T>*>(draws_b + in_half * (uint32_t)sizeof(DrawRec<T>));
        {
            const V2 v = *reinterpret_cast<const V2*>(pos_b + (w[q] * (uint32_t)h_dims + (uint32_t)i0c) * (uint32_t)sizeof(T));
            own[q][0] = (active[q] && col_ok) ? v.x : (T)0;
            own[q][1] = (active[q] && col_ok) ? v.y : (T)0;
        }
        lp_old[q] = *reinterpret_cast<const T*>(logp_b + w[q] * (uint32_t)sizeof(T));
    }

    const StepCtl* ctl_mine = hot_ctl_in;
    const RunInfo* run_mine = a.run;
    if (MC && chain != 0)  // (the chain's own records, ChainGeometry)
    {
        ctl_mine = reinterpret_cast<const StepCtl*>(reinterpret_cast<const char*>(ctl_mine) + (size_t)chain * kCtlChainStride);
        run_mine = reinterpret_cast<const RunInfo*>(reinterpret_cast<const char*>(run_mine) + (size_t)chain * kCtlChainStride);
    }
    const StepCtl ctl = *ctl_mine;
    const RunInfo run = *run_mine;
    // The walkers' next draws, in the shadow of the partner gather or (LATE) behind the accept.  A wavefront of 16 walkers used
    // to make its own 48 (lane l: draw l % 3 of walker l / 3) -- three kinds of arithmetic (partner index | z and (D-1) ln z |
    // ln U) in one wavefront, run one after the other at a third of the lanes each.  The draws do not depend on the walkers,
    // so the four wavefronts of a workgroup divide the workgroup's 64 walkers' draws BY KIND instead: three of them make one
    // kind each for all 64 walkers (64 lanes, one path), the fourth makes none; the kind rotates with the workgroup so that no
    // SIMD collects all the logarithms.  (Issue-bound kernel, profiles/r03_post_sq_breakdown_secondary.txt: two of three paths
    // fewer per wavefront.)  Wavefronts without walkers of their own still make their share.
    constexpr bool kDrawsByKind = !DW && NW * kWavesPerBlock == 64;
    const int draw_kind = kDrawsByKind ? (int)(((threadIdx.x >> 6) + blockIdx.x) & 3u) : 0;
    const int draw_first = kDrawsByKind ? (int)blockIdx.x * kWavesPerBlock * NW : first;
    const int slot_a = kDrawsByKind ? lane : lane / 3, k_a = kDrawsByKind ? draw_kind : lane - 3 * slot_a;
    const bool draws_here = kDrawsByKind ? (draw_kind < 3 && draw_first + slot_a < h_shard_count) : (wave_active && lane < 3 * NW && first + slot_a < h_shard_count);
    const bool direct_jump = a.task_jump != nullptr;
    Affine128 j_a, j_b;
    // (macros, not lambdas: a closure that captures the jump entries by reference keeps them in scratch memory)
#define MCMCPP_LOAD_DRAW_JUMPS()                                                  \
    do                                                                            \
    {                                                                             \
        const int i_a = h_shard_begin + max(min(draw_first + slot_a, last_li), 0); \
        if (direct_jump)                                                          \
            j_a = a.task_jump[3 * i_a + min(k_a, 2)]; /* (the wavefront without draws: a valid entry) */ \
        else                                                                      \
        {                                                                         \
            j_a = a.jump_hi[i_a >> 8];                                            \
            j_b = a.jump_lo[i_a & 255];                                           \
        }                                                                         \
    } while (0)
#define MCMCPP_MAKE_NEXT_DRAWS()                                                                                                                     \
    do                                                                                                                                               \
    {                                                                                                                                                \
        if (draws_here) compute_draw<T>(a, ctl.state2, j_a, j_b, direct_jump, k_a, h_draws_next + h_shard_begin + draw_first + slot_a);              \
    } while (0)
    if constexpr (!DW && !LATE) MCMCPP_LOAD_DRAW_JUMPS();

    // ---- second round trip: the partner rows of all passes ----------------------------------------------------------
    T par[P][2];
#pragma unroll
    for (int q = 0; q < P; ++q)
    {
        const uint32_t pw = (uint32_t)other_base + (active[q] ? rec[q].partner : 0u);
        const V2 v = *reinterpret_cast<const V2*>(pos_b + (pw * (uint32_t)h_dims + (uint32_t)i0c) * (uint32_t)sizeof(T));
        par[q][0] = (active[q] && col_ok) ? v.x : (T)0;
        par[q][1] = (active[q] && col_ok) ? v.y : (T)0;
    }
    // the wavefront's share of P^T: 8 x 16 bytes per lane, the same 8 KiB for every wavefront (L2 hits), issued behind
    // the partner gather so that it does not compete with the records the gather waits for
    asm volatile("" ::: "memory");
    McB<T> matB;
    // P^T (zero-padded to 32 x 32 by the host, behind the sixteenth preloaded pointer).  With a draw wavefront in the
    // workgroup every update wavefront fetches its share into registers (8 x 16 bytes per lane, the same 8 KiB for every
    // wavefront: L2 hits).  Without one, the workgroup's four wavefronts fetch a quarter each, issued behind the partner
    // gather, and share it through LDS behind a barrier that comes after the work done in the gather's shadow: a quarter
    // of the L2 reads (8 KiB per 16 walkers was two thirds of the rows' own traffic) and 24 registers fewer where the
    // pressure is highest (profiles/r03_lean_probe.txt).
    constexpr bool kMatrixViaLds = !DW;
    T* const sh_pt = reinterpret_cast<T*>(smem) + kWavesPerBlock * (kStageRows * XS);
    typedef T V4 __attribute__((ext_vector_type(4)));
    V4 mpiece;
    if constexpr (kMatrixViaLds)
        mpiece = *reinterpret_cast<const V4*>(hot_matrix + 4 * threadIdx.x);
    else
        mc_load_b(hot_matrix, sub, grp, matB);
    MCMCPP_STAMP(1);  // records landed, partner gather issued

    // ---- in its shadow: hand-over to the next launch, the walkers' next draws ------------------------------------------
    // one lane of the grid hands the stream and the counters to the next launch (in the shadow of its gather wait;
    // in the extra wavefront it would lengthen the last wavefront to finish: measured)
    if (blockIdx.x == 0 && threadIdx.x == 0)
        hand_over<T>(a, ctl, run, h_color, reinterpret_cast<StepCtl*>(reinterpret_cast<char*>(a.ctl_out) + (size_t)chain * kCtlChainStride));
    long long save_slot = -1;
    if (a.direct_save_slot >= 0)
        save_slot = a.direct_save_slot;
    else if (h_use_ctl_save && run.chain != nullptr && ctl.save_phase + 1u == (uint32_t)run.interval)
        save_slot = run.chain_slot_base + ctl.chain_slot;
    if constexpr (!DW && !LATE) MCMCPP_MAKE_NEXT_DRAWS();
    MCMCPP_STAMP(2);  // next draws done

    // ---- proposals (StretchMove.h:105-108) -----------------------------------------------------------------------
    T prop[P][2];
#pragma unroll
    for (int q = 0; q < P; ++q)
    {
#pragma unroll
        for (int e = 0; e < 2; ++e)
        {
            const T d = own[q][e] - par[q][e];
            const T zd = rec[q].z * d;
            prop[q][e] = par[q][e] + zd;
        }
    }
    MCMCPP_STAMP(3);  // partner rows landed

    // ---- Y = X * P^T on the matrix cores, products and the canonical tree ------------------------------------------
    T lp_new[P];
    if constexpr (kMatrixViaLds)
    {
        *reinterpret_cast<V4*>(sh_pt + 4 * threadIdx.x) = mpiece;
        __syncthreads();  // (every wavefront of the workgroup comes here, those without walkers too)
        mc_load_b(sh_pt, sub, grp, matB);
    }
    if (!wave_active)
    {
        if constexpr (LATE && kDrawsByKind)
        {
            MCMCPP_LOAD_DRAW_JUMPS();
            MCMCPP_MAKE_NEXT_DRAWS();
        }
        return;
    }
    mc_eval<P>(matB, sh_x, sub, grp, h_dims, prop, lp_new);
    MCMCPP_STAMP(4);  // calculator done
    if constexpr (LATE) MCMCPP_LOAD_DRAW_JUMPS();

    // ---- Metropolis accept in place, chain store, counters ------------------------------------------------------------
    unsigned accepted_here = 0;
#pragma unroll
    for (int q = 0; q < P; ++q)
    {
        const T zs = rec[q].zs, ln_u = rec[q].ln_u;
        const T delta = zs + lp_new[q] - lp_old[q];
        const bool accept = active[q] && (ln_u < delta);
        if (active[q] && sub == 0)
        {
            const T margin = dev_abs(ln_u - delta);
            const T scale = dev_abs(ln_u) + dev_abs(zs) + dev_abs(lp_new[q]) + dev_abs(lp_old[q]);
            if (margin <= a.tie_eps * scale) count_near_tie(a.diag);
        }
        if (accept)
        {
            if (col_ok)
                *reinterpret_cast<V2*>(const_cast<char*>(pos_b) + (w[q] * (uint32_t)h_dims + (uint32_t)i0) * (uint32_t)sizeof(T)) = Vec2<T>::make(prop[q][0], prop[q][1]);
            if (sub == 0)
            {
                *reinterpret_cast<T*>(const_cast<char*>(logp_b) + w[q] * (uint32_t)sizeof(T)) = lp_new[q];
                // (an add that returns nothing instead of a load in the first round trip and a store here: the counter is
                //  this lane group's alone, and a register per pass is free while everything else is live)
                __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(const_cast<char*>(nacc_b) + w[q] * 4u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (save_slot >= 0 && active[q])
        {
            T* crow = reinterpret_cast<T*>(run.chain) + ((size_t)save_slot * (size_t)(2 * h_n) + (size_t)w[q]) * h_dims;
            if (col_ok)
                // (a stored step is the exception: the row as it stands, from memory -- this lane's own store just above
                //  included -- instead of registers kept for it across the tile)
                *reinterpret_cast<V2*>(crow + i0) = *reinterpret_cast<const V2*>(pos_b + (w[q] * (uint32_t)h_dims + (uint32_t)i0) * (uint32_t)sizeof(T));
        }
        accepted_here += (unsigned)__popcll(__ballot(accept && sub == 0));
    }
    if constexpr (LATE) MCMCPP_MAKE_NEXT_DRAWS();
#undef MCMCPP_LOAD_DRAW_JUMPS
#undef MCMCPP_MAKE_NEXT_DRAWS
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
        a.partials[(((size_t)chain * (size_t)a.partial_slots + (size_t)ctl.partial_slot) * 2 + (size_t)h_color) * (size_t)a.partial_waves + (size_t)wave] = accepted_here;
}

// Before the first full-step launch of a run(): no row of the second position buffer can be trusted (set_state and
// the half-step kernels only ever touch the first), so every walker is marked "moved" (full_step_kernel.hpp).
#ifdef MCMCPP_DEFINE_REDUCE_KERNEL
__global__ void __launch_bounds__(256) mark_rows_moved_kernel(uint32_t* n_accept, int walkers, uint32_t bit)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < walkers) n_accept[w] |= bit;
}
#endif

// Sums the per-wavefront accepted counts of the last `count` ensemble steps into RunInfo.accepted_per_step.
// Runs once after every graph replay (one workgroup per step); `ctl_after` is the control record the
// last black launch left behind, so step_in_run is the number of steps finished in this run.
#ifdef MCMCPP_DEFINE_REDUCE_KERNEL  // one definition, in mcmcpp_hip.hip
__global__ void __launch_bounds__(256)
accepted_reduce_kernel(const uint32_t* partials, int partial_slots, int partial_waves, int count, const StepCtl* ctl_after,
                       const RunInfo* run_ptr, StepCtl* ctl_keep)
{
    __shared__ unsigned sums[4];
    // (one chain, records made a replay ahead: the control record at the head of the next replay, out of the step launches' way)
    if (ctl_keep != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *ctl_keep = *ctl_after;
    // (blockIdx.y: the chain, see ChainGeometry)
    partials += (size_t)blockIdx.y * (size_t)partial_slots * 2 * (size_t)partial_waves;
    ctl_after = reinterpret_cast<const StepCtl*>(reinterpret_cast<const char*>(ctl_after) + (size_t)blockIdx.y * kCtlChainStride);
    run_ptr = reinterpret_cast<const RunInfo*>(reinterpret_cast<const char*>(run_ptr) + (size_t)blockIdx.y * kCtlChainStride);
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
    typename Calc::Prefetch calc_pf;
    Calc::block_prefetch(calc_pf, calc_params, dims, vec_ok != 0, (int)threadIdx.x, 64 * kWavesPerBlock);
    Calc::block_commit(calc_pf, sh_block, calc_params, dims, vec_ok != 0, (int)threadIdx.x, 64 * kWavesPerBlock);
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

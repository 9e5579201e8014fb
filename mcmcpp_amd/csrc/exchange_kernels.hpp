// exchange_kernels.hpp -- a split ensemble's ranks exchange ONLY THE ROWS THAT MOVED (BASELINE config 5).
//
// The reference's workers share one address space, so "the other half as it stands" costs them nothing
// (MCMCpp/Threading/RedBlkCtrlerSpinLock.h:240-322: two barriers per step, no data motion).  Ranks on different GPUs
// keep replicas, and an all-gather of every rank's slice moves rows nobody changed: at C5's acceptance rate one walker in
// six moves per step.  Instead, behind every step (or half-step) launch a rank
//   packs   the walkers of its slice whose accepted counter changed since the last exchange -- global index,
//           log-posterior, row -- into a block of `cap` slots (exchange_pack_kernel),
//   gathers the ranks' blocks (one ncclAllGather of G equal blocks: the only collective of the step),
//   scatters the other ranks' rows into its replica (exchange_scatter_kernel) -- into BOTH position buffers when the
//           full-step kernels ping-pong between two, so that a remote walker that stays put next step is already where
//           that step's output buffer expects it.
// A block holds its slot bound and the number of walkers that moved; more moved walkers than slots is an OVERFLOW:
// every rank sees it in the gathered headers, raises a sticky flag, and the host -- which looks at the flag once per
// chunk of steps -- rolls the ensemble back to the chunk's snapshot and repeats the chunk with blocks that hold a whole
// slice (mcmcpp_hip.hip: run_split).  A run therefore never continues on a replica that missed a row.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcmcpp
{
constexpr uint32_t kAcceptCountMask = 0x7fffffffu;  // (the top bit of an accepted counter is the full-step kernels' "moved" mark)

// [header 16 B][idx: cap x u32][logp: cap x T][rows: cap x D x T], every piece 16-byte aligned
struct XBlockHeader
{
    uint32_t count;  // walkers of the slice that moved (may exceed cap: overflow)
    uint32_t cap;
    uint32_t pad[2];
};
__host__ __device__ inline size_t xblock_align16(size_t b) { return (b + 15) & ~(size_t)15; }
__host__ __device__ inline size_t xblock_idx_offset() { return sizeof(XBlockHeader); }
template <class T>
__host__ __device__ inline size_t xblock_logp_offset(uint32_t cap) { return xblock_idx_offset() + xblock_align16((size_t)cap * sizeof(uint32_t)); }
template <class T>
__host__ __device__ inline size_t xblock_rows_offset(uint32_t cap) { return xblock_logp_offset<T>(cap) + xblock_align16((size_t)cap * sizeof(T)); }
template <class T>
__host__ __device__ inline size_t xblock_bytes(uint32_t cap, int dims) { return xblock_rows_offset<T>(cap) + xblock_align16((size_t)cap * (size_t)dims * sizeof(T)); }

// what the scatter kernels leave for the host (read once per chunk of steps)
struct XStats
{
    uint32_t overflow;   // sticky: some rank's block of some exchange had more moved walkers than slots
    uint32_t max_count;  // largest number of moved walkers any rank packed in one exchange
};

// own slice: seen[w] <- accepted counter of w (start of a run, after a roll-back)
__global__ void __launch_bounds__(256) exchange_sync_seen_kernel(const uint32_t* n_accept, uint32_t* seen, int n, int shard_begin, int shard_count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * shard_count) return;
    const int w = (i / shard_count) * n + shard_begin + i % shard_count;
    seen[w] = n_accept[w] & kAcceptCountMask;
}

// A wavefront looks at kPackWalkersPerWave walkers of the slice (colours [color0, color0 + colors)), one per lane of its
// first lanes, and copies the rows of those that moved with all 64 lanes -- few walkers per wavefront on purpose: the rows
// of a wavefront go one after the other (each a dependent load -> store), so the launch is as long as the busiest
// wavefront's list.  Slots are reserved per WORKGROUP of 16 wavefronts (wave counts through LDS, one atomic on the block's
// count, a prefix over the wavefronts): one atomic per wavefront on the same address costs ~12 ns each -- 1 024 of them
// made this launch 17 us at C5's slice size, 64 make it 3.  The block's count must be zero on entry (the scatter kernel
// of the previous exchange, or the host at the start of a chunk, sees to that).
constexpr int kPackWalkersPerWave = 16;
constexpr int kPackWavesPerBlock = 16;
template <class T>
__global__ void __launch_bounds__(64 * kPackWavesPerBlock) exchange_pack_kernel(const T* pos, const T* logp, const uint32_t* n_accept, uint32_t* seen, char* block,
                                                                                   uint32_t cap, int n, int dims, int shard_begin, int shard_count, int color0, int colors)
{
    __shared__ uint32_t wave_count[kPackWavesPerBlock];
    __shared__ uint32_t block_base;
    __shared__ int moved_walker[kPackWavesPerBlock][kPackWalkersPerWave];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int wave = (int)blockIdx.x * kPackWavesPerBlock + wib;
    const int i = wave * kPackWalkersPerWave + lane;
    const bool in_range = lane < kPackWalkersPerWave && i < colors * shard_count;
    const int w = in_range ? (color0 + i / shard_count) * n + shard_begin + i % shard_count : 0;
    bool moved = false;
    if (in_range)
    {
        const uint32_t now = n_accept[w] & kAcceptCountMask;
        moved = now != seen[w];
        if (moved) seen[w] = now;
    }
    const unsigned long long ballot = __ballot(moved);
    if (lane == 0) wave_count[wib] = (uint32_t)__popcll(ballot);
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t total = 0;
        for (int k = 0; k < kPackWavesPerBlock; ++k) total += wave_count[k];
        block_base = total ? atomicAdd(&reinterpret_cast<XBlockHeader*>(block)->count, total) : 0u;
    }
    __syncthreads();
    if (ballot == 0) return;
    uint32_t base = block_base;
    for (int k = 0; k < wib; ++k) base += wave_count[k];
    uint32_t* idx = reinterpret_cast<uint32_t*>(block + xblock_idx_offset());
    T* blogp = reinterpret_cast<T*>(block + xblock_logp_offset<T>(cap));
    T* brows = reinterpret_cast<T*>(block + xblock_rows_offset<T>(cap));
    const uint32_t my_slot = base + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
    if (moved && my_slot < cap)
    {
        idx[my_slot] = (uint32_t)w;
        blogp[my_slot] = logp[w];
    }
    // the rows: the wavefront's moved walkers listed through LDS (rank inside the wavefront -> walker), then copied
    // 64 / pieces rows at a time (a 512-byte row is 32 pieces of 16 bytes: two rows per round), all loads of a round in flight
    const uint32_t mine = (uint32_t)__popcll(ballot);
    if (moved) moved_walker[wib][my_slot - base] = w;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool vec = ((size_t)dims * sizeof(T)) % 16 == 0;
    const int pieces = vec ? (int)((size_t)dims * sizeof(T) / 16) : dims;
    int lpr = 1;  // lanes per row: a power of two, at most 64
    while (lpr < pieces && lpr < 64) lpr <<= 1;
    const int rows_per_round = 64 / lpr, sub = lane % lpr, rr = lane / lpr;
    for (uint32_t r0 = 0; r0 < mine; r0 += (uint32_t)rows_per_round)
    {
        const uint32_t r = r0 + (uint32_t)rr;
        if (r >= mine || base + r >= cap) continue;
        const int sw = moved_walker[wib][r];
        if (vec)
        {
            const uint4* sp = reinterpret_cast<const uint4*>(pos + (size_t)sw * dims);
            uint4* dp = reinterpret_cast<uint4*>(brows + (size_t)(base + r) * dims);
            for (int k = sub; k < pieces; k += lpr) dp[k] = sp[k];
        }
        else
            for (int k = sub; k < dims; k += lpr) brows[(size_t)(base + r) * dims + k] = pos[(size_t)sw * dims + k];
    }
}

// blocks: [ranks][block_bytes] as the all-gather left them.  Row `slot` of peer p (blockIdx.y counts the peers, skipping
// this rank) goes to pos_a (and pos_b, logp_b unless null).  One workgroup row of 256 threads handles 256 / lanes_per_row
// slots.  Thread 0 of the grid reads every header for the statistics and clears this rank's own count for the next pack.
template <class T>
__global__ void __launch_bounds__(256) exchange_scatter_kernel(char* blocks, size_t block_bytes, uint32_t cap, int ranks, int rank, int dims, T* pos_a, T* pos_b,
                                                               T* logp_a, T* logp_b, XStats* stats)
{
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    {
        uint32_t worst = 0;
        for (int p = 0; p < ranks; ++p)
        {
            const uint32_t c = reinterpret_cast<const XBlockHeader*>(blocks + block_bytes * (size_t)p)->count;
            worst = c > worst ? c : worst;
        }
        if (worst > cap) atomicOr(&stats->overflow, 1u);
        atomicMax(&stats->max_count, worst);
        reinterpret_cast<XBlockHeader*>(blocks + block_bytes * (size_t)rank)->count = 0;
    }
    const int peer = (int)blockIdx.y + ((int)blockIdx.y >= rank ? 1 : 0);
    const char* block = blocks + block_bytes * (size_t)peer;
    uint32_t count = reinterpret_cast<const XBlockHeader*>(block)->count;
    if (count > cap) count = cap;
    const bool vec = ((size_t)dims * sizeof(T)) % 16 == 0;
    const int pieces = vec ? (int)((size_t)dims * sizeof(T) / 16) : dims;
    int lpr = 1;  // lanes per row: a power of two, at most 64
    while (lpr < pieces && lpr < 64) lpr <<= 1;
    const int rows_per_block = 256 / lpr;
    const uint32_t slot = blockIdx.x * (uint32_t)rows_per_block + threadIdx.x / (uint32_t)lpr;
    if (slot >= count) return;
    const int sub = threadIdx.x % lpr;
    const uint32_t* idx = reinterpret_cast<const uint32_t*>(block + xblock_idx_offset());
    const T* blogp = reinterpret_cast<const T*>(block + xblock_logp_offset<T>(cap));
    const T* brows = reinterpret_cast<const T*>(block + xblock_rows_offset<T>(cap));
    const size_t w = idx[slot];
    if (sub == 0)
    {
        const T lp = blogp[slot];
        logp_a[w] = lp;
        if (logp_b) logp_b[w] = lp;
    }
    if (vec)
    {
        const uint4* s = reinterpret_cast<const uint4*>(brows + (size_t)slot * dims);
        uint4* da = reinterpret_cast<uint4*>(pos_a + w * dims);
        uint4* db = pos_b ? reinterpret_cast<uint4*>(pos_b + w * dims) : nullptr;
        for (int k = sub; k < pieces; k += lpr)
        {
            const uint4 v = s[k];
            da[k] = v;
            if (db) db[k] = v;
        }
    }
    else
        for (int k = sub; k < dims; k += lpr)
        {
            const T v = brows[(size_t)slot * dims + k];
            pos_a[w * dims + k] = v;
            if (pos_b) pos_b[w * dims + k] = v;
        }
}
}  // namespace mcmcpp

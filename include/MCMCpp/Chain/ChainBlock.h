/*
 * Chain/ChainBlock.h -- one contiguous slab of stored ensemble steps.
 *
 * Output format kept from the reference (/root/reference/MCMCpp/Chain/ChainBlock.h:125-131): positions
 * only, cell = step*W*D + walker*D + param, every step holds all W walkers.  Unlike the reference's fixed
 * 10 000-step blocks indexed with 32-bit int (ChainBlock.h:31,115-123 -- the size product overflows at
 * 16 384 x 32 and above), a block here is sized in bytes and indexed with 64-bit integers.
 */
#ifndef MCMCPP_CHAIN_CHAINBLOCK_H
#define MCMCPP_CHAIN_CHAINBLOCK_H

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace MCMC
{
namespace Chain
{
namespace Detail
{
/// Default slab size; the number of steps per block follows from it and the ensemble size.
static const unsigned long long DefaultBlockBytes = 256ULL << 20;

/// Where a block's memory comes from.  The default is the heap (64-byte aligned like the reference's autoAlignedAlloc,
/// Utility/Misc.h:77-102); the MI355X samplers hand out pinned host memory (mcmcpp_hip_host_alloc) so that the step
/// launches write stored steps straight into the block.  A failed `obtain` falls back to the heap.
struct BlockMemory
{
    void* (*obtain)(unsigned long long bytes);
    void (*release)(void*);
    BlockMemory() : obtain(nullptr), release(nullptr) {}
    BlockMemory(void* (*get)(unsigned long long), void (*put)(void*)) : obtain(get), release(put) {}
};
}

template <class ParamType>
class ChainBlock
{
public:
    /// touchPages: write to every page of a heap block now (the Chain does this on the thread that obtains blocks ahead of
    /// their use, so that the first-touch page faults -- about 45 ms per 256 MiB -- are not paid by whoever fills the block)
    ChainBlock(std::int64_t stepsInBlock, std::int64_t cellsInStep, const Detail::BlockMemory& mem = Detail::BlockMemory(), bool touchPages = false)
        : capacitySteps(stepsInBlock), cellsPerStep(cellsInStep), usedSteps(0), cells(nullptr), releaseFn(nullptr)
    {
        const std::size_t bytes = static_cast<std::size_t>(capacitySteps) * static_cast<std::size_t>(cellsPerStep) * sizeof(ParamType);
        void* p = nullptr;
        if (mem.obtain && mem.release) p = mem.obtain(bytes ? bytes : 64);
        if (p)
            releaseFn = mem.release;
        else if (posix_memalign(&p, 64, bytes ? bytes : 64) != 0)
            p = nullptr;
        else if (touchPages)
            for (std::size_t off = 0; off < bytes; off += 4096) static_cast<volatile char*>(p)[off] = 0;
        cells = static_cast<ParamType*>(p);
    }
    ~ChainBlock()
    {
        if (releaseFn)
            releaseFn(cells);
        else
            std::free(cells);
    }
    ChainBlock(const ChainBlock&) = delete;
    ChainBlock& operator=(const ChainBlock&) = delete;

    bool valid() const { return cells != nullptr; }
    bool full() const { return usedSteps >= capacitySteps; }
    std::int64_t capacity() const { return capacitySteps; }
    std::int64_t used() const { return usedSteps; }
    void setUsed(std::int64_t n) { usedSteps = n; }
    ParamType* step(std::int64_t k) { return cells + static_cast<std::size_t>(k) * static_cast<std::size_t>(cellsPerStep); }
    const ParamType* step(std::int64_t k) const { return cells + static_cast<std::size_t>(k) * static_cast<std::size_t>(cellsPerStep); }

private:
    std::int64_t capacitySteps;
    std::int64_t cellsPerStep;
    std::int64_t usedSteps;
    ParamType* cells;
    void (*releaseFn)(void*);
};

}  // namespace Chain
}  // namespace MCMC
#endif  // MCMCPP_CHAIN_CHAINBLOCK_H

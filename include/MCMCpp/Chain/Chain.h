/*
 * Chain/Chain.h -- the sampler's output: every stored ensemble step, walker-major.
 *
 * Same role and cell layout as the reference's Chain (/root/reference/MCMCpp/Chain/Chain.h:61-151): the
 * sampler stores one full step at a time, users read it back through ChainStepIterator (W*D contiguous
 * values per step) or ChainPsetIterator (D values per walker).  Differences, all deliberate:
 *   - 64-bit step and cell indices and byte-sized blocks (the reference overflows 32-bit int at the
 *     BASELINE ensemble sizes, ChainBlock.h:31,116-128);
 *   - whole steps arrive from the GPU, so besides the reference's per-walker storeWalker() there is a bulk
 *     path (stepsContiguousFrom / commitSteps) that lets the device-to-host copy land in place;
 *   - iterators address steps by global index instead of walking a linked list, which removes the
 *     off-by-one defects of the reference's ChainPsetIterator (ChainPsetIterator.h:131-142,154).
 */
#ifndef MCMCPP_CHAIN_CHAIN_H
#define MCMCPP_CHAIN_CHAIN_H

#include <cstdint>
#include <cstring>
#include <future>
#include <vector>

#include "ChainBlock.h"
#include "ChainPsetIterator.h"
#include "ChainStepIterator.h"

namespace MCMC
{
namespace Chain
{

/// Result of advancing the chain by one stored step (names as in the reference, Chain.h:34-38).
enum class IncrementStatus : char
{
    NormalIncrement,  ///< stayed inside the current block
    NewBlock,         ///< the next step will land in a fresh block
    EndOfChain        ///< the byte budget is exhausted: nothing more can be stored
};

template <class ParamType>
class Chain
{
public:
    typedef ChainPsetIterator<ParamType> PsetIterator;
    typedef ChainStepIterator<ParamType> StepIterator;

    /// maxSize: byte budget for stored steps (the reference's maxChainSizeBytes).  At least one step is always storable.
    Chain(int numWalkers, int numParams, unsigned long long maxSize, unsigned long long blockBytes = Detail::DefaultBlockBytes,
          const Detail::BlockMemory& memory = Detail::BlockMemory())
        : blockMemory(memory), walkerCount(numWalkers), cellsPerWalker(numParams),
          cellsPerStep(static_cast<std::int64_t>(numWalkers) * numParams), stepCount(0)
    {
        const unsigned long long stepBytes = static_cast<unsigned long long>(cellsPerStep) * sizeof(ParamType);
        maxSteps = static_cast<std::int64_t>(maxSize / stepBytes);
        if (maxSteps < 1) maxSteps = 1;
        stepsPerBlock = static_cast<std::int64_t>(blockBytes / stepBytes);
        if (stepsPerBlock < 1) stepsPerBlock = 1;
        if (stepsPerBlock > maxSteps) stepsPerBlock = maxSteps;
    }
    ~Chain()
    {
        if (nextBlock.valid()) delete nextBlock.get();
        for (ChainBlock<ParamType>* b : blocks) delete b;
    }
    Chain(const Chain&) = delete;
    Chain& operator=(const Chain&) = delete;

    // ---- writing, one walker at a time (reference API: Chain.h:93,204-236) ---------------------------
    /// Copy one walker's parameters into the step being assembled.
    void storeWalker(int walkerNum, const ParamType* walkerData)
    {
        ParamType* dst = writableStep();
        if (dst) std::memcpy(dst + static_cast<std::size_t>(walkerNum) * cellsPerWalker, walkerData, sizeof(ParamType) * cellsPerWalker);
    }
    /// Close the step being assembled.
    IncrementStatus incrementChainStep()
    {
        if (stepCount >= maxSteps) return IncrementStatus::EndOfChain;
        if (!writableStep()) return IncrementStatus::EndOfChain;
        ++stepCount;
        blocks[static_cast<std::size_t>((stepCount - 1) / stepsPerBlock)]->setUsed((stepCount - 1) % stepsPerBlock + 1);
        if (prefetchBlocks) maybePrefetch(stepCount);
        if (stepCount >= maxSteps) return IncrementStatus::EndOfChain;
        return (stepCount % stepsPerBlock == 0) ? IncrementStatus::NewBlock : IncrementStatus::NormalIncrement;
    }

    // ---- writing, whole steps at once (device-to-host copies land here) -------------------------------
    /// Room left under the byte budget.
    std::int64_t remainingSteps() const { return maxSteps - stepCount; }
    /// Pointer where the next stored step goes and how many consecutive steps fit there contiguously (0 when full).
    ParamType* stepsContiguousFrom(std::int64_t* contiguous)
    {
        ParamType* p = (stepCount < maxSteps) ? writableStep() : nullptr;
        if (contiguous)
        {
            std::int64_t room = p ? stepsPerBlock - stepCount % stepsPerBlock : 0;
            if (room > maxSteps - stepCount) room = maxSteps - stepCount;
            *contiguous = room;
        }
        return p;
    }
    /// Announce that `count` steps are about to be written at the pointer handed out by stepsContiguousFrom (a device run
    /// in progress): lets the block behind the current one be obtained while they are being made.
    void expectSteps(std::int64_t count)
    {
        if (prefetchBlocks) maybePrefetch(stepCount + count);
    }
    /// Declare `count` steps written at the pointer handed out by stepsContiguousFrom.
    void commitSteps(std::int64_t count)
    {
        for (std::int64_t k = 0; k < count; ++k) incrementChainStep();
    }

    /// Obtain every block one block ahead of its use, on a helper thread (the samplers switch this on; off by default so
    /// that a Chain used alone never starts a thread).
    void setBlockPrefetch(bool on) { prefetchBlocks = on; }

    // ---- bookkeeping -------------------------------------------------------------------------------------
    std::int64_t getStoredStepCount() const { return stepCount; }
    std::int64_t getMaxStepCount() const { return maxSteps; }
    int getWalkerCount() const { return walkerCount; }
    int getCellsPerWalker() const { return cellsPerWalker; }
    std::int64_t getCellsPerStep() const { return cellsPerStep; }

    /// Forget every stored step, keep the memory (reference: Chain.h:255-266).
    void resetChain()
    {
        stepCount = 0;
        for (ChainBlock<ParamType>* b : blocks) b->setUsed(0);
        if (nextBlock.valid()) delete nextBlock.get();  // (a block obtained ahead of a chain that starts over)
    }

    /// Drop `burnInSamples` leading steps, then keep every `interval`-th of the rest, compacting in place
    /// (reference: Chain.h:268-305, same special cases).
    void resetChainForSubSampling(int burnInSamples, int interval)
    {
        if (burnInSamples == 0 && interval == 1) return;
        if (stepCount <= burnInSamples || (stepCount - burnInSamples) < interval)
        {
            resetChain();
            return;
        }
        const std::int64_t stored = stepCount;
        std::int64_t kept = 0;
        for (std::int64_t src = burnInSamples; src < stored; src += interval, ++kept)
            if (src != kept) std::memmove(stepPtr(kept), stepPtr(src), sizeof(ParamType) * static_cast<std::size_t>(cellsPerStep));
        stepCount = kept;
        for (std::size_t b = 0; b < blocks.size(); ++b)
        {
            const std::int64_t lo = static_cast<std::int64_t>(b) * stepsPerBlock;
            std::int64_t used = stepCount - lo;
            if (used < 0) used = 0;
            if (used > stepsPerBlock) used = stepsPerBlock;
            blocks[b]->setUsed(used);
        }
    }

    // ---- reading ---------------------------------------------------------------------------------------------
    PsetIterator getPsetIteratorBegin() { return PsetIterator(this, 0); }
    PsetIterator getPsetIteratorEnd() { return PsetIterator(this, stepCount * walkerCount); }
    StepIterator getStepIteratorBegin() { return StepIterator(this, 0); }
    StepIterator getStepIteratorEnd() { return StepIterator(this, stepCount); }

    /// W*D contiguous values of stored step k (k < getStoredStepCount()).
    ParamType* stepPtr(std::int64_t k)
    {
        return blocks[static_cast<std::size_t>(k / stepsPerBlock)]->step(k % stepsPerBlock);
    }

private:
    /// Once the last block is half full, obtain the one behind it on a helper thread (a run that never gets that far never
    /// pays for a block it does not use).
    void maybePrefetch(std::int64_t stepsSoon)
    {
        if (nextBlock.valid() || blocks.empty()) return;
        const std::int64_t lastLo = static_cast<std::int64_t>(blocks.size() - 1) * stepsPerBlock;
        if (stepsSoon - lastLo < (blocks.back()->capacity() + 1) / 2) return;
        const std::int64_t nextLo = static_cast<std::int64_t>(blocks.size()) * stepsPerBlock;
        std::int64_t nextWant = stepsPerBlock;
        if (nextLo + nextWant > maxSteps) nextWant = maxSteps - nextLo;
        if (nextWant < 1) return;
        const std::int64_t cells = cellsPerStep;
        const Detail::BlockMemory mem = blockMemory;
        nextBlock = std::async(std::launch::async, [nextWant, cells, mem]() { return new ChainBlock<ParamType>(nextWant, cells, mem, true); });
    }

    /// The step being assembled (allocating its block on first use); nullptr when memory is exhausted.
    ParamType* writableStep()
    {
        const std::size_t b = static_cast<std::size_t>(stepCount / stepsPerBlock);
        while (blocks.size() <= b)
        {
            std::int64_t want = stepsPerBlock;
            const std::int64_t lo = static_cast<std::int64_t>(blocks.size()) * stepsPerBlock;
            if (lo + want > maxSteps) want = maxSteps - lo;
            if (want < 1) return nullptr;
            // A block is obtained ahead, on a thread of its own, while the sampler fills the second half of the current one: pinned
            // memory from a provider costs some 35 ms per 256 MiB block, heap memory as much in first-touch page faults
            // (which that thread takes by writing to every page).
            ChainBlock<ParamType>* nb = nullptr;
            if (nextBlock.valid())
            {
                nb = nextBlock.get();
                if (nb && nb->capacity() != want)
                {
                    delete nb;
                    nb = nullptr;
                }
            }
            if (!nb) nb = new ChainBlock<ParamType>(want, cellsPerStep, blockMemory);
            if (!nb->valid())
            {
                delete nb;
                return nullptr;
            }
            blocks.push_back(nb);
        }
        return blocks[b]->step(stepCount % stepsPerBlock);
    }

    std::vector<ChainBlock<ParamType>*> blocks;
    std::future<ChainBlock<ParamType>*> nextBlock;  ///< the block behind the last one, being obtained
    bool prefetchBlocks = false;                    ///< obtain blocks one ahead on a helper thread (setBlockPrefetch)
    Detail::BlockMemory blockMemory;
    int walkerCount;
    int cellsPerWalker;
    std::int64_t cellsPerStep;
    std::int64_t stepsPerBlock;
    std::int64_t maxSteps;
    std::int64_t stepCount;
};

}  // namespace Chain
}  // namespace MCMC
#endif  // MCMCPP_CHAIN_CHAIN_H

// CPU unit test of the host Chain and its iterators (include/MCMCpp/Chain/*.h): cell layout, block
// boundaries, saturation of the iterators, burn-in/thinning compaction and the byte budget -- the
// behaviours the reference documents in MCMCpp/Chain/{Chain,ChainStepIterator,ChainPsetIterator}.h.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "Chain/Chain.h"

static int failures = 0;
#define CHECK(cond)                                                         \
    do                                                                      \
    {                                                                       \
        if (!(cond))                                                        \
        {                                                                   \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);     \
            ++failures;                                                     \
        }                                                                   \
    } while (0)

typedef MCMC::Chain::Chain<double> ChainD;
using MCMC::Chain::IncrementStatus;

// value stored for (step, walker, param): distinct everywhere
static double cell(long s, int w, int p) { return 1000.0 * s + 10.0 * w + p; }

static void fill(ChainD& c, long steps, int W, int D, long first = 0)
{
    std::vector<double> row(D);
    for (long s = 0; s < steps; ++s)
    {
        for (int w = 0; w < W; ++w)
        {
            for (int p = 0; p < D; ++p) row[p] = cell(first + s, w, p);
            c.storeWalker(w, row.data());
        }
        c.incrementChainStep();
    }
}

// a block-memory provider that counts its calls
struct ProviderCalls
{
    static long obtained, released;
    static unsigned long long lastBytes;
    static bool fail;
    static void* obtain(unsigned long long bytes)
    {
        if (fail) return nullptr;
        ++obtained;
        lastBytes = bytes;
        void* p = nullptr;
        return posix_memalign(&p, 64, bytes) == 0 ? p : nullptr;
    }
    static void release(void* p)
    {
        ++released;
        std::free(p);
    }
};
long ProviderCalls::obtained = 0, ProviderCalls::released = 0;
unsigned long long ProviderCalls::lastBytes = 0;
bool ProviderCalls::fail = false;

int main()
{
    const int W = 6, D = 3;
    const unsigned long long stepBytes = sizeof(double) * W * D;
    {
        // 5 steps per block, budget 23 steps: blocks of 5,5,5,5,3
        ChainD c(W, D, 23 * stepBytes, 5 * stepBytes);
        CHECK(c.getMaxStepCount() == 23);
        fill(c, 12, W, D);
        CHECK(c.getStoredStepCount() == 12);
        // step iterator: contents, ++ across block boundaries, saturation at both ends
        long s = 0;
        MCMC::Chain::ChainStepIterator<double> end = c.getStepIteratorEnd();
        for (MCMC::Chain::ChainStepIterator<double> it = c.getStepIteratorBegin(); it != end; ++it, ++s)
            for (int w = 0; w < W; ++w)
                for (int p = 0; p < D; ++p) CHECK((*it)[w * D + p] == cell(s, w, p));
        CHECK(s == 12);
        MCMC::Chain::ChainStepIterator<double> it = c.getStepIteratorBegin();
        --it;
        CHECK(it == c.getStepIteratorBegin());
        it += 7;
        CHECK((*it)[0] == cell(7, 0, 0));
        it -= 3;
        CHECK((*it)[D] == cell(4, 1, 0));
        it += 1000;
        CHECK(it == end);
        ++it;
        CHECK(it == end);
        it -= 1000;
        CHECK(it == c.getStepIteratorBegin());
        // parameter-set iterator: W*steps sets in order, no phantom cell at block boundaries
        long k = 0;
        MCMC::Chain::ChainPsetIterator<double> pend = c.getPsetIteratorEnd();
        for (MCMC::Chain::ChainPsetIterator<double> pit = c.getPsetIteratorBegin(); pit != pend; ++pit, ++k)
            for (int p = 0; p < D; ++p) CHECK((*pit)[p] == cell(k / W, (int)(k % W), p));
        CHECK(k == 12L * W);
        MCMC::Chain::ChainPsetIterator<double> pit = c.getPsetIteratorEnd();
        --pit;
        CHECK((*pit)[D - 1] == cell(11, W - 1, D - 1));
        // burn 2, keep every 3rd: steps 2, 5, 8, 11
        c.resetChainForSubSampling(2, 3);
        CHECK(c.getStoredStepCount() == 4);
        const long kept[4] = {2, 5, 8, 11};
        s = 0;
        for (MCMC::Chain::ChainStepIterator<double> jt = c.getStepIteratorBegin(); jt != c.getStepIteratorEnd(); ++jt, ++s)
            CHECK((*jt)[W * D - 1] == cell(kept[s], W - 1, D - 1));
        // special cases of the reference: (0,1) is a no-op, too few steps clears the chain
        c.resetChainForSubSampling(0, 1);
        CHECK(c.getStoredStepCount() == 4);
        c.resetChainForSubSampling(4, 1);
        CHECK(c.getStoredStepCount() == 0);
        // the budget: the 23rd step reports EndOfChain, nothing more is stored after it
        fill(c, 22, W, D);
        for (int w = 0; w < W; ++w)
        {
            double row[3] = {1, 2, 3};
            c.storeWalker(w, row);
        }
        CHECK(c.incrementChainStep() == IncrementStatus::EndOfChain);
        CHECK(c.getStoredStepCount() == 23 && c.remainingSteps() == 0);
        CHECK(c.incrementChainStep() == IncrementStatus::EndOfChain);
        CHECK(c.getStoredStepCount() == 23);
        c.resetChain();
        CHECK(c.getStoredStepCount() == 0 && c.remainingSteps() == 23);
    }
    {
        // bulk path used by the device-to-host copies
        ChainD c(W, D, 11 * stepBytes, 4 * stepBytes);
        std::int64_t room = 0;
        double* dst = c.stepsContiguousFrom(&room);
        CHECK(dst != nullptr && room == 4);
        for (long s = 0; s < 3; ++s)
            for (int w = 0; w < W; ++w)
                for (int p = 0; p < D; ++p) dst[(s * W + w) * D + p] = cell(s, w, p);
        c.commitSteps(3);
        dst = c.stepsContiguousFrom(&room);
        CHECK(room == 1);
        for (int w = 0; w < W; ++w)
            for (int p = 0; p < D; ++p) dst[w * D + p] = cell(3, w, p);
        c.commitSteps(1);
        dst = c.stepsContiguousFrom(&room);
        CHECK(room == 4);
        c.commitSteps(4);
        dst = c.stepsContiguousFrom(&room);
        CHECK(room == 3);  // 11-step budget: last block holds 3
        c.commitSteps(3);
        dst = c.stepsContiguousFrom(&room);
        CHECK(dst == nullptr && room == 0);
        CHECK((*c.getStepIteratorBegin())[5] == cell(0, 1, 2));
        MCMC::Chain::ChainStepIterator<double> it = c.getStepIteratorBegin();
        it += 3;
        CHECK((*it)[0] == cell(3, 0, 0));
    }
    {
        // sizes that overflow the reference's 32-bit indexing (ChainBlock.h:116-128): 131072 x 64 walkers
        MCMC::Chain::Chain<float> big(131072, 64, 3ULL * 131072 * 64 * sizeof(float));
        CHECK(big.getMaxStepCount() == 3 && big.getCellsPerStep() == 131072LL * 64);
    }
    {
        // blocks from a caller-supplied memory provider (the samplers hand out pinned host memory,
        // Device/SamplerCore.h): every block is obtained from it and returned to it, contents and iterators as before;
        // a provider that fails falls back to the heap
        ProviderCalls::obtained = ProviderCalls::released = 0;
        {
            ChainD c(W, D, 11 * stepBytes, 4 * stepBytes, MCMC::Chain::Detail::BlockMemory(&ProviderCalls::obtain, &ProviderCalls::release));
            fill(c, 11, W, D);
            CHECK(ProviderCalls::obtained == 3 && ProviderCalls::released == 0);
            CHECK(ProviderCalls::lastBytes == 3 * stepBytes);  // the last block holds what is left of the budget
            long s = 0;
            for (MCMC::Chain::ChainStepIterator<double> it = c.getStepIteratorBegin(); it != c.getStepIteratorEnd(); ++it, ++s)
                CHECK((*it)[W * D - 1] == cell(s, W - 1, D - 1));
            CHECK(s == 11);
            std::int64_t room = 0;
            CHECK(c.stepsContiguousFrom(&room) == nullptr && room == 0);
        }
        CHECK(ProviderCalls::released == 3);
        ProviderCalls::fail = true;
        {
            ChainD c(W, D, 4 * stepBytes, 4 * stepBytes, MCMC::Chain::Detail::BlockMemory(&ProviderCalls::obtain, &ProviderCalls::release));
            fill(c, 4, W, D);
            CHECK((*c.getStepIteratorBegin())[5] == cell(0, 1, 2));
        }
        CHECK(ProviderCalls::released == 3);  // (heap blocks are not handed to the provider's release)
        ProviderCalls::fail = false;
        // blocks obtained ahead on a helper thread once the current one is half full (what the samplers switch on): same
        // contents; a block that was obtained ahead and never used is returned when the chain goes (or starts over)
        ProviderCalls::obtained = ProviderCalls::released = 0;
        {
            ChainD c(W, D, 12 * stepBytes, 4 * stepBytes, MCMC::Chain::Detail::BlockMemory(&ProviderCalls::obtain, &ProviderCalls::release));
            c.setBlockPrefetch(true);
            fill(c, 5, W, D);  // the second block holds one step of four: nothing is obtained behind it yet
            CHECK(ProviderCalls::obtained == 2);
            fill(c, 2, W, D, 5);
            long s = 0;
            for (MCMC::Chain::ChainStepIterator<double> it = c.getStepIteratorBegin(); it != c.getStepIteratorEnd(); ++it, ++s)
                CHECK((*it)[W * D - 1] == cell(s, W - 1, D - 1));
            CHECK(s == 7);
            c.resetChain();  // the third block was obtained ahead and goes back; the two in use stay
            CHECK(ProviderCalls::obtained == 3 && ProviderCalls::released == 1);
            c.expectSteps(3);  // a device run about to write three steps: the second block is not the last one in use
            CHECK(ProviderCalls::obtained == 3);
        }
        CHECK(ProviderCalls::obtained == 3 && ProviderCalls::released == 3);
        {
            ChainD c(W, D, 12 * stepBytes, 4 * stepBytes);  // heap blocks, touched ahead
            c.setBlockPrefetch(true);
            fill(c, 12, W, D);
            CHECK((*c.getStepIteratorBegin())[5] == cell(0, 1, 2) && c.remainingSteps() == 0);
        }
    }
    {
        // The reference's own plumbing test at scale (/root/reference/test/sequential/InnerBenchmark/src/main.cpp:9-13,32:
        // 2400 walkers x 4 parameters x 20 000 steps, chain budget 3.3e9 bytes, Mover::SequenceMove advancing parameter j
        // of every walker by stepSize[j] = j + 1 per step, Movers/Diagnostic/SequenceMove.h): a walker-by-walker store
        // as the reference's movers do it, then the known answer -- stored step k holds k * {1, 2, 3, 4} in every walker --
        // through both iterators.  1.54 GB of chain, blocks of 256 MiB.
        const int BW = 2400, BD = 4, steps = 20000;
        ChainD c(BW, BD, 3300000000ULL);
        std::vector<double> walkers((size_t)BW * BD, 0.0);
        for (int w = 0; w < BW; ++w) c.storeWalker(w, walkers.data() + (size_t)w * BD);  // setInitialWalkerPos: step 0
        CHECK(c.incrementChainStep() != IncrementStatus::EndOfChain);
        bool full = false;
        for (int s = 0; s < steps && !full; ++s)
        {
            for (int w = 0; w < BW; ++w)
            {
                double* x = walkers.data() + (size_t)w * BD;
                for (int j = 0; j < BD; ++j) x[j] += (double)(j + 1);
                c.storeWalker(w, x);
            }
            full = c.incrementChainStep() == IncrementStatus::EndOfChain;
        }
        CHECK(!full && c.getStoredStepCount() == steps + 1);
        long k = 0, bad = 0;
        for (MCMC::Chain::ChainStepIterator<double> it = c.getStepIteratorBegin(); it != c.getStepIteratorEnd(); ++it, ++k)
            for (int w = 0; w < BW; ++w)
                for (int j = 0; j < BD; ++j) bad += (*it)[w * BD + j] != (double)k * (j + 1);
        CHECK(k == steps + 1 && bad == 0);
        long sets = 0;
        bad = 0;
        for (MCMC::Chain::ChainPsetIterator<double> pit = c.getPsetIteratorBegin(); pit != c.getPsetIteratorEnd(); ++pit, ++sets)
            bad += (*pit)[3] != (double)(sets / BW) * 4.0;
        CHECK(sets == (long)(steps + 1) * BW && bad == 0);
        // the reference's test then reads nothing else; burning and slicing at this size: keep every 1000th step after 1
        c.resetChainForSubSampling(1, 1000);
        CHECK(c.getStoredStepCount() == 20);
        k = 0;
        for (MCMC::Chain::ChainStepIterator<double> it = c.getStepIteratorBegin(); it != c.getStepIteratorEnd(); ++it, ++k)
            CHECK((*it)[1] == (double)(1 + 1000 * k) * 2.0);
    }
    if (failures == 0) std::printf("chain_test OK\n");
    return failures == 0 ? 0 : 1;
}

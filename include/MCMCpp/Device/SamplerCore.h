/*
 * Device/SamplerCore.h -- what EnsembleSampler and ParallelEnsembleSampler share on the MI355X path.
 *
 * Reference behaviour kept (paths relative to /root/reference/MCMCpp):
 *   - walkers [0, W/2) are the red set, [W/2, W) the black set; chain slot = walker index  (EnsembleSampler.h:211-215)
 *   - setInitialWalkerPos stores the initial placement as chain step 0 and counts it as one accepted
 *     step per walker                                                       (EnsembleSampler.h:220-230, Walker/Walker.h:76,162-170)
 *   - runMCMC(n): n stored steps; with sub-sampling interval k every stored step is preceded by k-1
 *     unstored ensemble steps; returns false once the chain's byte budget is exhausted   (EnsembleSampler.h:284-310)
 *   - the PostStepAction runs once per ensemble step and sees the chain without the step being made
 *                                                                            (EnsembleSampler.h:291-293,356-359)
 *   - reset() forgets chain and counters, keeps positions and the random stream          (EnsembleSampler.h:312-322)
 * Not kept: the parallel sampler's run-to-run non-determinism (ParallelEnsembleSampler.h:71-76) and its
 * sub-sampling defect (Threading/RedBlkCtrlerSpinLock.h:297-300) -- both samplers follow the sequential
 * trajectory, the only reproducible one.
 */
#ifndef MCMCPP_DEVICE_SAMPLERCORE_H
#define MCMCPP_DEVICE_SAMPLERCORE_H

#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../Chain/Chain.h"
#include "../Utility/NoAction.h"
#include "../Utility/UserOjbectsTest.h"
#include "HipBackend.h"

namespace MCMC
{
namespace Device
{

template <class ParamType, class Mover, class PostStepAction>
class SamplerCore
{
public:
    typedef Chain::Chain<ParamType> ChainType;
    typedef Chain::ChainPsetIterator<ParamType> PsetItt;
    typedef Chain::ChainStepIterator<ParamType> StepItt;

    static_assert(Mover::RunsOnDevice, "the MI355X samplers run movers that exist as device kernels: MCMC::Mover::StretchMove, MCMC::Mover::DifferentialEvolution");
    static_assert(Utility::CheckPerformAction<PostStepAction, void, const StepItt&, const StepItt&>::value,
                  "the PostStepAction needs 'void performAction(const StepItt& start, const StepItt& end)'");

    SamplerCore(int randSeed, long long stream, int numWalker, int numParameter, const Mover& move,
                unsigned long long maxChainSizeBytes, PostStepAction* stepAct, const Placement& where = Placement::fromEnvironment())
        : stepAction(stepAct),
          markovChain(numWalker, numParameter, maxChainSizeBytes, Chain::Detail::DefaultBlockBytes, chainMemory()),
          moveProposer(move), numParams(numParameter), numWalkers(numWalker), initialPlacementCounted(false)
    {
        markovChain.setBlockPrefetch(true);
        assert(numWalkers % 2 == 0);             // EnsembleSampler.h:207
        assert(numWalkers > (2 * numParams));    // EnsembleSampler.h:208
        moveProposer.setPrng(randSeed, stream);  // EnsembleSampler.h:217 / RedBlkUpdater.h:86
        mcmcpp_hip_config cfg = mcmcpp_hip_config();
        cfg.struct_size = sizeof(cfg);
        cfg.dtype = HipDtype<ParamType>::value;
        cfg.num_walkers = numWalkers;
        cfg.num_params = numParams;
        cfg.calc_id = Mover::CalculatorType::hipCalcId;
        cfg.calc_params = moveProposer.getCalculator().hipParams();
        cfg.calc_params_len = moveProposer.getCalculator().hipParamCount();
        cfg.seed = static_cast<std::uint64_t>(static_cast<long long>(randSeed));  // sign-extends like MultiSampler::setPrng
        cfg.stream = static_cast<std::uint64_t>(stream);
        cfg.gw_alpha_num = Mover::DistributionType::Numerator;
        cfg.gw_alpha_den = Mover::DistributionType::Denominator;
        cfg.mover = Mover::HipMoverId;
        const int G = where.count();
        ranks.resize(static_cast<std::size_t>(G));
        if (G == 1 && !where.splitEnsemble)
        {
            cfg.device = where.device(0);
            ranks[0].create(cfg);
            return;
        }
        // One ensemble split over G devices: one handle per device, all ranks of one RCCL communicator.  The communicator's
        // rendezvous blocks until every rank has joined, so the handles are created concurrently.
        {
            // Honest about its status: the exchange logic of this path is validated (several ranks on one GPU through a
            // loop-back collective library), RCCL's transport between DIFFERENT GPUs has carried it with one rank only.
            bool distinct = false;
            for (int r = 1; r < G; ++r) distinct = distinct || where.device(r) != where.device(0);
            if (distinct && std::getenv("MCMCPP_QUIET") == nullptr)
                std::fprintf(stderr, "MCMCpp (MI355X): one ensemble split over %d GPUs -- note: this path has not yet run on multi-GPU hardware "
                                     "(DESIGN.md section 5); results are checkable against a single-GPU run, which they must equal bit for bit\n", G);
        }
        unsigned char commId[MCMCPP_HIP_COMM_ID_BYTES];
        HipHandle::checkCreate("mcmcpp_hip_comm_unique_id", mcmcpp_hip_comm_unique_id(commId));
        std::vector<std::thread> joiners;
        for (int r = 0; r < G; ++r)
            joiners.push_back(std::thread([this, cfg, r, G, &where, &commId]() {
                mcmcpp_hip_config mine = cfg;
                mine.device = where.device(r);
                mine.comm_world = G;
                mine.comm_rank = r;
                mine.comm_id = commId;
                ranks[static_cast<std::size_t>(r)].create(mine);
            }));
        for (std::thread& t : joiners) t.join();
    }

    void setInitialWalkerPos(ParamType* positions, ParamType* auxValues)
    {
        for (HipHandle& h : ranks) h.check("mcmcpp_hip_set_state", mcmcpp_hip_set_state(h.get(), positions, auxValues));
        for (int w = 0; w < numWalkers; ++w) markovChain.storeWalker(w, positions + static_cast<std::size_t>(w) * numParams);
        markovChain.incrementChainStep();
        initialPlacementCounted = true;
    }

    void storeCurrentWalkerPositions()
    {
        std::int64_t room = 0;
        ParamType* dst = markovChain.stepsContiguousFrom(&room);
        if (!dst || room < 1) return;
        ranks[0].check("mcmcpp_hip_get_state", mcmcpp_hip_get_state(ranks[0].get(), dst, nullptr, nullptr));
        markovChain.commitSteps(1);
    }

    /// numSteps stored steps, each the last of `interval` ensemble steps.  False when the chain filled up first.
    bool run(int numSteps, int interval)
    {
        std::int64_t left = numSteps;
        while (left > 0)
        {
            std::int64_t room = 0;
            ParamType* dst = markovChain.stepsContiguousFrom(&room);
            if (!dst || room < 1) return false;
            const std::int64_t now = left < room ? left : room;
            markovChain.expectSteps(now);
            if (stepAction == nullptr && ranks.size() == 1)
            {
                ranks[0].check("mcmcpp_hip_run", mcmcpp_hip_run(ranks[0].get(), now, interval, dst, nullptr));
                markovChain.commitSteps(now);
            }
            else
            {
                // Every rank steps on its handle's worker thread (a split ensemble's ranks meet in the exchanges), rank 0
                // receives the stored steps.  The PostStepAction runs here, beside the device: once per ensemble step, and
                // -- as in the reference, which calls it before the chain moves on to the next step
                // (EnsembleSampler.h:291-293,356-359) -- it sees the chain WITHOUT the step being made.  Nothing it can
                // observe changes during the interval-1 unstored steps, so their calls follow each other directly.
                for (std::size_t r = 0; r < ranks.size(); ++r)
                    ranks[r].check("mcmcpp_hip_run_async", mcmcpp_hip_run_async(ranks[r].get(), now, interval, r == 0 ? dst : nullptr, nullptr));
                for (std::int64_t k = 0; k < now; ++k)
                {
                    if (stepAction != nullptr)
                    {
                        ranks[0].check("mcmcpp_hip_wait_stored", mcmcpp_hip_wait_stored(ranks[0].get(), k + 1));
                        for (int j = 0; j < interval; ++j) stepAction->performAction(markovChain.getStepIteratorBegin(), markovChain.getStepIteratorEnd());
                        markovChain.commitSteps(1);
                    }
                }
                for (HipHandle& h : ranks) h.check("mcmcpp_hip_run_wait", mcmcpp_hip_run_wait(h.get()));
                if (stepAction == nullptr) markovChain.commitSteps(now);
            }
            left -= now;
            if (markovChain.remainingSteps() == 0) return false;  // budget reached with this step (EnsembleSampler.h:293,306)
        }
        return true;
    }

    void reset()
    {
        markovChain.resetChain();
        for (HipHandle& h : ranks) h.check("mcmcpp_hip_reset_counters", mcmcpp_hip_reset_counters(h.get()));
        initialPlacementCounted = false;
    }

    unsigned long long acceptedSteps()
    {
        unsigned long long total = 0;
        for (HipHandle& h : ranks)  // (a rank counts the walkers it updates)
        {
            std::uint64_t acc = 0;
            h.check("mcmcpp_hip_get_counters", mcmcpp_hip_get_counters(h.get(), &acc, nullptr, nullptr, nullptr));
            total += acc;
        }
        return total + (initialPlacementCounted ? static_cast<unsigned long long>(numWalkers) : 0ULL);
    }
    unsigned long long totalSteps()
    {
        std::uint64_t steps = 0;
        ranks[0].check("mcmcpp_hip_get_counters", mcmcpp_hip_get_counters(ranks[0].get(), nullptr, &steps, nullptr, nullptr));
        return static_cast<unsigned long long>(numWalkers) * (steps + (initialPlacementCounted ? 1ULL : 0ULL));
    }
    ParamType acceptanceFraction()
    {
        return static_cast<ParamType>(acceptedSteps()) / static_cast<ParamType>(totalSteps());
    }

    /// Walker positions, log-posteriors and per-walker accepted counts as they stand on the device (any may be null).
    void currentState(ParamType* positions, ParamType* logp, std::uint32_t* nAccept)
    {
        ranks[0].check("mcmcpp_hip_get_state", mcmcpp_hip_get_state(ranks[0].get(), positions, logp, nAccept));
    }
    /// Parity diagnostics of the device path (see include/mcmcpp_hip.h), summed over the devices.
    void diagnostics(std::uint64_t* nearTies, std::uint64_t* redraws)
    {
        std::uint64_t t = 0, r = 0;
        for (HipHandle& h : ranks)
        {
            std::uint64_t ht = 0, hr = 0;
            h.check("mcmcpp_hip_get_counters", mcmcpp_hip_get_counters(h.get(), nullptr, nullptr, &ht, &hr));
            t += ht;
            r += hr;
        }
        if (nearTies) *nearTies = t;
        if (redraws) *redraws = r;
    }
    int deviceCount() const { return static_cast<int>(ranks.size()); }

    ChainType& chain() { return markovChain; }

protected:
    static void* pinnedObtain(unsigned long long bytes) { return mcmcpp_hip_host_alloc(bytes); }
    static void pinnedRelease(void* p) { mcmcpp_hip_host_free(p); }
    /// Chain blocks come from the heap (stored steps pass through the library's pinned staging ring and are copied out by
    /// the host while the launches continue) unless MCMCPP_CHAIN_MEMORY=pinned asks for pinned host memory, into which the
    /// step launches write stored steps directly.  Measured at C2 (tools/bench_facade.cpp): pinned blocks win
    /// when the memory already exists (a chain that is reset and refilled); obtaining them costs as much as the page
    /// faults of fresh heap memory but stalls the launches while it lasts, so a growing chain is faster on the heap.
    static Chain::Detail::BlockMemory chainMemory()
    {
        const char* v = std::getenv("MCMCPP_CHAIN_MEMORY");
        if (v && v[0] == 'p' && v[1] == 'i') return Chain::Detail::BlockMemory(&pinnedObtain, &pinnedRelease);
        return Chain::Detail::BlockMemory();
    }

    PostStepAction* stepAction;
    ChainType markovChain;
    Mover moveProposer;
    std::vector<HipHandle> ranks;  ///< one handle per device of the placement
    int numParams;
    int numWalkers;
    bool initialPlacementCounted;
};

}  // namespace Device
}  // namespace MCMC
#endif  // MCMCPP_DEVICE_SAMPLERCORE_H

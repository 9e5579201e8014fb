/*
 * Device/SamplerCore.h -- what EnsembleSampler and ParallelEnsembleSampler share on the MI355X path.
 *
 * Reference behaviour kept (paths relative to /root/reference/MCMCpp):
 *   - walkers [0, W/2) are the red set, [W/2, W) the black set; chain slot = walker index  (EnsembleSampler.h:211-215)
 *   - setInitialWalkerPos stores the initial placement as chain step 0 and counts it as one accepted
 *     step per walker                                                       (EnsembleSampler.h:220-230, Walker/Walker.h:76,162-170)
 *   - runMCMC(n): n stored steps; with sub-sampling interval k every stored step is preceded by k-1
 *     unstored ensemble steps; returns false once the chain's byte budget is exhausted   (EnsembleSampler.h:284-310)
 *   - the PostStepAction runs once after every ensemble step                 (EnsembleSampler.h:356-359)
 *   - reset() forgets chain and counters, keeps positions and the random stream          (EnsembleSampler.h:312-322)
 * Not kept: the parallel sampler's run-to-run non-determinism (ParallelEnsembleSampler.h:71-76) and its
 * sub-sampling defect (Threading/RedBlkCtrlerSpinLock.h:297-300) -- both samplers follow the sequential
 * trajectory, the only reproducible one.
 */
#ifndef MCMCPP_DEVICE_SAMPLERCORE_H
#define MCMCPP_DEVICE_SAMPLERCORE_H

#include <cassert>
#include <cstdint>
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
                unsigned long long maxChainSizeBytes, PostStepAction* stepAct)
        : stepAction(stepAct), markovChain(numWalker, numParameter, maxChainSizeBytes), moveProposer(move),
          numParams(numParameter), numWalkers(numWalker), initialPlacementCounted(false)
    {
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
        cfg.device = -1;
        cfg.gw_alpha_num = Mover::DistributionType::Numerator;
        cfg.gw_alpha_den = Mover::DistributionType::Denominator;
        cfg.mover = Mover::HipMoverId;
        device.create(cfg);
    }

    void setInitialWalkerPos(ParamType* positions, ParamType* auxValues)
    {
        device.check("mcmcpp_hip_set_state", mcmcpp_hip_set_state(device.get(), positions, auxValues));
        for (int w = 0; w < numWalkers; ++w) markovChain.storeWalker(w, positions + static_cast<std::size_t>(w) * numParams);
        markovChain.incrementChainStep();
        initialPlacementCounted = true;
    }

    void storeCurrentWalkerPositions()
    {
        std::int64_t room = 0;
        ParamType* dst = markovChain.stepsContiguousFrom(&room);
        if (!dst || room < 1) return;
        device.check("mcmcpp_hip_get_state", mcmcpp_hip_get_state(device.get(), dst, nullptr, nullptr));
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
            // a PostStepAction sees the chain after every ensemble step, so it forces one step per launch batch
            std::int64_t now = (stepAction != nullptr) ? 1 : (left < room ? left : room);
            if (stepAction != nullptr && interval > 1)
            {
                // interval-1 unstored steps, the action after each, then the stored one
                for (int j = 1; j < interval; ++j)
                {
                    device.check("mcmcpp_hip_run", mcmcpp_hip_run(device.get(), 1, 1, nullptr, nullptr));
                    stepAction->performAction(markovChain.getStepIteratorBegin(), markovChain.getStepIteratorEnd());
                }
                device.check("mcmcpp_hip_run", mcmcpp_hip_run(device.get(), 1, 1, dst, nullptr));
            }
            else
                device.check("mcmcpp_hip_run", mcmcpp_hip_run(device.get(), now, interval, dst, nullptr));
            markovChain.commitSteps(now);
            if (stepAction != nullptr) stepAction->performAction(markovChain.getStepIteratorBegin(), markovChain.getStepIteratorEnd());
            left -= now;
            if (markovChain.remainingSteps() == 0) return false;  // budget reached with this step (EnsembleSampler.h:293,306)
        }
        return true;
    }

    void reset()
    {
        markovChain.resetChain();
        device.check("mcmcpp_hip_reset_counters", mcmcpp_hip_reset_counters(device.get()));
        initialPlacementCounted = false;
    }

    unsigned long long acceptedSteps()
    {
        std::uint64_t acc = 0;
        device.check("mcmcpp_hip_get_counters", mcmcpp_hip_get_counters(device.get(), &acc, nullptr, nullptr, nullptr));
        return acc + (initialPlacementCounted ? static_cast<unsigned long long>(numWalkers) : 0ULL);
    }
    unsigned long long totalSteps()
    {
        std::uint64_t steps = 0;
        device.check("mcmcpp_hip_get_counters", mcmcpp_hip_get_counters(device.get(), nullptr, &steps, nullptr, nullptr));
        return static_cast<unsigned long long>(numWalkers) * (steps + (initialPlacementCounted ? 1ULL : 0ULL));
    }
    ParamType acceptanceFraction()
    {
        return static_cast<ParamType>(acceptedSteps()) / static_cast<ParamType>(totalSteps());
    }

    /// Walker positions, log-posteriors and per-walker accepted counts as they stand on the device (any may be null).
    void currentState(ParamType* positions, ParamType* logp, std::uint32_t* nAccept)
    {
        device.check("mcmcpp_hip_get_state", mcmcpp_hip_get_state(device.get(), positions, logp, nAccept));
    }
    /// Parity diagnostics of the device path (see include/mcmcpp_hip.h).
    void diagnostics(std::uint64_t* nearTies, std::uint64_t* redraws)
    {
        device.check("mcmcpp_hip_get_counters", mcmcpp_hip_get_counters(device.get(), nullptr, nullptr, nearTies, redraws));
    }

    ChainType& chain() { return markovChain; }

protected:
    PostStepAction* stepAction;
    ChainType markovChain;
    Mover moveProposer;
    HipHandle device;
    int numParams;
    int numWalkers;
    bool initialPlacementCounted;
};

}  // namespace Device
}  // namespace MCMC
#endif  // MCMCPP_DEVICE_SAMPLERCORE_H

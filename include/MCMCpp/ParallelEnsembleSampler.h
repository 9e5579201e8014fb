/*
 * ParallelEnsembleSampler.h -- the parallel sampler facade: the drop-in target named by BASELINE.json.
 *
 * Same template parameters and public surface as /root/reference/MCMCpp/ParallelEnsembleSampler.h:78-210.
 * The reference parallelises with a pool of std::threads, a red/black barrier controller and one pcg
 * stream per thread (ParallelEnsembleSampler.h:250-261, Threading/RedBlkUpdater.h:81-87), which makes its
 * results depend on thread scheduling (ParallelEnsembleSampler.h:71-76).  Here the parallelism is the GPU
 * grid -- one kernel launch per half-step, the launch boundary being the reference's mid/end-step barrier
 * (Threading/RedBlkCtrlerSpinLock.h:240-322) -- and every walker addresses the single stream-0 sequence of
 * the sequential sampler, so the result is reproducible and equal to EnsembleSampler's.
 * threadCount and UseSpinLocks are accepted for source compatibility and do not affect the device path; which GPUs
 * the sampler uses is a Device::Placement (constructor overload, or MCMCPP_DEVICES in the environment for unchanged
 * user code): several devices split the one ensemble between them (Device/SamplerCore.h).
 * Every public method takes the sampler mutex, as in the reference.
 */
#ifndef MCMCPP_PARALLELENSEMBLESAMPLER_H
#define MCMCPP_PARALLELENSEMBLESAMPLER_H

#include <cassert>
#include <mutex>

#include "Device/SamplerCore.h"

namespace MCMC
{

template <class ParamType, class Mover, class PostStepAction = Utility::NoAction<ParamType>, bool UseSpinLocks = true>
class ParallelEnsembleSampler : private Device::SamplerCore<ParamType, Mover, PostStepAction>
{
    typedef Device::SamplerCore<ParamType, Mover, PostStepAction> Core;
    typedef std::unique_lock<std::mutex> Lock;

public:
    typedef typename Core::ChainType ChainType;
    typedef typename Core::PsetItt PsetItt;
    typedef typename Core::StepItt StepItt;

    ParallelEnsembleSampler(int randSeed, int threadCount, int numWalker, int numParameter, const Mover& move,
                            unsigned long long maxChainSizeBytes = 2147483648ULL, PostStepAction* stepAct = nullptr)
        : Core(randSeed, 0, numWalker, numParameter, move, maxChainSizeBytes, stepAct), numThreads(threadCount), subSamplingInterval(1)
    {
        assert(threadCount > 0);
    }
    /// The same on an explicit set of GPUs (not in the reference, whose only placement argument is threadCount): several
    /// devices split the ONE ensemble between them, exchanging the updated rows over RCCL once per ensemble step.
    ParallelEnsembleSampler(int randSeed, int threadCount, int numWalker, int numParameter, const Mover& move, unsigned long long maxChainSizeBytes,
                            PostStepAction* stepAct, const Device::Placement& where)
        : Core(randSeed, 0, numWalker, numParameter, move, maxChainSizeBytes, stepAct, where), numThreads(threadCount), subSamplingInterval(1)
    {
        assert(threadCount > 0);
    }
    ParallelEnsembleSampler(const ParallelEnsembleSampler&) = delete;
    ParallelEnsembleSampler& operator=(const ParallelEnsembleSampler&) = delete;

    void setInitialWalkerPos(ParamType* positions, ParamType* auxValues)
    {
        Lock lock(samplerMutex);
        Core::setInitialWalkerPos(positions, auxValues);
    }
    void storeCurrentWalkerPositions()
    {
        Lock lock(samplerMutex);
        Core::storeCurrentWalkerPositions();
    }
    /// numSteps stored steps, each preceded by subSamplingInt-1 unstored ones (see setSamplingMode).
    bool runMCMC(int numSteps)
    {
        Lock lock(samplerMutex);
        return Core::run(numSteps, subSamplingInterval);
    }
    void reset()
    {
        Lock lock(samplerMutex);
        Core::reset();
    }
    ParamType getAcceptanceFraction()
    {
        Lock lock(samplerMutex);
        return Core::acceptanceFraction();
    }
    /// Sub-sampling interval for future steps; also thins/burns what is already stored
    /// (reference: ParallelEnsembleSampler.h:322-330).
    void setSamplingMode(int subSamplingInt = 1, int burnIn = 0)
    {
        assert(subSamplingInt > 0);
        assert(burnIn >= 0);
        Lock lock(samplerMutex);
        subSamplingInterval = subSamplingInt;
        this->markovChain.resetChainForSubSampling(burnIn, subSamplingInt);
    }
    int getStoredSteps()
    {
        Lock lock(samplerMutex);
        return static_cast<int>(this->markovChain.getStoredStepCount());
    }
    PsetItt getParamSetIttBegin()
    {
        Lock lock(samplerMutex);
        return this->markovChain.getPsetIteratorBegin();
    }
    PsetItt getParamSetIttEnd()
    {
        Lock lock(samplerMutex);
        return this->markovChain.getPsetIteratorEnd();
    }
    StepItt getStepIttBegin()
    {
        Lock lock(samplerMutex);
        return this->markovChain.getStepIteratorBegin();
    }
    StepItt getStepIttEnd()
    {
        Lock lock(samplerMutex);
        return this->markovChain.getStepIteratorEnd();
    }

    // ---- additions of the device path (not in the reference) -------------------------------------------------
    unsigned long long getAcceptedSteps()
    {
        Lock lock(samplerMutex);
        return Core::acceptedSteps();
    }
    unsigned long long getTotalSteps()
    {
        Lock lock(samplerMutex);
        return Core::totalSteps();
    }
    using Core::currentState;
    using Core::diagnostics;
    using Core::deviceCount;

private:
    std::mutex samplerMutex;
    int numThreads;
    int subSamplingInterval;
};

}  // namespace MCMC
#endif  // MCMCPP_PARALLELENSEMBLESAMPLER_H

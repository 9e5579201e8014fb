/*
 * EnsembleSampler.h -- the sequential sampler facade, running on one MI355X.
 *
 * Drop-in for /root/reference/MCMCpp/EnsembleSampler.h: same template parameters
 * (ParamType, Mover, PostStepAction = NoAction), same constructor (EnsembleSampler.h:66-67) and the same
 * public methods (EnsembleSampler.h:89-176).  The step loop the reference runs on the host
 * (runMCMC -> performStep -> Mover::updateWalker, EnsembleSampler.h:284-360) is executed by gfx950 kernels
 * behind the C ABI of libmcmcpp_hip.so; the Chain and its iterators stay host objects.
 */
#ifndef MCMCPP_ENSEMBLESAMPLER_H
#define MCMCPP_ENSEMBLESAMPLER_H

#include <cassert>

#include "Device/SamplerCore.h"

namespace MCMC
{

template <class ParamType, class Mover, class PostStepAction = Utility::NoAction<ParamType> >
class EnsembleSampler : private Device::SamplerCore<ParamType, Mover, PostStepAction>
{
    typedef Device::SamplerCore<ParamType, Mover, PostStepAction> Core;

public:
    typedef typename Core::ChainType ChainType;
    typedef typename Core::PsetItt PsetItt;
    typedef typename Core::StepItt StepItt;

    /// randSeed seeds the pcg64 stream (stream number 0, as in the reference); numWalker must be even and
    /// exceed 2*numParameter; maxChainSizeBytes bounds the host memory of stored steps.
    EnsembleSampler(int randSeed, int numWalker, int numParameter, const Mover& move,
                    unsigned long long maxChainSizeBytes = 2147483648ULL, PostStepAction* stepAct = nullptr)
        : Core(randSeed, 0, numWalker, numParameter, move, maxChainSizeBytes, stepAct), subSamplingInterval(1), subSampling(false)
    {
    }
    EnsembleSampler(const EnsembleSampler&) = delete;
    EnsembleSampler& operator=(const EnsembleSampler&) = delete;

    /// positions: numWalker*numParameter values, walker-major; auxValues: the log-posterior of each walker.
    void setInitialWalkerPos(ParamType* positions, ParamType* auxValues) { Core::setInitialWalkerPos(positions, auxValues); }
    /// Append the walkers' current positions to the chain (useful after reset()).
    void storeCurrentWalkerPositions() { Core::storeCurrentWalkerPositions(); }

    /// Store numSteps more steps (numSteps*slicingInterval ensemble steps when slicing).  False when the
    /// chain's byte budget ran out first.
    bool runMCMC(int numSteps) { return Core::run(numSteps, subSampling ? subSamplingInterval : 1); }

    /// Forget the chain and the counters; the walkers stay where they are.
    void reset() { Core::reset(); }

    ParamType getAcceptanceFraction() { return Core::acceptanceFraction(); }
    unsigned long long getAcceptedSteps() { return Core::acceptedSteps(); }
    unsigned long long getTotalSteps() { return Core::totalSteps(); }

    void setSlicingMode(bool useSlicing = false, int slicingInterval = 1)
    {
        assert(slicingInterval > 0);
        subSampling = useSlicing;
        subSamplingInterval = slicingInterval;
    }
    /// Drop burnIn leading stored steps and keep every slicingInterval-th of the rest.
    void sliceAndBurnChain(int slicingInterval, int burnIn)
    {
        assert(slicingInterval > 0);
        assert(burnIn >= 0);
        this->markovChain.resetChainForSubSampling(burnIn, slicingInterval);
    }

    int getStoredSteps() { return static_cast<int>(this->markovChain.getStoredStepCount()); }

    PsetItt getParamSetIttBegin() { return this->markovChain.getPsetIteratorBegin(); }
    PsetItt getParamSetIttEnd() { return this->markovChain.getPsetIteratorEnd(); }
    StepItt getStepIttBegin() { return this->markovChain.getStepIteratorBegin(); }
    StepItt getStepIttEnd() { return this->markovChain.getStepIteratorEnd(); }

    // ---- additions of the device path (not in the reference) -------------------------------------------------
    using Core::currentState;
    using Core::diagnostics;

private:
    int subSamplingInterval;
    bool subSampling;
};

}  // namespace MCMC
#endif  // MCMCPP_ENSEMBLESAMPLER_H

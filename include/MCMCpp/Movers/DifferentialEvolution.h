/*
 * Movers/DifferentialEvolution.h -- the differential-evolution mover (ter Braak) as seen by a program written against
 * the reference (/root/reference/MCMCpp/Movers/DifferentialEvolution.h:28-126).
 *
 * Same template parameters (ParamType, Calculator) and constructor (numParams, prngInit, Calculator).  In the
 * reference the mover object itself draws ind1, ind2 != ind1, D uniform jitters and one exponential per update and
 * moves one walker at a time; here it is a description -- dimension, seed, Calculator -- that EnsembleSampler /
 * ParallelEnsembleSampler hand to libmcmcpp_hip.so (config.mover = MCMCPP_HIP_MOVER_DIFFERENTIAL_EVOLUTION), whose
 * kernels follow the reference's single pcg64 stream draw for draw, thrown-away draws included (csrc/diffevo_kernel.hpp):
 * same chain as the reference's sequential sampler, bit for bit (tests/test_diffevo.py).
 *
 * overrideGamma / overrideRandBounds exist and do what they do in the reference: nothing a sampler sees.  The
 * samplers copy the mover (EnsembleSampler.h:199-204), and the reference's copy constructor re-initialises gamma to
 * 2.38 / sqrt(2 D) and the jitter bounds to +-1e-4 (DifferentialEvolution.h:65-69,120-121); the kernels use those values.
 */
#ifndef MCMCPP_MOVERS_DIFFERENTIALEVOLUTION_H
#define MCMCPP_MOVERS_DIFFERENTIALEVOLUTION_H

#include <cmath>
#include <type_traits>

#include "../Utility/GwDistribution.h"
#include "../Utility/UserOjbectsTest.h"

namespace MCMC
{
namespace Mover
{
template <class ParamType, class Calculator>
class DifferentialEvolution
{
public:
    typedef ParamType ValueType;
    typedef Calculator CalculatorType;
    typedef Utility::GwDistribution<ParamType, 2, 1> DistributionType;  // (the reference's unused DistType; not part of this move)
    /// Marks the movers the MI355X samplers can run, and which kernels they select (include/mcmcpp_hip.h).
    static const bool RunsOnDevice = true;
    static const unsigned HipMoverId = 1u;  // MCMCPP_HIP_MOVER_DIFFERENTIAL_EVOLUTION

    static_assert(Utility::CheckCalcLogPostProb<Calculator, ParamType, ParamType*>::value,
                  "DifferentialEvolution: the Calculator needs 'ParamType calcLogPostProb(ParamType* paramSet)'");
    static_assert(std::is_copy_constructible<Calculator>::value, "DifferentialEvolution: the Calculator must be copy constructible");
    static_assert(Utility::CheckDeviceCalculator<Calculator>::value,
                  "DifferentialEvolution (MI355X): the Calculator must also name its device functor -- hipCalcId, hipParams(), "
                  "hipParamCount() -- see MCMCpp/Device/Calculators.h; arbitrary host code cannot run inside the GPU kernel");

    DifferentialEvolution(int numParams, long long prngInit, const Calculator& orig)
        : paramCount(numParams), gamma(static_cast<ParamType>(2.38 / std::sqrt(static_cast<double>(2 * numParams)))), smallRandWidth(2.0e-4),
          smallRandLowEdge(-1.0e-4), prngSeed(prngInit), prngStream(0), calc(orig)
    {
    }
    /// As in the reference (DifferentialEvolution.h:65-69): a copy starts from the default gamma and jitter bounds.
    DifferentialEvolution(const DifferentialEvolution& rhs)
        : paramCount(rhs.paramCount), gamma(static_cast<ParamType>(2.38 / std::sqrt(static_cast<double>(2 * rhs.paramCount)))), smallRandWidth(2.0e-4),
          smallRandLowEdge(-1.0e-4), prngSeed(rhs.prngSeed), prngStream(rhs.prngStream), calc(rhs.calc)
    {
    }
    DifferentialEvolution& operator=(const DifferentialEvolution&) = delete;

    void overrideGamma(ParamType newGamma) { gamma = newGamma; }
    void overrideRandBounds(ParamType newRand)
    {
        smallRandLowEdge = -newRand;
        smallRandWidth = 2 * newRand;
    }
    void setPrng(long long seed, long long stream)
    {
        prngSeed = seed;
        prngStream = stream;
    }

    int getNumParams() const { return paramCount; }
    long long getSeed() const { return prngSeed; }
    long long getStream() const { return prngStream; }
    ParamType getGamma() const { return gamma; }
    const Calculator& getCalculator() const { return calc; }
    Calculator& getCalculator() { return calc; }

private:
    int paramCount;
    ParamType gamma;
    ParamType smallRandWidth;
    ParamType smallRandLowEdge;
    long long prngSeed;
    long long prngStream;
    Calculator calc;
};

}  // namespace Mover
}  // namespace MCMC
#endif  // MCMCPP_MOVERS_DIFFERENTIALEVOLUTION_H

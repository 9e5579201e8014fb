/*
 * Movers/StretchMove.h -- the affine-invariant stretch move of Goodman & Weare, as a Mover descriptor.
 *
 * In the reference this class IS the hot loop: updateWalker draws a partner and a stretch factor on the
 * host and evaluates the Calculator (/root/reference/MCMCpp/Movers/StretchMove.h:100-123).  Here the
 * walkers live in HBM and that loop is the gfx950 kernel stretch_half_step_kernel
 * (mcmcpp_amd/csrc/stretch_kernel.hpp); the class keeps the reference's template parameters and
 * constructor (StretchMove.h:42-43,62-66,90) and carries what the kernel needs: the Calculator's device
 * identity and parameters, the dimension, the stretch-scale alpha and the stream selection.
 * There is deliberately no host updateWalker: a host fallback of the hot path would defeat the drop-in.
 */
#ifndef MCMCPP_MOVERS_STRETCHMOVE_H
#define MCMCPP_MOVERS_STRETCHMOVE_H

#include <type_traits>

#include "../Utility/GwDistribution.h"
#include "../Utility/UserOjbectsTest.h"

namespace MCMC
{
namespace Mover
{
namespace Detail
{
template <class D>
struct IsGwDistribution : std::false_type
{
};
template <class T, int N, int M>
struct IsGwDistribution<Utility::GwDistribution<T, N, M> > : std::true_type
{
};
}  // namespace Detail

template <class ParamType, class Calculator, class CustomDistribution = Utility::GwDistribution<ParamType, 2, 1> >
class StretchMove
{
public:
    typedef ParamType ValueType;
    typedef Calculator CalculatorType;
    typedef CustomDistribution DistributionType;
    /// Marks the movers the MI355X samplers can run.
    static const bool RunsOnDevice = true;
    static const unsigned HipMoverId = 0u;  // MCMCPP_HIP_MOVER_STRETCH

    static_assert(Utility::CheckCalcLogPostProb<Calculator, ParamType, ParamType*>::value,
                  "StretchMove: the Calculator needs 'ParamType calcLogPostProb(ParamType* paramSet)'");
    static_assert(std::is_copy_constructible<Calculator>::value, "StretchMove: the Calculator must be copy constructible");
    static_assert(Utility::CheckDeviceCalculator<Calculator>::value,
                  "StretchMove (MI355X): the Calculator must also name its device functor -- hipCalcId, hipParams(), "
                  "hipParamCount() -- see MCMCpp/Device/Calculators.h; arbitrary host code cannot run inside the GPU kernel");
    static_assert(Utility::CheckFunctor<CustomDistribution, ParamType, ParamType>::value,
                  "StretchMove: the CustomDistribution needs 'ParamType operator()(ParamType)'");
    static_assert(Detail::IsGwDistribution<CustomDistribution>::value,
                  "StretchMove (MI355X): the stretch-factor distribution must be Utility::GwDistribution<ParamType, Num, Denom>");

    /// numParams: dimension D; prngInit: seed (the samplers re-seed through setPrng, as in the reference);
    /// orig: the Calculator, copied.
    StretchMove(int numParams, long long prngInit, const Calculator& orig)
        : paramCount(numParams), prngSeed(prngInit), prngStream(0), calc(orig)
    {
    }

    /// Seed and stream of the pcg64 engine every random draw comes from (reference: StretchMove.h:90).
    void setPrng(long long seed, long long stream)
    {
        prngSeed = seed;
        prngStream = stream;
    }

    int getNumParams() const { return paramCount; }
    long long getSeed() const { return prngSeed; }
    long long getStream() const { return prngStream; }
    const Calculator& getCalculator() const { return calc; }
    Calculator& getCalculator() { return calc; }

private:
    int paramCount;
    long long prngSeed;
    long long prngStream;
    Calculator calc;
};

}  // namespace Mover
}  // namespace MCMC
#endif  // MCMCPP_MOVERS_STRETCHMOVE_H

/*
 * Analysis/AutoCorrCalc.h -- integrated autocorrelation time of every parameter in a chain, computed on the MI355X.
 *
 * Same class, constructor and methods as the reference (/root/reference/MCMCpp/Analysis/AutoCorrCalc.h:39-147):
 *
 *     MCMC::Analysis::AutoCorrCalc<double> ac(numParams, numWalkers);
 *     ac.setAutoCorrScaleFactor(5);                                        // optional, default 4 as in the reference
 *     ac.calcAutoCorrTimes(sampler.getStepIttBegin(), sampler.getStepIttEnd(), numSamples [, numWalkersToUse]);
 *     ac.retrieveAutoCorrelationTime(paramIndex);
 *
 * The reference transforms one (walker, parameter) series after another on one core; here the chain is handed to
 * libmcmcpp_hip.so (include/mcmcpp_hip.h, mcmcpp_hip_autocorr_times): one workgroup per series, both FFTs in LDS, the
 * reference's own butterflies, sums and divisions in the reference's order -- the result is bit-identical to the
 * restatement in oracle/ that is pinned to the reference's Detail::AutoCov (tests/test_autocorr.py).
 *
 * Two things are deliberately NOT taken over:
 *   - the reference's transferWalker adds every series onto what its scratch array holds -- the previous walker's
 *     autocovariance function, uninitialised memory for the first (AutoCorrCalc.h:239-245,307-320); here the series is
 *     what gets transformed, as the class documents;
 *   - with 0 < numWalkersToUse < numWalkers the reference selects walkers with an engine seeded from
 *     std::random_device, irreproducibly; here the subset is the evenly spaced one, floor(i * numWalkers / use).
 * No GPU, no result: failures abort with the library's message, like everything else in this facade.
 */
#ifndef MCMCPP_ANALYSIS_AUTOCORRCALC_H
#define MCMCPP_ANALYSIS_AUTOCORRCALC_H

#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../Chain/ChainStepIterator.h"
#include "../Device/HipBackend.h"

namespace MCMC
{
namespace Analysis
{
template <class ParamType>
class AutoCorrCalc
{
public:
    typedef Chain::ChainStepIterator<ParamType> IttType;

    AutoCorrCalc(int numParams, int numWalkers) : paramCount(numParams), walkerCount(numWalkers), acorrTimeList(static_cast<size_t>(numParams), ParamType(0))
    {
        assert(paramCount > 0);
        assert(walkerCount > 0);
    }
    AutoCorrCalc(const AutoCorrCalc&) = delete;
    AutoCorrCalc& operator=(const AutoCorrCalc&) = delete;

    /// numSamples: the number of steps in [start, end), as in the reference (it sizes the transform).
    void calcAutoCorrTimes(const IttType& start, const IttType& end, int numSamples, int numWalkersToUse = 0)
    {
        stepList.clear();
        for (IttType itt(start); itt != end; ++itt) stepList.push_back(*itt);
        assert(static_cast<size_t>(numSamples) == stepList.size());
        (void)numSamples;
        const int rc = mcmcpp_hip_autocorr_times(Device::HipDtype<ParamType>::value, -1, stepList.data(), static_cast<std::int64_t>(stepList.size()), walkerCount,
                                                 paramCount, numWalkersToUse == walkerCount ? 0 : numWalkersToUse, windowScaling, acorrTimeList.data(), nullptr);
        if (rc != MCMCPP_HIP_OK)
        {
            std::fprintf(stderr, "MCMCpp (MI355X): mcmcpp_hip_autocorr_times failed with code %d: %s\n", rc, mcmcpp_hip_autocorr_last_error());
            std::abort();
        }
    }

    void setAutoCorrScaleFactor(int scaleFactor = 5) { windowScaling = scaleFactor; }
    ParamType retrieveAutoCorrelationTime(int paramIndex) { return acorrTimeList[static_cast<size_t>(paramIndex)]; }

private:
    int paramCount;
    int walkerCount;
    int windowScaling = 4;
    std::vector<ParamType> acorrTimeList;
    std::vector<const void*> stepList;
};

}  // namespace Analysis
}  // namespace MCMC
#endif  // MCMCPP_ANALYSIS_AUTOCORRCALC_H

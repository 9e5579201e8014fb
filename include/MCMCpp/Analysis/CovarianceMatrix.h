/*
 * Analysis/CovarianceMatrix.h -- covariance and correlation matrix of the samples in a chain, computed on the
 * MI355X.
 *
 * Same class, constructor and methods as the reference (/root/reference/MCMCpp/Analysis/CovarianceMatrix.h:44-126):
 *
 *     MCMC::Analysis::CovarianceMatrix<double> cm(numParams, numWalkers);
 *     cm.calculateCovar(sampler.getStepIttBegin(), sampler.getStepIttEnd(), sliceInterval);
 *     cm.getCovarianceMatrixElement(i, j);  cm.getCorrelationMatrixElement(i, j);
 *
 * The reference walks the chain sample by sample and Kahan-sums x_i and x_i*x_j on one core; here the selected
 * steps are handed to libmcmcpp_hip.so (include/mcmcpp_hip.h, mcmcpp_hip_moments_*), which accumulates the same
 * sums as a rank-N update X^T X on the matrix cores and applies the reference's finalizeMatrix arithmetic
 * (CovarianceMatrix.h:178-224).  A parallel sum cannot keep the reference's order of additions, so the results
 * agree with it to rounding, not bit for bit (tests/test_moments.py states the tolerance).  Runs of steps that are
 * contiguous in the chain's memory are uploaded in one piece.  No GPU, no result: failures abort with the
 * library's message, like everything else in this facade.
 */
#ifndef MCMCPP_ANALYSIS_COVARIANCEMATRIX_H
#define MCMCPP_ANALYSIS_COVARIANCEMATRIX_H

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
class CovarianceMatrix
{
public:
    typedef Chain::ChainStepIterator<ParamType> IttType;

    CovarianceMatrix(int numParams, int numWalkers)
        : pCount(numParams), wCount(numWalkers), covarMat(static_cast<size_t>(numParams) * numParams, ParamType(0)),
          corrMat(static_cast<size_t>(numParams) * numParams, ParamType(0)), handle(nullptr)
    {
        assert(pCount > 0);
        assert(wCount > 0);
        const int rc = mcmcpp_hip_moments_create(Device::HipDtype<ParamType>::value, -1, wCount, pCount, &handle);
        if (rc != MCMCPP_HIP_OK) die("mcmcpp_hip_moments_create", rc, mcmcpp_hip_moments_last_error(nullptr));
    }
    ~CovarianceMatrix()
    {
        if (handle) mcmcpp_hip_moments_destroy(handle);
    }
    CovarianceMatrix(const CovarianceMatrix&) = delete;
    CovarianceMatrix& operator=(const CovarianceMatrix&) = delete;

    /// Uses every sliceInterval'th step of [start, end), beginning with `start` (CovarianceMatrix.h:154-173).
    void calculateCovar(IttType start, IttType end, int sliceInterval = 1)
    {
        assert(sliceInterval >= 1);
        check("mcmcpp_hip_moments_reset", mcmcpp_hip_moments_reset(handle));
        const std::int64_t stepElems = static_cast<std::int64_t>(wCount) * pCount;
        // gather runs of selected steps that lie sliceInterval steps apart in memory: one upload call per run
        ParamType* runStart = nullptr;
        ParamType* prev = nullptr;
        std::int64_t runLength = 0;
        std::int64_t used = 0;
        for (IttType itt(start); itt != end; itt += sliceInterval)
        {
            ParamType* p = *itt;
            if (runLength > 0 && p == prev + stepElems * sliceInterval)
                ++runLength;
            else
            {
                if (runLength > 0) check("mcmcpp_hip_moments_add_steps", mcmcpp_hip_moments_add_steps(handle, runStart, runLength, sliceInterval));
                runStart = p;
                runLength = 1;
            }
            prev = p;
            ++used;
        }
        if (runLength > 0) check("mcmcpp_hip_moments_add_steps", mcmcpp_hip_moments_add_steps(handle, runStart, runLength, sliceInterval));
        if (used == 0) return;  // (the reference would divide by zero here)
        check("mcmcpp_hip_moments_finish", mcmcpp_hip_moments_finish(handle, nullptr, nullptr, covarMat.data(), corrMat.data()));
    }

    ParamType getCovarianceMatrixElement(int row, int col) { return covarMat[static_cast<size_t>(row) * pCount + col]; }
    ParamType getCorrelationMatrixElement(int row, int col) { return corrMat[static_cast<size_t>(row) * pCount + col]; }

private:
    void check(const char* what, int rc) const
    {
        if (rc != MCMCPP_HIP_OK) die(what, rc, mcmcpp_hip_moments_last_error(handle));
    }
    static void die(const char* what, int rc, const char* msg)
    {
        std::fprintf(stderr, "MCMCpp (MI355X): %s failed with code %d: %s\n", what, rc, msg ? msg : "");
        std::abort();
    }

    int pCount;
    int wCount;
    std::vector<ParamType> covarMat;
    std::vector<ParamType> corrMat;
    mcmcpp_hip_moments* handle;
};

}  // namespace Analysis
}  // namespace MCMC
#endif  // MCMCPP_ANALYSIS_COVARIANCEMATRIX_H

/*
 * ref_driver.cpp -- drives the REFERENCE sampler (headers compiled from /root/reference where they
 * lie; nothing of the reference is copied into this repo) so that its outputs can pin the oracle and
 * serve as the CPU baseline.  TEST INFRASTRUCTURE ONLY; built by oracle/Makefile into
 * oracle/_ref/libmcmcpp_ref.so (git-ignored, never part of the product path).
 *
 * It instantiates the reference's own
 *   MCMC::EnsembleSampler<T, MCMC::Mover::StretchMove<T, Calculator>>          (EnsembleSampler.h:39)
 *   MCMC::ParallelEnsembleSampler<T, ...>                                      (ParallelEnsembleSampler.h:78)
 * with this repo's Calculators (include/MCMCpp/Device/Calculators.h -- any class with
 * calcLogPostProb(T*) is a legal reference Calculator) and with the reference's SkewedGaussianTwoDim.
 */
#include <chrono>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

// pcg-cpp: un-vendored submodule of the reference; the image's copy lives in namespace arrow_vendored.
#include PCG_HEADER
using arrow_vendored::pcg32;
using arrow_vendored::pcg64;

#include "EnsembleSampler.h"
#include "Movers/StretchMove.h"
#include "Movers/DifferentialEvolution.h"
#include "Movers/WalkMove.h"
#include "ParallelEnsembleSampler.h"
#include "Common/SkewedGaussian.h"
#include "Analysis/CovarianceMatrix.h"
#include "Analysis/AutoCorrCalc.h"
#include "Movers/Diagnostic/AutoRegressiveMove.h"

#include "../include/MCMCpp/Device/Calculators.h"

namespace
{
template <class T, class MoverType>
int runMover(MoverType& mover, int W, int D, int seed, const T* initPos, const T* initLogp, int nCalls,
             int stepsPerCall, int slicing, T* chainOut, long long chainCapacitySteps,
             unsigned long long* acceptedAfterCall, unsigned long long* totalAfterCall, int* storedSteps,
             double* seconds)
{
    MCMC::EnsembleSampler<T, MoverType> sampler(seed, W, D, mover);
    if (slicing > 1) sampler.setSlicingMode(true, slicing);
    sampler.setInitialWalkerPos(const_cast<T*>(initPos), const_cast<T*>(initLogp));
    bool ok = true;
    auto t0 = std::chrono::steady_clock::now();
    for (int c = 0; c < nCalls; ++c)
    {
        ok = sampler.runMCMC(stepsPerCall) && ok;
        if (acceptedAfterCall) acceptedAfterCall[c] = sampler.getAcceptedSteps();
        if (totalAfterCall) totalAfterCall[c] = sampler.getTotalSteps();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    if (storedSteps) *storedSteps = sampler.getStoredSteps();
    if (chainOut)
    {
        long long k = 0;
        auto end = sampler.getStepIttEnd();
        for (auto it = sampler.getStepIttBegin(); it != end && k < chainCapacitySteps; ++it, ++k)
            std::memcpy(chainOut + static_cast<size_t>(k) * W * D, *it, sizeof(T) * static_cast<size_t>(W) * D);
    }
    return ok ? 0 : 1;
}

template <class T, class Calc, class Dist>
int runSequential(Calc& calc, int W, int D, int seed, const T* initPos, const T* initLogp, int nCalls,
                  int stepsPerCall, int slicing, T* chainOut, long long chainCapacitySteps,
                  unsigned long long* acceptedAfterCall, unsigned long long* totalAfterCall, int* storedSteps,
                  double* seconds)
{
    typedef MCMC::Mover::StretchMove<T, Calc, Dist> MoverType;
    MoverType mover(D, seed, calc);
    return runMover<T, MoverType>(mover, W, D, seed, initPos, initLogp, nCalls, stepsPerCall, slicing, chainOut, chainCapacitySteps, acceptedAfterCall,
                                  totalAfterCall, storedSteps, seconds);
}

/* Mover::DifferentialEvolution (next row f3) through the same sequential sampler */
template <class T, class Calc>
int runDiffEvo(Calc& calc, int W, int D, int seed, const T* initPos, const T* initLogp, int nCalls,
               int stepsPerCall, int slicing, T* chainOut, long long chainCapacitySteps,
               unsigned long long* acceptedAfterCall, unsigned long long* totalAfterCall, int* storedSteps,
               double* seconds)
{
    typedef MCMC::Mover::DifferentialEvolution<T, Calc> MoverType;
    MoverType mover(D, seed, calc);
    return runMover<T, MoverType>(mover, W, D, seed, initPos, initLogp, nCalls, stepsPerCall, slicing, chainOut, chainCapacitySteps, acceptedAfterCall,
                                  totalAfterCall, storedSteps, seconds);
}

/* Mover::WalkMove with the reference test's 6 sampled walkers (test/sequential/SkewedGaussian/WalkMove/src/main.cpp:35):
 * timed only -- DESIGN.md section 7 records why this mover has no device kernel */
template <class T, class Calc>
int runWalk(Calc& calc, int W, int D, int seed, const T* initPos, const T* initLogp, int nCalls,
            int stepsPerCall, int slicing, T* chainOut, long long chainCapacitySteps,
            unsigned long long* acceptedAfterCall, unsigned long long* totalAfterCall, int* storedSteps,
            double* seconds)
{
    typedef MCMC::Mover::WalkMove<T, Calc> MoverType;
    MoverType mover(D, seed, calc, 6);
    return runMover<T, MoverType>(mover, W, D, seed, initPos, initLogp, nCalls, stepsPerCall, slicing, chainOut, chainCapacitySteps, acceptedAfterCall,
                                  totalAfterCall, storedSteps, seconds);
}

template <class T, class Calc>
int runParallel(Calc& calc, int threads, int W, int D, int seed, const T* initPos, const T* initLogp, int nSteps,
                double* seconds, double* acceptanceFraction)
{
    typedef MCMC::Mover::StretchMove<T, Calc> MoverType;
    MoverType mover(D, seed, calc);
    MCMC::ParallelEnsembleSampler<T, MoverType> sampler(seed, threads, W, D, mover);
    sampler.setInitialWalkerPos(const_cast<T*>(initPos), const_cast<T*>(initLogp));
    auto t0 = std::chrono::steady_clock::now();
    bool ok = sampler.runMCMC(nSteps);
    auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    if (acceptanceFraction) *acceptanceFraction = static_cast<double>(sampler.getAcceptanceFraction());
    return ok ? 0 : 1;
}

template <class T>
int dispatch(int calcId, int threads, int alphaCode, int W, int D, const T* params, int seed, const T* initPos, const T* initLogp,
             int nCalls, int stepsPerCall, int slicing, T* chainOut, long long chainCapacitySteps,
             unsigned long long* acc, unsigned long long* tot, int* stored, double* seconds, double* fraction)
{
#define MCMCPP_REF_GO(CALC)                                                                                        \
    if (threads <= 0 && alphaCode == 3)                                                                            \
        return runWalk<T, decltype(CALC)>(CALC, W, D, seed, initPos, initLogp, nCalls, stepsPerCall, slicing,      \
                                          chainOut, chainCapacitySteps, acc, tot, stored, seconds);                \
    if (threads <= 0 && alphaCode == 2)                                                                            \
        return runDiffEvo<T, decltype(CALC)>(CALC, W, D, seed, initPos, initLogp, nCalls, stepsPerCall, slicing,   \
                                             chainOut, chainCapacitySteps, acc, tot, stored, seconds);             \
    if (threads <= 0 && alphaCode == 1)                                                                            \
        return runSequential<T, decltype(CALC), MCMC::Utility::GwDistribution<T, 3, 2> >(                          \
            CALC, W, D, seed, initPos, initLogp, nCalls, stepsPerCall, slicing, chainOut, chainCapacitySteps, acc, \
            tot, stored, seconds);                                                                                 \
    if (threads <= 0)                                                                                              \
        return runSequential<T, decltype(CALC), MCMC::Utility::GwDistribution<T, 2, 1> >(                          \
            CALC, W, D, seed, initPos, initLogp, nCalls, stepsPerCall, slicing, chainOut, chainCapacitySteps, acc, \
            tot, stored, seconds);                                                                                 \
    return runParallel<T>(CALC, threads, W, D, seed, initPos, initLogp, nCalls * stepsPerCall, seconds, fraction)
    switch (calcId)
    {
    case MCMC::Device::IsoGaussianId:
    {
        MCMC::Device::IsoGaussian<T> c(D);
        MCMCPP_REF_GO(c);
    }
    case MCMC::Device::DenseGaussianId:
    {
        MCMC::Device::DenseGaussian<T> c(D, params);
        MCMCPP_REF_GO(c);
    }
    case MCMC::Device::RosenbrockId:
    {
        MCMC::Device::Rosenbrock<T> c(D, params[0], params[1], params[2]);
        MCMCPP_REF_GO(c);
    }
    case MCMC::Device::SkewedGaussian2DId:
    {
        SkewedGaussianTwoDim<T> c(params[0]);  // the reference's own test Calculator
        MCMCPP_REF_GO(c);
    }
    case 103:
    {
        MCMC::Device::SkewedGaussian2D<T> c(params[0]);  // this repo's restatement of it
        MCMCPP_REF_GO(c);
    }
    default: return -6;
    }
#undef MCMCPP_REF_GO
}
/* The reference's Analysis::CovarianceMatrix over a chain given as [n_steps][W][D]: the steps are put into a
 * reference Chain (Chain::storeWalker / incrementChainStep, as the Walkers do) and walked with its own iterators. */
template <class T>
static int refCovariance(const T* steps, long long n_steps, int W, int D, int slice, T* cov, T* corr)
{
    MCMC::Chain::Chain<T> chain(W, D, static_cast<unsigned long long>(n_steps + 1) * W * D * sizeof(T) + (1ull << 20));
    for (long long s = 0; s < n_steps; ++s)
    {
        for (int w = 0; w < W; ++w) chain.storeWalker(w, const_cast<T*>(steps + (static_cast<size_t>(s) * W + w) * D));
        chain.incrementChainStep();
    }
    MCMC::Analysis::CovarianceMatrix<T> cm(D, W);
    cm.calculateCovar(chain.getStepIteratorBegin(), chain.getStepIteratorEnd(), slice);
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j)
        {
            cov[static_cast<size_t>(i) * D + j] = cm.getCovarianceMatrixElement(i, j);
            corr[static_cast<size_t>(i) * D + j] = cm.getCorrelationMatrixElement(i, j);
        }
    return 0;
}

/* The reference's Analysis::AutoCorrCalc over a chain given as [n_steps][W][D], all walkers used.  Its scratch array
 * comes from new[] uninitialised and transferWalker adds onto it (AutoCorrCalc.h:239-245,307-320): results are only
 * defined when that memory happens to be zero, which holds for arrays big enough to come straight from mmap
 * (n_steps >= 32768 values here). */
template <class T>
static int refAutoCorr(const T* steps, long long n_steps, int W, int D, int window_scaling, T* out)
{
    MCMC::Chain::Chain<T> chain(W, D, static_cast<unsigned long long>(n_steps + 1) * W * D * sizeof(T) + (1ull << 20));
    for (long long s = 0; s < n_steps; ++s)
    {
        for (int w = 0; w < W; ++w) chain.storeWalker(w, const_cast<T*>(steps + (static_cast<size_t>(s) * W + w) * D));
        chain.incrementChainStep();
    }
    MCMC::Analysis::AutoCorrCalc<T> ac(D, W);
    ac.setAutoCorrScaleFactor(window_scaling);
    ac.calcAutoCorrTimes(chain.getStepIteratorBegin(), chain.getStepIteratorEnd(), static_cast<int>(n_steps));
    for (int i = 0; i < D; ++i) out[i] = ac.retrieveAutoCorrelationTime(i);
    return 0;
}

}  // namespace

extern "C"
{
/* threads <= 0: MCMC::EnsembleSampler; threads >= 1: MCMC::ParallelEnsembleSampler (timing only:
 * it is non-deterministic above one thread, ParallelEnsembleSampler.h:71-76).
 * alpha_code 0: StretchMove's default GwDistribution<T,2,1>; 1: GwDistribution<T,3,2>; 2: Mover::DifferentialEvolution;
 * 3: Mover::WalkMove with 6 sampled walkers (timing only).
 * dtype 0 = double, 1 = float.  chain_out holds up to chain_capacity_steps steps of W*D values
 * (chain step 0 is the initial placement, EnsembleSampler.h:228-229). */
int ref_run(int dtype, int threads, int alpha_code, int W, int D, int calc_id, const void* params, int seed, const void* init_pos,
            const void* init_logp, int n_calls, int steps_per_call, int slicing, void* chain_out,
            long long chain_capacity_steps, unsigned long long* accepted_after_call,
            unsigned long long* total_after_call, int* stored_steps, double* seconds, double* fraction)
{
    if (dtype == 0)
        return dispatch<double>(calc_id, threads, alpha_code, W, D, static_cast<const double*>(params), seed,
                                static_cast<const double*>(init_pos), static_cast<const double*>(init_logp), n_calls,
                                steps_per_call, slicing, static_cast<double*>(chain_out), chain_capacity_steps,
                                accepted_after_call, total_after_call, stored_steps, seconds, fraction);
    return dispatch<float>(calc_id, threads, alpha_code, W, D, static_cast<const float*>(params), seed,
                           static_cast<const float*>(init_pos), static_cast<const float*>(init_logp), n_calls,
                           steps_per_call, slicing, static_cast<float*>(chain_out), chain_capacity_steps,
                           accepted_after_call, total_after_call, stored_steps, seconds, fraction);
}

int ref_chain_covariance(int dtype, const void* steps, long long n_steps, int W, int D, int slice, void* cov, void* corr)
{
    if (dtype == 0)
        return refCovariance<double>(static_cast<const double*>(steps), n_steps, W, D, slice, static_cast<double*>(cov), static_cast<double*>(corr));
    return refCovariance<float>(static_cast<const float*>(steps), n_steps, W, D, slice, static_cast<float*>(cov), static_cast<float*>(corr));
}

/* Analysis::Detail::AutoCov::calcNormAutoCov on one series, in place */
int ref_norm_autocov(int dtype, void* chain, double avg, int n)
{
    if (dtype == 0)
    {
        MCMC::Analysis::Detail::AutoCov<double> ac;
        ac.calcNormAutoCov(static_cast<double*>(chain), avg, n);
    }
    else
    {
        MCMC::Analysis::Detail::AutoCov<float> ac;
        ac.calcNormAutoCov(static_cast<float*>(chain), static_cast<float>(avg), n);
    }
    return 0;
}

int ref_autocorr_times(int dtype, const void* steps, long long n_steps, int W, int D, int window_scaling, void* out)
{
    if (dtype == 0) return refAutoCorr<double>(static_cast<const double*>(steps), n_steps, W, D, window_scaling, static_cast<double*>(out));
    return refAutoCorr<float>(static_cast<const float*>(steps), n_steps, W, D, window_scaling, static_cast<float*>(out));
}

/* The reference's AutoCorrCalc known-answer test (test/sequential/AcTime/src/main.cpp:23-82) with its statements in its
 * order: mover, sampler, initial points from the test's own mover object, runMCMC, AutoCorrCalc over all stored steps.
 * chain_out: NULL or [n_steps + 1][W][D]; times_out: [D]. */
int ref_actime_test(int run_number, int W, int D, int n_steps, const double* offsets, const double* phis, const double* vars, double* chain_out,
                    double* times_out)
{
    MCMC::Mover::AutoRegressiveMove<double> mover(D, offsets, phis, vars);
    MCMC::EnsembleSampler<double, MCMC::Mover::AutoRegressiveMove<double> > sampler(run_number, W, D, mover);
    std::vector<double> init(static_cast<size_t>(W) * D), aux(W);
    mover.getInitialPoints(init.data(), aux.data(), W);
    sampler.setInitialWalkerPos(init.data(), aux.data());
    sampler.runMCMC(n_steps);
    if (chain_out)
    {
        size_t k = 0;
        for (auto it = sampler.getStepIttBegin(); it != sampler.getStepIttEnd(); ++it, ++k)
            std::memcpy(chain_out + k * W * D, *it, sizeof(double) * static_cast<size_t>(W) * D);
    }
    MCMC::Analysis::AutoCorrCalc<double> ac(D, W);
    ac.calcAutoCorrTimes(sampler.getStepIttBegin(), sampler.getStepIttEnd(), sampler.getStoredSteps());
    for (int i = 0; i < D; ++i) times_out[i] = ac.retrieveAutoCorrelationTime(i);
    return sampler.getStoredSteps();
}

/* Initial walker placement of the reference's SkewedGaussian/StretchMove test
 * (test/sequential/SkewedGaussian/StretchMove/src/main.cpp:141-205): pcg32(extraRunNumber) feeding four
 * std::normal_distribution<double> in a 16-walker pattern.  Restated compactly (the draw order is
 * row-major over the 4x4 pattern) so that the fixture can be generated; the arrays it returns are
 * stored in tests/golden/ because libstdc++'s normal_distribution is implementation-defined. */
void ref_skewed_initial_values(double* init_vals, double* aux_vals, int num_walkers, double eps, int extra_run_number)
{
    const double averages[2] = {3.1, -3.5};
    const double deviations[2] = {3.3, 4.1};
    pcg32 engine(extra_run_number);
    std::normal_distribution<double> nd[4] = {std::normal_distribution<double>(averages[0], deviations[0]),
                                              std::normal_distribution<double>(averages[1], deviations[1]),
                                              std::normal_distribution<double>(averages[0], deviations[1]),
                                              std::normal_distribution<double>(averages[1], deviations[0])};
    SkewedGaussianTwoDim<double> likelihood(eps);
    for (int i = 0; i < num_walkers; i += 16)
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b)
            {
                const int w = i + 4 * a + b;
                init_vals[2 * w] = nd[a](engine);
                init_vals[2 * w + 1] = nd[b](engine);
                aux_vals[w] = likelihood.calcLogPostProb(init_vals + 2 * w);
            }
}
}

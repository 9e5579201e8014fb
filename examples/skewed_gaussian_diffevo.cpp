// Example: the reference's SkewedGaussian/DiffEvo demo, written against this repository's headers.
//
// What the reference's program does (/root/reference/test/sequential/SkewedGaussian/DiffEvo/src/main.cpp):
// 320 walkers sample the two-dimensional skewed Gaussian logp = -((x/2-y)^2/eps + (x/2+y)^2)/2, eps = 0.13,
// storing every 10th of 400 thousand ensemble steps (Mover::DifferentialEvolution), then drop 20 stored steps of burn-in and report the
// acceptance fraction and the sample covariance (analytic: [[1+eps, (1-eps)/2], [(1-eps)/2, (1+eps)/4]]).
// The same user-level calls run here on one MI355X.
//
//   g++ -std=c++11 -O2 -I include/MCMCpp -I include examples/skewed_gaussian_diffevo.cpp
//       -L mcmcpp_amd -lmcmcpp_hip -Wl,-rpath,$PWD/mcmcpp_amd -o skewed
//   ./skewed [stored_steps] [initial_values.bin]     (the optional file holds 320*2 positions as doubles)
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "Device/Calculators.h"
#include "EnsembleSampler.h"
#include "Movers/DifferentialEvolution.h"

int main(int argc, char** argv)
{
    typedef MCMC::Device::SkewedGaussian2D<double> Likelihood;
    typedef MCMC::Mover::DifferentialEvolution<double, Likelihood> Mover;
    const int runNumber = 0, numWalkers = 320, numParams = 2;
    const int numSteps = argc > 1 ? std::atoi(argv[1]) : 40019;
    const double eps = 0.13;

    Likelihood likelihood(eps);
    Mover mover(numParams, runNumber, likelihood);
    MCMC::EnsembleSampler<double, Mover> sampler(runNumber, numWalkers, numParams, mover);
    sampler.setSlicingMode(true, 10);

    std::vector<double> initVals(numWalkers * numParams), auxVals(numWalkers);
    if (argc > 2)
    {
        FILE* fp = std::fopen(argv[2], "rb");
        if (!fp || std::fread(initVals.data(), sizeof(double), initVals.size(), fp) != initVals.size())
        {
            std::fprintf(stderr, "cannot read %s\n", argv[2]);
            return 2;
        }
        std::fclose(fp);
    }
    else
    {
        std::mt19937_64 engine(53);
        std::normal_distribution<double> spread(0.0, 3.5);
        for (double& v : initVals) v = spread(engine);
    }
    for (int w = 0; w < numWalkers; ++w) auxVals[w] = likelihood.calcLogPostProb(&initVals[w * numParams]);
    sampler.setInitialWalkerPos(initVals.data(), auxVals.data());

    const bool complete = sampler.runMCMC(numSteps);
    std::printf("%s\n", complete ? "Sampling completed normally" : "Sampling finished when chain ran out of space");
    sampler.sliceAndBurnChain(1, 20);
    std::printf("Acceptance Fraction: %llu/%llu | %g\n", sampler.getAcceptedSteps(), sampler.getTotalSteps(),
                sampler.getAcceptanceFraction());

    // sample covariance over every stored parameter set
    double n = 0, mx = 0, my = 0, sxx = 0, sxy = 0, syy = 0;
    for (auto it = sampler.getParamSetIttBegin(); it != sampler.getParamSetIttEnd(); ++it)
    {
        const double x = (*it)[0], y = (*it)[1];
        n += 1;
        mx += x;
        my += y;
        sxx += x * x;
        sxy += x * y;
        syy += y * y;
    }
    mx /= n;
    my /= n;
    std::printf("Stored steps: %d\n", sampler.getStoredSteps());
    std::printf("Covariance matrix\n%g, %g\n%g, %g\n", sxx / n - mx * mx, sxy / n - mx * my, sxy / n - mx * my, syy / n - my * my);
    std::printf("Analytic\n%g, %g\n%g, %g\n", 1 + eps, (1 - eps) / 2, (1 - eps) / 2, (1 + eps) / 4);
    return 0;
}

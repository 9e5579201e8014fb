// covariance_facade.cpp -- the facade's Analysis::CovarianceMatrix (device) against the oracle's restatement of the
// reference's (CPU, linked here: tests may use the oracle), on a chain the facade's own sampler produced.
//   usage: covariance_facade            (needs an MI355X)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "Analysis/CovarianceMatrix.h"
#include "Device/Calculators.h"
#include "EnsembleSampler.h"
#include "Movers/StretchMove.h"

extern "C" int so_chain_covariance(int dtype, const void* steps, long long n_steps, int walkers, int dims, int slice, void* mean, void* cov,
                                   void* corr);

template <class T>
static int run(int W, int D, int steps, int slice, double tol)
{
    typedef MCMC::Device::Rosenbrock<T> Target;
    typedef MCMC::Mover::StretchMove<T, Target> Mover;
    Target target(D, T(1), T(100), T(0.05));
    Mover mover(D, 11, target);
    MCMC::EnsembleSampler<T, Mover> sampler(11, W, D, mover);
    std::vector<T> pos(static_cast<size_t>(W) * D), aux(W);
    unsigned long long s = 12345;
    for (size_t k = 0; k < pos.size(); ++k)
    {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        pos[k] = T(((s >> 11) * (1.0 / 9007199254740992.0)) * 4.0 - 2.0);
    }
    for (int w = 0; w < W; ++w) aux[w] = target.calcLogPostProb(&pos[static_cast<size_t>(w) * D]);
    sampler.setInitialWalkerPos(pos.data(), aux.data());
    sampler.runMCMC(steps);

    MCMC::Analysis::CovarianceMatrix<T> cm(D, W);
    cm.calculateCovar(sampler.getStepIttBegin(), sampler.getStepIttEnd(), slice);

    // the same steps, gathered through the iterators, into the oracle
    std::vector<T> all;
    long long n = 0;
    for (auto it = sampler.getStepIttBegin(); it != sampler.getStepIttEnd(); ++it, ++n) all.insert(all.end(), *it, *it + static_cast<size_t>(W) * D);
    std::vector<T> cov(static_cast<size_t>(D) * D), corr(static_cast<size_t>(D) * D);
    if (so_chain_covariance(sizeof(T) == 8 ? 0 : 1, all.data(), n, W, D, slice, nullptr, cov.data(), corr.data()) != 0) return 1;
    int bad = 0;
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j)
        {
            const double scale = std::sqrt(std::fabs(static_cast<double>(cov[i * D + i]) * cov[j * D + j]));
            if (std::fabs(cm.getCovarianceMatrixElement(i, j) - cov[i * D + j]) > tol * scale) ++bad;
            if (std::fabs(cm.getCorrelationMatrixElement(i, j) - corr[i * D + j]) > tol) ++bad;
        }
    if (bad) std::printf("FAIL W=%d D=%d steps=%d slice=%d: %d elements out of tolerance\n", W, D, steps, slice, bad);
    return bad;
}

int main()
{
    int bad = 0;
    bad += run<double>(64, 6, 150, 1, 1e-10);
    bad += run<double>(200, 33, 40, 3, 1e-10);
    bad += run<float>(96, 5, 120, 2, 2e-4);
    if (!bad) std::printf("covariance_facade OK\n");
    return bad ? 1 : 0;
}

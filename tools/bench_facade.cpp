// bench_facade.cpp -- the C2 workload (16 384 walkers x 32 dims, correlated Gaussian) through the header-only C++ facade:
// walker-steps/s of ParallelEnsembleSampler::runMCMC with and without a PostStepAction, at the bench's slicing interval
// (100) and at interval 1 (every ensemble step stored).  The reference calls the action once per ensemble step
// (/root/reference/MCMCpp/EnsembleSampler.h:356-359); here it runs on the calling thread beside the device
// (include/MCMCpp/Device/SamplerCore.h).
//   g++ -std=c++11 -O2 -I include/MCMCpp -I include tools/bench_facade.cpp -L mcmcpp_amd -lmcmcpp_hip -Wl,-rpath,$PWD/mcmcpp_amd -o tools/bench_facade.bin
#include <chrono>
#include <cstdio>
#include <vector>

#include "Device/Calculators.h"
#include "Movers/StretchMove.h"
#include "ParallelEnsembleSampler.h"

using namespace MCMC;

struct LookAtLastStep  // a PostStepAction that reads something of the chain each time it is called
{
    double sum = 0.0;
    long calls = 0;
    void performAction(const Chain::ChainStepIterator<double>& start, const Chain::ChainStepIterator<double>& end)
    {
        ++calls;
        if (end.stepIndex() > start.stepIndex())
        {
            Chain::ChainStepIterator<double> last = end;
            --last;
            sum += (*last)[0];
        }
    }
};

static double splitmix_unit(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

template <class Sampler>
static double timed(Sampler& s, int calls, int storedPerCall)
{
    s.runMCMC(storedPerCall);  // warm-up (graphs, pinned chain blocks)
    const auto t0 = std::chrono::steady_clock::now();
    for (int c = 0; c < calls; ++c) s.runMCMC(storedPerCall);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

int main()
{
    const int W = 16384, D = 32;
    std::vector<double> P((size_t)D * D, 0.0), pos((size_t)W * D), logp(W);
    const double rho = 0.5, d = 1.0 - rho * rho;
    for (int i = 0; i < D; ++i)
    {
        P[(size_t)i * D + i] = ((i == 0 || i == D - 1) ? 1.0 : 1.0 + rho * rho) / d;
        if (i + 1 < D) P[(size_t)i * D + i + 1] = P[(size_t)(i + 1) * D + i] = -rho / d;
    }
    for (size_t k = 0; k < pos.size(); ++k) pos[k] = 4.0 * splitmix_unit(k) - 2.0;
    typedef Device::DenseGaussian<double> Calc;
    typedef Mover::StretchMove<double, Calc> MoverType;
    Calc calc(D, P.data());
    for (int w = 0; w < W; ++w) logp[w] = calc.calcLogPostProb(&pos[(size_t)w * D]);
    MoverType mover(D, 0, calc);
    const unsigned long long budget = 6ULL << 30;  // chain budget: room for every stored step of the runs below
    for (int interval : {100, 1})
    {
        const int storedPerCall = interval == 100 ? 20 : 200, calls = interval == 100 ? 20 : 4;
        double plain, acted;
        long actionCalls;
        {
            ParallelEnsembleSampler<double, MoverType> s(0, 8, W, D, mover, budget);
            s.setSamplingMode(interval, 0);
            s.setInitialWalkerPos(pos.data(), logp.data());
            plain = timed(s, calls, storedPerCall);
        }
        {
            LookAtLastStep action;
            ParallelEnsembleSampler<double, MoverType, LookAtLastStep> s(0, 8, W, D, mover, budget, &action);
            s.setSamplingMode(interval, 0);
            s.setInitialWalkerPos(pos.data(), logp.data());
            acted = timed(s, calls, storedPerCall);
            actionCalls = action.calls;
        }
        const double ws = (double)W * interval * storedPerCall * calls;
        std::printf("interval %3d: facade without action %.3e walker-steps/s, with a PostStepAction %.3e (%ld calls): slowdown x%.3f\n", interval, ws / plain,
                    ws / acted, actionCalls, acted / plain);
    }
    return 0;
}

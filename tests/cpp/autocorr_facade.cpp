// autocorr_facade.cpp -- the facade's Analysis::AutoCorrCalc (device) against the oracle's restatement of the
// reference's (CPU, linked here: tests may use the oracle), on a chain the facade's own sampler produced.  Bit for bit.
//   usage: autocorr_facade            (needs an MI355X)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "Analysis/AutoCorrCalc.h"
#include "Device/Calculators.h"
#include "EnsembleSampler.h"
#include "Movers/StretchMove.h"

extern "C" int so_autocorr_times(int dtype, const void* steps, int n_steps, int walkers, int dims, int window_scaling, int emulate_defect, void* out,
                                 void* functions);

template <class T>
static int run(int W, int D, int steps, int scale, int use)
{
    typedef MCMC::Device::IsoGaussian<T> Target;
    typedef MCMC::Mover::StretchMove<T, Target> Mover;
    Target target(D);
    Mover mover(D, 3, target);
    MCMC::EnsembleSampler<T, Mover> sampler(3, W, D, mover);
    std::vector<T> pos(static_cast<size_t>(W) * D), aux(W);
    unsigned long long s = 4321;
    for (size_t k = 0; k < pos.size(); ++k)
    {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        pos[k] = T(((s >> 11) * (1.0 / 9007199254740992.0)) * 4.0 - 2.0);
    }
    for (int w = 0; w < W; ++w) aux[w] = target.calcLogPostProb(&pos[static_cast<size_t>(w) * D]);
    sampler.setInitialWalkerPos(pos.data(), aux.data());
    sampler.runMCMC(steps);

    int n = 0;
    for (auto it = sampler.getStepIttBegin(); it != sampler.getStepIttEnd(); ++it) ++n;
    MCMC::Analysis::AutoCorrCalc<T> ac(D, W);
    ac.setAutoCorrScaleFactor(scale);
    ac.calcAutoCorrTimes(sampler.getStepIttBegin(), sampler.getStepIttEnd(), n, use);

    // the same steps (of the walkers used), gathered through the iterators, into the oracle
    const int used = use == 0 ? W : use;
    std::vector<T> all;
    for (auto it = sampler.getStepIttBegin(); it != sampler.getStepIttEnd(); ++it)
        for (int i = 0; i < used; ++i)
        {
            const T* row = *it + static_cast<size_t>((static_cast<long long>(i) * W) / used) * D;
            all.insert(all.end(), row, row + D);
        }
    std::vector<T> want(D);
    if (so_autocorr_times(sizeof(T) == 8 ? 0 : 1, all.data(), n, used, D, scale, 0, want.data(), nullptr) != 0) return 1;
    int bad = 0;
    for (int p = 0; p < D; ++p)
    {
        const T got = ac.retrieveAutoCorrelationTime(p);
        if (std::memcmp(&got, &want[p], sizeof(T)) != 0)
        {
            ++bad;
            std::printf("  parameter %d: %.17g, oracle %.17g\n", p, static_cast<double>(got), static_cast<double>(want[p]));
        }
    }
    if (bad) std::printf("FAIL W=%d D=%d steps=%d: %d times differ\n", W, D, n, bad);
    else std::printf("  W=%d D=%d %d steps: tau[0] = %.4f\n", W, D, n, static_cast<double>(ac.retrieveAutoCorrelationTime(0)));
    return bad;
}

int main()
{
    int bad = 0;
    bad += run<double>(64, 4, 700, 4, 0);
    bad += run<double>(100, 7, 300, 5, 20);
    bad += run<float>(96, 5, 500, 4, 0);
    if (!bad) std::printf("autocorr_facade OK\n");
    return bad ? 1 : 0;
}

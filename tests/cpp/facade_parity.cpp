// End-to-end test of the header-only facade (include/MCMCpp) against a golden fixture produced by the
// reference: the same user code one would write against the reference's EnsembleSampler /
// ParallelEnsembleSampler + StretchMove, linked against libmcmcpp_hip.so instead.
//
//   facade_parity <fixture.bin> [de]        (de: the fixture is a Mover::DifferentialEvolution run)
// fixture: int32 W, D, steps, slicing, calc, nparams, nkept, dtype (0 = double, 1 = float); T params[nparams];
//          T pos[W*D]; T logp[W]; int32 kept_step[nkept]; T kept[nkept][W*D]; uint64 accepted_total, total_steps
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "Device/Calculators.h"
#include "EnsembleSampler.h"
#include "Movers/DifferentialEvolution.h"
#include "Movers/StretchMove.h"
#include "ParallelEnsembleSampler.h"

using namespace MCMC;

static int failures = 0;
#define CHECK(cond)                                                     \
    do                                                                  \
    {                                                                   \
        if (!(cond))                                                    \
        {                                                               \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                 \
        }                                                               \
    } while (0)

template <class T>
struct Fixture
{
    int W, D, steps, slicing, calc, nparams, nkept;
    std::vector<T> params, pos, logp, kept;
    std::vector<int> keptStep;
    unsigned long long acceptedTotal, totalSteps;
};

static int peekDtype(const char* path)
{
    FILE* fp = std::fopen(path, "rb");
    if (!fp)
    {
        std::perror(path);
        std::exit(2);
    }
    int hdr[8];
    if (std::fread(hdr, sizeof(int), 8, fp) != 8) std::exit(2);
    std::fclose(fp);
    return hdr[7];
}

template <class T>
static Fixture<T> load(const char* path)
{
    Fixture<T> f;
    FILE* fp = std::fopen(path, "rb");
    if (!fp)
    {
        std::perror(path);
        std::exit(2);
    }
    int hdr[8];
    if (std::fread(hdr, sizeof(int), 8, fp) != 8) std::exit(2);
    f.W = hdr[0], f.D = hdr[1], f.steps = hdr[2], f.slicing = hdr[3], f.calc = hdr[4], f.nparams = hdr[5], f.nkept = hdr[6];
    f.params.resize(f.nparams);
    f.pos.resize((size_t)f.W * f.D);
    f.logp.resize(f.W);
    f.keptStep.resize(f.nkept);
    f.kept.resize((size_t)f.nkept * f.W * f.D);
    size_t ok = 1;
    if (f.nparams) ok &= std::fread(f.params.data(), sizeof(T), f.nparams, fp) == (size_t)f.nparams;
    ok &= std::fread(f.pos.data(), sizeof(T), f.pos.size(), fp) == f.pos.size();
    ok &= std::fread(f.logp.data(), sizeof(T), f.logp.size(), fp) == f.logp.size();
    ok &= std::fread(f.keptStep.data(), 4, f.nkept, fp) == (size_t)f.nkept;
    ok &= std::fread(f.kept.data(), sizeof(T), f.kept.size(), fp) == f.kept.size();
    ok &= std::fread(&f.acceptedTotal, 8, 1, fp) == 1;
    ok &= std::fread(&f.totalSteps, 8, 1, fp) == 1;
    std::fclose(fp);
    if (!ok) std::exit(2);
    return f;
}

// a PostStepAction (reference concept: Utility/NoAction.h) that counts its calls and looks at the chain
template <class T>
struct CountingAction
{
    long calls = 0;
    long lastSeenSteps = 0;
    std::vector<long> seen;  // stored steps visible at every call
    void performAction(const Chain::ChainStepIterator<T>& start, const Chain::ChainStepIterator<T>& end)
    {
        ++calls;
        lastSeenSteps = end.stepIndex() - start.stepIndex();
        seen.push_back(lastSeenSteps);
    }
};

template <class Sampler, class T>
static void checkChain(Sampler& s, const Fixture<T>& f)
{
    CHECK(s.getStoredSteps() == f.steps + 1);
    // step 0 is the initial placement
    auto it = s.getStepIttBegin();
    CHECK(std::memcmp(*it, f.pos.data(), sizeof(T) * f.pos.size()) == 0);
    for (int k = 0; k < f.nkept; ++k)
    {
        auto jt = s.getStepIttBegin();
        jt += f.keptStep[k];
        CHECK(std::memcmp(*jt, f.kept.data() + (size_t)k * f.W * f.D, sizeof(T) * f.W * f.D) == 0);
    }
    // the parameter-set view walks the same memory walker by walker
    auto pit = s.getParamSetIttBegin();
    for (int w = 0; w < f.W; ++w, ++pit) CHECK(std::memcmp(*pit, f.pos.data() + (size_t)w * f.D, sizeof(T) * f.D) == 0);
    long sets = 0;
    for (auto p = s.getParamSetIttBegin(); p != s.getParamSetIttEnd(); ++p) ++sets;
    CHECK(sets == (long)(f.steps + 1) * f.W);
}

struct UseStretch
{
    template <class T, class Calc>
    struct Of
    {
        typedef Mover::StretchMove<T, Calc> type;
    };
    static const bool drawsVary = false;
};
struct UseDiffEvo
{
    template <class T, class Calc>
    struct Of
    {
        typedef Mover::DifferentialEvolution<T, Calc> type;
    };
    static const bool drawsVary = true;  // the mover throws draws away: the redraw counter is not an error count
};

template <class Sel, class T, class Calc>
static void runCase(const Fixture<T>& f, Calc calc)
{
    typedef typename Sel::template Of<T, Calc>::type MoverType;
    std::vector<T> pos(f.pos), logp(f.logp);
    // the initial auxValues are what user code computes with the Calculator (reference test main.cpp:141-205)
    for (int w = 0; w < f.W; ++w) CHECK(calc.calcLogPostProb(pos.data() + (size_t)w * f.D) == f.logp[w]);
    {
        MoverType mover(f.D, 0, calc);
        EnsembleSampler<T, MoverType> sampler(0, f.W, f.D, mover);
        if (f.slicing > 1) sampler.setSlicingMode(true, f.slicing);
        sampler.setInitialWalkerPos(pos.data(), logp.data());
        CHECK(sampler.runMCMC(f.steps / 2));
        CHECK(sampler.runMCMC(f.steps - f.steps / 2));
        checkChain(sampler, f);
        CHECK(sampler.getAcceptedSteps() == f.acceptedTotal);
        CHECK(sampler.getTotalSteps() == f.totalSteps);
        CHECK(sampler.getAcceptanceFraction() == (T)f.acceptedTotal / (T)f.totalSteps);
        std::uint64_t ties = 1, redraws = 1;
        sampler.diagnostics(&ties, &redraws);
        CHECK(ties == 0 && (Sel::drawsVary || redraws == 0));
        // reset keeps the walkers, forgets chain and counters (EnsembleSampler.h:312-322)
        std::vector<T> now((size_t)f.W * f.D);
        sampler.currentState(now.data(), nullptr, nullptr);
        sampler.reset();
        CHECK(sampler.getStoredSteps() == 0 && sampler.getAcceptedSteps() == 0);
        sampler.storeCurrentWalkerPositions();
        CHECK(sampler.getStoredSteps() == 1);
        CHECK(std::memcmp(*sampler.getStepIttBegin(), now.data(), sizeof(T) * now.size()) == 0);
        CHECK(sampler.runMCMC(3));
        CHECK(sampler.getTotalSteps() == (unsigned long long)f.W * 3 * (f.slicing > 1 ? f.slicing : 1));
        sampler.sliceAndBurnChain(2, 1);
        CHECK(sampler.getStoredSteps() == 2);
    }
    {
        // the parallel facade follows the same (sequential, reproducible) trajectory
        MoverType mover(f.D, 0, calc);
        CountingAction<T> action;
        ParallelEnsembleSampler<T, MoverType, CountingAction<T> > sampler(0, 8, f.W, f.D, mover, 2147483648ULL, &action);
        sampler.setSamplingMode(f.slicing, 0);
        sampler.setInitialWalkerPos(pos.data(), logp.data());
        CHECK(sampler.runMCMC(f.steps));
        checkChain(sampler, f);
        CHECK(sampler.getAcceptedSteps() == f.acceptedTotal);
        CHECK(sampler.getAcceptanceFraction() == (T)f.acceptedTotal / (T)f.totalSteps);
        CHECK(action.calls == (long)f.steps * f.slicing);  // once per ensemble step (EnsembleSampler.h:356-359)
        // ... and BEFORE the chain moves on to the step being made (EnsembleSampler.h:291-293): the calls of stored step k
        // (counting the initial placement as step 0) see k stored steps
        CHECK(action.lastSeenSteps == f.steps);
        bool ordered = action.seen.size() == (size_t)f.steps * f.slicing;
        for (size_t i = 0; ordered && i < action.seen.size(); ++i) ordered = action.seen[i] == 1 + (long)(i / (size_t)f.slicing);
        CHECK(ordered);
    }
    if (!Sel::drawsVary)
    {
        // the same ensemble through the split path: one rank of an RCCL communicator (all a one-GPU box has); several
        // devices differ only in the number of ranks
        MoverType mover(f.D, 0, calc);
        Device::Placement where;
        where.splitEnsemble = true;
        ParallelEnsembleSampler<T, MoverType> sampler(0, 8, f.W, f.D, mover, 2147483648ULL, nullptr, where);
        CHECK(sampler.deviceCount() == 1);
        sampler.setSamplingMode(f.slicing, 0);
        sampler.setInitialWalkerPos(pos.data(), logp.data());
        CHECK(sampler.runMCMC(f.steps / 2));
        CHECK(sampler.runMCMC(f.steps - f.steps / 2));
        checkChain(sampler, f);
        CHECK(sampler.getAcceptedSteps() == f.acceptedTotal);
        std::uint64_t ties = 1, redraws = 1;
        sampler.diagnostics(&ties, &redraws);
        CHECK(ties == 0 && redraws == 0);
    }
    if (!Sel::drawsVary && std::getenv("FACADE_PARITY_PLACEMENT_RANKS"))
    {
        // one ensemble split over several ranks named by an explicit placement (tests/test_split_loopback.py: all on device 0,
        // exchanging through the loop-back collective library), with a PostStepAction watching the chain
        const int G = std::atoi(std::getenv("FACADE_PARITY_PLACEMENT_RANKS"));
        MoverType mover(f.D, 0, calc);
        CountingAction<T> action;
        Device::Placement where(std::vector<int>((size_t)G, 0));
        ParallelEnsembleSampler<T, MoverType, CountingAction<T> > sampler(0, 8, f.W, f.D, mover, 2147483648ULL, &action, where);
        CHECK(sampler.deviceCount() == G);
        sampler.setSamplingMode(f.slicing, 0);
        sampler.setInitialWalkerPos(pos.data(), logp.data());
        CHECK(sampler.runMCMC(f.steps / 3));
        CHECK(sampler.runMCMC(f.steps - f.steps / 3));
        checkChain(sampler, f);
        CHECK(sampler.getAcceptedSteps() == f.acceptedTotal);
        CHECK(sampler.getAcceptanceFraction() == (T)f.acceptedTotal / (T)f.totalSteps);
        CHECK(action.calls == (long)f.steps * f.slicing);
        std::uint64_t ties = 1, redraws = 1;
        sampler.diagnostics(&ties, &redraws);
        CHECK(ties == 0 && redraws == 0);
        std::printf("placement of %d ranks checked\n", G);
    }
    {
        // chain budget: room for 5 steps only -> runMCMC reports false when it fills (EnsembleSampler.h:293)
        MoverType mover(f.D, 0, calc);
        EnsembleSampler<T, MoverType> sampler(0, f.W, f.D, mover, 5ULL * f.W * f.D * sizeof(T));
        sampler.setInitialWalkerPos(pos.data(), logp.data());
        CHECK(sampler.runMCMC(3));
        CHECK(!sampler.runMCMC(10));
        CHECK(sampler.getStoredSteps() == 5);
    }
}

template <class Sel, class T>
static int runFixture(const char* path)
{
    const Fixture<T> f = load<T>(path);
    switch (f.calc)
    {
    case Device::IsoGaussianId: runCase<Sel, T>(f, Device::IsoGaussian<T>(f.D)); break;
    case Device::DenseGaussianId: runCase<Sel, T>(f, Device::DenseGaussian<T>(f.D, f.params.data())); break;
    case Device::RosenbrockId: runCase<Sel, T>(f, Device::Rosenbrock<T>(f.D, f.params[0], f.params[1], f.params[2])); break;
    case Device::SkewedGaussian2DId: runCase<Sel, T>(f, Device::SkewedGaussian2D<T>(f.params[0])); break;
    default: std::printf("unknown calculator %d\n", f.calc); return 2;
    }
    if (failures == 0)
        std::printf("facade_parity OK (%d walkers x %d params, %d stored steps, slicing %d, %s)\n", f.W, f.D, f.steps, f.slicing,
                    sizeof(T) == 8 ? "double" : "float");
    return failures == 0 ? 0 : 1;
}

int main(int argc, char** argv)
{
    if (argc < 2)
    {
        std::printf("usage: facade_parity fixture.bin\n");
        return 2;
    }
    const bool de = argc > 2 && std::strcmp(argv[2], "de") == 0;
    if (de) return peekDtype(argv[1]) == 0 ? runFixture<UseDiffEvo, double>(argv[1]) : runFixture<UseDiffEvo, float>(argv[1]);
    return peekDtype(argv[1]) == 0 ? runFixture<UseStretch, double>(argv[1]) : runFixture<UseStretch, float>(argv[1]);
}

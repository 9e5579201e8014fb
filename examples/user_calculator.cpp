// Example, host side: a user-defined Calculator run through the reference's own template surface on one MI355X.
//
// What a user of the reference writes (/root/reference/MCMCpp/Movers/StretchMove.h:42-54: any class with
// `ParamType calcLogPostProb(ParamType*)` is a Calculator) stays as it is; two things are added:
//   1. the class names the device functor that evaluates it inside the kernels (hipCalcId >= 1000, hipParams,
//      hipParamCount) -- examples/user_calculator_device.hip, compiled with hipcc into libuser_calculator.so;
//   2. the program registers that functor once (mcmcpp_hip_register_calculator).
//
//   g++ -std=c++11 -O2 -I include/MCMCpp -I include examples/user_calculator.cpp -L <dir of libuser_calculator.so> -luser_calculator
//       -L mcmcpp_amd -lmcmcpp_hip -Wl,-rpath,... -o user_calculator
//   ./user_calculator W D steps init.bin chain_out.bin
//
// init.bin: W*D doubles (initial positions).  The program samples the target twice with ParallelEnsembleSampler --
// once with means 0 and precisions 1, where it must equal the library's built-in isotropic Gaussian bit for bit (checked
// here against Device::IsoGaussian through the same sampler, and by tests/test_plugin.py against the oracle through
// chain_out.bin), once with means and precisions of its own, where the sample means are checked -- and prints a verdict.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "Device/Calculators.h"
#include "Movers/StretchMove.h"
#include "ParallelEnsembleSampler.h"
#include "mcmcpp_hip.h"

extern "C" const void* mcmcpp_hip_plugin_shifted_diagonal_gaussian_f64(void);  // (from MCMCPP_HIP_PLUGIN_CALCULATOR)
extern "C" const void* mcmcpp_hip_plugin_shifted_diagonal_gaussian_f32(void);

// logp(x) = -1/2 sum_j w_j (x_j - mu_j)^2 -- the host twin of ShiftedDiagonalGaussianFn: same operations, same order
class ShiftedDiagonalGaussian
{
public:
    static const int hipCalcId = MCMCPP_HIP_CALC_USER_BASE;  // 1000: the id the functor is registered under
    ShiftedDiagonalGaussian(const std::vector<double>& means, const std::vector<double>& precisions)
        : dims(static_cast<int>(means.size())), params(means), terms(means.size())
    {
        params.insert(params.end(), precisions.begin(), precisions.end());
    }
    MCMCPP_STRICT_FP_FN double calcLogPostProb(double* x)
    {
        MCMCPP_STRICT_FP_BEGIN
        for (int j = 0; j < dims; ++j)
        {
            const double d = x[j] - params[j];
            const double dd = d * d;
            terms[j] = params[dims + j] * dd;
        }
        const double s = MCMC::Device::Detail::treeSum(terms.data(), 0, MCMC::Device::Detail::pow2AtLeast(dims), dims);
        return -0.5 * s;
    }
    const double* hipParams() const { return params.data(); }
    int hipParamCount() const { return 2 * dims; }

private:
    int dims;
    std::vector<double> params, terms;
};

template <class Calc>
static std::vector<double> sample(Calc calc, int W, int D, int steps, const std::vector<double>& init, unsigned long long* accepted, int keepEvery = 1)
{
    typedef MCMC::Mover::StretchMove<double, Calc> Mover;
    Mover mover(D, 0, calc);
    MCMC::ParallelEnsembleSampler<double, Mover> sampler(0, 4, W, D, mover);
    sampler.setSamplingMode(keepEvery, 0);
    std::vector<double> pos(init), aux(W);
    for (int w = 0; w < W; ++w) aux[w] = calc.calcLogPostProb(&pos[(size_t)w * D]);  // user code, on the host, as with the reference
    sampler.setInitialWalkerPos(pos.data(), aux.data());
    sampler.runMCMC(steps);
    *accepted = sampler.getAcceptedSteps();
    std::vector<double> chain;
    for (auto it = sampler.getStepIttBegin(); it != sampler.getStepIttEnd(); ++it) chain.insert(chain.end(), *it, *it + (size_t)W * D);
    return chain;
}

int main(int argc, char** argv)
{
    if (argc < 6)
    {
        std::fprintf(stderr, "usage: user_calculator W D steps init.bin chain_out.bin\n");
        return 2;
    }
    const int W = std::atoi(argv[1]), D = std::atoi(argv[2]), steps = std::atoi(argv[3]);
    std::vector<double> init((size_t)W * D);
    FILE* fp = std::fopen(argv[4], "rb");
    if (!fp || std::fread(init.data(), sizeof(double), init.size(), fp) != init.size())
    {
        std::fprintf(stderr, "cannot read %s\n", argv[4]);
        return 2;
    }
    std::fclose(fp);

    if (mcmcpp_hip_register_calculator(ShiftedDiagonalGaussian::hipCalcId, mcmcpp_hip_plugin_shifted_diagonal_gaussian_f64(),
                                       mcmcpp_hip_plugin_shifted_diagonal_gaussian_f32(), -1) != MCMCPP_HIP_OK)
    {
        std::fprintf(stderr, "registration failed: %s\n", mcmcpp_hip_last_error(nullptr));
        return 1;
    }

    int failures = 0;
    // 1. means 0, precisions 1: the user's target IS the isotropic Gaussian, through the user's own host class and functor
    unsigned long long accUser = 0, accBuiltin = 0;
    const std::vector<double> user = sample(ShiftedDiagonalGaussian(std::vector<double>(D, 0.0), std::vector<double>(D, 1.0)), W, D, steps, init, &accUser);
    const std::vector<double> builtin = sample(MCMC::Device::IsoGaussian<double>(D), W, D, steps, init, &accBuiltin);
    const bool same = user.size() == builtin.size() && std::memcmp(user.data(), builtin.data(), sizeof(double) * user.size()) == 0 && accUser == accBuiltin;
    std::printf("user functor against the built-in it clones: %s (%llu accepted)\n", same ? "identical chains" : "CHAINS DIFFER", accUser);
    failures += !same;
    fp = std::fopen(argv[5], "wb");
    if (!fp || std::fwrite(user.data(), sizeof(double), user.size(), fp) != user.size()) return 2;
    std::fclose(fp);

    // 2. means and precisions of its own: the chain must find them
    std::vector<double> mu(D), w(D);
    for (int j = 0; j < D; ++j)
    {
        mu[j] = 0.1 * j - 1.0;
        w[j] = 0.5 + 0.25 * (j % 5);
    }
    unsigned long long acc = 0;
    const std::vector<double> chain = sample(ShiftedDiagonalGaussian(mu, w), W, D, 2 * steps, init, &acc, 10);  // (every 10th of 20 * steps)
    const size_t stored = chain.size() / ((size_t)W * D), burn = stored / 2;
    for (int j = 0; j < D; ++j)
    {
        double m = 0, v = 0;
        size_t cnt = 0;
        for (size_t s = burn; s < stored; ++s)
            for (int k = 0; k < W; ++k, ++cnt) m += chain[(s * W + k) * D + j];
        m /= (double)cnt;
        for (size_t s = burn; s < stored; ++s)
            for (int k = 0; k < W; ++k)
            {
                const double d = chain[(s * W + k) * D + j] - m;
                v += d * d;
            }
        v /= (double)cnt;
        const bool ok = std::fabs(m - mu[j]) < 0.15 && std::fabs(v * w[j] - 1.0) < 0.25;
        if (!ok) std::printf("parameter %d: mean %g (want %g), variance %g (want %g)\n", j, m, mu[j], v, 1.0 / w[j]);
        failures += !ok;
    }
    std::printf("%s\n", failures == 0 ? "user_calculator OK" : "user_calculator FAILED");
    return failures == 0 ? 0 : 1;
}

/*
 * Device/Calculators.h -- log-posterior Calculators that exist on both sides of the HIP boundary.
 *
 * The reference lets any class with `ParamType calcLogPostProb(ParamType*)` act as the Calculator of
 * a Mover (/root/reference/MCMCpp/Movers/StretchMove.h:47-53, Utility/UserOjbectsTest.h:144-145).
 * A header-only host cannot inject arbitrary host code into GPU kernels, so the MI355X sampler
 * evaluates Calculators that also carry a device identity: a `hipCalcId` naming the device functor
 * compiled into libmcmcpp_hip.so (mcmcpp_amd/csrc/calculators.hpp) and a parameter blob handed over
 * the C ABI (include/mcmcpp_hip.h).  The host `calcLogPostProb` below is what user code calls to
 * fill the initial auxValues array, exactly as with the reference
 * (/root/reference/test/sequential/SkewedGaussian/StretchMove/src/main.cpp:141-205), and is a legal
 * reference Calculator: oracle/ref_driver.cpp runs these very classes through the reference sampler.
 *
 * Operation order is part of each Calculator's definition, because host, oracle and device must
 * agree bit for bit: element terms are combined by a canonical pairwise tree sum (pad to a power of
 * two with +0, add halves recursively), products and sums are individually rounded, and fused
 * multiply-adds appear only where written as std::fma.
 */
#ifndef MCMCPP_DEVICE_CALCULATORS_H
#define MCMCPP_DEVICE_CALCULATORS_H

#include <cmath>
#include <cstddef>
#include <vector>

// The arithmetic below must not be re-associated or contracted by the user's compiler flags.
#if defined(__clang__)
#define MCMCPP_STRICT_FP_BEGIN _Pragma("clang fp contract(off)")
#define MCMCPP_STRICT_FP_FN
#elif defined(__GNUC__)
#define MCMCPP_STRICT_FP_BEGIN
#define MCMCPP_STRICT_FP_FN __attribute__((optimize("fp-contract=off", "no-fast-math")))
#else
#define MCMCPP_STRICT_FP_BEGIN
#define MCMCPP_STRICT_FP_FN
#endif

namespace MCMC
{
namespace Device
{

/// Identifiers of the device functors built into libmcmcpp_hip.so (values = MCMCPP_HIP_CALC_*).
enum CalcId : int { IsoGaussianId = 0, DenseGaussianId = 1, RosenbrockId = 2, SkewedGaussian2DId = 3 };

namespace Detail
{
/// Canonical pairwise sum of t[lo .. lo+len) where cells at index >= count read as +0.
template <class T>
MCMCPP_STRICT_FP_FN inline T treeSum(const T* t, int lo, int len, int count)
{
    MCMCPP_STRICT_FP_BEGIN
    if (len == 1) return (lo < count) ? t[lo] : static_cast<T>(0);
    const int half = len >> 1;
    const T left = treeSum(t, lo, half, count);
    const T right = treeSum(t, lo + half, half, count);
    return left + right;
}

inline int pow2AtLeast(int n)
{
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}
}  // namespace Detail

/// logp(x) = -1/2 sum_j x_j^2
template <class ParamType>
class IsoGaussian
{
public:
    static const int hipCalcId = IsoGaussianId;
    explicit IsoGaussian(int numParams) : dims(numParams), terms(numParams) {}

    MCMCPP_STRICT_FP_FN ParamType calcLogPostProb(ParamType* paramSet)
    {
        MCMCPP_STRICT_FP_BEGIN
        for (int j = 0; j < dims; ++j) terms[j] = paramSet[j] * paramSet[j];
        const ParamType s = Detail::treeSum(terms.data(), 0, Detail::pow2AtLeast(dims), dims);
        return static_cast<ParamType>(-0.5) * s;
    }

    const ParamType* hipParams() const { return nullptr; }
    int hipParamCount() const { return 0; }

private:
    int dims;
    std::vector<ParamType> terms;
};

/// logp(x) = -1/2 x^T P x with a dense row-major precision matrix P (D x D).
template <class ParamType>
class DenseGaussian
{
public:
    static const int hipCalcId = DenseGaussianId;
    DenseGaussian(int numParams, const ParamType* precisionRowMajor)
        : dims(numParams), prec(precisionRowMajor, precisionRowMajor + static_cast<size_t>(numParams) * numParams),
          terms(numParams)
    {
    }

    MCMCPP_STRICT_FP_FN ParamType calcLogPostProb(ParamType* paramSet)
    {
        MCMCPP_STRICT_FP_BEGIN
        for (int i = 0; i < dims; ++i)
        {
            ParamType acc = static_cast<ParamType>(0);
            const ParamType* row = prec.data() + static_cast<size_t>(i) * dims;
            for (int j = 0; j < dims; ++j) acc = std::fma(row[j], paramSet[j], acc);
            terms[i] = paramSet[i] * acc;
        }
        const ParamType s = Detail::treeSum(terms.data(), 0, Detail::pow2AtLeast(dims), dims);
        return static_cast<ParamType>(-0.5) * s;
    }

    const ParamType* hipParams() const { return prec.data(); }
    int hipParamCount() const { return dims * dims; }

    /// Precision matrix of the AR(1)-correlated Gaussian Sigma_ij = rho^|i-j| (closed-form tridiagonal
    /// inverse, stored dense): the BASELINE "correlated Gaussian" workload.
    static std::vector<ParamType> ar1Precision(int numParams, ParamType rho)
    {
        std::vector<ParamType> p(static_cast<size_t>(numParams) * numParams, static_cast<ParamType>(0));
        const ParamType d = static_cast<ParamType>(1) - rho * rho;
        for (int i = 0; i < numParams; ++i)
        {
            const bool edge = (i == 0) || (i == numParams - 1);
            p[static_cast<size_t>(i) * numParams + i] =
                (edge ? static_cast<ParamType>(1) : (static_cast<ParamType>(1) + rho * rho)) / d;
            if (i + 1 < numParams)
            {
                p[static_cast<size_t>(i) * numParams + i + 1] = -rho / d;
                p[static_cast<size_t>(i + 1) * numParams + i] = -rho / d;
            }
        }
        return p;
    }

private:
    int dims;
    std::vector<ParamType> prec;
    std::vector<ParamType> terms;
};

/// logp(x) = -c * sum_{i<D-1} [ b (x_{i+1} - x_i^2)^2 + (a - x_i)^2 ]
template <class ParamType>
class Rosenbrock
{
public:
    static const int hipCalcId = RosenbrockId;
    Rosenbrock(int numParams, ParamType a = static_cast<ParamType>(1), ParamType b = static_cast<ParamType>(100),
               ParamType c = static_cast<ParamType>(0.05))
        : dims(numParams), terms(numParams)
    {
        abc[0] = a;
        abc[1] = b;
        abc[2] = c;
    }

    MCMCPP_STRICT_FP_FN ParamType calcLogPostProb(ParamType* paramSet)
    {
        MCMCPP_STRICT_FP_BEGIN
        for (int i = 0; i < dims; ++i)
        {
            if (i + 1 < dims)
            {
                const ParamType sq = paramSet[i] * paramSet[i];
                const ParamType u = paramSet[i + 1] - sq;
                const ParamType v = abc[0] - paramSet[i];
                const ParamType uu = u * u;
                const ParamType buu = abc[1] * uu;
                const ParamType vv = v * v;
                terms[i] = buu + vv;
            }
            else
            {
                terms[i] = static_cast<ParamType>(0);
            }
        }
        const ParamType s = Detail::treeSum(terms.data(), 0, Detail::pow2AtLeast(dims), dims);
        const ParamType scaled = s * abc[2];
        return -scaled;
    }

    const ParamType* hipParams() const { return abc; }
    int hipParamCount() const { return 3; }

private:
    int dims;
    ParamType abc[3];
    std::vector<ParamType> terms;
};

/// The reference's own two-dimensional test target
/// (/root/reference/test/sequential/SkewedGaussian/Common/SkewedGaussian.h:52-57), same operation order:
/// logp = -((x/2 - y)^2 / eps + (x/2 + y)^2) / 2.
template <class ParamType>
class SkewedGaussian2D
{
public:
    static const int hipCalcId = SkewedGaussian2DId;
    explicit SkewedGaussian2D(ParamType eps) : epsilon(eps) {}

    MCMCPP_STRICT_FP_FN ParamType calcLogPostProb(ParamType* paramSet)
    {
        MCMCPP_STRICT_FP_BEGIN
        const ParamType half = paramSet[0] / static_cast<ParamType>(2);
        const ParamType lo = half - paramSet[1];
        const ParamType hi = half + paramSet[1];
        const ParamType a = (lo * lo) / epsilon;
        const ParamType b = hi * hi;
        return (a + b) / static_cast<ParamType>(-2);
    }

    const ParamType* hipParams() const { return &epsilon; }
    int hipParamCount() const { return 1; }

private:
    ParamType epsilon;
};

}  // namespace Device
}  // namespace MCMC
#endif  // MCMCPP_DEVICE_CALCULATORS_H

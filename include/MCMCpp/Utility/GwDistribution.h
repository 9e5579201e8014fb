/*
 * Utility/GwDistribution.h -- inverse CDF of the Goodman & Weare stretch-factor density g(z) ~ 1/sqrt(z) on
 * [1/a, a]: z = ((sqrt(a) - 1/sqrt(a)) u + 1/sqrt(a))^2, a = AlphaNum/AlphaDenom
 * (/root/reference/MCMCpp/Utility/GwDistribution.h:45-58).  The device kernels evaluate the same three
 * operations (multiply, add, square, individually rounded) with the constants below.
 */
#ifndef MCMCPP_UTILITY_GWDISTRIBUTION_H
#define MCMCPP_UTILITY_GWDISTRIBUTION_H

#include <cmath>

namespace MCMC
{
namespace Utility
{
template <class ParamType, int AlphaNum, int AlphaDenom>
class GwDistribution
{
public:
    static const int Numerator = AlphaNum;
    static const int Denominator = AlphaDenom;
    static ParamType alpha() { return static_cast<ParamType>(AlphaNum) / static_cast<ParamType>(AlphaDenom); }
    static ParamType invSqrtAlpha() { return static_cast<ParamType>(1) / std::sqrt(alpha()); }
    static ParamType slope() { return std::sqrt(alpha()) - invSqrtAlpha(); }

    ParamType operator()(ParamType u) const
    {
        const ParamType scaled = slope() * u;
        const ParamType root = scaled + invSqrtAlpha();
        return root * root;
    }
};
}  // namespace Utility
}  // namespace MCMC
#endif  // MCMCPP_UTILITY_GWDISTRIBUTION_H

// fast_log.hpp -- natural logarithm with a short dependency chain, for the accept test of the stretch move.
//
// The two logarithms of StretchMove::updateWalker (ln z at MCMCpp/Movers/StretchMove.h:110 and the
// exponential variate -log(1-u) at :113 via libstdc++) only feed the comparison `lnU < (D-1) ln z + dlogp`;
// they are never stored.  The half-step kernel is latency-bound (one dependent chain per wavefront), and the
// library logarithm is the longest link of that chain, so the kernel uses this restatement of the classic
// argument-reduction + atanh-series algorithm (x = 2^k (1+f), s = f/(2+f), log(1+f) = f - s (f - R(s^2))),
// accurate to about 1 ulp.  A decision that a last-ulp difference could flip is counted as a near tie by
// the kernel and by the oracle alike (tests assert there are none), so parity stays exact.
//
// Plain C++ (no HIP dependency) so that a CPU test can compile it and check its accuracy against libm.
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define MCMCPP_HD __host__ __device__ __forceinline__
#else
#define MCMCPP_HD inline
#endif

namespace mcmcpp
{

// x must be a positive normal number (the kernel passes z in [1/2, 2] and 1-u in [2^-53, 1])
MCMCPP_HD double fast_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    // minimax coefficients of (log(1+f) - 2s)/s over s^2 <= 0.0295 (R(z) = Lg1 z + ... + Lg7 z^7)
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int k = (int)(bits >> 52) - 1023;
    bits = (bits & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;  // m in [1, 2)
    double m;
    memcpy(&m, &bits, 8);
    if (m > 1.41421356237309504880)
    {
        m *= 0.5;
        k += 1;
    }
    const double f = m - 1.0;  // in [sqrt(1/2)-1, sqrt(2)-1]
    const double dk = (double)k;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    // log(x) = k ln2 + f - (hfsq - s (hfsq + R)), assembled so that the small terms are summed first
    return __builtin_fma(dk, ln2_hi, f - (hfsq - __builtin_fma(s, hfsq + R, dk * ln2_lo)));
}

}  // namespace mcmcpp

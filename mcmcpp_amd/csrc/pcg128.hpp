// pcg128.hpp -- 128-bit LCG arithmetic of pcg64 (setseq_xsl_rr_128_64) for host and gfx950 device.
//
// The reference draws every random number from one `pcg64` engine (MCMCpp/Utility/MultiSampler.h:120,
// imneme/pcg-cpp).  The device does not walk that stream sequentially: lane i of half-step h jumps to
// draw 3*(h*n + i) through precomputed affine maps (state -> mult*state + plus), which reproduces the
// reference's sequential stream bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mcmcpp
{

struct U128
{
    uint64_t lo, hi;
};

__host__ __device__ __forceinline__ U128 make_u128(uint64_t hi, uint64_t lo)
{
    U128 r;
    r.lo = lo;
    r.hi = hi;
    return r;
}

__host__ __device__ __forceinline__ uint64_t mulhi64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// low 128 bits of a*b
__host__ __device__ __forceinline__ U128 mul128(U128 a, U128 b)
{
    U128 r;
    r.lo = a.lo * b.lo;
    r.hi = mulhi64(a.lo, b.lo) + a.lo * b.hi + a.hi * b.lo;
    return r;
}

__host__ __device__ __forceinline__ U128 add128(U128 a, U128 b)
{
    U128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
    return r;
}

// state' = mult*state + plus (mod 2^128)
struct Affine128
{
    U128 mult, plus;
};

__host__ __device__ __forceinline__ U128 apply(const Affine128& f, U128 s) { return add128(mul128(f.mult, s), f.plus); }

// pcg-cpp default 128-bit multiplier
__host__ __device__ __forceinline__ U128 pcg_multiplier() { return make_u128(2549297995355413924ULL, 4865540595714422341ULL); }

// one LCG step
__host__ __device__ __forceinline__ U128 pcg_step(U128 s, U128 inc) { return add128(mul128(s, pcg_multiplier()), inc); }

// XSL-RR 128 -> 64 of the post-step state
__host__ __device__ __forceinline__ uint64_t pcg_output(U128 s)
{
    const uint64_t x = s.hi ^ s.lo;
    const unsigned rot = (unsigned)(s.hi >> 58);
    return (x >> rot) | (x << ((0u - rot) & 63u));
}

// engine(seed, stream): inc = (stream << 1) | 1, state = (seed + inc)*M + inc   (MultiSampler.h:54)
inline void pcg_seed(uint64_t seed, uint64_t stream, U128* state, U128* inc)
{
    *inc = make_u128(stream >> 63, (stream << 1) | 1u);
    *state = pcg_step(add128(make_u128(0, seed), *inc), *inc);
}

// coefficients of `delta` LCG steps (Brown's arbitrary-stride algorithm), host only
inline Affine128 pcg_jump(U128 inc, unsigned __int128 delta)
{
    U128 cur_mult = pcg_multiplier(), cur_plus = inc;
    Affine128 acc;
    acc.mult = make_u128(0, 1);
    acc.plus = make_u128(0, 0);
    while (delta > 0)
    {
        if (delta & 1u)
        {
            acc.mult = mul128(acc.mult, cur_mult);
            acc.plus = add128(mul128(acc.plus, cur_mult), cur_plus);
        }
        cur_plus = mul128(add128(cur_mult, make_u128(0, 1)), cur_plus);
        cur_mult = mul128(cur_mult, cur_mult);
        delta >>= 1;
    }
    return acc;
}

// libstdc++ generate_canonical<T>(pcg64): T(r) rounded to nearest, divided by 2^64, clamped below 1
// (bits/random.tcc:3345-3380; MultiSampler.h:60,86 through uniform_real / exponential distributions)
__device__ __forceinline__ double canonical(uint64_t r, double)
{
    // u64 -> f64 round-to-nearest-even: hi*2^32 is exact, lo is exact, one rounded add
    const double hi = (double)(uint32_t)(r >> 32);
    const double lo = (double)(uint32_t)r;
    double u = __builtin_fma(hi, 4294967296.0, lo) * 5.42101086242752217003726400434970855712890625e-20;
    // fma(hi, 2^32, lo) rounds once (the product is exact), as the conversion instruction would
    if (u >= 1.0) u = 0.99999999999999988897769753748434595763683319091796875;
    return u;
}

__device__ __forceinline__ float canonical(uint64_t r, float)
{
    float u = (float)r * 5.42101086242752217003726400434970855712890625e-20f;
    if (u >= 1.0f) u = 0.999999940395355224609375f;
    return u;
}

}  // namespace mcmcpp

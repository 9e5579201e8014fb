// calculators.hpp -- device functors of the log-posterior Calculators (gfx950).
//
// Reference concept: `ParamType Calculator::calcLogPostProb(ParamType*)`
// (MCMCpp/Utility/UserOjbectsTest.h:144-145, called at MCMCpp/Movers/StretchMove.h:111).  On the
// device one walker's D-vector is spread over a group of LPW lanes of a wavefront, EPL consecutive
// elements per lane (element index = sub*EPL + e); a functor receives its lane's slice of the proposal
// and a GroupCtx for the few cross-lane operations it may need, and returns the log-posterior in every
// lane of the group.  Operation order mirrors include/MCMCpp/Device/Calculators.h exactly (canonical
// pairwise tree sum, individually rounded products, fma only where written).
#pragma once

#include <hip/hip_runtime.h>

namespace mcmcpp
{

// Cross-lane facilities of one walker's lane group.
template <class T, int EPL, int LPW>
struct GroupCtx
{
    int sub;        // lane index inside the group, 0..LPW-1
    int dims;       // D
    T* stage;       // wave-private LDS: 64*EPL elements, this lane's slice at [lane*EPL, lane*EPL+EPL)
    int lane;       // lane in wavefront

    __device__ __forceinline__ int first_index() const { return sub * EPL; }

    // canonical pairwise sum over the group's LPW*EPL cells (cells >= D must hold +0)
    __device__ __forceinline__ T tree_sum(const T (&t)[EPL]) const
    {
        T v[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) v[e] = t[e];
#pragma unroll
        for (int w = EPL; w > 1; w >>= 1)
        {
#pragma unroll
            for (int e = 0; e < w / 2; ++e) v[e] = v[2 * e] + v[2 * e + 1];
        }
        T s = v[0];
#pragma unroll
        for (int off = 1; off < LPW; off <<= 1) s = s + __shfl_xor(s, off, 64);
        return s;
    }

    // publish this lane's slice so that any lane of the group can read any element of the walker
    __device__ __forceinline__ void publish(const T (&x)[EPL]) const
    {
#pragma unroll
        for (int e = 0; e < EPL; ++e) stage[lane * EPL + e] = x[e];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // element j of this walker (after publish)
    __device__ __forceinline__ T element(int j) const { return stage[(lane - sub) * EPL + j]; }

    // first element held by the next lane of the group (x_{i+1} for the last element of this lane)
    __device__ __forceinline__ T next_lane_first(T x0) const { return __shfl_down(x0, 1, 64); }
};

// -1/2 sum x^2
template <class T>
struct IsoGaussianFn
{
    static constexpr bool kNeedsStage = false;
    template <int EPL, int LPW>
    __device__ __forceinline__ static T eval(const GroupCtx<T, EPL, LPW>& g, const T* /*params*/, const T (&x)[EPL])
    {
        T t[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) t[e] = x[e] * x[e];  // padded cells hold x = 0 -> +0
        return (T)-0.5 * g.tree_sum(t);
    }
};

// -1/2 x^T P x, params = P transposed (PT[j*D + i] = P[i][j]) so that a group's lanes read contiguously
template <class T>
struct DenseGaussianFn
{
    static constexpr bool kNeedsStage = true;
    template <int EPL, int LPW>
    __device__ __forceinline__ static T eval(const GroupCtx<T, EPL, LPW>& g, const T* pt, const T (&x)[EPL])
    {
        g.publish(x);
        const int D = g.dims;
        const int i0 = g.first_index();
        T acc[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = (T)0;
        if (i0 + EPL <= D)
        {
            for (int j = 0; j < D; ++j)
            {
                const T xj = g.element(j);
                const T* col = pt + (size_t)j * D + i0;
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc[e] = __builtin_fma(col[e], xj, acc[e]);
            }
        }
        else
        {
            for (int j = 0; j < D; ++j)
            {
                const T xj = g.element(j);
                const T* col = pt + (size_t)j * D + i0;
#pragma unroll
                for (int e = 0; e < EPL; ++e)
                    if (i0 + e < D) acc[e] = __builtin_fma(col[e], xj, acc[e]);
            }
        }
        T t[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) t[e] = (i0 + e < D) ? x[e] * acc[e] : (T)0;
        return (T)-0.5 * g.tree_sum(t);
    }
};

// -c sum_{i<D-1} b (x_{i+1} - x_i^2)^2 + (a - x_i)^2, params = {a, b, c}
template <class T>
struct RosenbrockFn
{
    static constexpr bool kNeedsStage = false;
    template <int EPL, int LPW>
    __device__ __forceinline__ static T eval(const GroupCtx<T, EPL, LPW>& g, const T* prm, const T (&x)[EPL])
    {
        const T a = prm[0], b = prm[1], c = prm[2];
        const T xn = g.next_lane_first(x[0]);
        const int i0 = g.first_index();
        T t[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e)
        {
            const T nxt = (e + 1 < EPL) ? x[(e + 1 < EPL) ? e + 1 : 0] : xn;
            const T sq = x[e] * x[e];
            const T u = nxt - sq;
            const T v = a - x[e];
            const T uu = u * u;
            const T buu = b * uu;
            const T vv = v * v;
            const T term = buu + vv;
            t[e] = (i0 + e + 1 < g.dims) ? term : (T)0;
        }
        const T s = g.tree_sum(t);
        return -(s * c);
    }
};

// the reference's SkewedGaussianTwoDim (test/sequential/SkewedGaussian/Common/SkewedGaussian.h:52-57);
// D == 2 lives in one lane (EPL == 2 for double; for float EPL == 4 with two padded cells)
template <class T>
struct SkewedGaussian2DFn
{
    static constexpr bool kNeedsStage = false;
    template <int EPL, int LPW>
    __device__ __forceinline__ static T eval(const GroupCtx<T, EPL, LPW>& /*g*/, const T* prm, const T (&x)[EPL])
    {
        static_assert(EPL >= 2, "SkewedGaussian2D needs both coordinates in one lane");
        const T eps = prm[0];
        const T half = x[0] / (T)2;
        const T lo = half - x[1];
        const T hi = half + x[1];
        const T p = (lo * lo) / eps;
        const T q = hi * hi;
        return (p + q) / (T)-2;
    }
};

}  // namespace mcmcpp

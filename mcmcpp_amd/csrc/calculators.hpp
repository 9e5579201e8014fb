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

// 16-byte vector of the element type (the widest per-lane access)
template <class T>
struct Vec16;
template <>
struct Vec16<double>
{
    typedef double2 type;
    static constexpr int N = 2;
};
template <>
struct Vec16<float>
{
    typedef float4 type;
    static constexpr int N = 4;
};

// block-shared LDS a calculator may claim for read-only tables (bytes are capped so that several
// workgroups still fit one CU's 160 KiB)
constexpr size_t kMaxBlockScratchBytes = 48 * 1024;

// whole-register DPP move of a float/double (all rows and banks enabled, bound_ctrl irrelevant: every lane is read)
template <int CTRL>
__device__ __forceinline__ float dpp_move(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)b, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// Cross-lane facilities of one walker's lane group.
template <class T, int EPL, int LPW>
struct GroupCtx
{
    int sub;        // lane index inside the group, 0..LPW-1
    int dims;       // D
    T* stage;       // wave-private LDS: 64*EPL elements, this lane's slice at [lane*EPL, lane*EPL+EPL)
    int lane;       // lane in wavefront
    const T* block_scratch;  // block-shared LDS filled by Calc::block_init (nullptr when the calculator claims none)
    bool vec_ok;    // D is a whole number of 16-byte vectors

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
        // Butterfly over the group's lanes.  At the step of width `off` every lane of one half-block of
        // 2*off lanes already holds the same partial sum, so any lane of the other half-block is a valid
        // partner: DPP quad permutes (off 1, 2) and row mirrors (off 4, 8) replace LDS-routed shuffles.
        if (LPW > 1) s = s + dpp_move<0xB1>(s);   // quad_perm [1,0,3,2]
        if (LPW > 2) s = s + dpp_move<0x4E>(s);   // quad_perm [2,3,0,1]
        if (LPW > 4) s = s + dpp_move<0x141>(s);  // row_half_mirror
        if (LPW > 8) s = s + dpp_move<0x140>(s);  // row_mirror
#pragma unroll
        for (int off = 16; off < LPW; off <<= 1) s = s + __shfl_xor(s, off, 64);
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
    __host__ __device__ static size_t block_scratch_elems(int) { return 0; }
    __device__ static void block_init(T*, const T*, int, bool, int, int) {}
    template <int EPL, int LPW>
    struct Regs
    {
    };
    template <int EPL, int LPW>
    __device__ __forceinline__ static void preload(const GroupCtx<T, EPL, LPW>&, const T*, Regs<EPL, LPW>&)
    {
    }
    template <int EPL, int LPW>
    __device__ __forceinline__ static T eval(const GroupCtx<T, EPL, LPW>& g, const T* /*params*/, const Regs<EPL, LPW>&,
                                             const T (&x)[EPL])
    {
        T t[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) t[e] = x[e] * x[e];  // padded cells hold x = 0 -> +0
        return (T)-0.5 * g.tree_sum(t);
    }
};

// -1/2 x^T P x, params = P transposed (PT[j*D + i] = P[i][j]) so that a group's lanes read contiguously.
// The matrix is read D times per walker, so each workgroup keeps a copy in LDS (when it fits); the
// proposal is published to wave-private LDS so that every lane can read every x_j (broadcast reads).
// y_i = sum_j P_ij x_j is accumulated as one fma chain in ascending j, as the host Calculator does.
template <class T>
struct DenseGaussianFn
{
    static constexpr bool kNeedsStage = true;
    __host__ __device__ static size_t block_scratch_elems(int D)
    {
        const size_t e = (size_t)D * (size_t)D;
        return e * sizeof(T) <= kMaxBlockScratchBytes ? e : 0;
    }
    __device__ static void block_init(T* scratch, const T* pt, int D, bool vec_ok, int tid, int nthreads)
    {
        if (block_scratch_elems(D) == 0) return;
        constexpr int VN = Vec16<T>::N;
        typedef typename Vec16<T>::type V;
        const int total = D * D;
        if (vec_ok)
            for (int k = tid * VN; k < total; k += nthreads * VN)
                *reinterpret_cast<V*>(scratch + k) = *reinterpret_cast<const V*>(pt + k);
        else
            for (int k = tid; k < total; k += nthreads) scratch[k] = pt[k];
    }

    // Walkers of up to 32 padded dimensions keep their lane's slice of the matrix in registers (loaded once
    // per wavefront, before the random draws are known); larger ones read it from the LDS / global copy.
    template <int EPL, int LPW>
    struct Regs
    {
        static constexpr bool kUse = (EPL * LPW <= 32);
        T col[kUse ? EPL * LPW : 1][EPL];
    };
    template <int EPL, int LPW>
    __device__ __forceinline__ static void preload(const GroupCtx<T, EPL, LPW>& g, const T* pt_global, Regs<EPL, LPW>& r)
    {
        if constexpr (Regs<EPL, LPW>::kUse)
        {
            // called after block_init + barrier: the workgroup's LDS copy of the matrix is the source
            if (g.block_scratch != nullptr)
                preload_from<EPL, LPW>(g, g.block_scratch, r);
            else
                preload_from<EPL, LPW>(g, pt_global, r);
        }
    }
    template <int EPL, int LPW, class PtrT>
    __device__ __forceinline__ static void preload_from(const GroupCtx<T, EPL, LPW>& g, PtrT pt, Regs<EPL, LPW>& r)
    {
        {
            constexpr int N2 = EPL * LPW;
            constexpr int VN = Vec16<T>::N;
            typedef typename Vec16<T>::type V;
            const int D = g.dims;
            const int i0 = g.first_index();
            // branch-free: every load is issued unconditionally from a clamped (always valid) address and the
            // cells outside the D x D matrix are zeroed afterwards, so all N2 loads are in flight together
            if (g.vec_ok)
            {
#pragma unroll
                for (int j = 0; j < N2; ++j)
                {
                    const int jc = j < D ? j : D - 1;
#pragma unroll
                    for (int v = 0; v < EPL / VN; ++v)
                    {
                        const int ic = (i0 + v * VN < D) ? i0 + v * VN : D - VN;
                        const V c = *reinterpret_cast<const V*>(pt + (size_t)jc * D + ic);
                        const T* cs = reinterpret_cast<const T*>(&c);
                        const bool inside = (j < D) && (i0 + v * VN < D);
#pragma unroll
                        for (int k = 0; k < VN; ++k) r.col[j][v * VN + k] = inside ? cs[k] : (T)0;
                    }
                }
            }
            else
            {
#pragma unroll
                for (int j = 0; j < N2; ++j)
                {
                    const int jc = j < D ? j : D - 1;
#pragma unroll
                    for (int e = 0; e < EPL; ++e)
                    {
                        const int ic = (i0 + e < D) ? i0 + e : D - 1;
                        const T c = pt[(size_t)jc * D + ic];
                        r.col[j][e] = ((j < D) && (i0 + e < D)) ? c : (T)0;
                    }
                }
            }
        }
    }

    template <int EPL, int LPW, class PtrT>
    __device__ __forceinline__ static void mat_vec(const GroupCtx<T, EPL, LPW>& g, PtrT pt, T (&acc)[EPL])
    {
        constexpr int VN = Vec16<T>::N;
        typedef typename Vec16<T>::type V;
        const int D = g.dims;
        const int i0 = g.first_index();
        const T* xs = g.stage + (g.lane - g.sub) * EPL;
        if (g.vec_ok)
        {
            for (int j = 0; j < D; j += VN)
            {
                const V xv = *reinterpret_cast<const V*>(xs + j);
                const T* xj = reinterpret_cast<const T*>(&xv);
#pragma unroll
                for (int jj = 0; jj < VN; ++jj)
                {
#pragma unroll
                    for (int v = 0; v < EPL / VN; ++v)
                    {
                        if (i0 + v * VN < D)
                        {
                            const V c = *reinterpret_cast<const V*>(pt + (size_t)(j + jj) * D + i0 + v * VN);
                            const T* cs = reinterpret_cast<const T*>(&c);
#pragma unroll
                            for (int k = 0; k < VN; ++k)
                                acc[v * VN + k] = __builtin_fma(cs[k], xj[jj], acc[v * VN + k]);
                        }
                    }
                }
            }
        }
        else
        {
            for (int j = 0; j < D; ++j)
            {
                const T xj = xs[j];
#pragma unroll
                for (int e = 0; e < EPL; ++e)
                    if (i0 + e < D) acc[e] = __builtin_fma(pt[(size_t)j * D + i0 + e], xj, acc[e]);
            }
        }
    }

    template <int EPL, int LPW>
    __device__ __forceinline__ static T eval(const GroupCtx<T, EPL, LPW>& g, const T* pt_global, const Regs<EPL, LPW>& r,
                                             const T (&x)[EPL])
    {
        g.publish(x);
        T acc[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = (T)0;
        if constexpr (Regs<EPL, LPW>::kUse)
        {
            constexpr int N2 = EPL * LPW;
            constexpr int VN = Vec16<T>::N;
            typedef typename Vec16<T>::type V;
            // the whole proposal of this walker, read back from the stage with broadcast 16-byte reads
            T xs[N2];
            const T* src = g.stage + (g.lane - g.sub) * EPL;
#pragma unroll
            for (int j = 0; j < N2; j += VN)
            {
                const V xv = *reinterpret_cast<const V*>(src + j);
                const T* xp = reinterpret_cast<const T*>(&xv);
#pragma unroll
                for (int k = 0; k < VN; ++k) xs[j + k] = xp[k];
            }
            // No guards: cells outside the matrix hold col = 0 and x = +0, and an fma chain that starts at +0
            // can never sit at -0 under round-to-nearest, so fma(0, 0, acc) == acc bit for bit.
#pragma unroll
            for (int j = 0; j < N2; ++j)
            {
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc[e] = __builtin_fma(r.col[j][e], xs[j], acc[e]);
            }
        }
        else if (g.block_scratch != nullptr)
            mat_vec<EPL, LPW>(g, g.block_scratch, acc);
        else
            mat_vec<EPL, LPW>(g, pt_global, acc);
        const int i0 = g.first_index();
        T t[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) t[e] = (i0 + e < g.dims) ? x[e] * acc[e] : (T)0;
        return (T)-0.5 * g.tree_sum(t);
    }
};

// -c sum_{i<D-1} b (x_{i+1} - x_i^2)^2 + (a - x_i)^2, params = {a, b, c}
template <class T>
struct RosenbrockFn
{
    static constexpr bool kNeedsStage = false;
    __host__ __device__ static size_t block_scratch_elems(int) { return 0; }
    __device__ static void block_init(T*, const T*, int, bool, int, int) {}
    template <int EPL, int LPW>
    struct Regs
    {
    };
    template <int EPL, int LPW>
    __device__ __forceinline__ static void preload(const GroupCtx<T, EPL, LPW>&, const T*, Regs<EPL, LPW>&)
    {
    }
    template <int EPL, int LPW>
    __device__ __forceinline__ static T eval(const GroupCtx<T, EPL, LPW>& g, const T* prm, const Regs<EPL, LPW>&,
                                             const T (&x)[EPL])
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
    __host__ __device__ static size_t block_scratch_elems(int) { return 0; }
    __device__ static void block_init(T*, const T*, int, bool, int, int) {}
    template <int EPL, int LPW>
    struct Regs
    {
    };
    template <int EPL, int LPW>
    __device__ __forceinline__ static void preload(const GroupCtx<T, EPL, LPW>&, const T*, Regs<EPL, LPW>&)
    {
    }
    template <int EPL, int LPW>
    __device__ __forceinline__ static T eval(const GroupCtx<T, EPL, LPW>& /*g*/, const T* prm, const Regs<EPL, LPW>&,
                                             const T (&x)[EPL])
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

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
    const T* block_scratch;  // block-shared LDS filled by Calc::block_commit (nullptr when the calculator claims none)
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
    template <int EPL, int LPW>
    struct MatrixCore
    {
        static constexpr bool kUse = false;
    };
    __host__ __device__ static size_t block_scratch_elems(int) { return 0; }
    struct Prefetch
    {
    };
    __device__ __forceinline__ static void block_prefetch(Prefetch&, const T*, int, bool, int, int) {}
    __device__ __forceinline__ static void block_commit(const Prefetch&, T*, const T*, int, bool, int, int) {}
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
// The matrix is read once per walker, so each workgroup keeps a copy in LDS: zero-padded to N2 x N2
// (N2 = the walker's padded dimension, a power of two) when N2 <= 64, which lets the product loop run
// without any bounds test; the proposal is published to wave-private LDS so that every lane can read every
// x_j with broadcast reads.  y_i = sum_j P_ij x_j is accumulated as one fma chain in ascending j, as the host
// Calculator does (padding contributes fma(0, 0, acc), which leaves acc untouched bit for bit: a chain that
// starts at +0 never sits at -0 under round-to-nearest).
template <class T>
struct DenseGaussianFn
{
    static constexpr bool kNeedsStage = true;
    static constexpr int kMaxPadded = 64;
    // walkers of 17..32 dimensions take the matrix-core kernels (stretch_half_step_mfma_kernel): the slot of that range
    // is 16 lanes x 2 elements for fp64, 8 lanes x 4 elements for fp32
    template <int EPL, int LPW>
    struct MatrixCore
    {
        static constexpr bool kUse = (sizeof(T) == 8 && EPL == 2 && LPW == 16) || (sizeof(T) == 4 && EPL == 4 && LPW == 8);
    };

    __host__ __device__ static int padded_dim(int D)
    {
        int n2 = Vec16<T>::N;
        while (n2 < D) n2 <<= 1;
        return n2;
    }
    __host__ __device__ static size_t block_scratch_elems(int D)
    {
        const int n2 = padded_dim(D);
        return n2 <= kMaxPadded ? (size_t)n2 * (size_t)n2 : 0;
    }

    // The workgroup's LDS copy is filled in two steps so that the global loads are issued with the launch's
    // first batch of loads and land in registers while other work proceeds: every thread prefetches up to
    // kPrefetchVecs 16-byte pieces, commit stores them (and any remainder) to LDS.
    static constexpr int kPrefetchVecs = 2;
    struct Prefetch
    {
        typename Vec16<T>::type v[kPrefetchVecs];
    };
    // piece k of the padded matrix (16 bytes = VN cells of row j starting at column i)
    __device__ __forceinline__ static typename Vec16<T>::type fetch_piece(const T* pt, int D, int n2, bool vec_ok, int k)
    {
        constexpr int VN = Vec16<T>::N;
        typedef typename Vec16<T>::type V;
        const int cell = k * VN;
        const int j = cell / n2, i = cell - j * n2;
        V out;
        T* o = reinterpret_cast<T*>(&out);
        if (vec_ok)
        {
            // D is a multiple of VN: a piece lies wholly inside or wholly outside the matrix
            const bool inside = j < D && i < D;
            const V c = *reinterpret_cast<const V*>(pt + (size_t)(inside ? j : 0) * D + (inside ? i : 0));
            const T* cs = reinterpret_cast<const T*>(&c);
#pragma unroll
            for (int e = 0; e < VN; ++e) o[e] = inside ? cs[e] : (T)0;
        }
        else
        {
#pragma unroll
            for (int e = 0; e < VN; ++e) o[e] = (j < D && i + e < D) ? pt[(size_t)j * D + i + e] : (T)0;
        }
        return out;
    }
    __device__ __forceinline__ static void block_prefetch(Prefetch& pf, const T* pt, int D, bool vec_ok, int tid, int nthreads)
    {
        const int n2 = padded_dim(D);
        if (n2 > kMaxPadded) return;
        const int pieces = n2 * n2 / Vec16<T>::N;
#pragma unroll
        for (int r = 0; r < kPrefetchVecs; ++r)
        {
            const int k = tid + r * nthreads;
            if (k < pieces) pf.v[r] = fetch_piece(pt, D, n2, vec_ok, k);
        }
    }
    __device__ __forceinline__ static void block_commit(const Prefetch& pf, T* scratch, const T* pt, int D, bool vec_ok, int tid,
                                                        int nthreads)
    {
        typedef typename Vec16<T>::type V;
        const int n2 = padded_dim(D);
        if (n2 > kMaxPadded) return;
        const int pieces = n2 * n2 / Vec16<T>::N;
        V* dst = reinterpret_cast<V*>(scratch);
#pragma unroll
        for (int r = 0; r < kPrefetchVecs; ++r)
        {
            const int k = tid + r * nthreads;
            if (k < pieces) dst[k] = pf.v[r];
        }
        for (int k = tid + kPrefetchVecs * nthreads; k < pieces; k += nthreads) dst[k] = fetch_piece(pt, D, n2, vec_ok, k);
    }

    template <int EPL, int LPW>
    struct Regs
    {
    };
    template <int EPL, int LPW>
    __device__ __forceinline__ static void preload(const GroupCtx<T, EPL, LPW>&, const T*, Regs<EPL, LPW>&)
    {
    }

    // matrix in global memory (D x D, row stride D): the fallback for padded dimensions above kMaxPadded
    template <int EPL, int LPW>
    __device__ __forceinline__ static void mat_vec_global(const GroupCtx<T, EPL, LPW>& g, const T* pt, T (&acc)[EPL])
    {
        const int D = g.dims;
        const int i0 = g.first_index();
        const T* xs = g.stage + (g.lane - g.sub) * EPL;
        for (int j = 0; j < D; ++j)
        {
            const T xj = xs[j];
#pragma unroll
            for (int e = 0; e < EPL; ++e)
                if (i0 + e < D) acc[e] = __builtin_fma(pt[(size_t)j * D + i0 + e], xj, acc[e]);
        }
    }

    template <int EPL, int LPW>
    __device__ __forceinline__ static T eval(const GroupCtx<T, EPL, LPW>& g, const T* pt_global, const Regs<EPL, LPW>&,
                                             const T (&x)[EPL])
    {
        constexpr int N2 = EPL * LPW;
        constexpr int VN = Vec16<T>::N;
        typedef typename Vec16<T>::type V;
        g.publish(x);
        T acc[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = (T)0;
        if constexpr (N2 <= kMaxPadded)
        {
            // no bounds tests; unrolled four x-vectors at a time so that a bounded number of LDS reads is in
            // flight (a full unroll makes the compiler hold the whole matrix slice in registers)
            const T* xs = g.stage + (g.lane - g.sub) * EPL;
            const T* pt = g.block_scratch + g.first_index();
#pragma unroll 4
            for (int j = 0; j < N2; j += VN)
            {
                const V xv = *reinterpret_cast<const V*>(xs + j);
                const T* xj = reinterpret_cast<const T*>(&xv);
#pragma unroll
                for (int jj = 0; jj < VN; ++jj)
                {
#pragma unroll
                    for (int v = 0; v < EPL / VN; ++v)
                    {
                        const V c = *reinterpret_cast<const V*>(pt + (j + jj) * N2 + v * VN);
                        const T* cs = reinterpret_cast<const T*>(&c);
#pragma unroll
                        for (int k = 0; k < VN; ++k) acc[v * VN + k] = __builtin_fma(cs[k], xj[jj], acc[v * VN + k]);
                    }
                }
            }
        }
        else
            mat_vec_global<EPL, LPW>(g, pt_global, acc);
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
    template <int EPL, int LPW>
    struct MatrixCore
    {
        static constexpr bool kUse = false;
    };
    __host__ __device__ static size_t block_scratch_elems(int) { return 0; }
    struct Prefetch
    {
    };
    __device__ __forceinline__ static void block_prefetch(Prefetch&, const T*, int, bool, int, int) {}
    __device__ __forceinline__ static void block_commit(const Prefetch&, T*, const T*, int, bool, int, int) {}
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
    template <int EPL, int LPW>
    struct MatrixCore
    {
        static constexpr bool kUse = false;
    };
    __host__ __device__ static size_t block_scratch_elems(int) { return 0; }
    struct Prefetch
    {
    };
    __device__ __forceinline__ static void block_prefetch(Prefetch&, const T*, int, bool, int, int) {}
    __device__ __forceinline__ static void block_commit(const Prefetch&, T*, const T*, int, bool, int, int) {}
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

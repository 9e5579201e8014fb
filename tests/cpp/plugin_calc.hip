// Test plug-in: two user-compiled device Calculators built against mcmcpp_amd/csrc/mcmcpp_hip_plugin.hpp.
//   iso_clone    the isotropic Gaussian again, under a user id: trajectories must equal the built-in's bit for bit
//   diag_shifted -1/2 sum_j w_j (x_j - mu_j)^2 with params = {mu[D], w[D]}: a functor the library does not ship
#include "mcmcpp_hip_plugin.hpp"

template <class T>
struct NoTables
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
    __device__ static void block_prefetch(Prefetch&, const T*, int, bool, int, int) {}
    __device__ static void block_commit(const Prefetch&, T*, const T*, int, bool, int, int) {}
    template <int EPL, int LPW>
    struct Regs
    {
    };
    template <int EPL, int LPW>
    __device__ static void preload(const mcmcpp::GroupCtx<T, EPL, LPW>&, const T*, Regs<EPL, LPW>&)
    {
    }
};

template <class T>
struct IsoClone : NoTables<T>
{
    template <int EPL, int LPW>
    __device__ static T eval(const mcmcpp::GroupCtx<T, EPL, LPW>& g, const T*, const typename NoTables<T>::template Regs<EPL, LPW>&,
                             const T (&x)[EPL])
    {
        T t[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) t[e] = x[e] * x[e];
        return (T)-0.5 * g.tree_sum(t);
    }
};

template <class T>
struct DiagShifted : NoTables<T>
{
    template <int EPL, int LPW>
    __device__ static T eval(const mcmcpp::GroupCtx<T, EPL, LPW>& g, const T* prm, const typename NoTables<T>::template Regs<EPL, LPW>&,
                             const T (&x)[EPL])
    {
        const int D = g.dims, i0 = g.first_index();
        T t[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e)
        {
            const int j = i0 + e;
            if (j < D)
            {
                const T d = x[e] - prm[j];
                const T dd = d * d;
                t[e] = prm[D + j] * dd;
            }
            else
                t[e] = (T)0;
        }
        return (T)-0.5 * g.tree_sum(t);
    }
};

MCMCPP_HIP_PLUGIN_CALCULATOR(IsoClone, iso_clone)
MCMCPP_HIP_PLUGIN_CALCULATOR(DiagShifted, diag_shifted)

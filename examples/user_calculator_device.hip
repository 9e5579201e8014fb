// Example, device side: a user's own log-posterior as a device functor (see examples/user_calculator.cpp for the host side).
//
// The reference takes the Calculator as a template argument of the Mover and calls calcLogPostProb on the host
// (/root/reference/MCMCpp/Movers/StretchMove.h:42-54,111).  Here the same target is written twice: a host class (for the
// initial auxValues, and so that the program still reads like a program for the reference) and this functor, which the
// step kernels evaluate for every proposal.  Both must perform the same operations in the same order.
//
//   hipcc -std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fPIC -shared \
//         -mllvm -amdgpu-kernarg-preload-count=16 -I mcmcpp_amd/csrc examples/user_calculator_device.hip -o libuser_calculator.so
//
// Target: independent Gaussians with their own means and precisions, logp(x) = -1/2 sum_j w_j (x_j - mu_j)^2,
// parameters {mu[D], w[D]}.  With mu = 0 and w = 1 every operation below returns what the library's own isotropic
// Gaussian computes (x - 0 = x, 1 * x^2 = x^2 exactly), so that case must follow the built-in's trajectory bit for bit.
#include "mcmcpp_hip_plugin.hpp"

template <class T>
struct ShiftedDiagonalGaussianFn
{
    // a functor that needs neither LDS staging, per-workgroup tables, registers of its own nor the matrix cores
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

    // this lane holds elements [g.first_index(), g.first_index() + EPL) of the proposal; cells at index >= D are +0
    template <int EPL, int LPW>
    __device__ static T eval(const mcmcpp::GroupCtx<T, EPL, LPW>& g, const T* prm, const Regs<EPL, LPW>&, const T (&x)[EPL])
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
        return (T)-0.5 * g.tree_sum(t);  // the canonical pairwise sum the host twin performs (Device::Detail::treeSum)
    }
};

MCMCPP_HIP_PLUGIN_CALCULATOR(ShiftedDiagonalGaussianFn, shifted_diagonal_gaussian)

// mcmcpp_hip_plugin.hpp -- user-compiled device Calculators.
//
// The reference accepts any class with `ParamType calcLogPostProb(ParamType*)` as a Calculator
// (MCMCpp/Utility/UserOjbectsTest.h:144-145).  On the MI355X path the Calculator runs inside the half-step
// kernel, so user code reaches it as a *device functor* compiled by hipcc against this header into a small
// shared library, which the application registers with libmcmcpp_hip.so under a calculator id >= 1000:
//
//     // my_calc.hip
//     #include "mcmcpp_hip_plugin.hpp"
//     template <class T> struct MyTarget {                       // see calculators.hpp for the built-in ones
//         static constexpr bool kNeedsStage = false;             // true: ctx.publish()/ctx.element(j) are used
//         template <int EPL, int LPW> struct MatrixCore { static constexpr bool kUse = false; };
//         __host__ __device__ static size_t block_scratch_elems(int) { return 0; }
//         struct Prefetch {};
//         __device__ static void block_prefetch(Prefetch&, const T*, int, bool, int, int) {}
//         __device__ static void block_commit(const Prefetch&, T*, const T*, int, bool, int, int) {}
//         template <int EPL, int LPW> struct Regs {};
//         template <int EPL, int LPW> __device__ static void preload(const mcmcpp::GroupCtx<T, EPL, LPW>&, const T*, Regs<EPL, LPW>&) {}
//         // this lane holds elements [g.first_index(), g.first_index()+EPL) of the proposal (cells >= D are +0);
//         // return the log-posterior in every lane of the walker's group
//         template <int EPL, int LPW>
//         __device__ static T eval(const mcmcpp::GroupCtx<T, EPL, LPW>& g, const T* params, const Regs<EPL, LPW>&, const T (&x)[EPL]);
//     };
//     MCMCPP_HIP_PLUGIN_CALCULATOR(MyTarget, my_target)
//
//     hipcc -std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -mllvm -amdgpu-kernarg-preload-count=16 \
//           -I <repo>/mcmcpp_amd/csrc my_calc.hip -o libmy_target.so
//
// and in the application (links libmy_target.so and libmcmcpp_hip.so):
//
//     extern "C" const void* mcmcpp_hip_plugin_my_target_f64(void);
//     extern "C" const void* mcmcpp_hip_plugin_my_target_f32(void);
//     mcmcpp_hip_register_calculator(1000, mcmcpp_hip_plugin_my_target_f64(), mcmcpp_hip_plugin_my_target_f32(), /*params*/ -1);
//
// after which `calc_id = 1000` (or a host Calculator class whose hipCalcId is 1000) selects it.
#pragma once

#include "launch_build.hpp"

#define MCMCPP_HIP_PLUGIN_CALCULATOR(FUNCTOR_TEMPLATE, NAME)                                                     \
    extern "C" const void* mcmcpp_hip_plugin_##NAME##_f64(void)                                                   \
    {                                                                                                            \
        return mcmcpp::make_launch_table<double, FUNCTOR_TEMPLATE>();                                            \
    }                                                                                                            \
    extern "C" const void* mcmcpp_hip_plugin_##NAME##_f32(void) { return mcmcpp::make_launch_table<float, FUNCTOR_TEMPLATE>(); }

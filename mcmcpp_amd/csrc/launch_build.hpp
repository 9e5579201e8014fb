// launch_build.hpp -- instantiates the kernels of one (element type, calculator functor) pair and builds the
// host-side launch table over them.  Used by the library's own instantiation units (instances.inc) and by
// user plug-ins (mcmcpp_hip_plugin.hpp).
#pragma once

#include "launch_table.hpp"

namespace mcmcpp
{
namespace build
{
template <class T, class Calc, int EPL, int LPW>
void launch_half(const HalfStepArgs<T>& a, unsigned grid, hipStream_t st)
{
    const size_t lds = LdsLayout<T, Calc, EPL>::bytes(a.dims);
    const int chains = a.chains > 1 ? a.chains : 1;
    const uint32_t bits = HotBits::pack(a.dims, a.passes, a.color, a.vec_ok, a.n_is_pow2, a.use_ctl_save, a.draw_parity, a.draw_wave, a.task_jump != nullptr) |
                          ((uint32_t)(chains - 1) << 28);
#define MCMCPP_LAUNCH_HALF(DW, MC, THREADS)                                                                                                              \
    hipLaunchKernelGGL((stretch_half_step_kernel<T, Calc, EPL, LPW, DW, MC>), dim3(grid, chains), dim3(THREADS), lds, st, a.draws, a.pos, a.logp, a.n_accept, a.n, \
                       bits, a.shard_begin, a.shard_count, a.ctl_in, a)
    // (four instantiations: with / without the draw wavefront, for one ensemble / several chains per launch)
    if (a.draw_wave && chains > 1)
        MCMCPP_LAUNCH_HALF(true, true, 64 * (kWavesPerBlock + 1));
    else if (a.draw_wave)
        MCMCPP_LAUNCH_HALF(true, false, 64 * (kWavesPerBlock + 1));
    else if (chains > 1)
        MCMCPP_LAUNCH_HALF(false, true, 64 * kWavesPerBlock);
    else
        MCMCPP_LAUNCH_HALF(false, false, 64 * kWavesPerBlock);
#undef MCMCPP_LAUNCH_HALF
}

template <class T, class Calc, int EPL, int LPW, int P, bool LATE = false>
void launch_half_mfma(const HalfStepArgs<T>& a, unsigned grid, hipStream_t st)
{
    // (staging rows of the workgroup's wavefronts, then the 32 x 32 matrix shared by workgroups without a draw wavefront)
    const size_t lds = ((size_t)kWavesPerBlock * (sizeof(T) == 8 ? 4 * P : 16) * kMcXS + 1024) * sizeof(T);
    const int chains = a.chains > 1 ? a.chains : 1;
    const uint32_t bits = HotBits::pack(a.dims, a.passes, a.color, a.vec_ok, a.n_is_pow2, a.use_ctl_save, a.draw_parity, a.draw_wave, a.task_jump != nullptr) |
                          ((uint32_t)(chains - 1) << 28);
#define MCMCPP_LAUNCH_HALF_MC(DW, MC, THREADS)                                                                                                            \
    hipLaunchKernelGGL((stretch_half_step_mfma_kernel<T, Calc, EPL, LPW, P, DW, MC, (LATE && !(DW))>), dim3(grid, chains), dim3(THREADS), lds, st, a.draws, a.pos, a.logp, a.n_accept, \
                       a.n, bits, a.shard_begin, a.shard_count, a.ctl_in, a.calc_params_padded, a)
    // (wavefronts of 16 walkers make their next draws themselves: the host never asks for a draw wavefront there)
    if constexpr (P < 4)
    {
        if (a.draw_wave && chains > 1)
        {
            MCMCPP_LAUNCH_HALF_MC(true, true, 64 * (kWavesPerBlock + 1));
            return;
        }
        if (a.draw_wave)
        {
            MCMCPP_LAUNCH_HALF_MC(true, false, 64 * (kWavesPerBlock + 1));
            return;
        }
    }
    if (chains > 1)
        MCMCPP_LAUNCH_HALF_MC(false, true, 64 * kWavesPerBlock);
    else
        MCMCPP_LAUNCH_HALF_MC(false, false, 64 * kWavesPerBlock);
#undef MCMCPP_LAUNCH_HALF_MC
}

template <class T, class Calc, int EPL, int LPW>
void launch_full(const HalfStepArgs<T>& a, unsigned grid, hipStream_t st)
{
    const size_t lds = LdsLayout<T, Calc, EPL>::bytes(a.dims);
    const int chains = a.chains > 1 ? a.chains : 1;
    const uint32_t bits = full_step_bits(HotBits::pack(a.dims, 1, 0, a.vec_ok, a.n_is_pow2, a.use_ctl_save, a.draw_parity, a.draw_wave, a.task_jump != nullptr), a.pos_parity) |
                          ((uint32_t)(chains - 1) << 28);
    if (chains > 1)
        hipLaunchKernelGGL((stretch_full_step_kernel<T, Calc, EPL, LPW, true>), dim3(grid, chains), dim3(64 * (kWavesPerBlock + (a.draw_wave ? kFullDrawWaves : 0))),
                           lds, st, a.draws, a.pos, a.pos_alt, a.logp, a.run, a.shard_begin, a.shard_count, a.n, bits, a.ctl_in, a);
    else
        hipLaunchKernelGGL((stretch_full_step_kernel<T, Calc, EPL, LPW, false>), dim3(grid), dim3(64 * (kWavesPerBlock + (a.draw_wave ? kFullDrawWaves : 0))), lds,
                           st, a.draws, a.pos, a.pos_alt, a.logp, a.run, a.shard_begin, a.shard_count, a.n, bits, a.ctl_in, a);
}

template <class T, class Calc, int EPL, int LPW>
void launch_full_mfma(const HalfStepArgs<T>& a, unsigned grid, hipStream_t st)
{
    const size_t lds = ((size_t)kWavesPerBlock * (sizeof(T) == 8 ? 3 : 4) * 8 * kMcXS) * sizeof(T);
    const int chains = a.chains > 1 ? a.chains : 1;
    // (the colour bit, meaningless for a full step, says that the draw records were made ahead: HalfStepArgs::draw_wave == 2;
    //  such a launch has one extra wavefront per workgroup, the forwarder of stored steps, instead of the draw wavefronts)
    const int prefilled = a.draw_wave == 2 ? 1 : 0;
    const int extra_waves = prefilled ? 1 : kFullDrawWaves;
    const uint32_t bits = full_step_bits(HotBits::pack(a.dims, 2, prefilled, a.vec_ok, a.n_is_pow2, a.use_ctl_save, a.draw_parity, 1, a.task_jump != nullptr), a.pos_parity) |
                          ((uint32_t)(chains - 1) << 28);
    // (logp_alt == logp + W, n_accept == logp + 2 W and the run record kRunBehindCtlBytes behind the control records: the
    //  kernel derives them and takes the padded matrix and the shard bounds as preloaded arguments instead)
    if (chains > 1 || a.chains < 0)  // (a.chains < 0: experiments -- the several-chains instantiation for one ensemble)
        hipLaunchKernelGGL((stretch_full_step_mfma_kernel<T, Calc, EPL, LPW, true>), dim3(grid, chains), dim3(64 * (kWavesPerBlock + extra_waves)), lds, st, a.draws,
                           a.pos, a.pos_alt, a.logp, a.calc_params_padded, a.shard_begin, a.shard_count, a.n, bits, a.ctl_in, a);
    else
        hipLaunchKernelGGL((stretch_full_step_mfma_kernel<T, Calc, EPL, LPW, false>), dim3(grid), dim3(64 * (kWavesPerBlock + extra_waves)), lds, st, a.draws,
                           a.pos, a.pos_alt, a.logp, a.calc_params_padded, a.shard_begin, a.shard_count, a.n, bits, a.ctl_in, a);
}

template <class T, class Calc, int EPL, int LPW>
void launch_calc(const T* pos, T* out, const T* prm, long long count, int dims, int vec_ok, unsigned grid, hipStream_t st)
{
    const size_t lds = LdsLayout<T, Calc, EPL>::bytes(dims);
    hipLaunchKernelGGL((calc_logp_kernel<T, Calc, EPL, LPW>), dim3(grid), dim3(64 * kWavesPerBlock), lds, st, pos, out, prm,
                       count, dims, vec_ok);
}

template <class T, class Calc, int EPL, int LPW>
void launch_de(const typename LaunchTable<T>::DeLaunch& l, const DeArgs<T>& a, unsigned grid, hipStream_t st)
{
    const size_t lds = LdsLayout<T, Calc, EPL>::bytes(l.dims);
    hipLaunchKernelGGL((de_update_kernel<T, Calc, EPL, LPW>), dim3(grid), dim3(64 * kWavesPerBlock), lds, st, l.pos, l.logp, l.n_accept, l.recs, l.jump_small, l.run, l.n,
                       de_hot_bits(l.dims, l.color, l.vec_ok), l.step, a);
}

template <class T, class Calc, int EPL, int LPW, int P>
void launch_de_mfma(const typename LaunchTable<T>::DeLaunch& l, const DeArgs<T>& a, unsigned grid, hipStream_t st)
{
    const size_t lds = ((size_t)kWavesPerBlock * (sizeof(T) == 8 ? 4 * P : 16) * kMcXS) * sizeof(T);
    // (the step inside the replay travels in the hot bits: the sixteenth preloaded dword pair is the matrix pointer)
    hipLaunchKernelGGL((de_update_mfma_kernel<T, Calc, EPL, LPW, P>), dim3(grid), dim3(64 * kWavesPerBlock), lds, st, l.pos, l.logp, l.n_accept, l.recs, l.jump_small, l.run,
                       l.n, de_hot_bits(l.dims, l.color, l.vec_ok, l.step), l.matrix_padded, a);
}

template <class T, class Calc, int LPWLOG, int EPLSHIFT>
void put(LaunchTable<T>& t)
{
    constexpr int kBase = Vec16<T>::N;
    t.half_step[LPWLOG][EPLSHIFT] = &launch_half<T, Calc, (kBase << EPLSHIFT), (1 << LPWLOG)>;
    t.full_step[LPWLOG][EPLSHIFT] = &launch_full<T, Calc, (kBase << EPLSHIFT), (1 << LPWLOG)>;
    if constexpr (Calc::template MatrixCore<(kBase << EPLSHIFT), (1 << LPWLOG)>::kUse)
    {
        // (the matrix-core kernels always map a walker to 16 lanes x 2 elements, whatever the slot's own mapping -- for
        //  fp32 walkers of 17..32 dimensions the slot is 8 lanes x 4 elements)
        t.half_step_mc[0][LPWLOG][EPLSHIFT] = &launch_half_mfma<T, Calc, 2, 16, 2>;
        t.half_step_mc[1][LPWLOG][EPLSHIFT] = &launch_half_mfma<T, Calc, 2, 16, 4>;
        t.half_step_mc[2][LPWLOG][EPLSHIFT] = &launch_half_mfma<T, Calc, 2, 16, 4, true>;
        t.full_step_mc[LPWLOG][EPLSHIFT] = &launch_full_mfma<T, Calc, 2, 16>;
        t.de_update_mc[0][LPWLOG][EPLSHIFT] = &launch_de_mfma<T, Calc, 2, 16, 2>;
        t.de_update_mc[1][LPWLOG][EPLSHIFT] = &launch_de_mfma<T, Calc, 2, 16, 4>;
    }
    t.calc[LPWLOG][EPLSHIFT] = &launch_calc<T, Calc, (kBase << EPLSHIFT), (1 << LPWLOG)>;
    t.de_update[LPWLOG][EPLSHIFT] = &launch_de<T, Calc, (kBase << EPLSHIFT), (1 << LPWLOG)>;
}

// OnlyLpw1: only the single-lane mapping is meaningful (a fixed low-dimensional target such as D = 2)
template <class T, class Calc, bool OnlyLpw1>
LaunchTable<T> make()
{
    LaunchTable<T> t = {};
    t.abi = kLaunchTableAbi;
    t.elem_size = (uint32_t)sizeof(T);
    put<T, Calc, 0, 0>(t);
    if constexpr (!OnlyLpw1)
    {
        put<T, Calc, 1, 0>(t);
        put<T, Calc, 2, 0>(t);
        put<T, Calc, 3, 0>(t);
        put<T, Calc, 4, 0>(t);
        put<T, Calc, 5, 0>(t);
        put<T, Calc, 6, 0>(t);
        put<T, Calc, 6, 1>(t);
        put<T, Calc, 6, 2>(t);
        if constexpr (sizeof(T) == 8) put<T, Calc, 6, 3>(t);  // D up to 1024 in both element types
    }
    return t;
}
}  // namespace build

template <class T, template <class> class CalcT, bool OnlyLpw1 = false>
const LaunchTable<T>* make_launch_table()
{
    static const LaunchTable<T> table = build::make<T, CalcT<T>, OnlyLpw1>();
    return &table;
}
}  // namespace mcmcpp

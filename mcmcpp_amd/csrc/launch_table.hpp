// launch_table.hpp -- host-side dispatch tables over the kernel instantiations.
#pragma once

#include <hip/hip_runtime.h>

#include "diffevo_kernel.hpp"
#include "full_step_kernel.hpp"
#include "stretch_kernel.hpp"

namespace mcmcpp
{

// index of EPL among {base, 2*base, 4*base, 8*base}; base = 16 bytes / sizeof(T)
constexpr int kMaxEplShift = 4;
constexpr int kLpwLevels = 7;  // LPW = 1,2,4,...,64

constexpr uint32_t kLaunchTableAbi = 0x4D430018u;  // bumped whenever HalfStepArgs or the launcher signatures change

template <class T>
struct LaunchTable
{
    uint32_t abi;        // kLaunchTableAbi of the headers the table was built from
    uint32_t elem_size;  // sizeof(T)
    typedef void (*HalfStepFn)(const HalfStepArgs<T>&, unsigned grid, hipStream_t);
    typedef void (*CalcFn)(const T* pos, T* out, const T* params, long long count, int dims, int vec_ok, unsigned grid,
                           hipStream_t);
    // [log2(LPW)][log2(EPL/base)]; nullptr where not built
    HalfStepFn half_step[kLpwLevels][kMaxEplShift];
    CalcFn calc[kLpwLevels][kMaxEplShift];
    // matrix-core variants (nullptr where the calculator has none): need even D and exactly 2 ([0]) / 4 ([1]) passes
    HalfStepFn half_step_mc[3][kLpwLevels][kMaxEplShift];  // [8 walkers per wavefront | 16 | 16 with the next draws behind the accept]
    // one launch per ensemble step (full_step_kernel.hpp); grid counts workgroups of 4 x (64/LPW) walkers per colour
    // (generic) or 32 per colour (matrix-core variant)
    HalfStepFn full_step[kLpwLevels][kMaxEplShift];
    HalfStepFn full_step_mc[kLpwLevels][kMaxEplShift];
    // Mover::DifferentialEvolution (diffevo_kernel.hpp): one launch updates a whole half from the records of that half-step
    struct DeLaunch
    {
        T* pos;
        T* logp;
        uint32_t* n_accept;
        const DeRec<T>* recs;  // the n records of this half-step
        const Affine128* jump_small;
        DeRunInfo* run;
        int n, dims, color, vec_ok;
        int step;  // ensemble step inside the replay (< 65536)
        const T* matrix_padded;  // matrix-core variants: P^T zero-padded to 32 x 32
    };
    typedef void (*DeFn)(const DeLaunch&, const DeArgs<T>&, unsigned grid, hipStream_t);
    DeFn de_update[kLpwLevels][kMaxEplShift];
    // matrix-core variants (nullptr where the calculator has none): 8 ([0]) / 16 ([1]) walkers per wavefront, even D only
    DeFn de_update_mc[2][kLpwLevels][kMaxEplShift];
};

// red_base != nullptr: black records, with partner2 (see DrawRec)
void launch_fill_draws(const HalfStepArgs<double>& a, U128 base, const U128* red_base, hipStream_t stream);
void launch_fill_draws(const HalfStepArgs<float>& a, U128 base, const U128* red_base, hipStream_t stream);
// the records of `steps` ensemble steps from the one whose control record is `ctl` on (fill_draws_batch_kernel)
void launch_fill_draws_batch(const HalfStepArgs<double>& a, const StepCtl* ctl, const Affine128* step_jump, DrawRec<double>* out, int steps, hipStream_t stream);
void launch_fill_draws_batch(const HalfStepArgs<float>& a, const StepCtl* ctl, const Affine128* step_jump, DrawRec<float>* out, int steps, hipStream_t stream);
void launch_accepted_reduce(const uint32_t* partials, int partial_slots, int partial_waves, int count,
                            const StepCtl* ctl_after, const RunInfo* run, hipStream_t stream, int chains = 1, StepCtl* ctl_keep = nullptr);

// one definition per (element type, calculator), each in its own translation unit
const LaunchTable<double>* launch_table_f64_iso();
const LaunchTable<double>* launch_table_f64_dense();
const LaunchTable<double>* launch_table_f64_rosenbrock();
const LaunchTable<double>* launch_table_f64_skewed();
const LaunchTable<float>* launch_table_f32_iso();
const LaunchTable<float>* launch_table_f32_dense();
const LaunchTable<float>* launch_table_f32_rosenbrock();
const LaunchTable<float>* launch_table_f32_skewed();

}  // namespace mcmcpp

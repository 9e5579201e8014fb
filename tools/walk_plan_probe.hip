// walk_plan_probe.hip -- diagnostic: the floor of a DEVICE plan pass for Mover::WalkMove
// (/root/reference/MCMCpp/Movers/WalkMove.h:101-186), measured to decide whether that mover gets a kernel.
//
// WalkMove consumes its single pcg64 stream in data-dependent amounts: selectWalkers (:130-151) is selection sampling
// with one uniform per EXAMINED walker (about n k / (k + 1) of them), then k normals from libstdc++'s polar method (a
// rejection loop whose spare variate is carried into the next update), then one exponential.  Where walker i + 1 starts
// in the stream is known only when walker i has finished: the walk over the walkers is serial whatever executes it.
// What a GPU can do is make every step of that walk wide: one wavefront draws 64 stream positions at once (jump-ahead:
// one 128-bit multiply-add per lane) and finds the k selections with k ballots.  This probe does exactly that -- the
// selection and the count of polar attempts of every update of one half-step, n = 160 and k = 6 as the reference's
// own test (test/sequential/SkewedGaussian/WalkMove/src/main.cpp:25,35) and n = 2048 -- and checks the stream
// positions it arrives at against a sequential host walk of the same stream.  Output: us per update.
//   hipcc -O3 --offload-arch=gfx950 -I mcmcpp_amd/csrc tools/walk_plan_probe.hip -o tools/walk_plan_probe.bin
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pcg128.hpp"

using namespace mcmcpp;

#define CHECK(x)                                                         \
    do                                                                   \
    {                                                                    \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess)                                            \
        {                                                                \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                                \
        }                                                                \
    } while (0)

static double host_canonical(uint64_t r)
{
    double u = (double)r * 5.42101086242752217003726400434970855712890625e-20;
    if (u >= 1.0) u = 0.99999999999999988897769753748434595763683319091796875;
    return u;
}

// one wavefront; jump[t] = map of t + 1 draws (t < 64); walks `updates` WalkMove updates, records the stream position
// at which each one starts (relative to the first)
__global__ void __launch_bounds__(64) plan_kernel(const Affine128* jump, U128 state, int n, int k, int updates, unsigned long long* start_of, unsigned long long* ticks)
{
    const int lane = threadIdx.x;
    const Affine128 mine = jump[lane];  // lane t: the state behind draw t + 1 of a round
    unsigned long long pos = 0;
    bool spare = false;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int w = 0; w < updates; ++w)
    {
        if (lane == 0) start_of[w] = pos;
        // ---- selectWalkers: the j-th selection is the first examined walker after the (j-1)-th whose uniform passes
        //      (n - examined) * u < k - j  (WalkMove.h:139-149) ----
        int selected = 0, examined = 0;
        while (selected < k)
        {
            const U128 s = apply(mine, state);
            const double u = canonical(pcg_output(s), double());
            const double lhs = (double)(n - (examined + lane)) * u;
            unsigned long long pending = ~0ULL;  // lanes not yet consumed in this round
            int used = 64;
            while (selected < k)
            {
                const unsigned long long hit = __ballot(!(lhs >= (double)(k - selected))) & pending;
                if (hit == 0) break;
                const int at = __ffsll((long long)hit) - 1;
                ++selected;
                pending = (at == 63) ? 0ULL : (~0ULL << (at + 1));
                if (selected == k) used = at + 1;
            }
            // the state behind the last draw this update consumed in the round
            const U128 behind = s;
            state.lo = __shfl(behind.lo, used - 1);
            state.hi = __shfl(behind.hi, used - 1);
            examined += used;
            pos += (unsigned long long)used;
        }
        // ---- k normals: polar attempts of two draws each until enough variates (bits/random.tcc, normal_distribution) ----
        int need = k - (spare ? 1 : 0);  // variates still to make
        while (need > 0)
        {
            const U128 s = apply(mine, state);
            const double u = canonical(pcg_output(s), double());
            const double x = 2.0 * u - 1.0;
            const double other = __shfl_xor(x, 1);
            const double r2 = x * x + other * other;
            const bool ok = !(r2 > 1.0 || r2 == 0.0);
            const unsigned long long good = __ballot(ok && (lane & 1) == 0);  // accepted attempts, by their first lane
            // an accepted attempt yields 2 variates: attempts needed = ceil(need / 2); find the lane of that one
            const int want = (need + 1) / 2;
            unsigned long long g = good;
            int at = -1, have = 0;
            while (g != 0 && have < want)
            {
                at = __ffsll((long long)g) - 1;
                g &= g - 1;
                ++have;
            }
            int used;
            if (have == want)
            {
                used = at + 2;
                spare = (need & 1) != 0;  // an odd request leaves the second variate of the last attempt behind
                need = 0;
            }
            else
            {
                used = 64;
                need -= 2 * have;
            }
            const U128 behind = s;
            state.lo = __shfl(behind.lo, used - 1);
            state.hi = __shfl(behind.hi, used - 1);
            pos += (unsigned long long)used;
        }
        // ---- the exponential of the accept test: one draw ----
        {
            const U128 s = apply(mine, state);
            state.lo = __shfl(s.lo, 0);
            state.hi = __shfl(s.hi, 0);
            pos += 1;
        }
    }
    if (lane == 0)
    {
        ticks[0] = __builtin_amdgcn_s_memrealtime() - t0;
        start_of[updates] = pos;
    }
}

int main()
{
    U128 state0, inc;
    pcg_seed(0, 0, &state0, &inc);
    std::vector<Affine128> jump(64);
    for (int t = 0; t < 64; ++t) jump[t] = pcg_jump(inc, (unsigned __int128)(t + 1));
    Affine128* d_jump;
    unsigned long long *d_start, *d_ticks;
    CHECK(hipMalloc(&d_jump, sizeof(Affine128) * 64));
    CHECK(hipMemcpy(d_jump, jump.data(), sizeof(Affine128) * 64, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_ticks, 8));
    const int k = 6;
    for (int n : {160, 2048})
    {
        const int updates = n;  // one half-step
        CHECK(hipMalloc(&d_start, 8 * (size_t)(updates + 1)));
        for (int rep = 0; rep < 2; ++rep)
        {
            hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(64), 0, 0, d_jump, state0, n, k, updates, d_start, d_ticks);
            CHECK(hipDeviceSynchronize());
        }
        std::vector<unsigned long long> start(updates + 1);
        unsigned long long ticks = 0;
        CHECK(hipMemcpy(start.data(), d_start, 8 * (size_t)(updates + 1), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost));
        // the same walk on the host, one draw after the other
        U128 s = state0;
        unsigned long long pos = 0;
        bool spare = false;
        int wrong = 0;
        const auto h0 = std::chrono::steady_clock::now();
        for (int w = 0; w < updates; ++w)
        {
            if (start[w] != pos) ++wrong;
            int selected = 0, examined = 0;
            while (selected < k)
            {
                s = pcg_step(s, inc);
                ++pos;
                const double u = host_canonical(pcg_output(s));
                if (!((double)(n - examined) * u >= (double)(k - selected))) ++selected;
                ++examined;
            }
            for (int j = 0; j < k; ++j)
            {
                if (spare)
                {
                    spare = false;
                    continue;
                }
                double x, y, r2;
                do
                {
                    s = pcg_step(s, inc);
                    x = 2.0 * host_canonical(pcg_output(s)) - 1.0;
                    s = pcg_step(s, inc);
                    y = 2.0 * host_canonical(pcg_output(s)) - 1.0;
                    pos += 2;
                    r2 = x * x + y * y;
                } while (r2 > 1.0 || r2 == 0.0);
                spare = true;
            }
            s = pcg_step(s, inc);
            ++pos;
        }
        const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
        if (start[updates] != pos) ++wrong;
        std::printf("n = %4d, k = %d: %llu stream draws per half-step (%.1f per update); device plan walk %.2f us per update (%.1f us per half-step), "
                    "the same walk on one host core %.2f us per update; stream positions %s\n",
                    n, k, pos, (double)pos / updates, ticks * 0.01 / updates, ticks * 0.01, host_us / updates, wrong ? "DIFFER" : "agree");
        CHECK(hipFree(d_start));
    }
    return 0;
}

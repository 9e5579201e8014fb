// diffevo.hip -- host side and stream-planning kernels of Mover::DifferentialEvolution on gfx950 (SURVEY.md 8f row f3;
// reference MCMCpp/Movers/DifferentialEvolution.h:80-112 inside EnsembleSampler::performStep, EnsembleSampler.h:342-354).
// See diffevo_kernel.hpp for the scheme: plan (parallel) -> update (a few dozen sequential steps from LDS, then parallel),
// two launches per half-step on one stream, the stream position and the error flag travelling in device memory.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "launch_table.hpp"
#include "sampler_base.hpp"

using namespace mcmcpp;

namespace
{
// Would an update of walker k that starts r draws late (r = 0..kDeMaxShift) throw draws away?  One wavefront per walker:
// lane j makes raw draw j behind (D+3)k (one table jump from the walker's base state), neighbouring lanes compare, and
// for the rare walker where the answer is yes for some r, lane r walks the update that starts at draw r and the table
// extra[r] goes to the candidate list.
__global__ void __launch_bounds__(256)
de_plan_kernel(DeCtl* ctl, DeCand* cand, const Affine128* jump_hi, const Affine128* jump_lo, const Affine128* jump_small, uint64_t threshold, int n)
{
    __shared__ uint64_t sh_raw[4][64];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const bool pow2 = (n & (n - 1)) == 0;
    const U128 state = ctl->state;
    const Affine128 j_draw = jump_small[(lane < kDeRaw ? lane : kDeRaw - 1) + 1];
    const int waves = gridDim.x * 4;
    for (int k = blockIdx.x * 4 + wib; k < n; k += waves)
    {
        const U128 base = apply(jump_lo[k & 255], apply(jump_hi[k >> 8], state));
        const uint64_t raw = pcg_output(apply(j_draw, base));
        const uint64_t nxt = __shfl_down(raw, 1);
        // clean(j): draws j and j+1 are both kept and name different walkers: an update starting at j throws nothing away
        const bool bad = lane <= kDeMaxShift && (raw < threshold || nxt < threshold || de_bounded(raw, n, pow2) == de_bounded(nxt, n, pow2));
        if (__ballot(bad) == 0) continue;
        uint32_t slot = 0;
        if (lane == 0) slot = atomicAdd(&ctl->cand_count, 1u);
        slot = __shfl(slot, 0);
        if (slot >= (uint32_t)kDeMaxCand)
        {
            if (lane == 0) atomicOr(&ctl->error, kDeErrCand);
            continue;
        }
        // (one wavefront: its LDS accesses execute in order; the fences keep the compiler from moving them)
        sh_raw[wib][lane] = raw;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane <= kDeMaxShift)
        {
            // DifferentialEvolution.h:83-87 from draw `lane` on
            const uint64_t* rw = sh_raw[wib];
            int at = lane;
            const int end = lane + kDeWindow;
            uint64_t v;
            do v = rw[at++];
            while (v < threshold && at < end);
            const uint32_t ind1 = de_bounded(v, n, pow2);
            uint32_t ind2 = ind1;
            bool overrun = v < threshold;
            do
            {
                if (at >= end)
                {
                    overrun = true;
                    break;
                }
                do v = rw[at++];
                while (v < threshold && at < end);
                if (v < threshold) overrun = true;
                ind2 = de_bounded(v, n, pow2);
            } while (ind2 == ind1);
            // (an overrun is an error only if the walk in the update kernel comes through this start)
            cand[slot].extra[lane] = overrun ? (uint8_t)kDeOverrun : (uint8_t)(at - lane - 2);
        }
        if (lane == 0) cand[slot].k = (uint32_t)k;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}


template <class T>
class DeSampler final : public mcmcpp_hip_sampler
{
public:
    ~DeSampler() override { release(); }

    int init(const mcmcpp_hip_config& c)
    {
        W = c.num_walkers;
        D = c.num_params;
        n = W / 2;
        table = static_cast<const LaunchTable<T>*>(launch_table_lookup(c.dtype, c.calc_id));
        if (!table) return fail(MCMCPP_HIP_E_ARG, "calc_id %d has no kernels for this element type", c.calc_id);
        if (table->abi != kLaunchTableAbi || table->elem_size != sizeof(T))
            return fail(MCMCPP_HIP_E_ARG, "calc_id %d: the plug-in was built against other headers (table abi %08x)", c.calc_id, table->abi);
        const int base = Vec16<T>::N;
        const int n2 = pow2_at_least(D > base ? D : base);
        lpw = n2 / base < 64 ? n2 / base : 64;
        epl = n2 / lpw;
        const int lpw_log = ilog2(lpw), epl_shift = ilog2(epl / base);
        if (epl_shift >= kMaxEplShift || !table->de_update[lpw_log][epl_shift])
            return fail(MCMCPP_HIP_E_UNSUPPORTED, "no differential-evolution kernel for D=%d with this calculator (LPW=%d EPL=%d)", D, lpw, epl);
        update_fn = table->de_update[lpw_log][epl_shift];
        calc_fn = table->calc[lpw_log][epl_shift];
        vec_ok = (D % base == 0) ? 1 : 0;

        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MCMCPP_HIP_E_NO_DEVICE, "no HIP device visible to this process");
        if (c.device >= ndev) return fail(MCMCPP_HIP_E_NO_DEVICE, "device %d out of range (%d visible)", c.device, ndev);
        if (c.device >= 0)
            device = c.device;
        else
            HIP_TRY(hipGetDevice(&device));
        HIP_TRY(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(MCMCPP_HIP_E_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        if (c.flags & MCMCPP_HIP_FLAG_CALLER_STREAM)
            stream = static_cast<hipStream_t>(c.hip_stream);
        else
        {
            HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            own_stream = true;
        }

        HIP_TRY(hipMalloc(&d_pos, sizeof(T) * (size_t)W * D));
        HIP_TRY(hipMalloc(&d_logp, sizeof(T) * (size_t)W));
        HIP_TRY(hipMalloc(&d_nacc, sizeof(uint32_t) * (size_t)W));
        HIP_TRY(hipMalloc(&d_diag, sizeof(Diag)));
        HIP_TRY(hipMalloc(&d_ctl, sizeof(DeCtl) * 2));
        HIP_TRY(hipMalloc(&d_cand, sizeof(DeCand) * kDeMaxCand));
        HIP_TRY(hipMemset(d_nacc, 0, sizeof(uint32_t) * (size_t)W));
        HIP_TRY(hipMemset(d_diag, 0, sizeof(Diag)));
        if (c.calc_params_len > 0)
        {
            // (the dense Gaussian's matrix goes over transposed, as for the stretch kernels: see DenseGaussianFn)
            std::vector<T> prm((const T*)c.calc_params, (const T*)c.calc_params + c.calc_params_len);
            if (c.calc_id == MCMCPP_HIP_CALC_DENSE_GAUSSIAN)
            {
                const T* p = (const T*)c.calc_params;
                for (int i = 0; i < D; ++i)
                    for (int j = 0; j < D; ++j) prm[(size_t)j * D + i] = p[(size_t)i * D + j];
            }
            HIP_TRY(hipMalloc(&d_params, sizeof(T) * prm.size()));
            HIP_TRY(hipMemcpy(d_params, prm.data(), sizeof(T) * prm.size(), hipMemcpyHostToDevice));
        }

        // the stream (MultiSampler.h:54) and its jump tables: D + 3 draws per update
        pcg_seed(c.seed, c.stream, &state0, &inc);
        const unsigned per = (unsigned)D + 3u;
        {
            std::vector<Affine128> lo(256), hi((size_t)(n + 255) / 256), small((size_t)D + kDeRaw + 1);
            Affine128 id;
            id.mult = make_u128(0, 1);
            id.plus = make_u128(0, 0);
            const Affine128 step_u = pcg_jump(inc, per), step_b = pcg_jump(inc, (unsigned __int128)per * 256u), step_1 = pcg_jump(inc, 1);
            lo[0] = hi[0] = small[0] = id;
            for (size_t j = 1; j < lo.size(); ++j) lo[j] = compose(step_u, lo[j - 1]);
            for (size_t m = 1; m < hi.size(); ++m) hi[m] = compose(step_b, hi[m - 1]);
            for (size_t j = 1; j < small.size(); ++j) small[j] = compose(step_1, small[j - 1]);
            HIP_TRY(hipMalloc(&d_jump_lo, sizeof(Affine128) * lo.size()));
            HIP_TRY(hipMalloc(&d_jump_hi, sizeof(Affine128) * hi.size()));
            HIP_TRY(hipMalloc(&d_jump_small, sizeof(Affine128) * small.size()));
            HIP_TRY(hipMemcpy(d_jump_lo, lo.data(), sizeof(Affine128) * lo.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(d_jump_hi, hi.data(), sizeof(Affine128) * hi.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(d_jump_small, small.data(), sizeof(Affine128) * small.size(), hipMemcpyHostToDevice));
        }
        half_jump = pcg_jump(inc, (unsigned __int128)per * (unsigned)n);
        threshold = (uint64_t)(0 - (uint64_t)n) % (uint64_t)n;
        gamma = (T)(2.38 / std::sqrt((double)(2 * D)));  // DifferentialEvolution.h:57
        return MCMCPP_HIP_OK;
    }

    int set_state(const void* pos, const void* logp) override
    {
        if (!pos || !logp) return fail(MCMCPP_HIP_E_ARG, "set_state: null pointer");
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipMemcpyAsync(d_pos, pos, sizeof(T) * (size_t)W * D, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(d_logp, logp, sizeof(T) * (size_t)W, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(d_nacc, 0, sizeof(uint32_t) * (size_t)W, stream));
        HIP_TRY(hipMemsetAsync(d_diag, 0, sizeof(Diag), stream));
        DeCtl c[2];
        std::memset(c, 0, sizeof c);
        c[0].state = state0;
        HIP_TRY(hipMemcpyAsync(d_ctl, c, sizeof c, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        half_steps = 0;
        steps_since_reset = 0;
        have_state = true;
        return MCMCPP_HIP_OK;
    }

    // EnsembleSampler::runMCMC (EnsembleSampler.h:284-310): interval-1 unsaved ensemble steps, one saved, n_saved times
    int run(int64_t n_saved, int32_t interval, void* chain_out, uint32_t* accepted_per_step) override
    {
        if (!have_state) return fail(MCMCPP_HIP_E_STATE, "run: set_state has not been called");
        if (n_saved < 0 || interval < 1) return fail(MCMCPP_HIP_E_ARG, "run: n_saved >= 0 and interval >= 1 required");
        HIP_TRY(hipSetDevice(device));
        const int64_t total = n_saved * (int64_t)interval;
        last_ms = 0.0;
        last_launches = 0;
        if (total == 0) return MCMCPP_HIP_OK;
        const auto t0 = std::chrono::steady_clock::now();
        const size_t step_bytes = sizeof(T) * (size_t)W * D;
        // stored steps leave in pieces of at most 256 MiB of device chain and 64 MiB of accepted counters
        int64_t piece = n_saved;
        if (accepted_per_step)
        {
            const int64_t per_stored = (int64_t)interval * kDeAccSlots * (int64_t)sizeof(uint32_t);
            if (per_stored > ((int64_t)1 << 30))
                return fail(MCMCPP_HIP_E_UNSUPPORTED, "run: interval %d is too large to report accepted counts per step; pass NULL for accepted_per_step", interval);
            const int64_t fit = ((int64_t)64 << 20) / per_stored;
            if (piece > fit) piece = fit < 1 ? 1 : fit;
        }
        if (chain_out)
        {
            const int64_t fit = (int64_t)(((size_t)256 << 20) / step_bytes);
            if (piece > fit) piece = fit < 1 ? 1 : fit;
            if ((size_t)piece * step_bytes > chain_bytes)
            {
                if (d_chain) HIP_TRY(hipFree(d_chain));
                d_chain = nullptr;
                chain_bytes = 0;
                HIP_TRY(hipMalloc(&d_chain, (size_t)piece * step_bytes));
                chain_bytes = (size_t)piece * step_bytes;
            }
        }
        const int64_t acc_piece = accepted_per_step ? piece * (int64_t)interval * kDeAccSlots : 0;
        if (accepted_per_step && (size_t)acc_piece > acc_count)
        {
            if (d_acc) HIP_TRY(hipFree(d_acc));
            d_acc = nullptr;
            acc_count = 0;
            HIP_TRY(hipMalloc(&d_acc, sizeof(uint32_t) * (size_t)acc_piece));
            acc_count = (size_t)acc_piece;
        }
        DeArgs<T> a;
        std::memset(&a, 0, sizeof a);
        a.pos = d_pos;
        a.logp = d_logp;
        a.n_accept = d_nacc;
        a.calc_params = d_params;
        a.cand = d_cand;
        a.half_jump = half_jump;
        a.jump_hi = d_jump_hi;
        a.jump_lo = d_jump_lo;
        a.jump_small = d_jump_small;
        a.diag = d_diag;
        a.chain = d_chain;
        a.threshold = threshold;
        a.inc = inc;
        a.gamma = gamma;
        a.jitter_width = (T)2.0e-4;  // DifferentialEvolution.h:120-121
        a.jitter_low = (T)-1.0e-4;
        a.tie_eps = sizeof(T) == 8 ? (T)1e-12 : (T)6e-7;
        a.n = n;
        a.dims = D;
        a.vec_ok = vec_ok;
        // a wavefront per walker, at most 8 wavefronts per SIMD's worth of workgroups
        const unsigned plan_grid = (unsigned)(n / 4 < 2048 ? (n + 3) / 4 : 2048);
        const int per_block = (64 / lpw) * kWavesPerBlock;
        const unsigned update_grid = (unsigned)((n + per_block - 1) / per_block);
        for (int64_t first = 0; first < n_saved; first += piece)
        {
            const int64_t now = n_saved - first < piece ? n_saved - first : piece;
            if (accepted_per_step) HIP_TRY(hipMemsetAsync(d_acc, 0, sizeof(uint32_t) * (size_t)(now * interval) * kDeAccSlots, stream));
            for (int64_t s = 0; s < now; ++s)
                for (int32_t j = 0; j < interval; ++j)
                {
                    const bool save = chain_out && j == interval - 1;
                    for (int color = 0; color < 2; ++color)
                    {
                        DeCtl* cur = d_ctl + (half_steps & 1);
                        DeCtl* nxt = d_ctl + ((half_steps + 1) & 1);
                        hipLaunchKernelGGL(de_plan_kernel, dim3(plan_grid), dim3(256), 0, stream, cur, d_cand, d_jump_hi, d_jump_lo, d_jump_small, threshold, n);
                        a.ctl = cur;
                        a.ctl_next = nxt;
                        a.color = color;
                        a.save_slot = save ? (long long)s : -1;
                        a.accepted = accepted_per_step ? d_acc + (s * interval + j) * kDeAccSlots : nullptr;
                        update_fn(a, update_grid, stream);
                        ++half_steps;
                        last_launches += 2;
                    }
                }
            HIP_TRY(hipGetLastError());
            if (chain_out)
                HIP_TRY(hipMemcpyAsync(static_cast<char*>(chain_out) + (size_t)first * step_bytes, d_chain, (size_t)now * step_bytes, hipMemcpyDeviceToHost, stream));
            if (accepted_per_step)
            {
                acc_host.resize((size_t)(now * interval) * kDeAccSlots);
                HIP_TRY(hipMemcpyAsync(acc_host.data(), d_acc, sizeof(uint32_t) * acc_host.size(), hipMemcpyDeviceToHost, stream));
            }
            HIP_TRY(hipStreamSynchronize(stream));
            if (accepted_per_step)
                for (int64_t e = 0; e < now * interval; ++e)
                {
                    uint32_t sum = 0;
                    for (int q = 0; q < kDeAccSlots; ++q) sum += acc_host[(size_t)e * kDeAccSlots + q];
                    accepted_per_step[first * interval + e] = sum;
                }
        }
        steps_since_reset += (uint64_t)total;
        DeCtl c;
        HIP_TRY(hipMemcpy(&c, d_ctl + (half_steps & 1), sizeof c, hipMemcpyDeviceToHost));
        last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (c.error)
        {
            have_state = false;
            return fail(MCMCPP_HIP_E_UNSUPPORTED,
                        "differential evolution: the random stream could not be followed (flags %u: 1 = more than %d draws thrown away in one "
                        "half-step, 2 = more than %d candidates, 4 = one update threw away more than %d draws); the state is undefined, call set_state",
                        c.error, kDeMaxShift, kDeMaxCand, kDeWindow - 2);
        }
        return MCMCPP_HIP_OK;
    }

    int get_state(void* pos, void* logp, uint32_t* n_accept) override
    {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamSynchronize(stream));
        if (pos) HIP_TRY(hipMemcpy(pos, d_pos, sizeof(T) * (size_t)W * D, hipMemcpyDeviceToHost));
        if (logp) HIP_TRY(hipMemcpy(logp, d_logp, sizeof(T) * (size_t)W, hipMemcpyDeviceToHost));
        if (n_accept) HIP_TRY(hipMemcpy(n_accept, d_nacc, sizeof(uint32_t) * (size_t)W, hipMemcpyDeviceToHost));
        return MCMCPP_HIP_OK;
    }

    int reset_counters() override
    {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipMemsetAsync(d_nacc, 0, sizeof(uint32_t) * (size_t)W, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        steps_since_reset = 0;
        return MCMCPP_HIP_OK;
    }

    int seek(uint64_t) override
    {
        return fail(MCMCPP_HIP_E_UNSUPPORTED, "seek: with the differential-evolution mover the stream position depends on the draws thrown away so far");
    }

    int get_counters(uint64_t* accepted, uint64_t* steps, uint64_t* ties, uint64_t* redraws) override
    {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamSynchronize(stream));
        if (accepted)
        {
            std::vector<uint32_t> acc((size_t)W);
            HIP_TRY(hipMemcpy(acc.data(), d_nacc, sizeof(uint32_t) * (size_t)W, hipMemcpyDeviceToHost));
            uint64_t s = 0;
            for (uint32_t v : acc) s += v;
            *accepted = s;
        }
        if (steps) *steps = steps_since_reset;
        if (ties)
        {
            Diag d;
            HIP_TRY(hipMemcpy(&d, d_diag, sizeof(Diag), hipMemcpyDeviceToHost));
            *ties = d.near_ties;
        }
        if (redraws)
        {
            DeCtl c;
            HIP_TRY(hipMemcpy(&c, d_ctl + (half_steps & 1), sizeof c, hipMemcpyDeviceToHost));
            *redraws = c.extra_total;  // every draw thrown away so far (bounded_rand, ind2 == ind1)
        }
        return MCMCPP_HIP_OK;
    }

    int calc_logp(const void* pos, int64_t count, void* out) override
    {
        if (count < 0 || (count > 0 && (!pos || !out))) return fail(MCMCPP_HIP_E_ARG, "calc_logp: bad arguments");
        if (count == 0) return MCMCPP_HIP_OK;
        HIP_TRY(hipSetDevice(device));
        struct Scratch  // freed on every way out
        {
            T *rows = nullptr, *out = nullptr;
            ~Scratch()
            {
                if (rows) (void)hipFree(rows);
                if (out) (void)hipFree(out);
            }
        } scratch;
        HIP_TRY(hipMalloc(&scratch.rows, sizeof(T) * (size_t)count * D));
        HIP_TRY(hipMalloc(&scratch.out, sizeof(T) * (size_t)count));
        T *dp = scratch.rows, *dout = scratch.out;
        HIP_TRY(hipMemcpyAsync(dp, pos, sizeof(T) * (size_t)count * D, hipMemcpyHostToDevice, stream));
        const long long per_block = (long long)(64 / lpw) * kWavesPerBlock;
        calc_fn(dp, dout, d_params, count, D, vec_ok, (unsigned)((count + per_block - 1) / per_block), stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out, dout, sizeof(T) * (size_t)count, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return MCMCPP_HIP_OK;
    }

    int last_run_timing(double* ms, int64_t* launches) override
    {
        if (ms) *ms = last_ms;
        if (launches) *launches = last_launches;
        return MCMCPP_HIP_OK;
    }
    int half_step_async(int32_t, int64_t) override { return unsupported("half_step_async"); }
    int bind_device_chain(void*, int64_t) override { return unsupported("bind_device_chain"); }
    void* device_positions() override { return d_pos; }
    int shard_span(int32_t color, int64_t* off, int64_t* cnt) override
    {
        if (color != 0 && color != 1) return fail(MCMCPP_HIP_E_ARG, "shard_span: colour must be 0 or 1");
        if (off) *off = (int64_t)(color ? n : 0) * D;
        if (cnt) *cnt = (int64_t)n * D;
        return MCMCPP_HIP_OK;
    }
    int synchronize() override
    {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamSynchronize(stream));
        return MCMCPP_HIP_OK;
    }
    int debug_stamps(unsigned long long*) override { return unsupported("debug_stamps"); }

private:
    int unsupported(const char* what) { return fail(MCMCPP_HIP_E_UNSUPPORTED, "%s: not available with the differential-evolution mover", what); }
    void release()
    {
        if (device >= 0) (void)hipSetDevice(device);
        if (stream && own_stream) (void)hipStreamSynchronize(stream);
        void* bufs[] = {d_pos, d_logp, d_nacc, d_diag, d_ctl, d_cand, d_params, d_jump_lo, d_jump_hi, d_jump_small, d_chain, d_acc};
        for (void* b : bufs)
            if (b) (void)hipFree(b);
        if (stream && own_stream) (void)hipStreamDestroy(stream);
    }

    const LaunchTable<T>* table = nullptr;
    typename LaunchTable<T>::DeFn update_fn = nullptr;
    typename LaunchTable<T>::CalcFn calc_fn = nullptr;
    int W = 0, D = 0, n = 0, lpw = 1, epl = 1, vec_ok = 0, device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false, have_state = false;
    T *d_pos = nullptr, *d_logp = nullptr, *d_params = nullptr, *d_chain = nullptr;
    uint32_t *d_nacc = nullptr, *d_acc = nullptr;
    Diag* d_diag = nullptr;
    DeCtl* d_ctl = nullptr;
    DeCand* d_cand = nullptr;
    Affine128 *d_jump_lo = nullptr, *d_jump_hi = nullptr, *d_jump_small = nullptr;
    size_t chain_bytes = 0, acc_count = 0;
    std::vector<uint32_t> acc_host;
    U128 state0, inc;
    Affine128 half_jump;
    uint64_t threshold = 0, half_steps = 0, steps_since_reset = 0;
    T gamma = 0;
    double last_ms = 0.0;
    int64_t last_launches = 0;
};
}  // namespace

namespace mcmcpp
{
mcmcpp_hip_sampler* make_de_sampler(const mcmcpp_hip_config& cfg, int* rc)
{
    if (cfg.dtype == MCMCPP_HIP_F64)
    {
        DeSampler<double>* s = new (std::nothrow) DeSampler<double>();
        if (s) *rc = s->init(cfg);
        return s;
    }
    DeSampler<float>* s = new (std::nothrow) DeSampler<float>();
    if (s) *rc = s->init(cfg);
    return s;
}
}  // namespace mcmcpp

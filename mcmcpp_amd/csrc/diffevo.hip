// diffevo.hip -- host side of Mover::DifferentialEvolution on gfx950 (reference
// MCMCpp/Movers/DifferentialEvolution.h:80-112 inside EnsembleSampler::performStep, EnsembleSampler.h:342-354).
// See diffevo_kernel.hpp for the scheme: ONE launch per half-step (update of half-step h beside the stream planning of
// h + 1 and h + 2), replayed from a hipGraph; the stream position, the error flags and the per-run counters travel in
// device memory.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "launch_table.hpp"
#include "sampler_base.hpp"

using namespace mcmcpp;

namespace
{
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
        HIP_TRY(hipMalloc(&d_ctl, sizeof(DeCtl) * 4));
        HIP_TRY(hipMalloc(&d_shared, sizeof(DeShared)));
        // a half-step has about D + 3 bad positions (one per walker collides with probability 1/n, (D+3)n positions)
        bad_capacity = 8 * (D + 3) + 512;
        HIP_TRY(hipMalloc(&d_bad, sizeof(DeBad) * 2 * (size_t)bad_capacity));
        HIP_TRY(hipMemset(d_bad, 0, sizeof(DeBad) * 2 * (size_t)bad_capacity));
        HIP_TRY(hipMalloc(&d_recs, sizeof(DeRec<T>) * 2 * (size_t)n));
        // the two per-step records and, right behind them, the run's constants: one allocation (the kernel reads both
        // through one preloaded pointer)
        HIP_TRY(hipMalloc(&d_step, sizeof(DeStepCtl) * 2 + sizeof(DeRunInfo)));
        d_run = reinterpret_cast<DeRunInfo*>(d_step + 2);
        HIP_TRY(hipMemset(d_shared, 0, sizeof(DeShared)));
        for (int k = 0; k < 2; ++k)
        {
            HIP_TRY(hipEventCreate(&ev_t0[k]));
            HIP_TRY(hipEventCreate(&ev_t1[k]));
        }
        // HIP cannot capture on the legacy default stream: a caller that hands it over gets plain launches
        {
            // MCMCPP_HIP_DE_FIND_WALKERS: runs of kDeScanRun stream positions one scanning lane takes (read once, here)
            const char* v = std::getenv("MCMCPP_HIP_DE_FIND_WALKERS");
            knob_walkers_per_find_wave = (v && *v) ? (int)std::strtol(v, nullptr, 10) : 1;
            if (knob_walkers_per_find_wave < 1) knob_walkers_per_find_wave = 1;
            v = std::getenv("MCMCPP_HIP_DE_SCAN_RUN");  // stream positions one scanning lane steps through
            scan_run = (v && *v) ? (int)std::strtol(v, nullptr, 10) : kDeScanRun;
            if (scan_run < 1) scan_run = 1;
            v = std::getenv("MCMCPP_HIP_DE_DEBUG");
            knob_debug = (v && *v) ? (int)std::strtol(v, nullptr, 10) : 0;
        }
        graph_steps = c.graph_steps == 0 ? 128 : c.graph_steps;
        if (!own_stream && (stream == nullptr || stream == hipStreamLegacy)) graph_steps = -1;
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
            scan_positions = (int)per * n + kDeMaxShift + 1;
            const size_t scan_lanes = ((size_t)scan_positions + scan_run - 1) / scan_run;
            std::vector<Affine128> lo(256), hi((size_t)(n + 255) / 256), small((size_t)(D > kDeMaxShift ? D : kDeMaxShift) + 2);
            std::vector<Affine128> slo(256), shi((scan_lanes + 255) / 256);
            Affine128 id;
            id.mult = make_u128(0, 1);
            id.plus = make_u128(0, 0);
            const Affine128 step_u = pcg_jump(inc, per), step_b = pcg_jump(inc, (unsigned __int128)per * 256u), step_1 = pcg_jump(inc, 1);
            lo[0] = hi[0] = small[0] = id;
            for (size_t j = 1; j < lo.size(); ++j) lo[j] = compose(step_u, lo[j - 1]);
            for (size_t m = 1; m < hi.size(); ++m) hi[m] = compose(step_b, hi[m - 1]);
            for (size_t j = 1; j < small.size(); ++j) small[j] = compose(step_1, small[j - 1]);
            const Affine128 step_r = pcg_jump(inc, (unsigned)scan_run), step_rb = pcg_jump(inc, (unsigned __int128)scan_run * 256u);
            slo[0] = shi[0] = id;
            for (size_t j = 1; j < slo.size(); ++j) slo[j] = compose(step_r, slo[j - 1]);
            for (size_t m = 1; m < shi.size(); ++m) shi[m] = compose(step_rb, shi[m - 1]);
            HIP_TRY(hipMalloc(&d_scan_lo, sizeof(Affine128) * slo.size()));
            HIP_TRY(hipMalloc(&d_scan_hi, sizeof(Affine128) * shi.size()));
            HIP_TRY(hipMemcpy(d_scan_lo, slo.data(), sizeof(Affine128) * slo.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(d_scan_hi, shi.data(), sizeof(Affine128) * shi.size(), hipMemcpyHostToDevice));
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
        DeCtl c[4];
        std::memset(c, 0, sizeof c);
        c[0].state = state0;
        HIP_TRY(hipMemcpyAsync(d_ctl, c, sizeof c, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(d_shared, 0, sizeof(DeShared), stream));
        HIP_TRY(hipStreamSynchronize(stream));
        half_steps = 0;
        steps_since_reset = 0;
        have_state = true;
        return MCMCPP_HIP_OK;
    }

    // EnsembleSampler::runMCMC (EnsembleSampler.h:284-310): interval-1 unsaved ensemble steps, one saved, n_saved times
    int run(int64_t n_saved, int32_t interval, void* chain_out, uint32_t* accepted_per_step) override
    {
        run_touched = false;
        const int rc = run_steps(n_saved, interval, chain_out, accepted_per_step);
        if (rc != MCMCPP_HIP_OK && run_touched)
        {
            // launches went out and the call failed: the walkers are ahead of the host's counters (and the stream may
            // be left capturing) -- nothing on the device can be trusted until the next set_state
            hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
            {
                hipGraph_t g = nullptr;
                (void)hipStreamEndCapture(stream, &g);
                if (g) (void)hipGraphDestroy(g);
            }
            (void)hipStreamSynchronize(stream);
            (void)hipGetLastError();
            have_state = false;
        }
        return rc;
    }

    int run_steps(int64_t n_saved, int32_t interval, void* chain_out, uint32_t* accepted_per_step)
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
        DeArgs<T>& a = args;
        std::memset(&a, 0, sizeof a);
        a.pos = d_pos;
        a.logp = d_logp;
        a.n_accept = d_nacc;
        a.calc_params = d_params;
        a.ctl = d_ctl;
        a.shared = d_shared;
        a.bad = d_bad;
        a.bad_capacity = bad_capacity;
        a.scan_positions = scan_positions;
        a.scan_run = scan_run;
        a.scan_hi = d_scan_hi;
        a.scan_lo = d_scan_lo;
        a.recs = d_recs;
        a.run = d_run;
        a.step_ctl = d_step;
        a.half_jump = half_jump;
        a.jump_hi = d_jump_hi;
        a.jump_lo = d_jump_lo;
        a.jump_small = d_jump_small;
        a.diag = d_diag;
        a.threshold = threshold;
        a.inc = inc;
        a.gamma = gamma;
        a.jitter_width = (T)2.0e-4;  // DifferentialEvolution.h:120-121
        a.jitter_low = (T)-1.0e-4;
        a.tie_eps = sizeof(T) == 8 ? (T)1e-12 : (T)6e-7;
        a.n = n;
        a.dims = D;
        a.vec_ok = vec_ok;
        if (knob_debug == 3 && !d_debug) HIP_TRY(hipMalloc(&d_debug, sizeof(unsigned long long) * 2 * 8192));
        a.debug_times = d_debug;
        const int per_block = (64 / lpw) * kWavesPerBlock;
        update_blocks = (n + per_block - 1) / per_block;
        record_blocks = (n + 64 * kWavesPerBlock - 1) / (64 * kWavesPerBlock);
        // scanners: a lane per kDeScanRun stream positions (knob: that many runs per lane)
        {
            const long lanes = ((long)scan_positions + scan_run - 1) / scan_run;
            const long per_wg = 64L * kWavesPerBlock * knob_walkers_per_find_wave;
            find_blocks = (int)((lanes + per_wg - 1) / per_wg);
            if (find_blocks < 1) find_blocks = 1;
        }

        // Priming: the planners run one half-step (records) and two half-steps (candidates) ahead of the updates.  Two
        // planning-only launches bring them there from the stream position the ring holds for the coming half-step.
        run_touched = true;
        HIP_TRY(hipMemsetAsync(d_shared, 0, offsetof(DeShared, error), stream));  // (the bad-position counters; the error flags stay)
        launch_step((int)((half_steps + 3) & 3), /*with_update=*/false, /*with_records=*/false);  // bad positions of the coming half-step
        launch_step((int)((half_steps + 3) & 3), false, true);                                     // its records, positions of the next
        HIP_TRY(hipGetLastError());

        double gpu_ms = 0.0;
        for (int64_t first = 0; first < n_saved; first += piece)
        {
            const int64_t now = n_saved - first < piece ? n_saved - first : piece;
            if (accepted_per_step) HIP_TRY(hipMemsetAsync(d_acc, 0, sizeof(uint32_t) * (size_t)(now * interval) * kDeAccSlots, stream));
            {
                DeRunInfo ri;
                std::memset(&ri, 0, sizeof ri);
                ri.chain = chain_out ? d_chain : nullptr;
                ri.accepted = accepted_per_step ? d_acc : nullptr;
                ri.interval = interval;
                DeStepCtl sc[2];
                std::memset(sc, 0, sizeof sc);
                HIP_TRY(hipMemcpyAsync(d_run, &ri, sizeof ri, hipMemcpyHostToDevice, stream));
                HIP_TRY(hipMemcpyAsync(d_step, sc, sizeof sc, hipMemcpyHostToDevice, stream));
                HIP_TRY(hipStreamSynchronize(stream));  // (the sources are on this stack frame)
            }
            HIP_TRY(hipEventRecord(ev_t0[0], stream));
            int rc = enqueue_steps(now * interval);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ev_t1[0], stream));
            if (chain_out)
                HIP_TRY(hipMemcpyAsync(static_cast<char*>(chain_out) + (size_t)first * step_bytes, d_chain, (size_t)now * step_bytes, hipMemcpyDeviceToHost, stream));
            if (accepted_per_step)
            {
                acc_host.resize((size_t)(now * interval) * kDeAccSlots);
                HIP_TRY(hipMemcpyAsync(acc_host.data(), d_acc, sizeof(uint32_t) * acc_host.size(), hipMemcpyDeviceToHost, stream));
            }
            HIP_TRY(hipStreamSynchronize(stream));
            {
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, ev_t0[0], ev_t1[0]));
                gpu_ms += ms;
            }
            if (chain_out) publish_stored(first + now);
            if (accepted_per_step)
                for (int64_t e = 0; e < now * interval; ++e)
                {
                    uint32_t sum = 0;
                    for (int q = 0; q < kDeAccSlots; ++q) sum += acc_host[(size_t)e * kDeAccSlots + q];
                    accepted_per_step[first * interval + e] = sum;
                }
        }
        steps_since_reset += (uint64_t)total;
        if (d_debug)
        {
            // diagnostics: when the workgroups of the LAST launch started and ended, by role (us from the first start)
            const int grid = update_blocks + record_blocks + find_blocks;
            std::vector<unsigned long long> t((size_t)2 * grid);
            HIP_TRY(hipMemcpy(t.data(), d_debug, sizeof(unsigned long long) * t.size(), hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ULL;
            for (int b = 0; b < grid; ++b) t0 = t[2 * b] < t0 ? t[2 * b] : t0;
            const char* names[3] = {"records", "scan", "update"};
            const int lo[3] = {0, record_blocks, record_blocks + find_blocks}, hi[3] = {record_blocks, record_blocks + find_blocks, grid};
            for (int r = 0; r < 3; ++r)
            {
                double s0 = 1e9, s1 = 0, e0 = 1e9, e1 = 0;
                for (int b = lo[r]; b < hi[r]; ++b)
                {
                    const double st = (t[2 * b] - t0) * 0.01, en = (t[2 * b + 1] - t0) * 0.01;
                    s0 = st < s0 ? st : s0, s1 = st > s1 ? st : s1, e0 = en < e0 ? en : e0, e1 = en > e1 ? en : e1;
                }
                std::fprintf(stderr, "[de debug] %-8s %4d workgroups: start %.2f..%.2f us, end %.2f..%.2f us\n", names[r], hi[r] - lo[r], s0, s1, e0, e1);
            }
        }
        DeShared sh;
        HIP_TRY(hipMemcpy(&sh, d_shared, sizeof sh, hipMemcpyDeviceToHost));
        last_ms = gpu_ms;  // GPU time of the step launches between HIP events on the launch stream (transfers excluded)
        last_launches = 2 * total;
        host_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (sh.error)
        {
            have_state = false;
            return fail(MCMCPP_HIP_E_UNSUPPORTED,
                        "differential evolution: the random stream could not be followed (flags %u: 1 = more than %d draws thrown away in one "
                        "half-step, 2 = more than %d possible starts among the bad positions, 4 = one update threw away more than %d draws); the state is undefined, call set_state",
                        sh.error, kDeMaxShift, kDeMaxEvents, kDeWindow - 2);
        }
        return MCMCPP_HIP_OK;
    }

    // one launch: the update of half-step `h4` (mod 4) beside the planning of the two half-steps behind it
    void launch_step(int h4, bool with_update, bool with_records)
    {
        DeArgs<T> a = args;
        a.half_step_mod4 = h4;
        a.update_blocks = with_update ? update_blocks : 0;
        a.record_blocks = with_records ? record_blocks : 0;
        unsigned grid = (unsigned)(a.update_blocks + a.record_blocks + find_blocks);
        // timing diagnostics only (MCMCPP_HIP_DE_DEBUG; the chain is wrong): 1 = regular launches without planners, 2 = without updates
        if (with_update && knob_debug == 1) grid = (unsigned)a.update_blocks;
        if (with_update && knob_debug == 2) a.update_blocks = 0, grid = (unsigned)(a.record_blocks + find_blocks);
        update_fn(a, grid, stream);
    }

    // hipGraph of `steps` ensemble steps that start at half-step residue `r0` (0 or 2)
    int graph_for(int steps, int r0, hipGraphExec_t* out)
    {
        const size_t key = (size_t)steps * 2 + (size_t)(r0 >> 1);
        if (graph_cache.size() <= key) graph_cache.resize(key + 1, nullptr);
        if (!graph_cache[key])
        {
            hipGraph_t g = nullptr;
            HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
            for (int s = 0; s < steps; ++s)
                for (int c = 0; c < 2; ++c) launch_step((r0 + 2 * s + c) & 3, true, true);
            HIP_TRY(hipStreamEndCapture(stream, &g));
            hipGraphExec_t ex = nullptr;
            HIP_TRY(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
            HIP_TRY(hipGraphDestroy(g));
            graph_cache[key] = ex;
        }
        *out = graph_cache[key];
        return MCMCPP_HIP_OK;
    }

    int enqueue_steps(int64_t steps)
    {
        int64_t left = steps;
        while (left > 0)
        {
            const int r0 = (int)(half_steps & 3);
            if (graph_steps >= 1)
            {
                const int now = (int)(left < graph_steps ? left : graph_steps);
                hipGraphExec_t ex = nullptr;
                int rc = graph_for(now, r0, &ex);
                if (rc) return rc;
                HIP_TRY(hipGraphLaunch(ex, stream));
                half_steps += 2 * (uint64_t)now;
                left -= now;
            }
            else
            {
                launch_step(r0, true, true);
                launch_step((r0 + 1) & 3, true, true);
                half_steps += 2;
                left -= 1;
            }
        }
        HIP_TRY(hipGetLastError());
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
            HIP_TRY(hipMemcpy(&c, d_ctl + (half_steps & 3), sizeof c, hipMemcpyDeviceToHost));
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
        for (hipGraphExec_t ex : graph_cache)
            if (ex) (void)hipGraphExecDestroy(ex);
        void* bufs[] = {d_pos, d_logp, d_nacc, d_diag, d_ctl, d_params, d_jump_lo, d_jump_hi, d_jump_small, d_chain, d_acc, d_shared, d_recs, d_step, d_bad, d_scan_lo, d_scan_hi, d_debug};
        for (void* b : bufs)
            if (b) (void)hipFree(b);
        for (int k = 0; k < 2; ++k)
        {
            if (ev_t0[k]) (void)hipEventDestroy(ev_t0[k]);
            if (ev_t1[k]) (void)hipEventDestroy(ev_t1[k]);
        }
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
    DeShared* d_shared = nullptr;
    DeBad* d_bad = nullptr;
    unsigned long long* d_debug = nullptr;
    Affine128 *d_scan_lo = nullptr, *d_scan_hi = nullptr;
    int bad_capacity = 0, scan_positions = 0, scan_run = kDeScanRun;
    DeRec<T>* d_recs = nullptr;
    DeRunInfo* d_run = nullptr;
    DeStepCtl* d_step = nullptr;
    DeArgs<T> args;
    int update_blocks = 0, record_blocks = 0, find_blocks = 1, graph_steps = 128, knob_walkers_per_find_wave = 2, knob_debug = 0;
    bool run_touched = false;
    hipEvent_t ev_t0[2] = {nullptr, nullptr}, ev_t1[2] = {nullptr, nullptr};
    std::vector<hipGraphExec_t> graph_cache;

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

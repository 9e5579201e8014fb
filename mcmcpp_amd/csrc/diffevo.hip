// diffevo.hip -- host side of Mover::DifferentialEvolution on gfx950 (reference
// MCMCpp/Movers/DifferentialEvolution.h:80-112 inside EnsembleSampler::performStep, EnsembleSampler.h:342-354).
// See diffevo_kernel.hpp for the scheme: the random stream is planned a batch of half-steps at a time (scan, resolve,
// records) on a second HIP stream beside the update launches of the batch before, one update launch per half-step; a
// run is replayed from hipGraphs whose edges are the event waits between the two streams; the stream head, the error
// flags and the per-run counters travel in device memory.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "diffevo_plan.hpp"
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

        HIP_TRY(hipStreamCreateWithFlags(&plan_stream, hipStreamNonBlocking));
        {
            const char* v = std::getenv("MCMCPP_HIP_DE_SCAN_RUN");  // stream positions one scanning lane steps through
            scan_run = (v && *v) ? (int)std::strtol(v, nullptr, 10) : kDeScanRun;
            if (scan_run < 1) scan_run = 1;
            v = std::getenv("MCMCPP_HIP_DE_BATCH");  // half-steps planned by one batch of planning launches
            batch_max = (v && *v) ? (int)std::strtol(v, nullptr, 10) : kDeBatchMax;
            v = std::getenv("MCMCPP_HIP_DE_DEBUG");  // timing diagnostics (the chain is wrong): 1 = the update launches alone, 2 = the planning alone
            knob_debug = (v && *v) ? (int)std::strtol(v, nullptr, 10) : 0;
            v = std::getenv("MCMCPP_HIP_DE_OVERLAP");  // 0: planning in the launch stream, in front of the updates it plans
            knob_overlap = (v && *v) ? (int)std::strtol(v, nullptr, 10) != 0 : true;
        }
        // the batch: as many half-steps as the position counters (32 bits), the resolver's lists and a sensible amount of
        // record memory (128 MiB) allow
        const unsigned per = (unsigned)D + 3u;
        {
            long long b = batch_max < 1 ? 1 : (batch_max > kDeBatchMax ? kDeBatchMax : batch_max);
            const long long by_positions = ((1LL << 30) - kDeShiftMax - 1) / ((long long)per * n);
            const long long by_lists = (kDeMaxBad * 5LL / 8) / per;
            const long long by_records = (4LL << 20) / n;
            b = b < by_positions ? b : by_positions;
            b = b < by_lists ? b : by_lists;
            b = b < by_records ? b : by_records;
            if (by_positions < 1) return fail(MCMCPP_HIP_E_UNSUPPORTED, "differential evolution: %d walkers x %d parameters exceed the planner's 2^30 stream positions per half-step", W, D);
            batch_max = b < 1 ? 1 : (int)b;
        }
        const size_t updates_max = (size_t)batch_max * n;
        positions_max = (long long)per * (long long)(updates_max - 1) + kDeShiftMax + 1;
        // a batch lists about (D + 3) bad positions per half-step (one position in n is bad), spread evenly over the lists
        bad_capacity = 4 * (int)((per * (unsigned)batch_max + kDeSegments - 1) / kDeSegments) + 64;

        HIP_TRY(hipMalloc(&d_pos, sizeof(T) * (size_t)W * D));
        HIP_TRY(hipMalloc(&d_logp, sizeof(T) * (size_t)W));
        HIP_TRY(hipMalloc(&d_nacc, sizeof(uint32_t) * (size_t)W));
        HIP_TRY(hipMalloc(&d_diag, sizeof(Diag)));
        HIP_TRY(hipMalloc(&d_head, sizeof(DeHead)));
        HIP_TRY(hipMalloc(&d_batch, sizeof(DeBatch)));
        HIP_TRY(hipMalloc(&d_counts, sizeof(uint32_t) * kDeSegments * kDeCountStride));
        HIP_TRY(hipMemset(d_counts, 0, sizeof(uint32_t) * kDeSegments * kDeCountStride));
        HIP_TRY(hipMalloc(&d_bad, sizeof(DeBad) * kDeSegments * (size_t)bad_capacity));
        HIP_TRY(hipMemset(d_bad, 0, sizeof(DeBad) * kDeSegments * (size_t)bad_capacity));
        HIP_TRY(hipMalloc(&d_recs, sizeof(DeRec<T>) * 2 * updates_max));  // two batches: one being planned, one being used
        for (int k = 0; k < 2; ++k)
        {
            HIP_TRY(hipEventCreate(&ev_t0[k]));
            HIP_TRY(hipEventCreate(&ev_t1[k]));
        }
        graph_steps = c.graph_steps == 0 ? 128 : c.graph_steps;
        // HIP cannot capture on the legacy default stream: a caller that hands it over gets plain launches
        if (!own_stream && (stream == nullptr || stream == hipStreamLegacy)) graph_steps = -1;
        replay_steps_max = graph_steps >= 1 ? graph_steps : 16;  // (plain launches: enqueued in groups of this many steps)
        const int per_block = (64 / lpw) * kWavesPerBlock;
        update_blocks = (n + per_block - 1) / per_block;
        partial_waves = update_blocks * kWavesPerBlock;
        // the run record and, right behind it, the wavefronts' accepted counts of a replay: one allocation (the kernel reaches
        // both through one preloaded pointer)
        {
            const size_t bytes = sizeof(DeRunInfo) + sizeof(uint32_t) * (size_t)replay_steps_max * 2 * (size_t)partial_waves;
            void* p = nullptr;
            HIP_TRY(hipMalloc(&p, bytes));
            d_run = static_cast<DeRunInfo*>(p);
            HIP_TRY(hipMemset(d_run, 0, bytes));
        }
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
        {
            const size_t scan_lanes = ((size_t)positions_max + scan_run - 1) / scan_run;
            std::vector<Affine128> lo(256), hi((updates_max + 255) / 256), small((size_t)kDeShiftMax + (size_t)D + 2);
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
        batch_jumps.resize((size_t)batch_max + 1);
        for (int b = 1; b <= batch_max; ++b) batch_jumps[(size_t)b] = pcg_jump(inc, (unsigned __int128)per * (unsigned)n * (unsigned)b);
        threshold = (uint64_t)(0 - (uint64_t)n) % (uint64_t)n;
        gamma = (T)(2.38 / std::sqrt((double)(2 * D)));  // DifferentialEvolution.h:57

        args.calc_params = d_params;
        args.diag = d_diag;
        args.inc = inc;
        args.gamma = gamma;
        args.jitter_width = (T)2.0e-4;  // DifferentialEvolution.h:120-121
        args.jitter_low = (T)-1.0e-4;
        args.tie_eps = sizeof(T) == 8 ? (T)1e-12 : (T)6e-7;
        args.partial_waves = partial_waves;
        return MCMCPP_HIP_OK;
    }

    int set_state(const void* pos, const void* logp) override
    {
        if (!pos || !logp) return fail(MCMCPP_HIP_E_ARG, "set_state: null pointer");
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamSynchronize(plan_stream));
        HIP_TRY(hipMemcpyAsync(d_pos, pos, sizeof(T) * (size_t)W * D, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(d_logp, logp, sizeof(T) * (size_t)W, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(d_nacc, 0, sizeof(uint32_t) * (size_t)W, stream));
        HIP_TRY(hipMemsetAsync(d_diag, 0, sizeof(Diag), stream));
        DeHead h;
        std::memset(&h, 0, sizeof h);
        h.state = state0;
        HIP_TRY(hipMemcpyAsync(d_head, &h, sizeof h, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(d_counts, 0, sizeof(uint32_t) * kDeSegments * kDeCountStride, stream));
        HIP_TRY(hipStreamSynchronize(stream));
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
            // launches went out and the call failed: the walkers are ahead of the host's counters (and the streams may
            // be left capturing) -- nothing on the device can be trusted until the next set_state
            hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
            {
                hipGraph_t g = nullptr;
                (void)hipStreamEndCapture(stream, &g);
                if (g) (void)hipGraphDestroy(g);
            }
            (void)hipStreamSynchronize(stream);
            (void)hipStreamSynchronize(plan_stream);
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
            const int64_t per_stored = (int64_t)interval * (int64_t)sizeof(uint32_t);
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
        const int64_t acc_piece = accepted_per_step ? piece * (int64_t)interval : 0;
        if (accepted_per_step && (size_t)acc_piece > acc_count)
        {
            if (d_acc) HIP_TRY(hipFree(d_acc));
            d_acc = nullptr;
            acc_count = 0;
            HIP_TRY(hipMalloc(&d_acc, sizeof(uint32_t) * (size_t)acc_piece));
            acc_count = (size_t)acc_piece;
        }

        run_touched = true;
        double gpu_ms = 0.0;
        for (int64_t first = 0; first < n_saved; first += piece)
        {
            const int64_t now = n_saved - first < piece ? n_saved - first : piece;
            {
                DeRunInfo ri;
                std::memset(&ri, 0, sizeof ri);
                ri.chain = chain_out ? d_chain : nullptr;
                ri.accepted = accepted_per_step ? d_acc : nullptr;
                ri.interval = (uint32_t)interval;
                HIP_TRY(hipMemcpyAsync(d_run, &ri, sizeof ri, hipMemcpyHostToDevice, stream));
                HIP_TRY(hipStreamSynchronize(stream));  // (the source is on this stack frame)
            }
            HIP_TRY(hipEventRecord(ev_t0[0], stream));
            int rc = enqueue_steps(now * interval);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ev_t1[0], stream));
            if (chain_out)
                HIP_TRY(hipMemcpyAsync(static_cast<char*>(chain_out) + (size_t)first * step_bytes, d_chain, (size_t)now * step_bytes, hipMemcpyDeviceToHost, stream));
            if (accepted_per_step)
                HIP_TRY(hipMemcpyAsync(accepted_per_step + first * interval, d_acc, sizeof(uint32_t) * (size_t)(now * interval), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            {
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, ev_t0[0], ev_t1[0]));
                gpu_ms += ms;
            }
            if (chain_out) publish_stored(first + now);
        }
        steps_since_reset += (uint64_t)total;
        DeHead h;
        HIP_TRY(hipMemcpy(&h, d_head, sizeof h, hipMemcpyDeviceToHost));
        last_ms = gpu_ms;  // GPU time of the launches (planning included) between HIP events on the launch stream (transfers excluded)
        last_launches = 2 * total;
        host_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (h.error)
        {
            have_state = false;
            return fail(MCMCPP_HIP_E_UNSUPPORTED,
                        "differential evolution: the random stream could not be followed (flags %u: 1 = more than %d draws thrown away in one "
                        "batch of half-steps, 2 = more bad stream positions or events than the planner's lists hold, 4 = one update threw away more than %d draws); the state "
                        "is undefined, call set_state",
                        h.error, kDeShiftMax, kDeWindow - 2);
        }
        return MCMCPP_HIP_OK;
    }

    // the three planning launches of `count` half-steps, records into buffer `buf`
    void launch_plan(int count, int buf, hipStream_t st)
    {
        const unsigned per = (unsigned)D + 3u;
        DePlanArgs p;
        std::memset(&p, 0, sizeof p);
        p.head = d_head;
        p.batch = d_batch;
        p.bad = d_bad;
        p.counts = d_counts;
        p.scan_hi = d_scan_hi;
        p.scan_lo = d_scan_lo;
        p.jump_hi = d_jump_hi;
        p.jump_lo = d_jump_lo;
        p.jump_small = d_jump_small;
        p.batch_jump = batch_jumps[(size_t)count];
        p.inc = inc;
        p.threshold = threshold;
        p.n = n;
        p.dims = D;
        p.updates = n * count;
        p.positions = (int)((long long)per * (long long)(p.updates - 1) + kDeShiftMax + 1);
        p.scan_run = scan_run;
        p.seg_len = (p.positions + kDeSegments - 1) / kDeSegments;
        p.bad_capacity = bad_capacity;
        const long long lanes = ((long long)p.positions + scan_run - 1) / scan_run;
        hipLaunchKernelGGL(de_scan_kernel, dim3((unsigned)((lanes + kDePlanThreads - 1) / kDePlanThreads)), dim3(kDePlanThreads), 0, st, p);
        hipLaunchKernelGGL(de_resolve_kernel, dim3(1), dim3(kDePlanThreads), 0, st, p);
        hipLaunchKernelGGL((de_records_kernel<T>), dim3((unsigned)((p.updates + kDePlanThreads - 1) / kDePlanThreads)), dim3(kDePlanThreads), 0, st, p,
                           d_recs + (size_t)buf * (size_t)batch_max * (size_t)n);
    }

    // `steps` ensemble steps (at most replay_steps_max) on the launch stream: batches of up to batch_max half-steps, the
    // planning of batch b + 1 on the planning stream beside the updates of batch b; behind them the accepted counts and
    // the run record.  Under stream capture this becomes the replay's graph (the event waits its edges).
    int enqueue_replay(int steps)
    {
        const int halves = 2 * steps;
        const int batches = (halves + batch_max - 1) / batch_max;
        while ((int)events.size() < 2 * batches + 1)
        {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            events.push_back(e);
        }
        hipEvent_t* planned = events.data();            // [batches]
        hipEvent_t* updated = events.data() + batches;  // [batches]
        hipEvent_t fork = events[2 * (size_t)batches];
        const bool overlap = knob_overlap && batches > 1;
        hipStream_t ps = overlap ? plan_stream : stream;
        auto count_of = [&](int b) { return halves - b * batch_max < batch_max ? halves - b * batch_max : batch_max; };
        if (overlap)
        {
            HIP_TRY(hipEventRecord(fork, stream));
            HIP_TRY(hipStreamWaitEvent(ps, fork, 0));
        }
        if (knob_debug != 1) launch_plan(count_of(0), 0, ps);
        if (overlap) HIP_TRY(hipEventRecord(planned[0], ps));
        typename LaunchTable<T>::DeLaunch l;
        l.pos = d_pos;
        l.logp = d_logp;
        l.n_accept = d_nacc;
        l.jump_small = d_jump_small;
        l.run = d_run;
        l.n = n;
        l.dims = D;
        l.vec_ok = vec_ok;
        for (int b = 0; b < batches; ++b)
        {
            if (b + 1 < batches)
            {
                // batch b + 1 is planned into the buffer batch b - 1 was read from
                if (overlap && b >= 1) HIP_TRY(hipStreamWaitEvent(ps, updated[b - 1], 0));
                if (knob_debug != 1) launch_plan(count_of(b + 1), (b + 1) & 1, ps);
                if (overlap) HIP_TRY(hipEventRecord(planned[b + 1], ps));
            }
            if (overlap) HIP_TRY(hipStreamWaitEvent(stream, planned[b], 0));
            const int count = count_of(b);
            for (int j = 0; j < count && knob_debug != 2; ++j)
            {
                const int h = b * batch_max + j;  // half-step inside the replay
                l.recs = d_recs + ((size_t)(b & 1) * (size_t)batch_max + (size_t)j) * (size_t)n;
                l.color = h & 1;
                l.step = h >> 1;
                update_fn(l, args, (unsigned)update_blocks, stream);
            }
            if (overlap && b + 2 < batches) HIP_TRY(hipEventRecord(updated[b], stream));
        }
        hipLaunchKernelGGL(de_accepted_kernel, dim3((unsigned)steps), dim3(256), 0, stream, d_run, partial_waves);
        hipLaunchKernelGGL(de_advance_kernel, dim3(1), dim3(1), 0, stream, d_run, steps);
        return MCMCPP_HIP_OK;
    }

    // hipGraph of `steps` ensemble steps
    int graph_for(int steps, hipGraphExec_t* out)
    {
        const size_t key = (size_t)steps;
        if (graph_cache.size() <= key) graph_cache.resize(key + 1, nullptr);
        if (!graph_cache[key])
        {
            hipGraph_t g = nullptr;
            HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
            int rc = enqueue_replay(steps);
            if (rc) return rc;
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
            const int now = (int)(left < replay_steps_max ? left : replay_steps_max);
            if (graph_steps >= 1)
            {
                hipGraphExec_t ex = nullptr;
                int rc = graph_for(now, &ex);
                if (rc) return rc;
                HIP_TRY(hipGraphLaunch(ex, stream));
            }
            else
            {
                int rc = enqueue_replay(now);
                if (rc) return rc;
            }
            left -= now;
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
            DeHead h;
            HIP_TRY(hipMemcpy(&h, d_head, sizeof h, hipMemcpyDeviceToHost));
            *redraws = h.extra_total;  // every draw thrown away so far (bounded_rand, ind2 == ind1)
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
        if (plan_stream) (void)hipStreamSynchronize(plan_stream);
        void* bufs[] = {d_pos, d_logp, d_nacc, d_diag, d_head, d_batch, d_counts, d_params, d_jump_lo, d_jump_hi, d_jump_small, d_chain, d_acc, d_recs, d_run, d_bad, d_scan_lo, d_scan_hi};
        for (void* b : bufs)
            if (b) (void)hipFree(b);
        for (int k = 0; k < 2; ++k)
        {
            if (ev_t0[k]) (void)hipEventDestroy(ev_t0[k]);
            if (ev_t1[k]) (void)hipEventDestroy(ev_t1[k]);
        }
        for (hipEvent_t e : events)
            if (e) (void)hipEventDestroy(e);
        if (plan_stream) (void)hipStreamDestroy(plan_stream);
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
    DeHead* d_head = nullptr;
    DeBatch* d_batch = nullptr;
    uint32_t* d_counts = nullptr;
    DeBad* d_bad = nullptr;
    Affine128 *d_scan_lo = nullptr, *d_scan_hi = nullptr;
    int bad_capacity = 0, scan_run = kDeScanRun, batch_max = kDeBatchMax;
    long long positions_max = 0;
    DeRec<T>* d_recs = nullptr;
    DeRunInfo* d_run = nullptr;
    DeArgs<T> args;
    int update_blocks = 0, partial_waves = 0, graph_steps = 128, replay_steps_max = 128, knob_debug = 0;
    bool knob_overlap = true;
    hipStream_t plan_stream = nullptr;  // the planning launches of the next batch, beside the updates of this one
    std::vector<hipEvent_t> events;     // fork/join between the two streams (graph edges under capture)
    std::vector<Affine128> batch_jumps; // [half-steps of a batch]: (D+3) * n * that many draws
    bool run_touched = false;
    hipEvent_t ev_t0[2] = {nullptr, nullptr}, ev_t1[2] = {nullptr, nullptr};
    std::vector<hipGraphExec_t> graph_cache;

    Affine128 *d_jump_lo = nullptr, *d_jump_hi = nullptr, *d_jump_small = nullptr;
    size_t chain_bytes = 0, acc_count = 0;
    U128 state0, inc;
    uint64_t threshold = 0, steps_since_reset = 0;
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

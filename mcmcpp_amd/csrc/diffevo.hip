// diffevo.hip -- host side of Mover::DifferentialEvolution on gfx950 (reference
// MCMCpp/Movers/DifferentialEvolution.h:80-112 inside EnsembleSampler::performStep, EnsembleSampler.h:342-354).
// See diffevo_kernel.hpp for the scheme: one update launch per half-step; the random stream is planned a batch of half-steps
// at a time (scan, resolve, records) by two launches at every batch boundary; a run is replayed from hipGraphs; the stream
// head, the error flags and the per-run counters travel in device memory.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <unordered_map>
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
        walkers_per_block = (64 / lpw) * kWavesPerBlock;
        {
            // the dense Gaussian's product on the matrix cores (de_update_mfma_kernel): fp64, even D, 8 walkers per wavefront
            const char* v = std::getenv("MCMCPP_HIP_MATRIX_CORE_MIN_WALKERS");
            const long mc_min = (v && *v) ? std::strtol(v, nullptr, 10) : 0;
            if (table->de_update_mc[0][lpw_log][epl_shift] && c.calc_id == MCMCPP_HIP_CALC_DENSE_GAUSSIAN && D % 2 == 0 && D <= 32 && mc_min >= 0 && n >= mc_min)
            {
                // 16 walkers per wavefront once the chip is full (as the stretch kernels: MCMCPP_HIP_MATRIX_CORE_4PASS_WALKERS)
                const char* v4 = std::getenv("MCMCPP_HIP_MATRIX_CORE_4PASS_WALKERS");
                const long four_pass = (v4 && *v4) ? std::strtol(v4, nullptr, 10) : 32768;
                const int big = n >= four_pass ? 1 : 0;
                update_fn = table->de_update_mc[big][lpw_log][epl_shift];
                walkers_per_block = (big ? 16 : 8) * kWavesPerBlock;
                matrix_core = true;
            }
        }
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

        {
            const char* v = std::getenv("MCMCPP_HIP_DE_SCAN_RUN");  // stream positions one scanning lane steps through
            scan_run = (v && *v) ? (int)std::strtol(v, nullptr, 10) : kDeScanRun;
            if (scan_run < 1) scan_run = 1;
            v = std::getenv("MCMCPP_HIP_DE_BATCH");  // half-steps planned together
            batch_max = (v && *v) ? (int)std::strtol(v, nullptr, 10) : kDeBatchMax;
#ifdef MCMCPP_DE_TIMING_DIAGNOSTICS
            // Experiment builds only (make VARIANT=detiming EXTRA=-DMCMCPP_DE_TIMING_DIAGNOSTICS): launches left out to time
            // the others -- THE CHAIN IS WRONG: 1 = the update launches alone, 2 = the planning launches alone, 3 / 4 / 5 = the
            // boundary without its records / resolve / scan.  The library that ships has no such switch.
            v = std::getenv("MCMCPP_HIP_DE_DEBUG");
            knob_debug = (v && *v) ? (int)std::strtol(v, nullptr, 10) : 0;
#endif
        }
        // the batch: as many half-steps as the position counters (32 bits), the resolver's lists and a sensible amount of
        // record memory (256 MiB) allow
        const unsigned per = (unsigned)D + 3u;
        {
            long long b = batch_max < 1 ? 1 : (batch_max > kDeBatchMax ? kDeBatchMax : batch_max);
            const long long by_positions = ((1LL << 30) - 2 * (kDeShiftMax + 1)) / ((long long)per * n);
            const long long by_lists = (kDeMaxBad * 5LL / 8 - 2 * (kDeShiftMax + 1) / n) / per;
            const long long by_records = (8LL << 20) / n;
            b = b < by_positions ? b : by_positions;
            b = b < by_lists ? b : by_lists;
            b = b < by_records ? b : by_records;
            if (b < 1)
                return fail(MCMCPP_HIP_E_UNSUPPORTED, "differential evolution: %d walkers x %d parameters exceed what the stream planner holds (per half-step: 2^30 stream "
                            "positions, 8 Mi updates)", W, D);
            batch_max = (int)b;
        }
        const size_t updates_max = (size_t)batch_max * n;
        positions_max = (long long)per * (long long)(updates_max - 1) + 2 * (kDeShiftMax + 1);  // (what a scan looks at)
        // one stream position in n is bad: a batch lists about (D + 3) of them per half-step (and as many again as the
        // kDeShiftMax positions behind its end hold, which matters for tiny ensembles), spread evenly over the lists
        {
            const long long expected = positions_max / n + 1;
            bad_capacity = (int)(4 * ((expected + kDeSegments - 1) / kDeSegments) + 64);
            const long long cap = 2 * expected + 512;
            resolve_capacity = cap > kDeMaxBad ? kDeMaxBad : (int)cap;
        }
        scan_blocks = (int)(((positions_max + scan_run - 1) / scan_run + kDePlanThreads - 1) / kDePlanThreads);

        HIP_TRY(hipMalloc(&d_pos, sizeof(T) * (size_t)W * D));
        HIP_TRY(hipMalloc(&d_logp, sizeof(T) * (size_t)W));
        HIP_TRY(hipMalloc(&d_nacc, sizeof(uint32_t) * (size_t)W));
        HIP_TRY(hipMalloc(&d_diag, sizeof(Diag)));
        HIP_TRY(hipMalloc(&d_counts, sizeof(uint32_t) * 2 * kDeSegments * kDeCountStride));  // two sets of lists (launch_boundary)
        HIP_TRY(hipMemset(d_counts, 0, sizeof(uint32_t) * 2 * kDeSegments * kDeCountStride));
        HIP_TRY(hipMalloc(&d_bad, sizeof(DeBad) * 2 * kDeSegments * (size_t)bad_capacity));
        HIP_TRY(hipMemset(d_bad, 0, sizeof(DeBad) * 2 * kDeSegments * (size_t)bad_capacity));
        HIP_TRY(hipMalloc(&d_recs, sizeof(DeRec<T>) * updates_max));  // the records of the batch being stepped through
        for (int k = 0; k < 2; ++k)
        {
            HIP_TRY(hipEventCreate(&ev_t0[k]));
            HIP_TRY(hipEventCreate(&ev_t1[k]));
        }
        graph_steps = c.graph_steps == 0 ? 128 : (c.graph_steps > 32768 ? 32768 : c.graph_steps);  // (the step inside a replay travels in 16 bits)
        // HIP cannot capture on the legacy default stream: a caller that hands it over gets plain launches
        if (!own_stream && (stream == nullptr || stream == hipStreamLegacy)) graph_steps = -1;
        replay_steps_max = graph_steps >= 1 ? graph_steps : 16;  // (plain launches: enqueued in groups of this many steps)
        const int per_block = walkers_per_block;
        update_blocks = (n + per_block - 1) / per_block;
        partial_waves = update_blocks * kWavesPerBlock;
        HIP_TRY(hipMalloc(&d_head, sizeof(DeHead)));
        HIP_TRY(hipMalloc(&d_batch, sizeof(DeBatch) * 2));  // batch b resolves into record b & 1
        // the run record and, right behind it, the wavefronts' accepted counts of a replay: one allocation (the update kernel
        // reaches both through one preloaded pointer)
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
            if (matrix_core)
            {
                // the matrix-core update kernels read P^T zero-padded to 32 x 32 straight into registers
                std::vector<T> pad((size_t)32 * 32, (T)0);
                for (int k = 0; k < D; ++k)
                    for (int i = 0; i < D; ++i) pad[(size_t)k * 32 + i] = prm[(size_t)k * D + i];
                HIP_TRY(hipMalloc(&d_params_padded, sizeof(T) * pad.size()));
                HIP_TRY(hipMemcpy(d_params_padded, pad.data(), sizeof(T) * pad.size(), hipMemcpyHostToDevice));
            }
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
            std::vector<Affine128> all;  // one allocation
            all.insert(all.end(), small.begin(), small.end());
            all.insert(all.end(), slo.begin(), slo.end());
            all.insert(all.end(), lo.begin(), lo.end());
            all.insert(all.end(), hi.begin(), hi.end());
            all.insert(all.end(), shi.begin(), shi.end());
            HIP_TRY(hipMalloc(&d_tables, sizeof(Affine128) * all.size()));
            HIP_TRY(hipMemcpy(d_tables, all.data(), sizeof(Affine128) * all.size(), hipMemcpyHostToDevice));
            d_jump_small = d_tables;
            d_scan_lo = d_jump_small + small.size();
            d_jump_lo = d_scan_lo + slo.size();
            d_jump_hi = d_jump_lo + lo.size();
            d_scan_hi = d_jump_hi + hi.size();
        }
        batch_jump = pcg_jump(inc, (unsigned __int128)per * (unsigned)n * (unsigned)batch_max);
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
        HIP_TRY(hipMemcpyAsync(d_pos, pos, sizeof(T) * (size_t)W * D, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(d_logp, logp, sizeof(T) * (size_t)W, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(d_nacc, 0, sizeof(uint32_t) * (size_t)W, stream));
        HIP_TRY(hipMemsetAsync(d_diag, 0, sizeof(Diag), stream));
        DeHead h;
        std::memset(&h, 0, sizeof h);
        h.state = state0;
        h.provisional[0] = state0;  // (batches 0 and 1 are scanned before anything has been resolved)
        h.provisional[1] = apply(batch_jump, state0);
        HIP_TRY(hipMemcpyAsync(d_head, &h, sizeof h, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemsetAsync(d_counts, 0, sizeof(uint32_t) * 2 * kDeSegments * kDeCountStride, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        steps_since_reset = 0;
        half_steps = 0;
        primed = false;
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

    // The boundary launch in front of a batch: the resolve of batch `resolve_batch` (scanned before), the records of batch
    // `record_batch` (resolved before) and the scan of batch `scan_batch`; any of them < 0: not in this launch (priming).
    void launch_boundary(long long resolve_batch, long long record_batch, long long scan_batch)
    {
        const unsigned per = (unsigned)D + 3u;
        DePlanArgs p;
        std::memset(&p, 0, sizeof p);
        p.head = d_head;
        p.scan_hi = d_scan_hi;
        p.scan_lo = d_scan_lo;
        p.jump_hi = d_jump_hi;
        p.jump_lo = d_jump_lo;
        p.jump_small = d_jump_small;
        p.batch_jump = batch_jump;
        p.inc = inc;
        p.threshold = threshold;
        p.n = n;
        p.dims = D;
        p.updates = n * batch_max;
        p.positions = (int)((long long)per * (long long)(p.updates - 1) + kDeShiftMax + 1);
        p.scan_positions = p.positions + kDeShiftMax + 1;
        p.scan_run = scan_run;
        p.seg_len = (p.scan_positions + kDeSegments - 1) / kDeSegments;
        p.bad_capacity = bad_capacity;
        // two sets of lists: batch b is scanned into set b & 1 (its resolve runs beside the scan of batch b + 1)
        const long long sb = scan_batch >= 0 ? scan_batch : resolve_batch + 1;
        p.scan_parity = (int)(sb & 1);
        p.bad = d_bad + (size_t)(sb & 1) * kDeSegments * (size_t)bad_capacity;
        p.counts = d_counts + (size_t)(sb & 1) * kDeSegments * kDeCountStride;
        p.resolve_bad = d_bad + (size_t)((sb + 1) & 1) * kDeSegments * (size_t)bad_capacity;
        p.resolve_counts = d_counts + (size_t)((sb + 1) & 1) * kDeSegments * kDeCountStride;
        p.batch = d_batch + (resolve_batch >= 0 ? (resolve_batch & 1) : 0);
        const int resolve = resolve_batch >= 0 ? 1 : 0;
        const int record_blocks = record_batch >= 0 ? (p.updates + kDePlanThreads - 1) / kDePlanThreads : 0;
        const int scan_now = scan_batch >= 0 ? scan_blocks : 0;
        size_t lds = record_blocks ? sizeof(DePlan) * kDeMaxEvents : 0;
        const size_t need = resolve ? de_resolve_lds_bytes(resolve_capacity, D + 3) : 0;
        lds = need > lds ? need : lds;
        hipLaunchKernelGGL((de_boundary_kernel<T>), dim3((unsigned)(resolve + record_blocks + scan_now)), dim3(kDePlanThreads), lds, stream, p, resolve, resolve_capacity,
                           record_blocks, d_batch + (record_batch >= 0 ? (record_batch & 1) : 0), d_recs);
    }

    // Behind a set_state: batch 0 scanned and resolved, batch 1 scanned -- what the boundary in front of a batch finds
    // (see enqueue_replay).
    int prime()
    {
        launch_boundary(-1, -1, 0);
        launch_boundary(0, -1, 1);
        HIP_TRY(hipGetLastError());
        primed = true;
        return MCMCPP_HIP_OK;
    }

    // `steps` ensemble steps (at most replay_steps_max) from half-step h0 (counted from the set_state) on the launch stream:
    // one update launch per half-step; in front of the first half-step of batch b the boundary launch -- the resolve of
    // batch b + 1 (scanned at the boundary before), the records of batch b (resolved at the boundary before) and the scan
    // of batch b + 2.  Behind the steps: the accepted counts and the run record.
    int enqueue_replay(int steps, uint64_t h0)
    {
        typename LaunchTable<T>::DeLaunch l;
        l.pos = d_pos;
        l.logp = d_logp;
        l.n_accept = d_nacc;
        l.jump_small = d_jump_small;
        l.run = d_run;
        l.n = n;
        l.dims = D;
        l.vec_ok = vec_ok;
        l.matrix_padded = d_params_padded;
        for (int i = 0; i < 2 * steps; ++i)
        {
            const uint64_t h = h0 + (uint64_t)i;
            const uint64_t b = h / (uint64_t)batch_max;
            const int j = (int)(h % (uint64_t)batch_max);
            if (j == 0 && knob_debug != 1)
            {
                // (timing diagnostics 3 / 4 / 5: the boundary without its records / without its resolve / without its scan)
                launch_boundary(knob_debug == 4 ? -1 : (long long)b + 1, knob_debug == 3 ? -1 : (long long)b, knob_debug == 5 ? -1 : (long long)b + 2);
            }
            if (knob_debug >= 2) continue;
            l.recs = d_recs + (size_t)j * (size_t)n;
            l.color = (int)(h & 1);
            l.step = i >> 1;
            update_fn(l, args, (unsigned)update_blocks, stream);
        }
        hipLaunchKernelGGL(de_accepted_kernel, dim3((unsigned)steps), dim3(256), 0, stream, d_run, partial_waves);
        hipLaunchKernelGGL(de_advance_kernel, dim3(1), dim3(1), 0, stream, d_run, steps);
        return MCMCPP_HIP_OK;
    }

    // hipGraph of `steps` ensemble steps that start at half-step h0: where the batch boundaries fall, and which of the two
    // batch records a boundary writes, depends on h0 mod two batches
    int graph_for(int steps, uint64_t h0, hipGraphExec_t* out)
    {
        const uint64_t phase = h0 % (2 * (uint64_t)batch_max);
        const uint64_t key = (uint64_t)steps * (2 * (uint64_t)batch_max) + phase;
        auto it = graph_cache.find(key);
        if (it == graph_cache.end())
        {
            if (graph_cache.size() >= 64)
            {
                // (runs of ever-changing lengths: the graphs still queued have been launched, not destroyed under them)
                HIP_TRY(hipStreamSynchronize(stream));
                for (auto& kv : graph_cache)
                    if (kv.second) (void)hipGraphExecDestroy(kv.second);
                graph_cache.clear();
            }
            hipGraph_t g = nullptr;
            HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
            int rc = enqueue_replay(steps, phase);
            if (rc) return rc;
            HIP_TRY(hipStreamEndCapture(stream, &g));
            hipGraphExec_t ex = nullptr;
            HIP_TRY(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
            HIP_TRY(hipGraphDestroy(g));
            it = graph_cache.emplace(key, ex).first;
        }
        *out = it->second;
        return MCMCPP_HIP_OK;
    }

    int enqueue_steps(int64_t steps)
    {
        if (!primed)
        {
            int rc = prime();
            if (rc) return rc;
        }
        int64_t left = steps;
        while (left > 0)
        {
            const int now = (int)(left < replay_steps_max ? left : replay_steps_max);
            if (graph_steps >= 1)
            {
                hipGraphExec_t ex = nullptr;
                int rc = graph_for(now, half_steps, &ex);
                if (rc) return rc;
                HIP_TRY(hipGraphLaunch(ex, stream));
            }
            else
            {
                int rc = enqueue_replay(now, half_steps);
                if (rc) return rc;
            }
            half_steps += 2 * (uint64_t)now;
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
            // every draw thrown away so far (bounded_rand, ind2 == ind1): the stream is planned ahead of the updates, so
            // count from the record of the batch the next half-step belongs to
            *redraws = 0;
            if (primed)
            {
                std::vector<DeBatch> rec(1);
                const uint64_t b = half_steps / (uint64_t)batch_max;
                HIP_TRY(hipMemcpy(rec.data(), d_batch + (b & 1), sizeof(DeBatch), hipMemcpyDeviceToHost));
                const uint64_t done = (half_steps % (uint64_t)batch_max) * (uint64_t)n;  // updates of this batch behind us
                uint64_t shift = 0;
                for (uint32_t e = 0; e < rec[0].events && e < (uint32_t)kDeMaxEvents && (uint64_t)rec[0].plan[e].m < done; ++e) shift = rec[0].plan[e].shift_after;
                *redraws = rec[0].extra_base + shift;
            }
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
        for (auto& kv : graph_cache)
            if (kv.second) (void)hipGraphExecDestroy(kv.second);
        void* bufs[] = {d_pos, d_logp, d_nacc, d_diag, d_head, d_batch, d_run, d_counts, d_params, d_params_padded, d_tables, d_chain, d_acc, d_recs, d_bad};
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
    int W = 0, D = 0, n = 0, lpw = 1, epl = 1, vec_ok = 0, device = -1, walkers_per_block = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false, have_state = false;
    T *d_pos = nullptr, *d_logp = nullptr, *d_params = nullptr, *d_params_padded = nullptr, *d_chain = nullptr;
    bool matrix_core = false;
    uint32_t *d_nacc = nullptr, *d_acc = nullptr;
    Diag* d_diag = nullptr;
    DeHead* d_head = nullptr;
    DeBatch* d_batch = nullptr;
    Affine128* d_tables = nullptr;  // jump_small, scan_lo, jump_lo, jump_hi, scan_hi
    uint32_t* d_counts = nullptr;
    DeBad* d_bad = nullptr;
    Affine128 *d_scan_lo = nullptr, *d_scan_hi = nullptr;
    int bad_capacity = 0, scan_run = kDeScanRun, batch_max = kDeBatchMax;
    long long positions_max = 0;
    DeRec<T>* d_recs = nullptr;
    DeRunInfo* d_run = nullptr;
    DeArgs<T> args;
    int update_blocks = 0, partial_waves = 0, graph_steps = 128, replay_steps_max = 128, knob_debug = 0;
    int resolve_capacity = 0, scan_blocks = 0;
    bool primed = false;    // batches 0 and 1 planned behind the last set_state
    Affine128 batch_jump;   // (D+3) * n * batch_max draws
    bool run_touched = false;
    hipEvent_t ev_t0[2] = {nullptr, nullptr}, ev_t1[2] = {nullptr, nullptr};
    std::unordered_map<uint64_t, hipGraphExec_t> graph_cache;

    Affine128 *d_jump_lo = nullptr, *d_jump_hi = nullptr, *d_jump_small = nullptr;
    size_t chain_bytes = 0, acc_count = 0;
    U128 state0, inc;
    uint64_t threshold = 0, steps_since_reset = 0, half_steps = 0;
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

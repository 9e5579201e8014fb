// sampler_base.hpp -- what the C ABI handle is behind include/mcmcpp_hip.h: one abstract interface, implemented by the
// stretch-move sampler (mcmcpp_hip.hip) and by the differential-evolution sampler (diffevo.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <string>
#include <thread>

#include "../../include/mcmcpp_hip.h"

struct mcmcpp_hip_sampler
{
    std::string error;
    // host-side cost of the last run (mcmcpp_hip_last_run_host_timing)
    double host_enqueue_ms = 0.0, host_wall_ms = 0.0, exchange_us_per_step = 0.0;
    // split ensembles (mcmcpp_hip_last_run_exchange): bytes this rank received per ensemble step of the last run, chunks of
    // steps that had to be repeated with larger exchange blocks, slots of a block at the end of the run
    double xchg_bytes_per_step = 0.0;
    int64_t xchg_rollbacks = 0, xchg_cap_slots = 0;
    // mcmcpp_hip_run_async: the run executes on a worker thread owned by the handle; stored steps are announced as they
    // reach the caller's memory
    std::thread async_worker;
    std::mutex async_mutex;
    std::condition_variable async_cv;
    int64_t async_stored = 0;   // stored steps of the current run that are complete in chain_out
    bool async_active = false;  // a worker has been started and not yet joined
    bool async_done = true;     // the worker's run has returned
    int async_rc = 0;
    void publish_stored(int64_t count)
    {
        {
            std::lock_guard<std::mutex> lock(async_mutex);
            if (count > async_stored) async_stored = count;
        }
        async_cv.notify_all();
    }
    virtual ~mcmcpp_hip_sampler()
    {
        if (async_worker.joinable()) async_worker.join();
    }
    virtual int set_state(const void* pos, const void* logp) = 0;
    virtual int run(int64_t n_saved, int32_t interval, void* chain_out, uint32_t* accepted_per_step) = 0;
    virtual int get_state(void* pos, void* logp, uint32_t* n_accept) = 0;
    virtual int reset_counters() = 0;
    virtual int seek(uint64_t steps_done) = 0;
    virtual int get_counters(uint64_t* accepted, uint64_t* steps, uint64_t* ties, uint64_t* redraws) = 0;
    virtual int calc_logp(const void* pos, int64_t count, void* out) = 0;
    virtual int last_run_timing(double* ms, int64_t* launches) = 0;
    virtual int half_step_async(int32_t color, int64_t save_slot) = 0;
    virtual int bind_device_chain(void* chain, int64_t slots) = 0;
    virtual void* device_positions() = 0;
    virtual int shard_span(int32_t color, int64_t* off, int64_t* cnt) = 0;
    virtual int synchronize() = 0;
    virtual int debug_stamps(unsigned long long* out8) = 0;

    // A call refused because the handle is busy with an asynchronous run: the message is a literal kept beside `error`,
    // which belongs to the worker thread while it runs (only the caller's thread touches `refused`).
    const char* refused = nullptr;
    int refuse(const char* literal)
    {
        refused = literal;
        return MCMCPP_HIP_E_STATE;
    }

    int fail(int code, const char* fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        error = buf;
        return code;
    }
};

#define HIP_TRY(expr)                                                                                       \
    do                                                                                                      \
    {                                                                                                       \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return fail(MCMCPP_HIP_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)


#include "pcg128.hpp"

namespace mcmcpp
{
inline int pow2_at_least(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}
inline int ilog2(int v)
{
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
inline Affine128 compose(const Affine128& g, const Affine128& f)  // g after f
{
    Affine128 r;
    r.mult = mul128(g.mult, f.mult);
    r.plus = add128(mul128(g.mult, f.plus), g.plus);
    return r;
}

// launch table (LaunchTable<double> / LaunchTable<float>) of a built-in or registered calculator, or nullptr
const void* launch_table_lookup(int dtype, int calc_id);
// Mover::DifferentialEvolution (diffevo.hip); *rc receives the init result, the handle carries the message
mcmcpp_hip_sampler* make_de_sampler(const mcmcpp_hip_config& cfg, int* rc);
}  // namespace mcmcpp

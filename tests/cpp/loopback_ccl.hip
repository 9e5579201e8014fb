// loopback_ccl.hip -- TEST INFRASTRUCTURE ONLY: a loop-back collective library.
//
// Exports the ten nccl* symbols libmcmcpp_hip.so binds at run time (mcmcpp_amd/csrc/rccl_dyn.hpp) and implements them for
// G ranks that live INSIDE ONE PROCESS ON ONE DEVICE: one host thread per rank, each with its own handle, replica and
// stream.  Selected with MCMCPP_HIP_RCCL_LIB=<this .so>; the product never loads it otherwise.  Its purpose: the box the
// tests run on has one GPU, and the world > 1 branch of the split ensemble (run_split / the exchanges' offsets and
// ordering, the facade's one-thread-per-rank constructor) must execute somewhere.  It validates the exchange LOGIC --
// which rows go where, in which order relative to the launches -- not RCCL's transport.
//
// Semantics kept from NCCL: every rank calls the same collectives in the same order; a collective is enqueued on the
// caller's stream and returns at once; rank r's output is complete on r's stream behind the call; rank r's input may be
// overwritten by whatever r enqueues behind the call (the call orders it behind every peer's reads); ncclCommInitRank
// blocks until all ranks of the id have joined; calls between ncclGroupStart / ncclGroupEnd are issued together at the
// outermost ncclGroupEnd.  Mismatched calls (kind, count, type) fail with ncclInvalidUsage on every rank instead of
// hanging, and a rank that never arrives fails the others after a timeout (LOOPBACK_CCL_TIMEOUT_S, default 120).
//
// How one (group of) collective(s) runs, per rank thread:
//   publish the calls; BARRIER; rank 0 compares them and sizes the staging area; BARRIER;
//   all-reduce inputs -> staging[rank] on the own stream; record ready[rank] on the own stream; BARRIER;
//   wait for every peer's ready event on the own stream; all-gather: copy each peer's input slice into the own output;
//   all-reduce: one small kernel combines the G staged inputs into the own output; record done[rank]; BARRIER;
//   wait for every peer's done event on the own stream (a rank's input and the staging stay intact until all have read);
//   BARRIER (nobody re-records an event or republishes before every wait above has been enqueued).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace
{
struct Call
{
    int kind;  // 0 all-gather, 1 all-reduce
    const void* send;
    void* recv;
    size_t count;
    ncclDataType_t type;
    ncclRedOp_t op;
};

size_t type_bytes(ncclDataType_t t)
{
    switch (t)
    {
    case ncclInt8:
    case ncclUint8: return 1;
    case ncclFloat16:
    case ncclBfloat16: return 2;
    case ncclInt32:
    case ncclUint32:
    case ncclFloat32: return 4;
    case ncclInt64:
    case ncclUint64:
    case ncclFloat64: return 8;
    default: return 0;
    }
}

struct Group
{
    int nranks = 0;
    int device = -1;
    std::mutex m;
    std::condition_variable cv;
    int joined = 0, destroyed = 0;
    // barrier
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    // what the ranks published for the collective in flight
    std::vector<std::vector<Call>> calls;
    ncclResult_t verdict = ncclSuccess;
    std::vector<hipEvent_t> ready, done;
    char* staging = nullptr;  // [nranks][staging_per_rank]
    size_t staging_per_rank = 0;
    uint64_t collectives = 0, bytes_moved = 0;
};

struct Comm
{
    Group* g;
    int rank;
};

std::mutex g_registry_mutex;
std::map<std::string, Group*> g_groups;  // by unique id
uint64_t g_next_id = 1;

double timeout_seconds()
{
    const char* v = std::getenv("LOOPBACK_CCL_TIMEOUT_S");
    return (v && *v) ? std::atof(v) : 120.0;
}

// all ranks of the group meet here; false when the group is broken (a rank timed out or failed)
bool barrier(Group* g)
{
    std::unique_lock<std::mutex> lock(g->m);
    if (g->broken) return false;
    const uint64_t gen = g->generation;
    if (++g->arrived == g->nranks)
    {
        g->arrived = 0;
        ++g->generation;
        g->cv.notify_all();
        return true;
    }
    const bool ok = g->cv.wait_for(lock, std::chrono::duration<double>(timeout_seconds()), [&]() { return g->generation != gen || g->broken; });
    if (!ok || g->broken)
    {
        g->broken = true;
        g->cv.notify_all();
        return false;
    }
    return true;
}

template <class T>
__global__ void combine_kernel(T* out, const T* staged, size_t per_rank_elems, size_t count, int nranks, int op)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    T acc = staged[i];
    for (int r = 1; r < nranks; ++r)
    {
        const T v = staged[(size_t)r * per_rank_elems + i];
        acc = op == 0 ? (T)(acc + v) : (op == 1 ? (v > acc ? v : acc) : (v < acc ? v : acc));
    }
    out[i] = acc;
}

template <class T>
void launch_combine(void* out, const char* staged, size_t per_rank_bytes, size_t count, int nranks, int op, hipStream_t stream)
{
    const unsigned grid = (unsigned)((count + 255) / 256);
    hipLaunchKernelGGL(combine_kernel<T>, dim3(grid), dim3(256), 0, stream, (T*)out, (const T*)staged, per_rank_bytes / sizeof(T), count, nranks, op);
}

thread_local int t_group_depth = 0;
struct Pending
{
    Comm* comm;
    hipStream_t stream;
    Call call;
};
thread_local std::vector<Pending> t_pending;

#define HIP_OK(expr)                          \
    do                                        \
    {                                         \
        if ((expr) != hipSuccess)             \
        {                                     \
            std::lock_guard<std::mutex> l(g->m); \
            g->broken = true;                 \
            g->cv.notify_all();               \
            return ncclUnhandledCudaError;    \
        }                                     \
    } while (0)

ncclResult_t issue(Comm* comm, hipStream_t stream, const std::vector<Call>& mine)
{
    Group* g = comm->g;
    const int rank = comm->rank, G = g->nranks;
    g->calls[(size_t)rank] = mine;
    if (!barrier(g)) return ncclSystemError;
    size_t need = 0;  // staging bytes per rank: the all-reduce inputs of this group, 256-byte aligned each
    for (const Call& c : mine)
        if (c.kind == 1) need += (c.count * type_bytes(c.type) + 255) & ~(size_t)255;
    if (rank == 0)
    {
        g->verdict = ncclSuccess;
        for (int r = 1; r < G; ++r)
        {
            const std::vector<Call>& other = g->calls[(size_t)r];
            if (other.size() != mine.size()) g->verdict = ncclInvalidUsage;
            for (size_t k = 0; g->verdict == ncclSuccess && k < mine.size(); ++k)
                if (other[k].kind != mine[k].kind || other[k].count != mine[k].count || other[k].type != mine[k].type || other[k].op != mine[k].op)
                    g->verdict = ncclInvalidUsage;
        }
        for (const Call& c : mine)
            if (type_bytes(c.type) == 0) g->verdict = ncclInvalidArgument;
        if (g->verdict == ncclSuccess && need > g->staging_per_rank)
        {
            // (every rank's stream may still read the old area: wait for the device before replacing it)
            if (hipDeviceSynchronize() != hipSuccess) g->verdict = ncclUnhandledCudaError;
            if (g->staging) (void)hipFree(g->staging);
            g->staging = nullptr;
            g->staging_per_rank = 0;
            void* p = nullptr;
            if (hipMalloc(&p, need * (size_t)G) != hipSuccess)
                g->verdict = ncclUnhandledCudaError;
            else
            {
                g->staging = (char*)p;
                g->staging_per_rank = need;
            }
        }
        ++g->collectives;
    }
    if (!barrier(g)) return ncclSystemError;
    if (g->verdict != ncclSuccess) return g->verdict;
    const size_t per_rank = g->staging_per_rank;

    size_t off = 0;
    for (const Call& c : mine)
        if (c.kind == 1)
        {
            const size_t bytes = c.count * type_bytes(c.type);
            HIP_OK(hipMemcpyAsync(g->staging + per_rank * (size_t)rank + off, c.send, bytes, hipMemcpyDeviceToDevice, stream));
            off += (bytes + 255) & ~(size_t)255;
        }
    HIP_OK(hipEventRecord(g->ready[(size_t)rank], stream));
    if (!barrier(g)) return ncclSystemError;

    for (int p = 0; p < G; ++p)
        if (p != rank) HIP_OK(hipStreamWaitEvent(stream, g->ready[(size_t)p], 0));
    off = 0;
    for (size_t k = 0; k < mine.size(); ++k)
    {
        const Call& c = mine[k];
        const size_t bytes = c.count * type_bytes(c.type);
        if (c.kind == 0)
        {
            for (int p = 0; p < G; ++p)
            {
                const void* src = g->calls[(size_t)p][k].send;
                char* dst = (char*)c.recv + bytes * (size_t)p;
                if (src == (const void*)dst) continue;  // (in place: the own slice is where it belongs)
                HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
                if (p != rank)
                {
                    std::lock_guard<std::mutex> l(g->m);
                    g->bytes_moved += bytes;
                }
            }
        }
        else
        {
            const int op = c.op == ncclSum ? 0 : (c.op == ncclMax ? 1 : (c.op == ncclMin ? 2 : -1));
            if (op < 0) return ncclInvalidArgument;
            const char* staged = g->staging + off;
            switch (c.type)
            {
            case ncclUint32: launch_combine<uint32_t>(c.recv, staged, per_rank, c.count, G, op, stream); break;
            case ncclInt32: launch_combine<int32_t>(c.recv, staged, per_rank, c.count, G, op, stream); break;
            case ncclUint64: launch_combine<uint64_t>(c.recv, staged, per_rank, c.count, G, op, stream); break;
            case ncclInt64: launch_combine<int64_t>(c.recv, staged, per_rank, c.count, G, op, stream); break;
            case ncclFloat32: launch_combine<float>(c.recv, staged, per_rank, c.count, G, op, stream); break;
            case ncclFloat64: launch_combine<double>(c.recv, staged, per_rank, c.count, G, op, stream); break;
            default: return ncclInvalidArgument;
            }
            HIP_OK(hipGetLastError());
            off += (bytes + 255) & ~(size_t)255;
        }
    }
    HIP_OK(hipEventRecord(g->done[(size_t)rank], stream));
    if (!barrier(g)) return ncclSystemError;
    for (int p = 0; p < G; ++p)
        if (p != rank) HIP_OK(hipStreamWaitEvent(stream, g->done[(size_t)p], 0));
    if (!barrier(g)) return ncclSystemError;
    return ncclSuccess;
}

ncclResult_t submit(Comm* comm, hipStream_t stream, const Call& call)
{
    if (!comm || !comm->g) return ncclInvalidArgument;
    if (t_group_depth > 0)
    {
        t_pending.push_back(Pending{comm, stream, call});
        return ncclSuccess;
    }
    return issue(comm, stream, std::vector<Call>(1, call));
}
}  // namespace

extern "C"
{
ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    std::memset(id, 0, sizeof *id);
    const uint64_t v = g_next_id++;
    std::memcpy(id->internal, "LOOPBACK", 8);
    std::memcpy(id->internal + 8, &v, sizeof v);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    int device = -1;
    if (hipGetDevice(&device) != hipSuccess) return ncclUnhandledCudaError;
    const std::string key(id.internal, sizeof id.internal);
    Group* g = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        Group*& slot = g_groups[key];
        if (!slot)
        {
            slot = new Group();
            slot->nranks = nranks;
            slot->device = device;
            slot->calls.resize((size_t)nranks);
            slot->ready.assign((size_t)nranks, nullptr);
            slot->done.assign((size_t)nranks, nullptr);
        }
        g = slot;
    }
    if (g->nranks != nranks || g->device != device) return ncclInvalidUsage;  // (all ranks of a loop-back group share one device)
    if (hipEventCreateWithFlags(&g->ready[(size_t)rank], hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventCreateWithFlags(&g->done[(size_t)rank], hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    {
        // as ncclCommInitRank: return when every rank has joined
        std::unique_lock<std::mutex> lock(g->m);
        ++g->joined;
        g->cv.notify_all();
        const bool ok = g->cv.wait_for(lock, std::chrono::duration<double>(timeout_seconds()), [&]() { return g->joined >= g->nranks || g->broken; });
        if (!ok || g->broken)
        {
            g->broken = true;
            g->cv.notify_all();
            return ncclSystemError;
        }
    }
    Comm* c = new Comm{g, rank};
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclInvalidArgument;
    Group* g = c->g;
    bool last = false;
    {
        std::lock_guard<std::mutex> lock(g->m);
        last = ++g->destroyed == g->nranks;
    }
    if (last)
    {
        (void)hipDeviceSynchronize();
        for (hipEvent_t e : g->ready)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : g->done)
            if (e) (void)hipEventDestroy(e);
        if (g->staging) (void)hipFree(g->staging);
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        for (std::map<std::string, Group*>::iterator it = g_groups.begin(); it != g_groups.end(); ++it)
            if (it->second == g)
            {
                g_groups.erase(it);
                break;
            }
        delete g;
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count)
{
    const Comm* c = reinterpret_cast<const Comm*>(comm);
    if (!c || !count) return ncclInvalidArgument;
    *count = c->g->nranks;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank)
{
    const Comm* c = reinterpret_cast<const Comm*>(comm);
    if (!c || !rank) return ncclInvalidArgument;
    *rank = c->rank;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
    return submit(reinterpret_cast<Comm*>(comm), stream, Call{0, sendbuff, recvbuff, sendcount, datatype, ncclSum});
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
    return submit(reinterpret_cast<Comm*>(comm), stream, Call{1, sendbuff, recvbuff, count, datatype, op});
}

ncclResult_t ncclGroupStart()
{
    ++t_group_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (t_group_depth <= 0) return ncclInvalidUsage;
    if (--t_group_depth > 0) return ncclSuccess;
    std::vector<Pending> pending;
    pending.swap(t_pending);
    if (pending.empty()) return ncclSuccess;
    // (the library under test issues a group on one communicator and one stream)
    std::vector<Call> calls;
    for (const Pending& p : pending)
    {
        if (p.comm != pending[0].comm || p.stream != pending[0].stream) return ncclInvalidUsage;
        calls.push_back(p.call);
    }
    return issue(pending[0].comm, pending[0].stream, calls);
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r)
    {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "loop-back: a HIP call failed";
    case ncclSystemError: return "loop-back: a rank did not arrive (timeout) or the group is broken";
    case ncclInvalidArgument: return "loop-back: invalid argument";
    case ncclInvalidUsage: return "loop-back: the ranks issued different collectives";
    default: return "loop-back: error";
    }
}

// test hook (not an nccl symbol): collectives issued and bytes a rank received from its peers, summed over the group
void loopback_ccl_stats(ncclComm_t comm, uint64_t* collectives, uint64_t* bytes_moved)
{
    const Comm* c = reinterpret_cast<const Comm*>(comm);
    if (!c) return;
    std::lock_guard<std::mutex> lock(c->g->m);
    if (collectives) *collectives = c->g->collectives;
    if (bytes_moved) *bytes_moved = c->g->bytes_moved;
}
}

// rccl_dyn.hpp -- RCCL (the collective library over xGMI) bound at run time.
//
// A split ensemble (BASELINE config 5) exchanges the rows its ranks updated with ncclAllGather on the
// launch stream.  Single-GPU users never need a collective library, and librccl.so is half a gigabyte of code
// objects, so libmcmcpp_hip.so does not carry a link-time dependency on it: the first handle that asks for a
// communicator loads it (dlopen by soname: inside a process that already uses RCCL -- e.g. through
// torch.distributed -- that is the very same library instance, so communicators can be shared with it).
// MCMCPP_HIP_RCCL_LIB names another file.  Types come from <rccl/rccl.h>; no function of it is called directly.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <mutex>
#include <string>

namespace mcmcpp
{
struct Rccl
{
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string load_error;
    bool ok = false;

    // the process-wide binding; nullptr (with *why set) when the library or one of its symbols is missing
    static const Rccl* get(std::string* why)
    {
        static Rccl inst;
        static std::once_flag once;
        std::call_once(once, []() { inst.load(); });
        if (!inst.ok)
        {
            if (why) *why = inst.load_error;
            return nullptr;
        }
        return &inst;
    }

private:
    void load()
    {
        const char* forced = std::getenv("MCMCPP_HIP_RCCL_LIB");
        const char* names[] = {forced && *forced ? forced : "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void* lib = nullptr;
        for (const char* nm : names)
        {
            lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
            load_error = std::string("cannot load RCCL: ") + dlerror();
        }
        if (!lib) return;
        bool all = true;
        auto sym = [&](const char* nm) -> void* {
            void* p = dlsym(lib, nm);
            if (!p)
            {
                all = false;
                load_error = std::string("RCCL lacks ") + nm;
            }
            return p;
        };
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        CommCount = reinterpret_cast<decltype(CommCount)>(sym("ncclCommCount"));
        CommUserRank = reinterpret_cast<decltype(CommUserRank)>(sym("ncclCommUserRank"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        ok = all;
    }
};

template <class T>
struct RcclType;
template <>
struct RcclType<double>
{
    static constexpr ncclDataType_t value = ncclFloat64;
};
template <>
struct RcclType<float>
{
    static constexpr ncclDataType_t value = ncclFloat32;
};
}  // namespace mcmcpp

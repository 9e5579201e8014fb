/*
 * Device/HipBackend.h -- the one place where the header-only facade touches the C ABI of
 * libmcmcpp_hip.so (include/mcmcpp_hip.h).  RAII over a mcmcpp_hip_sampler handle; any failure is fatal
 * and loud, in the spirit of the reference's asserts (/root/reference/MCMCpp/EnsembleSampler.h:207-208):
 * there is no CPU fallback to fall back to.
 */
#ifndef MCMCPP_DEVICE_HIPBACKEND_H
#define MCMCPP_DEVICE_HIPBACKEND_H

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "../../mcmcpp_hip.h"

namespace MCMC
{
namespace Device
{
/// Element types the kernels are built for (the reference also admits long double; gfx950 has no such type).
template <class ParamType>
struct HipDtype
{
    static_assert(sizeof(ParamType) == 0, "the MI355X samplers run with ParamType = double or float");
};
template <>
struct HipDtype<double>
{
    static const int value = MCMCPP_HIP_F64;
};
template <>
struct HipDtype<float>
{
    static const int value = MCMCPP_HIP_F32;
};

/// Which GPUs a sampler runs on.  One device (the default: the process's current device) steps the whole ensemble; several
/// split ONE ensemble between them -- every device keeps the full replica of the positions, updates its slice of both
/// halves and exchanges the updated rows over RCCL once per ensemble step (the role of ParallelEnsembleSampler's
/// threadCount workers and barrier controller, /root/reference/MCMCpp/ParallelEnsembleSampler.h:119-120,285-291).
/// Results do not depend on the placement.  The reference's constructors have no such argument, so unchanged user code
/// selects a placement through the environment: MCMCPP_DEVICES="0,1,2,3" (HIP device ordinals).
struct Placement
{
    std::vector<int> devices;  ///< HIP device ordinals; empty = the current device
    bool splitEnsemble;        ///< true: go through the communicator even with one device (tests of the split path)
    Placement() : splitEnsemble(false) {}
    explicit Placement(int device) : devices(1, device), splitEnsemble(false) {}
    explicit Placement(const std::vector<int>& list) : devices(list), splitEnsemble(false) {}
    static Placement fromEnvironment()
    {
        Placement p;
        const char* v = std::getenv("MCMCPP_DEVICES");
        while (v && *v)
        {
            char* end = nullptr;
            const long d = std::strtol(v, &end, 10);
            if (end == v) break;
            p.devices.push_back(static_cast<int>(d));
            v = (*end == ',') ? end + 1 : end;
        }
        return p;
    }
    int count() const { return devices.empty() ? 1 : static_cast<int>(devices.size()); }
    int device(int k) const { return devices.empty() ? -1 : devices[static_cast<std::size_t>(k)]; }
};

class HipHandle
{
public:
    HipHandle() : handle(nullptr) {}
    ~HipHandle()
    {
        if (handle) mcmcpp_hip_destroy(handle);
    }
    HipHandle(const HipHandle&) = delete;
    HipHandle& operator=(const HipHandle&) = delete;
    HipHandle(HipHandle&& other) noexcept : handle(other.handle) { other.handle = nullptr; }

    void create(const mcmcpp_hip_config& cfg)
    {
        checkCreate("mcmcpp_hip_create", mcmcpp_hip_create(&cfg, &handle));
    }
    /// The same for calls that have no handle yet (the library keeps their message per thread).
    static void checkCreate(const char* what, int rc)
    {
        if (rc != MCMCPP_HIP_OK) die(what, rc, mcmcpp_hip_last_error(nullptr));
    }
    /// Abort with the library's message unless rc is MCMCPP_HIP_OK.
    void check(const char* what, int rc) const
    {
        if (rc != MCMCPP_HIP_OK) die(what, rc, mcmcpp_hip_last_error(handle));
    }
    mcmcpp_hip_sampler* get() const { return handle; }

private:
    static void die(const char* what, int rc, const char* msg)
    {
        std::fprintf(stderr, "MCMCpp (MI355X): %s failed with code %d: %s\n", what, rc, msg ? msg : "");
        std::abort();
    }
    mcmcpp_hip_sampler* handle;
};

}  // namespace Device
}  // namespace MCMC
#endif  // MCMCPP_DEVICE_HIPBACKEND_H

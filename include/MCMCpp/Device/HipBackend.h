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

    void create(const mcmcpp_hip_config& cfg)
    {
        const int rc = mcmcpp_hip_create(&cfg, &handle);
        if (rc != MCMCPP_HIP_OK) die("mcmcpp_hip_create", rc, mcmcpp_hip_last_error(nullptr));
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

// how long does a pinned Chain block take to allocate?  (mcmcpp_hip_host_alloc = hipHostMalloc)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include "mcmcpp_hip.h"
int main()
{
    for (unsigned long long mib : {4ULL, 64ULL, 256ULL, 256ULL, 1024ULL})
    {
        auto t0 = std::chrono::steady_clock::now();
        void* p = mcmcpp_hip_host_alloc(mib << 20);
        auto t1 = std::chrono::steady_clock::now();
        void* q = nullptr;
        if (posix_memalign(&q, 64, mib << 20)) return 1;
        std::memset(q, 0, mib << 20);
        auto t2 = std::chrono::steady_clock::now();
        std::printf("%5llu MiB: pinned alloc %.2f ms (%.1f GB/s); malloc + first touch %.2f ms\n", mib, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                    (double)(mib << 20) / std::chrono::duration<double>(t1 - t0).count() / 1e9, std::chrono::duration<double, std::milli>(t2 - t1).count());
        mcmcpp_hip_host_free(p);
        std::free(q);
    }
    return 0;
}

// Instantiations of the two-group sample kernel (sample_kernel_x2.hip.h): eight float streams per workgroup, PARITY arithmetic,
// one per register-resident items-per-lane variant.  A translation unit of its own so that it builds beside the others.
#include "sample_kernel_x2.hip.h"
#include <mutex>

template <int NW>
static int launch_x2(int grid, int lds, hipStream_t st, const LpcnSampleArgs *d_args)
{
    auto k = lpcn::sample_kernel_x2<NW>;
    static std::mutex mu;
    static int limit[64];                                    // per HIP device: the dynamic-LDS size already granted
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> g(mu);
        if (dev < 0 || dev >= 64 || limit[dev] < lds) {
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return (int)e;
            if (dev >= 0 && dev < 64) limit[dev] = lds;
        }
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(LPCN_WG_THREADS), lds, st, d_args);
    return (int)hipGetLastError();
}

// returns a hipError_t value (0 = launched); hipErrorInvalidValue: no such variant (the caller falls back to four streams per workgroup)
extern "C" int lpcn_launch_sample_x2(int nw, int grid, int lds, hipStream_t st, const LpcnSampleArgs *d_args)
{
    switch (nw) {
#ifdef LPCN_ONLY_BENCH_VARIANT
    case 30: return launch_x2<30>(grid, lds, st, d_args);
#else
    case 24: return launch_x2<24>(grid, lds, st, d_args);
    case 28: return launch_x2<28>(grid, lds, st, d_args);
    case 30: return launch_x2<30>(grid, lds, st, d_args);
    case 32: return launch_x2<32>(grid, lds, st, d_args);
#endif
    default: return (int)hipErrorInvalidValue;
    }
}
extern "C" int lpcn_x2_lds_bytes(int nb_b) { return lpcn::LdsX2::total(nb_b); }

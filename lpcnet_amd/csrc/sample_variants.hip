// Instantiations of the persistent sample kernel for ONE streams-per-workgroup value (compile with -DLPCN_S=1|2|4):
// items per lane (register-resident GRU-A variants) x blob flavour (fp32 / int8) x arithmetic (PARITY / FAST).
// Split from engine.hip so that the three values build in parallel.
#include "sample_kernel.hip.h"
#include <mutex>

#ifndef LPCN_S
#error "compile with -DLPCN_S=1, 2 or 4"
#endif
#define LPCN_CAT2(a, b) a##b
#define LPCN_CAT(a, b) LPCN_CAT2(a, b)

template <int NW, bool I8, bool FAST, bool PACK2 = false>
static int launch(int grid, int lds, hipStream_t st, const LpcnSampleArgs *d_args)
{
    auto k = lpcn::sample_kernel<LPCN_S, NW, I8, FAST, PACK2>;
    // the dynamic-LDS limit of a variant is raised once per (device, size), not at every launch
    static std::mutex mu;
    static int limit[64];                                    // per HIP device: the size already granted
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
    // (the arguments stay a device-resident block read through scalar loads: passing the struct by value was measured --
    // 23 more spilled SGPRs, 105.6 vs 107.3 M samples/s on the float kernel, +1.7 % on the int8 one)
    hipLaunchKernelGGL(k, dim3(grid), dim3(LPCN_WG_THREADS), lds, st, d_args);
    return (int)hipGetLastError();
}

template <bool FAST>
static int pick(int nw, int is_int8, int pack2, int grid, int lds, hipStream_t st, const LpcnSampleArgs *d_args)
{
#ifdef LPCN_ONLY_BENCH_VARIANT       // tools: compile only the benchmark model's PARITY float kernel (quick assembly listings)
    if constexpr (FAST) return (int)hipErrorInvalidValue;
#if LPCN_ONLY_BENCH_VARIANT == 2     // ... the int8 one (32 items per lane, two workgroups per CU; build with -DLPCN_S=2)
    else return (is_int8 && nw == 32 && pack2) ? launch<32, true, false, (LPCN_S <= 2)>(grid, lds, st, d_args) : (int)hipErrorInvalidValue;
#else
    else return (!is_int8 && nw == 30) ? launch<30, false, false>(grid, lds, st, d_args) : (int)hipErrorInvalidValue;
#endif
#else
    if (is_int8) {
        switch (nw) {
        case 32: return (pack2 && LPCN_S <= 2) ? launch<32, true, FAST, (LPCN_S <= 2)>(grid, lds, st, d_args) : launch<32, true, FAST>(grid, lds, st, d_args);
        case 48: return launch<48, true, FAST>(grid, lds, st, d_args);
        case 64: return launch<64, true, FAST>(grid, lds, st, d_args);
        default: return launch<96, true, FAST>(grid, lds, st, d_args);      // (a row group may list every one of its 96 input blocks: trained, heavy-tailed sparsity)
        }
    }
    switch (nw) {
    case 24: return launch<24, false, FAST>(grid, lds, st, d_args);
    case 28: return launch<28, false, FAST>(grid, lds, st, d_args);
    case 30: return launch<30, false, FAST>(grid, lds, st, d_args);
    case 32: return launch<32, false, FAST>(grid, lds, st, d_args);
    case 36: return launch<36, false, FAST>(grid, lds, st, d_args);
    case 40: return launch<40, false, FAST>(grid, lds, st, d_args);
    case 48: return launch<48, false, FAST>(grid, lds, st, d_args);      // (more than 32 items per lane: the items past the 28th are streamed from L2, sample_kernel.hip.h)
    case 64: return launch<64, false, FAST>(grid, lds, st, d_args);
    case 80: return launch<80, false, FAST>(grid, lds, st, d_args);
    default: return launch<96, false, FAST>(grid, lds, st, d_args);      // (a full row: every one of its 96 input blocks)
    }
#endif
}

// returns a hipError_t value (0 = launched)
// flags: bit 0 = FAST arithmetic, bit 1 = PACK2 (two workgroups per CU; int8, 32 items per lane only)
extern "C" int LPCN_CAT(lpcn_launch_sample_s, LPCN_S)(int nw, int is_int8, int flags, int grid, int lds, hipStream_t st, const LpcnSampleArgs *d_args)
{
    const int pack2 = (flags >> 1) & 1;
    return (flags & 1) ? pick<true>(nw, is_int8, pack2, grid, lds, st, d_args) : pick<false>(nw, is_int8, pack2, grid, lds, st, d_args);
}

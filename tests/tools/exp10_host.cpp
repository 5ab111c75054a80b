// TEST INFRASTRUCTURE: host build of the engine's 10^x (lpcnet_amd/csrc/lpcnet_exp10.h) next to glibc's pow, the function
// the reference calls (src/freq.c:317-318: `pow(10.f, tmp[i])*compensation[i]` stored to float).
//   g++ -O2 -ffp-contract=off -shared -fPIC -I lpcnet_amd/csrc tests/tools/exp10_host.cpp -o <tmp>/libexp10_host.so
#include <cmath>
#include <cstdint>
#include <cstring>
#include "lpcnet_exp10.h"

static const float band_comp[18] = {0.8f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.666667f, 0.5f, 0.5f, 0.5f,
                                    0.333333f, 0.25f, 0.25f, 0.2f, 0.166667f, 0.173913f};      // src/freq.c:51-53

extern "C" {
void exp10_engine(const float *x, double *out, long n) { for (long i = 0; i < n; ++i) out[i] = lpcn_exp10(x[i]); }
void exp10_glibc(const float *x, double *out, long n) { for (long i = 0; i < n; ++i) out[i] = pow((double)10.f, (double)x[i]); }

// float `e` with biased exponents [e_lo, e_hi), both signs, every `stride`-th mantissa: counts[0] = evaluations,
// counts[1] = doubles that differ from glibc, counts[2] = (value x band compensation) floats that differ (18 bands each)
void exp10_sweep(int e_lo, int e_hi, int stride, long *counts)
{
    long n = 0, bad_d = 0, bad_f = 0;
    for (int sign = 0; sign < 2; ++sign)
        for (int e = e_lo; e < e_hi; ++e)
            for (uint32_t m = 0; m < (1u << 23); m += (uint32_t)stride) {
                const uint32_t bits = ((uint32_t)sign << 31) | ((uint32_t)e << 23) | m;
                float x;
                memcpy(&x, &bits, 4);
                const double a = lpcn_exp10(x), b = pow((double)10.f, (double)x);
                ++n;
                if (a != b) {
                    ++bad_d;
                    for (int c = 0; c < 18; ++c) if ((float)(a * (double)band_comp[c]) != (float)(b * (double)band_comp[c])) ++bad_f;
                }
            }
    counts[0] = n; counts[1] = bad_d; counts[2] = bad_f;
}
}

// 10^x for a float x, correctly rounded to double for all practical purposes, without a math library.
//
// Why: lpc_from_cepstrum evaluates `pow(10.f, x)` in double and rounds the product with the band compensation to float
// (src/freq.c:317-318).  The reference gets that double from glibc (< 1 ULP, correctly rounded except when the exact
// value lies within ~2^-15 ULP of a rounding boundary); a device math library that is merely "<= 1 ULP" returns the
// neighbouring double for a sizeable fraction of arguments, and whenever that double pair straddles a float rounding
// boundary (probability ~2^-29 per call) one LPC coefficient changes in its last bit -- enough to send a free-running
// stream down a different trajectory for good.  This routine carries ~2^-70 relative error before the final rounding
// (double-double argument reduction, table and product), so it disagrees with a correctly rounded pow only when the
// exact value is within ~2^-17 ULP of a double rounding boundary AND that double decides a float rounding:
// ~2^-46 per call.  tests/test_exp10.py sweeps every float in the reachable range against glibc.
//
//   10^x = 2^n * 2^(j/128) * e^u,   x*log2(10) = n + j/128 + r,  |r| <= 2^-8,  u = r*ln2
#pragma once
#include <math.h>
#ifndef LPCN_EXP10_TABLE_QUAL
# ifdef __HIPCC__
#  define LPCN_EXP10_TABLE_QUAL static __device__
#  define LPCN_EXP10_FN __device__ __forceinline__
# else
#  define LPCN_EXP10_TABLE_QUAL static
#  define LPCN_EXP10_FN static inline
# endif
#endif
#include "lpcnet_exp10_gen.h"

LPCN_EXP10_FN double lpcn_exp10(const float xf)
{
    const double x = (double)xf;
    if (!(x == x)) return x;                                  // NaN
    if (x > 310.0) return HUGE_VAL;
    if (x < -330.0) return 0.0;
    // t = x * log2(10) as t_hi + t_lo (x has 24 significant bits: the error of t_lo is ~2^-106 |t|)
    const double t_hi = x * LPCN_LOG2_10_HI;
    const double t_lo = fma(x, LPCN_LOG2_10_HI, -t_hi) + x * LPCN_LOG2_10_LO;
    const double k = rint(t_hi * 128.0);                      // n*128 + j
    const double r_hi = t_hi - k * 0.0078125;                 // exact: both are multiples of ulp(t_hi), |r_hi| <= 2^-8
    const int ki = (int)k;
    const int n = ki >> 7, j = ki & 127;
    // u = (r_hi + t_lo) * ln2 as u_hi + u_lo
    const double u_hi = r_hi * LPCN_LN2_HI;
    const double u_lo = fma(r_hi, LPCN_LN2_HI, -u_hi) + (r_hi * LPCN_LN2_LO + t_lo * LPCN_LN2_HI);
    // e^u - 1 = u + q,  q = u^2/2 + ... + u^7/5040 (|u| < 2.8e-3: q < 4e-6, its rounding error is < 2^-70)
    const double q = u_hi * u_hi * (0.5 + u_hi * (1.0 / 6 + u_hi * (1.0 / 24 + u_hi * (1.0 / 120 + u_hi * (1.0 / 720 + u_hi * (1.0 / 5040))))));
    const double e_lo0 = u_lo + q + u_hi * u_lo;
    const double e_hi = u_hi + e_lo0;                          // fast two-sum (|u_hi| >= |e_lo0| unless both are tiny)
    const double e_lo = e_lo0 - (e_hi - u_hi);
    // T * (1 + e) = T_hi + (T_hi*e_hi + (T_lo + T_hi*e_lo + T_lo*e_hi))
    const double T_hi = lpcn_exp2_tab[j][0], T_lo = lpcn_exp2_tab[j][1];
    const double p_hi = T_hi * e_hi;
    const double p_lo = fma(T_hi, e_hi, -p_hi);
    const double s_hi = T_hi + p_hi;                          // two-sum, |T_hi| > |p_hi|
    const double s_lo = p_hi - (s_hi - T_hi);
    const double lo = s_lo + (p_lo + (T_lo + (T_hi * e_lo + T_lo * e_hi)));
    return ldexp(s_hi + lo, n);
}

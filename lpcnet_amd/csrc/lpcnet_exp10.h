// 10^x for a float x, correctly rounded to double for all practical purposes, without a math library.
//
// Why: lpc_from_cepstrum evaluates `pow(10.f, x)` in double and rounds the product with the band compensation to float
// (src/freq.c:317-318).  The PARITY TARGET is what the reference computes on its host: glibc's pow (2.35 on the boxes this
// was pinned on), which is accurate to < 1 ULP but NOT correctly rounded.  A device math library that is merely
// "<= 1 ULP" returns a neighbouring double for a sizeable fraction of arguments, and whenever that pair straddles a float
// rounding boundary one LPC coefficient changes in its last bit -- enough to send a free-running stream down a different
// trajectory for good.  This routine carries ~2^-70 relative error before the final rounding (double-double argument
// reduction, table and product), i.e. it IS the correctly rounded value except within ~2^-17 ULP of a double boundary.
// Measured against glibc 2.35 (tests/test_exp10.py): the DOUBLES differ for ~2^-10 of all arguments (16 775 of
// 2e7; 80-digit arithmetic says ours is the correctly rounded one each time), but none of those differences is visible
// after the rounding to float: all 2.7e8 floats of the reachable range x 18 band factors give identical floats.  The
// residual risk is therefore a glibc-vs-exact disagreement that ALSO straddles a float boundary, ~2^-10 x 2^-29 = 2^-39 per
// call -- at 8192 streams x 100 frames/s x 18 bands about one last-bit LPC flip per ten hours of a full 8-GPU node, after
// which that one free-running stream no longer matches the reference bit for bit (teacher-forced parity is unaffected).
// A build for a host whose libm differs (another glibc, musl) has the same exposure against THAT libm.
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

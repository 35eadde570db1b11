/*
 * fastmath.h -- lean exp() for the fast-mode kernels: exp(y) for y in [-inf, 0].
 *
 * The device library's exp() costs ~100 instructions per call on gfx950 (measured in
 * the objective kernel's ISA); the transition probability needs one exp(-alpha d) per
 * site and lane, so it was half of that kernel.  This version is 20 instructions:
 * clamp, k = rint(y log2 e), two-step Cody-Waite reduction, degree-13 Taylor
 * polynomial on |r| <= ln2/2 (truncation r^14/14! < 5e-18), ldexp.  Error < 1 ulp
 * (tests/test_detmath.py compares a host build against libm).
 */
#ifndef NGH_FASTMATH_H
#define NGH_FASTMATH_H

#if defined(__HIPCC__)
#define NGH_FM_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define NGH_FM_HD static inline
#endif

NGH_FM_HD double exp_nonpos(double y) {
  y = (y > -746.0) ? y : -746.0; /* exp(-746) underflows to 0; also maps -inf and NaN-free */
  const double k = __builtin_rint(y * 1.44269504088896338700e+00);
  double r = __builtin_fma(k, -6.93147180369123816490e-01, y);
  r = __builtin_fma(k, -1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0;                 /* 1/13! */
  p = __builtin_fma(p, r, 1.0 / 479001600.0);    /* 1/12! */
  p = __builtin_fma(p, r, 1.0 / 39916800.0);
  p = __builtin_fma(p, r, 1.0 / 3628800.0);
  p = __builtin_fma(p, r, 1.0 / 362880.0);
  p = __builtin_fma(p, r, 1.0 / 40320.0);
  p = __builtin_fma(p, r, 1.0 / 5040.0);
  p = __builtin_fma(p, r, 1.0 / 720.0);
  p = __builtin_fma(p, r, 1.0 / 120.0);
  p = __builtin_fma(p, r, 1.0 / 24.0);
  p = __builtin_fma(p, r, 1.0 / 6.0);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)k);
}

#endif

/*
 * detmath.h -- platform-independent exp()/log() for IEEE-754 binary64.
 *
 * Why this exists: the reference computes everything in log space with libm
 * exp/log (shared/gen_func.cpp:135-151 logsum, shared/HMM.cpp:130-139
 * calc_trans).  Its finite-difference L-BFGS-B M-step (shared/bfgs.cpp:22-65)
 * amplifies last-bit differences of those two functions into ~1e-5 differences
 * of the final indF (SURVEY.md finding 4).  To make "GPU result == CPU result"
 * a statement that can be tested bit for bit, the exact-mode HIP kernels and
 * the oracle's `det` build both call THESE functions.  They use only IEEE
 * add/sub/mul/div and integer bit operations (no FMA contraction, no table
 * whose rounding depends on the host), so gcc on x86-64 and hipcc on gfx950
 * produce identical bits for identical inputs.
 *
 * Algorithms: the classic argument-reduction + minimax polynomial schemes
 * published with FreeBSD/Sun fdlibm (e_exp.c, e_log.c; "freely distributable"
 * Sun Microsystems 1993), restated here.  Accuracy < 1 ulp (checked against
 * glibc in tests/test_detmath.py).
 *
 * Both translation units that include this header MUST be compiled with
 * -ffp-contract=off.
 */
#ifndef NGH_DETMATH_H
#define NGH_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define NGH_HD __host__ __device__ __forceinline__
#else
#define NGH_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

NGH_HD uint64_t ngh_bits(double x) {
  uint64_t u;
  __builtin_memcpy(&u, &x, sizeof u);
  return u;
}

NGH_HD double ngh_from_bits(uint64_t u) {
  double x;
  __builtin_memcpy(&x, &u, sizeof x);
  return x;
}

NGH_HD int32_t ngh_hi(double x) { return (int32_t)(ngh_bits(x) >> 32); }
NGH_HD uint32_t ngh_lo(double x) { return (uint32_t)(ngh_bits(x) & 0xffffffffu); }

NGH_HD double ngh_with_hi(double x, int32_t hi) {
  return ngh_from_bits(((uint64_t)(uint32_t)hi << 32) | (ngh_bits(x) & 0xffffffffull));
}

/* Natural logarithm.  log(+-0) = -inf, log(x<0) = NaN, log(inf) = inf. */
NGH_HD double det_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  const double two54 = 1.80143985094819840000e+16;
  const double Lg1 = 6.666666666666735130e-01;
  const double Lg2 = 3.999999999940941908e-01;
  const double Lg3 = 2.857142874366239149e-01;
  const double Lg4 = 2.222219843214978396e-01;
  const double Lg5 = 1.818357216161805012e-01;
  const double Lg6 = 1.531383769920937332e-01;
  const double Lg7 = 1.479819860511658591e-01;

  int32_t hx = ngh_hi(x);
  uint32_t lx = ngh_lo(x);
  int32_t k = 0;

  if (hx < 0x00100000) { /* zero, subnormal or negative */
    if (((hx & 0x7fffffff) | (int32_t)lx) == 0)
      return ngh_from_bits(0xfff0000000000000ull); /* -inf */
    if (hx < 0)
      return ngh_from_bits(0x7ff8000000000000ull); /* NaN */
    k -= 54;
    x *= two54;
    hx = ngh_hi(x);
  }
  if (hx >= 0x7ff00000)
    return x + x; /* inf or NaN */

  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  int32_t i = (hx + 0x95f64) & 0x100000;
  x = ngh_with_hi(x, hx | (i ^ 0x3ff00000)); /* x in [sqrt(2)/2, sqrt(2)) */
  k += (i >> 20);
  double f = x - 1.0;
  double dk = (double)k;

  if ((0x000fffff & (2 + hx)) < 3) { /* |f| < 2^-20 */
    if (f == 0.0) {
      if (k == 0)
        return 0.0;
      return dk * ln2_hi + dk * ln2_lo;
    }
    double R0 = f * f * (0.5 - 0.33333333333333333 * f);
    if (k == 0)
      return f - R0;
    return dk * ln2_hi - ((R0 - dk * ln2_lo) - f);
  }

  double s = f / (2.0 + f);
  double z = s * s;
  double w = z * z;
  i = hx - 0x6147a;
  int32_t j = 0x6b851 - hx;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  double R = t2 + t1;
  if (i > 0) {
    double hfsq = 0.5 * f * f;
    if (k == 0)
      return f - (hfsq - s * (hfsq + R));
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  if (k == 0)
    return f - s * (f - R);
  return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

/* Exponential.  exp(-inf) = 0, exp(x < -745.13) = 0, exp(x > 709.78) = inf. */
NGH_HD double det_exp(double x) {
  const double o_threshold = 7.09782712893383973096e+02;
  const double u_threshold = -7.45133219101941108420e+02;
  const double ln2HI = 6.93147180369123816490e-01;
  const double ln2LO = 1.90821492927058770002e-10;
  const double invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01;
  const double P2 = -2.77777777770155933842e-03;
  const double P3 = 6.61375632143793436117e-05;
  const double P4 = -1.65339022054652515390e-06;
  const double P5 = 4.13813679705723846039e-08;
  const double twom1000 = 9.33263618503218878990e-302;

  int32_t hx = ngh_hi(x);
  int32_t xsb = (hx >> 31) & 1;
  hx &= 0x7fffffff;
  double hi = 0.0, lo = 0.0;
  int32_t k = 0;

  if (hx >= 0x40862E42) { /* |x| >= 709.78 */
    if (hx >= 0x7ff00000) {
      if (((hx & 0xfffff) | (int32_t)ngh_lo(x)) != 0)
        return x + x; /* NaN */
      return xsb ? 0.0 : x; /* exp(+-inf) */
    }
    if (x > o_threshold)
      return ngh_from_bits(0x7ff0000000000000ull); /* overflow */
    if (x < u_threshold)
      return 0.0; /* underflow */
  }

  if (hx > 0x3fd62e42) { /* |x| > 0.5 ln2 */
    if (hx < 0x3FF0A2B2) { /* |x| < 1.5 ln2 */
      if (xsb) {
        hi = x + ln2HI;
        lo = -ln2LO;
        k = -1;
      } else {
        hi = x - ln2HI;
        lo = ln2LO;
        k = 1;
      }
    } else {
      k = (int32_t)(invln2 * x + (xsb ? -0.5 : 0.5));
      double t = (double)k;
      hi = x - t * ln2HI; /* exact */
      lo = t * ln2LO;
    }
    x = hi - lo;
  } else if (hx < 0x3e300000) { /* |x| < 2^-28 */
    return 1.0 + x;
  }

  double t = x * x;
  double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0)
    return 1.0 - ((x * c) / (c - 2.0) - x);
  double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
  if (k >= -1021)
    return ngh_from_bits(ngh_bits(y) + ((uint64_t)(int64_t)k << 52));
  y = ngh_from_bits(ngh_bits(y) + ((uint64_t)(int64_t)(k + 1000) << 52));
  return y * twom1000;
}

#endif /* NGH_DETMATH_H */

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
#define NGH_HD __host__ __device__ inline __attribute__((always_inline))
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

/*
 * Select-form variants for the exact-mode recursion kernels.
 *
 * A recursion chain is one lane walking the sites one after the other: what it costs is the
 * number of instructions its wave issues per site, and a wave whose lanes take different
 * branches of det_exp / det_log issues all of them one after the other (both divisions of
 * det_exp's two endings, for instance).  The functions below perform, for every argument,
 * exactly the operations det_exp / det_log perform for that argument, on the same operands
 * in the same order -- the case analysis is done with selects on values every lane computes
 * -- so they return the same bits (tests/test_detmath.py compares them over 10^7 arguments
 * including every case boundary).
 */

/* det_exp(x), any x. */
NGH_HD double det_exp_sel(double x) {
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

  const int32_t hxs = ngh_hi(x);
  const int32_t hx = hxs & 0x7fffffff;
  const int neg = hxs < 0;
  const int is_nan = x != x;
  const int over = x > o_threshold;      /* includes +inf: det_exp returns +inf */
  const int under = x < u_threshold;     /* includes -inf: det_exp returns 0 */
  const int tiny = hx < 0x3e300000;      /* |x| < 2^-28: det_exp returns 1 + x */
  const int out = is_nan | over | under;
  const double xm = out ? 0.0 : x;       /* keeps the main path's int conversion in range */
  const int mid = hx > 0x3fd62e42;       /* |x| > 0.5 ln2 */
  const int near1 = hx < 0x3FF0A2B2;     /* ... and < 1.5 ln2: k = -1 or 1 */
  int32_t k = (int32_t)(invln2 * xm + (neg ? -0.5 : 0.5));
  k = near1 ? (neg ? -1 : 1) : k;
  k = (mid && !out) ? k : 0;
  const double t = (double)k;
  /* |k| = 1: xm - (+-ln2HI) and (+-1) * ln2LO are det_exp's xm -+ ln2HI and +-ln2LO;
   * k = 0: hi = xm, lo = +0, hi - lo = xm */
  const double hi = xm - t * ln2HI;
  const double lo = t * ln2LO;
  const double xr = hi - lo;
  const double tt = xr * xr;
  const double c = xr - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
  const double xc = xr * c;
  const double den = (k == 0) ? c - 2.0 : 2.0 - c;
  const double q = xc / den;
  const double y = (k == 0) ? 1.0 - (q - xr) : 1.0 - ((lo - q) - hi);
  const int deep = k < -1021;
  const int32_t kk = deep ? k + 1000 : k;
  const double ys = ngh_from_bits(ngh_bits(y) + ((uint64_t)(int64_t)kk << 52));
  double r = deep ? ys * twom1000 : ys;
  r = tiny ? 1.0 + x : r;
  r = under ? 0.0 : r;
  r = over ? ngh_from_bits(0x7ff0000000000000ull) : r;
  r = is_nan ? x + x : r;
  return r;
}

/* det_log(x): positive normal finite x in select form, anything else through det_log. */
NGH_HD double det_log_pos(double x) {
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01;
  const double Lg2 = 3.999999999940941908e-01;
  const double Lg3 = 2.857142874366239149e-01;
  const double Lg4 = 2.222219843214978396e-01;
  const double Lg5 = 1.818357216161805012e-01;
  const double Lg6 = 1.531383769920937332e-01;
  const double Lg7 = 1.479819860511658591e-01;

  int32_t hx = ngh_hi(x);
  if (hx < 0x00100000 || hx >= 0x7ff00000) return det_log(x);
  int32_t k = (hx >> 20) - 1023;
  hx &= 0x000fffff;
  const int32_t i = (hx + 0x95f64) & 0x100000;
  x = ngh_with_hi(x, hx | (i ^ 0x3ff00000));
  k += (i >> 20);
  const double f = x - 1.0;
  const double dk = (double)k;
  const int k0 = k == 0;
  /* |f| < 2^-20 */
  const int small = (0x000fffff & (2 + hx)) < 3;
  const double R0 = f * f * (0.5 - 0.33333333333333333 * f);
  double rs = k0 ? f - R0 : dk * ln2_hi - ((R0 - dk * ln2_lo) - f);
  const double rz = k0 ? 0.0 : dk * ln2_hi + dk * ln2_lo;
  rs = (f == 0.0) ? rz : rs;
  /* the general case */
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const int32_t ii = (hx - 0x6147a) | (0x6b851 - hx);
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double sa = s * (hfsq + R);
  const double ra = k0 ? f - (hfsq - sa) : dk * ln2_hi - ((hfsq - (sa + dk * ln2_lo)) - f);
  const double sb = s * (f - R);
  const double rb = k0 ? f - sb : dk * ln2_hi - ((sb - dk * ln2_lo) - f);
  const double rg = (ii > 0) ? ra : rb;
  return small ? rs : rg;
}

/* logsum of two terms (shared/gen_func.cpp:135-151 with n = 2; its max() is the macro
 * a >= b ? a : b) when the larger term M is finite: then its own exponent is exp(M - M) =
 * exp(0) = 1 exactly, 0 + x = x, and the two additions of the reference's loop are 1 + E
 * in either order (IEEE addition commutes), E the other term's exponential.  Any other M
 * (-inf: the reference returns -inf; +inf, NaN) takes the loop as written. */
NGH_HD double det_logsum2(double a0, double a1) {
  const int ge = a1 >= a0;
  const double M = ge ? a1 : a0;
  const double m = ge ? a0 : a1;
  const uint32_t hm = (uint32_t)ngh_hi(M) & 0x7fffffffu;
  if (hm >= 0x7ff00000u) { /* M is infinite or NaN */
    if (M == ngh_from_bits(0xfff0000000000000ull)) return M;
    double sum = 0;
    sum += det_exp(a0 - M);
    sum += det_exp(a1 - M);
    return det_log(sum) + M;
  }
  const double E = det_exp_sel(m - M);
  return det_log_pos(E + 1.0) + M;
}

/*
 * Chain forms: the two calls a recursion chain makes per site (det_logsum2_chain below), cut
 * down to what their arguments can be there.  det_exp's argument is (smaller term) - (larger
 * term) <= 0 and det_log's is 1 + exp(.) in [1, 2]; for those,
 *   - det_exp's ending for k = 0, 1 - (xc / (c - 2) - x), equals its general ending
 *     1 - ((lo - xc / (2 - c)) - hi) with hi = x, lo = +0: c - 2 = -(2 - c) and a quotient
 *     changes sign with its divisor, exactly; 0 - q = -q;
 *   - det_log's three endings all have the shape dk ln2_hi - (T - f), and its k = 0 forms
 *     f - T are that shape with dk = 0: 0 - (T - f) = f - T in round-to-nearest, T - 0 = T;
 * so one form serves all cases and the case analysis shrinks to three selects.  Anything
 * outside the fast domain (NaN, +-inf, x < -708 that does not underflow, subnormal sums) makes
 * the whole wave take the select forms above: NGH_ANY is a wave vote on the device, so the
 * branch is uniform.  Same bits as det_exp / det_log for every argument
 * (tests/test_detmath.py).
 */
#if defined(__HIP_DEVICE_COMPILE__)
#define NGH_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0)
#else
#define NGH_ANY(c) (c)
#endif
/* the rare routes out of line, so that a chain's hot loop stays a short straight run */
#if defined(__HIPCC__)
#define NGH_COLD static __host__ __device__ __attribute__((noinline))
#else
#define NGH_COLD static __attribute__((noinline))
#endif
NGH_COLD double det_exp_sel_cold(double x) { return det_exp_sel(x); }
NGH_COLD double det_log_pos_cold(double x) { return det_log_pos(x); }
NGH_COLD double det_logsum2_cold(double a0, double a1) { return det_logsum2(a0, a1); }

/* det_exp(x) for -708 <= x <= -2^-28, or x < u_threshold (`under`: det_exp returns 0):
 * k >= -1021, so there is no second scaling step */
NGH_HD double det_exp_chain_fast(double x, int under) {
  const double ln2HI = 6.93147180369123816490e-01;
  const double ln2LO = 1.90821492927058770002e-10;
  const double invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01;
  const double P2 = -2.77777777770155933842e-03;
  const double P3 = 6.61375632143793436117e-05;
  const double P4 = -1.65339022054652515390e-06;
  const double P5 = 4.13813679705723846039e-08;

  const int32_t hx = ngh_hi(x) & 0x7fffffff;
#if defined(__HIP_DEVICE_COMPILE__)
  const double xm = x; /* v_cvt_i32_f64 saturates; an underflowing lane's result is unused */
#else
  const double xm = under ? -746.0 : x; /* keeps the int conversion defined on the host */
#endif
  int32_t k = (int32_t)(invln2 * xm + -0.5);
  k = (hx < 0x3FF0A2B2) ? -1 : k;    /* |x| < 1.5 ln2 */
  k = (hx > 0x3fd62e42) ? k : 0;     /* |x| <= 0.5 ln2 */
  const double t = (double)k;
  const double hi = xm - t * ln2HI;
  const double lo = t * ln2LO;
  const double xr = hi - lo;
  const double tt = xr * xr;
  const double c = xr - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
  const double q = (xr * c) / (2.0 - c);
  const double y = 1.0 - ((lo - q) - hi);
  const double r = ngh_from_bits(ngh_bits(y) + ((uint64_t)(int64_t)k << 52));
  return under ? 0.0 : r;
}

/* whether x is in det_exp_chain_fast's domain */
NGH_HD int det_exp_chain_ok(double x, int under) {
  return (x <= -3.7252902984619140625e-09 /* -2^-28 */ && x >= -708.0) || under;
}

NGH_HD double det_exp_chain(double x) {
  const int under = x < -7.45133219101941108420e+02;
  if (NGH_ANY(!det_exp_chain_ok(x, under))) return det_exp_sel_cold(x);
  return det_exp_chain_fast(x, under);
}

/* det_log(x) for positive normal finite x */
NGH_HD double det_log_chain_fast(double x) {
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01;
  const double Lg2 = 3.999999999940941908e-01;
  const double Lg3 = 2.857142874366239149e-01;
  const double Lg4 = 2.222219843214978396e-01;
  const double Lg5 = 1.818357216161805012e-01;
  const double Lg6 = 1.531383769920937332e-01;
  const double Lg7 = 1.479819860511658591e-01;

  int32_t hx = ngh_hi(x);
  int32_t k = (hx >> 20) - 1023;
  hx &= 0x000fffff;
  const int32_t i = (hx + 0x95f64) & 0x100000;
  x = ngh_with_hi(x, hx | (i ^ 0x3ff00000));
  k += (i >> 20);
  const double f = x - 1.0;
  const double dk = (double)k;
  const double dh = dk * ln2_hi, dl = dk * ln2_lo;
  const double Ts = f * f * (0.5 - 0.33333333333333333 * f) - dl; /* |f| < 2^-20 */
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double Ta = hfsq - (s * (hfsq + R) + dl);
  const double Tb = s * (f - R) - dl;
  double T = (((hx - 0x6147a) | (0x6b851 - hx)) > 0) ? Ta : Tb;
  T = ((0x000fffff & (2 + hx)) < 3) ? Ts : T;
  return dh - (T - f);
}

NGH_HD double det_log_chain(double x) {
  const int32_t hx = ngh_hi(x);
  if (NGH_ANY(hx < 0x00100000 || hx >= 0x7ff00000)) return det_log_pos_cold(x);
  return det_log_chain_fast(x);
}

/* det_logsum2 for a recursion chain */
NGH_HD double det_logsum2_chain(double a0, double a1) {
  const int ge = a1 >= a0;
  const double M = ge ? a1 : a0;
  const double m = ge ? a0 : a1;
  if (NGH_ANY(((uint32_t)ngh_hi(M) & 0x7fffffffu) >= 0x7ff00000u)) return det_logsum2_cold(a0, a1);
  return det_log_chain(det_exp_chain(m - M) + 1.0) + M;
}

#endif /* NGH_DETMATH_H */

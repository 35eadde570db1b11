// exact_dev.hpp -- device helpers shared by the exact-mode kernels (kernels_exact.hip,
// kernels_exact_pc.hip): the reference's scalar routines in its operation order, with exp/log
// from detmath.h.  Included INSIDE `namespace nghmm { namespace {` of a translation unit that
// has already included <cstdint> and detmath.h and is compiled with -ffp-contract=off.
#pragma once

constexpr double kINF = 1e15;      // shared/gen_func.hpp:15
constexpr double kEPS = 1e-5;      // shared/gen_func.hpp:16
constexpr uint64_t kUnreadBits = 0x7ff8dead00000001ull;  // NGHMM_GL_UNREAD (include/nghmm.h)
#define NEG_INFINITY (-__builtin_huge_val())

// shared/gen_func.cpp:135-151, n = 2.  max() there is the macro (a >= b ? a : b).
__device__ __forceinline__ double logsum2(double a0, double a1) {
  double M = a0;
  M = (a1 >= M) ? a1 : M;
  if (M == NEG_INFINITY) return NEG_INFINITY;
  double sum = 0;
  sum += det_exp(a0 - M);
  sum += det_exp(a1 - M);
  return det_log(sum) + M;
}

// shared/gen_func.cpp:135-151, n = 3
__device__ __forceinline__ double logsum3(double a0, double a1, double a2) {
  double M = a0;
  M = (a1 >= M) ? a1 : M;
  M = (a2 >= M) ? a2 : M;
  if (M == NEG_INFINITY) return NEG_INFINITY;
  double sum = 0;
  sum += det_exp(a0 - M);
  sum += det_exp(a1 - M);
  sum += det_exp(a2 - M);
  return det_log(sum) + M;
}

// conv_space(.., log) for one value (shared/gen_func.cpp:123-130)
__device__ __forceinline__ double log_or_minf(double v) {
  double r = det_log(v);
  return (r == NEG_INFINITY) ? -kINF : r;
}

// calc_HWE, log scale (shared/gen_func.cpp:938-957)
__device__ __forceinline__ void hwe_log(double maf, double F, double& h0, double& h1, double& h2) {
  h0 = (1 - maf) * (1 - maf) + (1 - maf) * maf * F;
  h1 = 2 * (1 - maf) * maf - 2 * (1 - maf) * maf * F;
  h2 = maf * maf + (1 - maf) * maf * F;
  h0 = log_or_minf(h0);
  h1 = log_or_minf(h1);
  h2 = log_or_minf(h2);
  if (F == 1) h1 = -kINF;
}

// calc_emission (shared/HMM.cpp:144-154)
__device__ __forceinline__ double emission_log(double g0, double g1, double g2, double maf, int k) {
  double h0, h1, h2;
  hwe_log(maf, (double)k, h0, h1, h2);
  return logsum3(g0 + h0, g1 + h1, g2 + h2);
}

// the four log transition probabilities of one site (shared/HMM.cpp:130-139)
struct Trans {
  double t00, t10, t01, t11;  // t[k][l]
};
__device__ __forceinline__ Trans calc_trans_all(double q0, double q1, double alpha, double d) {
  Trans t;
  double c = det_exp(-alpha * d);
  double b0 = (1 - c) * q0;
  double b1 = (1 - c) * q1;
  t.t10 = det_log(b0);      // k=1 -> l=0
  t.t00 = det_log(b0 + c);  // k=0 -> l=0
  t.t01 = det_log(b1);      // k=0 -> l=1
  t.t11 = det_log(b1 + c);  // k=1 -> l=1
  return t;
}

// shared/gen_func.cpp:55-70
__device__ __forceinline__ double check_interv(double v, bool& isnan_flag) {
  if (v != v) {
    isnan_flag = true;
    return v;
  }
  if (v < kEPS)
    v = 0;
  else if (v > 1 - kEPS)
    v = 1;
  return v;
}


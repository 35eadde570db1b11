// fast_dev.hpp -- what the fast-mode translation units share on the device side: the layout
// constants and index helpers, the 2x2 operator algebra with integer exponents, the per-site
// sources of a forward walk (stored emission ratios, or dense / packed likelihoods on the way),
// the group descriptor of an objective round.  Included by kernels_fast.hip (layouts, loading,
// handle state), kernels_fast_walks.hip (objective rounds), kernels_fast_estep.hip (E-step and
// the site-shard edges) and kernels_fast_estmaf.hip (allele-frequency step); everything is in an
// anonymous namespace, i.e. private to each of them.  kernels_fast.hip's header comment has the
// design.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "detmath.h"
#include "fastmath.h"
#include "kernels.hpp"
#include "kernels_fast.hpp"

#pragma clang fp contract(fast)

namespace nghmm {

namespace {

constexpr double kINF = 1e15;
constexpr double kEPS = 1e-5;
constexpr int MAXP = 5;   // probe points per group (f(x) + 4 finite-difference probes)
constexpr int RENORM = 8; // sites between rescalings
constexpr int CK = 8;     // sites between forward checkpoints (== RENORM: stored right after a rescale)
constexpr int UF = 4;     // prefetch depth (sites) of the E-step sweeps
constexpr int NB = 4;     // objective kernel: load buffers in flight ...
constexpr int UG = 2;     // ... of this many sites each (NB * UG == RENORM)

// The E-step's posteriors, "tile" layout: one tile per tile row c*T + t (the 64 sites
// (c*64 + l)*T + t, l = 0..63) -- [tile row][i / 8][l][i % 8]: the eight individuals of a group
// are adjacent, so the 64 B sector around a posterior holds ONE site (est_maf fetches nothing
// it does not use), and a group's 64 x 8 block of a tile row is 4 KB contiguous (the backward
// sweep writes it with full wave-stores from eight waves through LDS).
constexpr bool kPost8 = true;
__host__ __device__ constexpr uint64_t post_tile_doubles(uint64_t I) {
  return kPost8 ? ((I + 7) / 8) * 512 : I * 64;
}
// offset of individual i inside a tile, relative to (i = 0, lane l)
__device__ __forceinline__ uint64_t post_ind_off(uint64_t i) {
  return kPost8 ? (i >> 3) * 512 + (i & 7) : i * 64;
}
// posterior of (tile row, individual 0, lane l)
__device__ __forceinline__ uint64_t post_lane_off(uint64_t tile_row, uint64_t l, uint64_t I) {
  return tile_row * post_tile_doubles(I) + (kPost8 ? l * 8 : l);
}

struct GroupDesc {
  uint32_t ind;
  uint32_t np;
  uint32_t mode;     // 0 = general points, else fd_mode(nf, na, small): see lkl_run_fd
  uint32_t pad;
  double F[MAXP];
  double A[MAXP];
  uint32_t out_idx[MAXP];
  uint32_t pad2;
};

// ---- 2x2 operators with a binary exponent -------------------------------
struct Op {
  double a00, a01, a10, a11;
  int ex;
};

__device__ __forceinline__ int exp_of(double mx) {
  // exponent e with mx = m * 2^e, m in [0.5, 1); 0 for mx == 0 or non-finite
  return (mx > 0.0 && mx < __builtin_huge_val()) ? __builtin_amdgcn_frexp_exp(mx) : 0;
}

__device__ __forceinline__ void renorm(Op& m) {
  const double mx = fmax(fmax(m.a00, m.a01), fmax(m.a10, m.a11));
  const int e = exp_of(mx);
  m.a00 = __builtin_ldexp(m.a00, -e);
  m.a01 = __builtin_ldexp(m.a01, -e);
  m.a10 = __builtin_ldexp(m.a10, -e);
  m.a11 = __builtin_ldexp(m.a11, -e);
  m.ex += e;
}

__device__ __forceinline__ void renorm2(double& v0, double& v1, int& ex) {
  const int e = exp_of(fmax(v0, v1));
  v0 = __builtin_ldexp(v0, -e);
  v1 = __builtin_ldexp(v1, -e);
  ex += e;
}

// L applied first, then R (row-vector convention v' = v M)
__device__ __forceinline__ Op op_mul(const Op& L, const Op& R) {
  Op o;
  o.a00 = fma(L.a00, R.a00, L.a01 * R.a10);
  o.a01 = fma(L.a00, R.a01, L.a01 * R.a11);
  o.a10 = fma(L.a10, R.a00, L.a11 * R.a10);
  o.a11 = fma(L.a10, R.a01, L.a11 * R.a11);
  o.ex = L.ex + R.ex;
  renorm(o);
  return o;
}

__device__ __forceinline__ Op op_load(const double* __restrict__ m) {
  return Op{m[0], m[1], m[2], m[3], (int)m[4]};
}

__device__ __forceinline__ Op op_shfl_down(const Op& m, int off) {
  Op o;
  o.a00 = __shfl_down(m.a00, off);
  o.a01 = __shfl_down(m.a01, off);
  o.a10 = __shfl_down(m.a10, off);
  o.a11 = __shfl_down(m.a11, off);
  o.ex = __shfl_down(m.ex, off);
  return o;
}

// one site applied to both rows of an operator:  row' = (c row + a (row.1) q) * e
// with the products ce_k = c e_k and g_k = a e_k q_k formed by the caller
__device__ __forceinline__ void op_step(Op& m, double ce0, double ce1, double g0, double g1) {
  const double s0 = m.a00 + m.a01;
  const double s1 = m.a10 + m.a11;
  m.a00 = fma(g0, s0, ce0 * m.a00);
  m.a01 = fma(g1, s0, ce1 * m.a01);
  m.a10 = fma(g0, s1, ce0 * m.a10);
  m.a11 = fma(g1, s1, ce1 * m.a11);
}

// The same site with the operator kept as M / c ("kappa form", the objective kernels' SMALL
// versions): M_s / c_s = (I + kappa_s 1 q^T) diag(1, rho_s) with kappa = (1 - c) / c = expm1(alpha d),
//   row' = (row_0 + g0 (row.1),  rho row_1 + g1 (row.1)),   g0 = kappa q0,  g1 = kappa q1 rho
// -- four instructions per row instead of five, every term non-negative as before (no
// cancellation), and the factor left out is known in closed form: prod_s c_s = exp(-alpha sum_s d_s)
// over the lane-chunk (chunk_scale, formed at load), put back once at the end of the walk.
// Chromosome starts (c = 0; stored as d = kDStart) need no select: the argument alpha d is
// clamped to KAPPA_XMAX = 2^16, where the polynomial gives K ~ 2^99.7 for every alpha >= 1e-15 -- the
// row's own part is then 1 / (K q) of the other, at most 1e-15 (q >= 1e-15, EM.cpp:425; 1e-24 at
// q = 1e-6): the rank-one operator 1 q^T diag(e) to rounding, times a constant that the end of
// the walk divides out again per start.  Eight starts in a row between two rescales are 2^798:
// 2^225 of the double range are left for that block's emission ratios (the c form had 2^1023).
// The alpha probes' small exponential sees the distance clamped to KAPPA_DCLAMP (above every
// finite distance these kernels are given: |alpha_0 - alpha_probe| d_max <= 1e-3 with probes >= 4e-6
// apart), so their constant is that of point 0 within a few per cent and the points' common
// exponent holds however many starts a lane-chunk has.
constexpr double kDStart = 1e22;           // a chromosome start in pos_il (exp(-alpha d) = 0, alpha >= 1e-15)
constexpr double KAPPA_XMAX = 65536.0;     // 2^16 <= 1e-15 * kDStart
constexpr double KAPPA_DCLAMP = 1000.0;
// min of two numbers that are not NaN: the one instruction (fmin() canonicalises a loaded operand
// first, a v_max_f64 x, x per site)
__device__ __forceinline__ double min_num(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ void op_step_k(Op& m, double rho, double g0, double g1) {
  const double s0 = m.a00 + m.a01;
  const double s1 = m.a10 + m.a11;
  m.a00 = fma(g0, s0, m.a00);
  m.a01 = fma(g1, s0, rho * m.a01);
  m.a10 = fma(g0, s1, m.a10);
  m.a11 = fma(g1, s1, rho * m.a11);
}

// expm1(x) / x for 0 <= x <= 2^-6 (the first dropped term x^7/8! < 6e-18): the caller multiplies
__device__ __forceinline__ double expm1_over_x_tiny(double x) {
  double p = 1.0 / 5040.0;
  p = fma(p, x, 1.0 / 720.0);
  p = fma(p, x, 1.0 / 120.0);
  p = fma(p, x, 1.0 / 24.0);
  p = fma(p, x, 1.0 / 6.0);
  p = fma(p, x, 0.5);
  return fma(p, x, 1.0);
}

// expm1(x) of the alpha probes' x = (alpha_probe - alpha_0) d, to the degree exp_small<DEG> has
template <int DEG>
__device__ __forceinline__ double expm1_small(double x) {
  if constexpr (DEG == 2) return fma(0.5 * x, x, x);
  else return fma(fma(x, fma(x, 1.0 / 24, 1.0 / 6), 0.5) * x, x, x);
}

// sum over the wave in a fixed (butterfly) order: the same bits in every lane and every run
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__device__ __forceinline__ double coanc(double alpha, double d) {
  // exp(-alpha d): 0 at chromosome starts (d = +inf), 1 on padding sites (d = 0)
  return exp_nonpos(-alpha * d);
}

// by-products of a forward walk that the E-step consumes (see "E-step" below)
struct EmitPtrs {
  double* __restrict__ lane_ops;  // [I][J][5]: operator of the whole lane-chunk
  double2* __restrict__ ckpt;     // see "E-step" below
};

__device__ __forceinline__ void emit_checkpoint(double2* __restrict__ ck, uint64_t wave,
                                                uint64_t nblk, uint64_t b, int lane, const Op& R) {
  double2* o = ck + ((wave * nblk + b) * 2) * 64 + lane;
  o[0] = double2{R.a00, R.a01};
  o[64] = double2{R.a10, R.a11};
}

__device__ __forceinline__ void emit_lane_op(double* __restrict__ lane_ops, uint64_t wave, int lane,
                                             const Op& R) {
  double* out = lane_ops + (wave * 64 + lane) * 5;
  out[0] = R.a00;
  out[1] = R.a01;
  out[2] = R.a10;
  out[3] = R.a11;
  out[4] = (double)R.ex;
}

// ---- objective: chunk operators of <= 5 points per individual ------------
// 1/x to full precision (v_rcp_f64 + two Newton steps); 1/0 = inf as in IEEE
__device__ __forceinline__ double rcp_nr2(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
}

// 1/x to ~46 bits: v_rcp_f64 (about 23 bits) + one Newton step; x in (0, 3]
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
}

// exp(x) of the alpha probes, x = (alpha_0 - alpha_probe) d: degree 4 for |x| <= 1e-3 (x^5/120
// is the first dropped term), degree 2 for |x| <= 1e-5 (x^3/6 <= 1.7e-16: below half an ulp of
// the result) -- the probes sit eh = (1e-8 (alpha + 1))^0.67 <= 2.3e-5 from alpha
// (shared/bfgs.cpp:33), so degree 2 serves every data set whose largest finite distance is
// below ~0.4 Mb; the host picks per group (fd_pattern)
template <int DEG>
__device__ __forceinline__ double exp_small(double x) {
  if constexpr (DEG == 2) return fma(x, fma(x, 0.5, 1.0), 1.0);
  else return fma(x, fma(x, fma(x, fma(x, 1.0 / 24, 1.0 / 6), 0.5), 1.0), 1.0);
}

// GroupDesc::mode: 0 = general points; else the finite-difference pattern of
// shared/bfgs.cpp:22-43 -- point 0 = x, then NF probes that differ from it in F only,
// then NA probes that differ in alpha only, by so little that exp(-(alpha +- eh) d) =
// exp(-alpha d) * exp(-+ eh d) with a tiny second argument.
constexpr uint32_t FD_FLAG = 0x100, FD_SMALL = 0x200, FD_XDEG2 = 0x400;
// an exponent of its own for every point (below: "Dynamic range"); OR-ed onto an fd_mode
constexpr uint32_t FD_OWNEX = 0x800;
__host__ __device__ constexpr uint32_t fd_mode(int nf, int na, bool small, bool xdeg2 = false) {
  return FD_FLAG | (small ? FD_SMALL : 0u) | (xdeg2 ? FD_XDEG2 : 0u) | ((uint32_t)nf << 2) |
         (uint32_t)na;
}

// Recognise the finite-difference pattern of one objective + gradient evaluation
// (bfgs_batch.cpp plan(): x, then the F probes, then the alpha probes) with alpha probes
// close enough for exp_small<4> (or <2>) on every finite distance of this data set.
//
// alpha_small_min: the small-alpha versions run in the kappa form (op_step_k), which forms
// kappa = expm1(alpha d) to full precision, where the reference's 1 - exp(-alpha d) (HMM.cpp:130-139)
// carries the rounding of exp(-alpha d) to the 2^-53 grid below 1 -- a RELATIVE error 1e-16 / (alpha d)
// of the switching probability, which a tract boundary that the data force into a stretch of n
// sites turns into ~1e-16 / (sqrt(n) alpha d) of log-likelihood: nothing at alpha d ~ 1e-3, visible
// at alpha d < ~1e-8, 0.1 at alpha's lower bound 1e-15 (measured).  Individuals whose alpha is
// that small (alpha * mean finite distance < 1e-6) take the general-exp version, whose c = exp(-alpha
// d), 1 - c round as the reference's do.
__host__ __device__ inline uint32_t fd_pattern(const GroupDesc& G, double dmax, uint64_t T, bool forced_visits,
                           double alpha_small_min) {
  if (G.np < 2 || !(G.F[0] > 0 && G.F[0] < 1) || !(G.A[0] > 0)) return 0;
  int nf = 0, na = 0;
  bool ownex = false;
  double xmax = 0;  // largest |alpha_0 - alpha_probe| d over the data's finite distances
  double damax = 0;  // largest |alpha_0 - alpha_probe|
  for (uint32_t p = 1; p < G.np; ++p) {
    if (G.A[p] == G.A[0] && G.F[p] > 0 && G.F[p] < 1) {
      if (na) return 0;  // F probes come first
      // The pattern kernel rescales all points by point 0's exponent.  A site that forces
      // the non-IBD state (a called heterozygote: e1 = 0) multiplies an F probe's operator by
      // rho0 = (1 - F_p) / (1 - F_0) relative to point 0's; with F_0 at its upper bound that
      // is ~1e10 per such site and would overflow within a lane-chunk.  For called genotypes
      // (packed handles: such sites exist by construction) keep rho0^T inside the double
      // range, else the general kernel (an exponent per point) takes the group.  Likelihood
      // data have no forced visits; should a probe overflow there all the same, its value
      // comes back non-finite and the host re-evaluates it with the general kernel.
      const double rho0 = (1 - G.F[p]) / (1 - G.F[0]);
      if (forced_visits && !(fabs(log(rho0)) * (double)T <= 600.0)) {
        // ... or, where the probe stays in range over the eight sites between two rescales
        // (always, with F inside [1e-15, 1 - 1e-15]), the pattern kernel with an exponent per
        // point: the shared transition terms are still formed once per site
        if (!(fabs(log(rho0)) * 8.0 <= 600.0)) return 0;
        ownex = true;
      }
      ++nf;
    } else if (G.F[p] == G.F[0] && fabs(G.A[p] - G.A[0]) * dmax <= 1e-3) {
      ++na;
      xmax = fmax(xmax, fabs(G.A[p] - G.A[0]) * dmax);
      damax = fmax(damax, fabs(G.A[p] - G.A[0]));
    } else {
      return 0;
    }
  }
  const bool ok = (nf == 2 && na == 2) || (nf == 1 && na == 2) || (nf == 2 && na == 1) ||
                  (nf == 1 && na == 1) || (nf == 2 && na == 0) || (nf == 0 && na == 2);
  if (!ok) return 0;
  // The kappa form's alpha probes see the distance clamped to KAPPA_DCLAMP (op_step_k above): right
  // only while every finite distance is below it, and their constant at a chromosome start stays
  // near point 0's only while |alpha_0 - alpha_probe| KAPPA_DCLAMP is small.  The M-step's own
  // probes (eh <= 2.3e-5 apart, distances <= 46 Mb: dbfgs_available) are far inside both; points a
  // caller of nghmm_lkl_batch chooses need not be, and then take the general-exp version.
  const bool kappa_ok = na == 0 || (dmax < KAPPA_DCLAMP && damax * KAPPA_DCLAMP <= 0.05);
  return fd_mode(nf, na, G.A[0] * dmax <= 0.015625 && G.A[0] >= alpha_small_min && kappa_ok,
                 na > 0 && xmax <= 1e-5) |
         (ownex ? FD_OWNEX : 0u);
}

// Every recognised pattern runs on the kernel of the FULL one -- two F probes, two alpha probes:
// a probe the optimizer did not ask for (a parameter on a bound: a one-sided difference,
// shared/bfgs.cpp:36-41; a fixed parameter) is filled with a copy of point 0, in the positions
// the kernel expects (0: x, 1-2: F probes, 3-4: alpha probes).  The points of a group do not enter
// each other's arithmetic (one exponent, point 0's, rescales all of them), so the real points'
// values are the same bits as in a kernel of their own pattern; a copy takes point 0's operations
// (F ratios 1, alpha difference 0) and writes point 0's value to point 0's place.  One kernel per
// (small-alpha, degree, exponent) instead of one per pattern as well: a round of individuals at
// different bounds is one launch, not up to six (and the library a fifth of the objective code).
__host__ __device__ inline void fd_pad(GroupDesc& G) {
  if (!(G.mode & FD_FLAG)) return;
  const uint32_t nf = (G.mode >> 2) & 3u, na = G.mode & 3u;
  if (nf == 2 && na == 2) return;
  GroupDesc P = G;
  for (uint32_t k = 0; k < 2; ++k) {
    const uint32_t sf = k < nf ? 1 + k : 0, sa = k < na ? 1 + nf + k : 0;
    P.F[1 + k] = G.F[sf];
    P.A[1 + k] = G.A[sf];
    P.out_idx[1 + k] = G.out_idx[sf];
    P.F[3 + k] = G.F[sa];
    P.A[3 + k] = G.A[sa];
    P.out_idx[3 + k] = G.out_idx[sa];
  }
  P.np = 5;
  P.mode = (G.mode & ~0xfu) | (2u << 2) | 2u;
  G = P;
}

// compact index of a mode (0 = general): the device-planned rounds keep one worklist per mode
constexpr uint32_t kModeSlots = 1 + 8 * 16;
__host__ __device__ constexpr uint32_t mode_slot(uint32_t mode) {
  return mode == 0 ? 0u : 1u + ((mode >> 9) & 7u) * 16u + (mode & 15u);
}

// Where a forward walk gets its per-site inputs from.  Plain: the materialised emission
// ratios e_il.  Fresh: the first walk after an allele-frequency update computes the
// emissions itself from the interleaved linear genotype likelihoods and the new
// frequencies (calc_emission, shared/HMM.cpp:144-154, in linear space: e_k = sum_g p_g
// HWE_g(f, F = k)) and WRITES their ratio to e_il for every later pass (and the sum of
// log e0 to base_c) -- the separate refresh pass
// (24 B read + 8 B written per site and individual) disappears into a kernel that is
// FP64-bound anyway.
struct LklArrays {
  const double* __restrict__ e_il;      // emission ratios rho = e1 / e0
  const double* __restrict__ pos_il;
  const double2* __restrict__ glq_il;   // likelihoods relative to the cell's largest (glq_decode)
  const double* __restrict__ freq_il;
  double* __restrict__ e_out;           // == e_il, written by the fresh walk
  const uint32_t* __restrict__ geno_il; // packed handle: 2-bit codes, 16 sites of a lane per word
  double u_lin;                         // packed handle: linear likelihood of a uniform cell
  double* __restrict__ base_c;          // [I][C]: sum of log e0 over the wave's sites (see top)
  const double* __restrict__ gl_scale_c; // [I][C]: sum of the cells' largest log likelihoods, or null
  const double2* __restrict__ chunk_scale; // [C][64]: (sum of the finite distances, number of
                                           // chromosome starts) of a lane-chunk (op_step_k)
};

// The interleaved copy of the likelihoods holds each cell RELATIVE TO ITS LARGEST value: a
// common factor of (p0, p1, p2) scales both emissions alike, so it moves from the ratio
// into `base` -- as the cell's largest LOG likelihood, summed per wave at load
// (gl_scale_c).  One of the three is then exactly 1 and need not be stored: 16 B per cell
// instead of 24, the other two in index order with the position of the 1 in their sign bits
// (likelihoods are non-negative).
__device__ __forceinline__ double2 glq_encode(double l0, double l1, double l2) {
  const int tag = (l0 >= l1 && l0 >= l2) ? 0 : (l1 >= l2 ? 1 : 2);
  const double m = tag == 0 ? l0 : tag == 1 ? l1 : l2;
  const double a = exp((tag == 0 ? l1 : l0) - m), b = exp((tag == 2 ? l1 : l2) - m);
  return double2{(tag & 1) ? -a : a, (tag & 2) ? -b : b};
}
__device__ __forceinline__ void glq_decode(double2 q, double& p0, double& p1, double& p2) {
  // sign of x: the 1 is p1; sign of y: the 1 is p2; neither: p0 (tests on the high words: the
  // stored value may be -0.0)
  const bool sx = (int)(ngh_bits(q.x) >> 32) < 0, sy = (int)(ngh_bits(q.y) >> 32) < 0;
  const double a = fabs(q.x), b = fabs(q.y);
  p0 = (sx || sy) ? a : 1.0;
  p1 = sx ? 1.0 : (sy ? b : a);
  p2 = sy ? 1.0 : b;
}

// Running product of the e0 of a lane's sites (fresh walks): one multiply per site, the
// exponent taken out every RENORM sites -- e0 >= freq^2 or (1 - freq)^2 times the largest
// likelihood of the cell, so eight factors stay far inside the double range.
struct BaseAcc {
  double P = 1.0;
  int ex = 0;
  __device__ __forceinline__ void mul(double e0) { P *= e0; }
  __device__ __forceinline__ void rescale() {
    const int e = exp_of(P);
    P = __builtin_ldexp(P, -e);
    ex += e;
  }
  // log of the product; -inf when a site has no probability mass (e0 = 0), NaN for NaN
  __device__ __forceinline__ double log_value() const {
    return log(P) + (double)ex * 0.6931471805599453094;
  }
};

// emissions of one cell -> the ratio the walks run on; e0 = 0 (no mass) gives inf or NaN,
// which ends as a non-finite likelihood like the zero emissions themselves would
__device__ __forceinline__ double emission_ratio(double e0, double e1) {
  return e1 * rcp_nr2(e0);
}

// Every source hands a site to the walk as (rho, d): the emissions are (1, rho).
struct SrcPlain {
  const double* __restrict__ ep;
  const double* __restrict__ dp;
  struct Buf {
    double r;
    double d;
  };
  __device__ __forceinline__ SrcPlain(const LklArrays& A, uint64_t wave_base, uint64_t pos_base)
      : ep(A.e_il + wave_base), dp(A.pos_il + pos_base) {}
  __device__ __forceinline__ Buf load(uint64_t t) const { return Buf{ep[t * 64], dp[t * 64]}; }
  __device__ __forceinline__ void get(const Buf& b, uint64_t, double& rho, double& d) {
    rho = b.r;
    d = b.d;
  }
  __device__ __forceinline__ void rescale() {}
};

struct SrcFresh {
  const double2* __restrict__ gq;
  const double* __restrict__ fp;
  const double* __restrict__ dp;
  double* __restrict__ eo;
  BaseAcc base;
  struct Buf {
    double2 q;
    double f, d;
  };
  __device__ __forceinline__ SrcFresh(const LklArrays& A, uint64_t wave_base, uint64_t pos_base)
      : gq(A.glq_il + wave_base), fp(A.freq_il + pos_base), dp(A.pos_il + pos_base),
        eo(A.e_out + wave_base) {}
  __device__ __forceinline__ Buf load(uint64_t t) const {
    return Buf{gq[t * 64], fp[t * 64], dp[t * 64]};
  }
  __device__ __forceinline__ void get(const Buf& b, uint64_t t, double& rho, double& d) {
    // calc_HWE (gen_func.cpp:938-957) for F = 0 and F = 1; with F = 1 the heterozygote
    // weight is exp(-1e15) = 0
    const double maf = b.f, om = 1 - maf;
    const double bb = om * maf;
    const double h00 = om * om, h02 = maf * maf;
    double p0, p1, p2;
    glq_decode(b.q, p0, p1, p2);
    const double e0 = fma(p0, h00, fma(p1, 2 * bb, p2 * h02));
    const double e1 = fma(p0, h00 + bb, p2 * (h02 + bb));
    base.mul(e0);
    rho = emission_ratio(e0, e1);
    d = b.d;
    eo[t * 64] = rho;  // sites past T never get here
  }
  __device__ __forceinline__ void rescale() { base.rescale(); }
};

// Fresh walk of a PACKED handle (called genotypes, glview.hpp): the cell is a 2-bit code, 16
// consecutive sites of a lane share one 32-bit word, and the linear likelihoods are exactly
// (1,0,0), (0,1,0), (0,0,1) or (u,u,u): the emission of SrcFresh::get collapses to a select
// among four per-site values.  0.25 B read instead of 16 B per site and individual.
struct SrcFreshPacked {
  const uint32_t* __restrict__ gw;
  const double* __restrict__ fp;
  const double* __restrict__ dp;
  double* __restrict__ eo;
  BaseAcc base;
  struct Buf {
    uint32_t w;
    double f, d;
  };
  __device__ __forceinline__ SrcFreshPacked(const LklArrays& A, uint64_t wave_base,
                                            uint64_t pos_base)
      : gw(A.geno_il + (((wave_base & ~63ull) >> 4) + (wave_base & 63))), fp(A.freq_il + pos_base),
        dp(A.pos_il + pos_base), eo(A.e_out + wave_base) {}
  __device__ __forceinline__ Buf load(uint64_t t) const {
    return Buf{gw[(t >> 4) * 64], fp[t * 64], dp[t * 64]};
  }
  __device__ __forceinline__ void get(const Buf& b, uint64_t t, double& rho, double& d) {
    const uint32_t code = (b.w >> ((uint32_t)(t & 15) * 2)) & 3u;
    const double maf = b.f, om = 1 - maf;
    const double bb = om * maf;
    const double h00 = om * om, h02 = maf * maf;
    // the four classes through SrcFresh's formula on what glq_encode makes of them: p in
    // {0, 1} (exact), and (1, 1, 1) for a uniform cell, whose common factor u is part of
    // gl_scale_c like any cell's largest likelihood -- the same bits as the dense handle
    const double u0 = fma(1.0, h00, fma(1.0, 2 * bb, h02));
    const double u1 = fma(1.0, h00 + bb, h02 + bb);
    const double e0 = code == 0 ? h00 : code == 1 ? 2 * bb : code == 2 ? h02 : u0;
    const double e1 = code == 0 ? h00 + bb : code == 1 ? 0.0 : code == 2 ? h02 + bb : u1;
    base.mul(e0);
    rho = emission_ratio(e0, e1);
    d = b.d;
    eo[t * 64] = rho;
  }
  __device__ __forceinline__ void rescale() { base.rescale(); }
};

// which per-site source a forward walk reads: the materialised emissions, or (first walk
// after a frequency update) the dense / packed likelihoods
enum { SRC_PLAIN = 0, SRC_FRESH = 1, SRC_FRESH_PACKED = 2 };
template <int SRC>
using SrcOf = std::conditional_t<SRC == SRC_PLAIN, SrcPlain,
                                 std::conditional_t<SRC == SRC_FRESH, SrcFresh, SrcFreshPacked>>;


// sum of base_c[0..C) over a wave: lane l adds the K = ceil(C / 64) entries l*K .. l*K + K - 1,
// then the fixed butterfly (C <= 64: one entry per lane)
__device__ __forceinline__ double base_sum(const double* __restrict__ base_c, uint32_t C, int lane) {
  const uint32_t K = (C + 63) / 64;
  double acc = 0.0;
  for (uint32_t u = 0; u < K; ++u) {
    const uint32_t k = (uint32_t)lane * K + u;
    if (k < C) acc += base_c[k];
  }
  return wave_sum(acc);
}

// ---- the end of an objective round: lkl = log( q . prod_c R_c . 1 ) ----
// The ordered product of one point's C chunk operators (part_g: the group's [C][MAXP][5]) by a
// wave: lane l multiplies the operators of chunks l*K .. l*K + K - 1 (K = ceil(C / 64): one chunk
// per lane up to 64 chunks), an ordered shuffle tree the lanes'; lane 0 holds the result.
// Shared by k_fast_lkl_finish and the planning kernel (kernels_bfgs.hip), which finishes its own
// individual's points: the same operations in the same order, the same bits.
__device__ __forceinline__ Op lkl_point_product(const double* __restrict__ part_g, uint32_t C, uint32_t p,
                                                int lane) {
  const uint32_t K = (C + 63) / 64;
  Op m{1.0, 0.0, 0.0, 1.0, 0};
  if ((uint32_t)lane * K < C) m = op_load(part_g + (((uint64_t)(uint32_t)lane * K) * MAXP + p) * 5);
  for (uint32_t u = 1; u < K; ++u) {
    const uint32_t k = (uint32_t)lane * K + u;
    if (k < C) m = op_mul(m, op_load(part_g + ((uint64_t)k * MAXP + p) * 5));
  }
  for (int off = 1; off < 64; off <<= 1) {
    const Op o = op_shfl_down(m, off);
    if ((lane & (2 * off - 1)) == 0) m = op_mul(m, o);
  }
  return m;
}

// base = the sum of log e0 over the individual's sites (the same for every point)
__device__ __forceinline__ double lkl_point_value(const Op& m, double F, double base) {
  const double q0 = 1 - F, q1 = F;
  const double v0 = fma(q0, m.a00, q1 * m.a10), v1 = fma(q0, m.a01, q1 * m.a11);
  return base + (log(v0 + v1) + (double)m.ex * 0.6931471805599453094);
}

template <typename T>
bool dalloc(T** p, size_t n) {
  if (n == 0) n = 1;
  return hipMalloc((void**)p, n * sizeof(T)) == hipSuccess;
}

inline LklArrays lkl_arrays(const FastState& fs) {
  return LklArrays{fs.e_il, fs.pos_il, reinterpret_cast<const double2*>(fs.glq_il), fs.freq_il,
                   fs.e_il,  fs.geno_il, fs.u_lin, fs.base_c, fs.gl_scale_c,
                   reinterpret_cast<const double2*>(fs.chunk_scale)};
}

}  // namespace

}  // namespace nghmm

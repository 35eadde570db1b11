// kernels_fast.hip -- fast-mode HIP kernels for gfx950: the throughput path.
//
// Same model as the reference (two-state HMM with distance-dependent transitions,
// shared/HMM.cpp), reformulated for the hardware:
//
//  * LINEAR space with rescaling instead of log space.  The reference spends 14
//    exp/log calls per site per recursion (HMM.cpp:130-139, gen_func.cpp:135-151);
//    here a site costs one exp (e^{-alpha d}) and ~8 FMAs per tracked vector.
//    Per-call results agree with the log-space oracle to ~1e-13 relative (the
//    tests allow 1e-9); the running magnitude is kept in an integer exponent.
//
//  * CHUNK-PARALLEL over sites.  A 2-state forward step is a 2x2 linear operator
//      M_s = (c_s I + (1-c_s) 1 q^T) diag(e_s),   c_s = exp(-alpha d_s),
//    so a run of sites is the product of its operators.  An individual's sites are
//    cut into J = 64*C runs of T sites; LANE j of the individual's C waves walks
//    run j sequentially and the 64*C per-lane operators are combined afterwards
//    (in-wave ordered shuffle tree, then a tiny cross-wave pass).  No lane ever
//    waits for another inside the main loop.
//
//  * INTERLEAVED layout [I][C][T][64]: element (t, lane) of wave (i, c) is site
//    (c*64 + lane)*T + t.  Every load of the main loops is one contiguous
//    1 KiB (double2 emissions) or 512 B segment per wave-instruction.
//
//  * MATERIALISED emission RATIOS (8 B per site-individual) rather than recomputing the
//    emissions from the 24 B genotype likelihoods in every pass of an iteration.  Scaling both
//    emissions of a site by a common factor changes neither posteriors nor paths and adds
//    the factor's log to the likelihood, whatever (indF, alpha) are: with
//      rho_s = e1_s / e0_s,   base = sum_s log e0_s
//    every walk runs on the emissions (1, rho_s) and the likelihood is base + the walk's
//    result.  rho and base are refreshed by the first forward walk after a frequency
//    update, on its way; base is one number per wave (individual, chunk).
//
//  * the <= 5 probe points of one individual's finite-difference gradient
//    (shared/bfgs.cpp:22-43) share ONE pass over that individual's emissions, in a
//    kernel specialised for the pattern of the points.
//
//  * ONE forward walk per EM iteration serves the M-step's first objective round, the
//    E-step (it leaves lane operators and a checkpoint every 8 sites; the backward
//    sweep recomputes forward vectors block-wise) and the emission refresh.
//
//  * posteriors in a TILE-MAJOR layout ([tile row c*T + t][i / 8][l][i % 8]: kPost8 below) the
//    backward sweep writes with contiguous stores; est_maf reads it in place.
//
// The streaming kernels are HBM-bound, the objective rounds and est_maf FP64-issue
// bound (no MFMA: there is no contraction longer than 2).  DESIGN.md section 4.
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
// NGHMM_POST8=0: the earlier [tile row][i][l] (a sector = 8 sites of one individual, which
// est_maf shares between neighbouring workgroups through L2 as far as L2 keeps it).
#ifndef NGHMM_POST8
#define NGHMM_POST8 1
#endif
constexpr bool kPost8 = NGHMM_POST8 != 0;
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

// exp(y) for -2^-6 <= y <= 0 without range reduction (y^8/8! < 1e-19)
__device__ __forceinline__ double exp_tiny7(double y) {
  double p = 1.0 / 5040.0;
  p = fma(p, y, 1.0 / 720.0);
  p = fma(p, y, 1.0 / 120.0);
  p = fma(p, y, 1.0 / 24.0);
  p = fma(p, y, 1.0 / 6.0);
  p = fma(p, y, 0.5);
  p = fma(p, y, 1.0);
  return fma(p, y, 1.0);
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

// The main loop of one wave for the finite-difference pattern.  Per site and lane:
// one exp (a degree-7 polynomial when alpha * d_max <= 2^-6: SMALL), the products shared
// by all points, 12 instructions per F-probe and 20 per alpha-probe; one exponent (point
// 0's) rescales all points, which are perturbations of each other.
template <int NF, int NA, bool SMALL, bool EMIT, int XDEG, bool OWNEX, typename Src>
__device__ __forceinline__ void lkl_run_fd(Src& src, uint64_t T, const GroupDesc& G,
                                           Op (&R)[MAXP], EmitPtrs emit, uint64_t wave,
                                           int lane) {
  static_assert(NB * UG == CK && RENORM == CK, "checkpoints are stored right after a rescale");
  constexpr int NPT = 1 + NF + NA;
  const uint64_t nblk = T / CK;
  const double al0 = G.A[0];
  const double q1 = G.F[0], q0 = 1 - q1;
  double rho0[NF > 0 ? NF : 1], rho1[NF > 0 ? NF : 1], dal[NA > 0 ? NA : 1];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    rho0[f] = (1 - G.F[1 + f]) / q0;
    rho1[f] = G.F[1 + f] / q1;
  }
#pragma unroll
  for (int a = 0; a < NA; ++a) dal[a] = al0 - G.A[1 + NF + a];
  int exc = 0;
  // Software pipeline: NB buffers of UG sites.  A buffer is refilled right after it
  // has been consumed, i.e. (NB-1) groups = 6 sites before it is needed again, which
  // covers HBM latency at two to three waves per SIMD.  T is a multiple of NB*UG and the
  // arrays carry one group of slack at the end, so neither the prologue nor the refills
  // need bound checks (values read past T are never used); sites past S are identity
  // operators (e = 1, d = 0).
  typename Src::Buf buf[NB][UG];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int u = 0; u < UG; ++u) buf[b][u] = src.load((uint64_t)b * UG + u);
  }
  for (uint64_t t0 = 0; t0 < T; t0 += NB * UG) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        double rho, d;
        src.get(buf[b][u], t0 + (uint64_t)b * UG + u, rho, d);
        double c0;
        if constexpr (SMALL) {
          // chromosome starts are stored as d = 1e30: c = 0 there (the polynomial of the
          // huge argument is finite; masking its bits keeps the loop body branch-free)
          const uint64_t keep = (d < 1e30) ? ~0ull : 0ull;
          c0 = ngh_from_bits(ngh_bits(exp_tiny7(-al0 * d)) & keep);
        } else {
          c0 = coanc(al0, d);
        }
        const double a0 = 1 - c0;
        const double ce0 = c0, ce1 = c0 * rho;  // emissions (1, rho)
        const double eq0 = q0, eq1 = rho * q1;
        const double g0 = a0 * eq0, g1 = a0 * eq1;
        op_step(R[0], ce0, ce1, g0, g1);
#pragma unroll
        for (int f = 0; f < NF; ++f) op_step(R[1 + f], ce0, ce1, g0 * rho0[f], g1 * rho1[f]);
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          // |x| <= 1e-3 on every finite distance (checked by the host); at d = 1e30 the
          // polynomial is huge but finite and multiplies c0 = 0
          const double m = exp_small<XDEG>(dal[a] * d);
          const double am = fma(-c0, m, 1.0);
          op_step(R[1 + NF + a], ce0 * m, ce1 * m, am * eq0, am * eq1);
        }
      }
#pragma unroll
      for (int u = 0; u < UG; ++u) buf[b][u] = src.load(t0 + (uint64_t)(b + NB) * UG + u);
    }
    if constexpr (OWNEX) {  // every point by its own exponent (kept in R[p].ex)
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        const double mx = fmax(fmax(R[p].a00, R[p].a01), fmax(R[p].a10, R[p].a11));
        const int e = exp_of(mx);
        const double sc = __builtin_ldexp(1.0, -e);
        R[p].ex += e;
        R[p].a00 *= sc;
        R[p].a01 *= sc;
        R[p].a10 *= sc;
        R[p].a11 *= sc;
      }
      src.rescale();
    } else {  // rescale every point by point 0's exponent
      const double mx = fmax(fmax(R[0].a00, R[0].a01), fmax(R[0].a10, R[0].a11));
      const int e = exp_of(mx);
      const double sc = __builtin_ldexp(1.0, -e);
      exc += e;
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        R[p].a00 *= sc;
        R[p].a01 *= sc;
        R[p].a10 *= sc;
        R[p].a11 *= sc;
      }
      src.rescale();
    }
    if constexpr (EMIT) {  // first round of an M-step: point 0 is the E-step's forward walk
      // (no bound check, to keep the loop one basic block: the store after the last block
      // lands in the unused slot 0 of the next wave, or in the array's slack)
      emit_checkpoint(emit.ckpt, wave, nblk, t0 / CK + 1, lane, R[0]);
    }
  }
  if constexpr (!OWNEX) {
#pragma unroll
    for (int p = 0; p < NPT; ++p) R[p].ex = exc;
  }
}

// ordered product of the 64 lanes' operators; lane 0 stores the wave's operator
__device__ __forceinline__ void lkl_store_wave_op(Op r, int lane, double* __restrict__ out) {
  renorm(r);
  Op m = r;
  for (int off = 1; off < 64; off <<= 1) {
    const Op o = op_shfl_down(m, off);
    if ((lane & (2 * off - 1)) == 0) m = op_mul(m, o);
  }
  if (lane == 0) {
    out[0] = m.a00;
    out[1] = m.a01;
    out[2] = m.a10;
    out[3] = m.a11;
    out[4] = (double)m.ex;
  }
}

// One kernel per loop-body version (each gets its own register allocation); the host
// sorts the groups of a round by mode and launches every version on its range
// [g_begin, g_begin + gridDim.x / C).
template <int NF, int NA, bool SMALL, bool EMIT, int SRC, int XDEG, bool OWNEX = false>
__global__ void __launch_bounds__(64)
k_fast_lkl_fd(LklArrays arr, uint64_t T, uint32_t C, const GroupDesc* __restrict__ groups,
              uint32_t g_begin, double* __restrict__ part, EmitPtrs emit) {
  static_assert(EMIT || SRC == SRC_PLAIN, "the fresh walk is the first round of an M-step");
  // chunk-major: the waves resident at a time walk the same few slices of the shared
  // distance / frequency tables, which then stay in L2
  const uint32_t n_g = gridDim.x / C;
  const uint32_t g = g_begin + blockIdx.x % n_g;
  const uint32_t c = blockIdx.x / n_g;
  const int lane = threadIdx.x;
  const GroupDesc& G = groups[g];
  const uint64_t i = G.ind;
  Op R[MAXP];
#pragma unroll
  for (int p = 0; p < MAXP; ++p) R[p] = Op{1.0, 0.0, 0.0, 1.0, 0};
  const uint64_t wave_base = ((i * C + c) * T) * 64 + lane;
  const uint64_t pos_base = ((uint64_t)c * T) * 64 + lane;
  using Src = SrcOf<SRC>;
  Src src(arr, wave_base, pos_base);
  static_assert(!(OWNEX && EMIT), "an emitting round's checkpoints assume point 0's scale");
  lkl_run_fd<NF, NA, SMALL, EMIT, XDEG, OWNEX>(src, T, G, R, emit, i * C + c, lane);
  if constexpr (SRC != SRC_PLAIN) {  // fresh walk: the wave's part of sum log e0
    const double bl = wave_sum(src.base.log_value());
    if (lane == 0) arr.base_c[i * C + c] = bl + (arr.gl_scale_c ? arr.gl_scale_c[i * C + c] : 0.0);
  }
  if constexpr (EMIT) {
    Op r0 = R[0];
    renorm(r0);
    emit_lane_op(emit.lane_ops, i * C + c, lane, r0);
  }
#pragma unroll
  for (int p = 0; p < 1 + NF + NA; ++p)
    lkl_store_wave_op(R[p], lane, part + (((uint64_t)g * C + c) * MAXP + p) * 5);
}

template <int NP_MAX, int SRC>
__global__ void __launch_bounds__(64)
k_fast_lkl_chunks(LklArrays arr, uint64_t T, uint32_t C, const GroupDesc* __restrict__ groups,
                  uint32_t g_begin, double* __restrict__ part, EmitPtrs emit) {
  // chunk-major: the waves resident at a time walk the same few slices of the shared
  // distance / frequency tables, which then stay in L2
  const uint32_t n_g = gridDim.x / C;
  const uint32_t g = g_begin + blockIdx.x % n_g;
  const uint32_t c = blockIdx.x / n_g;
  const int lane = threadIdx.x;
  const GroupDesc& G = groups[g];
  const uint32_t np = G.np;
  const uint64_t i = G.ind;

  Op R[NP_MAX];
  double q0[NP_MAX], q1[NP_MAX], al[NP_MAX];
#pragma unroll
  for (int p = 0; p < NP_MAX; ++p) {
    const double f = (p < (int)np) ? G.F[p] : 0.5;
    q1[p] = f;
    q0[p] = 1 - f;
    al[p] = (p < (int)np) ? G.A[p] : 1.0;
    R[p] = Op{1.0, 0.0, 0.0, 1.0, 0};
  }

  using Src = SrcOf<SRC>;
  Src src(arr, ((i * C + c) * T) * 64 + lane, ((uint64_t)c * T) * 64 + lane);
  typename Src::Buf buf[NB][UG];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int u = 0; u < UG; ++u) buf[b][u] = src.load((uint64_t)b * UG + u);
  }
  for (uint64_t t0 = 0; t0 < T; t0 += NB * UG) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        double rho, d;
        src.get(buf[b][u], t0 + (uint64_t)b * UG + u, rho, d);
#pragma unroll
        for (int p = 0; p < NP_MAX; ++p) {
          if (p < (int)np) {
            const double cc = coanc(al[p], d);
            const double a = 1 - cc;
            op_step(R[p], cc, cc * rho, a * q0[p], a * rho * q1[p]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UG; ++u) buf[b][u] = src.load(t0 + (uint64_t)(b + NB) * UG + u);
    }
#pragma unroll
    for (int p = 0; p < NP_MAX; ++p)
      if (p < (int)np) renorm(R[p]);
    src.rescale();
    if (emit.ckpt) {
      const uint64_t b = t0 / CK + 1;
      if (b < T / CK) emit_checkpoint(emit.ckpt, i * C + c, T / CK, b, lane, R[0]);
    }
  }
  if constexpr (SRC != SRC_PLAIN) {
    const double bl = wave_sum(src.base.log_value());
    if (lane == 0) arr.base_c[i * C + c] = bl + (arr.gl_scale_c ? arr.gl_scale_c[i * C + c] : 0.0);
  }
  if (emit.lane_ops) {
    Op r0 = R[0];
    renorm(r0);
    emit_lane_op(emit.lane_ops, i * C + c, lane, r0);
  }
#pragma unroll
  for (int p = 0; p < NP_MAX; ++p)
    if (p < (int)np) lkl_store_wave_op(R[p], lane, part + (((uint64_t)g * C + c) * MAXP + p) * 5);
}

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

// lkl = log( q . prod_c R_c . 1 ): one workgroup per group, one wave per point; lane l holds the
// ordered product of the operators of chunks l*K .. l*K + K - 1 of the wave's point (K =
// ceil(C / 64): one chunk per lane up to 64 chunks) and an ordered shuffle tree multiplies them.
// SHARD (a handle that holds a site range of a larger data set, SiteShard): the product and
// the sum of log e0 of this range go to the send buffer instead, six doubles per point in point
// order; k_fast_shard_combine finishes the value once every range's part has arrived.
template <bool SHARD>
__global__ void __launch_bounds__(64 * MAXP)
k_fast_lkl_finish(const GroupDesc* __restrict__ groups, uint32_t n_groups, uint32_t C,
                  const double* __restrict__ part, const double* __restrict__ base_c,
                  double* __restrict__ lkl_out, int* __restrict__ flags) {
  const uint32_t g = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const GroupDesc& G = groups[g];
  const uint32_t p = threadIdx.x >> 6;
  if (p >= G.np) return;
  // sum of log e0 over the individual's sites: the same for every point
  const double base = base_sum(base_c + (uint64_t)G.ind * C, C, lane);
  {
    const uint32_t K = (C + 63) / 64;
    Op m{1.0, 0.0, 0.0, 1.0, 0};
    if ((uint32_t)lane * K < C) m = op_load(part + (((uint64_t)g * C + (uint32_t)lane * K) * MAXP + p) * 5);
    for (uint32_t u = 1; u < K; ++u) {
      const uint32_t k = (uint32_t)lane * K + u;
      if (k < C) m = op_mul(m, op_load(part + (((uint64_t)g * C + k) * MAXP + p) * 5));
    }
    for (int off = 1; off < 64; off <<= 1) {
      const Op o = op_shfl_down(m, off);
      if ((lane & (2 * off - 1)) == 0) m = op_mul(m, o);
    }
    if (lane == 0) {
      if constexpr (SHARD) {
        double* o = lkl_out + (uint64_t)G.out_idx[p] * 6;
        o[0] = m.a00;
        o[1] = m.a01;
        o[2] = m.a10;
        o[3] = m.a11;
        o[4] = (double)m.ex;
        o[5] = base;
      } else {
        const double q0 = 1 - G.F[p], q1 = G.F[p];
        const double v0 = fma(q0, m.a00, q1 * m.a10), v1 = fma(q0, m.a01, q1 * m.a11);
        const double l = base + (log(v0 + v1) + (double)m.ex * 0.6931471805599453094);
        lkl_out[G.out_idx[p]] = l;
        // NaN or +-inf: overflow of a probe against point 0's scale, or no probability mass left
        // in linear space; the host re-evaluates such points with the general kernel
        if (!(fabs(l) < __builtin_huge_val())) flags[FLAG_INVALID_LKL] = 1;
      }
    }
  }
}

// ---- site shards ------------------------------------------------------------
// A run of sites is the product of its operators, so the SITE axis can be cut between GPUs as
// it is cut between lane-chunks: every handle holds all individuals for a contiguous site
// range, walks it as if it were a data set of its own, and what the ranges owe each other per
// individual is one 2x2 operator (+ exponent, + the range's sum of log e0): six doubles.  They
// travel by an all-gather the caller provides (SiteShard::allgather, stream-ordered); every
// handle then multiplies the ranges' operators in rank order, so all of them see the same
// bits and run the same L-BFGS-B steps.  est_maf has every individual of its sites at hand:
// the frequency step needs no exchange at all.
//
// recv = [world][n][6]; one thread per point: lkl = sum_r base_r + log(q . prod_r M_r . 1)
__global__ void __launch_bounds__(256)
k_fast_shard_combine(const GroupDesc* __restrict__ groups, uint32_t n_groups,
                     const double* __restrict__ recv, uint32_t world, uint64_t n,
                     double* __restrict__ lkl_out, int* __restrict__ flags) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t g = (uint32_t)(t / MAXP), p = (uint32_t)(t % MAXP);
  if (g >= n_groups) return;
  const GroupDesc& G = groups[g];
  if (p >= G.np) return;
  const uint64_t idx = G.out_idx[p];
  Op m = op_load(recv + idx * 6);
  double base = recv[idx * 6 + 5];
  for (uint32_t r = 1; r < world; ++r) {
    const double* o = recv + ((uint64_t)r * n + idx) * 6;
    m = op_mul(m, op_load(o));
    base += o[5];
  }
  const double q0 = 1 - G.F[p], q1 = G.F[p];
  const double v0 = fma(q0, m.a00, q1 * m.a10), v1 = fma(q0, m.a01, q1 * m.a11);
  const double l = base + (log(v0 + v1) + (double)m.ex * 0.6931471805599453094);
  lkl_out[idx] = l;
  if (!(fabs(l) < __builtin_huge_val())) flags[FLAG_INVALID_LKL] = 1;
}

// E-step: the operator of the handle's whole site range per individual (ordered product of its
// lane-chunk operators) and the range's sum of log e0 -> send[i][6]; one wave per individual
__global__ void __launch_bounds__(64)
k_fast_shard_reduce(const double* __restrict__ lane_ops, uint64_t J, uint32_t C,
                    const double* __restrict__ base_c, double* __restrict__ send) {
  const uint64_t i = blockIdx.x;
  const int lane = threadIdx.x;
  const double* ops = lane_ops + (i * J + (uint64_t)lane * C) * 5;
  constexpr uint32_t PF = 8;
  Op L{1.0, 0.0, 0.0, 1.0, 0};
  for (uint32_t k0 = 0; k0 < C; k0 += PF) {
    Op o[PF];
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u)
      o[u] = op_load(ops + (uint64_t)(k0 + u < C ? k0 + u : C - 1) * 5);
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u)
      if (k0 + u < C) L = op_mul(L, o[u]);
  }
  for (int off = 1; off < 64; off <<= 1) {
    const Op o = op_shfl_down(L, off);
    if ((lane & (2 * off - 1)) == 0) L = op_mul(L, o);
  }
  const double base = base_sum(base_c + i * C, C, lane);
  if (lane == 0) {
    double* o = send + i * 6;
    o[0] = L.a00;
    o[1] = L.a01;
    o[2] = L.a10;
    o[3] = L.a11;
    o[4] = (double)L.ex;
    o[5] = base;
  }
}

// The first objective round of an M-step carries every individual's current parameters as its
// point 0: that point's gathered operators ARE the ranges' operators the E-step needs, so the
// E-step's own all-gather is saved (stride 6 doubles per point, n points per rank)
__global__ void __launch_bounds__(256)
k_fast_shard_edges_from_round(const GroupDesc* __restrict__ groups, uint32_t n_groups,
                              const double* __restrict__ recv, uint32_t world, uint32_t rank,
                              uint64_t n, double* __restrict__ edges) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const GroupDesc& G = groups[g];
  const uint64_t idx = G.out_idx[0];
  const double f = G.F[0];
  double u0 = 1 - f, u1 = f;
  int uex = 0;
  double base = 0.0;
  for (uint32_t r = 0; r < world; ++r) {
    const double* o = recv + ((uint64_t)r * n + idx) * 6;
    base += o[5];
    if (r < rank) {
      const Op m = op_load(o);
      const double n0 = fma(u0, m.a00, u1 * m.a10), n1 = fma(u0, m.a01, u1 * m.a11);
      u0 = n0;
      u1 = n1;
      uex += m.ex;
      renorm2(u0, u1, uex);
    }
  }
  double x0 = 1.0, x1 = 1.0;
  int xex = 0;
  for (uint32_t r = world; r-- > rank + 1;) {
    const Op m = op_load(recv + ((uint64_t)r * n + idx) * 6);
    const double n0 = fma(m.a00, x0, m.a01 * x1), n1 = fma(m.a10, x0, m.a11 * x1);
    x0 = n0;
    x1 = n1;
    xex += m.ex;
    renorm2(x0, x1, xex);
  }
  double* e = edges + (uint64_t)G.ind * 8;
  e[0] = u0;
  e[1] = u1;
  e[2] = (double)uex;
  e[3] = x0;
  e[4] = x1;
  e[5] = (double)xex;
  e[6] = base;
  e[7] = 0.0;
}

// edges[i][8] = the row vector entering this range from the left (u0, u1, exponent), the column
// vector entering it from the right (x0, x1, exponent), the sum of log e0 over all ranges
__global__ void __launch_bounds__(256)
k_fast_shard_edges(const double* __restrict__ recv, uint32_t world, uint32_t rank, uint64_t I,
                   const double* __restrict__ indF, double* __restrict__ edges) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= I) return;
  const double f = indF[i];
  double u0 = 1 - f, u1 = f;
  int uex = 0;
  double base = 0.0;
  for (uint32_t r = 0; r < world; ++r) {
    const double* o = recv + ((uint64_t)r * I + i) * 6;
    base += o[5];
    if (r < rank) {
      const Op m = op_load(o);
      const double n0 = fma(u0, m.a00, u1 * m.a10), n1 = fma(u0, m.a01, u1 * m.a11);
      u0 = n0;
      u1 = n1;
      uex += m.ex;
      renorm2(u0, u1, uex);
    }
  }
  double x0 = 1.0, x1 = 1.0;
  int xex = 0;
  for (uint32_t r = world; r-- > rank + 1;) {
    const Op m = op_load(recv + ((uint64_t)r * I + i) * 6);
    const double n0 = fma(m.a00, x0, m.a01 * x1), n1 = fma(m.a10, x0, m.a11 * x1);
    x0 = n0;
    x1 = n1;
    xex += m.ex;
    renorm2(x0, x1, xex);
  }
  double* e = edges + i * 8;
  e[0] = u0;
  e[1] = u1;
  e[2] = (double)uex;
  e[3] = x0;
  e[4] = x1;
  e[5] = (double)xex;
  e[6] = base;
  e[7] = 0.0;
}

// ---- E-step ---------------------------------------------------------------
// The E-step of a lane-chunk needs the forward vector at every site and the backward
// vector at every site.  Storing either per site costs 8 B written and 8 B read back per
// site and individual; instead the forward pass leaves a CHECKPOINT every CK sites -- the
// 2x2 prefix operator of the lane-chunk up to there, 32 B per CK sites -- and the backward
// sweep recomputes the forward vectors of a block of CK sites from its checkpoint, in
// registers, before walking the block backwards (k_fast_bwd_recompute).  The prefix
// operators do not depend on the vector entering the lane-chunk, so they are produced by
// whichever kernel walks the chunk forward first: k_fast_chunk_ops for a stand-alone
// E-step, or the first objective round of the M-step (same parameters, same emissions:
// lkl_run_fd / k_fast_lkl_chunks with `emit`), which then replaces phase A altogether.
//
// checkpoint layout: ck[((i*C + c)*NBLK + b)*2 + h][64] of double2 = row h of the prefix
// operator of sites [0, b*CK) of lane-chunk (c, lane); b = 0 (identity) is not stored.
// phase A: the operator of every lane-chunk and its checkpoints, one point per individual
__global__ void __launch_bounds__(64)
k_fast_chunk_ops(const double* __restrict__ e_il, const double* __restrict__ pos_il, uint64_t T,
                 uint32_t C, const double* __restrict__ indF, const double* __restrict__ alpha,
                 EmitPtrs out) {
  const uint64_t i = blockIdx.x / C;
  const uint32_t c = blockIdx.x % C;
  const int lane = threadIdx.x;
  const double f = indF[i], al = alpha[i];
  const double q0 = 1 - f, q1 = f;
  Op R{1.0, 0.0, 0.0, 1.0, 0};
  const double* ep = e_il + ((i * C + c) * T) * 64 + lane;
  const double* dp = pos_il + ((uint64_t)c * T) * 64 + lane;
  const uint64_t nblk = T / CK;
  double ecur[UF], enxt[UF];  // emission ratios: the emissions are (1, rho)
  double dcur[UF], dnxt[UF];
#pragma unroll
  for (int u = 0; u < UF; ++u) {
    ecur[u] = ep[(uint64_t)u * 64];  // T is a multiple of 8; arrays carry a group of slack
    dcur[u] = dp[(uint64_t)u * 64];
  }
  for (uint64_t t0 = 0; t0 < T; t0 += UF) {
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const uint64_t t = t0 + UF + u;
      enxt[u] = ep[t * 64];
      dnxt[u] = dp[t * 64];
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const double cc = coanc(al, dcur[u]);
      const double a = 1 - cc;
      op_step(R, cc, cc * ecur[u], a * q0, a * ecur[u] * q1);
    }
    if (((t0 / UF) % (RENORM / UF)) == (RENORM / UF - 1)) {
      renorm(R);
      const uint64_t b = (t0 + UF) / CK;  // checkpoint in front of block b
      if (b < nblk) emit_checkpoint(out.ckpt, i * C + c, nblk, b, lane, R);
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      ecur[u] = enxt[u];
      dcur[u] = dnxt[u];
    }
  }
  renorm(R);
  emit_lane_op(out.lane_ops, i * C + c, lane, R);
}

// phase B: per individual, the vector entering every lane-chunk from the left
// (forward) and from the right (backward), the log-likelihood and the Fw/Bw check.
// One wave per individual: lane l owns the C consecutive lane-chunks l*C .. l*C + C-1,
// multiplies their operators, an ordered shuffle scan over the 64 lanes gives every lane
// the product of everything to its left (right), and the lane then walks its own chunks.
__device__ __forceinline__ Op op_shfl_up(const Op& m, int off) {
  Op o;
  o.a00 = __shfl_up(m.a00, off);
  o.a01 = __shfl_up(m.a01, off);
  o.a10 = __shfl_up(m.a10, off);
  o.a11 = __shfl_up(m.a11, off);
  o.ex = __shfl_up(m.ex, off);
  return o;
}

__global__ void __launch_bounds__(64)
k_fast_bounds(const double* __restrict__ lane_ops, uint64_t J, uint32_t C,
              const double* __restrict__ indF, const double* __restrict__ base_c,
              double* __restrict__ bound, double* __restrict__ ind_lkl, int* __restrict__ flags,
              const double* __restrict__ edges) {
  const uint64_t i = blockIdx.x;
  const int lane = threadIdx.x;
  const double f = indF[i];
  // what enters the handle's sites from the left and from the right: the initial distribution
  // and (1, 1), or -- a site shard -- the other ranges' products (k_fast_shard_edges)
  double q0 = 1 - f, q1 = f, x0 = 1.0, x1 = 1.0;
  int uex = 0, xex = 0;
  if (edges) {
    const double* e = edges + i * 8;
    q0 = e[0];
    q1 = e[1];
    uex = (int)e[2];
    x0 = e[3];
    x1 = e[4];
    xex = (int)e[5];
  }
  const double LN2 = 0.6931471805599453094;
  const double* ops = lane_ops + (i * J + (uint64_t)lane * C) * 5;
  double* bd = bound + (i * J + (uint64_t)lane * C) * 4;

  // A lane's C operators are read three times below, each time as a chain of dependent
  // multiplications: the loads of PF operators are issued together ahead of their chain link,
  // or every link would wait a memory round trip (100 us per call at C = 51, whatever I is).
  constexpr uint32_t PF = 8;
  Op L{1.0, 0.0, 0.0, 1.0, 0};
  for (uint32_t k0 = 0; k0 < C; k0 += PF) {
    Op o[PF];
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u)
      o[u] = op_load(ops + (uint64_t)(k0 + u < C ? k0 + u : C - 1) * 5);
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u)
      if (k0 + u < C) L = op_mul(L, o[u]);
  }

  // forward: product of the lanes to the left
  Op P = L;
  for (int off = 1; off < 64; off <<= 1) {
    const Op o = op_shfl_up(P, off);
    if (lane >= off) P = op_mul(o, P);
  }
  Op E = op_shfl_up(P, 1);
  if (lane == 0) E = Op{1.0, 0.0, 0.0, 1.0, 0};
  double v0 = fma(q0, E.a00, q1 * E.a10), v1 = fma(q0, E.a01, q1 * E.a11);
  int ex = E.ex + uex;
  renorm2(v0, v1, ex);
  for (uint32_t k0 = 0; k0 < C; k0 += PF) {
    Op o[PF];
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u)
      o[u] = op_load(ops + (uint64_t)(k0 + u < C ? k0 + u : C - 1) * 5);
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u) {
      const uint32_t k = k0 + u;
      if (k < C) {
        bd[(uint64_t)k * 4 + 0] = v0;
        bd[(uint64_t)k * 4 + 1] = v1;
        const double n0 = fma(v0, o[u].a00, v1 * o[u].a10);
        const double n1 = fma(v0, o[u].a01, v1 * o[u].a11);
        v0 = n0;
        v1 = n1;
        ex += o[u].ex;
        renorm2(v0, v1, ex);
      }
    }
  }
  // (x = (1, 1): v0 + v1 exactly)
  const double lf = __shfl(log(fma(v0, x0, v1 * x1)) + (double)(ex + xex) * LN2, 63);  // lane 63 has walked it all

  // backward: product of the lanes to the right
  Op Sx = L;
  for (int off = 1; off < 64; off <<= 1) {
    const Op o = op_shfl_down(Sx, off);
    if (lane + off < 64) Sx = op_mul(Sx, o);
  }
  Op X = op_shfl_down(Sx, 1);
  if (lane == 63) X = Op{1.0, 0.0, 0.0, 1.0, 0};
  double w0 = fma(X.a00, x0, X.a01 * x1), w1 = fma(X.a10, x0, X.a11 * x1);
  int exb = X.ex + xex;
  renorm2(w0, w1, exb);
  for (uint32_t kk0 = C; kk0 > 0; kk0 = kk0 > PF ? kk0 - PF : 0) {
    Op o[PF];  // operators kk0-1, kk0-2, ...
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u)
      o[u] = op_load(ops + (uint64_t)(kk0 > u ? kk0 - 1 - u : 0) * 5);
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u) {
      if (kk0 > u) {
        const uint32_t k = kk0 - 1 - u;
        bd[(uint64_t)k * 4 + 2] = w0;
        bd[(uint64_t)k * 4 + 3] = w1;
        const double n0 = fma(o[u].a00, w0, o[u].a01 * w1);
        const double n1 = fma(o[u].a10, w0, o[u].a11 * w1);
        w0 = n0;
        w1 = n1;
        exb += o[u].ex;
        renorm2(w0, w1, exb);
      }
    }
  }
  const double lb = __shfl(log(fma(q0, w0, q1 * w1)) + (double)(exb + uex) * LN2, 0);
  // the walks ran on the emissions (1, rho): add sum log e0 (as k_fast_lkl_finish does)
  const double base = edges ? edges[i * 8 + 6] : base_sum(base_c + i * C, C, lane);
  if (lane == 0) {
    ind_lkl[i] = base + lf;
    if (lf != lf || lb != lb || base != base) flags[FLAG_INVALID_LKL] = 1;
    if (fabs(lf - lb) > 0.001) flags[FLAG_FW_BW] = 1;  // EM.cpp:167
  }
}

// phase C: backward sweep with block-wise forward recomputation.  Posterior of the IBD
// state, snapped like check_interv (gen_func.cpp:55-70).  This version (kPost8 == false)
// writes the TILE-MAJOR layout
//   post[(c*T + t)*I + i][lane]      (site (c*64 + lane)*T + t)
// i.e. every wave-store is one contiguous 512 B segment and no transposition pass is
// needed: est_maf reads the layout directly (k_fast_estmaf<.., TILE>); the version in use,
// k_fast_bwd_recompute8, follows.
__global__ void __launch_bounds__(64)
k_fast_bwd_recompute(const double* __restrict__ e_il, const double* __restrict__ pos_il,
                     uint64_t T, uint32_t C, uint64_t S, uint64_t I,
                     const double* __restrict__ indF, const double* __restrict__ alpha,
                     const double* __restrict__ bound, const double2* __restrict__ ckpt,
                     double* __restrict__ post, int* __restrict__ flags) {
  const uint64_t i = blockIdx.x / C;
  const uint32_t c = blockIdx.x % C;
  const int lane = threadIdx.x;
  const double f = indF[i], al = alpha[i];
  const double q0 = 1 - f, q1 = f;
  const uint64_t J = (uint64_t)C * 64;
  const uint64_t j = (uint64_t)c * 64 + lane;
  const double* bd = bound + (i * J + j) * 4;
  const double vin0 = bd[0], vin1 = bd[1];
  double w0 = bd[2], w1 = bd[3];
  const double* ep = e_il + ((i * C + c) * T) * 64 + lane;
  const double* dp = pos_il + ((uint64_t)c * T) * 64 + lane;
  double* pp = post + ((uint64_t)c * T * I + i) * 64 + lane;  // site step t at pp[t * I * 64]
  const uint64_t nblk = T / CK;
  const double2* ck = ckpt + ((i * C + c) * nblk * 2) * 64 + lane;
  const uint64_t tstride = I * 64;
  bool nanflag = false;
  int exd = 0;

  double ecur[CK], enxt[CK];  // emission ratios: the emissions are (1, rho)
  double dcur[CK], dnxt[CK];
  double2 r0c, r1c, r0n, r1n;  // rows of the prefix operator in front of the block
  {
    const uint64_t b = nblk - 1;
#pragma unroll
    for (int u = 0; u < CK; ++u) {
      ecur[u] = ep[(b * CK + u) * 64];
      dcur[u] = dp[(b * CK + u) * 64];
    }
    r0c = b ? ck[(b * 2) * 64] : double2{1.0, 0.0};
    r1c = b ? ck[(b * 2 + 1) * 64] : double2{0.0, 1.0};
  }
  for (uint64_t b = nblk;;) {
    --b;
    if (b > 0) {  // the block in front: in flight while this one is computed
      const uint64_t bn = b - 1;
#pragma unroll
      for (int u = 0; u < CK; ++u) {
        enxt[u] = ep[(bn * CK + u) * 64];
        dnxt[u] = dp[(bn * CK + u) * 64];
      }
      r0n = bn ? ck[(bn * 2) * 64] : double2{1.0, 0.0};
      r1n = bn ? ck[(bn * 2 + 1) * 64] : double2{0.0, 1.0};
    }
    // forward vectors of the block's sites, from the checkpoint
    double v0 = fma(vin0, r0c.x, vin1 * r1c.x);
    double v1 = fma(vin0, r0c.y, vin1 * r1c.y);
    double f0[CK], f1[CK], cc[CK];
#pragma unroll
    for (int u = 0; u < CK; ++u) {
      cc[u] = coanc(al, dcur[u]);
      const double a = 1 - cc[u];
      const double sm = v0 + v1;
      v0 = fma(a * q0, sm, cc[u] * v0);
      v1 = fma(a * q1, sm, cc[u] * v1) * ecur[u];
      if (u == CK / 2 - 1) {  // the scale of (v0, v1) does not matter: keep it in range
        int dummy = 0;
        renorm2(v0, v1, dummy);
      }
      f0[u] = v0;
      f1[u] = v1;
    }
    // backward through the block: posterior, then the beta step
#pragma unroll
    for (int u = CK - 1; u >= 0; --u) {
      const uint64_t t = b * CK + u;
      const double x0 = f0[u] * w0, x1 = f1[u] * w1;
      double g1 = x1 * rcp_nr2(x0 + x1);  // 0/0 (no probability mass) stays NaN
      if (j * T + t < S) {
        if (g1 != g1) nanflag = true;
        // check_interv (gen_func.cpp:55-70)
        if (g1 < kEPS) g1 = 0;
        else if (g1 > 1 - kEPS) g1 = 1;
        pp[t * tstride] = g1;
      }
      // beta step: w'_k = c u_k + a (q . u),  u = e * w
      const double a = 1 - cc[u];
      const double u0 = w0, u1 = ecur[u] * w1;
      const double sq = a * fma(q0, u0, q1 * u1);
      w0 = fma(cc[u], u0, sq);
      w1 = fma(cc[u], u1, sq);
    }
    renorm2(w0, w1, exd);
    if (b == 0) break;
#pragma unroll
    for (int u = 0; u < CK; ++u) {
      ecur[u] = enxt[u];
      dcur[u] = dnxt[u];
    }
    r0c = r0n;
    r1c = r1n;
  }
  if (nanflag) flags[FLAG_NAN] = 1;
}

// The same sweep for the [tile row][i / 8][l][i % 8] layout (kPost8).  A workgroup of four waves
// is the eight individuals of a group x one half (32) of the lane-chunks of chunk c: a wave
// walks 32 lane-chunks of two individuals (every load two 256 B segments), the posteriors of a
// block of CK sites are staged in LDS and leave as contiguous 2 KB stores.  Four waves, because
// three such workgroups fit a CU at this kernel's three waves per SIMD (eight-wave workgroups
// -- a whole tile row block per store -- fit once: 4.5 instead of 4.1 ms).
__global__ void __launch_bounds__(256)
k_fast_bwd_recompute8(const double* __restrict__ e_il, const double* __restrict__ pos_il,
                     uint64_t T, uint32_t C, uint64_t S, uint64_t I,
                     const double* __restrict__ indF, const double* __restrict__ alpha,
                     const double* __restrict__ bound, const double2* __restrict__ ckpt,
                     double* __restrict__ post, int* __restrict__ flags) {
  // (a partial last group repeats its last individual: the copies land in the layout's padding)
  __shared__ double stage[2][CK][8][40];  // [buffer][site of the block][individual][chunk, padded]
  const uint64_t grp = blockIdx.x / ((uint64_t)C * 2);
  const uint32_t c = (uint32_t)((blockIdx.x >> 1) % C);
  const int half = blockIdx.x & 1;
  const int lp = threadIdx.x & 31;                                    // chunk within the half
  const int m8 = (threadIdx.x >> 6) * 2 + ((threadIdx.x >> 5) & 1);  // individual within the group
  const int lane = half * 32 + lp;                                    // the lane-chunk, 0..63
  const uint64_t i = (grp * 8 + m8 < I) ? grp * 8 + m8 : I - 1;
  const double f = indF[i], al = alpha[i];
  const double q0 = 1 - f, q1 = f;
  const uint64_t J = (uint64_t)C * 64;
  const uint64_t j = (uint64_t)c * 64 + lane;
  const double* bd = bound + (i * J + j) * 4;
  const double vin0 = bd[0], vin1 = bd[1];
  double w0 = bd[2], w1 = bd[3];
  const double* ep = e_il + ((i * C + c) * T) * 64 + lane;
  const double* dp = pos_il + ((uint64_t)c * T) * 64 + lane;
  // the half's 32 x 8 block of tile row c*T + t: 256 contiguous doubles, thread k's at [k]
  double* pp = post + (uint64_t)c * T * post_tile_doubles(I) + grp * 512 + half * 256 + threadIdx.x;
  const uint64_t nblk = T / CK;
  const double2* ck = ckpt + ((i * C + c) * nblk * 2) * 64 + lane;
  const uint64_t tstride = post_tile_doubles(I);
  bool nanflag = false;
  int exd = 0;

  double ecur[CK], enxt[CK];  // emission ratios: the emissions are (1, rho)
  double dcur[CK], dnxt[CK];
  double2 r0c, r1c, r0n, r1n;  // rows of the prefix operator in front of the block
  {
    const uint64_t b = nblk - 1;
#pragma unroll
    for (int u = 0; u < CK; ++u) {
      ecur[u] = ep[(b * CK + u) * 64];
      dcur[u] = dp[(b * CK + u) * 64];
    }
    r0c = b ? ck[(b * 2) * 64] : double2{1.0, 0.0};
    r1c = b ? ck[(b * 2 + 1) * 64] : double2{0.0, 1.0};
  }
  for (uint64_t b = nblk;;) {
    --b;
    if (b > 0) {  // the block in front: in flight while this one is computed
      const uint64_t bn = b - 1;
#pragma unroll
      for (int u = 0; u < CK; ++u) {
        enxt[u] = ep[(bn * CK + u) * 64];
        dnxt[u] = dp[(bn * CK + u) * 64];
      }
      r0n = bn ? ck[(bn * 2) * 64] : double2{1.0, 0.0};
      r1n = bn ? ck[(bn * 2 + 1) * 64] : double2{0.0, 1.0};
    }
    // forward vectors of the block's sites, from the checkpoint
    double v0 = fma(vin0, r0c.x, vin1 * r1c.x);
    double v1 = fma(vin0, r0c.y, vin1 * r1c.y);
    double f0[CK], f1[CK], cc[CK];
#pragma unroll
    for (int u = 0; u < CK; ++u) {
      cc[u] = coanc(al, dcur[u]);
      const double a = 1 - cc[u];
      const double sm = v0 + v1;
      v0 = fma(a * q0, sm, cc[u] * v0);
      v1 = fma(a * q1, sm, cc[u] * v1) * ecur[u];
      if (u == CK / 2 - 1) {  // the scale of (v0, v1) does not matter: keep it in range
        int dummy = 0;
        renorm2(v0, v1, dummy);
      }
      f0[u] = v0;
      f1[u] = v1;
    }
    // backward through the block: posterior, then the beta step
#pragma unroll
    for (int u = CK - 1; u >= 0; --u) {
      const uint64_t t = b * CK + u;
      const double x0 = f0[u] * w0, x1 = f1[u] * w1;
      double g1 = x1 * rcp_nr2(x0 + x1);  // 0/0 (no probability mass) stays NaN
      if (j * T + t < S && g1 != g1) nanflag = true;
      // check_interv (gen_func.cpp:55-70)
      if (g1 < kEPS) g1 = 0;
      else if (g1 > 1 - kEPS) g1 = 1;
      stage[b & 1][u][m8][lp] = g1;
      // beta step: w'_k = c u_k + a (q . u),  u = e * w
      const double a = 1 - cc[u];
      const double u0 = w0, u1 = ecur[u] * w1;
      const double sq = a * fma(q0, u0, q1 * u1);
      w0 = fma(cc[u], u0, sq);
      w1 = fma(cc[u], u1, sq);
    }
    renorm2(w0, w1, exd);
    // the block's 8 x 32 x 8 posteriors leave through LDS: thread k writes chunk k / 8 of
    // individual k % 8, i.e. the 256 threads store 2 KB contiguous per site (the buffers
    // alternate: the barrier of block b orders its reads before the writes of block b - 2)
    __syncthreads();
    {
      const int rl = threadIdx.x >> 3, rm = threadIdx.x & 7;
#pragma unroll
      for (int u = 0; u < CK; ++u) pp[(b * CK + u) * tstride] = stage[b & 1][u][rm][rl];
    }
    if (b == 0) break;
#pragma unroll
    for (int u = 0; u < CK; ++u) {
      ecur[u] = enxt[u];
      dcur[u] = dnxt[u];
    }
    r0c = r0n;
    r1c = r1n;
  }
  if (nanflag) flags[FLAG_NAN] = 1;
}

// tile-major posteriors -> site-major [S][I] (multi-GPU packing, host read-back, est_maf
// with more individuals than one wave holds); tile = (c, t) x 64 lanes x 64 individuals,
// 16-byte accesses on both sides: a thread reads two lanes of one individual and writes
// two individuals of one site (I even; odd I takes the 8-byte path)
template <bool PAIRS>
__global__ void __launch_bounds__(256)
k_fast_post_to_site_major(const double* __restrict__ post, uint64_t I, uint64_t S, uint64_t T,
                          uint32_t C, double* __restrict__ marg) {
  constexpr int TI = 64;
  __shared__ double tile[64][TI + 2];  // [lane][individual]
  const uint64_t n_it = (I + TI - 1) / TI;
  const uint64_t ct = blockIdx.x / n_it;  // c * T + t
  const uint64_t i0 = (blockIdx.x % n_it) * TI;
  const uint64_t c = ct / T, t = ct % T;
  if constexpr (PAIRS) {
    const int lp = threadIdx.x & 31, ty = threadIdx.x >> 5;  // lane pair, 8 individuals a pass
    for (int ii = ty; ii < TI; ii += 8) {
      const uint64_t i = i0 + ii;
      if (i < I) {
        if constexpr (kPost8) {  // lanes are 8 doubles apart: two 8-byte reads
          tile[2 * lp][ii] = post[post_lane_off(ct, 2 * lp, I) + post_ind_off(i)];
          tile[2 * lp + 1][ii] = post[post_lane_off(ct, 2 * lp + 1, I) + post_ind_off(i)];
        } else {
          const double2 v = *reinterpret_cast<const double2*>(post + (ct * I + i) * 64 + 2 * lp);
          tile[2 * lp][ii] = v.x;
          tile[2 * lp + 1][ii] = v.y;
        }
      }
    }
    __syncthreads();
    const int ip = threadIdx.x & 31, tz = threadIdx.x >> 5;  // individual pair, 8 sites a pass
    for (int ll = tz; ll < 64; ll += 8) {
      const uint64_t s = (c * 64 + ll) * T + t;
      const uint64_t i = i0 + 2 * ip;
      if (s < S && i < I)  // I even: i + 1 < I as well, and s * I + i is even
        *reinterpret_cast<double2*>(marg + s * I + i) = double2{tile[ll][2 * ip], tile[ll][2 * ip + 1]};
    }
  } else {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int ii = ty; ii < TI; ii += 4) {
      const uint64_t i = i0 + ii;
      if (i < I) tile[tx][ii] = post[post_lane_off(ct, tx, I) + post_ind_off(i)];
    }
    __syncthreads();
    for (int ll = ty; ll < 64; ll += 4) {
      const uint64_t s = (c * 64 + ll) * T + t;
      const uint64_t i = i0 + tx;
      if (s < S && i < I) marg[s * I + i] = tile[ll][tx];
    }
  }
}

// ---- emissions --------------------------------------------------------------
// Stand-alone refresh of the emission ratios and of sum log e0 (an E-step or an objective
// call that no fresh forward walk precedes): one wave per (individual, chunk) runs the fresh
// walk's source over its sites -- the interleaved likelihoods (or codes) and frequencies in,
// the ratios out, every access one contiguous segment per wave-instruction.
template <int SRC>
__global__ void __launch_bounds__(64)
k_fast_refresh(LklArrays arr, uint64_t T, uint32_t C) {
  static_assert(SRC != SRC_PLAIN, "a refresh computes the emissions");
  const uint64_t w = blockIdx.x;  // i * C + c
  const uint32_t c = (uint32_t)(w % C);
  const int lane = threadIdx.x;
  using Src = SrcOf<SRC>;
  Src src(arr, (w * T) * 64 + lane, ((uint64_t)c * T) * 64 + lane);
  for (uint64_t t0 = 0; t0 < T; t0 += RENORM) {  // T is a multiple of RENORM
    typename Src::Buf buf[RENORM];
#pragma unroll
    for (int u = 0; u < RENORM; ++u) buf[u] = src.load(t0 + u);
#pragma unroll
    for (int u = 0; u < RENORM; ++u) {
      double rho, d;
      src.get(buf[u], t0 + u, rho, d);
    }
    src.rescale();
  }
  const double bl = wave_sum(src.base.log_value());
  if (lane == 0) arr.base_c[w] = bl + (arr.gl_scale_c ? arr.gl_scale_c[w] : 0.0);
}

// site-major codes [S][I] (2 bits per cell) -> interleaved words [I][C][T/16][64]: word
// (i, c, tb, lane) holds the codes of sites (c*64 + lane)*T + tb*16 .. + 15.  Padding sites
// get code 0, which the padding frequency 0 turns into the identity emission (1, 1).
__global__ void __launch_bounds__(256)
k_fast_geno_interleave(const uint32_t* __restrict__ codes, uint64_t cell0, uint64_t I, uint64_t S,
                       uint64_t T, uint32_t C, uint32_t* __restrict__ geno_il) {
  const uint64_t TB = T >> 4;
  const uint64_t n = I * C * TB * 64;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t lane = k & 63, r = k >> 6;
    const uint64_t tb = r % TB, wv = r / TB;
    const uint64_t c = wv % C, i = wv / C;
    const uint64_t s0 = (c * 64 + lane) * T + tb * 16;
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint64_t s = s0 + j;
      if (s < S) w |= gl_code(codes, cell0 + s * I + i) << (2 * j);
    }
    geno_il[k] = w;
  }
}

// linear-space copy of the genotype likelihoods: they do not change during a run, and
// both the emission refresh and est_maf would otherwise exponentiate all 3 I S of them
// in every EM iteration
__global__ void __launch_bounds__(256)
k_fast_exp(const double* in, double* out, uint64_t n) {  // in == out allowed
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (uint64_t)gridDim.x * blockDim.x)
    out[k] = exp(in[k]);
}

// largest finite distance (bit pattern order == value order for non-negative doubles)
__global__ void __launch_bounds__(256)
k_fast_max_finite(const double* __restrict__ pos, uint64_t S, unsigned long long* __restrict__ out) {
  unsigned long long m = 0;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < S;
       k += (uint64_t)gridDim.x * blockDim.x) {
    const double d = pos[k];
    if (d >= 0 && d < 1e30) {
      const unsigned long long b = ngh_bits(d);
      m = b > m ? b : m;
    }
  }
  atomicMax(out, m);
}

__global__ void __launch_bounds__(256)
k_fast_pos_interleave(const double* __restrict__ pos, uint64_t S, uint64_t T, uint32_t C,
                      double* __restrict__ pos_il) {
  const uint64_t n = (uint64_t)C * T * 64;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t lane = k & 63, ct = k >> 6;
    const uint64_t c = ct / T, t = ct % T;
    const uint64_t s = (c * 64 + lane) * T + t;
    // padding: d = 0 -> c = 1 -> identity transition; chromosome starts: +inf is stored
    // as 1e30 (exp(-alpha 1e30) = 0 for every alpha >= 1e-15, and no inf*0 can arise)
    const double d = (s < S) ? pos[s] : 0.0;
    pos_il[k] = (d < 1e30) ? d : 1e30;
  }
}

// freq[S] -> interleaved [C][T][64] for the fresh forward walk (padding: 0, which makes both
// the dense padding likelihoods (1, 1, 1) and the packed padding code 0 the identity emission
// (1, 1) exactly); flags an
// allele frequency outside [0, 1] like calc_emission does (shared/HMM.cpp:145-146)
__global__ void __launch_bounds__(256)
k_fast_freq_interleave(const double* __restrict__ freq, uint64_t S, uint64_t T, uint32_t C,
                       double* __restrict__ freq_il, int* __restrict__ flags) {
  const uint64_t n = (uint64_t)C * T * 64;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t lane = k & 63, ct = k >> 6;
    const uint64_t c = ct / T, t = ct % T;
    const uint64_t s = (c * 64 + lane) * T + t;
    double f = 0.0;
    if (s < S) {
      f = freq[s];
      if (f < 0 || f > 1) {  // the reference's predicate (HMM.cpp:145): NaN passes, and
                             // surfaces later as "invalid Lkl found!" like in exact mode
        flags[FLAG_INVALID_MAF] = 1;
        f = __builtin_nan("");
      }
    }
    freq_il[k] = f;
  }
}

// site-major LOG GL [S][I][3] -> interleaved, each cell relative to its largest likelihood
// (glq_encode), layout of e_il; padding sites get (1, 1, 1), which any frequency turns into
// the identity emission (1, 1).
// tile = (c, t) x 64 lanes (sites T apart) x 32 individuals
__global__ void __launch_bounds__(256)
k_fast_gl_interleave(const double* __restrict__ gl_log, uint64_t I, uint64_t S, uint64_t T,
                     uint32_t C, double2* __restrict__ glq_il) {
  __shared__ double2 tq[32][65];
  const uint64_t n_it = (I + 31) / 32;
  const uint64_t ct = blockIdx.x / n_it;
  const uint64_t i0 = (blockIdx.x % n_it) * 32;
  const uint64_t c = ct / T, t = ct % T;
  {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 individuals x 8
    const uint64_t i = i0 + tx;
    for (int ll = ty; ll < 64; ll += 8) {
      const uint64_t s = (c * 64 + ll) * T + t;
      double2 q{1.0, 1.0};
      if (s < S && i < I) {
        const double* g = gl_log + (s * I + i) * 3;
        q = glq_encode(g[0], g[1], g[2]);
      }
      tq[tx][ll] = q;
    }
  }
  __syncthreads();
  {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 lanes x 4
    for (int ii = ty; ii < 32; ii += 4) {
      const uint64_t i = i0 + ii;
      if (i < I) glq_il[((i * C + c) * T + t) * 64 + tx] = tq[ii][tx];
    }
  }
}

// scale_c[i][c] = sum over the sites of wave (i, c) of the cell's largest log likelihood (what
// glq_encode divides out).  Block = chunk c x 32 individuals; a thread adds its sites in
// order, the eight partial sums of an individual are added in order: the same bits every run.
__global__ void __launch_bounds__(256)
k_fast_gl_scale(const GlView gl_log, uint64_t I, uint64_t S, uint64_t T, uint32_t C,
                double* __restrict__ scale_c) {
  __shared__ double part[8][33];
  const uint64_t n_it = (I + 31) / 32;
  const uint64_t c = blockIdx.x / n_it;
  const uint64_t i0 = (blockIdx.x % n_it) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const uint64_t i = i0 + tx;
  double acc = 0;
  if (i < I) {
    for (uint64_t j = c * 64 + ty; j < c * 64 + 64; j += 8) {
      for (uint64_t t = 0; t < T; ++t) {
        const uint64_t s = j * T + t;
        if (s >= S) break;
        double g0, g1, g2;
        gl_fetch(gl_log, s * I + i, g0, g1, g2);
        acc += fmax(g0, fmax(g1, g2));
      }
    }
  }
  part[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && i < I) {
    double sum = 0;
    for (int k = 0; k < 8; ++k) sum += part[k][tx];
    scale_c[i * C + c] = sum;
  }
}

// out[i][s][k] = log emission of state k (test/debug read-back): recomputed from the
// likelihoods (or interleaved codes) and frequencies with the fresh walk's expressions --
// only the ratio of the two is kept between walks
__global__ void __launch_bounds__(256)
k_fast_export_e(LklArrays arr, bool packed, const double* __restrict__ gl_lin, uint64_t I,
                uint64_t S, uint64_t T, uint32_t C, double* __restrict__ out) {
  const uint64_t n = I * S;
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
       k += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t i = k / S, s = k % S;
    const uint64_t j = s / T, t = s % T;
    const uint64_t c = j >> 6, lane = j & 63;
    const uint64_t wv = i * C + c;
    const double maf = arr.freq_il[(c * T + t) * 64 + lane], om = 1 - maf;
    const double bb = om * maf;
    const double h00 = om * om, h02 = maf * maf;
    double e0, e1;
    if (packed) {
      const uint32_t w = arr.geno_il[(wv * (T >> 4) + (t >> 4)) * 64 + lane];
      const uint32_t code = (w >> ((uint32_t)(t & 15) * 2)) & 3u;
      const double u = arr.u_lin;
      const double u0 = fma(u, h00, fma(u, 2 * bb, u * h02));
      const double u1 = fma(u, h00 + bb, u * (h02 + bb));
      e0 = code == 0 ? h00 : code == 1 ? 2 * bb : code == 2 ? h02 : u0;
      e1 = code == 0 ? h00 + bb : code == 1 ? 0.0 : code == 2 ? h02 + bb : u1;
    } else {
      const double* g = gl_lin + (s * I + i) * 3;  // the site-major linear copy est_maf reads
      e0 = fma(g[0], h00, fma(g[1], 2 * bb, g[2] * h02));
      e1 = fma(g[0], h00 + bb, g[2] * (h02 + bb));
    }
    out[k * 2] = log(e0);
    out[k * 2 + 1] = log(e1);
  }
}

// ---- est_maf ----------------------------------------------------------------

// reference-order log-space term for a cell whose linear weights all vanish
// (e.g. a called heterozygote with posterior IBD = 1): gen_func.cpp:984-1000
__device__ double2 estmaf_term_logspace(const double* g, double freq, double F) {
  double h[3];
  h[0] = (1 - freq) * (1 - freq) + (1 - freq) * freq * F;
  h[1] = 2 * (1 - freq) * freq - 2 * (1 - freq) * freq * F;
  h[2] = freq * freq + (1 - freq) * freq * F;
  double pp[3];
  for (int k = 0; k < 3; ++k) {
    double l = log(h[k]);
    if (l == -__builtin_huge_val()) l = -kINF;
    h[k] = l;
  }
  if (F == 1) h[1] = -kINF;
  double M = g[0] + h[0];
  for (int k = 0; k < 3; ++k) {
    pp[k] = g[k] + h[k];
    M = (pp[k] >= M) ? pp[k] : M;
  }
  double sum = 0;
  for (int k = 0; k < 3; ++k) sum += exp(pp[k] - M);
  const double norm = log(sum) + M;
  for (int k = 0; k < 3; ++k) pp[k] = exp(pp[k] - norm);
  return double2{pp[1] + pp[2] * (2 - F), 2 * pp[1] + (pp[0] + pp[2]) * (2 - F)};
}

// ---- wave-wide sum that ends in a wave-uniform value -----------------------
// DPP moves stay inside the SIMD (no LDS round trip as with ds_bpermute), which
// matters here: est_maf has one dependent reduction per pass and ~100 passes.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  // full row mask and in-row permutations: every lane is written, so the "old" operand
  // is irrelevant (mov_dpp leaves it undefined and saves the two zeroing moves)
  const uint64_t b = ngh_bits(v);
  const int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)b, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(b >> 32), CTRL, 0xf, 0xf, true);
  return ngh_from_bits(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

__device__ __forceinline__ double lane_value(double v, int lane) {
  const uint64_t b = ngh_bits(v);
  const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)b, lane);
  const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
  return ngh_from_bits(((uint64_t)hi << 32) | lo);
}

// total in the lanes of the last row (48..63); other lanes hold partial sums
__device__ __forceinline__ double wave_sum_lastrow(double v) {
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror: every lane holds its row total
  {  // row_bcast15 into rows 1 and 3, then row_bcast31 into rows 2 and 3
    const uint64_t b = ngh_bits(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x142, 0xa, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x142, 0xa, 0xf, false);
    v += ngh_from_bits(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
  }
  {
    const uint64_t b = ngh_bits(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x143, 0xc, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x143, 0xc, 0xf, false);
    v += ngh_from_bits(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
  }
  return v;
}

__device__ __forceinline__ double wave_sum_uniform(double v) {
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror: every lane now holds its 16-lane row total
  return ((lane_value(v, 0) + lane_value(v, 16)) + lane_value(v, 32)) + lane_value(v, 48);
}

// Sums of two per-lane values over the wave in ONE reduction tree: the first step swaps
// the upper half of pn with the lower half of pd (v_permlane32_swap, gfx950), so lanes
// 0..31 carry pn partials and lanes 32..63 pd partials; four in-row DPP steps and one
// row_bcast15 finish both.  Returns the value whose lane 31 holds sum(pn) and lane 63
// sum(pd).
__device__ __forceinline__ double wave_sum_pair(double pn, double pd) {
  const uint64_t bn = ngh_bits(pn), bd = ngh_bits(pd);
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)bn, (unsigned)bd, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(bn >> 32), (unsigned)(bd >> 32),
                                                   false, false);
  double v = ngh_from_bits(((uint64_t)hi[0] << 32) | lo[0]) +
             ngh_from_bits(((uint64_t)hi[1] << 32) | lo[1]);
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror: every lane holds its row total
  {  // row_bcast15 into rows 1 and 3
    const uint64_t b = ngh_bits(v);
    const int l = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x142, 0xa, 0xf, false);
    const int h = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x142, 0xa, 0xf, false);
    v += ngh_from_bits(((uint64_t)(uint32_t)h << 32) | (uint32_t)l);
  }
  return v;
}

// ---- est_maf: certified interpolation of the per-pass sums -------------------
// The reference's loop (gen_func.cpp:981-1006) is a running average: pass k evaluates
// two sums over all individuals at the odds r_k of the current frequency and adds them
// to num/den; r_k creeps towards its limit like 1/k, so nearly every site runs into the
// 100-pass cap.  Both sums are rational functions of r whose poles all lie in Re r <= 0
// (their denominators sA + r sb + r^2 sC have non-negative coefficients), hence analytic
// in a disc of radius >= r around any r > 0.  In the Moebius variable of the build below an
// interval of ratio hi / lo = 2 has Bernstein-ellipse parameter 11.7: EN = 12 Chebyshev nodes
// reproduce the sums to 11.7^-12 = 1.5e-13 before the constant (measured against all-exact
// passes at five full-size shapes: frequencies within 6.7e-13; 14 nodes, the previous
// default, 8e-15 -- three and a half orders inside the 1e-9 the frequencies are held to,
// for two evaluations of all individuals fewer per site: est_maf 8.8 -> 8.1 ms at 1000 x 1M).
// So after a few exact passes the kernel evaluates the sums exactly at the EN Chebyshev nodes
// of an interval ahead of r_k (as expensive as EN passes), CHECKS the interpolant against the
// next exact pass (relative EST_TOL = 1e-11, else the site stays on exact passes), and hands
// the site to k_fast_estmaf_interp, where one LANE per site runs the
// remaining passes on the barycentric formula: the same recursion, same pass count,
// same stopping rule, at ~1/60 of the cost per pass.  A pass whose stopping decision
// would be closer than 1e-9 (relative) to the threshold, or whose r leaves the
// interval, goes back to exact evaluation (one more build is allowed per site).
#ifndef NGHMM_EST_EN
#define NGHMM_EST_EN 12
#endif
constexpr int EN = NGHMM_EST_EN;        // Chebyshev nodes per interval
constexpr int EST_SCALARS = 8;          // num, den, pnum, pden, iters, mid, half, tF
constexpr int EST_FIELDS = EST_SCALARS + 2 * EN;
enum : uint8_t { EST_DONE = 0, EST_INTERP = 1, EST_EXACT = 2 };
#ifndef NGHMM_EST_K0
#define NGHMM_EST_K0 2
#endif
constexpr int EST_K0 = NGHMM_EST_K0;    // exact passes before the first interval
constexpr int EST_MIN_GAIN = 24;        // build only if about this many passes remain
#ifndef NGHMM_EST_DMAX
#define NGHMM_EST_DMAX 0.85
#endif
#ifndef NGHMM_EST_MULT
#define NGHMM_EST_MULT 32.0
#endif
constexpr double EST_DMAX = NGHMM_EST_DMAX;  // interval length <= EST_DMAX * r ahead ...
constexpr double EST_BACK = 0.1;        // ... plus this fraction of it behind
constexpr double EST_MULT = NGHMM_EST_MULT;  // ... and about this many current steps
#ifndef NGHMM_EST_FIT
#define NGHMM_EST_FIT 0.72
#endif
#ifndef NGHMM_EST_KMAX
#define NGHMM_EST_KMAX 32
#endif
constexpr double EST_FIT = NGHMM_EST_FIT;    // build once k * step <= EST_FIT * EST_DMAX * r ...
constexpr int EST_KMAX = NGHMM_EST_KMAX;     // ... or after this many passes at the latest
#ifndef NGHMM_EST_TOL
#define NGHMM_EST_TOL 1e-11
#endif
constexpr double EST_TOL = NGHMM_EST_TOL;  // interpolant vs exact pass, relative
constexpr double EST_GUARD = 1e-9;      // stopping decisions this close go back to exact
// cos((2j+1) pi/(2 EN)) and (-1)^j sin((2j+1) pi/(2 EN)): first-kind Chebyshev nodes and
// their barycentric weights
#if NGHMM_EST_EN == 8
__constant__ double kChebC[EN] = {0.9807852804032304, 0.8314696123025452, 0.5555702330196023, 0.19509032201612833, -0.1950903220161282, -0.555570233019602, -0.8314696123025453, -0.9807852804032304};
__constant__ double kChebW[EN] = {0.19509032201612825, -0.5555702330196022, 0.8314696123025452, -0.9807852804032304, 0.9807852804032304, -0.8314696123025455, 0.5555702330196022, -0.1950903220161286};
#elif NGHMM_EST_EN == 12
__constant__ double kChebC[EN] = {0.9914448613738104, 0.9238795325112867, 0.7933533402912352, 0.6087614290087207, 0.38268343236508984, 0.1305261922200517, -0.1305261922200516, -0.3826834323650895, -0.6087614290087207, -0.793353340291235, -0.9238795325112867, -0.9914448613738104};
__constant__ double kChebW[EN] = {0.13052619222005157, -0.3826834323650898, 0.6087614290087207, -0.7933533402912352, 0.9238795325112867, -0.9914448613738104, 0.9914448613738104, -0.9238795325112868, 0.7933533402912352, -0.6087614290087209, 0.3826834323650899, -0.130526192220052};
#elif NGHMM_EST_EN == 14
__constant__ double kChebC[EN] = {0.9937122098932426, 0.9438833303083676, 0.8467241992282841, 0.7071067811865476, 0.5320320765153366, 0.3302790619551673, 0.11196447610330769, -0.11196447610330758, -0.3302790619551672, -0.5320320765153365, -0.7071067811865475, -0.8467241992282841, -0.9438833303083676, -0.9937122098932426};
__constant__ double kChebW[EN] = {0.11196447610330786, -0.3302790619551671, 0.5320320765153366, -0.7071067811865475, 0.8467241992282841, -0.9438833303083675, 0.9937122098932426, -0.9937122098932426, 0.9438833303083675, -0.8467241992282842, 0.7071067811865476, -0.5320320765153367, 0.3302790619551672, -0.11196447610330798};
#elif NGHMM_EST_EN == 16
__constant__ double kChebC[EN] = {0.9951847266721969, 0.9569403357322088, 0.881921264348355, 0.773010453362737, 0.6343932841636455, 0.4713967368259978, 0.29028467725446233, 0.09801714032956077, -0.09801714032956065, -0.29028467725446216, -0.4713967368259977, -0.6343932841636454, -0.773010453362737, -0.8819212643483549, -0.9569403357322088, -0.9951847266721968};
__constant__ double kChebW[EN] = {0.0980171403295606, -0.29028467725446233, 0.47139673682599764, -0.6343932841636455, 0.773010453362737, -0.8819212643483549, 0.9569403357322089, -0.9951847266721968, 0.9951847266721969, -0.9569403357322089, 0.881921264348355, -0.7730104533627371, 0.6343932841636455, -0.47139673682599786, 0.2902846772544624, -0.09801714032956083};
#else
#error "NGHMM_EST_EN must be 8, 12, 14 or 16"
#endif

// W = BLOCK/64 waves per site, NI individuals per lane held in registers.  With
//   A = (1-f)^2, b = (1-f) f, C = f^2
// the weights w_g = p_g * HWE_g(f, F) of calc_HWE/post_prob (gen_func.cpp:920-957) are
// linear in (A, b, C):  w0 = p0 (A + bF), w1 = b c1 with c1 = 2 p1 (1-F), w2 = p2 (C + bF),
// and the reference's per-individual terms (gen_func.cpp:999-1000) become
//   num-term = (w1 + (2-F) w2) / sum
//   den-term = (2 w1 + (2-F)(w0 + w2)) / sum = (2-F) + F w1 / sum
// Dividing every weight by (1-f)^2 leaves, in the odds r = f/(1-f),
//   sum' = sA + r sb + r^2 sC,  num-term = r (u0 + r nC) / sum',  den-term = (2-F) + r fc / sum'
// six constants per individual, 8 FP64 instructions per individual and evaluation with
// the reciprocals taken four at a time, and the (2-F) part of the denominator a per-site
// constant.  Nothing is read from memory again after the constants are formed.  A site
// with a cell whose weights all vanish (a called heterozygote at posterior IBD = 1, ...)
// ends with a non-finite frequency, is flagged and redone by k_fast_estmaf_stream, which
// takes the reference-order log-space route for such cells.
//
// Up to 1024 individuals one wave holds the whole site (NI <= 16: 192 VGPRs of constants,
// two waves per SIMD) and an evaluation is 128 + ~45 instructions; beyond that W waves
// share a site (e.g. the site-sharded frequency step of a multi-GPU run), their partial
// sums meet in LDS once per evaluation (double-buffered, one barrier) and are added in
// wave order, so the result does not depend on scheduling.
//
// fresh != 0: every site starts the loop; else only sites whose status is EST_EXACT
// resume from `state`.  n_exact passes are evaluated exactly, then (allow_build) the
// interval is built and checked; a site that ends here writes freq_out/redo.
constexpr int ESTMAF_MAXW = 16;

// is site (c*64 + l)*T + t in the tile rows c*T + t of [row0, row1)?  (tile_T == 0: no tiles,
// every site is)
__device__ __forceinline__ bool in_tile_rows(uint64_t site, uint64_t tile_T, uint64_t row0,
                                             uint64_t row1) {
  if (tile_T == 0) return true;
  const uint64_t j = site / tile_T, t = site - j * tile_T;
  const uint64_t row = (j >> 6) * tile_T + t;
  return row >= row0 && row < row1;
}
// TILE: the posteriors are read from the E-step's tile-major layout (post_lane_off /
// post_ind_off; site (c*64 + l)*T + t), one wave per site.  With kPost8 a wave-load of 64
// consecutive individuals is eight fully used 64 B sectors.  (Without: a lane's 8-byte loads
// are 512 B apart and the sector around each holds the eight sites l0..l0+7 of one individual;
// workgroups go round-robin to the 8 XCDs, each with its own L2, so the blockIdx -> site map
// gives XCD x the sites l = 8x..8x+7 of every tile row in eight consecutive workgroups, for
// the sector to be fetched once and hit in that L2 seven times.  The map is kept.)
// one site on the W = BLOCK / 64 waves of a workgroup (see above); the shared arrays are the
// calling kernel's
// The size dispatch of fast_estmaf gives the variant (NI, BLOCK) only to cohorts larger than
// the previous variant holds, so its first slots are full for every thread: no masking there.
__host__ __device__ constexpr int estmaf_full_slots(int NI, int BLOCK) {
  return BLOCK == 64 ? (NI == 16 ? 12 : NI == 12 ? 8 : NI == 8 ? 4 : NI == 4 ? 2 : NI == 2 ? 1 : 0)
                     : (NI == 16 ? 8 : 0);  // 128: > 1024 = 8 x 128; 256: > 2048; 512: > 4096
}

template <int NI, int BLOCK, bool TILE>
__device__ __forceinline__ void estmaf_site(
    const GlView& gl, const double* __restrict__ marg_blocks, uint64_t S_own, uint64_t I_tot,
    uint64_t I_blk, double* __restrict__ freq_out, uint8_t* __restrict__ redo,
    uint8_t* __restrict__ status, double* __restrict__ state, uint64_t state_stride, int fresh,
    int n_exact, int allow_build, uint64_t site, const double* __restrict__ tile_col,
    double (&xch)[2][ESTMAF_MAXW][2],
    double2 (&nodebuf)[(BLOCK == 64 && NI >= 8) ? EN : 1][(BLOCK == 64 && NI >= 8) ? 65 : 1],
    double2 (&xnode)[(BLOCK == 64 && NI >= 8) ? 1 : EN][(BLOCK == 64 && NI >= 8) ? 1 : BLOCK / 64]) {
  constexpr int W = BLOCK / 64;
  constexpr bool PARK = (W == 1 && NI >= 8);
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint32_t tix = threadIdx.x;  // index among the site's threads
  constexpr uint64_t stride = BLOCK;
  const uint64_t cell_s = site * I_tot;  // first cell of the site's row

  double tF_lane_out;
  // The loads of eight slots (32 per lane) are issued before anything waits on them
  // (out-of-range slots re-read the last individual and are masked afterwards): a wave
  // has two memory round trips here, not NI of them.
  double sA[NI], sb[NI], sC[NI], u0[NI], nC[NI], fc[NI];
  {
#ifndef NGHMM_EST_NB
#define NGHMM_EST_NB 8
#endif
    // slots per batch of loads; NI = 12: two batches of six
    constexpr int NB = NI < NGHMM_EST_NB ? NI : (NI % NGHMM_EST_NB ? NI / 2 : NGHMM_EST_NB);
    static_assert(NI % NB == 0, "whole batches");
    const bool one_block = (I_blk == I_tot);
    const uint32_t ib = (uint32_t)I_blk;
    double tF_acc = 0;
#pragma unroll
    for (int k0 = 0; k0 < NI; k0 += NB) {
      double r0[NB], r1[NB], r2[NB], rF[NB];
      uint64_t ic[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const uint64_t i = (uint64_t)tix + stride * (k0 + j);
        // (a slot that is full for every thread needs no clamp: its addresses are the lane's
        // plus a constant)
        ic[j] = (k0 + j < estmaf_full_slots(NI, BLOCK) || i < I_tot) ? i : I_tot - 1;
        gl_fetch(gl, cell_s + ic[j], r0[j], r1[j], r2[j]);
      }
      if constexpr (TILE) {
#pragma unroll
        for (int j = 0; j < NB; ++j)  // (post_ind_off is additive over multiples of 8)
          rF[j] = tile_col[k0 + j < estmaf_full_slots(NI, BLOCK)
                               ? post_ind_off(tix) + (uint64_t)(k0 + j) * post_ind_off(stride)
                               : post_ind_off(ic[j])];
      } else if (one_block) {
#pragma unroll
        for (int j = 0; j < NB; ++j) rF[j] = marg_blocks[site * I_blk + ic[j]];
      } else {  // posteriors arrive in rank blocks [I_tot / I_blk][S_own][I_blk]
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const uint32_t q = (uint32_t)ic[j] / ib;
          rF[j] = marg_blocks[((uint64_t)q * S_own + site) * I_blk + ((uint32_t)ic[j] - q * ib)];
        }
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int k = k0 + j;
        const bool valid = k < estmaf_full_slots(NI, BLOCK) || (uint64_t)tix + stride * k < I_tot;
        // empty slot: likelihoods (1, 0, 0) at posterior 0 give sum' = 1 and numerators 0 --
        // it contributes nothing (four selects on the inputs instead of six on the results)
        const double p0 = valid ? r0[j] : 1.0, p1 = valid ? r1[j] : 0.0;  // linear GL
        const double p2 = valid ? r2[j] : 0.0, F = valid ? rF[j] : 0.0;
        // (at F = 1 the heterozygote's weight is the reference's exp(-1e15) = 0: so is the
        // product, p1 being finite)
        const double cc = p1 * fma(-2.0, F, 2.0);  // = 2 p1 (1 - F), the same bits, one op fewer
        const double n2 = (2 - F) * p2;
        sA[k] = p0;
        sb[k] = fma(F, p0 + p2, cc);
        sC[k] = p2;
        u0[k] = fma(n2, F, cc);
        nC[k] = n2;
        fc[k] = F * cc;
        tF_acc += valid ? 2 - F : 0.0;
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the next batch's loads out of this one
    }
    tF_lane_out = tF_acc;
  }
  double tF_sum = wave_sum_uniform(tF_lane_out);
  if constexpr (W > 1) {
    if (lane == 0) xch[1][wv][0] = tF_sum;
    __syncthreads();
    tF_sum = xch[1][0][0];
#pragma unroll
    for (int w = 1; w < W; ++w) tF_sum += xch[1][w][0];
    __syncthreads();
  }

  // The loop carries num and den only.  The odds of freq = num/den are num/(den - num):
  // one reciprocal on the serial path instead of two, and the reference's stopping rule
  // |prev - freq| > EPSILON (gen_func.cpp:1006) is tested cross-multiplied,
  // |num_prev den - num den_prev| > EPSILON den den_prev, which needs no quotient.
  int iters = 0;
  int buf = 0;
  double num = 0, den = 0;
  double pnum = 0.01, pden = 1.0;  // freq = 0.01 (gen_func.cpp:976)
  if (!fresh) {
    num = state[0 * state_stride + site];
    den = state[1 * state_stride + site];
    pnum = state[2 * state_stride + site];
    pden = state[3 * state_stride + site];
    iters = (int)state[4 * state_stride + site];
  }
  bool built = !allow_build;  // at most one interval per launch
  int n_before = n_exact;     // exact passes before deciding on it
  bool check = false, interp_ok = false;
  double mid = 0, half = 0, my_gn = 0, my_gd = 0, rprev = 0;  // mid: centre a, half: h (see the build)
  // this lane's part of the two per-pass sums at odds r
  auto lane_sums = [&](double r, double& pn, double& pd) {
    pn = 0;
    pd = 0;
    if constexpr (NI >= 4) {
      // reciprocals four at a time (Montgomery's trick): one v_rcp_f64 + Newton step and
      // 9 multiplies instead of four reciprocals; v_rcp_f64 is the slow instruction
#pragma unroll
      for (int k0 = 0; k0 < NI; k0 += 4) {
        double sm[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sm[j] = fma(r, fma(r, sC[k0 + j], sb[k0 + j]), sA[k0 + j]);
        const double p01 = sm[0] * sm[1], p23 = sm[2] * sm[3];
        // a vanishing sum makes R infinite and the site's freq non-finite, which ends the
        // loop (the comparison below is false for NaN) and flags the site after it
        const double R = rcp_nr(p01 * p23);
        const double r01 = R * p23, r23 = R * p01;
        const double inv0 = r01 * sm[1], inv1 = r01 * sm[0];
        const double inv2 = r23 * sm[3], inv3 = r23 * sm[2];
        pn = fma(fma(nC[k0], r, u0[k0]), inv0, pn);
        pd = fma(fc[k0], inv0, pd);
        pn = fma(fma(nC[k0 + 1], r, u0[k0 + 1]), inv1, pn);
        pd = fma(fc[k0 + 1], inv1, pd);
        pn = fma(fma(nC[k0 + 2], r, u0[k0 + 2]), inv2, pn);
        pd = fma(fc[k0 + 2], inv2, pd);
        pn = fma(fma(nC[k0 + 3], r, u0[k0 + 3]), inv3, pn);
        pd = fma(fc[k0 + 3], inv3, pd);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const double inv = rcp_nr(fma(r, fma(r, sC[k], sb[k]), sA[k]));
        pn = fma(fma(nC[k], r, u0[k]), inv, pn);
        pd = fma(fc[k], inv, pd);
      }
    }
  };
  for (;;) {
    const double r = pnum * rcp_nr2(pden - pnum);
    double pn, pd;
    lane_sums(r, pn, pd);
    const double v = wave_sum_pair(pn, pd);
    double sn = lane_value(v, 31), sd = lane_value(v, 63);
    if constexpr (W > 1) {
      if (lane == 0) {
        xch[buf][wv][0] = sn;
        xch[buf][wv][1] = sd;
      }
      __syncthreads();
      sn = xch[buf][0][0];
      sd = xch[buf][0][1];
#pragma unroll
      for (int w = 1; w < W; ++w) {
        sn += xch[buf][w][0];
        sd += xch[buf][w][1];
      }
      buf ^= 1;
    }
    if (check) {  // the pass after a build: exact sums in hand, compare the interpolant
      const int nj = lane < EN ? lane : 0;
      const double t = (r - mid) / (r + mid) - half * kChebC[nj];
      const double q = (lane < EN) ? kChebW[nj] / t : 0.0;
      const double Sq = wave_sum_uniform(q);
      const double bn = wave_sum_uniform(q * my_gn) / Sq, bd = wave_sum_uniform(q * my_gd) / Sq;
      interp_ok = fabs(bn - sn) <= EST_TOL * fabs(sn) && fabs(bd - sd) <= EST_TOL * fabs(sd);
      check = false;
    }
    num = fma(r, sn, num);
    den = fma(r, sd, den + tF_sum);
    const double lhs = fabs(fma(pnum, den, -(num * pden))), thr = kEPS * (den * pden);
    const bool again = (lhs > thr) && (iters++ < 100);
    rprev = r;
    pnum = num;
    pden = den;
    if (!again) break;
    if (interp_ok) {  // hand the site to k_fast_estmaf_interp
      if (wv == 0) {
        if (lane < EN) {
          state[(EST_SCALARS + lane) * state_stride + site] = my_gn;
          state[(EST_SCALARS + EN + lane) * state_stride + site] = my_gd;
        }
        if (lane == 0) {
          state[0 * state_stride + site] = num;
          state[1 * state_stride + site] = den;
          state[2 * state_stride + site] = pnum;
          state[3 * state_stride + site] = pden;
          state[4 * state_stride + site] = (double)iters;
          state[5 * state_stride + site] = mid;
          state[6 * state_stride + site] = half;
          state[7 * state_stride + site] = tF_sum;
          status[site] = EST_INTERP;
        }
      }
      return;
    }
    if (!built && --n_before <= 0) {
      // |delta freq| shrinks roughly like 1/k^2: about k (sqrt(|delta|/EPSILON) - 1)
      // passes remain, and the odds still travel about k times their last step.  An
      // interval costs EN evaluations, so short tails stay exact; and a site whose
      // remaining travel does not fit into one interval yet (a frequency far from the
      // 0.01 every site starts at) takes a few more exact passes first, rather than
      // leaving its interval half way and paying for a second one.
      const double m_est = (double)iters * (sqrt(lhs / thr) - 1.0);
      const double rn = pnum * rcp_nr2(pden - pnum);
      const double step = fabs(rn - rprev);
      // (an interval reaches EST_DMAX * r ahead when r grows, down to r / (1 + EST_DMAX) when
      // it shrinks: the same ratio both ways)
      const double reach = (rn >= rprev) ? EST_DMAX * rn : EST_DMAX / (1 + EST_DMAX) * rn;
      const bool fits = (double)iters * step <= EST_FIT * reach;
      if (!fits && iters < EST_KMAX && m_est >= EST_MIN_GAIN) {
        n_before = 1;  // look again after the next exact pass
      } else {
        built = true;
      }
      if (built && m_est >= EST_MIN_GAIN && 100 - iters >= EST_MIN_GAIN) {
        const double g = fmin(EST_DMAX, fmax(EST_MULT * step / rn, 1e-3));  // relative length
        double lo, hi;
        if (rn >= rprev) {
          lo = rn * (1 - EST_BACK * g);
          hi = rn * (1 + g);
        } else {
          lo = rn / (1 + g);
          hi = rn * (1 + EST_BACK * g);
        }
        // Interpolation variable t = (r - a) / (r + a), a = sqrt(lo hi): the half plane
        // Re r <= 0 that holds every pole of the sums is the OUTSIDE of the unit disc in t, and
        // [lo, hi] becomes [-h, h] around 0 -- far from everything, so the Chebyshev
        // interpolant on EN nodes converges like rho^-EN with (rho + 1/rho) / 2 = 1/h: an
        // interval of ratio hi / lo = 2 has rho = 11.7, where the same nodes in r itself
        // (nearest pole at distance >= lo from an interval of length lo) would have rho = 5.8.
        mid = sqrt(lo * hi);
        half = (hi - mid) / (hi + mid);
        // a degenerate interval (rn not finite or not positive) keeps the site exact
        if (half > 0 && lo > 0 && hi < 1e300) {
          if constexpr (PARK) {
            // One wave holds the site: the node evaluations do not depend on each other,
            // so every lane parks its partial sums in LDS and the 16 x 64 partials are
            // added up once at the end -- no reduction tree (and its latency) per node.
            // Lane q*16 + j adds quarter q of node j's partials; two shuffles join the
            // quarters.
            // the nodes' odds: lane nd forms node nd's once, the loop reads them lane by lane
            // (a reciprocal and its Newton steps per node otherwise)
            const double tnl = half * kChebC[lane < EN ? lane : 0];
            const double r_nodes = mid * (1 + tnl) * rcp_nr2(1 - tnl);
#pragma unroll 1
            for (int nd = 0; nd < EN; ++nd) {
              double pn, pd;
              lane_sums(lane_value(r_nodes, nd), pn, pd);
              nodebuf[nd][lane] = double2{pn, pd};
            }
            __syncthreads();  // one wave: orders the LDS writes before the reads
            const int j = lane & 15, q4 = lane >> 4;
            double an = 0, ad = 0;
            if (j < EN) {
#pragma unroll
              for (int l = 0; l < 16; ++l) {
                const double2 t2 = nodebuf[j][q4 * 16 + l];
                an += t2.x;
                ad += t2.y;
              }
            }
            an += __shfl_xor(an, 16);
            ad += __shfl_xor(ad, 16);
            an += __shfl_xor(an, 32);
            ad += __shfl_xor(ad, 32);
            my_gn = an;  // lanes 0..EN-1 hold node `lane` (every quarter has the total)
            my_gd = ad;
            check = true;
          } else {
            // several waves per site: every wave reduces its own part of each node and the
            // waves' parts meet in LDS once for the whole interval (one barrier instead of
            // one per node), added in wave order
            const double tnl = half * kChebC[lane < EN ? lane : 0];
            const double r_nodes = mid * (1 + tnl) * rcp_nr2(1 - tnl);
#pragma unroll 1
            for (int nd = 0; nd < EN; ++nd) {
              double pn, pd;
              lane_sums(lane_value(r_nodes, nd), pn, pd);
              const double v = wave_sum_pair(pn, pd);
              const double sn = lane_value(v, 31), sd = lane_value(v, 63);
              if (lane == 0) xnode[nd][wv] = double2{sn, sd};
            }
            __syncthreads();
            double an = 0, ad = 0;
            if (lane < EN) {
#pragma unroll
              for (int w = 0; w < W; ++w) {
                const double2 t2 = xnode[lane][w];
                an += t2.x;
                ad += t2.y;
              }
            }
            my_gn = an;
            my_gd = ad;
            check = true;
          }
        }
      }
    }
  }
  if (tix == 0) {
    // non-finite or out-of-range result: a cell with vanishing weights (or f reaching 1);
    // the careful kernel redoes the site in the reference's log-space order
    const double freq = num / den;
    const bool ok = freq >= 0 && freq < 1;
    freq_out[site] = freq;
    redo[site] = ok ? 0 : 1;
    status[site] = EST_DONE;
  }
}

// tile-major posteriors: where site (c*64 + l)*T + t finds individual i at col[i * 64]
__device__ __forceinline__ const double* estmaf_tile_col(const double* marg_blocks, uint64_t site,
                                                         uint64_t tile_T, uint64_t I_tot) {
  const uint64_t j = site / tile_T, t = site - j * tile_T;
  return marg_blocks + post_lane_off((j >> 6) * tile_T + t, j & 63, I_tot);
}

#define ESTMAF_SHARED(NI, BLOCK)                                                              \
  constexpr bool PARK_ = ((BLOCK) == 64 && (NI) >= 8);                                         \
  __shared__ double xch[2][ESTMAF_MAXW][2];                      /* [buffer][wave][num, den] */ \
  __shared__ double2 nodebuf[PARK_ ? EN : 1][PARK_ ? 65 : 1];                                  \
  __shared__ double2 xnode[PARK_ ? 1 : EN][PARK_ ? 1 : (BLOCK) / 64] /* !PARK: per-wave node sums */

// every site from the start (freq = 0.01): one workgroup per site
template <int NI, int BLOCK, bool TILE>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2)))
k_fast_estmaf(const GlView gl, const double* __restrict__ marg_blocks,
              uint64_t S_own, uint64_t I_tot, uint64_t I_blk, uint64_t tile_T,
              double* __restrict__ freq_out, uint8_t* __restrict__ redo,
              uint8_t* __restrict__ status, double* __restrict__ state, uint64_t state_stride,
              int n_exact, int allow_build, uint64_t blk0) {
  // W == 1: per-lane partial sums of the interval's nodes (see the build); the pad makes lane
  // j's reads of row j conflict-free (few individuals per lane, NI < 8: the 16 KB would cap
  // the waves per CU for nothing -- those kernels reduce every node in registers like the
  // multi-wave ones)
  ESTMAF_SHARED(NI, BLOCK);
  uint64_t site;
  const double* tile_col = nullptr;  // TILE: posterior of individual i at tile_col[post_ind_off(i)]
  if constexpr (TILE) {
    // blk0: first block of the launch's part of the grid (whole tile rows: blk0 % 64 == 0)
    const uint64_t b = blockIdx.x + blk0, x = b & 7, k = b >> 3;
    const uint64_t q = ((k >> 3) << 6) + (x << 3) + (k & 7);
    const uint64_t tile_row = q >> 6, l = q & 63;  // tile_row = c * T + t
    const uint64_t c = tile_row / tile_T, t = tile_row - c * tile_T;
    site = (c * 64 + l) * tile_T + t;
    if (site >= S_own) return;  // padding of the interleaved layout
    tile_col = marg_blocks + post_lane_off(tile_row, l, I_tot);
  } else {
    site = blockIdx.x;
  }
  estmaf_site<NI, BLOCK, TILE>(gl, marg_blocks, S_own, I_tot, I_blk, freq_out, redo, status, state,
                               state_stride, 1, n_exact, allow_build, site, tile_col, xch, nodebuf,
                               xnode);
}

// The sites k_fast_estmaf_interp handed back (status EST_EXACT) resume from `state`.  They are
// few: instead of a workgroup per site that finds nothing to do, a workgroup reads the status
// of 64 sites at a time and takes the flagged ones in turn (tile rows [row0, row1) only).
template <int NI, int BLOCK, bool TILE>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2)))
k_fast_estmaf_resume(const GlView gl, const double* __restrict__ marg_blocks,
                     uint64_t S_own, uint64_t I_tot, uint64_t I_blk, uint64_t tile_T,
                     double* __restrict__ freq_out, uint8_t* __restrict__ redo,
                     uint8_t* __restrict__ status, double* __restrict__ state,
                     uint64_t state_stride, int n_exact, int allow_build, uint64_t row0,
                     uint64_t row1) {
  ESTMAF_SHARED(NI, BLOCK);
  const int lane = threadIdx.x & 63;
  for (uint64_t base = (uint64_t)blockIdx.x * 64; base < S_own; base += (uint64_t)gridDim.x * 64) {
    const uint64_t s = base + lane;
    const bool need = s < S_own && in_tile_rows(s, TILE ? tile_T : 0, row0, row1) &&
                      status[s] == EST_EXACT;
    uint64_t mask = __ballot(need);  // the same in every wave of the workgroup
    if constexpr (BLOCK > 64) __syncthreads();  // ... all have read before anyone writes a status
    while (mask) {
      const uint64_t site = base + (uint64_t)__builtin_ctzll(mask);
      mask &= mask - 1;
      const double* tile_col = TILE ? estmaf_tile_col(marg_blocks, site, tile_T, I_tot) : nullptr;
      estmaf_site<NI, BLOCK, TILE>(gl, marg_blocks, S_own, I_tot, I_blk, freq_out, redo, status,
                                   state, state_stride, 0, n_exact, allow_build, site, tile_col, xch,
                                   nodebuf, xnode);
      if constexpr (BLOCK > 64) __syncthreads();  // the shared buffers serve the next site
    }
  }
}

// Small cohorts: a 64-lane wave per site leaves most lanes empty below ~128 individuals and
// pays the per-pass bookkeeping for one site only.  Here a wave holds FOUR sites, one per
// 16-lane DPP row (individual i of the site in lane i % 16, slot i / 16; up to 16 NI = 128
// individuals), and every reduction is the four in-row DPP steps, which leave the row's total
// in all of its lanes -- no cross-lane reads, no LDS.  Same recursion, same interval logic,
// same hand-over to k_fast_estmaf_interp as k_fast_estmaf; the rows of a wave run their own
// sites independently (a row whose site is finished idles).
// The lanes of a row take their decisions (stop, hand over, build) each for itself from these
// totals, so the totals must be the SAME BITS in all 16 lanes: every step adds the two
// partners' values, a + b in one lane and b + a in the other.  That only holds for plain
// additions -- were the first one contracted with a multiplication that produced the argument
// (fma(x, y, partner's rounded x'y') here, fma(x', y', rounded xy) there) the partners would
// differ in the last bit, and a row's lanes would part ways at a threshold.  __dadd_rn is
// never contracted.
__device__ __forceinline__ double row_sum(double v) {
  v = __dadd_rn(v, dpp_move<0xB1>(v));   // quad_perm [1,0,3,2]
  v = __dadd_rn(v, dpp_move<0x4E>(v));   // quad_perm [2,3,0,1]
  v = __dadd_rn(v, dpp_move<0x141>(v));  // row_half_mirror
  v = __dadd_rn(v, dpp_move<0x140>(v));  // row_mirror: every lane holds its 16-lane row total
  return v;
}

// four sites on the four 16-lane rows of a wave; a row with done = true idles (site and
// tile_col must still be readable)
template <int NI, bool TILE>
__device__ __forceinline__ void estmaf_rows_sites(
    const GlView& gl, const double* __restrict__ marg_blocks, uint64_t S_own, uint64_t I_tot,
    uint64_t I_blk, double* __restrict__ freq_out, uint8_t* __restrict__ redo,
    uint8_t* __restrict__ status, double* __restrict__ state, uint64_t state_stride, int fresh,
    int n_exact, int allow_build, uint64_t site, const double* __restrict__ tile_col, bool done) {
  static_assert(EN <= 16, "a row's lanes hold the interval's node sums");
  // Control flow is kept WAVE-UNIFORM: the rows of a wave are at different points of their
  // recursions (one hands its site over while another still needs exact passes), but every
  // DPP reduction runs with all 64 lanes enabled -- a finished row computes along on its stale
  // values and ignores the results -- and the per-row decisions are applied under `!done`.
  const int lane = threadIdx.x, j = lane & 15;
  const uint64_t cell_s = site * I_tot;

  double sA[NI], sb[NI], sC[NI], u0[NI], nC[NI], fc[NI];
  double tF_acc = 0;
  {
    const bool one_block = (I_blk == I_tot);
    const uint32_t ib = (uint32_t)I_blk;
    double r0[NI], r1[NI], r2[NI], rF[NI];
    uint64_t ic[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const uint64_t i = (uint64_t)j + 16 * k;
      ic[k] = i < I_tot ? i : I_tot - 1;
      gl_fetch(gl, cell_s + ic[k], r0[k], r1[k], r2[k]);
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      if constexpr (TILE) {
        rF[k] = tile_col[post_ind_off(ic[k])];
      } else if (one_block) {
        rF[k] = marg_blocks[site * I_blk + ic[k]];
      } else {
        const uint32_t q = (uint32_t)ic[k] / ib;
        rF[k] = marg_blocks[((uint64_t)q * S_own + site) * I_blk + ((uint32_t)ic[k] - q * ib)];
      }
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const bool valid = (uint64_t)j + 16 * k < I_tot;
      const double p0 = valid ? r0[k] : 1.0, p1 = valid ? r1[k] : 0.0;  // (see estmaf_site)
      const double p2 = valid ? r2[k] : 0.0, F = valid ? rF[k] : 0.0;
      const double cc = p1 * fma(-2.0, F, 2.0);  // 2 p1 (1 - F): 0 at F = 1, as the reference's exp(-1e15)
      const double n2 = (2 - F) * p2;
      sA[k] = p0;
      sb[k] = fma(F, p0 + p2, cc);
      sC[k] = p2;
      u0[k] = fma(n2, F, cc);
      nC[k] = n2;
      fc[k] = F * cc;
      tF_acc += valid ? 2 - F : 0.0;
    }
  }
  const double tF_sum = row_sum(tF_acc);

  int iters = 0;
  double num = 0, den = 0;
  double pnum = 0.01, pden = 1.0;  // freq = 0.01 (gen_func.cpp:976)
  if (!fresh && !done) {
    num = state[0 * state_stride + site];
    den = state[1 * state_stride + site];
    pnum = state[2 * state_stride + site];
    pden = state[3 * state_stride + site];
    iters = (int)state[4 * state_stride + site];
  }
  bool built = !allow_build;
  int n_before = n_exact;
  bool check = false;
  double mid = 1, half = 0.5, my_gn = 0, my_gd = 0, rprev = 0;
  auto lane_sums = [&](double r, double& pn, double& pd) {
    pn = 0;
    pd = 0;
    if constexpr (NI >= 4) {
#pragma unroll
      for (int k0 = 0; k0 < NI; k0 += 4) {
        double sm[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sm[k] = fma(r, fma(r, sC[k0 + k], sb[k0 + k]), sA[k0 + k]);
        const double p01 = sm[0] * sm[1], p23 = sm[2] * sm[3];
        const double R = rcp_nr(p01 * p23);
        const double r01 = R * p23, r23 = R * p01;
        const double inv0 = r01 * sm[1], inv1 = r01 * sm[0];
        const double inv2 = r23 * sm[3], inv3 = r23 * sm[2];
        pn = fma(fma(nC[k0], r, u0[k0]), inv0, pn);
        pd = fma(fc[k0], inv0, pd);
        pn = fma(fma(nC[k0 + 1], r, u0[k0 + 1]), inv1, pn);
        pd = fma(fc[k0 + 1], inv1, pd);
        pn = fma(fma(nC[k0 + 2], r, u0[k0 + 2]), inv2, pn);
        pd = fma(fc[k0 + 2], inv2, pd);
        pn = fma(fma(nC[k0 + 3], r, u0[k0 + 3]), inv3, pn);
        pd = fma(fc[k0 + 3], inv3, pd);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const double inv = rcp_nr(fma(r, fma(r, sC[k], sb[k]), sA[k]));
        pn = fma(fma(nC[k], r, u0[k]), inv, pn);
        pd = fma(fc[k], inv, pd);
      }
    }
  };
  for (;;) {
    // ---- one exact pass of every row (all lanes) ----
    const double r = pnum * rcp_nr2(pden - pnum);
    double pn, pd;
    lane_sums(r, pn, pd);
    const double sn = row_sum(pn), sd = row_sum(pd);
    bool interp_ok = false;
    if (__builtin_amdgcn_ballot_w64(check && !done) != 0) {
      // the pass after a build: exact sums in hand, compare the interpolant (rows that did
      // not just build compute along and ignore the outcome)
      const int nj = j < EN ? j : 0;
      const double t = (r - mid) / (r + mid) - half * kChebC[nj];
      const double q = (j < EN) ? kChebW[nj] / t : 0.0;
      const double Sq = row_sum(q);
      const double bn = row_sum(q * my_gn) / Sq, bd = row_sum(q * my_gd) / Sq;
      if (check && !done)
        interp_ok = fabs(bn - sn) <= EST_TOL * fabs(sn) && fabs(bd - sd) <= EST_TOL * fabs(sd);
      check = false;
    }
    // ---- the recursion and the row's decisions ----
    bool want_build = false;
    if (!done) {
      num = fma(r, sn, num);
      den = fma(r, sd, den + tF_sum);
      const double lhs = fabs(fma(pnum, den, -(num * pden))), thr = kEPS * (den * pden);
      const bool again = (lhs > thr) && (iters++ < 100);
      rprev = r;
      pnum = num;
      pden = den;
      if (!again) {
        if (j == 0) {
          const double freq = num / den;
          const bool ok = freq >= 0 && freq < 1;
          freq_out[site] = freq;
          redo[site] = ok ? 0 : 1;
          status[site] = EST_DONE;
        }
        done = true;
      } else if (interp_ok) {  // hand the site to k_fast_estmaf_interp
        if (j < EN) {
          state[(EST_SCALARS + j) * state_stride + site] = my_gn;
          state[(EST_SCALARS + EN + j) * state_stride + site] = my_gd;
        }
        if (j == 0) {
          state[0 * state_stride + site] = num;
          state[1 * state_stride + site] = den;
          state[2 * state_stride + site] = pnum;
          state[3 * state_stride + site] = pden;
          state[4 * state_stride + site] = (double)iters;
          state[5 * state_stride + site] = mid;
          state[6 * state_stride + site] = half;
          state[7 * state_stride + site] = tF_sum;
          status[site] = EST_INTERP;
        }
        done = true;
      } else if (!built && --n_before <= 0) {  // see k_fast_estmaf for the reasoning
        const double m_est = (double)iters * (sqrt(lhs / thr) - 1.0);
        const double rn = pnum * rcp_nr2(pden - pnum);
        const double step = fabs(rn - rprev);
        const double reach = (rn >= rprev) ? EST_DMAX * rn : EST_DMAX / (1 + EST_DMAX) * rn;
        const bool fits = (double)iters * step <= EST_FIT * reach;
        if (!fits && iters < EST_KMAX && m_est >= EST_MIN_GAIN) {
          n_before = 1;
        } else {
          built = true;
        }
        if (built && m_est >= EST_MIN_GAIN && 100 - iters >= EST_MIN_GAIN) {
          const double g = fmin(EST_DMAX, fmax(EST_MULT * step / rn, 1e-3));
          double lo, hi;
          if (rn >= rprev) {
            lo = rn * (1 - EST_BACK * g);
            hi = rn * (1 + g);
          } else {
            lo = rn / (1 + g);
            hi = rn * (1 + EST_BACK * g);
          }
          const double a = sqrt(lo * hi), h = (hi - a) / (hi + a);
          if (h > 0 && lo > 0 && hi < 1e300) {
            mid = a;
            half = h;
            want_build = true;
          }
        }
      }
    }
    // ---- the interval's node sums, for the rows that build (all lanes compute) ----
    if (__builtin_amdgcn_ballot_w64(want_build) != 0) {
#pragma unroll 1
      for (int nd = 0; nd < EN; ++nd) {
        const double tn = half * kChebC[nd];
        double qn, qd;
        lane_sums(mid * (1 + tn) * rcp_nr2(1 - tn), qn, qd);
        const double gn = row_sum(qn), gd = row_sum(qd);
        if (want_build && j == nd) {  // lane nd of the row keeps node nd
          my_gn = gn;
          my_gd = gd;
        }
      }
      if (want_build) check = true;
    }
    if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
  }
}

template <int NI, bool TILE>
__global__ void __launch_bounds__(64)
k_fast_estmaf_rows(const GlView gl, const double* __restrict__ marg_blocks, uint64_t S_own,
                   uint64_t I_tot, uint64_t I_blk, uint64_t tile_T, double* __restrict__ freq_out,
                   uint8_t* __restrict__ redo, uint8_t* __restrict__ status,
                   double* __restrict__ state, uint64_t state_stride, int n_exact,
                   int allow_build) {
  const int lane = threadIdx.x, row = lane >> 4;
  uint64_t site;
  const double* tile_col = nullptr;
  if constexpr (TILE) {
    // as in k_fast_estmaf, XCD x gets the sites l = 8x..8x+7 of a tile row -- here in two
    // consecutive workgroups of four sites each
    const uint64_t b = blockIdx.x, x = b & 7, k = b >> 3;
    const uint64_t q = ((k >> 1) << 6) + (x << 3) + ((k & 1) << 2) + row;
    const uint64_t tile_row = q >> 6, l = q & 63;
    const uint64_t c = tile_row / tile_T, t = tile_row - c * tile_T;
    site = (c * 64 + l) * tile_T + t;
    tile_col = marg_blocks + post_lane_off(tile_row, l, I_tot);
  } else {
    site = (uint64_t)blockIdx.x * 4 + row;
  }
  bool done = site >= S_own;          // padding of the layout / past the end
  if (done) {                         // read something valid, write nothing
    site = 0;
    if constexpr (TILE) tile_col = marg_blocks;
  }
  if (__builtin_amdgcn_ballot_w64(!done) == 0) return;  // nothing to do in this wave
  estmaf_rows_sites<NI, TILE>(gl, marg_blocks, S_own, I_tot, I_blk, freq_out, redo, status, state,
                              state_stride, 1, n_exact, allow_build, site, tile_col, done);
}

// resuming sites (see k_fast_estmaf_resume): the wave reads 64 statuses at a time and gives the
// flagged sites to its rows four at a time
template <int NI, bool TILE>
__global__ void __launch_bounds__(64)
k_fast_estmaf_rows_resume(const GlView gl, const double* __restrict__ marg_blocks, uint64_t S_own,
                          uint64_t I_tot, uint64_t I_blk, uint64_t tile_T,
                          double* __restrict__ freq_out, uint8_t* __restrict__ redo,
                          uint8_t* __restrict__ status, double* __restrict__ state,
                          uint64_t state_stride, int n_exact, int allow_build) {
  const int lane = threadIdx.x, row = lane >> 4;
  for (uint64_t base = (uint64_t)blockIdx.x * 64; base < S_own; base += (uint64_t)gridDim.x * 64) {
    const uint64_t s = base + lane;
    uint64_t mask = __ballot(s < S_own && status[s] == EST_EXACT);
    while (mask) {
      uint64_t site = 0;
      bool done = true;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (mask) {
          const uint64_t cand = base + (uint64_t)__builtin_ctzll(mask);
          mask &= mask - 1;
          if (r == row) {
            site = cand;
            done = false;
          }
        }
      }
      const double* tile_col =
          TILE ? (done ? marg_blocks : estmaf_tile_col(marg_blocks, site, tile_T, I_tot)) : nullptr;
      estmaf_rows_sites<NI, TILE>(gl, marg_blocks, S_own, I_tot, I_blk, freq_out, redo, status,
                                  state, state_stride, 0, n_exact, allow_build, site, tile_col, done);
    }
  }
}

// The passes between a checked interval and either the end of the loop or the point
// (a launch may cover only the tile rows [row0, row1) of the E-step's layout: see fast_estmaf)
// where exact evaluation is needed again: one lane per site.  The EN node values of each sum
// become Chebyshev coefficients once (c_k = 2/EN sum_j f_j T_k(x_j), the T_k by their
// recurrence: 3 EN^2 instructions), and a pass is then two Clenshaw recurrences of EN steps
// -- no divisions, where the barycentric formula spends one per node and pass (the ~80 passes
// of a site cost 3x less; the checked interpolant is the same polynomial).
__global__ void __launch_bounds__(256)
k_fast_estmaf_interp(uint64_t S_own, double* __restrict__ freq_out, uint8_t* __restrict__ redo,
                     uint8_t* __restrict__ status, double* __restrict__ state,
                     uint64_t state_stride, uint64_t tile_T, uint64_t row0, uint64_t row1) {
  const uint64_t site = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (site >= S_own || !in_tile_rows(site, tile_T, row0, row1) || status[site] != EST_INTERP)
    return;
  double num = state[0 * state_stride + site], den = state[1 * state_stride + site];
  double pnum = state[2 * state_stride + site], pden = state[3 * state_stride + site];
  int iters = (int)state[4 * state_stride + site];
  const double mid = state[5 * state_stride + site], half = state[6 * state_stride + site];
  const double tF_sum = state[7 * state_stride + site];
  double gn[EN], gd[EN];
#pragma unroll
  for (int j = 0; j < EN; ++j) {
    gn[j] = state[(EST_SCALARS + j) * state_stride + site];
    gd[j] = state[(EST_SCALARS + EN + j) * state_stride + site];
  }
  // node values -> Chebyshev coefficients (in place would need a second array anyway)
  double cn[EN], cd[EN];
#pragma unroll
  for (int k = 0; k < EN; ++k) cn[k] = cd[k] = 0;
#pragma unroll
  for (int j = 0; j < EN; ++j) {
    const double x = kChebC[j], fn = gn[j] * (2.0 / EN), fd = gd[j] * (2.0 / EN);
    double t0 = 1.0, t1 = x;
    cn[0] += fn;
    cd[0] += fd;
#pragma unroll
    for (int k = 1; k < EN; ++k) {
      cn[k] = fma(fn, t1, cn[k]);
      cd[k] = fma(fd, t1, cd[k]);
      const double t2 = fma(2 * x, t1, -t0);
      t0 = t1;
      t1 = t2;
    }
  }
  const double inv_half = 1.0 / half;
  uint8_t st = EST_EXACT;
  for (;;) {
    const double r = pnum * rcp_nr2(pden - pnum);  // the expression of k_fast_estmaf
    const double tt = (r - mid) * rcp_nr2(r + mid);  // the interpolation variable (see the build)
    if (!(fabs(tt) <= half)) break;                // left the interval (or not finite)
    // Clenshaw: p(x) = c_0 / 2 + sum_{k >= 1} c_k T_k(x),  x = tt / half in [-1, 1]
    const double x2 = 2 * (tt * inv_half);
    double bn1 = 0, bn2 = 0, bd1 = 0, bd2 = 0;
#pragma unroll
    for (int k = EN - 1; k >= 1; --k) {
      const double bn0 = fma(x2, bn1, cn[k] - bn2), bd0 = fma(x2, bd1, cd[k] - bd2);
      bn2 = bn1;
      bn1 = bn0;
      bd2 = bd1;
      bd1 = bd0;
    }
    const double sn = fma(0.5 * x2, bn1, 0.5 * cn[0] - bn2);
    const double sd = fma(0.5 * x2, bd1, 0.5 * cd[0] - bd2);
    const double num2 = fma(r, sn, num), den2 = fma(r, sd, den + tF_sum);
    const double lhs = fabs(fma(pnum, den2, -(num2 * pden))), thr = kEPS * (den2 * pden);
    if (!(fabs(lhs - thr) >= EST_GUARD * thr)) break;  // too close to call (or not finite)
    num = num2;
    den = den2;
    const bool again = (lhs > thr) && (iters++ < 100);
    pnum = num;
    pden = den;
    if (!again) {
      st = EST_DONE;
      break;
    }
  }
  if (st == EST_DONE) {
    const double freq = num / den;
    const bool ok = freq >= 0 && freq < 1;
    freq_out[site] = freq;
    redo[site] = ok ? 0 : 1;
  } else {
    state[0 * state_stride + site] = num;
    state[1 * state_stride + site] = den;
    state[2 * state_stride + site] = pnum;
    state[3 * state_stride + site] = pden;
    state[4 * state_stride + site] = (double)iters;
  }
  status[site] = st;
}

// ---- est_maf for CALLED genotypes (packed handles): the per-pass sums in closed form ----
// A called genotype (--call_geno, called-genotype input: gen_func.cpp:886-914,
// read_data.cpp:88-98) has linear likelihoods (1,0,0), (0,1,0), (0,0,1) or -- missing -- (u,u,u),
// so the genotype posterior of est_maf's pass (calc_HWE + post_prob, gen_func.cpp:984-1000) is
// a unit vector whatever the frequency, except for missing cells, where it is HWE itself:
//   genotype 0:  num += 0         den += 2 - F
//   genotype 1:  num += 1         den += 2                 (F < 1; at F = 1 the reference's weights
//                                                           all vanish: the site is redone in its
//                                                           log-space order, k_fast_estmaf_stream)
//   genotype 2:  num += 2 - F     den += 2 - F
//   missing:     num += h1 + h2 (2 - F),  den += 2 h1 + (h0 + h2)(2 - F),  with
//                h0 = (1-f)^2 + f(1-f)F, h1 = 2 f(1-f)(1-F), h2 = f^2 + f(1-f)F -- polynomials
//                in f whose coefficients are sums over the site's missing individuals of
//                (1-F), (2-F) and F(2-F).
// The whole <= 101-pass recursion of a site is therefore a SCALAR recursion on five sums over
// its individuals: one sweep over the 2-bit codes and the posteriors (8.25 B per cell, bound by
// HBM) leaves the sums, and one LANE per site runs the reference's passes -- same recursion,
// same pass count, same stopping rule as k_fast_estmaf, nothing interpolated.  (The general
// kernel spends 17 evaluations of every individual per site, 17.5 ps per cell at 5000
// individuals; this is one load of every cell.)
constexpr int ESTC_A0 = 0, ESTC_B0 = 1, ESTC_M0 = 2, ESTC_M1 = 3, ESTC_M2 = 4;

template <bool TILE>
__global__ void __launch_bounds__(64)
k_fast_estmaf_called_sums(const GlView gl, const double* __restrict__ marg_blocks, uint64_t S_own,
                          uint64_t I_tot, uint64_t I_blk, uint64_t tile_T,
                          uint8_t* __restrict__ redo, double* __restrict__ state,
                          uint64_t state_stride, uint64_t blk0) {
  const int lane = threadIdx.x;
  uint64_t site;
  const double* tile_col = nullptr;
  if constexpr (TILE) {  // the blockIdx -> site map of k_fast_estmaf<.., TILE>
    const uint64_t b = blockIdx.x + blk0, x = b & 7, k = b >> 3;
    const uint64_t q = ((k >> 3) << 6) + (x << 3) + (k & 7);
    const uint64_t tile_row = q >> 6, l = q & 63;
    const uint64_t c = tile_row / tile_T, t = tile_row - c * tile_T;
    site = (c * 64 + l) * tile_T + t;
    if (site >= S_own) return;
    tile_col = marg_blocks + post_lane_off(tile_row, l, I_tot);
  } else {
    site = blockIdx.x;
  }
  const uint64_t cell_s = gl.cell0 + site * I_tot;
  const bool one_block = (I_blk == I_tot);
  double n1 = 0, s2 = 0, s02 = 0, m0 = 0, m1 = 0, m2 = 0;
  bool bad = false;
  auto posterior = [&](uint64_t i) -> double {
    if constexpr (TILE) return tile_col[post_ind_off(i)];
    if (one_block) return marg_blocks[site * I_blk + i];
    const uint64_t q = i / I_blk;  // rank blocks [I_tot / I_blk][S_own][I_blk]
    return marg_blocks[(q * S_own + site) * I_blk + (i - q * I_blk)];
  };
  auto take = [&](uint32_t code, double F) {
    const double tF = 2 - F;
    const bool g1 = code == 1, g2 = code == 2, g3 = code == 3;
    n1 += g1 ? 1.0 : 0.0;
    bad |= g1 && !(F < 1);
    s2 += g2 ? tF : 0.0;
    s02 += (code == 0 || g2) ? tF : 0.0;
    m0 += g3 ? 1 - F : 0.0;
    m1 += g3 ? tF : 0.0;
    m2 += g3 ? F * tF : 0.0;
  };
  uint64_t i = lane;
  for (; i + 192 < I_tot; i += 256) {  // four loads of each kind in flight
    double F[4];
    uint32_t cd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      F[j] = posterior(i + 64 * j);
      cd[j] = gl_code(gl.codes, cell_s + i + 64 * j);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) take(cd[j], F[j]);
  }
  for (; i < I_tot; i += 64) take(gl_code(gl.codes, cell_s + i), posterior(i));
  const double A0 = wave_sum(n1 + s2), B0 = wave_sum(2 * n1 + s02);
  m0 = wave_sum(m0);
  m1 = wave_sum(m1);
  m2 = wave_sum(m2);
  const bool any_bad = __ballot(bad) != 0;
  if (lane == 0) {
    state[ESTC_A0 * state_stride + site] = A0;
    state[ESTC_B0 * state_stride + site] = B0;
    state[ESTC_M0 * state_stride + site] = m0;
    state[ESTC_M1 * state_stride + site] = m1;
    state[ESTC_M2 * state_stride + site] = m2;
    redo[site] = any_bad ? 1 : 0;
  }
}

// the passes themselves (gen_func.cpp:976-1006): one lane per site
__global__ void __launch_bounds__(256)
k_fast_estmaf_called_passes(uint64_t S_own, double* __restrict__ freq_out, uint8_t* __restrict__ redo,
                            uint8_t* __restrict__ status, const double* __restrict__ state,
                            uint64_t state_stride, uint64_t tile_T, uint64_t row0, uint64_t row1) {
  const uint64_t site = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (site >= S_own || !in_tile_rows(site, tile_T, row0, row1)) return;
  status[site] = EST_DONE;
  if (redo[site]) return;
  const double A0 = state[ESTC_A0 * state_stride + site], B0 = state[ESTC_B0 * state_stride + site];
  const double m0 = state[ESTC_M0 * state_stride + site], m1 = state[ESTC_M1 * state_stride + site];
  const double m2 = state[ESTC_M2 * state_stride + site];
  int iters = 0;
  double num = 0, den = 0, pnum = 0.01, pden = 1.0;  // freq = 0.01 (gen_func.cpp:976)
  for (;;) {
    const double f = pnum / pden, om = 1 - f;
    const double b = f * om, ff = f * f;
    const double miss_n = fma(2 * b, m0, fma(ff, m1, b * m2));
    const double miss_d = fma(4 * b, m0, fma(fma(om, om, ff), m1, 2 * b * m2));
    num += A0 + miss_n;
    den += B0 + miss_d;
    // |prev - freq| > EPSILON (gen_func.cpp:1006), cross-multiplied as in k_fast_estmaf
    const double lhs = fabs(fma(pnum, den, -(num * pden))), thr = kEPS * (den * pden);
    const bool again = (lhs > thr) && (iters++ < 100);
    pnum = num;
    pden = den;
    if (!again) break;
  }
  const double freq = num / den;
  const bool ok = freq >= 0 && freq < 1;
  freq_out[site] = freq;
  redo[site] = ok ? 0 : 1;
}

// any number of individuals: re-reads the (L2-resident) site row every pass.  One wave per
// site; a wave looks at the flags of 64 sites at a time (normally none is set: the launch is
// then a few thousand waves reading a cache line each, whatever the number of sites) and
// takes the flagged ones in turn.
__device__ void estmaf_stream_site(const GlView& gl, const double* __restrict__ marg_blocks,
                                   uint64_t S_own, uint64_t I_tot, uint64_t I_blk, uint64_t tile_T,
                                   double* __restrict__ freq_out, uint64_t site, int lane) {
  const uint64_t cell_s = site * I_tot;
  // tile_T != 0: posteriors in the tile-major layout (see k_fast_estmaf<.., TILE>)
  const uint64_t tj = tile_T ? site / tile_T : 0;  // lane-chunk c*64 + l; t = site - tj*T
  const double* trow =
      tile_T ? marg_blocks + post_lane_off((tj >> 6) * tile_T + (site - tj * tile_T), tj & 63, I_tot)
             : nullptr;
  int iters = 0;
  double num = 0, den = 0, freq = 0.01, prev;
  bool again;
  do {
    prev = freq;
    const double om = 1 - freq;
    const double b = om * freq;
    const double A = om * om, Cq = freq * freq;
    double pn = 0, pd = 0;
    for (uint64_t i = lane; i < I_tot; i += 64) {
      const double F = trow ? trow[post_ind_off(i)]
                            : marg_blocks[((i / I_blk) * S_own + site) * I_blk + (i % I_blk)];
      const double bF = b * F;
      const double h0 = A + bF, h2 = Cq + bF;
      const double h1 = (F == 1) ? 0.0 : (2 * b - 2 * bF);
      double p0, p1, p2;
      gl_fetch(gl, cell_s + i, p0, p1, p2);
      const double w0 = p0 * h0, w1 = p1 * h1, w2 = p2 * h2;
      const double sum = w0 + w1 + w2;
      const double tF = 2 - F;
      if (sum > 0) {
        const double inv = 1.0 / sum;
        pn += fma(w2, tF, w1) * inv;
        pd += fma(w0 + w2, tF, 2 * w1) * inv;
      } else {
        // a called genotype's impossible classes are -1e15 in the reference (read_data.cpp:21),
        // not -inf: the packed view knows it holds such cells
        double lg[3] = {log(p0), log(p1), log(p2)};
        if (!gl.dense)
          for (int k = 0; k < 3; ++k)
            if (lg[k] == -__builtin_huge_val()) lg[k] = -kINF;
        const double2 tt = estmaf_term_logspace(lg, freq, F);
        pn += tt.x;
        pd += tt.y;
      }
    }
    num += wave_sum(pn);
    den += wave_sum(pd);
    freq = num / den;
    again = (fabs(prev - freq) > kEPS) && (iters++ < 100);
  } while (again);
  if (lane == 0) freq_out[site] = freq;
}

__global__ void __launch_bounds__(256)
k_fast_estmaf_stream(const GlView gl, const double* __restrict__ marg_blocks,
                     uint64_t S_own, uint64_t I_tot, uint64_t I_blk, uint64_t tile_T,
                     double* __restrict__ freq_out, const uint8_t* __restrict__ redo,
                     uint64_t row0, uint64_t row1) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
  if (!redo) {  // every site (more individuals than the register kernels hold): a wave each
    for (uint64_t site = wave; site < S_own; site += n_waves)
      if (in_tile_rows(site, tile_T, row0, row1))
        estmaf_stream_site(gl, marg_blocks, S_own, I_tot, I_blk, tile_T, freq_out, site, lane);
    return;
  }
  for (uint64_t base = wave * 64; base < S_own; base += n_waves * 64) {
    const uint64_t s = base + lane;
    const bool need = s < S_own && in_tile_rows(s, tile_T, row0, row1) && redo[s];
    uint64_t mask = __ballot(need);
    while (mask) {
      const int b = __builtin_ctzll(mask);
      mask &= mask - 1;
      estmaf_stream_site(gl, marg_blocks, S_own, I_tot, I_blk, tile_T, freq_out, base + b, lane);
    }
  }
}

struct SwitchName {
  const char* name;
  int Switches::*field;
};
const SwitchName kSwitches[] = {
    {"pipeline", &Switches::pipeline}, {"no_bg", &Switches::no_bg},
    {"bg_parts", &Switches::bg_parts}, {"no_fuse", &Switches::no_fuse},
    {"eager_emission", &Switches::eager_emission}, {"estmaf_interp", &Switches::estmaf_interp},
    {"estmaf_sitemajor", &Switches::estmaf_sitemajor}, {"estmaf_no_rows", &Switches::estmaf_no_rows},
    {"estmaf_no_called", &Switches::estmaf_no_called},
    {"no_xdeg2", &Switches::no_xdeg2},
    {"fast_c", &Switches::fast_c}, {"exact_serial", &Switches::exact_serial},
    {"estmaf_exact_lanes", &Switches::estmaf_exact_lanes}, {"exact_bg_waves", &Switches::exact_bg_waves},
    {"exact_bg_depth", &Switches::exact_bg_depth},
    {"spin_sync", &Switches::spin_sync}, {"timing", &Switches::timing},
    {"debug_modes", &Switches::debug_modes}};

template <typename T>
bool dalloc(T** p, size_t n) {
  if (n == 0) n = 1;
  return hipMalloc((void**)p, n * sizeof(T)) == hipSuccess;
}

}  // namespace

// ---------------------------------------------------------------------------
Switches Switches::from_env() {
  Switches sw;
  for (const auto& k : kSwitches) {
    std::string env = "NGHMM_";
    for (const char* c = k.name; *c; ++c) env += (char)std::toupper((unsigned char)*c);
    if (const char* v = std::getenv(env.c_str())) {
      char* end = nullptr;
      const long n = std::strtol(v, &end, 10);
      sw.*(k.field) = (end != v) ? (int)n : 1;  // set without a number: on
    }
  }
  return sw;
}

bool Switches::set(const char* name, long value) {
  for (const auto& k : kSwitches)
    if (std::strcmp(k.name, name) == 0) {
      this->*(k.field) = (int)value;
      return true;
    }
  return false;
}

// the arrays an EM run writes (everything but the data): emissions, frequency table,
// posteriors, checkpoints, boundary operators
static bool fast_alloc_run_state(FastState& fs) {
  const size_t cells = (size_t)fs.I * fs.Spad;
  const size_t slack = 8 * 64;  // the pipelines read one group past the last lane-chunk
  if (!dalloc(&fs.e_il, cells + slack)) return false;
  if (hipMemset(fs.e_il + cells, 0, slack * sizeof(double)) != hipSuccess) return false;
  if (!dalloc(&fs.base_c, (size_t)fs.I * fs.C)) return false;
  if (!dalloc(&fs.freq_il, (size_t)fs.Spad + slack)) return false;
  if (hipMemset(fs.freq_il + fs.Spad, 0, slack * sizeof(double)) != hipSuccess) return false;
  const size_t post_cells = (size_t)fs.C * fs.T * post_tile_doubles(fs.I);
  if (!dalloc(&fs.post, post_cells)) return false;
  // the reference starts from marg_prob = 0 (parse_args.cpp:403-405): `--freq e` and a
  // `--log` print before the first E-step read the posteriors
  if (hipMemset(fs.post, 0, post_cells * sizeof(double)) != hipSuccess) return false;
  if (!dalloc(&fs.ckpt, cells / CK * 4 + 4 * 64)) return false;  // T is a multiple of CK; slack:
                                                                 // see lkl_run_fd
  if (!dalloc(&fs.lane_ops, (size_t)fs.I * fs.J * 5)) return false;
  if (!dalloc(&fs.bound, (size_t)fs.I * fs.J * 4)) return false;
  fs.e_stale = true;
  return true;
}

bool fast_create(FastState& fs, uint64_t I, uint64_t S, bool packed) {
  fs.I = I;
  fs.S = S;
  fs.packed = packed;
  fs.owns_data = true;
  // Waves per individual.  The objective rounds run 4 waves per SIMD = 4096 at a time and
  // all their waves take equally long, so a launch of n waves wastes the unfilled part of
  // its last batch, and the later rounds launch only the still-active individuals: ~49k waves
  // per 1000 individuals (measured at 1000 x 1M, ms per EM iteration: C = 9: 45, 33: 28.85,
  // 40: 28.3, 48: 28.1-28.3, 56: 28.2, 64: 28.9, 96: 29.0, 128: 29.8 -- beyond ~56 a wave's
  // fixed cost, the operator tree of every point, outweighs the fuller batches).  At least 16
  // sites per lane.
  uint64_t C = (49152 + I - 1) / I;
  if (fs.sw.fast_c >= 1) C = (uint64_t)fs.sw.fast_c;  // tuning knob: waves per individual
  if (C > 256) C = 256;  // (measured at 125 x 1M: 5.50 ms per iteration with 64, 5.30 with 256)
  // at least 32 sites per lane above 64 waves per individual (16 below): a wave's fixed cost
  // -- the 64-lane operator tree of every point -- is that of about a dozen sites
  while (C > 1 && (S + 64 * C - 1) / (64 * C) < (C > 64 ? 32u : 16u)) --C;
  if (C < 1) C = 1;
  if (fs.sw.fast_c < 1) {
    // sites per lane are rounded up to whole groups of 8 (16: packed), which at a few dozen
    // sites per lane pads a lot (100 x 100k: 24.4 -> 32 sites per lane, 31 %): take the count
    // within a quarter below the target that pads least
    const uint64_t q = packed ? 16 : 8;
    auto padded = [&](uint64_t c) { return 64 * c * ((((S + 64 * c - 1) / (64 * c)) + q - 1) / q * q); };
    uint64_t best = C;
    for (uint64_t c = C; c >= 1 && 4 * c >= 3 * C; --c)
      if (padded(c) < padded(best)) best = c;
    C = best;
  }
  fs.C = (uint32_t)C;
  fs.J = 64 * C;
  fs.T = (S + fs.J - 1) / fs.J;
  fs.T = (fs.T + 7) & ~7ull;  // whole prefetch groups: the main loops need no bound checks
  if (packed) fs.T = (fs.T + 15) & ~15ull;  // 16 sites of a lane per code word
  fs.Spad = fs.J * fs.T;
  const size_t cells = (size_t)I * fs.Spad;
  const size_t slack = 8 * 64;  // the pipelines read one group past the last lane-chunk
  if (!dalloc(&fs.pos_il, (size_t)fs.Spad + slack)) return false;
  if (hipMemset(fs.pos_il + fs.Spad, 0, slack * sizeof(double)) != hipSuccess) return false;
  if (!dalloc(&fs.gl_scale_c, (size_t)I * C)) return false;
  if (packed) {
    const size_t words = cells / 16 + slack;
    if (!dalloc(&fs.geno_il, words)) return false;
    if (hipMemset(fs.geno_il, 0, words * sizeof(uint32_t)) != hipSuccess) return false;
    if (!dalloc(&fs.cls_lin, (size_t)12)) return false;
  } else {
    if (!dalloc(&fs.glq_il, (cells + slack) * 2)) return false;
    if (hipMemset(fs.glq_il + cells * 2, 0, slack * 2 * sizeof(double)) != hipSuccess) return false;
    if (!dalloc(&fs.gl_lin, (size_t)I * S * 3)) return false;
  }
  return fast_alloc_run_state(fs);
}

bool fast_create_replica(FastState& fs, const FastState& parent) {
  fs = FastState();
  fs.sw = parent.sw;
  fs.I = parent.I;
  fs.S = parent.S;
  fs.T = parent.T;
  fs.C = parent.C;
  fs.J = parent.J;
  fs.Spad = parent.Spad;
  fs.packed = parent.packed;
  fs.owns_data = false;
  fs.gl_log = parent.gl_log;
  fs.d_pos = parent.d_pos;
  fs.geno_il = parent.geno_il;
  fs.cls_lin = parent.cls_lin;
  fs.u_lin = parent.u_lin;
  fs.called_table = parent.called_table;
  fs.gl_lin = parent.gl_lin;
  fs.pos_il = parent.pos_il;
  fs.glq_il = parent.glq_il;
  fs.gl_scale_c = parent.gl_scale_c;
  fs.dmax_finite = parent.dmax_finite;
  return fast_alloc_run_state(fs);
}

void fast_destroy(FastState& fs) {
  void* run[] = {fs.e_il, fs.base_c, fs.freq_il, fs.post, fs.ckpt, fs.lane_ops, fs.bound, fs.lanes[0].part,
                 fs.lanes[0].grp_dev, fs.lanes[1].part, fs.lanes[1].grp_dev, fs.redo, fs.est_status,
                 fs.est_state, fs.shard.edges};
  for (void* p : run)
    if (p) (void)hipFree(p);
  if (fs.owns_data) {
    void* data[] = {fs.pos_il, fs.glq_il, fs.gl_scale_c, fs.gl_lin, fs.geno_il, fs.cls_lin};
    for (void* p : data)
      if (p) (void)hipFree(p);
  }
  fs = FastState();
}

void fast_exp(hipStream_t st, const double* d_in, double* d_out, uint64_t n) {
  if (n) hipLaunchKernelGGL(k_fast_exp, dim3(16384), dim3(256), 0, st, d_in, d_out, n);
}

GlView fast_gl_lin(const FastState& fs) {
  return fs.packed ? gl_packed(fs.gl_log.codes, fs.cls_lin, fs.gl_log.cell0) : gl_dense(fs.gl_lin);
}

bool fast_load(FastState& fs, hipStream_t st, const GlView& gl_log, const double* d_pos) {
  fs.gl_log = gl_log;
  fs.d_pos = d_pos;
  fs.e_stale = true;
  if (fs.packed) {
    if (gl_log.dense) return false;
    // linear class table = exp of the prepared log table, as gl_lin is of the dense GL
    fast_exp(st, gl_log.table, fs.cls_lin, 12);
    double t[12];
    if (hipMemcpyAsync(t, fs.cls_lin, sizeof t, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    if (hipStreamSynchronize(st) != hipSuccess) return false;
    fs.u_lin = t[9];
    // the four classes of a called genotype: unit rows and a uniform one (est_maf's closed form)
    fs.called_table = t[0] == 1 && t[1] == 0 && t[2] == 0 && t[3] == 0 && t[4] == 1 && t[5] == 0 &&
                      t[6] == 0 && t[7] == 0 && t[8] == 1 && t[9] > 0 && t[10] == t[9] && t[11] == t[9];
    hipLaunchKernelGGL(k_fast_geno_interleave, dim3(8192), dim3(256), 0, st, gl_log.codes,
                       gl_log.cell0, fs.I, fs.S, fs.T, fs.C, fs.geno_il);
  } else {
    if (!gl_log.dense) return false;
    const double* lg = gl_log.dense + gl_log.cell0 * 3;
    fast_exp(st, lg, fs.gl_lin, fs.I * fs.S * 3);
    const uint64_t n_it = (fs.I + 31) / 32;
    hipLaunchKernelGGL(k_fast_gl_interleave, dim3((unsigned)((uint64_t)fs.C * fs.T * n_it)),
                       dim3(256), 0, st, lg, fs.I, fs.S, fs.T, fs.C,
                       reinterpret_cast<double2*>(fs.glq_il));
  }
  hipLaunchKernelGGL(k_fast_gl_scale, dim3((unsigned)((uint64_t)fs.C * ((fs.I + 31) / 32))), dim3(256),
                     0, st, gl_log, fs.I, fs.S, fs.T, fs.C, fs.gl_scale_c);
  hipLaunchKernelGGL(k_fast_pos_interleave, dim3(1024), dim3(256), 0, st, d_pos, fs.S, fs.T, fs.C,
                     fs.pos_il);
  // the largest finite distance decides when the objective kernel may use its
  // small-argument exp for the alpha +- eh probes
  unsigned long long* d_m = reinterpret_cast<unsigned long long*>(fs.bound);
  if (hipMemsetAsync(d_m, 0, sizeof(unsigned long long), st) != hipSuccess) return false;
  hipLaunchKernelGGL(k_fast_max_finite, dim3(256), dim3(256), 0, st, d_pos, fs.S, d_m);
  unsigned long long bits = 0;
  if (hipMemcpyAsync(&bits, d_m, sizeof bits, hipMemcpyDeviceToHost, st) != hipSuccess) return false;
  if (hipStreamSynchronize(st) != hipSuccess) return false;
  std::memcpy(&fs.dmax_finite, &bits, sizeof bits);
  return hipGetLastError() == hipSuccess;
}

// After an allele-frequency update: the interleaved frequency table (and the MAF check);
// the emissions themselves are refreshed either by fast_refresh_emissions or, for free, by
// the next forward walk (fast_lkl_launch with fresh emissions).
bool fast_refresh_freq_table(FastState& fs, hipStream_t st, const double* d_freq, int* d_flags) {
  hipLaunchKernelGGL(k_fast_freq_interleave, dim3(1024), dim3(256), 0, st, d_freq, fs.S, fs.T, fs.C,
                     fs.freq_il, d_flags);
  fs.e_stale = true;
  return hipGetLastError() == hipSuccess;
}

static LklArrays lkl_arrays(const FastState& fs) {
  return LklArrays{fs.e_il, fs.pos_il, reinterpret_cast<const double2*>(fs.glq_il), fs.freq_il,
                   fs.e_il,  fs.geno_il, fs.u_lin, fs.base_c, fs.gl_scale_c};
}

bool fast_refresh_emissions(FastState& fs, hipStream_t st, const double* d_freq, int* d_flags) {
  // from the frequencies as they are now (the table may predate a parameter upload)
  if (!fast_refresh_freq_table(fs, st, d_freq, d_flags)) return false;
  const dim3 grid((unsigned)(fs.I * fs.C)), block(64);
  if (fs.packed)
    hipLaunchKernelGGL((k_fast_refresh<SRC_FRESH_PACKED>), grid, block, 0, st, lkl_arrays(fs), fs.T,
                       fs.C);
  else
    hipLaunchKernelGGL((k_fast_refresh<SRC_FRESH>), grid, block, 0, st, lkl_arrays(fs), fs.T, fs.C);
  fs.e_stale = false;
  return hipGetLastError() == hipSuccess;
}

// Recognise the finite-difference pattern of one objective + gradient evaluation
// (bfgs_batch.cpp plan(): x, then the F probes, then the alpha probes) with alpha probes
// close enough for exp_small<4> (or <2>) on every finite distance of this data set.
static uint32_t fd_pattern(const GroupDesc& G, double dmax, uint64_t T, bool forced_visits,
                           bool allow_xdeg2) {
  if (G.np < 2 || !(G.F[0] > 0 && G.F[0] < 1) || !(G.A[0] > 0)) return 0;
  int nf = 0, na = 0;
  bool ownex = false;
  double xmax = 0;  // largest |alpha_0 - alpha_probe| d over the data's finite distances
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
      if (forced_visits && !(std::fabs(std::log(rho0)) * (double)T <= 600.0)) {
        // ... or, where the probe stays in range over the eight sites between two rescales
        // (always, with F inside [1e-15, 1 - 1e-15]), the pattern kernel with an exponent per
        // point: the shared transition terms are still formed once per site
        if (!(std::fabs(std::log(rho0)) * 8.0 <= 600.0)) return 0;
        ownex = true;
      }
      ++nf;
    } else if (G.F[p] == G.F[0] && std::fabs(G.A[p] - G.A[0]) * dmax <= 1e-3) {
      ++na;
      xmax = std::fmax(xmax, std::fabs(G.A[p] - G.A[0]) * dmax);
    } else {
      return 0;
    }
  }
  const bool ok = (nf == 2 && na == 2) || (nf == 1 && na == 2) || (nf == 2 && na == 1) ||
                  (nf == 1 && na == 1) || (nf == 2 && na == 0) || (nf == 0 && na == 2);
  if (!ok) return 0;
  return fd_mode(nf, na, G.A[0] * dmax <= 0.015625, allow_xdeg2 && na > 0 && xmax <= 1e-5) |
         (ownex ? FD_OWNEX : 0u);
}

bool fast_lkl_prepare(FastState& fs, hipStream_t st, uint32_t n_pts, const uint32_t* h_ind,
                      const double* h_F, const double* h_A, bool force_general) {
  FastState::LklLane& L = fs.lanes[fs.cur_lane];
  L.n_groups = 0;
  L.n_pts = n_pts;
  if (n_pts == 0) return true;
  // group the points by individual (<= MAXP per group): stable counting sort on the
  // individual index (the caller has checked ind < I)
  std::vector<uint32_t> order(n_pts), start(fs.I + 1, 0);
  for (uint32_t p = 0; p < n_pts; ++p) ++start[h_ind[p] + 1];
  for (uint64_t i = 0; i < fs.I; ++i) start[i + 1] += start[i];
  for (uint32_t p = 0; p < n_pts; ++p) order[start[h_ind[p]]++] = p;
  std::vector<GroupDesc> groups;
  groups.reserve(n_pts / 3 + 1);
  for (uint32_t k = 0; k < n_pts;) {
    GroupDesc G;
    std::memset(&G, 0, sizeof G);
    G.ind = h_ind[order[k]];
    uint32_t np = 0;
    while (k < n_pts && np < (uint32_t)MAXP && h_ind[order[k]] == G.ind) {
      G.F[np] = h_F[order[k]];
      G.A[np] = h_A[order[k]];
      G.out_idx[np] = order[k];
      ++np;
      ++k;
    }
    G.np = np;
    G.mode = force_general ? 0u : fd_pattern(G, fs.dmax_finite, fs.T, fs.packed, !fs.sw.no_xdeg2);
    groups.push_back(G);
  }
  // one kernel per loop-body version: sort the groups by mode (stable, so still in
  // individual order inside a mode) and remember the ranges
  std::stable_sort(groups.begin(), groups.end(),
                   [](const GroupDesc& a, const GroupDesc& b) { return a.mode < b.mode; });
  L.mode_ranges.clear();
  for (uint32_t k = 0; k < groups.size();) {
    uint32_t e = k;
    while (e < groups.size() && groups[e].mode == groups[k].mode) ++e;
    L.mode_ranges.push_back({groups[k].mode, k, e - k});
    k = e;
  }
  if (fs.sw.debug_modes) {  // which loop-body versions this round uses
    std::fprintf(stderr, "[nghmm modes] d_max %.6g:", fs.dmax_finite);
    for (const auto& r : L.mode_ranges) {
      if (r.mode)
        std::fprintf(stderr, " %uF%uA%s%s%s x%u", (r.mode >> 2) & 3, r.mode & 3,
                     (r.mode & FD_SMALL) ? "s" : "", (r.mode & FD_XDEG2) ? "2" : "",
                     (r.mode & FD_OWNEX) ? "e" : "", r.count);
      else
        std::fprintf(stderr, " general x%u", r.count);
    }
    std::fprintf(stderr, "\n");
  }
  const uint32_t ng = (uint32_t)groups.size();
  const size_t gbytes = (size_t)ng * sizeof(GroupDesc);
  if (gbytes > L.grp_cap) {
    if (L.grp_dev) (void)hipFree(L.grp_dev);
    L.grp_dev = nullptr;
    L.grp_cap = 0;
    const size_t cap = gbytes + gbytes / 4 + 4096;
    if (hipMalloc(&L.grp_dev, cap) != hipSuccess) return false;
    L.grp_cap = cap;
  }
  const size_t pdoubles = (size_t)ng * fs.C * MAXP * 5;
  if (pdoubles > L.part_cap) {
    if (L.part) (void)hipFree(L.part);
    L.part = nullptr;
    L.part_cap = 0;
    const size_t cap = pdoubles + pdoubles / 4 + 1024;
    if (!dalloc(&L.part, cap)) return false;
    L.part_cap = cap;
  }
  // the descriptors must outlive the async copy
  L.grp_host.assign(reinterpret_cast<unsigned char*>(groups.data()),
                     reinterpret_cast<unsigned char*>(groups.data()) + gbytes);
  if (hipMemcpyAsync(L.grp_dev, L.grp_host.data(), gbytes, hipMemcpyHostToDevice, st) !=
      hipSuccess)
    return false;
  L.n_groups = ng;
  return true;
}

bool fast_lkl_launch(FastState& fs, hipStream_t st, double* d_lkl, int* d_flags, bool emit_estep) {
  FastState::LklLane& L = fs.lanes[fs.cur_lane];
  const uint32_t ng = L.n_groups;
  if (ng == 0) return true;
  const GroupDesc* dg = reinterpret_cast<const GroupDesc*>(L.grp_dev);
  const LklArrays arr = lkl_arrays(fs);
  // first round of an M-step inside nghmm_estep_mstep: point 0 of every individual is the
  // E-step's forward walk, whose lane operators and checkpoints it leaves behind; if the
  // emissions are stale (frequencies just updated) the same walk recomputes and stores them
  const EmitPtrs emit = emit_estep ? EmitPtrs{fs.lane_ops, reinterpret_cast<double2*>(fs.ckpt)}
                                   : EmitPtrs{nullptr, nullptr};
  const bool fresh = emit_estep && fs.e_stale;
  if (fs.e_stale && !fresh) return false;  // the caller refreshes the emissions first
  // (the kernel versions of one round side by side on helper streams, so that they share one
  // partly filled last wave batch: measured, no gain -- 26.6-26.9 vs 27.0-27.1 ms per iteration
  // at 1000 x 1M)
  for (const auto& r : L.mode_ranges) {
    const dim3 grid(r.count * fs.C), block(64);
    // a group that needs an exponent per point (FD_OWNEX) in a round that also emits the
    // E-step's by-products goes to the general kernel as before
    const uint32_t mode = ((r.mode & FD_OWNEX) && emit_estep) ? 0u : r.mode;
    switch (mode) {
#define FD_LAUNCH(NF, NA, SM, EM, FR, XD)                                                    \
  hipLaunchKernelGGL((k_fast_lkl_fd<NF, NA, SM, EM, FR, XD>), grid, block, 0, st, arr, fs.T,    \
                     fs.C, dg, r.begin, L.part, emit)
#define FD_CASE1(NF, NA, SM, XD)                                              \
  case fd_mode(NF, NA, SM, XD == 2):                                          \
    if (fresh && fs.packed) FD_LAUNCH(NF, NA, SM, true, SRC_FRESH_PACKED, XD); \
    else if (fresh) FD_LAUNCH(NF, NA, SM, true, SRC_FRESH, XD);               \
    else if (emit_estep) FD_LAUNCH(NF, NA, SM, true, SRC_PLAIN, XD);          \
    else FD_LAUNCH(NF, NA, SM, false, SRC_PLAIN, XD);                         \
    break;                                                                    \
  case fd_mode(NF, NA, SM, XD == 2) | FD_OWNEX:                               \
    hipLaunchKernelGGL((k_fast_lkl_fd<NF, NA, SM, false, SRC_PLAIN, XD, true>), grid, block, 0, st, arr, \
                       fs.T, fs.C, dg, r.begin, L.part, emit);                \
    break;
#define FD_CASE(NF, NA)       \
  FD_CASE1(NF, NA, false, 4)  \
  FD_CASE1(NF, NA, true, 4)   \
  FD_CASE1(NF, NA, false, 2)  \
  FD_CASE1(NF, NA, true, 2)
      FD_CASE(2, 2)
      FD_CASE(1, 2)
      FD_CASE(2, 1)
      FD_CASE(1, 1)
      FD_CASE(0, 2)
#undef FD_CASE
      FD_CASE1(2, 0, false, 4)  // no alpha probe: nothing for the degree to choose
      FD_CASE1(2, 0, true, 4)
#undef FD_CASE1
#undef FD_LAUNCH
      default:
        if (fresh && fs.packed)
          hipLaunchKernelGGL((k_fast_lkl_chunks<MAXP, SRC_FRESH_PACKED>), grid, block, 0, st, arr,
                             fs.T, fs.C, dg, r.begin, L.part, emit);
        else if (fresh)
          hipLaunchKernelGGL((k_fast_lkl_chunks<MAXP, SRC_FRESH>), grid, block, 0, st, arr, fs.T,
                             fs.C, dg, r.begin, L.part, emit);
        else
          hipLaunchKernelGGL((k_fast_lkl_chunks<MAXP, SRC_PLAIN>), grid, block, 0, st, arr, fs.T,
                             fs.C, dg, r.begin, L.part, emit);
    }
  }
  if (fresh) fs.e_stale = false;
  if (fs.shard.world > 1) {
    // this handle's sites are a range of the data set's: its operators to everybody, theirs back
    SiteShard& sh = fs.shard;
    if ((uint64_t)L.n_pts * 6 > sh.cap) return false;
    hipLaunchKernelGGL(k_fast_lkl_finish<true>, dim3(ng), dim3(64 * MAXP), 0, st, dg, ng, fs.C, L.part,
                       fs.base_c, sh.send, d_flags);
    if (hipGetLastError() != hipSuccess) return false;
    if (sh.allgather(sh.user, (uint64_t)L.n_pts * 6 * sizeof(double)) != 0) return false;
    ++sh.n_gathers;
    hipLaunchKernelGGL(k_fast_shard_combine, dim3(((unsigned)ng * MAXP + 255) / 256), dim3(256), 0, st, dg,
                       ng, sh.recv, sh.world, (uint64_t)L.n_pts, d_lkl, d_flags);
    sh.edges_from_round = false;
#ifndef NGHMM_NO_EDGE_MERGE  // (A/B builds: the E-step with an all-gather of its own)
    if (emit_estep && !fs.sw.no_fuse) {  // every individual is in the batch: the E-step's edges too
      hipLaunchKernelGGL(k_fast_shard_edges_from_round, dim3((ng + 255) / 256), dim3(256), 0, st, dg, ng,
                         sh.recv, sh.world, sh.rank, (uint64_t)L.n_pts, sh.edges);
      sh.edges_from_round = true;
    }
#endif
    return hipGetLastError() == hipSuccess;
  }
  hipLaunchKernelGGL(k_fast_lkl_finish<false>, dim3(ng), dim3(64 * MAXP), 0, st, dg, ng, fs.C, L.part,
                     fs.base_c, d_lkl, d_flags);
  return hipGetLastError() == hipSuccess;
}

bool fast_lkl_covers_everyone(const FastState& fs) {
  const FastState::LklLane& L = fs.lanes[fs.cur_lane];
  // one group per individual (points are grouped by individual, <= MAXP each; an M-step's
  // first round has <= 5 points per individual, so groups == individuals iff all are there)
  return L.n_groups == fs.I;
}

bool fast_estep(FastState& fs, hipStream_t st, const double* d_indF, const double* d_alpha,
                double* d_ind_lkl, int* d_flags, bool have_forward_walk) {
  const double* e2 = fs.e_il;
  double2* ck = reinterpret_cast<double2*>(fs.ckpt);
  const unsigned waves = (unsigned)(fs.I * fs.C);
  if (!have_forward_walk)
    hipLaunchKernelGGL(k_fast_chunk_ops, dim3(waves), dim3(64), 0, st, e2, fs.pos_il, fs.T, fs.C,
                       d_indF, d_alpha, EmitPtrs{fs.lane_ops, ck});
  const double* edges = nullptr;
  if (fs.shard.world > 1) {
    SiteShard& sh = fs.shard;
    if (!(have_forward_walk && sh.edges_from_round)) {
      if (fs.I * 6 > sh.cap) return false;
      hipLaunchKernelGGL(k_fast_shard_reduce, dim3((unsigned)fs.I), dim3(64), 0, st, fs.lane_ops, fs.J,
                         fs.C, fs.base_c, sh.send);
      if (hipGetLastError() != hipSuccess) return false;
      if (sh.allgather(sh.user, fs.I * 6 * sizeof(double)) != 0) return false;
      ++sh.n_gathers;
      hipLaunchKernelGGL(k_fast_shard_edges, dim3((unsigned)((fs.I + 255) / 256)), dim3(256), 0, st,
                         sh.recv, sh.world, sh.rank, fs.I, d_indF, sh.edges);
    }
    sh.edges_from_round = false;
    edges = sh.edges;
  }
  hipLaunchKernelGGL(k_fast_bounds, dim3((unsigned)fs.I), dim3(64), 0, st, fs.lane_ops, fs.J, fs.C,
                     d_indF, fs.base_c, fs.bound, d_ind_lkl, d_flags, edges);
  if (kPost8)
    hipLaunchKernelGGL(k_fast_bwd_recompute8, dim3((unsigned)(((fs.I + 7) / 8) * fs.C * 2)), dim3(256),
                       0, st, e2, fs.pos_il, fs.T, fs.C, fs.S, fs.I, d_indF, d_alpha, fs.bound, ck,
                       fs.post, d_flags);
  else
    hipLaunchKernelGGL(k_fast_bwd_recompute, dim3(waves), dim3(64), 0, st, e2, fs.pos_il, fs.T, fs.C,
                       fs.S, fs.I, d_indF, d_alpha, fs.bound, ck, fs.post, d_flags);
  return hipGetLastError() == hipSuccess;
}

bool fast_post_to_site_major(FastState& fs, hipStream_t st, double* d_marg) {
  const uint64_t n_it = (fs.I + 63) / 64;
  const dim3 grid((unsigned)((uint64_t)fs.C * fs.T * n_it)), block(256);
  // 16-byte stores need s * I + i even for even i, i.e. an even number of individuals, and
  // a 16-byte aligned destination
  if (fs.I % 2 == 0 && (reinterpret_cast<uintptr_t>(d_marg) & 15) == 0)
    hipLaunchKernelGGL((k_fast_post_to_site_major<true>), grid, block, 0, st, fs.post, fs.I, fs.S,
                       fs.T, fs.C, d_marg);
  else
    hipLaunchKernelGGL((k_fast_post_to_site_major<false>), grid, block, 0, st, fs.post, fs.I, fs.S,
                       fs.T, fs.C, d_marg);
  return hipGetLastError() == hipSuccess;
}

// called genotypes (a packed handle whose class table is the four unit / uniform rows): est_maf's
// per-pass sums exist in closed form
bool fast_estmaf_called(const FastState& fs, const GlView& gl) {
  return !gl.dense && gl.codes && fs.called_table && !fs.sw.estmaf_no_called;
}

bool fast_estmaf_in_place(const FastState& fs, uint64_t I_tot) {
  // est_maf on the E-step's tile-major posteriors, without the site-major copy: the register
  // kernels up to 4096 individuals (measured at 10^9 site-individuals: in place 13.6 vs 14.4 ms
  // via the copy at 4000 individuals, 19.7 vs 16.5 ms at 8000: a site group's sectors outgrow
  // L2); the called-genotype sweep reads every cell once, whole sectors, at any size
  if (fs.sw.estmaf_sitemajor) return false;
  return I_tot <= 4096 || (fs.packed && fs.called_table && !fs.sw.estmaf_no_called);
}

bool fast_estmaf_splittable(const FastState& fs, uint64_t I_tot, bool tile_major) {
  // the wave-per-site kernels on the E-step's tile-major posteriors: a part is a range of
  // tile rows, i.e. of workgroups
  if (tile_major && fs.packed && fs.called_table && !fs.sw.estmaf_no_called) return true;
  return tile_major && I_tot > 128 && I_tot <= 8192;
}

bool fast_estmaf(FastState& fs, hipStream_t st, const GlView& d_gl_sites,
                 const double* d_marg_blocks, uint64_t S_own, uint64_t I_tot, uint64_t I_blk,
                 double* d_freq_out, bool tile_major, uint32_t part, uint32_t n_parts) {
  if (S_own == 0) return true;
  if (n_parts == 0 || part >= n_parts) return false;
  if (n_parts > 1 && !fast_estmaf_splittable(fs, I_tot, tile_major)) return false;
  // tile rows [row0, row1) of this call (everything when n_parts == 1)
  const uint64_t n_rows = (uint64_t)fs.C * fs.T;
  const uint64_t row0 = n_parts > 1 ? n_rows * part / n_parts : 0;
  const uint64_t row1 = n_parts > 1 ? n_rows * (part + 1) / n_parts : n_rows;
  const uint64_t blk0 = row0 * 64, nblk = (row1 - row0) * 64;
  if (nblk == 0) return true;
  // tile-major posteriors (the E-step's own layout) only for the handle's whole site range
  // and individuals that fit the registers of one workgroup
  if (tile_major && !((I_tot <= 8192 || fast_estmaf_called(fs, d_gl_sites)) && I_blk == I_tot &&
                      S_own == fs.S))
    return false;
  const uint64_t tile_T = tile_major ? fs.T : 0;
  // k_fast_estmaf_stream, 4 waves per workgroup: a wave per site when it streams every site,
  // else 64 flags per wave and turn
  const bool stream_all = I_tot > 8192 && !tile_major;
  const uint64_t stream_wgs = stream_all ? (S_own + 3) / 4 : (S_own + 255) / 256;
  const dim3 grid((unsigned)(stream_wgs < 65536 ? stream_wgs : 65536)), block(256);
  if (S_own > fs.redo_cap) {
    if (fs.redo) (void)hipFree(fs.redo);
    if (fs.est_status) (void)hipFree(fs.est_status);
    if (fs.est_state) (void)hipFree(fs.est_state);
    fs.redo = fs.est_status = nullptr;
    fs.est_state = nullptr;
    fs.redo_cap = 0;
    if (hipMalloc((void**)&fs.redo, S_own) != hipSuccess) return false;
    if (hipMalloc((void**)&fs.est_status, S_own) != hipSuccess) return false;
    if (hipMalloc((void**)&fs.est_state, S_own * EST_FIELDS * sizeof(double)) != hipSuccess)
      return false;
    fs.redo_cap = S_own;
  }
  if (fast_estmaf_called(fs, d_gl_sites)) {
    // called genotypes: the per-pass sums in closed form (k_fast_estmaf_called_sums)
    if (tile_major)
      hipLaunchKernelGGL((k_fast_estmaf_called_sums<true>), dim3((unsigned)nblk), dim3(64), 0, st,
                         d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, tile_T, fs.redo, fs.est_state,
                         fs.redo_cap, blk0);
    else
      hipLaunchKernelGGL((k_fast_estmaf_called_sums<false>), dim3((unsigned)S_own), dim3(64), 0, st,
                         d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, tile_T, fs.redo, fs.est_state,
                         fs.redo_cap, (uint64_t)0);
    hipLaunchKernelGGL(k_fast_estmaf_called_passes, dim3((unsigned)((S_own + 255) / 256)), dim3(256), 0,
                       st, S_own, d_freq_out, fs.redo, fs.est_status, fs.est_state, fs.redo_cap, tile_T,
                       row0, row1);
    // a called heterozygote at posterior IBD = 1 (the reference keeps a finite -1e15 there)
    hipLaunchKernelGGL(k_fast_estmaf_stream, grid, block, 0, st, d_gl_sites, d_marg_blocks, S_own,
                       I_tot, I_blk, tile_T, d_freq_out, fs.redo, row0, row1);
    return hipGetLastError() == hipSuccess;
  }
  // Interpolated passes (see k_fast_estmaf) unless NGHMM_ESTMAF_INTERP=0, which runs
  // every pass exactly.
  bool interp = true;
  interp = fs.sw.estmaf_interp != 0;
  // waves per site (W) and individuals per lane (NI): 16 per lane at two waves per SIMD;
  // a workgroup must fit one CU
  // resuming launches: 64 statuses per workgroup and turn
  const uint64_t scan_wgs_all = (S_own + 63) / 64;
  const unsigned scan_wgs = (unsigned)(scan_wgs_all < 16384 ? scan_wgs_all : 16384);
#define LAUNCH_NI(N, B)                                                                         \
  do {                                                                                          \
    if (fresh)                                                                                  \
      hipLaunchKernelGGL((k_fast_estmaf<N, B, false>), dim3((unsigned)S_own), dim3(B), 0, st,    \
                         d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, (uint64_t)0,           \
                         d_freq_out, fs.redo, fs.est_status, fs.est_state, fs.redo_cap,         \
                         n_exact, allow_build, (uint64_t)0);                                    \
    else                                                                                        \
      hipLaunchKernelGGL((k_fast_estmaf_resume<N, B, false>), dim3(scan_wgs), dim3(B), 0, st,   \
                         d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, (uint64_t)0,           \
                         d_freq_out, fs.redo, fs.est_status, fs.est_state, fs.redo_cap,         \
                         n_exact, allow_build, row0, row1);                                     \
  } while (0)
#define LAUNCH_TILE(N, B)                                                                       \
  do {                                                                                          \
    if (fresh)                                                                                  \
      hipLaunchKernelGGL((k_fast_estmaf<N, B, true>), dim3((unsigned)nblk), dim3(B), 0, st,     \
                         d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, tile_T, d_freq_out,    \
                         fs.redo, fs.est_status, fs.est_state, fs.redo_cap, n_exact,            \
                         allow_build, blk0);                                                    \
    else                                                                                        \
      hipLaunchKernelGGL((k_fast_estmaf_resume<N, B, true>), dim3(scan_wgs), dim3(B), 0, st,    \
                         d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, tile_T, d_freq_out,    \
                         fs.redo, fs.est_status, fs.est_state, fs.redo_cap, n_exact,            \
                         allow_build, row0, row1);                                              \
  } while (0)
#define LAUNCH_ROWS(N, TL)                                                                      \
  do {                                                                                          \
    if (fresh)                                                                                  \
      hipLaunchKernelGGL((k_fast_estmaf_rows<N, TL>),                                            \
                         dim3((unsigned)((TL) ? fs.Spad / 4 : (S_own + 3) / 4)), dim3(64), 0,   \
                         st, d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, tile_T, d_freq_out, \
                         fs.redo, fs.est_status, fs.est_state, fs.redo_cap, n_exact,            \
                         allow_build);                                                          \
    else                                                                                        \
      hipLaunchKernelGGL((k_fast_estmaf_rows_resume<N, TL>), dim3(scan_wgs), dim3(64), 0, st,    \
                         d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, tile_T, d_freq_out,    \
                         fs.redo, fs.est_status, fs.est_state, fs.redo_cap, n_exact,            \
                         allow_build);                                                          \
  } while (0)
  // up to 128 individuals: four sites per wave (k_fast_estmaf_rows)
  const bool rows = I_tot <= 128 && !fs.sw.estmaf_no_rows;
  auto launch = [&](int fresh, int n_exact, int allow_build) -> bool {
    if (rows) {
      if (tile_major) {
        if (I_tot <= 16) LAUNCH_ROWS(1, true);
        else if (I_tot <= 32) LAUNCH_ROWS(2, true);
        else if (I_tot <= 64) LAUNCH_ROWS(4, true);
        else LAUNCH_ROWS(8, true);
      } else {
        if (I_tot <= 16) LAUNCH_ROWS(1, false);
        else if (I_tot <= 32) LAUNCH_ROWS(2, false);
        else if (I_tot <= 64) LAUNCH_ROWS(4, false);
        else LAUNCH_ROWS(8, false);
      }
      return true;
    }
    if (tile_major) {
      if (I_tot <= 64) LAUNCH_TILE(1, 64);
      else if (I_tot <= 128) LAUNCH_TILE(2, 64);
      else if (I_tot <= 256) LAUNCH_TILE(4, 64);
      else if (I_tot <= 512) LAUNCH_TILE(8, 64);
      else if (I_tot <= 768) LAUNCH_TILE(12, 64);
      else if (I_tot <= 1024) LAUNCH_TILE(16, 64);
      // several waves per site: a lane's slot k holds individual thread + BLOCK k, so a cohort
      // in the lower half of a size class leaves the upper slots of EVERY lane empty -- 12
      // instead of 16 slots there (5000 individuals on 512 threads: 9.8 slots in use)
      else if (I_tot <= 1536) LAUNCH_TILE(12, 128);
      else if (I_tot <= 2048) LAUNCH_TILE(16, 128);
      else if (I_tot <= 3072) LAUNCH_TILE(12, 256);
      else if (I_tot <= 4096) LAUNCH_TILE(16, 256);
      else if (I_tot <= 6144) LAUNCH_TILE(12, 512);
      else LAUNCH_TILE(16, 512);
    } else if (I_tot <= 64) LAUNCH_NI(1, 64);
    else if (I_tot <= 128) LAUNCH_NI(2, 64);
    else if (I_tot <= 256) LAUNCH_NI(4, 64);
    else if (I_tot <= 512) LAUNCH_NI(8, 64);
    else if (I_tot <= 768) LAUNCH_NI(12, 64);
    else if (I_tot <= 1024) LAUNCH_NI(16, 64);
    else if (I_tot <= 1536) LAUNCH_NI(12, 128);
    else if (I_tot <= 2048) LAUNCH_NI(16, 128);
    else if (I_tot <= 3072) LAUNCH_NI(12, 256);
    else if (I_tot <= 4096) LAUNCH_NI(16, 256);
    else if (I_tot <= 6144) LAUNCH_NI(12, 512);
    else LAUNCH_NI(16, 512);
    return true;
  };
  const uint8_t* redo = fs.redo;
  if (I_tot > 8192 && !tile_major) {
    redo = nullptr;  // more individuals than registers hold: stream every site
  } else if (!interp) {
    if (!launch(1, 0, 0)) return false;
  } else {
    const dim3 igrid((unsigned)((S_own + 255) / 256));
    if (!launch(1, EST_K0, 1)) return false;
    hipLaunchKernelGGL(k_fast_estmaf_interp, igrid, dim3(256), 0, st, S_own, d_freq_out, fs.redo,
                       fs.est_status, fs.est_state, fs.redo_cap, tile_T, row0, row1);
    if (!launch(0, 1, 1)) return false;  // sites that left their interval: one more
    hipLaunchKernelGGL(k_fast_estmaf_interp, igrid, dim3(256), 0, st, S_own, d_freq_out, fs.redo,
                       fs.est_status, fs.est_state, fs.redo_cap, tile_T, row0, row1);
    if (!launch(0, 0, 0)) return false;  // whatever is left finishes on exact passes
  }
  hipLaunchKernelGGL(k_fast_estmaf_stream, grid, block, 0, st, d_gl_sites, d_marg_blocks, S_own,
                     I_tot, I_blk, tile_T, d_freq_out, redo, row0, row1);
#undef LAUNCH_NI
#undef LAUNCH_TILE
#undef LAUNCH_ROWS
  return hipGetLastError() == hipSuccess;
}


bool fast_viterbi(FastState& fs, hipStream_t st, const double* d_freq, const double* d_indF,
                  const double* d_alpha, uint8_t* d_bp, uint8_t* d_path_sites, int* d_flags,
                  double* d_scratch, uint64_t chunk_sites) {
  // Decoding runs once per analysis and must give the reference's path, ties and
  // its in-place update included: use the exact-mode kernels on log emissions.  The
  // site-major log emissions [S][I][2] (16 B per site and individual; the fast path keeps
  // only the 8 B ratio) live in a buffer of their own for the duration of the call.
  double* eprob_log = nullptr;
  if (!dalloc(&eprob_log, (size_t)fs.I * fs.S * 2)) return false;
  launch_emission_exact(st, fs.gl_log, d_freq, eprob_log, fs.S, fs.I, d_flags);
  launch_viterbi_exact(st, eprob_log, fs.d_pos, fs.S, fs.I, d_indF, d_alpha, d_bp, d_path_sites,
                       d_scratch, chunk_sites, fs.sw.exact_serial != 0);
  const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
  (void)hipFree(eprob_log);
  return ok;
}

bool fast_viterbi_forward(FastState& fs, hipStream_t st, const double* d_freq, const double* d_indF,
                          const double* d_alpha, uint8_t* d_bp, int* d_flags, double* d_scratch,
                          uint64_t chunk_sites, bool chain_start) {
  double* eprob_log = nullptr;
  if (!dalloc(&eprob_log, (size_t)fs.I * fs.S * 2)) return false;
  launch_emission_exact(st, fs.gl_log, d_freq, eprob_log, fs.S, fs.I, d_flags);
  launch_viterbi_fwd_exact(st, eprob_log, fs.d_pos, fs.S, fs.I, d_indF, d_alpha, d_bp, d_scratch,
                           chunk_sites, chain_start, fs.sw.exact_serial != 0);
  const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
  (void)hipFree(eprob_log);
  return ok;
}

bool fast_export_emissions(FastState& fs, hipStream_t st, double* d_out) {
  hipLaunchKernelGGL(k_fast_export_e, dim3(2048), dim3(256), 0, st, lkl_arrays(fs), fs.packed,
                     fs.gl_lin, fs.I, fs.S, fs.T, fs.C, d_out);
  return hipGetLastError() == hipSuccess;
}

}  // namespace nghmm

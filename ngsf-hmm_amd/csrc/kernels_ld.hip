// kernels_ld.hip -- the allele-frequency step of --freq_est 2 / --e_prob 2 AS INTENDED.
//
// OPT-IN, PARITY UNPINNED.  The reference aborts on --freq_est 2 and --e_prob 2 at the first
// site (EM.cpp:235-238 hands haplo_freq freq[0] = -1 -> "invalid allele frequencies",
// shared/gen_func.cpp:1030-1031); past that, its log-space pair iteration discards a logsum
// (gen_func.cpp:1160) and its LD emission (EM.cpp:258-260) is unreachable.  There is no
// reference output to compare with.  What these kernels compute is the loop of EM.cpp:224-263
// exactly as written with those three defects repaired the smallest way (oracle:
// orc_em_mstep_freq_ld, which they are tested against; binary128 anchor: oracle/hp_anchor.c):
//   * sites in order, frequencies updated in place: site s uses the NEW freq[s-1] and the old
//     freq[s] for the genotype posteriors of its two sites and for the starting point of its
//     haplotype-frequency EM -- a chain through the sites, sequential by definition;
//   * no haplotype step at the first site (est_maf there, as the code's own `s == 1` cases);
//   * the normal-space pair iteration (gen_func.cpp:1076-1119) under haplo_freq's loop
//     (:1027-1063), on the exponentiated posteriors;
//   * calc_emissionLD (shared/HMM.cpp:175-236) for the sites after the first under --e_prob 2.
//
// One workgroup of 1024 threads walks the chain: a thread owns up to 8 individuals, every
// pair iteration ends in a sum over the individuals.  EXACT: the reference's operations in its
// order with detmath's exp / log, the four sums accumulated in individual order (through LDS,
// one lane per sum) -- bit-identical to the oracle's det build.  Fast: linear-space posteriors
// (no transcendental at all), wave and workgroup tree sums; within 1e-9 of the oracle.
// The pairs' haplotype frequencies are kept ([S][4]); the LD emissions are a second, fully
// parallel kernel (exact mode: the emissions are materialised there).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "detmath.h"
#include "glview.hpp"
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace nghmm {

namespace {

#include "exact_dev.hpp"

constexpr int LD_THREADS = 1024;
constexpr int LD_ITER_MAX = 100;  // shared/gen_func.hpp:18

// shared/gen_func.cpp:1070-1071
__device__ __forceinline__ constexpr int ld_g1(int h, int k) { return ((h >> 1) & 1) + ((k >> 1) & 1); }
__device__ __forceinline__ constexpr int ld_g2(int h, int k) { return (h & 1) + (k & 1); }

// genotype posterior of one cell in normal space: exp(post_prob(gl, calc_HWE(maf, F)))
// (EM.cpp:229-232, gen_func.cpp:920-957); EXACT: log GL in, the reference's log-space route;
// else linear GL in, weights p_g HWE_g normalised (the het weight at F = 1 is the reference's
// exp(-1e15) = 0)
template <bool EXACT>
__device__ __forceinline__ void ld_posterior(double g0, double g1, double g2, double maf, double F,
                                             double (&pp)[3]) {
  if constexpr (EXACT) {
    double h0, h1, h2;
    hwe_log(maf, F, h0, h1, h2);
    double p0 = g0 + h0, p1 = g1 + h1, p2 = g2 + h2;
    const double norm = logsum3(p0, p1, p2);
    pp[0] = det_exp(p0 - norm);
    pp[1] = det_exp(p1 - norm);
    pp[2] = det_exp(p2 - norm);
  } else {
    const double om = 1 - maf, b = om * maf;
    const double w0 = g0 * (om * om + b * F);
    const double w1 = (F == 1) ? 0.0 : g1 * (2 * b - 2 * b * F);
    const double w2 = g2 * (maf * maf + b * F);
    const double inv = 1.0 / ((w0 + w1) + w2);
    pp[0] = w0 * inv;
    pp[1] = w1 * inv;
    pp[2] = w2 * inv;
  }
}

// sum over the workgroup of four per-thread values; fast: shuffle tree per wave, the wave totals
// through LDS, every thread adds them in wave order.  xw alternates between two buffers from
// call to call (`flip`), so ONE barrier per call is enough: a thread that races ahead into the
// next call writes the other buffer, and the one after that starts behind the next barrier.
template <int TH>
__device__ __forceinline__ void ld_block_sum4(double (&v)[4], double (*xw)[TH / 64][4], int tid,
                                              int flip) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
    for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off);
  const int wv = tid >> 6;
  double (*buf)[4] = xw[flip & 1];
  if ((tid & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) buf[wv][k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double a = buf[0][k];
    for (int w = 1; w < TH / 64; ++w) a += buf[w][k];
    v[k] = a;
  }
}

// gl: EXACT log GL view / else linear GL view, site-major cells; marg [S][I]; freq_old [S];
// freq_new [S]: [0] holds site 0's new frequency on entry (est_maf), freq_est == 1: all of it
// does; hap_out [S][4] (row 0 unused).
template <bool EXACT, int NI, int TH>
__global__ void __launch_bounds__(TH)
k_freq_ld_chain(const GlView gl, const double* __restrict__ marg,
                const double* __restrict__ freq_old, double* __restrict__ freq_new,
                double* __restrict__ hap_out, uint64_t S, uint64_t I, int freq_est,
                int* __restrict__ flags) {
  __shared__ double contrib[EXACT ? TH : 1][4];  // one chunk of individuals' tmp / sum
  __shared__ double xw[2][TH / 64][4];
  int flip = 0;
  __shared__ double ffs[4];
  const int tid = threadIdx.x;
  double gp[NI][3], Fp[NI];
  bool valid[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const uint64_t i = (uint64_t)tid + (uint64_t)j * TH;
    valid[j] = i < I;
    const uint64_t ic = valid[j] ? i : I - 1;
    gl_fetch(gl, ic, gp[j][0], gp[j][1], gp[j][2]);
    Fp[j] = marg[ic];
  }
  double f_prev = freq_new[0];
  const double two_x = (double)(2 * I);  // gen_func.cpp:1109: ff[k] / (2 * x), x = n
  for (uint64_t s = 1; s < S; ++s) {
    const double f_cur = freq_old[s];
    if (f_prev < 0 || f_prev > 1 || f_cur < 0 || f_cur > 1) {  // gen_func.cpp:1030-1031
      if (tid == 0) flags[FLAG_LD_FREQ] = 1;
      return;                                                  // uniform: every thread sees it
    }
    double gc[NI][3], Fc[NI];
    double P0[NI][3], P1[NI][3];  // the two sites' genotype probabilities
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const uint64_t i = (uint64_t)tid + (uint64_t)j * TH;
      const uint64_t ic = valid[j] ? i : I - 1;
      gl_fetch(gl, s * I + ic, gc[j][0], gc[j][1], gc[j][2]);
      Fc[j] = marg[s * I + ic];
      ld_posterior<EXACT>(gp[j][0], gp[j][1], gp[j][2], f_prev, Fp[j], P0[j]);
      ld_posterior<EXACT>(gc[j][0], gc[j][1], gc[j][2], f_cur, Fc[j], P1[j]);
    }
    // haplo_freq (gen_func.cpp:1027-1063)
    double f[4] = {(1 - f_prev) * (1 - f_cur), (1 - f_prev) * f_cur, f_prev * (1 - f_cur),
                   f_prev * f_cur};
    for (int it = 0; it < LD_ITER_MAX; ++it) {
      double last[4] = {f[0], f[1], f[2], f[3]};
      // pair_freq_iter (gen_func.cpp:1076-1119)
      double fkh[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int h = 0; h < 4; ++h) fkh[k][h] = f[k] * f[h];
      double ff[4] = {0, 0, 0, 0};
      double c[NI][4];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        double sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int h = 0; h < 4; ++h) sum += fkh[k][h] * P0[j][ld_g1(k, h)] * P1[j][ld_g2(k, h)];
        const double inv_sum = EXACT ? 0.0 : 1.0 / sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          double tmp = 0;
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            // p0[G1(h,k)] p1[G2(h,k)] + p0[G1(k,h)] p1[G2(k,h)]: the same product twice
            const double x = P0[j][ld_g1(h, k)] * P1[j][ld_g2(h, k)];
            tmp += fkh[k][h] * (x + x);
          }
          // exact: the reference's quotient; fast: one reciprocal per individual and iteration
          // instead of four divisions (the four share their divisor)
          if constexpr (EXACT) c[j][k] = valid[j] ? tmp / sum : 0.0;
          else c[j][k] = valid[j] ? tmp * inv_sum : 0.0;
        }
      }
      if constexpr (EXACT) {
        // ff[k] += tmp / sum in individual order: chunk j holds the individuals j*1024 ..,
        // lane k of the first wave adds the chunk's entries one after the other
        for (int j = 0; j < NI; ++j) {
          __syncthreads();
#pragma unroll
          for (int k = 0; k < 4; ++k) contrib[tid][k] = c[j][k];
          __syncthreads();
          if (tid < 4) {
            const uint64_t base = (uint64_t)j * TH;
            const int n = base >= I ? 0 : (I - base < (uint64_t)TH ? (int)(I - base) : TH);
            double acc = (j == 0) ? 0.0 : ffs[tid];
#pragma unroll 8
            for (int t = 0; t < n; ++t) acc += contrib[t][tid];
            ffs[tid] = acc;
          }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) ff[k] = ffs[k];
      } else {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) ff[k] += c[j][k];
        ld_block_sum4<TH>(ff, xw, tid, flip++);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) f[k] = ff[k] / two_x;
      // "Normalize" in place, as written: f[1] is divided by a sum that already holds the
      // normalised f[0], and so on
      f[0] /= f[0] + f[1] + f[2] + f[3];
      f[1] /= f[0] + f[1] + f[2] + f[3];
      f[2] /= f[0] + f[1] + f[2] + f[3];
      f[3] /= f[0] + f[1] + f[2] + f[3];
      double eps = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double x = fabs(f[k] - last[k]);
        if (x > eps) eps = x;
      }
      if (eps < kEPS) break;  // uniform: f is
    }
    const double f_new = (freq_est == 1) ? freq_new[s] : f[1] + f[3];  // EM.cpp:245-249
    if (tid == 0) {
      if (freq_est != 1) freq_new[s] = f_new;
      hap_out[s * 4 + 0] = f[0];
      hap_out[s * 4 + 1] = f[1];
      hap_out[s * 4 + 2] = f[2];
      hap_out[s * 4 + 3] = f[3];
    }
    f_prev = f_new;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      gp[j][0] = gc[j][0];
      gp[j][1] = gc[j][1];
      gp[j][2] = gc[j][2];
      Fp[j] = Fc[j];
    }
  }
}

// shared/HMM.cpp:216-236 (F_p == F_c)
__device__ __forceinline__ double ld_joint(const double (&h)[4], int g_p, int g_c, int F) {
  if (g_p == 0 && g_c == 0) return F == 0 ? h[0] * h[0] : h[0];
  if (g_p == 0 && g_c == 1) return F == 0 ? 2 * h[0] * h[1] : 0;
  if (g_p == 0 && g_c == 2) return F == 0 ? h[1] * h[1] : h[1];
  if (g_p == 1 && g_c == 0) return F == 0 ? 2 * h[0] * h[2] : 0;
  if (g_p == 1 && g_c == 1) return F == 0 ? 2 * (h[0] * h[3] + h[1] * h[2]) : 0;
  if (g_p == 1 && g_c == 2) return F == 0 ? 2 * h[1] * h[3] : 0;
  if (g_p == 2 && g_c == 0) return F == 0 ? h[2] * h[2] : h[2];
  if (g_p == 2 && g_c == 1) return F == 0 ? 2 * h[2] * h[3] : 0;
  return F == 0 ? h[3] * h[3] : h[3];
}

// e_prob[s][i][k] for s >= 1 = calc_emissionLD (shared/HMM.cpp:175-212, the live branch)
__global__ void __launch_bounds__(256)
k_emission_ld_exact(const GlView gl, const double* __restrict__ freq,
                    const double* __restrict__ hap, double* __restrict__ eprob, uint64_t S,
                    uint64_t I, int* __restrict__ flags) {
  const uint64_t n = (S - 1) * I;
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n;
       c += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t s = 1 + c / I, i = c % I;
    const double maf_p = freq[s - 1], maf_c = freq[s];
    double e[2];
    if (maf_p < 0 || maf_p > 1 || maf_c < 0 || maf_c > 1) {
      flags[FLAG_INVALID_MAF] = 1;
      e[0] = e[1] = __builtin_nan("");
    } else {
      double glp[3], glc[3], sp[3], sc[3];
      gl_fetch(gl, (s - 1) * I + i, glp[0], glp[1], glp[2]);
      gl_fetch(gl, s * I + i, glc[0], glc[1], glc[2]);
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        sp[g] = det_exp(glp[g]);
        sc[g] = det_exp(glc[g]);
      }
      const double h[4] = {hap[s * 4], hap[s * 4 + 1], hap[s * 4 + 2], hap[s * 4 + 3]};
#pragma unroll
      for (int F = 0; F < 2; ++F) {
        double sum = 0;
#pragma unroll
        for (int g_c = 0; g_c < 3; ++g_c)
#pragma unroll
          for (int g_p = 0; g_p < 3; ++g_p) sum += ld_joint(h, g_p, g_c, F) * sp[g_p] * sc[g_c];
        e[F] = det_log(sum) - emission_log(glp[0], glp[1], glp[2], maf_p, F);
      }
    }
    eprob[(s * I + i) * 2] = e[0];
    eprob[(s * I + i) * 2 + 1] = e[1];
  }
}

}  // namespace

bool launch_freq_ld_chain(hipStream_t st, bool exact, const GlView& gl, const double* marg,
                          const double* freq_old, double* freq_new, double* hap, uint64_t S,
                          uint64_t I, int freq_est, int* flags) {
  if (S < 2) return true;
  // threads of the one workgroup: every pair iteration ends in a sum over all of them, which a
  // small workgroup does sooner -- fast mode: 256 threads with up to 8 individuals each up to
  // 2048 individuals (1000 individuals: 51 -> 30 us per site), 1024 threads beyond; exact mode
  // keeps 1024 (its per-individual transcendentals weigh more than the sum: 143 -> 276 us with
  // 256)
  const bool small = !exact && I <= 2048;
  const uint64_t th = small ? 256 : LD_THREADS;  // (128 threads: 45 us per site; 512: 27)
  const uint64_t ni = (I + th - 1) / th;
  if (ni > 8) return false;
#define LD_LAUNCH(EX, NI, TH)                                                                  \
  hipLaunchKernelGGL((k_freq_ld_chain<EX, NI, TH>), dim3(1), dim3(TH), 0, st, gl, marg,        \
                     freq_old, freq_new, hap, S, I, freq_est, flags)
#define LD_PICK(EX, TH)                     \
  do {                                      \
    if (ni <= 1) LD_LAUNCH(EX, 1, TH);      \
    else if (ni <= 2) LD_LAUNCH(EX, 2, TH); \
    else if (ni <= 4) LD_LAUNCH(EX, 4, TH); \
    else LD_LAUNCH(EX, 8, TH);              \
  } while (0)
  if (exact) {
    if (small) LD_PICK(true, 256);
    else LD_PICK(true, LD_THREADS);
  } else {
    if (small) LD_PICK(false, 256);
    else LD_PICK(false, LD_THREADS);
  }
#undef LD_PICK
#undef LD_LAUNCH
  return true;
}

void launch_emission_ld_exact(hipStream_t st, const GlView& gl, const double* freq,
                              const double* hap, double* eprob, uint64_t S, uint64_t I,
                              int* flags) {
  if (S < 2 || I == 0) return;
  uint64_t blocks = ((S - 1) * I + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(k_emission_ld_exact, dim3((unsigned)blocks), dim3(256), 0, st, gl, freq, hap,
                     eprob, S, I, flags);
}

}  // namespace nghmm

/*
 * hp_anchor.c -- a high-precision anchor for the numeric routines of the hot path.
 *
 * TEST INFRASTRUCTURE ONLY (tests/ link it through ctypes; nothing under ngsf-hmm_amd/ may).
 *
 * Why: shared/HMM.cpp, the pop-gen part of shared/gen_func.cpp and EM.cpp cannot be compiled in
 * this image (they include <gsl/gsl_rng.h>), so the oracle's restatement of forward / backward /
 * posteriors / est_maf (ngsfhmm_oracle.c) is pinned to no reference binary.  This file shares NO
 * code and NO formulation with the oracle or with the HIP kernels: it evaluates the MODEL those
 * routines implement -- the two-state HMM of shared/HMM.cpp:6-60,130-154 and the per-site
 * frequency recursion of shared/gen_func.cpp:974-1009 -- in linear space with per-site scaling
 * (Rabiner's scaled forward-backward), in IEEE binary128 (__float128, libquadmath: 113-bit
 * significand).  Its rounding error is ~1e-30 per operation, so its results are "the truth" to
 * every digit a double can hold, and both the log-space double formulation (the reference's,
 * restated by the oracle) and the linear-space double kernels (fast mode) can be measured
 * against it.
 *
 * The reference's finite stand-ins: log 0 = -1e15 (conv_space) and the heterozygote prior at
 * F = 1 (-1e15) are all probabilities exp(-1e15) = 0 here; data on which the reference's
 * answer is dominated by those stand-ins (zero probability mass) are outside this anchor.
 */
#include <quadmath.h>
#include <stdint.h>
#include <stdlib.h>

typedef __float128 q128;

static q128 q_exp(double x) { return expq((q128)x); }

/* HWE genotype probabilities with inbreeding F (calc_HWE, shared/gen_func.cpp:938-957), linear */
static void hwe(q128 f, q128 F, q128 h[3]) {
  const q128 om = 1 - f;
  h[0] = om * om + om * f * F;
  h[1] = 2 * om * f * (1 - F);
  h[2] = f * f + om * f * F;
}

/* linear emissions of one individual: e[s][k] = sum_g exp(gl[s][g]) HWE_g(freq[s], F = k)
 * (calc_emission, shared/HMM.cpp:144-154) */
static void emissions(const double* gl, const double* freq, uint64_t S, q128* e) {
  for (uint64_t s = 0; s < S; ++s) {
    q128 p[3], h0[3], h1[3];
    for (int g = 0; g < 3; ++g) p[g] = q_exp(gl[s * 3 + g]);
    hwe((q128)freq[s], 0, h0);
    hwe((q128)freq[s], 1, h1);
    e[s * 2] = p[0] * h0[0] + p[1] * h0[1] + p[2] * h0[2];
    e[s * 2 + 1] = p[0] * h1[0] + p[1] * h1[1] + p[2] * h1[2];
  }
}

/* Forward log-likelihood and posteriors P(IBD at s | data) of one individual.
 * gl [S][3] natural-log likelihoods, freq [S], pos_dist [S] (Mb, +inf at chromosome starts),
 * indF, alpha.  post (may be NULL) receives the UNSNAPPED posterior of state 1 per site.
 * Returns the log-likelihood rounded to double; *lkl_err (may be NULL) is unused slack. */
double hp_forward_backward(const double* gl, const double* freq, const double* pos_dist,
                           uint64_t S, double indF, double alpha, double* post) {
  const q128 q1 = (q128)indF, q0 = 1 - q1;
  q128* e = (q128*)malloc(sizeof(q128) * S * 2);
  q128* fw = (q128*)malloc(sizeof(q128) * (S + 1) * 2);
  q128* sc = (q128*)malloc(sizeof(q128) * (S + 1));
  emissions(gl, freq, S, e);
  fw[0] = q0;
  fw[1] = q1;
  q128 loglik = 0;
  for (uint64_t s = 0; s < S; ++s) {
    const double d = pos_dist[s];
    const q128 c = (d > 1e300) ? (q128)0 : expq(-(q128)alpha * (q128)d);
    const q128 a = 1 - c;
    const q128 tot = fw[s * 2] + fw[s * 2 + 1];
    q128 n0 = (c * fw[s * 2] + a * q0 * tot) * e[s * 2];
    q128 n1 = (c * fw[s * 2 + 1] + a * q1 * tot) * e[s * 2 + 1];
    const q128 z = n0 + n1;
    sc[s] = z;
    loglik += logq(z);
    fw[(s + 1) * 2] = n0 / z;
    fw[(s + 1) * 2 + 1] = n1 / z;
  }
  if (post) {
    q128 b0 = 1, b1 = 1;
    for (uint64_t s = S; s >= 1; --s) {
      const q128 x0 = fw[s * 2] * b0, x1 = fw[s * 2 + 1] * b1;
      post[s - 1] = (double)(x1 / (x0 + x1));
      const double d = pos_dist[s - 1];
      const q128 c = (d > 1e300) ? (q128)0 : expq(-(q128)alpha * (q128)d);
      const q128 a = 1 - c;
      const q128 u0 = e[(s - 1) * 2] * b0, u1 = e[(s - 1) * 2 + 1] * b1;
      const q128 mix = a * (q0 * u0 + q1 * u1);
      b0 = (c * u0 + mix) / sc[s - 1];
      b1 = (c * u1 + mix) / sc[s - 1];
    }
  }
  free(e);
  free(fw);
  free(sc);
  return (double)loglik;
}

/* est_maf (shared/gen_func.cpp:974-1009) of one site in binary128: start 0.01, num/den
 * accumulate over the passes, <= 101 passes, stop on |delta| <= 1e-5.  gl_site [n_ind][3]
 * log likelihoods, indF [n_ind] the IBD posteriors at the site.  *n_passes (may be NULL). */
double hp_est_maf(uint64_t n_ind, const double* gl_site, const double* indF, int* n_passes) {
  q128 num = 0, den = 0, freq = (q128)0.01, prev;
  q128* p = (q128*)malloc(sizeof(q128) * n_ind * 3);
  for (uint64_t k = 0; k < n_ind * 3; ++k) p[k] = q_exp(gl_site[k]);
  int iters = 0, passes = 0, again;
  do {
    prev = freq;
    ++passes;
    for (uint64_t i = 0; i < n_ind; ++i) {
      q128 h[3];
      const q128 F = (q128)indF[i];
      hwe(freq, F, h);
      const q128 w0 = p[i * 3] * h[0], w1 = p[i * 3 + 1] * h[1], w2 = p[i * 3 + 2] * h[2];
      const q128 sum = w0 + w1 + w2;
      num += (w1 + w2 * (2 - F)) / sum;
      den += (2 * w1 + (w0 + w2) * (2 - F)) / sum;
    }
    freq = num / den;
    again = (fabsq(prev - freq) > (q128)1e-5) && (iters++ < 100);
  } while (again);
  free(p);
  if (n_passes) *n_passes = passes;
  return (double)freq;
}

/* ---- --freq_est 2 as intended: the haplotype-frequency EM of a site pair in binary128 ----
 * The model: individual i carries genotype probabilities a_i[g] at the first site and b_i[g] at
 * the second (posteriors, given here in linear space as doubles); with haplotype frequencies f
 * over (first allele, second allele) in {BA, Ba, bA, ba} the probability of the individual's
 * data is sum over ordered haplotype pairs (k, h) of f_k f_h a[g1(k, h)] b[g2(k, h)], and one EM
 * step sets f_k to the expected share of haplotype k among the 2 n haplotypes (gen_func.cpp:
 * 1076-1119 computes the same from a different arrangement of the sum).  Iterated from the
 * product of the marginal frequencies until no frequency moves by 1e-5, at most 100 times
 * (gen_func.cpp:1027-1063).  Returns the number of iterations; hap_out receives f. */
int hp_haplo_freq(double hap_out[4], const double* a, const double* b, double maf1, double maf2,
                  uint64_t n) {
  q128 f[4] = {(1 - (q128)maf1) * (1 - (q128)maf2), (1 - (q128)maf1) * (q128)maf2,
               (q128)maf1 * (1 - (q128)maf2), (q128)maf1 * (q128)maf2};
  int it;
  for (it = 0; it < 100; ++it) {
    q128 cnt[4] = {0, 0, 0, 0};
    for (uint64_t i = 0; i < n; ++i) {
      q128 w[4][4], tot = 0;
      for (int k = 0; k < 4; ++k)
        for (int h = 0; h < 4; ++h) {
          const int g1 = (k >> 1) + (h >> 1), g2 = (k & 1) + (h & 1);
          w[k][h] = f[k] * f[h] * (q128)a[3 * i + g1] * (q128)b[3 * i + g2];
          tot += w[k][h];
        }
      /* expected copies of haplotype k in individual i: once as the first, once as the second
       * member of the ordered pair */
      for (int k = 0; k < 4; ++k) {
        q128 c = 0;
        for (int h = 0; h < 4; ++h) c += w[k][h] + w[h][k];
        cnt[k] += c / tot;
      }
    }
    q128 nf[4], norm = 0, eps = 0;
    for (int k = 0; k < 4; ++k) {
      nf[k] = cnt[k] / (2 * (q128)n);
      norm += nf[k];
    }
    for (int k = 0; k < 4; ++k) {
      nf[k] /= norm; /* (the counts already sum to 2 n: norm = 1 up to rounding) */
      const q128 d = fabsq(nf[k] - f[k]);
      if (d > eps) eps = d;
      f[k] = nf[k];
    }
    if (eps < (q128)1e-5) break;
  }
  for (int k = 0; k < 4; ++k) hap_out[k] = (double)f[k];
  return it;
}

/* calc_emissionLD's model (shared/HMM.cpp:175-236): P(data at both sites | state F) / P(data at
 * the previous site | state F), the pair's genotypes drawn from two haplotypes (F = 0) or from
 * one haplotype twice (F = 1); linear likelihoods exp(gl).  Returned as a log. */
double hp_emission_ld(const double hap[4], const double gl_p[3], const double gl_c[3], double maf_p,
                      int F) {
  q128 sp[3], sc[3], joint = 0, hw[3];
  for (int g = 0; g < 3; ++g) {
    sp[g] = q_exp(gl_p[g]);
    sc[g] = q_exp(gl_c[g]);
  }
  if (F == 0) {
    for (int k = 0; k < 4; ++k)
      for (int h = 0; h < 4; ++h)
        joint += (q128)hap[k] * (q128)hap[h] * sp[(k >> 1) + (h >> 1)] * sc[(k & 1) + (h & 1)];
  } else {
    for (int k = 0; k < 4; ++k) joint += (q128)hap[k] * sp[2 * (k >> 1)] * sc[2 * (k & 1)];
  }
  hwe((q128)maf_p, (q128)F, hw);
  const q128 marg = sp[0] * hw[0] + sp[1] * hw[1] + sp[2] * hw[2];
  return (double)(logq(joint) - logq(marg));
}

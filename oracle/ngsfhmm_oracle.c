/*
 * ngsfhmm_oracle.c -- CPU restatement of ngsF-HMM's EM hot path (see the header
 * for status and rules: TEST INFRASTRUCTURE ONLY).
 *
 * Operation order follows the reference line by line so that, with the same
 * libm, results are the reference's.  Build with -ffp-contract=off.
 */
#include "ngsfhmm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORC_DETMATH
#include "../ngsf-hmm_amd/csrc/detmath.h"
#define ORC_EXP det_exp
#define ORC_LOG det_log
int orc_detmath(void) { return 1; }
#else
#define ORC_EXP exp
#define ORC_LOG log
int orc_detmath(void) { return 0; }
#endif

double orc_exp(double x) { return ORC_EXP(x); }
double orc_log(double x) { return ORC_LOG(x); }

/* shared/gen_func.hpp:20-22: the reference's abs/min/max are macros. */
#define ORC_ABS(x) ((x) >= 0 ? (x) : -(x))
#define ORC_MAX(a, b) ((a) >= (b) ? (a) : (b))

/* ------------------------------------------------------------------ */
/* numerical kernels                                                   */
/* ------------------------------------------------------------------ */

/* shared/gen_func.cpp:135-151 */
double orc_logsum(const double* a, uint64_t n) {
  double sum = 0;
  double M = a[0];
  for (uint64_t i = 1; i < n; i++) M = ORC_MAX(a[i], M);
  if (M == -INFINITY) return -INFINITY;
  for (uint64_t i = 0; i < n; i++) sum += ORC_EXP(a[i] - M);
  return ORC_LOG(sum) + M;
}

static double logsum2(double a, double b) { /* gen_func.cpp:155-160 */
  double buf[2] = {a, b};
  return orc_logsum(buf, 2);
}

static double logsum3(double a, double b, double c) { /* gen_func.cpp:164-170 */
  double buf[3] = {a, b, c};
  return orc_logsum(buf, 3);
}

/* shared/HMM.cpp:130-139 */
double orc_calc_trans(int k, int l, double q_l, double alpha, double pos_dist) {
  double trans = 0;
  double coanc_change = ORC_EXP(-alpha * pos_dist);
  trans = (1 - coanc_change) * q_l;
  if (k == l) trans += coanc_change;
  return ORC_LOG(trans);
}

/* shared/gen_func.cpp:123-130 with func = log */
static void conv_space_log(double* geno, int n) {
  for (int g = 0; g < n; g++) {
    geno[g] = ORC_LOG(geno[g]);
    if (geno[g] == -INFINITY) geno[g] = -ORC_INF;
  }
}

/* shared/gen_func.cpp:123-130 with func = exp */
static void conv_space_exp(double* geno, int n) {
  for (int g = 0; g < n; g++) {
    geno[g] = ORC_EXP(geno[g]);
    if (geno[g] == -INFINITY) geno[g] = -ORC_INF;
  }
}

/* shared/gen_func.cpp:886-914 with the defaults the reference calls it with
 * (ngsF-HMM.cpp:105; gen_func.hpp: log scale, both thresholds 0, miss_data 0);
 * array_max_pos / array_min_pos: gen_func.cpp:73-98 (first maximum, first minimum) */
void orc_call_geno(double geno[3]) {
  int max_pos = 0, min_pos = 0;
  double mx = -INFINITY, mn = INFINITY;
  for (int g = 0; g < 3; g++)
    if (geno[g] > mx) { max_pos = g; mx = geno[g]; }
  for (int g = 0; g < 3; g++)
    if (geno[g] < mn) { min_pos = g; mn = geno[g]; }
  double max_pp = ORC_EXP(geno[max_pos]);
  if (geno[min_pos] == geno[max_pos]) max_pp = -1; /* missing data */
  if (max_pp < 0)
    for (int g = 0; g < 3; g++) geno[g] = ORC_LOG((double)1 / 3);
  if (max_pp >= 0) {
    for (int g = 0; g < 3; g++) geno[g] = -ORC_INF;
    geno[max_pos] = ORC_LOG(1);
  }
}

/* What one cell goes through between the input file and the EM: read_geno's conversion
 * to log space and first normalisation (shared/read_data.cpp:36-40,89-98), then the
 * optional genotype call and the second normalisation (ngsF-HMM.cpp:101-117). */
void orc_prepare_gl(double* gl, uint64_t n_cells, int space, int call_geno) {
  for (uint64_t c = 0; c < n_cells; c++) {
    double* g = gl + 3 * c;
    unsigned long long bits;
    memcpy(&bits, g, sizeof bits);
    if (bits == 0x7ff8dead00000001ull) {             /* cell of an empty text line: */
      g[0] = g[1] = g[2] = -ORC_INF;                 /* read_data.cpp:21,60-61 */
    } else {
      if (space == 1) conv_space_log(g, 3);          /* binary file: read_data.cpp:36-37 */
      if (space == 2)                                /* text file: plain log, :89 */
        for (int k = 0; k < 3; k++) g[k] = ORC_LOG(g[k]);
      orc_post_prob(g, g, NULL);
    }
    if (call_geno) orc_call_geno(g);
    orc_post_prob(g, g, NULL);
  }
}

/* shared/gen_func.cpp:938-957.  pow(x,2) is x*x in the reference's -O3 build. */
void orc_calc_hwe(double out[3], double maf, double F, int log_scale) {
  out[0] = (1 - maf) * (1 - maf) + (1 - maf) * maf * F;
  out[1] = 2 * (1 - maf) * maf - 2 * (1 - maf) * maf * F;
  out[2] = maf * maf + (1 - maf) * maf * F;
  if (log_scale) conv_space_log(out, 3);
  if (F == 1) {
    if (log_scale)
      out[1] = -ORC_INF;
    else
      out[1] = 1 / ORC_INF;
  }
}

/* shared/gen_func.cpp:920-932 */
void orc_post_prob(double pp[3], const double lkl[3], const double* prior) {
  for (int g = 0; g < 3; g++) {
    pp[g] = lkl[g];
    if (prior != NULL) pp[g] += prior[g];
  }
  double norm = orc_logsum(pp, 3);
  for (int g = 0; g < 3; g++) pp[g] -= norm;
}

/* shared/gen_func.cpp:55-70 */
double orc_check_interv(double value, int* is_nan) {
  if (isnan(value)) {
    if (is_nan) *is_nan = 1;
    return value;
  }
  if (value < ORC_EPSILON)
    value = 0;
  else if (value > 1 - ORC_EPSILON)
    value = 1;
  return value;
}

/* shared/HMM.cpp:144-154 */
double orc_calc_emission(const double gl[3], double maf, int k, int* bad) {
  if (maf < 0 || maf > 1) {
    if (bad) *bad = 1;
    return NAN;
  }
  double geno[3];
  orc_calc_hwe(geno, maf, (double)k, 1);
  return logsum3(gl[0] + geno[0], gl[1] + geno[1], gl[2] + geno[2]);
}

/* shared/gen_func.cpp:974-1009 (ignore_miss_data = false, indF != NULL) */
double orc_est_maf(uint64_t n_ind, const double* gl_site, const double* indF, int* n_passes) {
  int iters = 0;
  int passes = 0;
  double num = 0;
  double den = 0;
  double F, prev_freq, freq = 0.01;
  double prior[3], pp[3];
  do {
    prev_freq = freq;
    passes++;
    for (uint64_t i = 0; i < n_ind; i++) {
      F = indF[i];
      orc_calc_hwe(prior, freq, F, 1);
      orc_post_prob(pp, gl_site + 3 * i, prior);
      conv_space_exp(pp, 3);
      num += pp[1] + pp[2] * (2 - F);
      den += 2 * pp[1] + (pp[0] + pp[2]) * (2 - F);
    }
    freq = num / den;
  } while (ORC_ABS(prev_freq - freq) > ORC_EPSILON && iters++ < 100);
  if (n_passes) *n_passes = passes;
  return freq;
}

/* ------------------------------------------------------------------ */
/* --freq_est 2 / --e_prob 2 as INTENDED (opt-in; PARITY UNPINNED: the    */
/* reference aborts on both, SURVEY.md finding 3)                          */
/* ------------------------------------------------------------------ */
/* shared/gen_func.cpp:1070-1071 */
#define ORC_G1(h, k) (((h) >> 1 & 1) + ((k) >> 1 & 1))
#define ORC_G2(h, k) (((h) & 1) + ((k) & 1))

/* shared/gen_func.cpp:1076-1119, ignore_miss_data = false (EM.cpp:237): one EM iteration of
 * the four haplotype frequencies of a pair of sites from per-individual genotype
 * probabilities in NORMAL space, s1 / s2 = [n][3].  The in-place normalisation at the end
 * divides f[1] by a sum that already holds the normalised f[0], and so on: kept. */
uint64_t orc_pair_freq_iter(double f[4], const double* s1, const double* s2, uint64_t n) {
  double ff[4] = {0, 0, 0, 0};
  uint64_t x = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const double* p0 = s1 + 3 * i;
    const double* p1 = s2 + 3 * i;
    double sum, tmp;
    x++;
    sum = 0;
    for (int k = 0; k < 4; ++k)
      for (int h = 0; h < 4; ++h) sum += f[k] * f[h] * p0[ORC_G1(k, h)] * p1[ORC_G2(k, h)];
    for (int k = 0; k < 4; ++k) {
      tmp = 0;
      for (int h = 0; h < 4; ++h)
        tmp += f[k] * f[h] *
               (p0[ORC_G1(h, k)] * p1[ORC_G2(h, k)] + p0[ORC_G1(k, h)] * p1[ORC_G2(k, h)]);
      ff[k] += tmp / sum;
    }
  }
  for (int k = 0; k < 4; ++k) f[k] = ff[k] / (2 * x);
  for (int k = 0; k < 4; k++) f[k] /= f[0] + f[1] + f[2] + f[3];
  return x;
}

/* shared/gen_func.cpp:1027-1063 on the normal-space iteration (the log-space one discards a
 * logsum result at :1160 and returns NaN frequencies; the intended path is :1076-1119).
 * Returns the number of iterations, or -5 for "invalid allele frequencies" (:1030-1031). */
int orc_haplo_freq(double hap_freq[4], const double* gl1, const double* gl2, double maf1,
                   double maf2, uint64_t n_ind) {
  double last[4];
  if (maf1 < 0 || maf1 > 1 || maf2 < 0 || maf2 > 1) return -5;
  hap_freq[0] = (1 - maf1) * (1 - maf2);
  hap_freq[1] = (1 - maf1) * maf2;
  hap_freq[2] = maf1 * (1 - maf2);
  hap_freq[3] = maf1 * maf2;
  int n_iter;
  for (n_iter = 0; n_iter < 100; n_iter++) { /* ITER_MAX, gen_func.hpp:18 */
    double eps = 0;
    memcpy(last, hap_freq, sizeof last);
    orc_pair_freq_iter(hap_freq, gl1, gl2, n_ind);
    for (int j = 0; j < 4; j++) {
      double x = fabs(hap_freq[j] - last[j]);
      if (x > eps) eps = x;
    }
    if (eps < ORC_EPSILON) break;
  }
  return n_iter;
}

/* shared/HMM.cpp:216-236 (F_p == F_c; pow(x, 2) is x * x) */
double orc_joint_geno_prob(const double h[4], int g_p, int g_c, int F) {
  if (g_p == 0 && g_c == 0) return F == 0 ? h[0] * h[0] : h[0];
  if (g_p == 0 && g_c == 1) return F == 0 ? 2 * h[0] * h[1] : 0;
  if (g_p == 0 && g_c == 2) return F == 0 ? h[1] * h[1] : h[1];
  if (g_p == 1 && g_c == 0) return F == 0 ? 2 * h[0] * h[2] : 0;
  if (g_p == 1 && g_c == 1) return F == 0 ? 2 * (h[0] * h[3] + h[1] * h[2]) : 0;
  if (g_p == 1 && g_c == 2) return F == 0 ? 2 * h[1] * h[3] : 0;
  if (g_p == 2 && g_c == 0) return F == 0 ? h[2] * h[2] : h[2];
  if (g_p == 2 && g_c == 1) return F == 0 ? 2 * h[2] * h[3] : 0;
  if (g_p == 2 && g_c == 2) return F == 0 ? h[3] * h[3] : h[3];
  return -1;
}

/* shared/HMM.cpp:175-212, the live branch (:203-211): emission of the current site given the
 * previous one through the pair's haplotype frequencies */
double orc_calc_emission_ld(const double hap_freq[4], const double gl_p[3], const double gl_c[3],
                            double maf_p, double maf_c, int F, int* bad) {
  if (maf_p < 0 || maf_p > 1 || maf_c < 0 || maf_c > 1) {
    if (bad) *bad = 1;
    return NAN;
  }
  double s_p[3], s_c[3];
  for (int g = 0; g < 3; g++) {
    s_p[g] = ORC_EXP(gl_p[g]);
    s_c[g] = ORC_EXP(gl_c[g]);
  }
  double sum = 0;
  for (int g_c = 0; g_c < 3; g_c++)
    for (int g_p = 0; g_p < 3; g_p++)
      sum += orc_joint_geno_prob(hap_freq, g_p, g_c, F) * s_p[g_p] * s_c[g_c];
  return ORC_LOG(sum) - orc_calc_emission(gl_p, maf_p, F, bad);
}

/* shared/HMM.cpp:6-28 */
int orc_forward(double* Fw, const double q[2], double alpha, const double* e_prob,
                const double* pos_dist, uint64_t S, double* lkl) {
  double prev[2], cur[2], tmp[2];
  for (int k = 0; k < 2; k++) prev[k] = ORC_LOG(q[k]);
  if (Fw) {
    Fw[0] = prev[0];
    Fw[1] = prev[1];
  }
  for (uint64_t s = 1; s <= S; s++) {
    for (int l = 0; l < 2; l++) {
      for (int k = 0; k < 2; k++) {
        tmp[k] = prev[k] + orc_calc_trans(k, l, q[l], alpha, pos_dist[s - 1]);
        if (isnan(tmp[k])) return -1;
      }
      cur[l] = orc_logsum(tmp, 2) + e_prob[2 * (s - 1) + l];
    }
    prev[0] = cur[0];
    prev[1] = cur[1];
    if (Fw) {
      Fw[2 * s] = cur[0];
      Fw[2 * s + 1] = cur[1];
    }
  }
  *lkl = orc_logsum(prev, 2);
  return 0;
}

/* shared/HMM.cpp:33-60 */
int orc_backward(double* Bw, const double q[2], double alpha, const double* e_prob,
                 const double* pos_dist, uint64_t S, double* lkl) {
  double tmp[2];
  for (int k = 0; k < 2; k++) Bw[2 * S + k] = ORC_LOG(1);
  for (uint64_t s = S; s > 0; s--) {
    for (int k = 0; k < 2; k++) {
      for (int l = 0; l < 2; l++) {
        tmp[l] = orc_calc_trans(k, l, q[l], alpha, pos_dist[s - 1]) + e_prob[2 * (s - 1) + l] +
                 Bw[2 * s + l];
        if (isnan(tmp[l])) return -1;
      }
      Bw[2 * (s - 1) + k] = orc_logsum(tmp, 2);
    }
  }
  for (int k = 0; k < 2; k++) Bw[k] += ORC_LOG(q[k]);
  *lkl = orc_logsum(Bw, 2);
  return 0;
}

/* shared/gen_func.cpp:73-84 */
static int array_max_pos(const double* a, int n) {
  int res = 0;
  double mx = -INFINITY;
  for (int c = 0; c < n; c++)
    if (a[c] > mx) {
      res = c;
      mx = a[c];
    }
  return res;
}

/* shared/HMM.cpp:98-125.  Vi_prob is updated in place inside the l loop, so
 * state 1 at site s reads state 0's value for site s (reference behaviour). */
double orc_viterbi(const double q[2], double alpha, const double* e_prob, const double* pos_dist,
                   uint64_t S, char* path) {
  double Vi_prob[2];
  uint8_t* back = (uint8_t*)malloc(2 * (S + 1));
  for (int k = 0; k < 2; k++) Vi_prob[k] = ORC_LOG(q[k]);
  for (uint64_t s = 1; s <= S; s++) {
    for (int l = 0; l < 2; l++) {
      double vmax = -ORC_INF;
      int k_vmax = 0;
      for (int k = 0; k < 2; k++) {
        double pval = Vi_prob[k] + orc_calc_trans(k, l, q[l], alpha, pos_dist[s - 1]);
        if (vmax < pval) {
          vmax = pval;
          k_vmax = k;
        }
      }
      back[2 * s + l] = (uint8_t)k_vmax;
      Vi_prob[l] = vmax + e_prob[2 * (s - 1) + l];
    }
  }
  path[S] = (char)array_max_pos(Vi_prob, 2);
  for (uint64_t s = S; s > 0; s--) path[s - 1] = (char)back[2 * s + (int)path[s]];
  double r = Vi_prob[(int)path[S]];
  free(back);
  return r;
}

/* EM.cpp:449-464 */
double orc_lkl(const double* x, const void* data) {
  orc_lkl_data* p = (orc_lkl_data*)data;
  double lkl = 0;
  p->n_calls++;
  if (isnan(x[0]) || isinf(x[0]) || isnan(x[1]) || isinf(x[1])) {
    lkl = ORC_INF;
  } else {
    double F[2] = {1 - x[0], x[0]};
    if (orc_forward(NULL, F, x[1], p->e_prob, p->pos_dist, p->S, &lkl) != 0) {
      p->failed = 1;
      lkl = NAN;
    }
  }
  return -lkl;
}

/* ------------------------------------------------------------------ */
/* EM state                                                            */
/* ------------------------------------------------------------------ */

struct orc_em {
  uint64_t I, S;
  double* gl;       /* [S][I][3] site-major log GL (the binary input order) */
  double* pos_dist; /* [S] */
  double* freq;     /* [S] */
  double* e_prob;   /* [I][S][2] */
  double* marg;     /* [I][S][2] */
  double* indF;     /* [I] */
  double* alpha;    /* [I] */
  double* ind_lkl;  /* [I] */
  double tot_lkl, prev_tot_lkl;
  orc_findmax_fn optimizer;
  uint64_t lkl_calls, maf_passes;
};

orc_em* orc_em_create(uint64_t n_ind, uint64_t n_sites, const double* gl, const double* pos_dist) {
  orc_em* em = (orc_em*)calloc(1, sizeof(orc_em));
  em->I = n_ind;
  em->S = n_sites;
  size_t cells = (size_t)n_ind * n_sites;
  em->gl = (double*)malloc(cells * 3 * sizeof(double));
  memcpy(em->gl, gl, cells * 3 * sizeof(double));
  em->pos_dist = (double*)malloc(n_sites * sizeof(double));
  memcpy(em->pos_dist, pos_dist, n_sites * sizeof(double));
  em->freq = (double*)calloc(n_sites, sizeof(double));
  em->e_prob = (double*)calloc(cells * 2, sizeof(double));
  em->marg = (double*)calloc(cells * 2, sizeof(double));
  em->indF = (double*)calloc(n_ind, sizeof(double));
  em->alpha = (double*)calloc(n_ind, sizeof(double));
  em->ind_lkl = (double*)malloc(n_ind * sizeof(double));
  for (uint64_t i = 0; i < n_ind; i++) em->ind_lkl[i] = -INFINITY; /* parse_args.cpp:412 */
  em->tot_lkl = 0;                                                  /* parse_args.cpp:31-32 */
  em->prev_tot_lkl = 0;
  em->optimizer = orc_findmax_bfgs;
  return em;
}

void orc_em_destroy(orc_em* em) {
  if (!em) return;
  free(em->gl);
  free(em->pos_dist);
  free(em->freq);
  free(em->e_prob);
  free(em->marg);
  free(em->indF);
  free(em->alpha);
  free(em->ind_lkl);
  free(em);
}

void orc_em_set_params(orc_em* em, const double* indF, const double* alpha, const double* freq) {
  if (indF) memcpy(em->indF, indF, em->I * sizeof(double));
  if (alpha) memcpy(em->alpha, alpha, em->I * sizeof(double));
  if (freq) memcpy(em->freq, freq, em->S * sizeof(double));
}

void orc_em_set_optimizer(orc_em* em, orc_findmax_fn fn) {
  em->optimizer = fn ? fn : orc_findmax_bfgs;
}

static int refresh_emission_site(orc_em* em, uint64_t s) { /* EM.cpp:252-257 */
  int bad = 0;
  for (uint64_t i = 0; i < em->I; i++)
    for (int k = 0; k < 2; k++)
      em->e_prob[(i * em->S + s) * 2 + k] =
          orc_calc_emission(em->gl + (s * em->I + i) * 3, em->freq[s], k, &bad);
  return bad ? -3 : 0;
}

/* parse_args.cpp:372-387 */
int orc_em_init_emission(orc_em* em) {
  int rc = 0;
  for (uint64_t s = 0; s < em->S; s++)
    if (refresh_emission_site(em, s) != 0) rc = -3;
  return rc;
}

/* EM.cpp:147-185 */
int orc_em_estep(orc_em* em, int n_threads) {
  const uint64_t I = em->I, S = em->S;
  int rc = 0;
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
  for (uint64_t i = 0; i < I; i++) {
    double* Fw = (double*)malloc((S + 1) * 2 * sizeof(double));
    double* Bw = (double*)malloc((S + 1) * 2 * sizeof(double));
    double q[2] = {1 - em->indF[i], em->indF[i]}; /* EM.cpp:415 */
    const double* e = em->e_prob + i * S * 2;
    double lf = 0, lb = 0;
    int err = 0;
    if (orc_forward(Fw, q, em->alpha[i], e, em->pos_dist, S, &lf) != 0) err = -1;
    if (!err && orc_backward(Bw, q, em->alpha[i], e, em->pos_dist, S, &lb) != 0) err = -1;
    if (!err) {
      /* EM.cpp:166-170 */
      double fl = orc_logsum(Fw + 2 * S, 2), bl = orc_logsum(Bw, 2);
      if (ORC_ABS(fl - bl) > 0.001) err = -2;
    }
    if (!err) {
      /* EM.cpp:178-185 */
      em->ind_lkl[i] = orc_logsum(Fw + 2 * S, 2);
      for (uint64_t s = 1; s <= S; s++)
        for (int k = 0; k < 2; k++) {
          int isn = 0;
          em->marg[(i * S + (s - 1)) * 2 + k] =
              orc_check_interv(ORC_EXP(Bw[2 * s + k] + Fw[2 * s + k] - em->ind_lkl[i]), &isn);
          if (isn) err = -4;
        }
    }
    if (err) {
#pragma omp critical
      if (rc == 0) rc = err;
    }
    free(Fw);
    free(Bw);
  }
  return rc;
}

/* EM.cpp:189-206 + thread_slave type 4 (EM.cpp:423-440) */
int orc_em_mstep_indf(orc_em* em, int indF_fixed, int alpha_fixed, int n_threads) {
  const uint64_t I = em->I, S = em->S;
  int rc = 0;
  uint64_t calls = 0;
  if (indF_fixed && alpha_fixed) return 0;
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1) reduction(+ : calls)
  for (uint64_t i = 0; i < I; i++) {
    orc_lkl_data d;
    d.e_prob = em->e_prob + i * S * 2;
    d.pos_dist = em->pos_dist;
    d.S = S;
    d.n_calls = 0;
    d.failed = 0;
    double val[2] = {em->indF[i], em->alpha[i]};
    double l_bound[2] = {1 / ORC_INF, 1 / ORC_INF};
    double u_bound[2] = {1 - l_bound[0], 10};
    int lims[2] = {2, 2};
    if (indF_fixed) {
      l_bound[0] = em->indF[i];
      u_bound[0] = em->indF[i];
    }
    if (alpha_fixed) {
      l_bound[1] = em->alpha[i];
      u_bound[1] = em->alpha[i];
    }
    em->optimizer(2, val, &d, &orc_lkl, NULL, l_bound, u_bound, lims, -1);
    em->indF[i] = val[0];
    em->alpha[i] = val[1];
    calls += d.n_calls;
    if (d.failed) {
#pragma omp critical
      rc = -1;
    }
  }
  em->lkl_calls += calls;
  return rc;
}

/* EM.cpp:210-272 for freq_est 1 / e_prob 1 (freq_est 2 aborts in the reference at
 * s = 1: haplo_freq rejects freq[0] = -1, shared/gen_func.cpp:1030-1031). */
int orc_em_mstep_freq(orc_em* em, int freq_est, int n_threads) {
  const uint64_t I = em->I, S = em->S;
  if (freq_est == 0) return 0;
  if (freq_est == 2) return -5;
  if (freq_est != 1) return -6;
  int rc = 0;
  uint64_t passes = 0;
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads) reduction(+ : passes)
  {
    double* indF = (double*)malloc(I * sizeof(double));
#pragma omp for schedule(static)
    for (uint64_t s = 0; s < S; s++) {
      for (uint64_t i = 0; i < I; i++) indF[i] = em->marg[(i * S + s) * 2 + 1]; /* EM.cpp:226 */
      int np = 0;
      em->freq[s] = orc_est_maf(I, em->gl + s * I * 3, indF, &np); /* EM.cpp:244 */
      passes += (uint64_t)np;
      if (refresh_emission_site(em, s) != 0) {
#pragma omp critical
        rc = -3;
      }
    }
    free(indF);
  }
  em->maf_passes += passes;
  return rc;
}

/* EM.cpp:210-272 with --freq_est 2 and / or --e_prob 2 as INTENDED.  The reference aborts at
 * s = 1 (haplo_freq is handed freq[0] = -1), its log-space pair iteration loses a logsum
 * (gen_func.cpp:1160), and the LD emission of :258-260 sits inside an `if (e_prob_calc == 1
 * || s == 1)` that it can never pass.  Intended here = the loop exactly as written -- sites
 * in order, frequencies updated IN PLACE, so that site s sees the new freq[s-1] and the old
 * freq[s] -- with those three repaired the smallest way: no haplotype step at the first site
 * (it has no previous one; the code already special-cases s == 1 for est_maf and the
 * emissions), the normal-space iteration on the exponentiated posteriors, and the LD emission
 * reachable for s > 1.  PARITY UNPINNED: there is no reference output to compare with.
 * freq_est, e_prob_calc: 1 or 2.  Returns 0, -3 (invalid MAF), -5 (invalid allele
 * frequencies). */
int orc_em_mstep_freq_ld(orc_em* em, int freq_est, int e_prob_calc) {
  const uint64_t I = em->I, S = em->S;
  if ((freq_est != 1 && freq_est != 2) || (e_prob_calc != 1 && e_prob_calc != 2)) return -6;
  double prior[3], hap_freq[4] = {0, 0, 0, 0};
  double* indF = (double*)malloc(I * sizeof(double));
  double* prev_site = (double*)malloc(I * 3 * sizeof(double));
  double* curr_site = (double*)malloc(I * 3 * sizeof(double));
  int rc = 0;
  for (uint64_t s = 0; s < S && rc == 0; s++) { /* reference site s + 1 */
    for (uint64_t i = 0; i < I; i++) {
      indF[i] = em->marg[(i * S + s) * 2 + 1];
      if (s >= 1) {
        orc_calc_hwe(prior, em->freq[s - 1], em->marg[(i * S + s - 1) * 2 + 1], 1);
        orc_post_prob(prev_site + 3 * i, em->gl + ((s - 1) * I + i) * 3, prior);
      }
      orc_calc_hwe(prior, em->freq[s], em->marg[(i * S + s) * 2 + 1], 1);
      orc_post_prob(curr_site + 3 * i, em->gl + (s * I + i) * 3, prior);
    }
    if (s >= 1 && (freq_est == 2 || e_prob_calc == 2)) {
      conv_space_exp(prev_site, (int)(3 * I));
      conv_space_exp(curr_site, (int)(3 * I));
      if (orc_haplo_freq(hap_freq, prev_site, curr_site, em->freq[s - 1], em->freq[s], I) < 0) {
        rc = -5;
        break;
      }
    }
    const double maf_p = s >= 1 ? em->freq[s - 1] : -1; /* before freq[s] is overwritten */
    if (freq_est == 1 || s == 0) {
      int np = 0;
      em->freq[s] = orc_est_maf(I, em->gl + s * I * 3, indF, &np);
      em->maf_passes += (uint64_t)np;
    } else {
      em->freq[s] = hap_freq[1] + hap_freq[3];
    }
    for (uint64_t i = 0; i < I && rc == 0; i++)
      for (int k = 0; k < 2; k++) {
        int bad = 0;
        double* e = em->e_prob + (i * S + s) * 2 + k;
        if (e_prob_calc == 1 || s == 0)
          *e = orc_calc_emission(em->gl + (s * I + i) * 3, em->freq[s], k, &bad);
        else
          *e = orc_calc_emission_ld(hap_freq, em->gl + ((s - 1) * I + i) * 3,
                                    em->gl + (s * I + i) * 3, maf_p, em->freq[s], k, &bad);
        if (bad) rc = -3;
      }
  }
  free(indF);
  free(prev_site);
  free(curr_site);
  return rc;
}

/* EM.cpp:139-289 */
int orc_em_iter(orc_em* em, int freq_est, int indF_fixed, int alpha_fixed, int n_threads,
                int thread_freq) {
  int rc = orc_em_estep(em, n_threads);
  if (rc) return rc;
  rc = orc_em_mstep_indf(em, indF_fixed, alpha_fixed, n_threads);
  if (rc) return rc;
  return orc_em_mstep_freq(em, freq_est, thread_freq ? n_threads : 1);
}

/* EM.cpp:27-103 */
int orc_em_run(orc_em* em, int freq_est, int indF_fixed, int alpha_fixed, int min_iters,
               int max_iters, double min_epsilon, int n_threads) {
  const uint64_t I = em->I;
  int iter = 0;
  double max_lkl_epsilon = -INFINITY;
  double* prev_ind_lkl = (double*)malloc(I * sizeof(double));
  double* eps = (double*)malloc(I * sizeof(double));
  for (uint64_t i = 0; i < I; i++) prev_ind_lkl[i] = eps[i] = -INFINITY;
  while ((em->prev_tot_lkl - em->tot_lkl > min_epsilon || max_lkl_epsilon > min_epsilon ||
          iter < min_iters) &&
         iter < max_iters) {
    iter++;
    int rc = orc_em_iter(em, freq_est, indF_fixed, alpha_fixed, n_threads, 0);
    if (rc) {
      free(prev_ind_lkl);
      free(eps);
      return rc;
    }
    em->prev_tot_lkl = em->tot_lkl;
    em->tot_lkl = 0;
    for (uint64_t i = 0; i < I; i++) {
      em->tot_lkl += em->ind_lkl[i];
      eps[i] = (em->ind_lkl[i] - prev_ind_lkl[i]) / fabs(prev_ind_lkl[i]);
    }
    max_lkl_epsilon = eps[array_max_pos(eps, (int)I)];
    memcpy(prev_ind_lkl, em->ind_lkl, I * sizeof(double));
  }
  free(prev_ind_lkl);
  free(eps);
  return iter;
}

/* EM.cpp:105-116 */
int orc_em_viterbi(orc_em* em, uint8_t* path, int n_threads) {
  const uint64_t I = em->I, S = em->S;
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
  for (uint64_t i = 0; i < I; i++) {
    char* p = (char*)malloc(S + 1);
    double q[2] = {1 - em->indF[i], em->indF[i]};
    orc_viterbi(q, em->alpha[i], em->e_prob + i * S * 2, em->pos_dist, S, p);
    for (uint64_t s = 0; s < S; s++) path[i * S + s] = (uint8_t)p[s + 1];
    free(p);
  }
  return 0;
}

/* EM.cpp:367-376 */
void orc_em_geno_post(orc_em* em, const uint8_t* path, double* out) {
  const uint64_t I = em->I, S = em->S;
  double pp[3], prior[3];
  for (uint64_t s = 0; s < S; s++)
    for (uint64_t i = 0; i < I; i++) {
      orc_calc_hwe(prior, em->freq[s], (double)path[i * S + s], 1);
      orc_post_prob(pp, em->gl + (s * I + i) * 3, prior);
      conv_space_exp(pp, 3);
      memcpy(out + (s * I + i) * 3, pp, 3 * sizeof(double));
    }
}

const double* orc_em_indF(const orc_em* em) { return em->indF; }
const double* orc_em_alpha(const orc_em* em) { return em->alpha; }
const double* orc_em_freq(const orc_em* em) { return em->freq; }
const double* orc_em_ind_lkl(const orc_em* em) { return em->ind_lkl; }
const double* orc_em_marg(const orc_em* em) { return em->marg; }
const double* orc_em_eprob(const orc_em* em) { return em->e_prob; }
double orc_em_tot_lkl(const orc_em* em) { return em->tot_lkl; }
uint64_t orc_em_lkl_calls(const orc_em* em) { return em->lkl_calls; }
uint64_t orc_em_maf_passes(const orc_em* em) { return em->maf_passes; }

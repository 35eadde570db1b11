/*
 * ngsfhmm_oracle.h -- CPU restatement of ngsF-HMM's EM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ngsf-hmm_amd/ (the product) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / the timed CPU
 * baseline.
 *
 * Each function cites the reference lines (into /root/reference) it follows.
 * Arrays are flat and 0-based in the site index: site s here is the reference's
 * site s+1; the reference's virtual site 0 (Fw[0] = log q, pos_dist[0] = inf,
 * freq[0] = -1) is handled inside the recursions.
 *
 * Parity status: the L-BFGS-B core used by orc_em_iter is pinned bit for bit
 * against the reference's own shared/bfgs.cpp compiled from where it lies
 * (oracle/_ref, tests/test_lbfgsb_ref.py), and orc_em_iter can run with that
 * reference object as its optimizer.  The remaining ~300 lines (shared/HMM.cpp,
 * the pop-gen routines of shared/gen_func.cpp, EM.cpp) cannot be compiled here:
 * every one of them includes <gsl/gsl_rng.h>, which this image lacks, and no
 * stand-in header is written for it.  The reference ships no golden vectors for
 * this path either (examples/test.md5 pins outputs of inputs that need R to
 * regenerate).  For those routines parity is therefore UNPINNED against a
 * reference binary; they are checked against closed-form / brute-force answers
 * in tests/test_oracle.py instead.
 *
 * Two builds from the same source:
 *   liboracle_libm.so : exp/log from libm, as the reference calls them.
 *   liboracle_det.so  : exp/log from ngsf-hmm_amd/csrc/detmath.h, the same
 *                       functions the exact-mode HIP kernels use, so GPU results
 *                       can be compared bit for bit.
 */
#ifndef NGSFHMM_ORACLE_H
#define NGSFHMM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_INF 1e15      /* shared/gen_func.hpp:15 */
#define ORC_EPSILON 1e-5  /* shared/gen_func.hpp:16 */

/* 1 when built with detmath, 0 when built with libm. */
int orc_detmath(void);
double orc_exp(double x);
double orc_log(double x);

/* shared/gen_func.cpp:135-151 */
double orc_logsum(const double* a, uint64_t n);
/* shared/HMM.cpp:130-139 */
double orc_calc_trans(int k, int l, double q_l, double alpha, double pos_dist);
/* shared/gen_func.cpp:938-957 */
void orc_calc_hwe(double out[3], double maf, double F, int log_scale);
/* shared/gen_func.cpp:920-932 (prior may be NULL) */
void orc_post_prob(double pp[3], const double lkl[3], const double* prior);
/* call_geno with the reference's defaults (gen_func.cpp:886-914) */
void orc_call_geno(double geno[3]);
/* input preparation of every cell: log conversion (space 0: already log; 1: normal space
 * from a binary file, log 0 -> -1e15; 2: normal space from a text file, plain log),
 * normalisation, optional genotype call, normalisation (read_data.cpp:36-40,89-98;
 * ngsF-HMM.cpp:101-117); in place */
void orc_prepare_gl(double* gl, uint64_t n_cells, int space, int call_geno);
/* shared/gen_func.cpp:55-70; returns NaN flag through *is_nan */
double orc_check_interv(double v, int* is_nan);
/* shared/HMM.cpp:144-154; returns NaN and sets *bad if maf outside [0,1] */
double orc_calc_emission(const double gl[3], double maf, int k, int* bad);
/* shared/gen_func.cpp:974-1009; gl_site = [n_ind][3] log GLs of one site,
 * indF = per-individual IBD posterior at that site. n_passes (optional) gets
 * the number of passes over the individuals. */
double orc_est_maf(uint64_t n_ind, const double* gl_site, const double* indF, int* n_passes);

/* ---- --freq_est 2 / --e_prob 2 as INTENDED (opt-in; parity unpinned: the reference aborts) ----
 * shared/gen_func.cpp:1076-1119: one normal-space EM iteration of the haplotype frequencies
 * f[4] of a site pair; s1, s2 = [n][3] genotype probabilities.  Returns n. */
uint64_t orc_pair_freq_iter(double f[4], const double* s1, const double* s2, uint64_t n);
/* shared/gen_func.cpp:1027-1063 on that iteration; iterations made, or -5 */
int orc_haplo_freq(double hap_freq[4], const double* gl1, const double* gl2, double maf1,
                   double maf2, uint64_t n_ind);
/* shared/HMM.cpp:216-236 and :175-212 */
double orc_joint_geno_prob(const double h[4], int g_p, int g_c, int F);
double orc_calc_emission_ld(const double hap_freq[4], const double gl_p[3], const double gl_c[3],
                            double maf_p, double maf_c, int F, int* bad);

/* shared/HMM.cpp:6-28.  e_prob = [S][2] log emissions of one individual,
 * pos_dist = [S]; Fw = [(S+1)][2] or NULL (likelihood only).  Returns 0, or -1
 * when a NaN appears ("invalid Lkl found!"). */
int orc_forward(double* Fw, const double q[2], double alpha, const double* e_prob,
                const double* pos_dist, uint64_t S, double* lkl);
/* shared/HMM.cpp:33-60 */
int orc_backward(double* Bw, const double q[2], double alpha, const double* e_prob,
                 const double* pos_dist, uint64_t S, double* lkl);
/* shared/HMM.cpp:98-125; path = [S+1] states (0/1), path[0] is the virtual site */
double orc_viterbi(const double q[2], double alpha, const double* e_prob, const double* pos_dist,
                   uint64_t S, char* path);

/* EM.cpp:449-464: objective handed to the optimizer (= -forward log-likelihood). */
typedef struct {
  const double* e_prob;
  const double* pos_dist;
  uint64_t S;
  uint64_t n_calls;
  int failed;
} orc_lkl_data;
double orc_lkl(const double* x, const void* data);

/* Signature of the reference's findmax_bfgs (shared/bfgs.h:54-57). */
typedef double (*orc_findmax_fn)(int numpars, double* invec, const void* dats,
                                 double (*fun)(const double x[], const void*),
                                 void (*dfun)(const double x[], double y[]), double* lowbound,
                                 double* upbound, int* nbd, int noisy);
/* shared/bfgs.cpp:22-138 restated on top of the product's L-BFGS-B core. */
double orc_findmax_bfgs(int numpars, double* invec, const void* dats,
                        double (*fun)(const double x[], const void*),
                        void (*dfun)(const double x[], double y[]), double* lowbound,
                        double* upbound, int* nbd, int noisy);

/* ---- whole-EM state (ngsF-HMM.hpp:13-52 `params`, flat) ---- */
typedef struct orc_em orc_em;

orc_em* orc_em_create(uint64_t n_ind, uint64_t n_sites, const double* gl_site_major,
                      const double* pos_dist);
void orc_em_destroy(orc_em* em);
void orc_em_set_params(orc_em* em, const double* indF, const double* alpha, const double* freq);
/* use another optimizer (e.g. the reference's own findmax_bfgs from oracle/_ref) */
void orc_em_set_optimizer(orc_em* em, orc_findmax_fn fn);
/* parse_args.cpp:372-387: e_prob from the current freq. Returns 0 or -3 ("invalid MAF!"). */
int orc_em_init_emission(orc_em* em);
/* EM.cpp:139-289 iter_EM.  Returns 0, -1 "invalid Lkl found!", -2 "Fw and Bw lkl
 * do not match!", -3 "invalid MAF!", -4 "value is NaN!".  n_threads parallelises
 * the per-individual phases like the reference's pool; thread_freq != 0 also
 * threads the (serial in the reference) allele-frequency loop. */
int orc_em_iter(orc_em* em, int freq_est, int indF_fixed, int alpha_fixed, int n_threads,
                int thread_freq);
/* the E-step half only (EM.cpp:147-185) */
int orc_em_estep(orc_em* em, int n_threads);
/* the indF/alpha M-step only (EM.cpp:189-206) */
int orc_em_mstep_indf(orc_em* em, int indF_fixed, int alpha_fixed, int n_threads);
/* the allele-frequency M-step + emission refresh only (EM.cpp:210-272) */
int orc_em_mstep_freq(orc_em* em, int freq_est, int n_threads);
/* EM.cpp:210-272 with --freq_est 2 and / or --e_prob 2 as intended (see the .c file; parity
 * unpinned: the reference aborts); freq_est, e_prob_calc in {1, 2} */
int orc_em_mstep_freq_ld(orc_em* em, int freq_est, int e_prob_calc);
/* EM.cpp:27-135 EM(): loop + convergence test; returns number of iterations run or <0 */
int orc_em_run(orc_em* em, int freq_est, int indF_fixed, int alpha_fixed, int min_iters,
               int max_iters, double min_epsilon, int n_threads);
/* EM.cpp:105-116: Viterbi for every individual; path = [I][S] bytes 0/1 */
int orc_em_viterbi(orc_em* em, uint8_t* path, int n_threads);
/* EM.cpp:367-376: genotype posteriors [S][I][3] given a path [I][S] */
void orc_em_geno_post(orc_em* em, const uint8_t* path, double* out);

const double* orc_em_indF(const orc_em* em);
const double* orc_em_alpha(const orc_em* em);
const double* orc_em_freq(const orc_em* em);
const double* orc_em_ind_lkl(const orc_em* em);
const double* orc_em_marg(const orc_em* em);   /* [I][S][2] */
const double* orc_em_eprob(const orc_em* em);  /* [I][S][2] */
double orc_em_tot_lkl(const orc_em* em);
uint64_t orc_em_lkl_calls(const orc_em* em);   /* forward passes spent in the optimizer */
uint64_t orc_em_maf_passes(const orc_em* em);  /* est_maf passes summed over sites */

#ifdef __cplusplus
}
#endif
#endif

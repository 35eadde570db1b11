/*
 * nghmm.h -- C ABI of the MI355X-native ngsF-HMM EM hot path.
 *
 * The reference (fgvieira/ngsF-HMM v1.1.0) has no plugin/FFI interface: its EM
 * engine (EM.cpp) calls its numerical routines (shared/HMM.hpp:8-15,
 * shared/gen_func.hpp:94-103, shared/bfgs.h:54-57) directly through pthread-pool
 * tasks (EM.cpp:385-445).  This header is the seam a maintainer would bind
 * instead: each entry point names the reference call sites it replaces.  All
 * per-site-per-individual arithmetic runs in HIP kernels on one GPU per handle;
 * large state stays device-resident between calls.
 *
 * Conventions
 *   - plain pointers and sizes only; `double` is IEEE binary64;
 *   - sites are 0-based here: site s is the reference's site s+1; the
 *     reference's virtual site 0 is implicit;
 *   - every function returns 0 (NGHMM_OK) or a negative code; the codes -1..-5
 *     correspond to the reference's fatal error() messages, which a host maps
 *     back to the same text (nghmm_strerror); nghmm_last_error() gives detail;
 *   - host buffers are owned by the caller, device buffers by the handle;
 *   - a handle is bound to one HIP device and one stream; calls on different
 *     handles may be made from different threads.
 */
#ifndef NGHMM_H
#define NGHMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nghmm_handle nghmm_t;

enum {
  NGHMM_OK = 0,
  NGHMM_ERR_INVALID_LKL = -1, /* "invalid Lkl found!"          shared/HMM.cpp:18-21,45-48 */
  NGHMM_ERR_FW_BW = -2,       /* "Fw and Bw lkl do not match!" EM.cpp:166-170 */
  NGHMM_ERR_INVALID_MAF = -3, /* "invalid MAF!"                shared/HMM.cpp:145-146 */
  NGHMM_ERR_NAN = -4,         /* "value is NaN!"               shared/gen_func.cpp:56-57 */
  NGHMM_ERR_FREQ_EST2 = -5,   /* "invalid allele frequencies": --freq_est 2 aborts in the
                                 reference at the first site (EM.cpp:235-238,
                                 shared/gen_func.cpp:1030-1031) */
  NGHMM_ERR_ARG = -10,        /* bad argument / call order */
  NGHMM_ERR_HIP = -11,        /* HIP runtime error */
  NGHMM_ERR_NOMEM = -12,
  NGHMM_ERR_NOT_PACKABLE = -13 /* a packed handle met a cell that is not a called genotype */
};

/* Arithmetic mode of a handle. */
enum {
  /* Log-space recursions in the reference's operation order, exp/log from
   * csrc/detmath.h: results are bit-identical to the oracle's `det` build. */
  NGHMM_MODE_EXACT = 0,
  /* Linear-space, chunk-parallel scan over sites with rescaling (the
   * throughput path); per-call results within 1e-9 relative of exact mode. */
  NGHMM_MODE_FAST = 1,
  /* OR-ed into the mode: the genotype likelihoods are CALLED GENOTYPES (--call_geno, or a
   * called-genotype input file; ngsF-HMM.cpp:101-117, shared/read_data.cpp:88-98,
   * shared/gen_func.cpp:886-914), of which every cell is one of four -- genotype 0, 1, 2
   * or missing -- and is kept as a 2-bit code (0.25 B instead of 24 B per site and
   * individual).  Results equal those of an unpacked handle given the same cells: bit for bit
   * in exact mode.  A loader that meets a cell which is not a called genotype (one-hot or
   * uniform likelihoods) returns NGHMM_ERR_NOT_PACKABLE -- the host then falls back to an
   * unpacked handle; NGHMM_ERR_ARG is for input that is wrong either way: a reader genotype
   * > 2, or uniform cells carrying different values within one data set. */
  NGHMM_GENO_PACKED = 0x10
};

/* Statistics of one indF/alpha M-step (shared/bfgs.cpp rounds). */
typedef struct {
  uint32_t rounds;          /* objective rounds: the longest sequence of evaluations any
                               individual needed (= lock-step launches when every round
                               covers all individuals; the two-lane M-step launches more) */
  uint64_t points;          /* objective evaluations sent to the GPU                */
  uint64_t ref_forward_calls; /* forward passes the reference would have spent      */
  uint64_t ind_rounds;      /* sum over rounds of the individuals still being optimised:
                               each costs one pass over that individual's emissions */
} nghmm_mstep_stats;

/* The message of the last failing call ON THE CALLING THREAD (thread-local storage: handles may
 * be driven from several host threads -- replicas, the members of a group or chain -- and each
 * thread sees its own calls' messages; a failure inside a library-owned worker thread of
 * nghmm_group_* / nghmm_chain_* is carried back to the thread that made the call).  Valid until
 * that thread's next library call; "" when that call succeeded. */
const char* nghmm_last_error(void);
const char* nghmm_strerror(int code);
/* 1 if the library was built with its HIP kernels (always, for the shipped .so). */
int nghmm_has_hip(void);

/* Create the state for n_ind individuals x n_sites sites on HIP device `device`
 * (replaces the allocations of read_geno/init_output, shared/read_data.cpp:21,
 * parse_args.cpp:245-412, and iter_EM's per-iteration Fw/Bw, EM.cpp:140-143). */
int nghmm_create(nghmm_t** out, uint64_t n_ind, uint64_t n_sites, int device, int mode);
int nghmm_destroy(nghmm_t* h);

/* Multi-start (ngsF-HMM.sh:77-101 runs 20 replicates from different random starts and keeps
 * the one with the best likelihood): a REPLICA shares its parent's genotype likelihoods and
 * distances on the device (read-only, loaded once) and owns everything an EM run writes --
 * parameters, emissions, posteriors, optimizer state -- and its own HIP stream, so R replicas
 * driven from R host threads run R independent EM analyses concurrently on one GPU.  Every
 * call behaves exactly as on an independent handle loaded with the same data.  The parent must
 * be loaded, must not be reloaded while replicas exist and must be destroyed last. */
int nghmm_create_replica(nghmm_t** out, nghmm_t* parent);

/* Upload genotype likelihoods: natural-log, normalised, site-major [S][I][3]
 * (= the reference's binary --geno file order, shared/read_data.cpp:28-31, after
 * its normalisation) and per-site distances in Mb, +inf at chromosome starts
 * (ngsF-HMM.cpp:75-86).  Host pointers. */
int nghmm_load_gl(nghmm_t* h, const double* gl_site_major, const double* pos_dist_mb);
/* Same from RAW genotype likelihoods as they come out of the input file, prepared on the
 * device in the reference's operation order: conversion to log space, normalisation
 * (shared/read_data.cpp:36-40,89-98), optional genotype calling and second normalisation
 * (ngsF-HMM.cpp:101-117; call_geno, shared/gen_func.cpp:886-914, with its defaults).
 * space: how the values are encoded, see below.  check_nan != 0: a NaN cell after the first
 * normalisation returns NGHMM_ERR_NAN ("NaN found! Is the file format correct?",
 * read_data.cpp:42-45: the reference checks binary input only). */
enum {
  NGHMM_GL_LOG = 0,          /* natural-log likelihoods (--loglkl, or called genotypes) */
  NGHMM_GL_NORMAL_BINARY = 1, /* normal space, binary file: log 0 becomes -1e15 (conv_space) */
  NGHMM_GL_NORMAL_TEXT = 2    /* normal space, text file: plain log (log 0 = -inf)         */
};
/* A cell the reader never filled (the reference lets an empty text line consume a site,
 * read_data.cpp:60-61: its cells keep the initial -1e15 and see only the second
 * normalisation): its FIRST value carries this quiet-NaN bit pattern. */
#define NGHMM_GL_UNREAD_BITS 0x7ff8dead00000001ull
int nghmm_load_gl_raw(nghmm_t* h, const double* gl_raw_site_major, int space, int call_geno,
                      int check_nan, const double* pos_dist_mb);

/* Same, from buffers already resident on the handle's device. */
int nghmm_load_gl_device(nghmm_t* h, const double* d_gl_site_major, const double* d_pos_dist_mb);

/* Chunked loading, for inputs that should never exist in one piece on the host (the file
 * goes to the device a block of sites at a time; read_geno, shared/read_data.cpp:13-116,
 * holds the whole matrix): nghmm_load_begin with the distances, then every site exactly once
 * in chunks [site_begin, site_begin + n_sites) of any size and order, then nghmm_load_end.
 * "Exactly once" is checked: a chunk that overlaps sites this load has already received is
 * NGHMM_ERR_ARG (nothing of it is taken), and so is nghmm_load_end before all n_sites have
 * arrived.  A chunk that fails for another reason (NGHMM_ERR_NAN, NGHMM_ERR_NOT_PACKABLE, a
 * genotype > 2) may have left cells behind: it ends the load, which starts again with
 * nghmm_load_begin.
 *   nghmm_load_gl_raw_sites      raw likelihoods [n_sites][I][3] as nghmm_load_gl_raw takes them
 *   nghmm_load_gl_raw_sites_dev  the same from a device buffer (left unmodified)
 *   nghmm_load_geno_sites        called genotypes [n_sites][I] as the reader sees them, -1
 *                                (missing), 0, 1, 2 (shared/read_data.cpp:88-98); a value > 2 is
 *                                NGHMM_ERR_ARG ("Genotypes must be coded as {-1,0,1,2} !")
 * The whole-matrix loaders above are these three calls in a row.
 * Device buffers (the _dev / _device loaders): the library runs on a stream of its own that
 * waits for no other, so these calls wait for the WHOLE device (hipDeviceSynchronize) before
 * they read the caller's buffer -- whatever stream wrote it, it is complete. */
int nghmm_load_begin(nghmm_t* h, const double* pos_dist_mb);
int nghmm_load_begin_dev(nghmm_t* h, const double* d_pos_dist_mb);
int nghmm_load_gl_raw_sites(nghmm_t* h, uint64_t site_begin, uint64_t n_sites,
                            const double* gl_raw, int space, int call_geno, int check_nan);
int nghmm_load_gl_raw_sites_dev(nghmm_t* h, uint64_t site_begin, uint64_t n_sites,
                                const double* d_gl_raw, int space, int call_geno, int check_nan);
int nghmm_load_geno_sites(nghmm_t* h, uint64_t site_begin, uint64_t n_sites, const int8_t* geno);
int nghmm_load_end(nghmm_t* h);

/* indF[I], alpha[I], freq[S]; NULL leaves a vector unchanged (parse_args.cpp:245-363). */
int nghmm_set_params(nghmm_t* h, const double* indF, const double* alpha, const double* freq);
int nghmm_get_params(nghmm_t* h, double* indF, double* alpha, double* freq);

/* Emission probabilities of every cell from the current freq
 * (calc_emission, shared/HMM.cpp:144-154, as called at parse_args.cpp:381-386). */
int nghmm_emission(nghmm_t* h);

/* E-step: forward, backward, Fw/Bw consistency check, posteriors
 * (EM.cpp:147-185; shared/HMM.cpp:6-60).  ind_lkl[I] (host, may be NULL). */
int nghmm_estep(nghmm_t* h, double* ind_lkl);

/* Objective of the indF/alpha M-step for a batch of probe points: lkl[p] =
 * forward log-likelihood of individual ind[p] with q = (1-F[p], F[p]) and
 * transition rate alpha[p], under the current emissions (EM.cpp:449-464 returns
 * its negative).  Host pointers. */
int nghmm_lkl_batch(nghmm_t* h, uint32_t n_pts, const uint32_t* ind, const double* F,
                    const double* alpha, double* lkl);

/* indF/alpha M-step for all individuals: one bound-constrained L-BFGS-B problem
 * per individual (EM.cpp:198-201,423-440; shared/bfgs.cpp:83-138), advanced in
 * lock-step rounds with nghmm_lkl_batch as the objective.  stats may be NULL. */
int nghmm_mstep_indf(nghmm_t* h, int indF_fixed, int alpha_fixed, nghmm_mstep_stats* stats);

/* The same lock-step batched L-BFGS-B machinery with a caller-supplied objective
 * (host only, no GPU involved): fn returns the forward log-likelihood of
 * individual `ind` at (F, alpha).  indF/alpha are updated in place.  Lets a host
 * plug another objective in, and lets the CPU test-suite exercise the state
 * machines without a device. */
typedef double (*nghmm_objective_fn)(uint32_t ind, double F, double alpha, void* user);
int nghmm_bfgs_batch_host(uint64_t n_ind, double* indF, double* alpha, int indF_fixed,
                          int alpha_fixed, nghmm_objective_fn fn, void* user,
                          nghmm_mstep_stats* stats);
/* The same with flags: 1 = the finite-difference step eh = (1e-8 (|x| + 1))^0.67
 * (shared/bfgs.cpp:33) by the library's own exp / log instead of libm's pow (what fast mode
 * uses: host and device then agree bit for bit); 2 = run the solver type the device runs
 * (kernels_bfgs.hip), one problem after the other.  Test hook: both give identical bits. */
int nghmm_bfgs_batch_host2(uint64_t n_ind, double* indF, double* alpha, int indF_fixed,
                           int alpha_fixed, nghmm_objective_fn fn, void* user,
                           nghmm_mstep_stats* stats, int flags);

/* Allele-frequency M-step + emission refresh (EM.cpp:210-272; est_maf,
 * shared/gen_func.cpp:974-1009).  freq_est 0 = keep, 1 = per-site EM,
 * 2 = NGHMM_ERR_FREQ_EST2 (the reference aborts).
 *
 * OPT-IN, PARITY UNPINNED -- --freq_est 2 / --e_prob 2 AS INTENDED.  The reference aborts on
 * both at the first site (EM.cpp:235-238 -> shared/gen_func.cpp:1030-1031), so there is no
 * reference output; what is computed is its loop (EM.cpp:224-263) as written -- sites in
 * order, frequencies updated in place, haplotype frequencies of every adjacent site pair by
 * the pair EM of gen_func.cpp:1027-1119 -- with its three defects repaired the smallest way
 * (no pair step at the first site; the normal-space pair iteration, the log-space one loses a
 * logsum at :1160; the LD emission of EM.cpp:258-260 reachable).  Asked for by OR-ing
 * NGHMM_LD_INTENDED into freq_est:
 *   2 | NGHMM_LD_INTENDED                    freq[s] from the pair's haplotype frequencies
 *   2 | NGHMM_LD_INTENDED | NGHMM_EPROB_LD   ... and emissions by calc_emissionLD
 *                                            (shared/HMM.cpp:175-236) past the first site
 *   1 | NGHMM_LD_INTENDED | NGHMM_EPROB_LD   est_maf frequencies, LD emissions
 * NGHMM_EPROB_LD needs NGHMM_MODE_EXACT (materialised emissions).  One unsharded handle, at
 * most 8192 individuals; the chain through the sites is sequential by its definition.  Plain
 * 2 keeps returning the reference's abort. */
enum { NGHMM_LD_INTENDED = 0x20, NGHMM_EPROB_LD = 0x40 };
int nghmm_mstep_freq(nghmm_t* h, int freq_est);

/* E-step + indF/alpha M-step of one EM iteration (EM.cpp:147-201) in one call.  In fast
 * mode the two share a pass over the emissions: the M-step's first objective evaluation
 * f(x) at the current parameters (EM.cpp:449-464) IS the E-step's forward walk, so it runs
 * first and leaves the lane operators and checkpoints the E-step's backward sweep needs;
 * results are those of nghmm_estep followed by nghmm_mstep_indf.  after_estep (may be
 * NULL) is called once on the calling thread as soon as the posteriors are final, before
 * the remaining objective rounds: a multi-GPU host starts its posterior exchange there. */
typedef void (*nghmm_hook_fn)(void* user);
int nghmm_estep_mstep(nghmm_t* h, int indF_fixed, int alpha_fixed, double* ind_lkl,
                      nghmm_mstep_stats* stats, nghmm_hook_fn after_estep, void* user);

/* One whole EM iteration = iter_EM (EM.cpp:139-289). */
int nghmm_iter_em(nghmm_t* h, int freq_est, int indF_fixed, int alpha_fixed, double* ind_lkl,
                  nghmm_mstep_stats* stats);

/* Viterbi decoding with the current parameters (EM.cpp:105-116;
 * shared/HMM.cpp:98-125).  path[I][S] bytes 0/1 (host). */
int nghmm_viterbi(nghmm_t* h, uint8_t* path);

/* Posterior of the IBD state from the last E-step, marg_prob[i][s][1], as
 * [I][S] doubles (host) -- what EM.cpp:347-353 prints. */
int nghmm_get_posteriors(nghmm_t* h, double* marg_ibd);
/* Prepared genotype likelihoods [S][I][3] (natural log, normalised) back to the host:
 * what nghmm_load_gl_raw made of its input. */
int nghmm_get_gl(nghmm_t* h, double* gl_site_major);
/* Genotype posteriors of the .geno output (EM.cpp:367-376) for the sites
 * [site_begin, site_begin + n_sites): out[n_sites][n_ind][3] (host, normal space), with the
 * path of the last nghmm_viterbi call as the prior's F (all zeros before the first call, like
 * the reference's path[][] at an intermediate print_iter) and the current frequencies. */
int nghmm_geno_posteriors(nghmm_t* h, uint64_t site_begin, uint64_t n_sites, double* out);
/* The posterior lines of the .ibd file (EM.cpp:347-353) as text, formatted on the device:
 * for the individuals [ind_begin, ind_begin + n_ind) one line each of n_sites values
 * printed like printf("%f") -- "d.dddddd", the digits glibc prints -- separated by tabs and
 * ended by a newline, i.e. exactly 9 * n_sites bytes per individual; out (host) receives
 * n_ind * 9 * n_sites bytes. */
int nghmm_format_posteriors(nghmm_t* h, uint64_t ind_begin, uint64_t n_ind, char* out);
/* The formatter behind it, for caller-supplied values in [0, 1] (host): rows lines of cols
 * "%f" values, 9 * rows * cols bytes.  NGHMM_ERR_ARG if a value is outside [0, 1]. */
int nghmm_format_fixed6(nghmm_t* h, const double* values, uint64_t rows, uint64_t cols, char* out);

/* ---- multi-GPU (individuals sharded over ranks) ----
 * The allele-frequency step needs every individual of a site.  A rank owns the
 * individuals [ind_begin, ind_begin + n_ind) of n_ind_total for all sites, and
 * the sites [site_begin, site_begin + n_sites_own) for the frequency step.
 * The host moves posteriors between ranks with an all-to-all and frequencies with
 * an all-gather (RCCL through torch.distributed or rccl.h); these calls take raw
 * DEVICE pointers to the exchange buffers. */
int nghmm_shard_config(nghmm_t* h, uint64_t n_ind_total, uint64_t ind_begin, uint64_t site_begin,
                       uint64_t n_sites_own);
/* static site-shard copy of the GLs of ALL individuals: [n_sites_own][n_ind_total][3] (host) */
int nghmm_load_gl_site_shard(nghmm_t* h, const double* gl_site_shard);
/* same, from a device buffer */
int nghmm_load_gl_site_shard_dev(nghmm_t* h, const double* d_gl_site_shard);
/* packed handles: the own individuals' genotype codes of the sites [site_lo, site_hi) as one
 * byte per cell (0, 1, 2, 3 = missing), d_out[(s - site_lo) * n_ind + i] (device) -- what the
 * host exchanges once to build the site shards -- and the static site-shard copy from such
 * bytes, [n_sites_own][n_ind_total] (device) */
int nghmm_get_geno_codes_dev(nghmm_t* h, uint64_t site_lo, uint64_t site_hi, uint8_t* d_out);
int nghmm_load_geno_site_shard_dev(nghmm_t* h, const uint8_t* d_codes_bytes);
/* pack posteriors of the own individuals for destination rank r's site range:
 * d_out[(s - site_lo) * n_ind + i], s in [site_lo, site_hi); with [0, n_sites) the whole
 * site-major matrix = the send buffer of all equal contiguous ranges at once */
int nghmm_pack_posteriors_dev(nghmm_t* h, uint64_t site_lo, uint64_t site_hi, double* d_out);
/* d_marg_sites: [n_sites_own][n_ind_total] posteriors of the own site range (device);
 * runs est_maf on them, writes d_freq_out[n_sites_own] (device) */
int nghmm_mstep_freq_sites_dev(nghmm_t* h, const double* d_marg_sites, double* d_freq_out);
/* install the gathered freq[S] (device pointer) and refresh the own emissions */
int nghmm_set_freq_dev(nghmm_t* h, const double* d_freq_all);

/* ---- one process, several GPUs: a GROUP of n handles ----
 * What EM.cpp:147-272 does for all individuals at once, split over n handles (one per GPU; or
 * several on one GPU): handle r owns the individuals [r I, (r+1) I) of n I for all sites --
 * create and load it with those -- and, after nghmm_group_setup, the sites [r S/n, (r+1) S/n)
 * of the allele-frequency step.  nghmm_group_iter_em = iter_EM for the whole cohort: per
 * handle, on its own host thread, the E-step and the indF/alpha M-step; the posteriors move
 * to their site owners by direct peer copies (every GPU pair of an MI355X node has its own
 * xGMI link: the n (n-1) copies are the all-to-all) under the remaining objective rounds;
 * est_maf per site range in global individual order; the frequencies go to everybody.  The
 * result does not depend on n (tests/test_gpu_sharded.py).  Equal I and S, S divisible by n,
 * NGHMM_MODE_FAST for n > 1; ind_lkl [n I] (host, may be NULL).  Between processes the same
 * steps run over RCCL (ngsf-hmm_amd/distributed.py). */
int nghmm_group_setup(nghmm_t** handles, int n);
int nghmm_group_iter_em(nghmm_t** handles, int n, int freq_est, int indF_fixed, int alpha_fixed,
                        double* ind_lkl, nghmm_mstep_stats* stats);
/* the allele-frequency step alone (nghmm_mstep_freq for the cohort), from the posteriors the
 * handles hold: all zero before the first E-step, which is `--freq e` */
int nghmm_group_mstep_freq(nghmm_t** handles, int n, int freq_est);

/* ---- multi-GPU, fast mode: shard the SITES instead ----
 * (what bench.py --gpus N and ngsf-hmm_amd/distributed.py use by default.)  A handle holds ALL
 * individuals for a contiguous range of sites -- created and loaded like a data set of its
 * own, with the true distance in front of its first site (+inf only at a chromosome start) --
 * and the ranges follow each other in rank order.  A run of sites is a product of 2x2
 * operators, so what the ranges owe each other is six doubles per individual and E-step, and
 * per objective point and round: forward / backward vectors, log-likelihoods and objective
 * values are then those of the whole chain, and every handle computes the same values and
 * takes the same L-BFGS-B steps for all individuals (shared/HMM.cpp:6-60 and EM.cpp:423-464
 * over all sites).  The allele-frequency step (EM.cpp:224-247) has every individual of a
 * handle's sites at hand: no posterior ever leaves the GPU (the individual shards above move
 * 8 bytes per site and individual per iteration, 8 GB at 1000 x 1M, over one xGMI link per GPU
 * pair).  Not for NGHMM_MODE_EXACT: its log-space recursion is one chain of roundings over all
 * sites.
 *
 * nghmm_site_shard_setup: send_dev / recv_dev are the caller's device buffers of
 * nghmm_site_shard_bytes(h) and world times that many bytes; `allgather(user, n)` must make
 * recv_dev = [rank][n bytes] of every handle's first n bytes of send_dev, ORDERED ON THE
 * HANDLE'S STREAM (nghmm_stream): the library has enqueued the writes of send_dev there before
 * the call and enqueues the reads of recv_dev after it; the function may block (a host-staged
 * exchange) or only enqueue (an RCCL all-gather on that stream).  It is called from inside
 * nghmm_estep / _lkl_batch / _mstep_indf / _estep_mstep / _iter_em, by every handle of the
 * chain the same number of times with the same n.  Non-zero return = failure (NGHMM_ERR_HIP).
 * world == 1 detaches.  Everything else (est_maf, emissions, posteriors, parameters, output
 * formatting) works on the handle's own sites as on any handle; nghmm_get_params' indF / alpha
 * are the cohort's and equal on all handles.
 *
 * Viterbi (shared/HMM.cpp:98-125) over the chain: nghmm_viterbi_shard_forward in rank order
 * (scores_in = NULL on the first handle, else the scores_out [I][2] of the handle before), then
 * nghmm_viterbi_shard_back in reverse order (state_after = NULL on the last handle, else the
 * state_before [I] of the handle after); path = [I][n_sites of the handle].  The same
 * operations in the same order per individual as nghmm_viterbi on one handle over all sites. */
typedef int (*nghmm_allgather_fn)(void* user, uint64_t n_bytes);
uint64_t nghmm_site_shard_bytes(nghmm_t* h);
int nghmm_site_shard_setup(nghmm_t* h, int rank, int world, void* send_dev, void* recv_dev,
                           uint64_t bytes_per_rank, nghmm_allgather_fn allgather, void* user);
int nghmm_viterbi_shard_forward(nghmm_t* h, const double* scores_in, double* scores_out);
int nghmm_viterbi_shard_back(nghmm_t* h, const uint8_t* state_after, uint8_t* state_before,
                             uint8_t* path);

/* ---- one process, several GPUs, fast mode: a CHAIN of site shards ----
 * n handles of one process (one per GPU, or several on one), handle r holding all individuals
 * for the r-th site range: nghmm_chain_setup installs the all-gather of nghmm_site_shard_setup
 * among them (direct device-to-device copies between the handles' buffers -- over the GPU
 * pair's xGMI link where the devices differ -- between two barriers of the handles' host
 * threads) and owns the buffers; nghmm_chain_iter_em = iter_EM (EM.cpp:139-289) for the whole
 * chain, every handle on a host thread of its own; ind_lkl [I] (host, may be NULL).
 * nghmm_chain_mstep_freq: the allele-frequency step alone (`--freq e`), every handle on its
 * own sites.  nghmm_chain_viterbi: path [I][all sites] (host).  n == 1 is the plain handle.
 * This is what the C++ host's --n_gpus N uses; destroying a member dissolves the chain. */
int nghmm_chain_setup(nghmm_t** handles, int n);
int nghmm_chain_iter_em(nghmm_t** handles, int n, int freq_est, int indF_fixed, int alpha_fixed,
                        double* ind_lkl, nghmm_mstep_stats* stats);
int nghmm_chain_mstep_freq(nghmm_t** handles, int n, int freq_est);
int nghmm_chain_viterbi(nghmm_t** handles, int n, uint8_t* path);

/* Page-locked host memory for buffers the library copies results into (nghmm_get_*,
 * nghmm_format_posteriors, nghmm_geno_posteriors, nghmm_viterbi ...): copies from the device
 * into such a buffer run at the PCIe rate instead of through a staging buffer.  Any host memory
 * works; this is for hosts that move gigabytes (the .ibd / .geno writers).  NULL when it cannot
 * be had. */
void* nghmm_alloc_host(uint64_t bytes);
void nghmm_free_host(void* p);

/* Device pointer + stream access for host-side plumbing (torch tensors, events). */
void* nghmm_stream(nghmm_t* h);
int nghmm_synchronize(nghmm_t* h);
/* Measurement and debugging entry points (switches of a handle, kernel timing, counters of the
 * rare code paths, emission read-back) are declared in nghmm_debug.h: tests, bench.py and the
 * profiling scripts use them; a host that only runs analyses does not need them. */

#ifdef __cplusplus
}
#endif
#endif /* NGHMM_H */

/*
 * nghmm_debug.h -- measurement and debugging entry points of libnghmm.so.
 *
 * Not part of the drop-in boundary (include/nghmm.h): nothing here replaces a call of the
 * reference.  tests/, bench.py, profiles/ and tools/ use these to time kernel families, to
 * force a code path that the sizes at hand would not take, and to count how often the rare
 * paths were taken.  None of them changes a result beyond rounding.
 */
#ifndef NGHMM_DEBUG_H
#define NGHMM_DEBUG_H

#include "nghmm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement and debugging switches of a handle.  None changes a result beyond rounding (the
 * kernels, their order on the stream or what is printed; DESIGN.md section 7 says what each is
 * for).  A handle reads them from the environment ONCE, in nghmm_create -- NGHMM_<NAME> with
 * the name in capitals; a variable that is set without a number counts as 1 -- and a replica
 * inherits its parent's; later only this call changes them:
 *   pipeline          two-lane objective rounds: -1 by cohort size (default), 0 off, 1 on
 *   no_bg             1: backward sweep and est_maf after the objective rounds, not behind them
 *   bg_parts          est_maf goes behind the rounds in this many parts (default 2)
 *   no_fuse           1: E-step and M-step each make a forward walk of their own
 *   eager_emission    1: stored emissions refreshed right after every frequency update
 *   estmaf_interp     0: every est_maf pass evaluated over all individuals (default 1)
 *   estmaf_sitemajor  1: est_maf on a site-major copy of the posteriors
 *   estmaf_no_rows    1: small cohorts take a wave per site instead of four sites per wave
 *   estmaf_no_called  1: called genotypes (packed handles) through the general est_maf kernels
 *                     instead of their closed form (k_fast_estmaf_called_sums)
 *   no_xdeg2          1: the alpha probes' exp((alpha_0 - alpha_probe) d) always by the
 *                     degree-4 polynomial (default: degree 2 where |.| <= 1e-5, the same to
 *                     half an ulp)
 *   exact_serial      1: exact-mode recursions as one lane per chain (kernels_exact.hip)
 *                     instead of producer-consumer workgroups (kernels_exact_pc.hip): same bits
 *   estmaf_exact_lanes 1: exact-mode est_maf with a lane per site instead of a wave per site
 *                     (same bits; measured slower, kept for the comparison)
 *   estmaf_exact_sel  1: exact-mode est_maf on the select forms of det_exp / det_log (same
 *                     bits; measured slower since round 4's kernel, default 0)
 *   exact_bg_waves    exact mode, fused iteration: est_maf runs underneath the objective rounds
 *                     in 16 pieces capped at this many waves per SIMD (default 3; 0: uncapped;
 *                     -1: after the rounds); exact_bg_depth: pieces queued under a round (3)
 *   exact_estep_overlap 0: exact mode's fused iteration runs its E-step before the objective
 *                     rounds instead of next to the first of them (default 1)
 *   timing            1: host-side phase times of every M-step on stderr
 *   debug_modes       1: kernel versions of every objective round on stderr
 *   no_dev_bfgs       1: fast mode's L-BFGS-B machines on the host, every round a round trip
 *                     (rounds 1-4; default 0: on the device, kernels_bfgs.hip -- same results)
 *   no_bg_stream      1: with the machines on the device, backward sweep and est_maf between the
 *                     objective rounds on the handle's one stream instead of next to them on a
 *                     second (what bench.py --serial_kernels and profiles/collect.sh run: every
 *                     kernel's span is then its own)
 *   estmaf_w2         1: est_maf of 513 .. 1024 individuals on two waves of 8 per lane (measured
 *                     slower: 10.1 vs 8.5 ms at 1000 x 1M)
 *   spans             1: fast mode records the timing events behind nghmm_kernel_ms around the
 *                     kernel families of nghmm_mstep_indf / nghmm_estep_mstep / nghmm_iter_em
 *                     (default 0: the events are packets the queue works through between two
 *                     kernels, 0.05 ms per EM iteration -- 8 % of an iteration of 100 x 100 000;
 *                     nghmm_kernel_ms then reads 0 for those calls); exact mode always records
 * Fixed at creation (environment only): fast_c (waves per individual), spin_sync (replicas
 * wait spinning).  Unknown names return NGHMM_ERR_ARG.  Outside the handle: NGHMM_HOST_THREADS
 * (host threads of the L-BFGS-B state machines, read once per process). */
int nghmm_set_switch(nghmm_t* h, const char* name, long value);

/* Fast-mode layout of the site axis: every individual's sites are cut into 64 * waves
 * runs of sites_per_lane sites; 0, 0 in exact mode.  Diagnostic. */
int nghmm_fast_layout(nghmm_t* h, uint32_t* waves_per_individual, uint64_t* sites_per_lane);

/* Current emissions [I][S][2] (host), natural log: what calc_emission (shared/HMM.cpp:144-154)
 * holds in e_prob; fast mode recomputes them from the likelihoods and frequencies (it keeps only
 * their ratio between walks). */
int nghmm_get_emissions(nghmm_t* h, double* e_prob);

/* HIP-event timing of the last call of each kernel family, in milliseconds:
 * 0 emission, 1 forward(store), 2 backward+posterior, 3 lkl_batch (sum over rounds of
 * the last mstep_indf or the last lkl_batch call), 4 est_maf+emission, 5 viterbi, 6 the
 * part of slot 3 spent in the round that doubled as the E-step's forward walk
 * (nghmm_estep_mstep), 7 the kernels that advance the L-BFGS-B machines on the device between
 * two rounds (fast mode; 0 where the host advances them).  Also the launch count behind each
 * slot.  Fast mode's M-step and fused iteration time their kernels only with the switch
 * `spans` (nghmm_set_switch; NGHMM_SPANS=1) and report 0 without it. */
int nghmm_kernel_ms(nghmm_t* h, int slot, double* ms, uint32_t* launches);

/* How often the code paths off the common one were taken since the handle was created (or
 * since the last call with reset != 0).  Fast mode; zeros in exact mode.
 *
 * nghmm_debug_mode_counts: the objective rounds of the indF / alpha M-step (EM.cpp:423-464) run one
 * kernel version per finite-difference pattern (csrc/fast_dev.hpp: fd_pattern); out[k] = (mode,
 * individual-rounds evaluated by that version), n = how many versions were used (at most cap are
 * written).  mode 0 = the general kernel (an exp per point and site); else bits 0-1 alpha probes,
 * bits 2-3 F probes, 0x200 small-alpha (kappa form, polynomial instead of exp), 0x400 degree-2
 * alpha probes, 0x800 an exponent per point.
 *
 * nghmm_debug_estmaf_counts: sites of the allele-frequency step (est_maf, gen_func.cpp:974-1009)
 * that left its common route -- out[0] the interpolant's check failed (the site ran every
 * remaining pass over all individuals), out[1] left its interval and got a second, out[2] a
 * third, out[3] redone in the reference's log-space order (a cell whose linear weights all
 * vanish), out[4] left its third interval too and finished on exact passes. */
typedef struct {
  uint32_t mode;
  uint64_t ind_rounds;
} nghmm_mode_count;
int nghmm_debug_mode_counts(nghmm_t* h, nghmm_mode_count* out, uint32_t cap, uint32_t* n, int reset);
int nghmm_debug_estmaf_counts(nghmm_t* h, uint64_t out[5], int reset);

#ifdef __cplusplus
}
#endif
#endif /* NGHMM_DEBUG_H */

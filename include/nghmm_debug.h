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
 * kernels, or their order on the streams).  A handle reads them from the environment ONCE, in
 * nghmm_create -- NGHMM_<NAME> with the name in capitals; a variable that is set without a number
 * counts as 1 -- and a replica inherits its parent's; later only this call changes them:
 *   spans             1: fast mode records the timing events behind nghmm_kernel_ms around the
 *                     kernel families of nghmm_mstep_indf / nghmm_estep_mstep / nghmm_iter_em (default 0:
 *                     the events are packets the queue works through between two kernels, 0.05 ms per
 *                     EM iteration -- 8 % of an iteration of 100 x 100 000 -- and an iteration then ends
 *                     by a stream synchronisation instead of a polled word; nghmm_kernel_ms reads 0
 *                     for those calls without it); exact mode always records
 *   no_bg_stream      1: backward sweep and est_maf between the objective rounds on the handle's one
 *                     stream instead of next to them on a second (bench.py --serial_kernels,
 *                     profiles/collect.sh: every kernel's span is then its own)
 *   no_dev_bfgs       1: fast mode's L-BFGS-B machines on the host, every round a round trip (what
 *                     exact mode and groups of handles use; default 0: on the device,
 *                     kernels_bfgs.hip -- the same steps, bit for bit)
 *   estmaf_interp     0: every est_maf pass evaluated over all individuals (default 1: a checked
 *                     interpolant carries most of them; tests hold the two together)
 *   estmaf_no_called  1: called genotypes (packed handles) through the general est_maf kernels
 *                     instead of their closed form (k_fast_estmaf_called_sums)
 *   estmaf_no_rows    1: small cohorts take a wave per site instead of four sites per wave
 *   bg_parts          host-planned rounds: est_maf goes behind the rounds in this many parts (default 2;
 *                     0: backward sweep and est_maf after the rounds)
 *   exact_serial      1: exact-mode recursions as one lane per chain (kernels_exact.hip) instead of
 *                     producer-consumer workgroups (kernels_exact_pc.hip): the same bits
 *   dbg_abort_round   test hook: a device-planned M-step returns an error after this round
 * Fixed at creation (environment only): fast_c (waves per individual).  Unknown names return
 * NGHMM_ERR_ARG.  Outside the handle: NGHMM_HOST_THREADS (host threads of the L-BFGS-B state
 * machines, read once per process). */
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
 * alpha probes, 0x800 an exponent per point.  One more entry may follow them, mode 0xffffffff: the
 * number of ROUNDS (not individual-rounds) whose versions shared one launch -- a device-planned round
 * of at most 16384 waves with individuals of several versions (kernels_fast_walks.hip,
 * k_fast_lkl_fd_mix).
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

// kernels.hpp -- launch interface of the HIP kernels (gfx950) behind the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "glview.hpp"

namespace nghmm {

// Error flags raised by kernels (device int[NFLAGS]); the C ABI maps them to the
// reference's fatal messages.
enum Flag { FLAG_INVALID_LKL = 0, FLAG_FW_BW = 1, FLAG_INVALID_MAF = 2, FLAG_NAN = 3,
            FLAG_NOT_PACKABLE = 4, FLAG_BAD_GENO = 5, FLAG_LD_FREQ = 6, NFLAGS = 8 };

// ---------------- exact mode (kernels_exact.hip) ----------------
// All arrays site-major: gl [S][I][3], eprob [S][I][2], fw [S+1][I][2], marg [S][I].

// input preparation in place: log conversion (space 0 log / 1 binary-file normal space /
// 2 text-file normal space), post_prob, optional call_geno, post_prob
// (shared/read_data.cpp:36-40,89-98; ngsF-HMM.cpp:101-117); FLAG_NAN on a NaN cell
void launch_prepare_gl(hipStream_t st, double* gl, uint64_t n_cells, int space, int call_geno,
                       int* flags);

// ---- called genotypes as 2-bit codes (glview.hpp) ----
// prepared dense cells [n_cells][3] -> codes at cell index cell0 + c (codes zeroed beforehand);
// table = [3][3] prepared likelihoods of the called classes; *uniform_bits (initially ~0)
// learns the one value all uniform cells share; FLAG_NOT_PACKABLE on any other cell
void launch_pack_cells(hipStream_t st, const double* gl, uint64_t n_cells, uint64_t cell0,
                       const double* table, uint32_t* codes, unsigned long long* uniform_bits,
                       int* flags);
// reader genotypes (-1 missing, 0, 1, 2) -> codes; FLAG_BAD_GENO on a value > 2
void launch_pack_geno(hipStream_t st, const int8_t* geno, uint64_t n_cells, uint64_t cell0,
                      uint32_t* codes, int* flags);
// reader genotypes -> the raw likelihoods of shared/read_data.cpp:21,88-98 (dense [n_cells][3])
void launch_expand_geno(hipStream_t st, const int8_t* geno, uint64_t n_cells, double log_third,
                        double* gl, int* flags);
void launch_unpack_cells(hipStream_t st, const GlView& gl, uint64_t n_cells, double* out);
void launch_codes_to_bytes(hipStream_t st, const uint32_t* codes, uint64_t cell0, uint64_t n_cells,
                           uint8_t* out);
void launch_bytes_to_codes(hipStream_t st, const uint8_t* in, uint64_t n_cells, uint32_t* codes);

// genotype posteriors of the .geno output (EM.cpp:367-376) for sites s0 .. s0 + n_s from
// the blocked Viterbi path; out [n_s][I][3]
void launch_geno_post_exact(hipStream_t st, const GlView& gl, const double* freq,
                            const uint8_t* path16, uint64_t I, uint64_t s0, uint64_t n_s,
                            double* out);

// e_prob[s][i][k] = calc_emission(gl[s][i], freq[s], k)   (shared/HMM.cpp:144-154)
void launch_emission_exact(hipStream_t st, const GlView& gl, const double* freq, double* eprob,
                           uint64_t S, uint64_t I, int* flags);

// forward recursion (shared/HMM.cpp:6-28) for n_pts (individual, F, alpha) points;
// ind == nullptr means point p is individual p.  fw (nullable) receives Fw.
void launch_forward_exact(hipStream_t st, const double* eprob, const double* pos, uint64_t S,
                          uint64_t I, uint32_t n_pts, const uint32_t* ind, const double* F,
                          const double* alpha, double* lkl_out, double* fw, int* flags);

// backward recursion + posteriors + Fw/Bw check (shared/HMM.cpp:33-60, EM.cpp:166-185)
void launch_backward_exact(hipStream_t st, const double* eprob, const double* pos, const double* fw,
                           uint64_t S, uint64_t I, const double* indF, const double* alpha,
                           const double* ind_lkl, double* marg, int* flags);

// The same two recursions as producer-consumer workgroups (kernels_exact_pc.hip): the
// state-independent transition logs of a block of sites computed in parallel by producer
// waves, the sequential chain reduced to its two logsums per site, two lanes per chain.
// Bit-identical to the serial kernels above; what the C ABI launches.
void launch_forward_exact_pc(hipStream_t st, const double* eprob, const double* pos, uint64_t S,
                             uint64_t I, uint32_t n_pts, const uint32_t* ind, const double* F,
                             const double* alpha, double* lkl_out, double* fw, int* flags);
void launch_backward_exact_pc(hipStream_t st, const double* eprob, const double* pos,
                              const double* fw, uint64_t S, uint64_t I, const double* indF,
                              const double* alpha, const double* ind_lkl, double* marg,
                              int* flags);

// est_maf per site (shared/gen_func.cpp:974-1009): gl_sites [S_own][I_tot][3],
// marg_sites [S_own][I_tot] -> freq_out[S_own]; passes_out (nullable) counts passes.
// bg_waves = kExactBgWaves: the version capped at that many waves per SIMD, which runs on a second
// stream underneath the objective rounds of exact mode's fused iteration (in kExactBgDepth pieces
// queued at a time); 0: uncapped
constexpr int kExactBgWaves = 3, kExactBgDepth = 3;
void launch_estmaf_exact(hipStream_t st, const GlView& gl_sites, const double* marg_sites,
                         uint64_t S_own, uint64_t I_tot, double* freq_out, uint32_t* passes_out,
                         int bg_waves = 0);

// --freq_est 2 / --e_prob 2 AS INTENDED (kernels_ld.hip; opt-in, parity unpinned: the reference
// aborts).  The chain through the sites: gl = log GL (exact) or linear GL (fast), site-major;
// marg [S][I]; freq_old [S]; freq_new [S] holds site 0's new frequency (freq_est 2) or every
// site's (freq_est 1) on entry and the new frequencies afterwards; hap [S][4] receives the
// pairs' haplotype frequencies (row 0 unused).  FLAG_LD_FREQ: a frequency outside [0, 1].
// false: more than 8192 individuals.
bool launch_freq_ld_chain(hipStream_t st, bool exact, const GlView& gl, const double* marg,
                          const double* freq_old, double* freq_new, double* hap, uint64_t S,
                          uint64_t I, int freq_est, int* flags);
// e_prob[s][i][k] = calc_emissionLD (shared/HMM.cpp:175-236) for the sites s >= 1
void launch_emission_ld_exact(hipStream_t st, const GlView& gl, const double* freq,
                              const double* hap, double* eprob, uint64_t S, uint64_t I,
                              int* flags);

// Viterbi (shared/HMM.cpp:98-125): bp [viterbi_blocked_bytes + I] scratch bytes, path_sites
// [viterbi_blocked_bytes] bytes blocked [site/16][I][16] (launch_unblock_path gives [I][S]),
// scratch [chunk_sites*I*4 + I*2] doubles (chunk_sites from viterbi_chunk_sites)
// the two halves of launch_viterbi_exact, for a handle whose sites continue another's (a site
// shard): chain_start false = scratch's state doubles hold the scores the range before ended
// with; the I last-state bytes behind bp may be overwritten between the halves; state_before
// receives the state at the site in front of the handle's first
void launch_viterbi_fwd_exact(hipStream_t st, const double* eprob, const double* pos, uint64_t S,
                             uint64_t I, const double* indF, const double* alpha, uint8_t* bp,
                             double* scratch, uint64_t chunk_sites, bool chain_start,
                             bool serial = false);  // serial: one lane per individual, no loader waves
void launch_viterbi_back_exact(hipStream_t st, uint8_t* bp, uint64_t S, uint64_t I,
                               uint8_t* path_sites, uint8_t* state_before);
void launch_viterbi_exact(hipStream_t st, const double* eprob, const double* pos, uint64_t S,
                          uint64_t I, const double* indF, const double* alpha, uint8_t* bp,
                          uint8_t* path_sites, double* scratch, uint64_t chunk_sites, bool serial = false);
uint64_t viterbi_chunk_sites(uint64_t S, uint64_t I);
uint64_t viterbi_blocked_bytes(uint64_t S, uint64_t I);
void launch_unblock_path(hipStream_t st, const uint8_t* path16, uint64_t S, uint64_t I,
                         uint8_t* out);

// ---------------- layout helpers (kernels_util.hip) ----------------
// out[c][r] = in[r][c]
void launch_transpose_f64(hipStream_t st, const double* in, double* out, uint64_t rows, uint64_t cols);
void launch_transpose_u8(hipStream_t st, const uint8_t* in, uint8_t* out, uint64_t rows, uint64_t cols);
// eprob [S][I][2] -> out [I][S][2]
void launch_transpose_pairs_f64(hipStream_t st, const double* in, double* out, uint64_t rows,
                                uint64_t cols);
// "%f" text of rows x cols values in [0, 1]: 9 bytes per value (8 characters + tab, newline
// after the last value of a row); *bad = 1 if a value is outside [0, 1]
void launch_format_fixed6(hipStream_t st, const double* in, uint64_t rows, uint64_t cols, char* out,
                          int* bad);
// out[(s - lo) * I + i] = marg[s][i] for s in [lo, hi): a contiguous slice copy in site-major
void launch_copy_f64(hipStream_t st, const double* in, double* out, uint64_t n);

}  // namespace nghmm

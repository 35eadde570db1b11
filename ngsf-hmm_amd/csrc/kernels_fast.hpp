// kernels_fast.hpp -- fast-mode (linear-space, chunk-parallel) kernels: host interface.
// See kernels_fast.hip for the design.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace nghmm {

struct FastState {
  uint64_t I = 0, S = 0;
  uint64_t T = 0;        // steps per lane: each lane of an individual's 64*C lanes walks T sites
  uint32_t C = 0;        // waves (chunks) per individual
  uint64_t Spad = 0;     // 64 * C * T >= S
  const double* d_gl = nullptr;  // borrowed: site-major log GL [S][I][3]
  double* e_il = nullptr;        // linear emissions, interleaved [I][C][T][64] x 2 (double2)
  double* pos_il = nullptr;      // distances, interleaved [C][T][64]
  double* r_il = nullptr;        // forward odds / posteriors, interleaved [I][C][T][64]
  double* part = nullptr;        // per-lane chunk operators [.][64*C][5] (4 mantissas + exponent)
  double* bound = nullptr;       // per-lane incoming forward/backward vectors
  double* eprob_log = nullptr;   // lazily: log emissions site-major for Viterbi [S][I][2]
  uint32_t* grp = nullptr;       // device scratch for point groups
  size_t grp_cap = 0;
  double* pt_buf = nullptr;
  size_t pt_cap = 0;
};

bool fast_create(FastState& fs, uint64_t I, uint64_t S);
void fast_destroy(FastState& fs);
// (re)build the interleaved distance table; remembers the GL pointer
bool fast_load(FastState& fs, hipStream_t st, const double* d_gl, const double* d_pos);
// emissions of every cell from freq (linear space), into the interleaved layout
bool fast_refresh_site_tables(FastState& fs, hipStream_t st, const double* d_freq, int* d_flags);
bool fast_lkl_batch(FastState& fs, hipStream_t st, uint32_t n_pts, const uint32_t* d_ind,
                    const double* d_F, const double* d_A, double* d_lkl, int* d_flags);
// forward + backward + posteriors; marg out is site-major [S][I]
bool fast_estep(FastState& fs, hipStream_t st, const double* d_indF, const double* d_alpha,
                double* d_ind_lkl, double* d_marg, int* d_flags);
// est_maf on [S_own] sites: GL site-major log [S_own][I_tot][3], posteriors in
// rank blocks [I_tot / I_blk][S_own][I_blk]
bool fast_estmaf(FastState& fs, hipStream_t st, const double* d_gl_sites,
                 const double* d_marg_blocks, uint64_t S_own, uint64_t I_tot, uint64_t I_blk,
                 double* d_freq_out);
bool fast_viterbi(FastState& fs, hipStream_t st, const double* d_indF, const double* d_alpha,
                  uint8_t* d_bp, uint8_t* d_path_sites);
// log emissions [I][S][2] for host read-back
bool fast_export_emissions(FastState& fs, hipStream_t st, double* d_out);

}  // namespace nghmm

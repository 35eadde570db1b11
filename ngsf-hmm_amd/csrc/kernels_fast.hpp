// kernels_fast.hpp -- fast-mode (linear-space, chunk-parallel) kernels: host interface.
// See kernels_fast.hip for the design.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <vector>

#include "glview.hpp"

namespace nghmm {

// Measurement / debugging switches of a handle (include/nghmm.h lists them).  None changes a
// result beyond rounding.  Read from the environment ONCE, when the handle is created
// (NGHMM_<NAME in capitals>); afterwards only nghmm_set_switch changes them.
struct Switches {
  int bg_parts = 2;           // fused iteration, host-planned rounds: est_maf behind the rounds in this many parts (0: sweep and est_maf after the rounds)
  int estmaf_interp = 1;      // 0: every est_maf pass evaluated over all individuals
  int estmaf_no_rows = 0;     // small cohorts: a wave per site instead of four sites per wave
  int estmaf_no_called = 0;   // called genotypes through the general est_maf kernels too
  int fast_c = 0;             // waves per individual (0: by cohort size); at creation only
  int exact_serial = 0;       // exact mode: one lane per chain instead of producer-consumer workgroups (same bits)
  int dbg_abort_round = 0;    // test hook: a device-planned M-step returns an error after this round
  int no_dev_bfgs = 0;        // the M-step's L-BFGS-B machines on the host (rounds 1-4), not on the device
  int no_bg_stream = 0;       // device-planned rounds: backward sweep and est_maf between the rounds on the one stream, not next to them on a second
  int spans = 0;              // fast mode: timing events around the kernel families of a fused iteration, for nghmm_kernel_ms (0.05 ms per iteration: 8 % of configs[1]'s)
  static Switches from_env();
  // false: no switch of that name
  bool set(const char* name, long value);
};

// A handle that holds a contiguous SITE range (all individuals) of a data set whose sites are
// split over `world` handles, one per GPU (kernels_fast.hip, "site shards"; nghmm_site_shard_setup)
struct SiteShard {
  uint32_t rank = 0, world = 1;
  double* send = nullptr;   // the caller's device buffers: cap doubles, world * cap doubles
  double* recv = nullptr;
  uint64_t cap = 0;
  // all-gather of the first n_bytes of `send` into recv = [rank][n_bytes], ordered on the
  // handle's stream (it may block, it need not); non-zero = failure
  int (*allgather)(void* user, uint64_t n_bytes) = nullptr;
  void* user = nullptr;
  double* edges = nullptr;  // [I][8], owned: what enters the range from both sides (k_fast_shard_edges)
  bool edges_from_round = false;  // the last emitting objective round has left `edges` current
  uint64_t n_gathers = 0;   // accounting
};

struct FastState {
  Switches sw;
  SiteShard shard;
  uint64_t I = 0, S = 0;
  uint64_t T = 0;     // sites walked by one lane
  uint32_t C = 0;     // waves (chunks of 64 lanes) per individual
  uint64_t J = 0;     // 64 * C lane-chunks per individual
  uint64_t Spad = 0;  // J * T >= S
  GlView gl_log;                  // borrowed: site-major log GL [S][I] cells (dense or packed)
  const double* d_pos = nullptr;  // borrowed: [S]
  // packed handle (called genotypes as 2-bit codes, glview.hpp): no dense copy of the
  // likelihoods exists; the fresh forward walk reads geno_il, est_maf the site-major codes
  bool packed = false;
  bool owns_data = true;          // false: a replica, whose data arrays belong to its parent
  uint32_t* geno_il = nullptr;    // codes interleaved [I][C][T/16][64]: 16 sites of a lane per word
  double* cls_lin = nullptr;      // [4][3] linear likelihoods of the four classes (device)
  double u_lin = 0;               // linear likelihood of a uniform (missing) cell, host copy
  bool called_table = false;      // packed: the class table is (1,0,0), (0,1,0), (0,0,1), (u,u,u)
  double* gl_lin = nullptr;       // exp(GL), site-major [S][I][3] (emission refresh, est_maf)
  double* e_il = nullptr;         // emission ratios e1/e0, interleaved [I][C][T][64]
  double* base_c = nullptr;       // [I][C]: sum of log e0 over the wave's sites
  double* pos_il = nullptr;       // distances, interleaved [C][T][64]
  double* chunk_scale = nullptr;  // [C][64] x (sum of the finite distances, number of chromosome
                                  // starts) over a lane-chunk's T sites (fast_dev.hpp: op_step_k)
  double* glq_il = nullptr;       // linear GL relative to the cell's largest, interleaved like
                                  // e_il, 16 B per cell (kernels_fast.hip: glq_encode)
  double* gl_scale_c = nullptr;   // [I][C]: sum of the cells' largest log GL over the wave's sites
  double* freq_il = nullptr;      // allele frequencies, interleaved [C][T][64]
  bool e_stale = true;            // e_il older than freq_il (refreshed lazily)
  double* post = nullptr;         // posteriors, tile-major [c*T + t][i / 8][l][i % 8] (site
                                  // (c*64+l)*T + t; kernels_fast.hip: kPost8)
  double* ckpt = nullptr;         // forward checkpoints [I][C][T/8][2][64] x double2
  double* lane_ops = nullptr;     // per-lane chunk operators [I][J][5]
  double* bound = nullptr;        // per-lane incoming forward/backward vectors [I][J][4]
  // objective batches: two independent sets of descriptor / partial-operator buffers, so that
  // one batch can be prepared and evaluated while the host digests the other's results
  // (fast_lkl_prepare / _launch / _covers_everyone work on lanes[cur_lane])
  struct ModeRange { uint32_t mode, begin, count; };
  struct LklLane {
    double* part = nullptr;         // [groups][C][5 points][5]
    size_t part_cap = 0;
    void* grp_dev = nullptr;        // packed group descriptors
    size_t grp_cap = 0;
    std::vector<unsigned char> grp_host;
    uint32_t n_groups = 0;
    uint32_t n_pts = 0;
    std::vector<ModeRange> mode_ranges;  // groups sorted by loop-body version
  };
  LklLane lanes[2];
  int cur_lane = 0;

  // indF / alpha M-step advanced on the device (kernels_bfgs.hip): the per-individual problems
  // and solvers, the group descriptors of the round being planned (by individual), one worklist
  // per loop-body version + one of everybody, the objective values at [individual * 5 + slot].
  // The kernel that plans a round publishes (round, number of active individuals, modes and
  // their counts) in pinned host memory, which the host polls -- no copy, no event.
  struct DevBfgs {
    static constexpr uint32_t kRing = 4;           // control slots, by plan number % kRing
    static constexpr uint32_t kTableWords = 4 + 2 * 129 + 12;
    uint64_t cap_I = 0;
    void* prob = nullptr;        // BfgsProblem [I]
    void* solver = nullptr;      // LbfgsbT<PtrStore> [I]: the saved scalars
    double* arrays = nullptr;    // [I][LbfgsbPtrs::doubles(2, 10)]: the solvers' work arrays
    void* groups = nullptr;      // GroupDesc [I], by individual
    uint32_t* last_mode = nullptr;   // [I]: the mode an individual's points were last evaluated by
    uint32_t* worklists = nullptr;   // [2][kModeSlots][I]: two sets, by plan parity
    uint32_t* all = nullptr;         // [2][I]
    uint32_t* counts = nullptr;      // [kRing][kModeSlots + 3]: per-mode counts, everybody, ticket
    double* lkl = nullptr;           // [5 I]
    double* part = nullptr;          // [I][C][MAXP][5]
    double *d_F = nullptr, *d_A = nullptr;  // the handle's parameter arrays (borrowed, per M-step)
    // [2][I]: the parameters the M-step started from, which the E-step next to it reads -- two
    // sets, by M-step parity: the NEXT M-step's first plan is made (and its set written) while
    // this one's backward sweep may still be reading (dbfgs_preplan)
    double *snap_F = nullptr, *snap_A = nullptr;
    uint32_t snap_set = 0;           // the set of the M-step that dbfgs_begin started last
    // pinned host memory the kernels write: the plans (number, active individuals, modes and
    // counts, statistics), a finished individual's parameters
    volatile uint32_t* h_table = nullptr;  // [kRing][kTableWords]
    double *h_F = nullptr, *h_A = nullptr;  // [I]
    unsigned long long stats_host[6] = {0, 0, 0, 0, 0, 0};  // of the last plan waited for
    uint32_t seq_base = 0;           // plans of earlier M-steps (plans are numbered through the handle's life)
    uint32_t mstep_no = 0;
    bool clean = true;               // the last M-step reached its end (else: dbfgs_begin starts over)
    // The NEXT M-step's first plan, made when this one ended (the handle's parameters are final
    // then, and nothing else enters it): its planning kernel and the plan's way to the host are
    // off the next iteration's critical path.  Void as soon as anything writes the parameters or
    // reloads the data (dbfgs_invalidate).
    bool preplanned = false;
    bool pre_F_fixed = false, pre_alpha_fixed = false;
    // End of a fused iteration without a copy or an event: a one-workgroup kernel behind the last
    // piece of work writes the error flags and the log-likelihoods to pinned memory and, behind a
    // system-scope fence, a sequence number the host polls (dbfgs_epilogue / dbfgs_wait_epilogue).
    double* h_epi_lkl = nullptr;     // [I] pinned
    volatile uint32_t* h_epi = nullptr;  // [kEpiFlags] flags, then the sequence word
    uint32_t epi_seq = 0;
  } dev;

  double dmax_finite = 0;         // largest finite distance of the loaded data
  double alpha_small_min = 0;     // 1e-6 / mean finite distance: below it no small-alpha (kappa
                                  // form) objective kernel (fast_dev.hpp: fd_pattern)
  uint8_t* redo = nullptr;        // per-site "needs the careful est_maf route" flags
  size_t redo_cap = 0;
  uint8_t* est_status = nullptr;  // est_maf per-site state machine (see k_fast_estmaf):
  double* est_state = nullptr;    // [EST_FIELDS][redo_cap] loop state + interval node values
  // diagnostics (include/nghmm_debug.h): individual-rounds by loop-body version of the objective
  // kernels since the last reset; est_maf's rare routes, counted on the device by the sites that
  // take them (EstCount below)
  std::map<uint32_t, uint64_t> mode_ind_rounds;
  uint32_t* est_counts = nullptr;  // [EST_COUNTS]
};

// est_maf's routes off the common one (exact passes, one checked interpolant, the rest on it)
enum EstCount {
  EST_CNT_CHECK_FAILED = 0,  // the interpolant missed the next exact pass by > 1e-11: exact passes to the end
  EST_CNT_RESUMED = 1,       // left its interval or a stopping decision too close to call: a second interval
  EST_CNT_RESUMED_AGAIN = 2, // ... and once more: a third
  EST_CNT_LOGSPACE = 3,      // a cell whose linear weights all vanish: the reference-order log-space route
  EST_CNT_EXACT_TAIL = 4,    // left its third interval as well: exact passes to the end
  EST_COUNTS = 5
};

bool fast_create(FastState& fs, uint64_t I, uint64_t S, bool packed = false);
// a replica of `parent` (after its fast_load): shares the read-only data arrays (likelihoods in
// every layout, distances), owns everything an EM run writes
bool fast_create_replica(FastState& fs, const FastState& parent);
void fast_destroy(FastState& fs);
// (re)build the interleaved distance table; remembers the GL / distance pointers
bool fast_load(FastState& fs, hipStream_t st, const GlView& gl_log, const double* d_pos);
// after a frequency update: interleaved frequency table + MAF check; marks e_il stale
bool fast_refresh_freq_table(FastState& fs, hipStream_t st, const double* d_freq, int* d_flags);
// linear-space emissions of every cell from freq, into the interleaved layout
bool fast_refresh_emissions(FastState& fs, hipStream_t st, const double* d_freq, int* d_flags);
// objective for host-side points: prepare() groups them by individual and uploads
// the descriptors, launch() runs the two kernels; d_lkl (device) receives the values
// in point order
// force_general: every group through the general kernel (one exponent per point)
bool fast_lkl_prepare(FastState& fs, hipStream_t st, uint32_t n_pts, const uint32_t* h_ind,
                      const double* h_F, const double* h_A, bool force_general = false);
// emit_estep: the launch is the first round of an M-step (point 0 of every individual =
// its current parameters) and leaves the E-step's lane operators and checkpoints behind
bool fast_lkl_launch(FastState& fs, hipStream_t st, double* d_lkl, int* d_flags,
                     bool emit_estep = false);
bool fast_lkl_covers_everyone(const FastState& fs);
// ---- the M-step's rounds planned on the device (kernels_bfgs.hip) ----
// whether this handle's data admit them (every finite-difference probe inside the pattern
// kernels' range: else all groups would take the general kernel one point at a time)
bool dbfgs_available(const FastState& fs);
bool dbfgs_reserve(FastState& fs);
void dbfgs_destroy(FastState& fs);
// the parameters (or the data) are about to change under a plan made in advance: drop it
void dbfgs_invalidate(FastState& fs);
// iteration's end: flags [n_flags ints, device; zeroed again for the next iteration] and
// log-likelihoods [I, device; may be null] to pinned memory, then the sequence word; nothing waits
constexpr uint32_t kEpiFlags = 8;
bool dbfgs_epilogue(FastState& fs, hipStream_t st, int* d_flags, uint32_t n_flags, const double* d_lkl);
// wait (polling) for that word; flags_out [n_flags]; false: the stream ran dry without it
bool dbfgs_wait_epilogue(FastState& fs, hipStream_t st, int* flags_out, uint32_t n_flags, bool yield);
// plan round 1 from the current parameters; nothing waits
// (d_indF / d_alpha: a finished individual's new parameters are written there, and to pinned
// host memory)
bool dbfgs_begin(FastState& fs, hipStream_t st, double* d_indF, double* d_alpha, bool F_fixed,
                 bool alpha_fixed);
// the parameters that M-step started from (device, [I] each)
const double* dbfgs_start_F(const FastState& fs);
const double* dbfgs_start_A(const FastState& fs);
// the values of round `round` into the machines, round + 1 planned; nothing waits
// (n_in = the active individuals of that round, as dbfgs_wait_plan reported them)
bool dbfgs_advance(FastState& fs, hipStream_t st, uint32_t round, uint32_t n_in);
// wait (polling pinned memory) for the plan of round `round`: active individuals and the modes
// present; false: the stream ran dry without publishing it
// (yield: give the core up between looks -- handles that run next to others from host threads)
bool dbfgs_wait_plan(FastState& fs, hipStream_t st, uint32_t round, uint32_t* n_active,
                     std::vector<FastState::ModeRange>* ranges, bool yield = false);
// launch the planned round
bool dbfgs_launch_round(FastState& fs, hipStream_t st, uint32_t round, uint32_t n_active,
                        const std::vector<FastState::ModeRange>& ranges, bool emit_estep);
// after the empty plan of round last_round has arrived: fs.dev.h_F / h_A hold every individual's
// parameters, fs.dev.stats_host the statistics (the next M-step's plans are numbered on from here)
void dbfgs_end(FastState& fs, uint32_t last_round);
bool dbfgs_preplan(FastState& fs, hipStream_t st, bool F_fixed, bool alpha_fixed);
bool fast_lkl_launch_planned(FastState& fs, hipStream_t st, const void* d_groups_by_ind,
                             const std::vector<FastState::ModeRange>& ranges, uint32_t n_active,
                             const uint32_t* d_worklists, const uint32_t* d_all, double* part,
                             double* d_lkl, bool emit_estep);
// forward checkpoints (unless a preceding fast_lkl_launch(emit_estep) left them), boundary
// vectors, backward sweep with posteriors into fs.post (tile-major)
bool fast_estep(FastState& fs, hipStream_t st, const double* d_indF, const double* d_alpha,
                double* d_ind_lkl, int* d_flags, bool have_forward_walk);
// fs.post -> site-major [S][I]
bool fast_post_to_site_major(FastState& fs, hipStream_t st, double* d_marg);
// out = exp(in) elementwise (in == out allowed): linear genotype likelihoods
void fast_exp(hipStream_t st, const double* d_in, double* d_out, uint64_t n);
// est_maf on S_own sites: LINEAR GL site-major [S_own][I_tot] cells (fast_gl_lin(), or a
// site shard passed through fast_exp; packed: codes + fs.cls_lin), posteriors in rank blocks
// [I_tot / I_blk][S_own][I_blk] or (tile_major) fs.post itself
GlView fast_gl_lin(const FastState& fs);
// part / n_parts: only the sites of tile rows [R part / n_parts, R (part + 1) / n_parts) of
// the E-step's layout (R = C T) -- the caller spreads the parts between other work on the
// stream; needs fast_estmaf_splittable().  Every site's result is the one the whole call gives.
bool fast_estmaf_splittable(const FastState& fs, uint64_t I_tot, bool tile_major);
// est_maf reads the E-step's tile-major posteriors in place (else: a site-major copy first)
bool fast_estmaf_in_place(const FastState& fs, uint64_t I_tot);
bool fast_estmaf_reserve(FastState& fs, uint64_t S_own);
// called genotypes: the per-pass sums in closed form (k_fast_estmaf_called_sums)
bool fast_estmaf_called(const FastState& fs, const GlView& gl);
bool fast_estmaf(FastState& fs, hipStream_t st, const GlView& d_gl_lin_sites,
                 const double* d_marg_blocks, uint64_t S_own, uint64_t I_tot, uint64_t I_blk,
                 double* d_freq_out, bool tile_major = false, uint32_t part = 0,
                 uint32_t n_parts = 1);
bool fast_viterbi(FastState& fs, hipStream_t st, const double* d_freq, const double* d_indF,
                  const double* d_alpha, uint8_t* d_bp, uint8_t* d_path_sites, int* d_flags,
                  double* d_scratch, uint64_t chunk_sites);
// the forward half of fast_viterbi (site shards: launch_viterbi_fwd_exact)
bool fast_viterbi_forward(FastState& fs, hipStream_t st, const double* d_freq, const double* d_indF,
                          const double* d_alpha, uint8_t* d_bp, int* d_flags, double* d_scratch,
                          uint64_t chunk_sites, bool chain_start);
// log of the stored linear emissions as [I][S][2] (device), for host read-back
bool fast_export_emissions(FastState& fs, hipStream_t st, double* d_out);

}  // namespace nghmm

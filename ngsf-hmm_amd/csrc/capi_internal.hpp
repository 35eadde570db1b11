// capi_internal.hpp -- what the translation units of the C ABI share: the handle, error plumbing,
// and the helpers every entry point uses.  nghmm_capi.hip (handle life cycle, single-handle EM),
// capi_load.hip (loaders), capi_output.hip (read-back and output formatting) and capi_multi.hip
// (individual shards, site shards, groups and chains of handles) implement include/nghmm.h.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <new>
#include <functional>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nghmm.h"
#include "../../include/nghmm_debug.h"
#include "bfgs_batch.hpp"
#include "kernels.hpp"
#include "kernels_fast.hpp"

using namespace nghmm;

namespace capi {

// the message of the last failing call on this thread (include/nghmm.h, nghmm_last_error)
extern thread_local std::string g_last_error;
void set_error(const char* fmt, ...);

#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t e__ = (expr);                                                       \
    if (e__ != hipSuccess) {                                                       \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__,  \
                __LINE__);                                                         \
      return NGHMM_ERR_HIP;                                                        \
    }                                                                              \
  } while (0)

enum Slot { SLOT_EMISSION = 0, SLOT_FORWARD = 1, SLOT_BACKWARD = 2, SLOT_LKL = 3, SLOT_ESTMAF = 4,
            SLOT_VITERBI = 5, SLOT_LKL_FIRST = 6, SLOT_BFGS = 7, NSLOTS = 8 };

}  // namespace capi

using namespace capi;

struct ChainCtx;   // nghmm_chain_setup
struct nghmm_handle {
  uint64_t I = 0, S = 0;
  int device = 0, mode = NGHMM_MODE_EXACT;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_sync = nullptr;
  // pinned landing places of what every call reads back -- the error flags, and a fused iteration's
  // log-likelihoods: a copy to pageable memory is staged and waited for by the runtime, one each
  int* h_flags_pin = nullptr;      // [NFLAGS]
  double* h_lkl_pin = nullptr;     // [I]
  // exact mode, fused iteration: est_maf on a second stream underneath the objective rounds
  hipStream_t aux_stream = nullptr;
  hipEvent_t aux_ev0 = nullptr, aux_ev1 = nullptr, aux_go = nullptr, aux_done = nullptr;
  static constexpr uint32_t kAuxPieces = 16;   // exact mode: est_maf underneath the rounds, in pieces
  hipEvent_t aux_piece_ev[kAuxPieces] = {};
  hipEvent_t aux_estep_ev[3] = {};             // ... the E-step next to the first rounds: its timing
  double* d_aux_params = nullptr;              // ... and its own copies of indF / alpha [2][I]
  hipEvent_t param_snapshot_ev = nullptr;      // (borrowed) those copies are made: an M-step's new parameters wait for it
  bool blocking_sync = false;
  bool loaded = false;
  bool warmed = false;   // nghmm_emission has set up what the first EM iteration needs

  double *d_gl = nullptr, *d_pos = nullptr, *d_freq = nullptr, *d_eprob = nullptr, *d_fw = nullptr,
         *d_marg = nullptr, *d_indF = nullptr, *d_alpha = nullptr, *d_ind_lkl = nullptr;
  int* d_flags = nullptr;

  uint32_t* d_pt_ind = nullptr;
  double *d_pt_F = nullptr, *d_pt_A = nullptr, *d_pt_lkl = nullptr;
  size_t pt_cap = 0;

  uint8_t *d_bp = nullptr, *d_path_sites = nullptr, *d_path = nullptr;
  double* d_vit = nullptr;  // Viterbi scratch: transition logs of one site chunk + carry state
  double* d_tmp = nullptr;  // S*I*2 doubles, transposes for host read-back
  double* d_geno = nullptr;  // .geno posteriors of one site chunk
  size_t geno_cap = 0;
  char* d_text = nullptr;    // formatted posterior lines of one batch of individuals
  size_t text_cap = 0;
  bool tmp_is_posteriors = false;  // d_tmp holds the [I][S] posteriors of the last E-step
  uint32_t* d_passes = nullptr;
  double *d_freq_new = nullptr, *d_hap = nullptr;  // --freq_est 2 as intended: [S], [S][4]

  // multi-GPU shard
  uint64_t I_tot = 0, ind_begin = 0, site_begin = 0, S_own = 0;
  double* d_gl_shard = nullptr;

  // packed handle (NGHMM_GENO_PACKED): called genotypes as 2-bit codes (glview.hpp); d_gl
  // does not exist
  bool packed = false;
  uint32_t* d_codes = nullptr;        // [S][I] cells, 16 per word
  uint32_t* d_codes_shard = nullptr;  // [S_own][I_tot] cells of the frequency step's site range
  double* d_cls_log = nullptr;        // [4][3] prepared log likelihoods of the four classes
  double h_cls_proto[12] = {0};       // ... as nghmm_create prepared them (row 3: the reader's
                                      // missing genotype); a load starts from these
  unsigned long long* d_uniform = nullptr;  // the one value every uniform cell carries (~0: none yet)
  // chunked loading (nghmm_load_begin .. nghmm_load_end)
  uint64_t lkl_redone = 0;            // objective points re-evaluated by the general kernel
  // fast-mode M-step after its first round: the individuals in two halves, each with its own
  // buffers and events, so that the host advances one half's optimizers while the GPU
  // evaluates the other half's points (mstep_indf_impl)
  struct LklAsync {
    double* d_lkl = nullptr;   // device results
    size_t cap = 0;
    double* h_lkl = nullptr;   // pinned host results
    size_t h_cap = 0;
    int* d_flags = nullptr;
    int* h_flags = nullptr;    // pinned
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr;
    std::vector<uint32_t> ind;
    std::vector<double> F, A;
    uint64_t lo = 0, hi = 0;
    bool pending = false;
  } lane[2];
  // Background work of a fused EM iteration (mstep_indf_impl): the E-step's backward sweep and
  // the allele-frequency step do not depend on the objective rounds after the first, so they
  // go onto the stream in pieces right behind each round's kernels and run while the host
  // digests that round's values.  Error flags of their own (the rounds clear theirs), a pool
  // of timing events (one pair per piece, read when the iteration ends).
  struct BgSpan {
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int slot = 0;
  };
  int* d_flags_bg = nullptr;
  bool flags_bg_clear = false;   // the last iteration's epilogue kernel left d_flags_bg zeroed
  std::vector<BgSpan> bg_spans;
  size_t bg_used = 0;
  // replicas (nghmm_create_replica): share the parent's data arrays (d_gl / d_codes /
  // d_cls_log / d_pos and the fast-mode layouts of the likelihoods)
  nghmm_handle* parent = nullptr;
  std::atomic<int> n_replicas{0};
  // member of a group (nghmm_group_setup): exchange buffers on this handle's device and a
  // second stream for the peer copies, which run under the remaining objective rounds
  int g_n = 0, g_rank = 0;
  double *g_send = nullptr, *g_recv = nullptr, *g_freq_own = nullptr, *g_freq_all = nullptr;
  hipStream_t g_xstream = nullptr;
  // member of an in-process chain of site shards (nghmm_chain_setup): the exchange buffers of
  // fast.shard on this handle's device and the chain's shared state
  struct ChainCtx* chain = nullptr;
  double *c_send = nullptr, *c_recv = nullptr;
  bool loading = false;
  // sites that have arrived since nghmm_load_begin, as disjoint [begin, end) runs: every site
  // must arrive exactly once (a repeated site would OR two codes into a packed cell)
  std::map<uint64_t, uint64_t> load_cover;
  double* d_stage = nullptr;          // staging buffer of one chunk of raw likelihoods
  size_t stage_cap = 0;
  int8_t* d_stage8 = nullptr;         // ... of one chunk of reader genotypes
  size_t stage8_cap = 0;

  FastState fast;  // fast-mode layouts (kernels_fast.hip)
  BfgsBatch batch;  // one L-BFGS-B state machine per individual, storage reused across M-steps
  // fast mode keeps the posteriors tile-major (fast.post); the site-major copy d_marg is
  // made on demand (host read-back, multi-GPU packing, est_maf beyond 4096 individuals)
  bool marg_valid = false;

  std::vector<double> h_indF, h_alpha;
  double ms[NSLOTS] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t launches[NSLOTS] = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace capi {

hipError_t sync_stream(nghmm_t* h);
int use_device(nghmm_t* h);
void tic(nghmm_t* h);
int toc(nghmm_t* h, int slot, bool accumulate);
int clear_flags(nghmm_t* h);
int check_flags(nghmm_t* h, const int* d_flags = nullptr);
int ensure_points(nghmm_t* h, size_t n);
GlView own_gl(const nghmm_t* h);
int ensure_tmp(nghmm_t* h);
int ensure_marg(nghmm_t* h);
int fast_estep_impl(nghmm_t* h, double* ind_lkl, bool have_walk);
int redo_nonfinite(nghmm_t* h, uint32_t n_pts, const uint32_t* ind, const double* F,
                   const double* alpha, double* lkl);
int lkl_batch_impl(nghmm_t* h, uint32_t n_pts, const uint32_t* ind, const double* F,
                   const double* alpha, double* lkl, bool accumulate, bool* emit_estep = nullptr);
int lane_setup(nghmm_t* h, int k, size_t n);
int lkl_submit(nghmm_t* h, int k);
int lkl_wait(nghmm_t* h, int k);
int emission_impl(nghmm_t* h);
int bg_begin(nghmm_t* h);
int bg_open(nghmm_t* h, int slot, hipStream_t st = nullptr);   // (st: where the piece runs; default the handle's stream)
int bg_close(nghmm_t* h, hipStream_t st = nullptr);
int bg_finish(nghmm_t* h);
int ensure_emissions(nghmm_t* h);

// est_maf on the handle's own sites and individuals or on its frequency-step site range
int estmaf_and_refresh(nghmm_t* h, bool shard, const double* d_marg_blocks, uint64_t S_own,
                       uint64_t I_tot, uint64_t I_blk, double* d_freq_out);
// leaves its chain, which dissolves (capi_multi.hip)
void chain_release(nghmm_t* h);

template <typename T>
int dev_alloc(T** p, size_t n) {
  if (n == 0) n = 1;
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) {
    set_error("hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
    return NGHMM_ERR_NOMEM;
  }
  return NGHMM_OK;
}

}  // namespace capi

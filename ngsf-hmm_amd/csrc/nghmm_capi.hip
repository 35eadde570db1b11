// nghmm_capi.hip -- implementation of include/nghmm.h on one MI355X.
//
// Device-resident state, site-major:
//   gl     [S][I][3]  log GL, as uploaded            24 B / site-individual
//   eprob  [S][I][2]  log emissions                  16 B
//   fw     [S+1][I][2] forward variable (E-step)     16 B
//   marg   [S][I]     posterior of the IBD state      8 B
//   pos[S], freq[S], indF[I], alpha[I], ind_lkl[I]
// Nothing of size S*I crosses PCIe inside an EM iteration: per round of the
// indF/alpha M-step only the probe points (20 B each) go down and their
// log-likelihoods (8 B each) come back.
#include "capi_internal.hpp"

namespace capi {

thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
}


// Wait for the handle's stream.  Handles that run next to others from their own host threads
// (replicas) wait on a blocking event instead of spinning, so that more waiting threads than
// cores do not starve the threads that have work.
hipError_t sync_stream(nghmm_t* h) {
  if (!h->blocking_sync) return hipStreamSynchronize(h->stream);
  hipError_t e = hipEventRecord(h->ev_sync, h->stream);
  return e != hipSuccess ? e : hipEventSynchronize(h->ev_sync);
}

int use_device(nghmm_t* h) {
  HIP_TRY(hipSetDevice(h->device));
  return NGHMM_OK;
}


void tic(nghmm_t* h) { (void)hipEventRecord(h->ev0, h->stream); }

// stops the timer, synchronises the stream and stores/accumulates the time
int toc(nghmm_t* h, int slot, bool accumulate) {
  HIP_TRY(hipEventRecord(h->ev1, h->stream));
  HIP_TRY(hipEventSynchronize(h->ev1));
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  if (accumulate) {
    h->ms[slot] += ms;
    h->launches[slot] += 1;
  } else {
    h->ms[slot] = ms;
    h->launches[slot] = 1;
  }
  return NGHMM_OK;
}

int clear_flags(nghmm_t* h) {
  HIP_TRY(hipMemsetAsync(h->d_flags, 0, NFLAGS * sizeof(int), h->stream));
  return NGHMM_OK;
}

// the kernels' error flags as the reference's fatal errors
static int map_flags(const int* f);

// Reads the kernel error flags and maps them to the reference's fatal errors.
int check_flags(nghmm_t* h, const int* d_flags) {
  if (!h->h_flags_pin)
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_flags_pin), NFLAGS * sizeof(int), hipHostMallocDefault));
  int* f = h->h_flags_pin;
  HIP_TRY(hipMemcpyAsync(f, d_flags ? d_flags : h->d_flags, NFLAGS * sizeof(int), hipMemcpyDeviceToHost,
                         h->stream));
  HIP_TRY(sync_stream(h));
  return map_flags(f);
}

static int map_flags(const int* f) {
  if (f[FLAG_INVALID_LKL]) {
    set_error("invalid Lkl found!");
    return NGHMM_ERR_INVALID_LKL;
  }
  if (f[FLAG_FW_BW]) {
    set_error("Fw and Bw lkl do not match!");
    return NGHMM_ERR_FW_BW;
  }
  if (f[FLAG_INVALID_MAF]) {
    set_error("invalid MAF!");
    return NGHMM_ERR_INVALID_MAF;
  }
  if (f[FLAG_NAN]) {
    set_error("value is NaN!");
    return NGHMM_ERR_NAN;
  }
  return NGHMM_OK;
}

int ensure_points(nghmm_t* h, size_t n) {
  if (n <= h->pt_cap) return NGHMM_OK;
  size_t cap = n + n / 4 + 1024;
  if (h->d_pt_ind) (void)hipFree(h->d_pt_ind);
  if (h->d_pt_F) (void)hipFree(h->d_pt_F);
  if (h->d_pt_A) (void)hipFree(h->d_pt_A);
  if (h->d_pt_lkl) (void)hipFree(h->d_pt_lkl);
  h->d_pt_ind = nullptr;
  h->d_pt_F = h->d_pt_A = h->d_pt_lkl = nullptr;
  h->pt_cap = 0;
  int rc;
  if ((rc = dev_alloc(&h->d_pt_ind, cap))) return rc;
  if ((rc = dev_alloc(&h->d_pt_F, cap))) return rc;
  if ((rc = dev_alloc(&h->d_pt_A, cap))) return rc;
  if ((rc = dev_alloc(&h->d_pt_lkl, cap))) return rc;
  h->pt_cap = cap;
  return NGHMM_OK;
}

// the handle's own genotype likelihoods (log space), and those of its frequency-step site range
GlView own_gl(const nghmm_t* h) {
  return h->packed ? gl_packed(h->d_codes, h->d_cls_log) : gl_dense(h->d_gl);
}

int ensure_tmp(nghmm_t* h) {
  if (h->d_tmp) return NGHMM_OK;
  return dev_alloc(&h->d_tmp, h->S * h->I * 2);
}

int ensure_emissions(nghmm_t* h);

// site-major posteriors [S][I] in d_marg (fast mode: converted from the tile-major layout)
int ensure_marg(nghmm_t* h) {
  if (h->mode != NGHMM_MODE_FAST || h->marg_valid) return NGHMM_OK;
  int rc;
  if (!h->d_marg) {
    if ((rc = dev_alloc(&h->d_marg, (size_t)h->I * h->S))) return rc;
    HIP_TRY(hipMemsetAsync(h->d_marg, 0, (size_t)h->I * h->S * sizeof(double), h->stream));
  }
  if (!fast_post_to_site_major(h->fast, h->stream, h->d_marg)) return NGHMM_ERR_HIP;
  h->marg_valid = true;
  return NGHMM_OK;
}

// fast-mode E-step; have_walk: an objective round just left the forward walk behind
int fast_estep_impl(nghmm_t* h, double* ind_lkl, bool have_walk) {
  int rc;
  if ((rc = clear_flags(h))) return rc;
  if (!have_walk && (rc = ensure_emissions(h))) return rc;
  tic(h);
  if (!fast_estep(h->fast, h->stream, h->d_indF, h->d_alpha, h->d_ind_lkl, h->d_flags, have_walk))
    return NGHMM_ERR_HIP;
  if ((rc = toc(h, SLOT_FORWARD, false))) return rc;
  h->ms[SLOT_BACKWARD] = 0;
  h->launches[SLOT_BACKWARD] = 0;
  h->marg_valid = false;
  h->tmp_is_posteriors = false;
  HIP_TRY(hipGetLastError());
  if (ind_lkl)
    HIP_TRY(hipMemcpyAsync(ind_lkl, h->d_ind_lkl, h->I * sizeof(double), hipMemcpyDeviceToHost,
                           h->stream));
  return check_flags(h);
}

// Points whose value is not finite: the pattern kernels share one scale among the points of a
// group and can overflow when a probe's likelihood is many orders away from point 0's; the
// general kernel carries an exponent per point.  (By-products of an emitting round come from
// point 0 alone and stay valid.)  Synchronous; lkl[] is patched in place.
int redo_nonfinite(nghmm_t* h, uint32_t n_pts, const uint32_t* ind, const double* F,
                   const double* alpha, double* lkl) {
  std::vector<uint32_t> bad, bind;
  std::vector<double> bF, bA;
  for (uint32_t p = 0; p < n_pts; ++p)
    if (!std::isfinite(lkl[p])) {
      bad.push_back(p);
      bind.push_back(ind[p]);
      bF.push_back(F[p]);
      bA.push_back(alpha[p]);
    }
  if (bad.empty()) {
    set_error("invalid Lkl found!");
    return NGHMM_ERR_INVALID_LKL;
  }
  g_last_error.clear();
  int rc;
  if ((rc = ensure_points(h, bad.size()))) return rc;
  if ((rc = clear_flags(h))) return rc;
  if (!fast_lkl_prepare(h->fast, h->stream, (uint32_t)bad.size(), bind.data(), bF.data(),
                        bA.data(), true) ||
      !fast_lkl_launch(h->fast, h->stream, h->d_pt_lkl, h->d_flags, false)) {
    set_error("objective re-evaluation failed: %s", hipGetErrorString(hipGetLastError()));
    return NGHMM_ERR_HIP;
  }
  std::vector<double> redo(bad.size());
  HIP_TRY(hipMemcpyAsync(redo.data(), h->d_pt_lkl, bad.size() * sizeof(double),
                         hipMemcpyDeviceToHost, h->stream));
  rc = check_flags(h);
  for (size_t k = 0; k < bad.size(); ++k) lkl[bad[k]] = redo[k];
  h->lkl_redone += bad.size();
  return rc;
}

int lkl_batch_impl(nghmm_t* h, uint32_t n_pts, const uint32_t* ind, const double* F,
                   const double* alpha, double* lkl, bool accumulate, bool* emit_estep) {
  if (n_pts == 0) return NGHMM_OK;
  for (uint32_t p = 0; p < n_pts; ++p)
    if (ind[p] >= h->I) {
      set_error("nghmm_lkl_batch: individual index %u out of range", ind[p]);
      return NGHMM_ERR_ARG;
    }
  int rc;
  if ((rc = ensure_points(h, n_pts))) return rc;
  if ((rc = clear_flags(h))) return rc;
  if (h->mode == NGHMM_MODE_FAST) {
    // the fast path groups the points by individual on the host and uploads
    // compact group descriptors itself
    if (!fast_lkl_prepare(h->fast, h->stream, n_pts, ind, F, alpha)) {
      set_error("fast_lkl_prepare failed: %s", hipGetErrorString(hipGetLastError()));
      return NGHMM_ERR_HIP;
    }
    // *emit_estep in: the caller wants the E-step's forward walk as a by-product (first
    // round of an M-step); out: whether every individual was in the batch, i.e. it was left
    const bool emit = emit_estep && *emit_estep && fast_lkl_covers_everyone(h->fast);
    if (emit_estep) *emit_estep = emit;
    if (!emit && (rc = ensure_emissions(h))) return rc;  // else the walk refreshes them itself
    tic(h);  // after the descriptor upload: the timed span is the kernels
    if (!fast_lkl_launch(h->fast, h->stream, h->d_pt_lkl, h->d_flags, emit)) {
      set_error("fast_lkl_launch failed: %s", hipGetErrorString(hipGetLastError()));
      return NGHMM_ERR_HIP;
    }
  } else {
    if (emit_estep) *emit_estep = false;
    HIP_TRY(hipMemcpyAsync(h->d_pt_ind, ind, n_pts * sizeof(uint32_t), hipMemcpyHostToDevice,
                           h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_pt_F, F, n_pts * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_pt_A, alpha, n_pts * sizeof(double), hipMemcpyHostToDevice,
                           h->stream));
    tic(h);
    (h->fast.sw.exact_serial ? launch_forward_exact : launch_forward_exact_pc)(
        h->stream, h->d_eprob, h->d_pos, h->S, h->I, n_pts, h->d_pt_ind, h->d_pt_F, h->d_pt_A,
        h->d_pt_lkl, nullptr, h->d_flags);
  }
  const double ms_before = accumulate ? h->ms[SLOT_LKL] : 0.0;
  if ((rc = toc(h, SLOT_LKL, accumulate))) return rc;
  if (emit_estep && *emit_estep) {  // the round that doubles as the E-step's forward walk
    h->ms[SLOT_LKL_FIRST] = h->ms[SLOT_LKL] - ms_before;
    h->launches[SLOT_LKL_FIRST] = 1;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(lkl, h->d_pt_lkl, n_pts * sizeof(double), hipMemcpyDeviceToHost,
                         h->stream));
  rc = check_flags(h);
  if (rc == NGHMM_ERR_INVALID_LKL && h->mode == NGHMM_MODE_FAST)
    rc = redo_nonfinite(h, n_pts, ind, F, alpha, lkl);
  return rc;
}

// ---- the same evaluation, asynchronous, on one of the two lanes (fast mode) ----
int lane_setup(nghmm_t* h, int k, size_t n) {
  auto& L = h->lane[k];
  int rc;
  if (!L.ev0) {
    const unsigned evf = h->blocking_sync ? hipEventBlockingSync : hipEventDefault;
    HIP_TRY(hipEventCreateWithFlags(&L.ev0, evf));
    HIP_TRY(hipEventCreateWithFlags(&L.ev1, evf));
    HIP_TRY(hipEventCreateWithFlags(&L.ev_done, evf | hipEventDisableTiming));
    if ((rc = dev_alloc(&L.d_flags, (size_t)NFLAGS))) return rc;
    HIP_TRY(hipHostMalloc((void**)&L.h_flags, NFLAGS * sizeof(int), hipHostMallocDefault));
  }
  if (n > L.cap) {
    if (L.d_lkl) (void)hipFree(L.d_lkl);
    L.d_lkl = nullptr;
    L.cap = 0;
    const size_t cap = n + n / 4 + 1024;
    if ((rc = dev_alloc(&L.d_lkl, cap))) return rc;
    L.cap = cap;
  }
  if (n > L.h_cap) {
    if (L.h_lkl) (void)hipHostFree(L.h_lkl);
    L.h_lkl = nullptr;
    L.h_cap = 0;
    const size_t cap = n + n / 4 + 1024;
    HIP_TRY(hipHostMalloc((void**)&L.h_lkl, cap * sizeof(double), hipHostMallocDefault));
    L.h_cap = cap;
  }
  return NGHMM_OK;
}

// enqueue lane k's points (L.ind / L.F / L.A); nothing waits here
int lkl_submit(nghmm_t* h, int k) {
  auto& L = h->lane[k];
  const size_t n = L.ind.size();
  int rc;
  if ((rc = lane_setup(h, k, n))) return rc;
  HIP_TRY(hipMemsetAsync(L.d_flags, 0, NFLAGS * sizeof(int), h->stream));
  h->fast.cur_lane = k;
  const bool ok = fast_lkl_prepare(h->fast, h->stream, (uint32_t)n, L.ind.data(), L.F.data(),
                                   L.A.data());
  if (ok) {
    (void)hipEventRecord(L.ev0, h->stream);
    if (!fast_lkl_launch(h->fast, h->stream, L.d_lkl, L.d_flags, false)) {
      h->fast.cur_lane = 0;
      set_error("fast_lkl_launch failed: %s", hipGetErrorString(hipGetLastError()));
      return NGHMM_ERR_HIP;
    }
    (void)hipEventRecord(L.ev1, h->stream);
  }
  h->fast.cur_lane = 0;
  if (!ok) {
    set_error("fast_lkl_prepare failed: %s", hipGetErrorString(hipGetLastError()));
    return NGHMM_ERR_HIP;
  }
  HIP_TRY(hipMemcpyAsync(L.h_lkl, L.d_lkl, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(L.h_flags, L.d_flags, NFLAGS * sizeof(int), hipMemcpyDeviceToHost,
                         h->stream));
  HIP_TRY(hipEventRecord(L.ev_done, h->stream));
  L.pending = true;
  return NGHMM_OK;
}

// wait for lane k; its values are in L.h_lkl afterwards
int lkl_wait(nghmm_t* h, int k) {
  auto& L = h->lane[k];
  HIP_TRY(hipEventSynchronize(L.ev_done));
  L.pending = false;
  float ms = 0;
  HIP_TRY(hipEventElapsedTime(&ms, L.ev0, L.ev1));
  h->ms[SLOT_LKL] += ms;
  h->launches[SLOT_LKL] += 1;
  const int* f = L.h_flags;
  if (f[FLAG_INVALID_LKL]) {
    h->fast.cur_lane = k;  // the lane is idle: its descriptor buffers serve the re-evaluation
    const int rc = redo_nonfinite(h, (uint32_t)L.ind.size(), L.ind.data(), L.F.data(),
                                  L.A.data(), L.h_lkl);
    h->fast.cur_lane = 0;
    if (rc != NGHMM_OK) return rc;
  }
  if (f[FLAG_NAN]) {
    set_error("value is NaN!");
    return NGHMM_ERR_NAN;
  }
  return NGHMM_OK;
}

// Emissions from the current frequencies (calc_emission, shared/HMM.cpp:144-154).  Fast
// mode checks the frequencies and rebuilds its interleaved frequency table now, but leaves
// the 8 B per site and individual of e_il to whoever reads them next: the first forward
// walk of the next EM iteration recomputes them on its way (fast_lkl_launch), anything
// else calls ensure_emissions.
int emission_impl(nghmm_t* h) {
  int rc;
  if ((rc = clear_flags(h))) return rc;
  tic(h);
  if (h->mode == NGHMM_MODE_FAST) {
    if (!fast_refresh_freq_table(h->fast, h->stream, h->d_freq, h->d_flags)) return NGHMM_ERR_HIP;
  } else {
    launch_emission_exact(h->stream, own_gl(h), h->d_freq, h->d_eprob, h->S, h->I, h->d_flags);
  }
  if ((rc = toc(h, SLOT_EMISSION, false))) return rc;
  HIP_TRY(hipGetLastError());
  return check_flags(h);
}

// ---- background pieces of a fused iteration (see nghmm_handle::BgSpan) ----
// The pieces' timers are events on the stream, and an event is a packet the queue has to work
// through between two kernels: 20 of them per iteration are 0.05 ms -- 8 % of an iteration of
// 100 x 100 000, 2 % of a rank of eight's, nothing at 1000 x 1 M.  Fast mode records them only
// with the switch `spans` (nghmm_kernel_ms of a fused iteration needs it); exact mode, whose
// iterations take seconds, always.
static bool spans_on(const nghmm_t* h) { return h->mode != NGHMM_MODE_FAST || h->fast.sw.spans != 0; }

int bg_begin(nghmm_t* h) {
  int rc;
  if (!h->d_flags_bg && (rc = dev_alloc(&h->d_flags_bg, (size_t)NFLAGS))) return rc;
  // (the last iteration's epilogue kernel has read the flags and cleared them again: no packet)
  if (!h->flags_bg_clear) HIP_TRY(hipMemsetAsync(h->d_flags_bg, 0, NFLAGS * sizeof(int), h->stream));
  h->flags_bg_clear = false;
  h->bg_used = 0;
  if (!spans_on(h))  // (no piece of this iteration is timed: nghmm_kernel_ms reads 0, not an earlier call's time)
    for (int slot : {SLOT_EMISSION, SLOT_FORWARD, SLOT_BACKWARD, SLOT_LKL, SLOT_ESTMAF, SLOT_LKL_FIRST, SLOT_BFGS}) {
      h->ms[slot] = 0;
      h->launches[slot] = 0;
    }
  return NGHMM_OK;
}

// start / stop the timer of one piece (nothing waits)
int bg_open(nghmm_t* h, int slot, hipStream_t st) {
  if (!st) st = h->stream;
  if (!spans_on(h)) return NGHMM_OK;
  if (h->bg_used == h->bg_spans.size()) {
    nghmm_handle::BgSpan sp;
    const unsigned evf = h->blocking_sync ? hipEventBlockingSync : hipEventDefault;
    HIP_TRY(hipEventCreateWithFlags(&sp.ev0, evf));
    HIP_TRY(hipEventCreateWithFlags(&sp.ev1, evf));
    h->bg_spans.push_back(sp);
  }
  h->bg_spans[h->bg_used].slot = slot;
  HIP_TRY(hipEventRecord(h->bg_spans[h->bg_used].ev0, st));
  return NGHMM_OK;
}

int bg_close(nghmm_t* h, hipStream_t st) {
  if (!spans_on(h)) return NGHMM_OK;
  HIP_TRY(hipEventRecord(h->bg_spans[h->bg_used].ev1, st ? st : h->stream));
  ++h->bg_used;
  return NGHMM_OK;
}

// end of the iteration: wait for everything, book the pieces' times, map their flags
int bg_finish(nghmm_t* h) {
  const int rc = check_flags(h, h->d_flags_bg);  // synchronises the stream
  bool seen[NSLOTS] = {};
  for (size_t k = 0; k < h->bg_used; ++k) {
    const auto& sp = h->bg_spans[k];
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, sp.ev0, sp.ev1));
    if (!seen[sp.slot]) {
      seen[sp.slot] = true;
      h->ms[sp.slot] = 0;
      h->launches[sp.slot] = 1;
    }
    h->ms[sp.slot] += ms;
  }
  h->bg_used = 0;
  return rc;
}

int ensure_emissions(nghmm_t* h) {
  if (h->mode != NGHMM_MODE_FAST || !h->fast.e_stale) return NGHMM_OK;
  int rc;
  tic(h);
  if (!fast_refresh_emissions(h->fast, h->stream, h->d_freq, h->d_flags)) return NGHMM_ERR_HIP;
  if ((rc = toc(h, SLOT_EMISSION, true))) return rc;
  HIP_TRY(hipGetLastError());
  return NGHMM_OK;
}

}  // namespace capi


extern "C" {

const char* nghmm_last_error(void) { return g_last_error.c_str(); }

const char* nghmm_strerror(int code) {
  switch (code) {
    case NGHMM_OK: return "ok";
    case NGHMM_ERR_INVALID_LKL: return "invalid Lkl found!";
    case NGHMM_ERR_FW_BW: return "Fw and Bw lkl do not match!";
    case NGHMM_ERR_INVALID_MAF: return "invalid MAF!";
    case NGHMM_ERR_NAN: return "value is NaN!\n";
    case NGHMM_ERR_FREQ_EST2: return "invalid allele frequencies";
    case NGHMM_ERR_ARG: return "invalid argument";
    case NGHMM_ERR_HIP: return "HIP runtime error";
    case NGHMM_ERR_NOMEM: return "out of device memory";
    case NGHMM_ERR_NOT_PACKABLE: return "not a called genotype (packed handle)";
    default: return "unknown error";
  }
}

int nghmm_has_hip(void) { return 1; }

int nghmm_create(nghmm_t** out, uint64_t n_ind, uint64_t n_sites, int device, int mode) {
  g_last_error.clear();
  const bool packed = (mode & NGHMM_GENO_PACKED) != 0;
  mode &= ~NGHMM_GENO_PACKED;
  if (!out || n_ind == 0 || n_sites == 0 || n_ind > 0xffffffffull ||
      (mode != NGHMM_MODE_EXACT && mode != NGHMM_MODE_FAST)) {
    set_error("nghmm_create: bad argument");
    return NGHMM_ERR_ARG;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    set_error("no HIP device available (%s): the hot path has no CPU fallback",
              hipGetErrorString(e));
    return NGHMM_ERR_HIP;
  }
  if (device < 0 || device >= ndev) {
    set_error("nghmm_create: device %d out of range (%d devices)", device, ndev);
    return NGHMM_ERR_ARG;
  }
  nghmm_t* h = new (std::nothrow) nghmm_handle;
  if (!h) return NGHMM_ERR_NOMEM;
  h->I = n_ind;
  h->S = n_sites;
  h->device = device;
  h->mode = mode;
  h->packed = packed;
  h->I_tot = n_ind;
  h->S_own = n_sites;
  h->fast.sw = Switches::from_env();  // the only place the environment is read
  int rc = NGHMM_OK;
  do {
    if (hipSetDevice(device) != hipSuccess) { rc = NGHMM_ERR_HIP; break; }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { rc = NGHMM_ERR_HIP; break; }
    if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) { rc = NGHMM_ERR_HIP; break; }
    const size_t cells = (size_t)n_ind * n_sites;
    if (packed) {
      if ((rc = dev_alloc(&h->d_codes, cells / 16 + 2))) break;
      if ((rc = dev_alloc(&h->d_cls_log, (size_t)12))) break;
      if ((rc = dev_alloc(&h->d_uniform, (size_t)1))) break;
      if ((rc = dev_alloc(&h->d_flags, (size_t)NFLAGS))) break;
      // the prepared likelihoods of the four classes: what the reference's reader and its two
      // normalisations make of a called genotype 0 / 1 / 2 and of a missing one
      // (shared/read_data.cpp:21,88-98; ngsF-HMM.cpp:117); row 3 is replaced by the value the
      // data's own uniform cells carry, if they carry another (--call_geno: det_log(1/3))
      const double u = std::log((double)1 / 3), ninf = -1e15;
      const double proto[12] = {0.0, ninf, ninf, ninf, 0.0, ninf, ninf, ninf, 0.0, u, u, u};
      if (hipMemcpy(h->d_cls_log, proto, sizeof proto, hipMemcpyHostToDevice) != hipSuccess ||
          hipMemset(h->d_flags, 0, NFLAGS * sizeof(int)) != hipSuccess) { rc = NGHMM_ERR_HIP; break; }
      launch_prepare_gl(h->stream, h->d_cls_log, 4, NGHMM_GL_LOG, 0, h->d_flags);
      if (hipMemcpyAsync(h->h_cls_proto, h->d_cls_log, sizeof h->h_cls_proto, hipMemcpyDeviceToHost,
                         h->stream) != hipSuccess ||
          sync_stream(h) != hipSuccess) { rc = NGHMM_ERR_HIP; break; }
    } else {
      if ((rc = dev_alloc(&h->d_gl, cells * 3))) break;
      if ((rc = dev_alloc(&h->d_flags, (size_t)NFLAGS))) break;
    }
    if ((rc = dev_alloc(&h->d_pos, n_sites))) break;
    if ((rc = dev_alloc(&h->d_freq, n_sites))) break;
    if ((rc = dev_alloc(&h->d_indF, n_ind))) break;
    if ((rc = dev_alloc(&h->d_alpha, n_ind))) break;
    if ((rc = dev_alloc(&h->d_ind_lkl, n_ind))) break;
    if (mode == NGHMM_MODE_EXACT) {
      if ((rc = dev_alloc(&h->d_eprob, cells * 2))) break;
      if ((rc = dev_alloc(&h->d_fw, (cells + n_ind) * 2))) break;
      if ((rc = dev_alloc(&h->d_marg, cells))) break;
      if (hipMemset(h->d_marg, 0, cells * sizeof(double)) != hipSuccess) { rc = NGHMM_ERR_HIP; break; }
    }
    if (hipMemset(h->d_freq, 0, n_sites * sizeof(double)) != hipSuccess) { rc = NGHMM_ERR_HIP; break; }
    h->h_indF.assign(n_ind, 0.0);
    h->h_alpha.assign(n_ind, 0.0);
    if (mode == NGHMM_MODE_FAST && !fast_create(h->fast, n_ind, n_sites, packed)) { rc = NGHMM_ERR_NOMEM; break; }
  } while (0);
  if (rc != NGHMM_OK) {
    if (g_last_error.empty()) set_error("nghmm_create failed (%d)", rc);
    nghmm_destroy(h);
    return rc;
  }
  *out = h;
  return NGHMM_OK;
}


int nghmm_destroy(nghmm_t* h) {
  if (!h) return NGHMM_OK;
  if (h->n_replicas.load() > 0) {
    set_error("nghmm_destroy: %d replica(s) of this handle are still alive", h->n_replicas.load());
    return NGHMM_ERR_ARG;
  }
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  chain_release(h);
  void* own[] = {h->d_freq, h->d_eprob, h->d_fw, h->d_marg, h->d_indF,
                 h->d_alpha, h->d_ind_lkl, h->d_flags, h->d_pt_ind, h->d_pt_F, h->d_pt_A,
                 h->d_pt_lkl, h->d_bp, h->d_path_sites, h->d_path, h->d_tmp, h->d_passes, h->d_vit,
                 h->d_gl_shard, h->d_geno, h->d_text, h->d_codes_shard, h->d_uniform, h->d_stage,
                 h->d_freq_new, h->d_hap,
                 h->d_stage8, h->g_send, h->g_recv, h->g_freq_own, h->g_freq_all};
  if (h->g_xstream) (void)hipStreamDestroy(h->g_xstream);
  if (h->d_flags_bg) (void)hipFree(h->d_flags_bg);
  for (auto& sp : h->bg_spans)
    for (hipEvent_t e : {sp.ev0, sp.ev1})
      if (e) (void)hipEventDestroy(e);
  for (auto& L : h->lane) {
    if (L.d_lkl) (void)hipFree(L.d_lkl);
    if (L.d_flags) (void)hipFree(L.d_flags);
    if (L.h_lkl) (void)hipHostFree(L.h_lkl);
    if (L.h_flags) (void)hipHostFree(L.h_flags);
    for (hipEvent_t e : {L.ev0, L.ev1, L.ev_done})
      if (e) (void)hipEventDestroy(e);
  }
  for (void* p : own)
    if (p) (void)hipFree(p);
  if (!h->parent) {
    void* data[] = {h->d_gl, h->d_pos, h->d_codes, h->d_cls_log};
    for (void* p : data)
      if (p) (void)hipFree(p);
  } else {
    h->parent->n_replicas.fetch_sub(1);
  }
  fast_destroy(h->fast);
  for (hipEvent_t e : {h->aux_ev0, h->aux_ev1, h->aux_go, h->aux_done})
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->aux_piece_ev)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->aux_estep_ev)
    if (e) (void)hipEventDestroy(e);
  if (h->d_aux_params) (void)hipFree(h->d_aux_params);
  if (h->h_flags_pin) (void)hipHostFree(h->h_flags_pin);
  if (h->h_lkl_pin) (void)hipHostFree(h->h_lkl_pin);
  if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev_sync) (void)hipEventDestroy(h->ev_sync);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return NGHMM_OK;
}

int nghmm_create_replica(nghmm_t** out, nghmm_t* parent) {
  g_last_error.clear();
  if (!out || !parent || !parent->loaded || parent->parent) {
    set_error("nghmm_create_replica: the parent must be a loaded handle that is not itself a replica");
    return NGHMM_ERR_ARG;
  }
  if (parent->I_tot != parent->I || parent->fast.shard.world > 1) {
    set_error("nghmm_create_replica: sharded handles have no replicas");
    return NGHMM_ERR_ARG;
  }
  int rc;
  if ((rc = use_device(parent))) return rc;
  nghmm_t* h = new (std::nothrow) nghmm_handle;
  if (!h) return NGHMM_ERR_NOMEM;
  h->I = parent->I;
  h->S = parent->S;
  h->device = parent->device;
  h->mode = parent->mode;
  h->packed = parent->packed;
  h->fast.sw = parent->fast.sw;
  h->I_tot = h->I;
  h->S_own = h->S;
  h->parent = parent;
  parent->n_replicas.fetch_add(1);
  // replicas run their EM iterations side by side from host threads of their own: each M-step
  // keeps to its thread (R pools of OpenMP workers would fight over the cores)
  parent->batch.set_max_threads(1);
  h->batch.set_max_threads(1);
  h->d_gl = parent->d_gl;
  h->d_pos = parent->d_pos;
  h->d_codes = parent->d_codes;
  h->d_cls_log = parent->d_cls_log;
  const uint64_t n_ind = h->I, n_sites = h->S;
  do {
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { rc = NGHMM_ERR_HIP; break; }
    // replicas are driven from one host thread each: wait without spinning
    h->blocking_sync = true;
    const unsigned evf = h->blocking_sync ? hipEventBlockingSync : hipEventDefault;
    if (hipEventCreateWithFlags(&h->ev0, evf) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev1, evf) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_sync, evf | hipEventDisableTiming) != hipSuccess) { rc = NGHMM_ERR_HIP; break; }
    const size_t cells = (size_t)n_ind * n_sites;
    if ((rc = dev_alloc(&h->d_flags, (size_t)NFLAGS))) break;
    if ((rc = dev_alloc(&h->d_freq, n_sites))) break;
    if ((rc = dev_alloc(&h->d_indF, n_ind))) break;
    if ((rc = dev_alloc(&h->d_alpha, n_ind))) break;
    if ((rc = dev_alloc(&h->d_ind_lkl, n_ind))) break;
    if (h->mode == NGHMM_MODE_EXACT) {
      if ((rc = dev_alloc(&h->d_eprob, cells * 2))) break;
      if ((rc = dev_alloc(&h->d_fw, (cells + n_ind) * 2))) break;
      if ((rc = dev_alloc(&h->d_marg, cells))) break;
      if (hipMemset(h->d_marg, 0, cells * sizeof(double)) != hipSuccess) { rc = NGHMM_ERR_HIP; break; }
    }
    if (hipMemset(h->d_freq, 0, n_sites * sizeof(double)) != hipSuccess) { rc = NGHMM_ERR_HIP; break; }
    h->h_indF.assign(n_ind, 0.0);
    h->h_alpha.assign(n_ind, 0.0);
    if (h->mode == NGHMM_MODE_FAST && !fast_create_replica(h->fast, parent->fast)) { rc = NGHMM_ERR_NOMEM; break; }
    h->loaded = true;
  } while (0);
  if (rc != NGHMM_OK) {
    if (g_last_error.empty()) set_error("nghmm_create_replica failed (%d)", rc);
    nghmm_destroy(h);
    return rc;
  }
  *out = h;
  return NGHMM_OK;
}

int nghmm_set_params(nghmm_t* h, const double* indF, const double* alpha, const double* freq) {
  g_last_error.clear();
  if (!h) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  if (indF || alpha) dbfgs_invalidate(h->fast);
  if (indF) {
    std::memcpy(h->h_indF.data(), indF, h->I * sizeof(double));
    HIP_TRY(hipMemcpyAsync(h->d_indF, indF, h->I * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  if (alpha) {
    std::memcpy(h->h_alpha.data(), alpha, h->I * sizeof(double));
    HIP_TRY(hipMemcpyAsync(h->d_alpha, alpha, h->I * sizeof(double), hipMemcpyHostToDevice,
                           h->stream));
  }
  if (freq)
    HIP_TRY(hipMemcpyAsync(h->d_freq, freq, h->S * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(sync_stream(h));
  return NGHMM_OK;
}

int nghmm_get_params(nghmm_t* h, double* indF, double* alpha, double* freq) {
  g_last_error.clear();
  if (!h) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  if (indF) std::memcpy(indF, h->h_indF.data(), h->I * sizeof(double));
  if (alpha) std::memcpy(alpha, h->h_alpha.data(), h->I * sizeof(double));
  if (freq) {
    HIP_TRY(hipMemcpyAsync(freq, h->d_freq, h->S * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(sync_stream(h));
  }
  return NGHMM_OK;
}

int nghmm_emission(nghmm_t* h) {
  g_last_error.clear();
  if (!h || !h->loaded) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  // init_output's place (parse_args.cpp:372-387): once per run, right before the EM loop -- also
  // where what the first iteration would otherwise allocate on its way is set up (the metric is
  // EM iterations per second of a run that starts here)
  if (!h->warmed) {
    h->warmed = true;
    h->batch.reserve(h->I);
    if (h->mode == NGHMM_MODE_FAST) {
      if ((rc = lane_setup(h, 0, (size_t)h->I * 5)) || (rc = lane_setup(h, 1, (size_t)h->I * 5))) return rc;
      if (h->I_tot == h->I && !fast_estmaf_reserve(h->fast, h->S)) return NGHMM_ERR_NOMEM;
      if (!h->fast.sw.no_dev_bfgs && dbfgs_available(h->fast) && !dbfgs_reserve(h->fast)) return NGHMM_ERR_NOMEM;
    }
  }
  return emission_impl(h);
}

int nghmm_estep(nghmm_t* h, double* ind_lkl) {
  g_last_error.clear();
  if (!h || !h->loaded) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  if (h->mode == NGHMM_MODE_FAST) return fast_estep_impl(h, ind_lkl, false);
  if ((rc = clear_flags(h))) return rc;
  {
    tic(h);
    (h->fast.sw.exact_serial ? launch_forward_exact : launch_forward_exact_pc)(
        h->stream, h->d_eprob, h->d_pos, h->S, h->I, (uint32_t)h->I, nullptr, h->d_indF,
        h->d_alpha, h->d_ind_lkl, h->d_fw, h->d_flags);
    if ((rc = toc(h, SLOT_FORWARD, false))) return rc;
    tic(h);
    h->tmp_is_posteriors = false;
    (h->fast.sw.exact_serial ? launch_backward_exact : launch_backward_exact_pc)(
        h->stream, h->d_eprob, h->d_pos, h->d_fw, h->S, h->I, h->d_indF, h->d_alpha,
        h->d_ind_lkl, h->d_marg, h->d_flags);
    if ((rc = toc(h, SLOT_BACKWARD, false))) return rc;
  }
  HIP_TRY(hipGetLastError());
  if (ind_lkl)
    HIP_TRY(hipMemcpyAsync(ind_lkl, h->d_ind_lkl, h->I * sizeof(double), hipMemcpyDeviceToHost,
                           h->stream));
  return check_flags(h);
}

int nghmm_lkl_batch(nghmm_t* h, uint32_t n_pts, const uint32_t* ind, const double* F,
                    const double* alpha, double* lkl) {
  g_last_error.clear();
  if (!h || !h->loaded || (n_pts && (!ind || !F || !alpha || !lkl))) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  return lkl_batch_impl(h, n_pts, ind, F, alpha, lkl, false);
}

// indF/alpha M-step.  fuse_estep (fast mode, nghmm_iter_em): the E-step that the
// reference runs BEFORE this M-step (EM.cpp:147-185) reads the same emissions and the
// same parameters as the M-step's first objective evaluation f(x) (EM.cpp:449-464 at the
// start values), and neither step writes anything the other reads.  So the first round
// runs first and leaves the forward walk of every individual behind (lane operators and
// checkpoints); the E-step then needs no forward pass over the emissions of its own.
//
// fuse_freq (nghmm_iter_em with --freq_est 1 on an unsharded handle): the allele-frequency
// step (EM.cpp:209-257) reads the E-step's posteriors and the likelihoods only, so it does not
// have to wait for the objective rounds either.  The backward sweep and est_maf, cut into
// parts, go onto the stream right behind each round's kernels: the GPU works on them while the
// host digests the round's values and prepares the next -- otherwise ~0.25 ms of idle device
// per round.  Same kernels on the same data in a different order; every result is unchanged.
//
// The rounds themselves take one of three routes:
//   first_or_plain_round  a whole round through lkl_batch_impl, synchronous: the first round
//                         of every M-step (it may double as the E-step's forward walk), and
//                         every round of a handle with neither of the two below;
//   round_with_background a whole round on lane 0's pinned buffers with the next background
//                         piece enqueued behind its kernels before the host waits for it;
//   two_lane_rounds       after the first round, small cohorts: the individuals in two halves
//                         on two lanes -- while the GPU evaluates one half's points the host
//                         scatters the other half's values into its L-BFGS-B machines, gathers
//                         their next points and enqueues them (EM.cpp:198-201 has no coupling
//                         between individuals either).  Worth it where a round is short against
//                         the host's share of it (100 x 100k: 1.36 -> 1.25 ms per iteration); at
//                         1000 x 1M two half launches lose to their emptier last wave batches
//                         what the overlap gains (37.4-38.4 against 36.9-37.9 ms).
namespace {

struct MstepRun {
  nghmm_t* h;
  int indF_fixed, alpha_fixed;
  bool fuse_estep, fuse_freq;
  double* ind_lkl;
  nghmm_hook_fn after_estep;
  void* user;

  BfgsBatch& batch;
  // called at the start of every objective round (exact mode's fused iteration feeds est_maf
  // to the second stream a piece at a time)
  std::function<int()> before_round;
  bool estep_pending = false;
  bool bg_active = false;
  bool bg_begun = false;        // the iteration's timing spans and background flags are set up
  // where the background work goes: the handle's stream (between the rounds), or -- rounds
  // planned on the device -- a second stream, NEXT TO the rounds and their planning kernels
  hipStream_t bg_stream = nullptr;
  const double *estep_F = nullptr, *estep_A = nullptr;  // the parameters the E-step reads
  bool tile_major = false;      // est_maf reads the tile-major posteriors in place
  uint32_t bg_parts = 0, bg_next = 0;  // est_maf parts queued / already on the stream
  // host-side wall time of the phases (switch `timing`)
  double t_gather = 0, t_lkl = 0, t_estep = 0, t_scatter = 0;
  std::vector<uint32_t> ind;
  std::vector<double> F, A, lkl;

  using clock = std::chrono::steady_clock;
  static double since(clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(clock::now() - t0).count();
  }

  MstepRun(nghmm_t* h_, int f, int a, bool fe, double* il, nghmm_hook_fn hook, void* u, bool ff)
      : h(h_), indF_fixed(f), alpha_fixed(a), fuse_estep(fe), fuse_freq(ff), ind_lkl(il),
        after_estep(hook), user(u), batch(h_->batch) {}

  // the E-step, then the caller's hook (multi-GPU: start moving the posteriors while the
  // remaining objective rounds run)
  int estep_then_hook(bool have_walk) {
    const int r = fast_estep_impl(h, ind_lkl, have_walk);
    if (r == NGHMM_OK && after_estep) after_estep(user);
    return r;
  }

  bool wants_background() const {
    // (bg_parts = 0: the backward sweep and est_maf after the objective rounds, not behind them)
    return fuse_freq && fuse_estep && !after_estep && h->mode == NGHMM_MODE_FAST &&
           h->I_tot == h->I && h->fast.sw.bg_parts != 0;
  }

  bool wants_two_lanes() const {
    // a site shard: every handle of the chain must make the same sequence of exchanges, and the
    // rule below looks at the waves per individual, which depend on the handle's own number of
    // sites -- one lane there
    if (h->fast.shard.world > 1) return false;
    return h->mode == NGHMM_MODE_FAST && h->I >= 2 && (uint64_t)h->I * h->fast.C < 16384;
  }

  // the E-step's backward sweep onto the stream now, est_maf queued in parts
  int start_background(bool have_walk) {
    int r;
    if (!have_walk && (r = ensure_emissions(h))) return r;
    if (!bg_begun && (r = bg_begin(h))) return r;
    bg_begun = true;
    hipStream_t bs = bg_stream ? bg_stream : h->stream;
    if ((r = bg_open(h, SLOT_FORWARD, bs))) return r;
    if (!fast_estep(h->fast, bs, estep_F ? estep_F : h->d_indF, estep_A ? estep_A : h->d_alpha,
                    h->d_ind_lkl, h->d_flags_bg, have_walk))
      return NGHMM_ERR_HIP;
    if ((r = bg_close(h, bs))) return r;
    h->ms[SLOT_BACKWARD] = 0;
    h->launches[SLOT_BACKWARD] = 0;
    h->marg_valid = false;
    h->tmp_is_posteriors = false;
    tile_major = fast_estmaf_in_place(h->fast, h->I);
    bg_parts = 1;
    if (fast_estmaf_splittable(h->fast, h->I, tile_major)) {
      // measured at 1000 x 1M (ms per iteration): 1 part 32.4, 2 parts 32.27, 3 parts 32.35,
      // 6 parts 32.5 (every part ends in a tail of partly filled CUs); without any of this 32.9
      bg_parts = h->fast.sw.bg_parts >= 1 ? (uint32_t)h->fast.sw.bg_parts : 1;
    }
    bg_next = 0;
    bg_active = true;
    return NGHMM_OK;
  }

  // the next part of est_maf behind whatever is on the stream (nothing waits)
  int push_background_piece() {
    if (bg_next >= bg_parts) return NGHMM_OK;
    const uint32_t part = bg_next++;
    int q;
    hipStream_t bs = bg_stream ? bg_stream : h->stream;
    if ((q = bg_open(h, SLOT_ESTMAF, bs))) return q;
    if (!tile_major && (q = ensure_marg(h))) return q;
    if (!fast_estmaf(h->fast, bs, fast_gl_lin(h->fast),
                     tile_major ? h->fast.post : h->d_marg, h->S, h->I, h->I, h->d_freq,
                     tile_major, part, bg_parts))
      return NGHMM_ERR_HIP;
    return bg_close(h, bs);
  }

  int first_or_plain_round() {
    int rc;
    if (before_round && (rc = before_round())) return rc;
    auto t0 = clock::now();
    const size_t n = batch.gather(ind, F, A);
    lkl.resize(n);
    t_gather += since(t0);
    bool emit = estep_pending;
    t0 = clock::now();
    if (n) {
      if ((rc = lkl_batch_impl(h, (uint32_t)n, ind.data(), F.data(), A.data(), lkl.data(), true,
                               &emit)))
        return rc;
    } else {
      emit = false;
    }
    t_lkl += since(t0);
    if (estep_pending) {
      t0 = clock::now();
      if ((rc = wants_background() ? start_background(emit) : estep_then_hook(emit))) return rc;
      estep_pending = false;
      t_estep += since(t0);
    }
    t0 = clock::now();
    batch.scatter(lkl.data());
    t_scatter += since(t0);
    return NGHMM_OK;
  }

  int round_with_background() {
    int rc;
    auto& L = h->lane[0];
    auto t0 = clock::now();
    const size_t n = batch.gather(L.ind, L.F, L.A, 0, h->I);
    t_gather += since(t0);
    if (n) {
      t0 = clock::now();
      if ((rc = lkl_submit(h, 0))) return rc;
      if ((rc = push_background_piece())) return rc;
      if ((rc = lkl_wait(h, 0))) return rc;
      t_lkl += since(t0);
    }
    t0 = clock::now();
    batch.scatter(L.h_lkl, 0, h->I);
    t_scatter += since(t0);
    return NGHMM_OK;
  }

  // lane k's next points, enqueued; nothing waits
  int feed_lane(int k) {
    auto& L = h->lane[k];
    for (;;) {
      auto t1 = clock::now();
      const size_t n = batch.gather(L.ind, L.F, L.A, L.lo, L.hi);
      t_gather += since(t1);
      if (n) {
        t1 = clock::now();
        const int r = lkl_submit(h, k);
        t_lkl += since(t1);
        return r;
      }
      if (batch.active_in(L.lo, L.hi) == 0) return NGHMM_OK;
      batch.scatter(L.h_lkl, L.lo, L.hi);  // only non-finite points this round: no launch
    }
  }

  // every remaining round of the M-step
  int two_lane_rounds() {
    int rc;
    const uint64_t half = (h->I + 1) / 2;
    h->lane[0].lo = 0;
    h->lane[0].hi = half;
    h->lane[1].lo = half;
    h->lane[1].hi = h->I;
    if ((rc = ensure_emissions(h))) return rc;
    for (int k = 0; k < 2; ++k)
      if ((rc = feed_lane(k))) return rc;
    while (bg_active && bg_next < bg_parts)  // (fused iteration) est_maf behind the two halves
      if ((rc = push_background_piece())) return rc;
    while (h->lane[0].pending || h->lane[1].pending) {
      for (int k = 0; k < 2; ++k) {
        auto& L = h->lane[k];
        if (!L.pending) continue;
        auto t1 = clock::now();
        if ((rc = lkl_wait(h, k))) return rc;
        t_lkl += since(t1);
        t1 = clock::now();
        batch.scatter(L.h_lkl, L.lo, L.hi);
        t_scatter += since(t1);
        if ((rc = feed_lane(k))) return rc;
      }
    }
    return NGHMM_OK;
  }

  // ---- the rounds planned on the device (kernels_bfgs.hip) ----
  // Fast mode, a handle that holds whole chains (no site shard): the L-BFGS-B machines live in
  // device memory, k_bfgs_advance turns a round's values into the next round's points, and the
  // host only learns HOW MANY groups of which loop-body version the next round has -- by polling
  // a word of pinned memory the planning kernel's last workgroup writes -- and launches them.
  // No copy, no event wait, no host arithmetic between two rounds.
  bool wants_device() const {
    return h->mode == NGHMM_MODE_FAST && !before_round && h->g_n <= 1 &&
           !h->fast.sw.no_dev_bfgs && dbfgs_available(h->fast);
  }

  int run_device(nghmm_mstep_stats* stats, bool* freq_done) {
    FastState& fs = h->fast;
    int rc;
    if (!dbfgs_reserve(fs)) {
      set_error("out of device memory (device-side L-BFGS-B state)");
      return NGHMM_ERR_NOMEM;
    }
    if ((rc = bg_begin(h))) return rc;  // timing spans of the rounds; flags of the background work
    bg_begun = true;
    // The E-step's backward sweep and est_maf need round 1's walk and nothing else of the
    // M-step: they go onto a second stream and run NEXT TO the later rounds, so that the chip
    // has work while a planning kernel (one wave per individual) and its plan's way to the host
    // would leave it idle.  The E-step reads the parameters the M-step started from (a copy the
    // planning kernel of round 1 makes: finished individuals' new ones go into d_indF / d_alpha
    // while the sweep may still be running).
    const bool overlap = wants_background() && fast_estmaf_in_place(fs, h->I) && !fs.sw.no_bg_stream;
    struct AuxDrain {  // nothing may be left running on the second stream when this returns
      hipStream_t s = nullptr;
      ~AuxDrain() { if (s) (void)hipStreamSynchronize(s); }
    } drain;
    if (overlap) {
      if (!h->aux_stream) HIP_TRY(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
      if (!h->aux_go) HIP_TRY(hipEventCreateWithFlags(&h->aux_go, hipEventDisableTiming));
      if (!h->aux_done) HIP_TRY(hipEventCreateWithFlags(&h->aux_done, hipEventDisableTiming));
      if (!h->d_flags_bg && (rc = dev_alloc(&h->d_flags_bg, (size_t)NFLAGS))) return rc;
      bg_stream = drain.s = h->aux_stream;
    }
    h->ms[SLOT_BFGS] = 0;
    h->launches[SLOT_BFGS] = 0;
    estep_pending = fuse_estep;
    auto t0 = clock::now();
    if ((rc = bg_open(h, SLOT_BFGS))) return rc;
    if (!dbfgs_begin(fs, h->stream, h->d_indF, h->d_alpha, indF_fixed != 0, alpha_fixed != 0)) {
      set_error("dbfgs_begin failed: %s", hipGetErrorString(hipGetLastError()));
      return NGHMM_ERR_HIP;
    }
    if ((rc = bg_close(h))) return rc;
    if (overlap) {  // (the set this M-step's first planning kernel wrote -- maybe when the last M-step ended)
      estep_F = dbfgs_start_F(fs);
      estep_A = dbfgs_start_A(fs);
    }
    uint32_t round = 1, n_active = 0, prev_active = (uint32_t)h->I;
    std::vector<FastState::ModeRange> ranges;
    const bool yield = h->blocking_sync;
    // the iteration ends by a word in pinned memory (dbfgs_epilogue) unless its kernels are timed
    const bool zero_copy = !spans_on(h) && fs.dev.h_epi;
    // ... written by the SECOND stream behind est_maf and the frequency table, which nothing on the
    // handle's stream reads before the next iteration's walk: the M-step's end waits for nothing
    const bool aux_epilogue = zero_copy && fuse_freq && ind_lkl != nullptr;
    bool aux_tail_queued = true;   // (overlap) est_maf .. epilogue are on the second stream, behind the sweep
    struct StaleGuard {            // ... whose emission ratios are stale once this M-step is over, however it ends
      FastState& fs;
      bool on = false;
      ~StaleGuard() { if (on) fs.e_stale = true; }
    } stale_when_over{h->fast};
    auto queue_aux_tail = [&]() -> int {
      int q;
      if ((q = push_background_piece())) return q;
      if (aux_epilogue) {
        if (!fast_refresh_freq_table(h->fast, h->aux_stream, h->d_freq, h->d_flags_bg)) return NGHMM_ERR_HIP;
        // (the emission ratios the remaining rounds read are those of the OLD frequencies, as the
        // reference's M-step reads the old e_prob (EM.cpp:198-201 before :252-257): stale only once
        // this M-step is over)
        h->fast.e_stale = false;
        stale_when_over.on = true;
        if (!dbfgs_epilogue(fs, h->aux_stream, h->d_flags_bg, (uint32_t)NFLAGS, h->d_ind_lkl)) {
          set_error("the iteration's epilogue kernel failed to launch: %s", hipGetErrorString(hipGetLastError()));
          return NGHMM_ERR_HIP;
        }
      } else {
        HIP_TRY(hipEventRecord(h->aux_done, h->aux_stream));
      }
      aux_tail_queued = true;
      return NGHMM_OK;
    };
    for (;;) {
      if (!dbfgs_wait_plan(fs, h->stream, round, &n_active, &ranges, yield)) {
        set_error("the device-side M-step did not publish round %u: %s", round,
                  hipGetErrorString(hipGetLastError()));
        return NGHMM_ERR_HIP;
      }
      if (n_active == 0) break;
      // (what a plan can be: round 1 has everybody, later rounds never more than the one before)
      if ((round == 1 && n_active != h->I) || n_active > prev_active) {
        set_error("the device-side M-step planned %u individuals for round %u after %u", n_active,
                  round, prev_active);
        return NGHMM_ERR_HIP;
      }
      prev_active = n_active;
      for (const auto& r : ranges) fs.mode_ind_rounds[r.mode] += r.count;
      bool emit = false;
      if (round == 1) {
        emit = estep_pending && n_active == h->I;
        if (!emit && (rc = ensure_emissions(h))) return rc;
      }
      if ((rc = bg_open(h, round == 1 ? SLOT_LKL_FIRST : SLOT_LKL))) return rc;
      if (!dbfgs_launch_round(fs, h->stream, round, n_active, ranges, emit)) {
        set_error("objective round %u failed to launch: %s", round, hipGetErrorString(hipGetLastError()));
        return NGHMM_ERR_HIP;
      }
      if ((rc = bg_close(h))) return rc;
      if (overlap && round == 1) HIP_TRY(hipEventRecord(h->aux_go, h->stream));  // the walk is there
      if ((rc = bg_open(h, SLOT_BFGS))) return rc;
      if (!dbfgs_advance(fs, h->stream, round, n_active)) {
        set_error("k_bfgs_advance failed to launch: %s", hipGetErrorString(hipGetLastError()));
        return NGHMM_ERR_HIP;
      }
      if ((rc = bg_close(h))) return rc;
      if (fs.sw.dbg_abort_round > 0 && round == (uint32_t)fs.sw.dbg_abort_round) {
        // (test hook: an M-step that ends early, with plans published and counters left behind)
        set_error("M-step ended after round %u by the switch dbg_abort_round", round);
        return NGHMM_ERR_HIP;
      }
      // behind the round and its planning kernel: the E-step's backward sweep (round 1), est_maf
      // in parts (rounds 2, 3, ...) -- the GPU works on them while the plan travels to the host
      if (estep_pending) {
        if (overlap) {  // the whole E-step + frequency step, next to everything that follows here
          // (the sweep now; est_maf's launches -- six of them, 20-40 us of this thread -- once
          // round 2 is on its way: round 1's planning kernel is shorter than that, and its plan
          // would wait for the host)
          HIP_TRY(hipStreamWaitEvent(h->aux_stream, h->aux_go, 0));
          if ((rc = start_background(emit))) return rc;
          bg_parts = 1;
          aux_tail_queued = false;
        } else if ((rc = wants_background() ? start_background(emit) : estep_then_hook(emit))) {
          return rc;
        }
        estep_pending = false;
      } else if (overlap && bg_active && !aux_tail_queued) {
        if ((rc = queue_aux_tail())) return rc;
      } else if (bg_active && !overlap) {
        if ((rc = push_background_piece())) return rc;
      }
      ++round;
    }
    t_lkl += since(t0);
    if (estep_pending && (rc = estep_then_hook(false))) return rc;
    if (bg_active && overlap && !aux_tail_queued && (rc = queue_aux_tail())) return rc;  // (an M-step of one round)
    if (bg_active && !(overlap && aux_epilogue)) {  // what is left of the background work, then the frequency table
      if (overlap) HIP_TRY(hipStreamWaitEvent(h->stream, h->aux_done, 0));
      bg_stream = nullptr;
      while (bg_next < bg_parts)
        if ((rc = push_background_piece())) return rc;
      if ((rc = bg_open(h, SLOT_EMISSION))) return rc;
      if (!fast_refresh_freq_table(h->fast, h->stream, h->d_freq, h->d_flags_bg)) return NGHMM_ERR_HIP;
      if ((rc = bg_close(h))) return rc;
    }
    bg_stream = nullptr;
    // (the plan that came out empty was published by the last kernel that touched the machines:
    // every individual's parameters are in pinned memory, and on the device)
    dbfgs_end(fs, round);
    std::memcpy(h->h_indF.data(), fs.dev.h_F, h->I * sizeof(double));
    std::memcpy(h->h_alpha.data(), fs.dev.h_A, h->I * sizeof(double));
    const bool lkl_out = bg_active && ind_lkl;
    if (zero_copy) {
      // The iteration's end without a copy, an event or a stream synchronisation: a one-workgroup
      // kernel behind the background work (on ITS stream when it has one: flags and log-likelihoods
      // are its products) writes them to pinned memory and a word the host polls.  The next
      // M-step's first planning kernel goes onto the handle's stream at once (dbfgs_preplan: the
      // parameters are final): its 20-30 us and its plan's way to the host then cost the next
      // iteration nothing.
      if (!(overlap && aux_epilogue) &&
          !dbfgs_epilogue(fs, h->stream, h->d_flags_bg, (uint32_t)NFLAGS, lkl_out ? h->d_ind_lkl : nullptr)) {
        set_error("the iteration's epilogue kernel failed to launch: %s", hipGetErrorString(hipGetLastError()));
        return NGHMM_ERR_HIP;
      }
      h->flags_bg_clear = true;
      (void)dbfgs_preplan(fs, h->stream, indF_fixed != 0, alpha_fixed != 0);
      int f[NFLAGS];
      if (!dbfgs_wait_epilogue(fs, overlap && aux_epilogue ? h->aux_stream : h->stream, f, (uint32_t)NFLAGS, yield)) {
        set_error("the iteration's epilogue kernel did not report: %s", hipGetErrorString(hipGetLastError()));
        return NGHMM_ERR_HIP;
      }
      if (lkl_out) std::memcpy(ind_lkl, fs.dev.h_epi_lkl, h->I * sizeof(double));
      rc = map_flags(f);
    } else {
      if (lkl_out) {  // (through pinned memory: the copy waits for nothing, bg_finish waits once for everything)
        if (!h->h_lkl_pin)
          HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h->h_lkl_pin), h->I * sizeof(double), hipHostMallocDefault));
        HIP_TRY(hipMemcpyAsync(h->h_lkl_pin, h->d_ind_lkl, h->I * sizeof(double), hipMemcpyDeviceToHost,
                               h->stream));
      }
      rc = bg_finish(h);  // waits for the stream; the spans' times; the background work's flags
      if (lkl_out) std::memcpy(ind_lkl, h->h_lkl_pin, h->I * sizeof(double));
    }
    h->ms[SLOT_LKL] += h->ms[SLOT_LKL_FIRST];
    h->launches[SLOT_LKL] = round - 1;
    if (rc) return rc;
    if (bg_active && freq_done) *freq_done = true;
    const unsigned long long* sv = fs.dev.stats_host;
    if (sv[5]) {
      set_error("invalid Lkl found!");
      return NGHMM_ERR_INVALID_LKL;
    }
    h->lkl_redone += sv[4];
    if (stats) {
      stats->points = sv[0];
      stats->ref_forward_calls = sv[1];
      stats->ind_rounds = sv[2];
      stats->rounds = (uint32_t)sv[3];
    }
    return NGHMM_OK;
  }

  // *freq_done tells the caller that the frequency step (and the frequency-table refresh of
  // nghmm_init_emission) has been done here.
  int run(nghmm_mstep_stats* stats, bool* freq_done) {
    int rc;
    if (freq_done) *freq_done = false;
    if (stats) std::memset(stats, 0, sizeof *stats);
    h->ms[SLOT_LKL] = 0;
    h->launches[SLOT_LKL] = 0;
    h->ms[SLOT_LKL_FIRST] = 0;
    h->launches[SLOT_LKL_FIRST] = 0;
    if (indF_fixed && alpha_fixed)  // EM.cpp:191-193
      return fuse_estep ? estep_then_hook(false) : NGHMM_OK;

    if (wants_device()) return run_device(stats, freq_done);
    dbfgs_invalidate(h->fast);  // (this M-step writes the parameters a plan made in advance started from)

    auto t0 = clock::now();
    // (fast mode: getgradient's step by detmath on host and device alike, bfgs_problem.hpp)
    batch.set_det_pow(h->mode == NGHMM_MODE_FAST);
    batch.begin(h->I, h->h_indF.data(), h->h_alpha.data(), indF_fixed != 0, alpha_fixed != 0);
    t_gather += since(t0);
    estep_pending = fuse_estep;
    const bool two_lanes = wants_two_lanes();
    bool first_round = true;
    while (!batch.done()) {
      if (!first_round && two_lanes) {
        if ((rc = two_lane_rounds())) return rc;
        break;
      }
      rc = (bg_active && !first_round) ? round_with_background() : first_or_plain_round();
      if (rc) return rc;
      first_round = false;
    }
    if (estep_pending && (rc = estep_then_hook(false))) return rc;
    if (bg_active) {  // what is left of the background work, then the frequency table
      while (bg_next < bg_parts)
        if ((rc = push_background_piece())) return rc;
      if ((rc = bg_open(h, SLOT_EMISSION))) return rc;
      if (!fast_refresh_freq_table(h->fast, h->stream, h->d_freq, h->d_flags_bg)) return NGHMM_ERR_HIP;
      if ((rc = bg_close(h))) return rc;
    }
    batch.result(h->h_indF.data(), h->h_alpha.data());
    // (exact mode's overlapped E-step reads copies of the OLD parameters that its own stream
    // makes: the new ones must not land before those copies are done)
    if (h->param_snapshot_ev) HIP_TRY(hipStreamWaitEvent(h->stream, h->param_snapshot_ev, 0));
    HIP_TRY(hipMemcpyAsync(h->d_indF, h->h_indF.data(), h->I * sizeof(double), hipMemcpyHostToDevice,
                           h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_alpha, h->h_alpha.data(), h->I * sizeof(double),
                           hipMemcpyHostToDevice, h->stream));
    if (bg_active) {
      if (ind_lkl)
        HIP_TRY(hipMemcpyAsync(ind_lkl, h->d_ind_lkl, h->I * sizeof(double), hipMemcpyDeviceToHost,
                               h->stream));
      if ((rc = bg_finish(h))) return rc;
      if (freq_done) *freq_done = true;
    } else {
      HIP_TRY(sync_stream(h));
    }
    if (stats) {
      stats->rounds = batch.rounds();
      stats->points = batch.points();
      stats->ref_forward_calls = batch.ref_forward_calls();
      stats->ind_rounds = batch.ind_rounds();
    }
    return NGHMM_OK;
  }
};

}  // namespace

static int mstep_indf_impl(nghmm_t* h, int indF_fixed, int alpha_fixed, nghmm_mstep_stats* stats,
                           bool fuse_estep, double* ind_lkl, nghmm_hook_fn after_estep = nullptr,
                           void* user = nullptr, bool fuse_freq = false,
                           bool* freq_done = nullptr, std::function<int()> before_round = nullptr) {
  MstepRun run(h, indF_fixed, alpha_fixed, fuse_estep, ind_lkl, after_estep, user, fuse_freq);
  run.before_round = std::move(before_round);
  return run.run(stats, freq_done);
}

int nghmm_mstep_indf(nghmm_t* h, int indF_fixed, int alpha_fixed, nghmm_mstep_stats* stats) {
  g_last_error.clear();
  if (!h || !h->loaded) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  return mstep_indf_impl(h, indF_fixed, alpha_fixed, stats, false, nullptr);
}

int nghmm_bfgs_batch_host(uint64_t n_ind, double* indF, double* alpha, int indF_fixed,
                          int alpha_fixed, nghmm_objective_fn fn, void* user,
                          nghmm_mstep_stats* stats) {
  g_last_error.clear();
  if (!indF || !alpha || !fn) return NGHMM_ERR_ARG;
  if (stats) std::memset(stats, 0, sizeof *stats);
  if (indF_fixed && alpha_fixed) return NGHMM_OK;
  static thread_local BfgsBatch batch;  // solver storage reused from call to call
  batch.begin(n_ind, indF, alpha, indF_fixed != 0, alpha_fixed != 0);
  std::vector<uint32_t> ind;
  std::vector<double> F, A, lkl;
  while (!batch.done()) {
    const size_t n = batch.gather(ind, F, A);
    lkl.resize(n);
    for (size_t p = 0; p < n; ++p) lkl[p] = fn(ind[p], F[p], A[p], user);
    batch.scatter(lkl.data());
  }
  batch.result(indF, alpha);
  if (stats) {
    stats->rounds = batch.rounds();
    stats->points = batch.points();
    stats->ref_forward_calls = batch.ref_forward_calls();
    stats->ind_rounds = batch.ind_rounds();
  }
  return NGHMM_OK;
}

// flags: 1 = getgradient's step size by detmath (fast mode's, bfgs_problem.hpp: DetPow);
// 2 = the solver type the DEVICE runs (LbfgsbT<PtrStore> over a plain block of memory, the
// problems one after the other) instead of class Lbfgsb in lock-step rounds.  Same code path
// per problem either way (bfgs_problem.hpp): the CPU test-suite checks that both give the same
// bits, the GPU suite that k_bfgs_advance does.
int nghmm_bfgs_batch_host2(uint64_t n_ind, double* indF, double* alpha, int indF_fixed,
                           int alpha_fixed, nghmm_objective_fn fn, void* user,
                           nghmm_mstep_stats* stats, int flags) {
  g_last_error.clear();
  if (!indF || !alpha || !fn) return NGHMM_ERR_ARG;
  if (stats) std::memset(stats, 0, sizeof *stats);
  if (indF_fixed && alpha_fixed) return NGHMM_OK;
  const bool det = (flags & 1) != 0;
  if (!(flags & 2)) {
    BfgsBatch batch;
    batch.set_det_pow(det);
    batch.set_max_threads(1);
    batch.begin(n_ind, indF, alpha, indF_fixed != 0, alpha_fixed != 0);
    std::vector<uint32_t> ind;
    std::vector<double> F, A, lkl;
    while (!batch.done()) {
      const size_t n = batch.gather(ind, F, A);
      lkl.resize(n);
      for (size_t p = 0; p < n; ++p) lkl[p] = fn(ind[p], F[p], A[p], user);
      batch.scatter(lkl.data());
    }
    batch.result(indF, alpha);
    if (stats) {
      stats->rounds = batch.rounds();
      stats->points = batch.points();
      stats->ref_forward_calls = batch.ref_forward_calls();
      stats->ind_rounds = batch.ind_rounds();
    }
    return NGHMM_OK;
  }
  std::vector<double> block(LbfgsbPtrs::doubles(2, 10));
  uint64_t points = 0, ref_calls = 0, ind_rounds = 0;
  uint32_t rounds = 0;
  for (uint64_t i = 0; i < n_ind; ++i) {
    BfgsProblem p;
    bfgs_problem_begin(p, indF[i], alpha[i], indF_fixed != 0, alpha_fixed != 0);
    LbfgsbT<PtrStore> solver;
    solver.st_.bind(block.data(), 2, 10);
    for (;;) {
      if (det) bfgs_plan<DetPow>(p);
      else bfgs_plan<LibmPow>(p);
      ++p.n_rounds;
      ++ind_rounds;
      double lklv[5] = {0, 0, 0, 0, 0};
      for (int k = 0; k < 5; ++k)
        if (p.slot_used[k] && !p.slot_nonfinite[k]) {
          lklv[k] = fn((uint32_t)i, p.pt[k][0], p.pt[k][1], user);
          ++points;
        }
      const bool again = bfgs_consume(p, solver, lklv, ref_calls, [&](BfgsProblem& q) {
        const int nbd[2] = {2, 2};
        solver.start_bound(2, 10, q.x, q.lb, q.ub, nbd, 1.0e6, 1.0e-3);
      });
      if (!again) break;
    }
    if (p.n_rounds > rounds) rounds = p.n_rounds;
    indF[i] = p.x[0];
    alpha[i] = p.x[1];
  }
  if (stats) {
    stats->rounds = rounds;
    stats->points = points;
    stats->ref_forward_calls = ref_calls;
    stats->ind_rounds = ind_rounds;
  }
  return NGHMM_OK;
}

}  // extern "C"

// est_maf on the handle's own sites and individuals (shard = false) or on its frequency-step
// site range over all ranks' individuals (shard = true: the static site-shard copy of the
// likelihoods, posteriors in rank blocks)
int capi::estmaf_and_refresh(nghmm_t* h, bool shard, const double* d_marg_blocks, uint64_t S_own,
                              uint64_t I_tot, uint64_t I_blk, double* d_freq_out) {
  int rc;
  tic(h);
  if (h->mode == NGHMM_MODE_FAST) {
    // fast mode reads linear-space GL: its own copy of the handle's GL, or the site
    // shard, which the shard loaders exponentiate in place (packed: the linear class table)
    const GlView lin = !shard ? fast_gl_lin(h->fast)
                       : h->packed ? gl_packed(h->d_codes_shard, h->fast.cls_lin)
                                   : gl_dense(h->d_gl_shard);
    bool tile_major = false;
    if (!d_marg_blocks) {  // the handle's own posteriors of its whole site range
      tile_major = fast_estmaf_in_place(h->fast, I_tot);
      if (tile_major) {
        d_marg_blocks = h->fast.post;
      } else {
        if ((rc = ensure_marg(h))) return rc;
        d_marg_blocks = h->d_marg;
      }
    }
    if (!fast_estmaf(h->fast, h->stream, lin, d_marg_blocks, S_own, I_tot, I_blk, d_freq_out,
                     tile_major))
      return NGHMM_ERR_HIP;
  } else {
    if (I_blk != I_tot) {
      set_error("exact-mode est_maf expects one posterior block");
      return NGHMM_ERR_ARG;
    }
    const GlView lg = !shard ? own_gl(h)
                      : h->packed ? gl_packed(h->d_codes_shard, h->d_cls_log)
                                  : gl_dense(h->d_gl_shard);
    launch_estmaf_exact(h->stream, lg, d_marg_blocks, S_own, I_tot, d_freq_out, nullptr, 0);
  }
  if ((rc = toc(h, SLOT_ESTMAF, false))) return rc;
  HIP_TRY(hipGetLastError());
  return NGHMM_OK;
}

extern "C" {

// --freq_est 2 / --e_prob 2 AS INTENDED (opt-in; PARITY UNPINNED: the reference aborts on
// both): the loop of EM.cpp:224-263 as written, sites in order with the frequencies updated in
// place, minus its three defects -- kernels_ld.hip says which; oracle: orc_em_mstep_freq_ld.
static int mstep_freq_ld_impl(nghmm_t* h, int freq_est, int e_prob) {
  const bool exact = h->mode == NGHMM_MODE_EXACT;
  // an individual shard, a site shard (its range would start a pair chain of its own: est_maf
  // and no pair step at the range's first site, no f_prev carried in from the range before)
  // and a chain member all are pieces of a cohort: silently different frequencies
  if (h->I_tot != h->I || h->fast.shard.world > 1 || h->chain || h->g_n > 1) {
    set_error("the intended --freq_est 2 walks the sites in order on ONE handle: not available "
              "for a sharded cohort");
    return NGHMM_ERR_ARG;
  }
  if (e_prob == 2 && !exact) {
    set_error("the intended --e_prob 2 needs materialised emissions: NGHMM_MODE_EXACT only");
    return NGHMM_ERR_ARG;
  }
  if (h->I > 8192) {
    set_error("the intended --freq_est 2 holds a site pair's cohort in one workgroup: at most "
              "8192 individuals");
    return NGHMM_ERR_ARG;
  }
  int rc;
  if (!h->d_freq_new && (rc = dev_alloc(&h->d_freq_new, (size_t)h->S))) return rc;
  if (!h->d_hap && (rc = dev_alloc(&h->d_hap, (size_t)h->S * 4))) return rc;
  if ((rc = clear_flags(h))) return rc;
  if (!exact && (rc = ensure_marg(h))) return rc;  // site-major posteriors
  // the first site, or (freq_est 1) every site, by est_maf (EM.cpp:242-244)
  const uint64_t n_est = freq_est == 1 ? h->S : 1;
  tic(h);
  if (exact) {
    launch_estmaf_exact(h->stream, own_gl(h), h->d_marg, n_est, h->I, h->d_freq_new, nullptr, 0);
  } else if (!fast_estmaf(h->fast, h->stream, fast_gl_lin(h->fast), h->d_marg, n_est, h->I, h->I,
                          h->d_freq_new, false)) {
    return NGHMM_ERR_HIP;
  }
  if (!launch_freq_ld_chain(h->stream, exact, exact ? own_gl(h) : fast_gl_lin(h->fast), h->d_marg,
                            h->d_freq, h->d_freq_new, h->d_hap, h->S, h->I, freq_est, h->d_flags))
    return NGHMM_ERR_ARG;
  HIP_TRY(hipMemcpyAsync(h->d_freq, h->d_freq_new, h->S * sizeof(double), hipMemcpyDeviceToDevice,
                         h->stream));
  if ((rc = toc(h, SLOT_ESTMAF, false))) return rc;
  HIP_TRY(hipGetLastError());
  int f[NFLAGS];
  HIP_TRY(hipMemcpyAsync(f, h->d_flags, sizeof f, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(sync_stream(h));
  if (f[FLAG_LD_FREQ]) {
    set_error("invalid allele frequencies");  // shared/gen_func.cpp:1030-1031
    return NGHMM_ERR_FREQ_EST2;
  }
  if ((rc = emission_impl(h))) return rc;  // calc_emission for every site (EM.cpp:252-257)
  if (e_prob == 2) {                       // ... and calc_emissionLD past the first (:258-260)
    if ((rc = clear_flags(h))) return rc;
    launch_emission_ld_exact(h->stream, own_gl(h), h->d_freq, h->d_hap, h->d_eprob, h->S, h->I,
                             h->d_flags);
    HIP_TRY(hipGetLastError());
    if ((rc = check_flags(h))) return rc;
  }
  return NGHMM_OK;
}

int nghmm_mstep_freq(nghmm_t* h, int freq_est) {
  g_last_error.clear();
  if (!h || !h->loaded) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  if (freq_est & NGHMM_LD_INTENDED) {
    const int fe = freq_est & 3, ep = (freq_est & NGHMM_EPROB_LD) ? 2 : 1;
    if ((freq_est & ~(NGHMM_LD_INTENDED | NGHMM_EPROB_LD | 3)) || (fe != 1 && fe != 2)) {
      set_error("wrong MAF estimation method!");
      return NGHMM_ERR_ARG;
    }
    if (fe == 1 && ep == 1) freq_est = 1;  // nothing of the LD route is asked for
    else return mstep_freq_ld_impl(h, fe, ep);
  }
  if (freq_est == 0) return NGHMM_OK;  // EM.cpp:212-214
  if (freq_est == 2) {
    set_error("invalid allele frequencies");
    return NGHMM_ERR_FREQ_EST2;
  }
  if (freq_est != 1) {
    set_error("wrong MAF estimation method!");
    return NGHMM_ERR_ARG;
  }
  if (h->I_tot != h->I) {
    set_error("sharded handle: use nghmm_mstep_freq_sites_dev");
    return NGHMM_ERR_ARG;
  }
  if ((rc = estmaf_and_refresh(h, false, h->mode == NGHMM_MODE_FAST ? nullptr : h->d_marg, h->S,
                               h->I, h->I, h->d_freq)))
    return rc;
  return emission_impl(h);
}

int nghmm_estep_mstep(nghmm_t* h, int indF_fixed, int alpha_fixed, double* ind_lkl,
                      nghmm_mstep_stats* stats, nghmm_hook_fn after_estep, void* user) {
  g_last_error.clear();
  if (!h || !h->loaded) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  if (h->mode == NGHMM_MODE_FAST)
    return mstep_indf_impl(h, indF_fixed, alpha_fixed, stats, true, ind_lkl, after_estep, user);
  if ((rc = nghmm_estep(h, ind_lkl))) return rc;
  if (after_estep) after_estep(user);
  return nghmm_mstep_indf(h, indF_fixed, alpha_fixed, stats);
}

int nghmm_iter_em(nghmm_t* h, int freq_est, int indF_fixed, int alpha_fixed, double* ind_lkl,
                  nghmm_mstep_stats* stats) {
  g_last_error.clear();
  if (!h || !h->loaded) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  // fast mode: E-step and indF/alpha M-step share their first forward walk, and the frequency
  // step runs in the shadow of the objective rounds (mstep_indf_impl)
  if (h->mode == NGHMM_MODE_FAST) {
    bool freq_done = false;
    if ((rc = mstep_indf_impl(h, indF_fixed, alpha_fixed, stats, true, ind_lkl, nullptr, nullptr,
                              freq_est == 1, &freq_done)))
      return rc;
    return freq_done ? NGHMM_OK : nghmm_mstep_freq(h, freq_est);
  }
  if (h->mode == NGHMM_MODE_EXACT && freq_est == 1 && h->I_tot == h->I) {
    // Exact mode: est_maf (EM.cpp:209-257) reads the E-step's posteriors and the likelihoods
    // and writes the frequencies; the objective rounds (EM.cpp:198-201) read the emissions of
    // the OLD frequencies.  Neither touches what the other uses, and a round is a few hundred
    // latency-bound waves on a chip of 1024 SIMDs -- but they are the iteration's critical path
    // (its slowest individual's lock-step rounds x 0.30 s at 10^6 sites), and an est_maf that
    // holds every wave slot starves them: measured at 1000 x 1M, side by side at full occupancy
    // is no faster than one after the other (17.0 / 14.0 / 5.2 / 9.7 s either way).  So est_maf
    // goes onto a second stream in PIECES (ranges of sites): one piece per round, capped at
    // kExactBgWaves waves per SIMD (k_estmaf_exact<BG_WAVES>) -- the chains, whose waves raise
    // their issue priority, then lose ~8 % -- and whatever is left when the rounds are over
    // runs uncapped on the whole chip.  The emissions are refreshed when both are done.  Same
    // kernels, same data, same bits.
    // The E-step, too, is latency-bound chains (one per individual: 32 workgroups), and the
    // objective rounds need nothing from it -- both read the emissions of the old frequencies
    // and the current (indF, alpha).  It runs on the second stream NEXT TO the first rounds
    // (its own copies of indF / alpha, since the M-step uploads the
    // new ones when it ends; its own error flags and events), est_maf's pieces queue up behind
    // it, and its fatal conditions are looked at first when everything is done -- the
    // reference's E-step comes before its M-step (EM.cpp:147-201).
    constexpr bool overlap_estep = true;
    if (!h->aux_stream) {
      HIP_TRY(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
      HIP_TRY(hipEventCreate(&h->aux_ev0));
      HIP_TRY(hipEventCreate(&h->aux_ev1));
      HIP_TRY(hipEventCreateWithFlags(&h->aux_go, hipEventDisableTiming));
      for (auto& e : h->aux_piece_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      for (auto& e : h->aux_estep_ev) HIP_TRY(hipEventCreate(&e));
      if ((rc = dev_alloc(&h->d_aux_params, (size_t)h->I * 2))) return rc;
      if (!h->d_flags_bg && (rc = dev_alloc(&h->d_flags_bg, (size_t)NFLAGS))) return rc;
    }
    // whatever happens from here on, nothing may be left running on the second stream when
    // this call returns (the caller may destroy the handle or load other data next)
    struct AuxDrain {
      hipStream_t s;
      ~AuxDrain() { (void)hipStreamSynchronize(s); }
    } drain{h->aux_stream};
    HIP_TRY(hipEventRecord(h->aux_go, h->stream));
    HIP_TRY(hipStreamWaitEvent(h->aux_stream, h->aux_go, 0));
    if (overlap_estep) {
      hipStream_t as = h->aux_stream;
      double *aF = h->d_aux_params, *aA = h->d_aux_params + h->I;
      HIP_TRY(hipMemsetAsync(h->d_flags_bg, 0, NFLAGS * sizeof(int), as));
      HIP_TRY(hipMemcpyAsync(aF, h->d_indF, h->I * sizeof(double), hipMemcpyDeviceToDevice, as));
      HIP_TRY(hipMemcpyAsync(aA, h->d_alpha, h->I * sizeof(double), hipMemcpyDeviceToDevice, as));
      HIP_TRY(hipEventRecord(h->aux_estep_ev[0], as));
      h->param_snapshot_ev = h->aux_estep_ev[0];
      (h->fast.sw.exact_serial ? launch_forward_exact : launch_forward_exact_pc)(
          as, h->d_eprob, h->d_pos, h->S, h->I, (uint32_t)h->I, nullptr, aF, aA, h->d_ind_lkl, h->d_fw,
          h->d_flags_bg);
      HIP_TRY(hipEventRecord(h->aux_estep_ev[1], as));
      h->tmp_is_posteriors = false;
      (h->fast.sw.exact_serial ? launch_backward_exact : launch_backward_exact_pc)(
          as, h->d_eprob, h->d_pos, h->d_fw, h->S, h->I, aF, aA, h->d_ind_lkl, h->d_marg, h->d_flags_bg);
      HIP_TRY(hipEventRecord(h->aux_estep_ev[2], as));
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(h->aux_ev0, h->aux_stream));
    constexpr uint32_t kPieces = nghmm_t::kAuxPieces;
    const uint32_t n_pieces = h->S >= 64 * kPieces ? kPieces : 1;
    uint32_t next = 0, finished = 0;
    constexpr int cap = kExactBgWaves;
    auto push = [&](int bg_waves) -> int {
      const uint64_t s0 = h->S * next / n_pieces, s1 = h->S * (next + 1) / n_pieces;
      GlView gl = own_gl(h);
      gl.cell0 += s0 * h->I;
      launch_estmaf_exact(h->aux_stream, gl, h->d_marg + s0 * h->I, s1 - s0, h->I, h->d_freq + s0, nullptr,
                          bg_waves);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(h->aux_piece_ev[next], h->aux_stream));
      ++next;
      return NGHMM_OK;
    };
    auto before_round = [&]() -> int {  // at most two pieces queued underneath a round
      while (finished < next && hipEventQuery(h->aux_piece_ev[finished]) == hipSuccess) ++finished;
      (void)hipGetLastError();          // (hipErrorNotReady is not an error)
      if (next < n_pieces && (int)(next - finished) < kExactBgDepth) return push(cap);
      return NGHMM_OK;
    };
    rc = mstep_indf_impl(h, indF_fixed, alpha_fixed, stats, false, nullptr, nullptr, nullptr, false,
                         nullptr, before_round);
    h->param_snapshot_ev = nullptr;
    while (rc == NGHMM_OK && next < n_pieces) rc = push(0);  // the rest, on the whole chip
    HIP_TRY(hipEventRecord(h->aux_ev1, h->aux_stream));
    HIP_TRY(hipEventSynchronize(h->aux_ev1));  // also when the M-step failed
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->aux_ev0, h->aux_ev1));
    h->ms[SLOT_ESTMAF] = ms;
    h->launches[SLOT_ESTMAF] = 1;
    if (overlap_estep) {  // the E-step's outcome first, as in the reference's order
      HIP_TRY(hipEventElapsedTime(&ms, h->aux_estep_ev[0], h->aux_estep_ev[1]));
      h->ms[SLOT_FORWARD] = ms;
      h->launches[SLOT_FORWARD] = 1;
      HIP_TRY(hipEventElapsedTime(&ms, h->aux_estep_ev[1], h->aux_estep_ev[2]));
      h->ms[SLOT_BACKWARD] = ms;
      h->launches[SLOT_BACKWARD] = 1;
      if (ind_lkl)
        HIP_TRY(hipMemcpyAsync(ind_lkl, h->d_ind_lkl, h->I * sizeof(double), hipMemcpyDeviceToHost,
                               h->stream));
      const int rc_e = check_flags(h, h->d_flags_bg);  // synchronises the stream
      if (rc_e != NGHMM_OK) return rc_e;
    }
    if (rc != NGHMM_OK) return rc;
    return emission_impl(h);
  }
  if ((rc = nghmm_estep_mstep(h, indF_fixed, alpha_fixed, ind_lkl, stats, nullptr, nullptr)))
    return rc;
  return nghmm_mstep_freq(h, freq_est);
}

int nghmm_viterbi(nghmm_t* h, uint8_t* path) {
  g_last_error.clear();
  if (!h || !h->loaded || !path) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  const size_t cells = (size_t)h->I * h->S;
  const size_t blocked = viterbi_blocked_bytes(h->S, h->I);
  if (!h->d_bp && (rc = dev_alloc(&h->d_bp, blocked + h->I))) return rc;
  if (!h->d_path_sites && (rc = dev_alloc(&h->d_path_sites, blocked))) return rc;
  if (!h->d_path && (rc = dev_alloc(&h->d_path, cells))) return rc;
  const uint64_t chunk = viterbi_chunk_sites(h->S, h->I);
  if (!h->d_vit && (rc = dev_alloc(&h->d_vit, (size_t)chunk * h->I * 4 + h->I * 2))) return rc;
  tic(h);
  if (h->mode == NGHMM_MODE_FAST) {
    if ((rc = clear_flags(h))) return rc;
    if (!fast_viterbi(h->fast, h->stream, h->d_freq, h->d_indF, h->d_alpha, h->d_bp,
                      h->d_path_sites, h->d_flags, h->d_vit, chunk))
      return NGHMM_ERR_HIP;
  } else {
    launch_viterbi_exact(h->stream, h->d_eprob, h->d_pos, h->S, h->I, h->d_indF, h->d_alpha,
                         h->d_bp, h->d_path_sites, h->d_vit, chunk, h->fast.sw.exact_serial != 0);
  }
  launch_unblock_path(h->stream, h->d_path_sites, h->S, h->I, h->d_path);
  if ((rc = toc(h, SLOT_VITERBI, false))) return rc;
  HIP_TRY(hipGetLastError());
  // fast mode recomputes the log emissions first: "invalid MAF!" (HMM.cpp:145-146)
  if (h->mode == NGHMM_MODE_FAST && (rc = check_flags(h))) return rc;
  HIP_TRY(hipMemcpyAsync(path, h->d_path, cells, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(sync_stream(h));
  return NGHMM_OK;
}

void* nghmm_alloc_host(uint64_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

void nghmm_free_host(void* p) {
  if (p) (void)hipHostFree(p);
}

int nghmm_set_switch(nghmm_t* h, const char* name, long value) {
  g_last_error.clear();
  if (!h || !name) return NGHMM_ERR_ARG;
  if (std::strcmp(name, "fast_c") == 0) {
    set_error("nghmm_set_switch: fast_c is fixed when the handle is created (NGHMM_FAST_C in the environment)");
    return NGHMM_ERR_ARG;
  }
  if (!h->fast.sw.set(name, value)) {
    set_error("nghmm_set_switch: no switch named %s", name);
    return NGHMM_ERR_ARG;
  }
  return NGHMM_OK;
}

int nghmm_fast_layout(nghmm_t* h, uint32_t* waves_per_individual, uint64_t* sites_per_lane) {
  if (!h) return NGHMM_ERR_ARG;
  const bool fast = h->mode == NGHMM_MODE_FAST;
  if (waves_per_individual) *waves_per_individual = fast ? h->fast.C : 0;
  if (sites_per_lane) *sites_per_lane = fast ? h->fast.T : 0;
  return NGHMM_OK;
}

void* nghmm_stream(nghmm_t* h) { return h ? (void*)h->stream : nullptr; }

int nghmm_synchronize(nghmm_t* h) {
  g_last_error.clear();
  if (!h) return NGHMM_ERR_ARG;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(sync_stream(h));
  return NGHMM_OK;
}

int nghmm_kernel_ms(nghmm_t* h, int slot, double* ms, uint32_t* launches) {
  if (!h || slot < 0 || slot >= NSLOTS) return NGHMM_ERR_ARG;
  if (ms) *ms = h->ms[slot];
  if (launches) *launches = h->launches[slot];
  return NGHMM_OK;
}

int nghmm_debug_mode_counts(nghmm_t* h, nghmm_mode_count* out, uint32_t cap, uint32_t* n, int reset) {
  g_last_error.clear();
  if (!h || (!out && cap)) return NGHMM_ERR_ARG;
  uint32_t k = 0;
  for (const auto& kv : h->fast.mode_ind_rounds) {
    if (k < cap) out[k] = nghmm_mode_count{kv.first, kv.second};
    ++k;
  }
  if (n) *n = k;
  if (reset) h->fast.mode_ind_rounds.clear();
  return NGHMM_OK;
}

int nghmm_debug_estmaf_counts(nghmm_t* h, uint64_t out[5], int reset) {
  g_last_error.clear();
  if (!h || !out) return NGHMM_ERR_ARG;
  for (int k = 0; k < 5; ++k) out[k] = 0;
  if (!h->fast.est_counts) return NGHMM_OK;  // no frequency step yet (or exact mode)
  int rc;
  if ((rc = use_device(h))) return rc;
  uint32_t v[EST_COUNTS];
  HIP_TRY(sync_stream(h));
  HIP_TRY(hipMemcpy(v, h->fast.est_counts, sizeof v, hipMemcpyDeviceToHost));
  for (int k = 0; k < 5; ++k) out[k] = v[k];
  if (reset) {
    HIP_TRY(hipMemset(h->fast.est_counts, 0, sizeof v));
    HIP_TRY(hipDeviceSynchronize());
  }
  return NGHMM_OK;
}

}  // extern "C"


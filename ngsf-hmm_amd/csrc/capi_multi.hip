// capi_multi.hip -- several handles, one data set: individual shards (nghmm_shard_config, nghmm_group_*), site shards
// (nghmm_site_shard_setup, nghmm_chain_*), their Viterbi halves
// (implementation of include/nghmm.h; capi_internal.hpp has the handle and the shared helpers.)
#include "capi_internal.hpp"

// ---------------- multi-GPU ----------------

int nghmm_shard_config(nghmm_t* h, uint64_t n_ind_total, uint64_t ind_begin, uint64_t site_begin,
                       uint64_t n_sites_own) {
  g_last_error.clear();
  if (h && h->fast.shard.world > 1) {
    set_error("nghmm_shard_config: the handle is a site shard (one layout at a time)");
    return NGHMM_ERR_ARG;
  }
  if (!h || n_ind_total < h->I || ind_begin + h->I > n_ind_total ||
      site_begin + n_sites_own > h->S || n_ind_total % h->I != 0) {
    set_error("nghmm_shard_config: inconsistent shard (equal individuals per rank required)");
    return NGHMM_ERR_ARG;
  }
  h->I_tot = n_ind_total;
  h->ind_begin = ind_begin;
  h->site_begin = site_begin;
  h->S_own = n_sites_own;
  return NGHMM_OK;
}

// ---- site shards (fast mode): kernels_fast.hip, "site shards" ----
uint64_t nghmm_site_shard_bytes(nghmm_t* h) {
  // an objective round: <= 5 points per individual, six doubles each (the E-step: six per
  // individual); nghmm_lkl_batch calls with more points than that are refused
  return h ? (uint64_t)(h->I * 30 + 64) * sizeof(double) : 0;
}

int nghmm_site_shard_setup(nghmm_t* h, int rank, int world, void* send_dev, void* recv_dev,
                           uint64_t bytes_per_rank, nghmm_allgather_fn fn, void* user) {
  g_last_error.clear();
  if (!h || world < 1 || rank < 0 || rank >= world) return NGHMM_ERR_ARG;
  if (h->mode != NGHMM_MODE_FAST) {
    set_error("site shards are a fast-mode layout: the exact-mode recursion is one chain of "
              "roundings over all sites (shard by individual: nghmm_shard_config)");
    return NGHMM_ERR_ARG;
  }
  if (h->parent || h->n_replicas.load() > 0 || h->I_tot != h->I || h->g_n) {
    set_error("nghmm_site_shard_setup: not on a replica, a handle with replicas, an individual "
              "shard or a group member");
    return NGHMM_ERR_ARG;
  }
  int rc;
  if ((rc = use_device(h))) return rc;
  SiteShard& sh = h->fast.shard;
  if (world == 1) {
    sh = SiteShard{};
    return NGHMM_OK;
  }
  if (!send_dev || !recv_dev || !fn || bytes_per_rank < nghmm_site_shard_bytes(h)) {
    set_error("nghmm_site_shard_setup: buffers of nghmm_site_shard_bytes() (x world for recv) and "
              "an all-gather are needed");
    return NGHMM_ERR_ARG;
  }
  if (!sh.edges && (rc = dev_alloc(&sh.edges, (size_t)h->I * 8))) return rc;
  sh.rank = (uint32_t)rank;
  sh.world = (uint32_t)world;
  sh.send = static_cast<double*>(send_dev);
  sh.recv = static_cast<double*>(recv_dev);
  sh.cap = bytes_per_rank / sizeof(double);
  sh.allgather = fn;
  sh.user = user;
  return NGHMM_OK;
}

// Viterbi over a chain of site shards: forward in rank order (every handle starts from the
// scores the one before ended with), then back in reverse order (every handle starts from the
// state the one after found for the site in front of its first).  Same kernels, same order of
// operations per individual as one handle over all sites: the same path.
int nghmm_viterbi_shard_forward(nghmm_t* h, const double* scores_in, double* scores_out) {
  g_last_error.clear();
  if (!h || !h->loaded || !scores_out || h->mode != NGHMM_MODE_FAST) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  const size_t blocked = viterbi_blocked_bytes(h->S, h->I);
  if (!h->d_bp && (rc = dev_alloc(&h->d_bp, blocked + h->I))) return rc;
  if (!h->d_path_sites && (rc = dev_alloc(&h->d_path_sites, blocked))) return rc;
  if (!h->d_path && (rc = dev_alloc(&h->d_path, (size_t)h->I * h->S))) return rc;
  const uint64_t chunk = viterbi_chunk_sites(h->S, h->I);
  if (!h->d_vit && (rc = dev_alloc(&h->d_vit, (size_t)chunk * h->I * 4 + h->I * 2))) return rc;
  double* d_state = h->d_vit + (size_t)chunk * h->I * 4;
  if (scores_in)
    HIP_TRY(hipMemcpyAsync(d_state, scores_in, h->I * 2 * sizeof(double), hipMemcpyHostToDevice,
                           h->stream));
  if ((rc = clear_flags(h))) return rc;
  tic(h);
  if (!fast_viterbi_forward(h->fast, h->stream, h->d_freq, h->d_indF, h->d_alpha, h->d_bp, h->d_flags,
                            h->d_vit, chunk, scores_in == nullptr))
    return NGHMM_ERR_HIP;
  if ((rc = toc(h, SLOT_VITERBI, false))) return rc;
  if ((rc = check_flags(h))) return rc;  // "invalid MAF!" (HMM.cpp:145-146)
  HIP_TRY(hipMemcpyAsync(scores_out, d_state, h->I * 2 * sizeof(double), hipMemcpyDeviceToHost,
                         h->stream));
  HIP_TRY(sync_stream(h));
  return NGHMM_OK;
}

int nghmm_viterbi_shard_back(nghmm_t* h, const uint8_t* state_after, uint8_t* state_before,
                             uint8_t* path) {
  g_last_error.clear();
  if (!h || !h->loaded || !h->d_bp || !state_before || !path) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  uint8_t* d_last = h->d_bp + viterbi_blocked_bytes(h->S, h->I);
  if (state_after)  // else: the last range, whose forward half left the arg max there
    HIP_TRY(hipMemcpyAsync(d_last, state_after, h->I, hipMemcpyHostToDevice, h->stream));
  uint8_t* d_before = nullptr;
  if ((rc = dev_alloc(&d_before, (size_t)h->I))) return rc;
  launch_viterbi_back_exact(h->stream, h->d_bp, h->S, h->I, h->d_path_sites, d_before);
  launch_unblock_path(h->stream, h->d_path_sites, h->S, h->I, h->d_path);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess)
    e = hipMemcpyAsync(state_before, d_before, h->I, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(path, h->d_path, (size_t)h->I * h->S, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = sync_stream(h);
  (void)hipFree(d_before);
  HIP_TRY(e);
  return NGHMM_OK;
}

int nghmm_load_gl_site_shard(nghmm_t* h, const double* gl_site_shard) {
  g_last_error.clear();
  if (!h || !gl_site_shard) return NGHMM_ERR_ARG;
  if (h->packed) {
    set_error("packed handle: use nghmm_load_geno_site_shard_dev");
    return NGHMM_ERR_ARG;
  }
  int rc;
  if ((rc = use_device(h))) return rc;
  const size_t n = (size_t)h->S_own * h->I_tot * 3;
  if (h->d_gl_shard) (void)hipFree(h->d_gl_shard);
  h->d_gl_shard = nullptr;
  if ((rc = dev_alloc(&h->d_gl_shard, n))) return rc;
  HIP_TRY(hipMemcpyAsync(h->d_gl_shard, gl_site_shard, n * sizeof(double), hipMemcpyHostToDevice,
                         h->stream));
  if (h->mode == NGHMM_MODE_FAST) fast_exp(h->stream, h->d_gl_shard, h->d_gl_shard, n);
  HIP_TRY(sync_stream(h));
  return NGHMM_OK;
}

int nghmm_get_geno_codes_dev(nghmm_t* h, uint64_t site_lo, uint64_t site_hi, uint8_t* d_out) {
  g_last_error.clear();
  if (!h || !h->packed || !h->loaded || !d_out || site_lo > site_hi || site_hi > h->S)
    return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  launch_codes_to_bytes(h->stream, h->d_codes, site_lo * h->I, (site_hi - site_lo) * h->I, d_out);
  HIP_TRY(hipGetLastError());
  HIP_TRY(sync_stream(h));
  return NGHMM_OK;
}

int nghmm_load_geno_site_shard_dev(nghmm_t* h, const uint8_t* d_codes_bytes) {
  g_last_error.clear();
  if (!h || !h->packed || !d_codes_bytes) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  const size_t n = (size_t)h->S_own * h->I_tot;
  if (h->d_codes_shard) (void)hipFree(h->d_codes_shard);
  h->d_codes_shard = nullptr;
  if ((rc = dev_alloc(&h->d_codes_shard, n / 16 + 2))) return rc;
  HIP_TRY(hipDeviceSynchronize());  // the caller's buffer: written on a stream of its own (capi_load.hip)
  launch_bytes_to_codes(h->stream, d_codes_bytes, n, h->d_codes_shard);
  HIP_TRY(hipGetLastError());
  HIP_TRY(sync_stream(h));
  return NGHMM_OK;
}

int nghmm_load_gl_site_shard_dev(nghmm_t* h, const double* d_gl_site_shard) {
  g_last_error.clear();
  if (!h || !d_gl_site_shard) return NGHMM_ERR_ARG;
  if (h->packed) {
    set_error("packed handle: use nghmm_load_geno_site_shard_dev");
    return NGHMM_ERR_ARG;
  }
  int rc;
  if ((rc = use_device(h))) return rc;
  const size_t n = (size_t)h->S_own * h->I_tot * 3;
  if (h->d_gl_shard) (void)hipFree(h->d_gl_shard);
  h->d_gl_shard = nullptr;
  if ((rc = dev_alloc(&h->d_gl_shard, n))) return rc;
  HIP_TRY(hipDeviceSynchronize());  // the caller's buffer: written on a stream of its own (capi_load.hip)
  HIP_TRY(hipMemcpyAsync(h->d_gl_shard, d_gl_site_shard, n * sizeof(double),
                         hipMemcpyDeviceToDevice, h->stream));
  if (h->mode == NGHMM_MODE_FAST) fast_exp(h->stream, h->d_gl_shard, h->d_gl_shard, n);
  HIP_TRY(sync_stream(h));
  return NGHMM_OK;
}

int nghmm_pack_posteriors_dev(nghmm_t* h, uint64_t site_lo, uint64_t site_hi, double* d_out) {
  g_last_error.clear();
  if (!h || !d_out || site_lo > site_hi || site_hi > h->S) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  if (h->mode == NGHMM_MODE_FAST && site_lo == 0 && site_hi == h->S) {
    // every destination at once: the send buffer [rank][S_own][I] of equal contiguous site
    // ranges IS the site-major matrix, so convert the tile-major posteriors straight into it
    if (!fast_post_to_site_major(h->fast, h->stream, d_out)) return NGHMM_ERR_HIP;
    HIP_TRY(hipGetLastError());
    HIP_TRY(sync_stream(h));
    return NGHMM_OK;
  }
  // marg is site-major [S][I]: the slice of a destination rank is contiguous
  if ((rc = ensure_marg(h))) return rc;
  launch_copy_f64(h->stream, h->d_marg + site_lo * h->I, d_out, (site_hi - site_lo) * h->I);
  HIP_TRY(hipGetLastError());
  HIP_TRY(sync_stream(h));
  return NGHMM_OK;
}

int nghmm_mstep_freq_sites_dev(nghmm_t* h, const double* d_marg_blocks, double* d_freq_out) {
  g_last_error.clear();
  if (!h || !d_marg_blocks || !d_freq_out || !(h->packed ? (void*)h->d_codes_shard : (void*)h->d_gl_shard))
    return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  if (h->mode != NGHMM_MODE_FAST && h->I_tot != h->I) {
    // exact mode wants [S_own][I_tot]: re-block [rank][S_own][I] through the scratch buffer
    set_error("exact-mode sharded est_maf is not available; use NGHMM_MODE_FAST");
    return NGHMM_ERR_ARG;
  }
  if ((rc = estmaf_and_refresh(h, true, d_marg_blocks, h->S_own, h->I_tot, h->I, d_freq_out)))
    return rc;
  HIP_TRY(sync_stream(h));
  return NGHMM_OK;
}

int nghmm_set_freq_dev(nghmm_t* h, const double* d_freq_all) {
  g_last_error.clear();
  if (!h || !d_freq_all) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  HIP_TRY(hipMemcpyAsync(h->d_freq, d_freq_all, h->S * sizeof(double), hipMemcpyDeviceToDevice,
                         h->stream));
  return emission_impl(h);
}


// ---------------- one process, several GPUs ----------------
// The group functions orchestrate what ngsf-hmm_amd/distributed.py does with RCCL between
// processes, inside one process with direct peer copies: on an MI355X node every pair of GPUs
// has its own xGMI link, so n*(n-1) simultaneous point-to-point copies ARE the all-to-all.

namespace {

struct GroupHook {
  nghmm_t** hs;
  int n, r;
  int rc;
};

// after rank r's E-step: its posteriors, site-major, into the send buffer; then one slice to
// every rank's receive buffer, on the exchange stream (the objective rounds go on meanwhile)
void group_after_estep(void* user) {
  GroupHook* g = static_cast<GroupHook*>(user);
  nghmm_t* h = g->hs[g->r];
  g->rc = nghmm_pack_posteriors_dev(h, 0, h->S, h->g_send);
  if (g->rc != NGHMM_OK) return;
  const size_t blk = (size_t)h->S_own * h->I;
  for (int q = 0; q < g->n; ++q) {
    if (hipMemcpyAsync(g->hs[q]->g_recv + (size_t)g->r * blk, h->g_send + (size_t)q * blk,
                       blk * sizeof(double), hipMemcpyDeviceToDevice, h->g_xstream) != hipSuccess) {
      g->rc = NGHMM_ERR_HIP;
      return;
    }
  }
}

int for_each_rank(int n, const std::function<int(int)>& fn) {
  std::vector<int> rcs(n, NGHMM_OK);
  std::vector<std::string> msgs(n);
  if (n == 1) {
    rcs[0] = fn(0);
  } else {
    std::vector<std::thread> th;
    for (int r = 0; r < n; ++r)
      th.emplace_back([&, r] {
        rcs[r] = fn(r);
        if (rcs[r] != NGHMM_OK) msgs[r] = g_last_error;  // thread-local: carry it over
      });
    for (auto& t : th) t.join();
  }
  for (int r = 0; r < n; ++r)
    if (rcs[r] != NGHMM_OK) {
      if (!msgs[r].empty()) g_last_error = msgs[r];
      return rcs[r];
    }
  return NGHMM_OK;
}

// phases 2 and 3 of a group iteration, once every rank's posteriors have arrived: est_maf on
// the own site range over all individuals (rank blocks = the global individual order), the
// frequencies to everybody, and their installation (emissions follow lazily)
int group_freq_phases(nghmm_t** hs, int n) {
  const uint64_t S_own = hs[0]->S / n;
  int rc = for_each_rank(n, [&](int r) -> int {
    nghmm_t* h = hs[r];
    int rr = nghmm_mstep_freq_sites_dev(h, h->g_recv, h->g_freq_own);
    if (rr != NGHMM_OK) return rr;
    for (int q = 0; q < n; ++q)
      if (hipMemcpyAsync(hs[q]->g_freq_all + (size_t)r * S_own, h->g_freq_own, S_own * sizeof(double),
                         hipMemcpyDeviceToDevice, h->g_xstream) != hipSuccess)
        return NGHMM_ERR_HIP;
    return hipStreamSynchronize(h->g_xstream) == hipSuccess ? NGHMM_OK : NGHMM_ERR_HIP;
  });
  if (rc != NGHMM_OK) return rc;
  return for_each_rank(n, [&](int r) -> int { return nghmm_set_freq_dev(hs[r], hs[r]->g_freq_all); });
}

}  // namespace

int nghmm_group_setup(nghmm_t** hs, int n) {
  g_last_error.clear();
  if (!hs || n < 1) return NGHMM_ERR_ARG;
  for (int r = 0; r < n; ++r)  // the members' M-steps run side by side on host threads of their own
    if (hs[r]) hs[r]->batch.set_max_threads(n > 1 ? 1 : 64);
  nghmm_t* h0 = hs[0];
  for (int r = 0; r < n; ++r) {
    nghmm_t* h = hs[r];
    if (!h || !h->loaded || (n > 1 && (h->parent || h->n_replicas.load() > 0)) ||
        h->fast.shard.world > 1 || h->I != h0->I ||
        h->S != h0->S || h->mode != h0->mode ||
        h->packed != h0->packed) {
      set_error("nghmm_group_setup: the handles must be loaded and agree in size, mode and packing");
      return NGHMM_ERR_ARG;
    }
  }
  const uint64_t I = h0->I, S = h0->S, I_tot = I * n;
  if (S % n != 0) {
    set_error("nghmm_group_setup: %llu sites do not divide by %d handles", (unsigned long long)S, n);
    return NGHMM_ERR_ARG;
  }
  if (n > 1 && h0->mode != NGHMM_MODE_FAST) {
    set_error("nghmm_group_setup: several handles need NGHMM_MODE_FAST");
    return NGHMM_ERR_ARG;
  }
  const uint64_t S_own = S / n;
  int rc;
  // peer access between the devices involved (a no-op for handles that share a device)
  for (int r = 0; r < n; ++r)
    for (int q = 0; q < n; ++q)
      if (hs[r]->device != hs[q]->device) {
        int can = 0;
        HIP_TRY(hipDeviceCanAccessPeer(&can, hs[r]->device, hs[q]->device));
        if (!can) {
          set_error("nghmm_group_setup: device %d cannot access device %d", hs[r]->device, hs[q]->device);
          return NGHMM_ERR_HIP;
        }
        HIP_TRY(hipSetDevice(hs[r]->device));
        const hipError_t e = hipDeviceEnablePeerAccess(hs[q]->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_TRY(e);
        (void)hipGetLastError();
      }
  for (int r = 0; r < n; ++r) {
    nghmm_t* h = hs[r];
    if ((rc = use_device(h))) return rc;
    if ((rc = nghmm_shard_config(h, I_tot, (uint64_t)r * I, (uint64_t)r * S_own, S_own))) return rc;
    h->g_n = n;
    h->g_rank = r;
    if (n == 1) continue;
    void* old[] = {h->g_send, h->g_recv, h->g_freq_own, h->g_freq_all};
    for (void* p : old)
      if (p) (void)hipFree(p);
    h->g_send = h->g_recv = h->g_freq_own = h->g_freq_all = nullptr;
    if ((rc = dev_alloc(&h->g_send, (size_t)S * I))) return rc;
    if ((rc = dev_alloc(&h->g_recv, (size_t)S * I))) return rc;
    if ((rc = dev_alloc(&h->g_freq_own, (size_t)S_own))) return rc;
    if ((rc = dev_alloc(&h->g_freq_all, (size_t)S))) return rc;
    if (!h->g_xstream) HIP_TRY(hipStreamCreateWithFlags(&h->g_xstream, hipStreamNonBlocking));
  }
  if (n == 1) return NGHMM_OK;
  // static site-shard copies: rank r gets the likelihoods of ALL individuals for its site range,
  // pulled from every rank's own matrix with strided peer copies
  for (int r = 0; r < n; ++r) {
    nghmm_t* h = hs[r];
    if ((rc = use_device(h))) return rc;
    const uint64_t lo = (uint64_t)r * S_own;
    if (!h->packed) {
      if (h->d_gl_shard) (void)hipFree(h->d_gl_shard);
      h->d_gl_shard = nullptr;
      if ((rc = dev_alloc(&h->d_gl_shard, (size_t)S_own * I_tot * 3))) return rc;
      for (int q = 0; q < n; ++q)
        HIP_TRY(hipMemcpy2DAsync(h->d_gl_shard + (size_t)q * I * 3, I_tot * 3 * sizeof(double),
                                 hs[q]->d_gl + lo * I * 3, I * 3 * sizeof(double),
                                 I * 3 * sizeof(double), S_own, hipMemcpyDeviceToDevice, h->stream));
      fast_exp(h->stream, h->d_gl_shard, h->d_gl_shard, (size_t)S_own * I_tot * 3);
      HIP_TRY(hipGetLastError());
      HIP_TRY(sync_stream(h));
    } else {
      uint8_t *bytes = nullptr, *part = nullptr;  // [S_own][I_tot] on r; [S_own][I] on q
      if ((rc = dev_alloc(&bytes, (size_t)S_own * I_tot))) return rc;
      for (int q = 0; q < n && rc == NGHMM_OK; ++q) {
        hipError_t e = hipSetDevice(hs[q]->device);
        if (e == hipSuccess) e = hipMalloc((void**)&part, (size_t)S_own * I);
        if (e == hipSuccess) {
          launch_codes_to_bytes(hs[q]->stream, hs[q]->d_codes, lo * I, S_own * I, part);
          e = hipStreamSynchronize(hs[q]->stream);
        }
        if (e == hipSuccess) e = hipSetDevice(h->device);
        if (e == hipSuccess)
          e = hipMemcpy2DAsync(bytes + (size_t)q * I, I_tot, part, I, I, S_own,
                               hipMemcpyDeviceToDevice, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (part) {
          (void)hipSetDevice(hs[q]->device);
          (void)hipFree(part);
          part = nullptr;
          (void)hipSetDevice(h->device);
        }
        if (e != hipSuccess) {
          set_error("nghmm_group_setup: %s", hipGetErrorString(e));
          rc = NGHMM_ERR_HIP;
        }
      }
      if (rc == NGHMM_OK) rc = nghmm_load_geno_site_shard_dev(h, bytes);
      (void)hipFree(bytes);
      if (rc != NGHMM_OK) return rc;
    }
  }
  return NGHMM_OK;
}

int nghmm_group_iter_em(nghmm_t** hs, int n, int freq_est, int indF_fixed, int alpha_fixed,
                        double* ind_lkl, nghmm_mstep_stats* stats) {
  g_last_error.clear();
  if (!hs || n < 1 || !hs[0] || hs[0]->g_n != n) {
    set_error("nghmm_group_iter_em: call nghmm_group_setup on these handles first");
    return NGHMM_ERR_ARG;
  }
  if (n == 1) return nghmm_iter_em(hs[0], freq_est, indF_fixed, alpha_fixed, ind_lkl, stats);
  if (freq_est & NGHMM_LD_INTENDED) {
    set_error("the intended --freq_est 2 walks the sites in order on ONE handle: not available "
              "for a group of several");
    return NGHMM_ERR_ARG;
  }
  if (freq_est != 0 && freq_est != 1) {  // as nghmm_mstep_freq (EM.cpp:212-239)
    set_error(freq_est == 2 ? "invalid allele frequencies" : "wrong MAF estimation method!");
    return freq_est == 2 ? NGHMM_ERR_FREQ_EST2 : NGHMM_ERR_ARG;
  }
  const uint64_t I = hs[0]->I;
  std::vector<nghmm_mstep_stats> st(n);
  std::vector<GroupHook> hook(n);
  // phase 1: E-step + indF/alpha M-step per rank; posteriors leave for their site owners as
  // soon as they are final
  int rc = for_each_rank(n, [&](int r) -> int {
    hook[r] = GroupHook{hs, n, r, NGHMM_OK};
    int rr = nghmm_estep_mstep(hs[r], indF_fixed, alpha_fixed, ind_lkl ? ind_lkl + (size_t)r * I : nullptr,
                               &st[r], freq_est ? group_after_estep : nullptr, &hook[r]);
    if (rr == NGHMM_OK) rr = hook[r].rc;
    if (rr == NGHMM_OK && freq_est && hipStreamSynchronize(hs[r]->g_xstream) != hipSuccess)
      rr = NGHMM_ERR_HIP;
    return rr;
  });
  if (rc != NGHMM_OK) return rc;
  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    for (int r = 0; r < n; ++r) {
      stats->rounds = st[r].rounds > stats->rounds ? st[r].rounds : stats->rounds;
      stats->points += st[r].points;
      stats->ref_forward_calls += st[r].ref_forward_calls;
      stats->ind_rounds += st[r].ind_rounds;
    }
  }
  if (!freq_est) return NGHMM_OK;
  return group_freq_phases(hs, n);
}

// The allele-frequency step alone, from the posteriors the handles hold (all zero before the
// first E-step: --freq e, parse_args.cpp:312-318).
int nghmm_group_mstep_freq(nghmm_t** hs, int n, int freq_est) {
  g_last_error.clear();
  if (!hs || n < 1 || !hs[0] || hs[0]->g_n != n) {
    set_error("nghmm_group_mstep_freq: call nghmm_group_setup on these handles first");
    return NGHMM_ERR_ARG;
  }
  if (n == 1) return nghmm_mstep_freq(hs[0], freq_est);
  if (freq_est == 0) return NGHMM_OK;
  if (freq_est != 1) {
    set_error(freq_est == 2 ? "invalid allele frequencies" : "wrong MAF estimation method!");
    return freq_est == 2 ? NGHMM_ERR_FREQ_EST2 : NGHMM_ERR_ARG;
  }
  std::vector<GroupHook> hook(n);
  int rc = for_each_rank(n, [&](int r) -> int {
    hook[r] = GroupHook{hs, n, r, NGHMM_OK};
    group_after_estep(&hook[r]);
    if (hook[r].rc != NGHMM_OK) return hook[r].rc;
    return hipStreamSynchronize(hs[r]->g_xstream) == hipSuccess ? NGHMM_OK : NGHMM_ERR_HIP;
  });
  if (rc != NGHMM_OK) return rc;
  return group_freq_phases(hs, n);
}

// ---- one process, several GPUs, fast mode: a CHAIN of site shards ----
// (include/nghmm.h; between processes ngsf-hmm_amd/distributed.py does the same over RCCL.)
// The all-gather of nghmm_site_shard_setup among the handles of one process: every handle runs
// on a host thread of its own; at an exchange it waits for its stream (its part of the send
// buffers is complete), meets the others at a barrier, copies every handle's part into its own
// receive buffer -- direct device-to-device copies, over the GPU pair's xGMI link where the
// devices differ --, waits for the copies and meets the others again (nobody rewrites its part
// before everyone has read it).  A handle that fails aborts the barrier for the others.
struct ChainCtx {
  std::vector<nghmm_t*> hs;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  bool aborted = false;
  int refs = 0;
  // false: somebody aborted
  bool wait() {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return false;
    const uint64_t gen = generation;
    if (++arrived == (int)hs.size()) {
      arrived = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    cv.wait(lk, [&] { return generation != gen || aborted; });
    return !aborted;
  }
  void abort() {
    std::lock_guard<std::mutex> lk(mu);
    aborted = true;
    cv.notify_all();
  }
  void reset() {
    std::lock_guard<std::mutex> lk(mu);
    aborted = false;
    arrived = 0;
  }
};

namespace {

int chain_allgather(void* user, uint64_t n_bytes) {
  nghmm_t* h = static_cast<nghmm_t*>(user);
  ChainCtx* cx = h->chain;
  if (!cx) return 1;
  const int n = (int)cx->hs.size();
  for (int q = 0; q < n; ++q)
    if (!cx->hs[q]) return 1;  // (a dissolved chain: chain_release detaches every member)
  bool ok = hipStreamSynchronize(h->stream) == hipSuccess;
  if (!ok) cx->abort();
  if (!cx->wait()) return 1;
  for (int q = 0; q < n && ok; ++q) {
    nghmm_t* o = cx->hs[q];
    char* dst = reinterpret_cast<char*>(h->c_recv) + (size_t)q * n_bytes;
    const hipError_t e = o->device == h->device
                             ? hipMemcpyAsync(dst, o->c_send, n_bytes, hipMemcpyDeviceToDevice, h->stream)
                             : hipMemcpyPeerAsync(dst, h->device, o->c_send, o->device, n_bytes, h->stream);
    ok = e == hipSuccess;
  }
  if (ok) ok = hipStreamSynchronize(h->stream) == hipSuccess;
  if (!ok) cx->abort();
  return cx->wait() && ok ? 0 : 1;
}

}  // namespace

void capi::chain_release(nghmm_t* h) {
  if (!h->chain) return;
  // One member leaving DISSOLVES the chain (include/nghmm.h): every remaining member goes back
  // to being a plain handle over its own sites -- no all-gather installed, no exchange buffers,
  // no context whose barrier could never fill again.  (Leaving the survivors attached made a
  // direct nghmm_iter_em / nghmm_estep / nghmm_lkl_batch on one of them wait in ChainCtx::wait
  // for a member that no longer exists.)
  ChainCtx* cx = h->chain;
  cx->abort();
  std::vector<nghmm_t*> members;
  {
    std::lock_guard<std::mutex> lk(cx->mu);
    members = cx->hs;
    for (auto& m : cx->hs) m = nullptr;
  }
  int dev_before = -1;
  (void)hipGetDevice(&dev_before);
  for (nghmm_t* m : members) {
    if (!m || m->chain != cx) continue;
    m->chain = nullptr;
    m->fast.shard.world = 1;
    m->fast.shard.rank = 0;
    m->fast.shard.allgather = nullptr;
    m->fast.shard.user = nullptr;
    m->fast.shard.send = m->fast.shard.recv = nullptr;
    m->fast.shard.edges_from_round = false;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->c_send) (void)hipFree(m->c_send);
    if (m->c_recv) (void)hipFree(m->c_recv);
    m->c_send = m->c_recv = nullptr;
  }
  if (dev_before >= 0) (void)hipSetDevice(dev_before);
  delete cx;
}

namespace {

bool is_chain(nghmm_t** hs, int n) {
  if (!hs || n < 1 || !hs[0]) return false;
  if (n == 1) return hs[0]->chain == nullptr || hs[0]->chain->hs.size() == 1;
  ChainCtx* cx = hs[0]->chain;
  if (!cx || (int)cx->hs.size() != n) return false;
  for (int r = 0; r < n; ++r)
    if (hs[r] != cx->hs[r]) return false;
  return true;
}

}  // namespace

int nghmm_chain_setup(nghmm_t** hs, int n) {
  g_last_error.clear();
  if (!hs || n < 1) return NGHMM_ERR_ARG;
  for (int r = 0; r < n; ++r) {
    nghmm_t* h = hs[r];
    if (!h || h->I != hs[0]->I || h->mode != hs[0]->mode || h->packed != hs[0]->packed || h->parent ||
        h->n_replicas.load() > 0 || h->g_n > 1 || h->I_tot != h->I) {
      set_error("nghmm_chain_setup: plain handles of the same individuals, mode and packing are needed");
      return NGHMM_ERR_ARG;
    }
  }
  if (n > 1 && hs[0]->mode != NGHMM_MODE_FAST) {
    set_error("nghmm_chain_setup: site shards are a fast-mode layout");
    return NGHMM_ERR_ARG;
  }
  for (int r = 0; r < n; ++r) hs[r]->batch.set_max_threads(n > 1 ? 1 : 64);  // (as in nghmm_group_setup)
  for (int r = 0; r < n; ++r) chain_release(hs[r]);
  if (n == 1) return NGHMM_OK;
  int rc;
  for (int r = 0; r < n; ++r)
    for (int q = 0; q < n; ++q)
      if (hs[r]->device != hs[q]->device) {
        int can = 0;
        HIP_TRY(hipDeviceCanAccessPeer(&can, hs[r]->device, hs[q]->device));
        if (!can) {
          set_error("nghmm_chain_setup: device %d cannot access device %d", hs[r]->device, hs[q]->device);
          return NGHMM_ERR_HIP;
        }
        HIP_TRY(hipSetDevice(hs[r]->device));
        const hipError_t e = hipDeviceEnablePeerAccess(hs[q]->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_TRY(e);
        (void)hipGetLastError();
      }
  ChainCtx* cx = new (std::nothrow) ChainCtx;
  if (!cx) return NGHMM_ERR_NOMEM;
  cx->hs.assign(hs, hs + n);
  cx->refs = n;
  for (int r = 0; r < n; ++r) hs[r]->chain = cx;
  for (int r = 0; r < n; ++r) {
    nghmm_t* h = hs[r];
    const uint64_t bytes = nghmm_site_shard_bytes(h);
    if ((rc = use_device(h)) || (rc = dev_alloc(&h->c_send, (size_t)(bytes / sizeof(double)))) ||
        (rc = dev_alloc(&h->c_recv, (size_t)(bytes / sizeof(double)) * n)) ||
        (rc = nghmm_site_shard_setup(h, r, n, h->c_send, h->c_recv, bytes, chain_allgather, h))) {
      const std::string msg = g_last_error;
      for (int q = 0; q < n; ++q) chain_release(hs[q]);
      g_last_error = msg;
      return rc;
    }
  }
  return NGHMM_OK;
}

int nghmm_chain_iter_em(nghmm_t** hs, int n, int freq_est, int indF_fixed, int alpha_fixed,
                        double* ind_lkl, nghmm_mstep_stats* stats) {
  g_last_error.clear();
  if (!is_chain(hs, n)) {
    set_error("nghmm_chain_iter_em: call nghmm_chain_setup on these handles first");
    return NGHMM_ERR_ARG;
  }
  if (n == 1) return nghmm_iter_em(hs[0], freq_est, indF_fixed, alpha_fixed, ind_lkl, stats);
  if (freq_est & NGHMM_LD_INTENDED) {
    set_error("the intended --freq_est 2 walks the sites in order on ONE handle: not available "
              "for a chain of several");
    return NGHMM_ERR_ARG;
  }
  ChainCtx* cx = hs[0]->chain;
  cx->reset();
  std::vector<nghmm_mstep_stats> st(n);
  const int rc = for_each_rank(n, [&](int r) -> int {
    // every handle computes the chain's log-likelihoods: the first one's go to the caller
    const int rr = nghmm_iter_em(hs[r], freq_est, indF_fixed, alpha_fixed, r == 0 ? ind_lkl : nullptr,
                                 &st[r]);
    if (rr != NGHMM_OK) cx->abort();
    return rr;
  });
  if (stats) *stats = st[0];
  return rc;
}

int nghmm_chain_mstep_freq(nghmm_t** hs, int n, int freq_est) {
  g_last_error.clear();
  if (!is_chain(hs, n)) {
    set_error("nghmm_chain_mstep_freq: call nghmm_chain_setup on these handles first");
    return NGHMM_ERR_ARG;
  }
  // every handle has all individuals of its own sites: nothing to exchange
  return for_each_rank(n, [&](int r) -> int { return nghmm_mstep_freq(hs[r], freq_est); });
}

int nghmm_chain_viterbi(nghmm_t** hs, int n, uint8_t* path) {
  g_last_error.clear();
  if (!is_chain(hs, n) || !path) {
    set_error("nghmm_chain_viterbi: call nghmm_chain_setup on these handles first");
    return NGHMM_ERR_ARG;
  }
  if (n == 1) return nghmm_viterbi(hs[0], path);
  const uint64_t I = hs[0]->I;
  uint64_t S_tot = 0;
  for (int r = 0; r < n; ++r) S_tot += hs[r]->S;
  int rc;
  std::vector<double> scores((size_t)I * 2);
  for (int r = 0; r < n; ++r)
    if ((rc = nghmm_viterbi_shard_forward(hs[r], r ? scores.data() : nullptr, scores.data()))) return rc;
  std::vector<uint8_t> state(I), part;
  uint64_t hi = S_tot;
  for (int r = n - 1; r >= 0; --r) {
    const uint64_t S = hs[r]->S, lo = hi - S;
    part.resize((size_t)I * S);
    if ((rc = nghmm_viterbi_shard_back(hs[r], r == n - 1 ? nullptr : state.data(), state.data(),
                                       part.data())))
      return rc;
    for (uint64_t i = 0; i < I; ++i) std::memcpy(path + i * S_tot + lo, part.data() + i * S, S);
    hi = lo;
  }
  return NGHMM_OK;
}


// capi_load.hip -- the loaders: whole matrices and blocks of sites, from host or device buffers, raw likelihoods
// prepared on the device, called genotypes packed on the way in
// (implementation of include/nghmm.h; capi_internal.hpp has the handle and the shared helpers.)
#include "capi_internal.hpp"

static int after_gl_load(nghmm_t* h) {
  h->loaded = true;
  h->loading = false;
  h->marg_valid = false;  // nothing derived from earlier data survives a (re)load
  h->tmp_is_posteriors = false;
  if (h->packed) {
    // the value the data's uniform cells carry becomes row 3 of the class table
    unsigned long long bits = ~0ull;
    HIP_TRY(hipMemcpyAsync(&bits, h->d_uniform, sizeof bits, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(sync_stream(h));
    if (bits != ~0ull) {
      double u;
      std::memcpy(&u, &bits, sizeof u);
      const double row[3] = {u, u, u};
      HIP_TRY(hipMemcpyAsync(h->d_cls_log + 9, row, sizeof row, hipMemcpyHostToDevice, h->stream));
      HIP_TRY(sync_stream(h));
    }
  }
  if (h->mode == NGHMM_MODE_FAST) {
    if (!fast_load(h->fast, h->stream, own_gl(h), h->d_pos)) return NGHMM_ERR_HIP;
    HIP_TRY(sync_stream(h));
  }
  return NGHMM_OK;
}

static int ensure_stage(nghmm_t* h, size_t doubles) {
  if (doubles <= h->stage_cap) return NGHMM_OK;
  if (h->d_stage) (void)hipFree(h->d_stage);
  h->d_stage = nullptr;
  h->stage_cap = 0;
  int rc;
  if ((rc = dev_alloc(&h->d_stage, doubles))) return rc;
  h->stage_cap = doubles;
  return NGHMM_OK;
}

static int ensure_stage8(nghmm_t* h, size_t bytes) {
  if (bytes <= h->stage8_cap) return NGHMM_OK;
  if (h->d_stage8) (void)hipFree(h->d_stage8);
  h->d_stage8 = nullptr;
  h->stage8_cap = 0;
  int rc;
  if ((rc = dev_alloc(&h->d_stage8, bytes))) return rc;
  h->stage8_cap = bytes;
  return NGHMM_OK;
}

// flags a chunk loader looks at after its kernels
static int check_load_flags(nghmm_t* h, bool check_nan) {
  int f[NFLAGS];
  HIP_TRY(hipMemcpyAsync(f, h->d_flags, sizeof f, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(sync_stream(h));
  if (check_nan && f[FLAG_NAN]) {
    set_error("NaN found! Is the file format correct?");
    return NGHMM_ERR_NAN;
  }
  if (f[FLAG_BAD_GENO]) {
    set_error("wrong GENO file format. Genotypes must be coded as {-1,0,1,2} !");
    return NGHMM_ERR_ARG;
  }
  if (f[FLAG_NOT_PACKABLE]) {
    set_error("a cell is not a called genotype (one-hot or uniform likelihoods): a packed handle "
              "(NGHMM_GENO_PACKED) needs --call_geno or called-genotype input");
    return NGHMM_ERR_NOT_PACKABLE;
  }
  return NGHMM_OK;
}

// One chunk of sites [site_begin, site_begin + n_sites) from d_src (device; dense [n][I][3]):
// optional preparation, then into d_gl or, packed, into the codes.  d_src may be the staging
// buffer or the caller's; `prepare` works in place on a copy when the handle is packed.
static int ingest_chunk(nghmm_t* h, uint64_t site_begin, uint64_t n_sites, const double* d_src,
                        bool src_is_scratch, bool prepare, int space, int call_geno,
                        int check_nan) {
  int rc;
  const uint64_t n_cells = n_sites * h->I, cell0 = site_begin * h->I;
  if ((rc = clear_flags(h))) return rc;
  if (!h->packed) {
    double* dst = h->d_gl + cell0 * 3;
    if (d_src != dst)
      HIP_TRY(hipMemcpyAsync(dst, d_src, n_cells * 3 * sizeof(double), hipMemcpyDeviceToDevice,
                             h->stream));
    if (prepare) launch_prepare_gl(h->stream, dst, n_cells, space, call_geno, h->d_flags);
  } else {
    const double* cells = d_src;
    if (prepare) {
      if (!src_is_scratch) {  // never modify the caller's buffer
        if ((rc = ensure_stage(h, n_cells * 3))) return rc;
        HIP_TRY(hipMemcpyAsync(h->d_stage, d_src, n_cells * 3 * sizeof(double),
                               hipMemcpyDeviceToDevice, h->stream));
        cells = h->d_stage;
      }
      launch_prepare_gl(h->stream, const_cast<double*>(cells), n_cells, space, call_geno, h->d_flags);
    }
    launch_pack_cells(h->stream, cells, n_cells, cell0, h->d_cls_log, h->d_codes, h->d_uniform,
                      h->d_flags);
  }
  HIP_TRY(hipGetLastError());
  return check_load_flags(h, check_nan != 0);
}

// sites per chunk when a whole-matrix loader feeds a packed handle through the staging buffer
static uint64_t stage_sites(const nghmm_t* h) {
  uint64_t n = (256ull << 20) / (h->I * 24);
  if (n < 1) n = 1;
  return n < h->S ? n : h->S;
}

int nghmm_load_begin(nghmm_t* h, const double* pos) {
  g_last_error.clear();
  if (!h || !pos) return NGHMM_ERR_ARG;
  if (h->parent || h->n_replicas.load() > 0) {
    set_error("a replica shares its parent's data: load into the parent, before creating replicas");
    return NGHMM_ERR_ARG;
  }
  int rc;
  if ((rc = use_device(h))) return rc;
  HIP_TRY(hipMemcpyAsync(h->d_pos, pos, h->S * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (h->packed) {
    HIP_TRY(hipMemsetAsync(h->d_codes, 0, ((size_t)h->I * h->S / 16 + 2) * sizeof(uint32_t), h->stream));
    HIP_TRY(hipMemsetAsync(h->d_uniform, 0xff, sizeof(unsigned long long), h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_cls_log, h->h_cls_proto, sizeof h->h_cls_proto,
                           hipMemcpyHostToDevice, h->stream));  // an earlier load's uniform value
  }
  HIP_TRY(sync_stream(h));
  h->loaded = false;
  h->loading = true;
  h->load_cover.clear();
  return NGHMM_OK;
}

static int load_begin_dev(nghmm_t* h, const double* d_pos) {
  if (h->parent || h->n_replicas.load() > 0) {
    set_error("a replica shares its parent's data: load into the parent, before creating replicas");
    return NGHMM_ERR_ARG;
  }
  // The caller's device buffers were written on streams this library knows nothing of (torch's,
  // say), and the handle's own stream is a non-blocking one that waits for none of them: a load
  // waits for the whole device before it reads them.  (Round 5: a distance table still being
  // computed on torch's stream was copied too early -- an intermittent, silent wrong data set.)
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyAsync(h->d_pos, d_pos, h->S * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (h->packed) {
    HIP_TRY(hipMemsetAsync(h->d_codes, 0, ((size_t)h->I * h->S / 16 + 2) * sizeof(uint32_t), h->stream));
    HIP_TRY(hipMemsetAsync(h->d_uniform, 0xff, sizeof(unsigned long long), h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_cls_log, h->h_cls_proto, sizeof h->h_cls_proto,
                           hipMemcpyHostToDevice, h->stream));
  }
  h->loaded = false;
  h->loading = true;
  h->load_cover.clear();
  return NGHMM_OK;
}

int nghmm_load_begin_dev(nghmm_t* h, const double* d_pos) {
  g_last_error.clear();
  if (!h || !d_pos) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  if ((rc = load_begin_dev(h, d_pos))) return rc;
  HIP_TRY(sync_stream(h));
  return NGHMM_OK;
}

// chunked loading: claim the sites [b, b + n) of the current load; overlap is an error
static int claim_sites(nghmm_t* h, uint64_t b, uint64_t n) {
  const uint64_t e = b + n;
  auto it = h->load_cover.upper_bound(b);  // first run that begins after b
  if (it != h->load_cover.begin()) {
    auto prev = std::prev(it);
    if (prev->second > b) {
      set_error("chunk loader: sites [%llu, %llu) overlap [%llu, %llu), which this load has "
                "already received: every site exactly once",
                (unsigned long long)b, (unsigned long long)e, (unsigned long long)prev->first,
                (unsigned long long)prev->second);
      return NGHMM_ERR_ARG;
    }
  }
  if (it != h->load_cover.end() && it->first < e) {
    set_error("chunk loader: sites [%llu, %llu) overlap [%llu, %llu), which this load has "
              "already received: every site exactly once",
              (unsigned long long)b, (unsigned long long)e, (unsigned long long)it->first,
              (unsigned long long)it->second);
    return NGHMM_ERR_ARG;
  }
  // insert, merging with the neighbours it touches
  uint64_t nb = b, ne = e;
  if (it != h->load_cover.begin()) {
    auto prev = std::prev(it);
    if (prev->second == b) {
      nb = prev->first;
      h->load_cover.erase(prev);
    }
  }
  if (it != h->load_cover.end() && it->first == e) {
    ne = it->second;
    h->load_cover.erase(it);
  }
  h->load_cover[nb] = ne;
  return NGHMM_OK;
}

// a chunk that failed half way has left cells behind (a packed handle ORs codes in): the load
// is over, the caller starts again with nghmm_load_begin
static int fail_load(nghmm_t* h, int rc) {
  if (rc != NGHMM_OK) h->loading = false;
  return rc;
}

static int load_sites_impl(nghmm_t* h, uint64_t site_begin, uint64_t n_sites, const double* src,
                           bool src_on_device, bool prepare, int space, int call_geno,
                           int check_nan) {
  if (!h || !h->loading || !src || site_begin + n_sites > h->S ||
      space < NGHMM_GL_LOG || space > NGHMM_GL_NORMAL_TEXT) {
    set_error("chunk loader: bad argument, or nghmm_load_begin has not been called");
    return NGHMM_ERR_ARG;
  }
  if (n_sites == 0) return NGHMM_OK;
  int rc;
  if ((rc = use_device(h))) return rc;
  if ((rc = claim_sites(h, site_begin, n_sites))) return rc;
  const uint64_t n_cells = n_sites * h->I;
  if (src_on_device) {
    if (hipDeviceSynchronize() != hipSuccess) return fail_load(h, NGHMM_ERR_HIP);  // (as load_begin_dev)
    return fail_load(h, ingest_chunk(h, site_begin, n_sites, src, false, prepare, space,
                                     call_geno, check_nan));
  }
  double* dst = h->packed ? nullptr : h->d_gl + site_begin * h->I * 3;
  if (h->packed) {
    if ((rc = ensure_stage(h, n_cells * 3))) return fail_load(h, rc);
    dst = h->d_stage;
  }
  if (hipMemcpyAsync(dst, src, n_cells * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream) !=
      hipSuccess) {
    set_error("chunk loader: copy to the device failed");
    return fail_load(h, NGHMM_ERR_HIP);
  }
  return fail_load(h, ingest_chunk(h, site_begin, n_sites, dst, true, prepare, space, call_geno,
                                   check_nan));
}

int nghmm_load_gl_raw_sites(nghmm_t* h, uint64_t site_begin, uint64_t n_sites, const double* gl_raw,
                            int space, int call_geno, int check_nan) {
  g_last_error.clear();
  return load_sites_impl(h, site_begin, n_sites, gl_raw, false, true, space, call_geno, check_nan);
}

int nghmm_load_gl_raw_sites_dev(nghmm_t* h, uint64_t site_begin, uint64_t n_sites,
                                const double* d_gl_raw, int space, int call_geno, int check_nan) {
  g_last_error.clear();
  return load_sites_impl(h, site_begin, n_sites, d_gl_raw, true, true, space, call_geno, check_nan);
}

int nghmm_load_geno_sites(nghmm_t* h, uint64_t site_begin, uint64_t n_sites, const int8_t* geno) {
  g_last_error.clear();
  if (!h || !h->loading || !geno || site_begin + n_sites > h->S) {
    set_error("nghmm_load_geno_sites: bad argument, or nghmm_load_begin has not been called");
    return NGHMM_ERR_ARG;
  }
  if (n_sites == 0) return NGHMM_OK;
  int rc;
  if ((rc = use_device(h))) return rc;
  if ((rc = claim_sites(h, site_begin, n_sites))) return rc;
  // (every return below goes through fail_load: a failure after claim_sites gives the sites back)
  auto body = [&]() -> int {
    const uint64_t n_cells = n_sites * h->I, cell0 = site_begin * h->I;
    int rc;
    if ((rc = ensure_stage8(h, n_cells))) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_stage8, geno, n_cells, hipMemcpyHostToDevice, h->stream));
    if ((rc = clear_flags(h))) return rc;
    if (h->packed) {
      // the reader's missing genotype is log(1/3) x 3 (read_data.cpp:94), prepared: row 3 of the
      // class table as nghmm_create left it; a data set has ONE uniform value
      unsigned long long cur = ~0ull, want = 0;
      double u = 0;
      HIP_TRY(hipMemcpyAsync(&cur, h->d_uniform, sizeof cur, hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(hipMemcpyAsync(&u, h->d_cls_log + 9, sizeof u, hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(sync_stream(h));
      std::memcpy(&want, &u, sizeof want);
      if (cur == ~0ull)
        HIP_TRY(hipMemcpyAsync(h->d_uniform, &want, sizeof want, hipMemcpyHostToDevice, h->stream));
      else if (cur != want) {
        set_error("nghmm_load_geno_sites: mixed with likelihood chunks whose uniform cells differ");
        return NGHMM_ERR_ARG;
      }
      launch_pack_geno(h->stream, h->d_stage8, n_cells, cell0, h->d_codes, h->d_flags);
    } else {
      double* dst = h->d_gl + cell0 * 3;
      launch_expand_geno(h->stream, h->d_stage8, n_cells, std::log((double)1 / 3), dst, h->d_flags);
      launch_prepare_gl(h->stream, dst, n_cells, NGHMM_GL_LOG, 0, h->d_flags);
    }
    HIP_TRY(hipGetLastError());
    return check_load_flags(h, false);
  };
  return fail_load(h, body());
}

int nghmm_load_end(nghmm_t* h) {
  g_last_error.clear();
  if (!h || !h->loading) {
    set_error("nghmm_load_end: no load in progress (nghmm_load_begin not called, or a chunk failed)");
    return NGHMM_ERR_ARG;
  }
  int rc;
  if ((rc = use_device(h))) return rc;
  // every site exactly once: one run [0, S)
  if (h->load_cover.size() != 1 || h->load_cover.begin()->first != 0 ||
      h->load_cover.begin()->second != h->S) {
    uint64_t got = 0;
    for (const auto& r : h->load_cover) got += r.second - r.first;
    set_error("nghmm_load_end: %llu of %llu sites have been loaded", (unsigned long long)got,
              (unsigned long long)h->S);
    h->loading = false;
    return NGHMM_ERR_ARG;
  }
  return after_gl_load(h);
}

// whole-matrix loaders = one begin, chunks, end
static int load_whole(nghmm_t* h, const double* gl, bool on_device, bool prepare, int space,
                      int call_geno, int check_nan) {
  int rc;
  // a dense handle takes the matrix in one piece; a packed one through the staging buffer
  const uint64_t step = (h->packed && !(on_device && !prepare)) ? stage_sites(h) : h->S;
  for (uint64_t s0 = 0; s0 < h->S; s0 += step) {
    const uint64_t ns = (h->S - s0) < step ? (h->S - s0) : step;
    if ((rc = load_sites_impl(h, s0, ns, gl + s0 * h->I * 3, on_device, prepare, space, call_geno,
                              check_nan)))
      return rc;
  }
  return after_gl_load(h);
}

int nghmm_load_gl(nghmm_t* h, const double* gl, const double* pos) {
  g_last_error.clear();
  if (!h || !gl || !pos) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = nghmm_load_begin(h, pos))) return rc;
  return load_whole(h, gl, false, false, NGHMM_GL_LOG, 0, 0);
}

int nghmm_load_gl_raw(nghmm_t* h, const double* gl_raw, int space, int call_geno, int check_nan,
                      const double* pos) {
  g_last_error.clear();
  if (!h || !gl_raw || !pos || space < NGHMM_GL_LOG || space > NGHMM_GL_NORMAL_TEXT)
    return NGHMM_ERR_ARG;
  int rc;
  if ((rc = nghmm_load_begin(h, pos))) return rc;
  return load_whole(h, gl_raw, false, true, space, call_geno, check_nan);
}

int nghmm_get_gl(nghmm_t* h, double* gl) {
  g_last_error.clear();
  if (!h || !h->loaded || !gl) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  const size_t cells = (size_t)h->I * h->S;
  const double* src = h->d_gl;
  if (h->packed) {  // test / debug aid: unpack through the staging buffer
    if ((rc = ensure_stage(h, cells * 3))) return rc;
    launch_unpack_cells(h->stream, own_gl(h), cells, h->d_stage);
    HIP_TRY(hipGetLastError());
    src = h->d_stage;
  }
  HIP_TRY(hipMemcpyAsync(gl, src, cells * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(sync_stream(h));
  return NGHMM_OK;
}

int nghmm_load_gl_device(nghmm_t* h, const double* d_gl, const double* d_pos) {
  g_last_error.clear();
  if (!h || !d_gl || !d_pos) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  if ((rc = load_begin_dev(h, d_pos))) return rc;
  return load_whole(h, d_gl, true, false, NGHMM_GL_LOG, 0, 0);
}


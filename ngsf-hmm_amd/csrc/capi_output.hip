// capi_output.hip -- read-back and output formatting: posteriors, the .ibd posterior lines and the .geno genotype
// posteriors formatted on the device, emissions
// (implementation of include/nghmm.h; capi_internal.hpp has the handle and the shared helpers.)
#include "capi_internal.hpp"

// [I][S] posteriors of the last E-step in d_tmp (transposed once per E-step)
static int posteriors_ind_major(nghmm_t* h) {
  int rc;
  if ((rc = ensure_tmp(h))) return rc;
  if (h->tmp_is_posteriors) return NGHMM_OK;
  if ((rc = ensure_marg(h))) return rc;
  launch_transpose_f64(h->stream, h->d_marg, h->d_tmp, h->S, h->I);
  HIP_TRY(hipGetLastError());
  h->tmp_is_posteriors = true;
  return NGHMM_OK;
}

int nghmm_get_posteriors(nghmm_t* h, double* marg_ibd) {
  g_last_error.clear();
  if (!h || !h->loaded || !marg_ibd) return NGHMM_ERR_ARG;  // zeros before the first E-step
  int rc;
  if ((rc = use_device(h))) return rc;
  if ((rc = posteriors_ind_major(h))) return rc;
  HIP_TRY(hipMemcpyAsync(marg_ibd, h->d_tmp, (size_t)h->I * h->S * sizeof(double),
                         hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(sync_stream(h));
  return NGHMM_OK;
}

int nghmm_format_posteriors(nghmm_t* h, uint64_t ind_begin, uint64_t n_ind, char* out) {
  g_last_error.clear();
  if (!h || !h->loaded || !out || ind_begin + n_ind > h->I) return NGHMM_ERR_ARG;
  if (n_ind == 0) return NGHMM_OK;
  int rc;
  if ((rc = use_device(h))) return rc;
  if ((rc = posteriors_ind_major(h))) return rc;
  const size_t bytes = (size_t)n_ind * 9 * h->S;
  if (bytes > h->text_cap) {
    if (h->d_text) (void)hipFree(h->d_text);
    h->d_text = nullptr;
    h->text_cap = 0;
    if ((rc = dev_alloc(&h->d_text, bytes))) return rc;
    h->text_cap = bytes;
  }
  if ((rc = clear_flags(h))) return rc;
  launch_format_fixed6(h->stream, h->d_tmp + ind_begin * h->S, n_ind, h->S, h->d_text, h->d_flags);
  HIP_TRY(hipGetLastError());
  int bad = 0;
  HIP_TRY(hipMemcpyAsync(&bad, h->d_flags, sizeof bad, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(out, h->d_text, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(sync_stream(h));
  if (bad) {
    set_error("nghmm_format_posteriors: a posterior outside [0, 1]");
    return NGHMM_ERR_ARG;
  }
  return NGHMM_OK;
}

int nghmm_format_fixed6(nghmm_t* h, const double* values, uint64_t rows, uint64_t cols, char* out) {
  g_last_error.clear();
  if (!h || !values || !out) return NGHMM_ERR_ARG;
  if (rows == 0 || cols == 0) return NGHMM_OK;
  int rc;
  if ((rc = use_device(h))) return rc;
  const size_t n = (size_t)rows * cols;
  double* d_in = nullptr;
  char* d_out = nullptr;
  if ((rc = dev_alloc(&d_in, n))) return rc;
  if ((rc = dev_alloc(&d_out, n * 9))) {
    (void)hipFree(d_in);
    return rc;
  }
  int bad = 0;
  hipError_t e = hipMemcpyAsync(d_in, values, n * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipMemsetAsync(h->d_flags, 0, NFLAGS * sizeof(int), h->stream);
  if (e == hipSuccess) {
    launch_format_fixed6(h->stream, d_in, rows, cols, d_out, h->d_flags);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&bad, h->d_flags, sizeof bad, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n * 9, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = sync_stream(h);
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  if (e != hipSuccess) {
    set_error("nghmm_format_fixed6: %s", hipGetErrorString(e));
    return NGHMM_ERR_HIP;
  }
  if (bad) {
    set_error("nghmm_format_fixed6: a value outside [0, 1]");
    return NGHMM_ERR_ARG;
  }
  return NGHMM_OK;
}

int nghmm_geno_posteriors(nghmm_t* h, uint64_t site_begin, uint64_t n_sites, double* out) {
  g_last_error.clear();
  if (!h || !h->loaded || !out || site_begin + n_sites > h->S) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  if (!h->d_path_sites) {
    // not decoded yet: the reference's path[][] is still all zeros then (an intermediate
    // print_iter, EM.cpp:60-62)
    const size_t blocked = viterbi_blocked_bytes(h->S, h->I);
    if ((rc = dev_alloc(&h->d_path_sites, blocked))) return rc;
    HIP_TRY(hipMemsetAsync(h->d_path_sites, 0, blocked, h->stream));
  }
  const size_t n = (size_t)n_sites * h->I * 3;
  if (n > h->geno_cap) {
    if (h->d_geno) (void)hipFree(h->d_geno);
    h->d_geno = nullptr;
    h->geno_cap = 0;
    if ((rc = dev_alloc(&h->d_geno, n))) return rc;
    h->geno_cap = n;
  }
  launch_geno_post_exact(h->stream, own_gl(h), h->d_freq, h->d_path_sites, h->I, site_begin, n_sites,
                         h->d_geno);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, h->d_geno, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(sync_stream(h));
  return NGHMM_OK;
}

int nghmm_get_emissions(nghmm_t* h, double* e_prob) {
  g_last_error.clear();
  if (!h || !e_prob) return NGHMM_ERR_ARG;
  int rc;
  if ((rc = use_device(h))) return rc;
  if ((rc = ensure_tmp(h))) return rc;
  if (h->mode == NGHMM_MODE_FAST) {
    if ((rc = ensure_emissions(h))) return rc;
    h->tmp_is_posteriors = false;
    if (!fast_export_emissions(h->fast, h->stream, h->d_tmp)) return NGHMM_ERR_HIP;
  } else {
    h->tmp_is_posteriors = false;
    launch_transpose_pairs_f64(h->stream, h->d_eprob, h->d_tmp, h->S, h->I);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(e_prob, h->d_tmp, (size_t)h->I * h->S * 2 * sizeof(double),
                         hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(sync_stream(h));
  return NGHMM_OK;
}


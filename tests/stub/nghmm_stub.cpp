// nghmm_stub.cpp -- a CPU stand-in for libnghmm.so, TEST INFRASTRUCTURE ONLY.
//
// It implements the entry points of include/nghmm.h that the C++ host (csrc/host/ngsF-HMM.cpp)
// calls, WITHOUT computing anything: it checks the arguments the way the library does, touches
// every byte of every buffer it is handed (so that AddressSanitizer sees the host's buffer
// arithmetic: block readers, column splitting for --n_gpus, output batching) and returns
// deterministic filler.  tests/test_cli_cpu.py builds the host against it with
// -fsanitize=address,undefined and runs it on every input type.  Nothing here is a fallback:
// the shipped binary links the real library, which has no CPU path.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nghmm.h"

struct nghmm_handle {
  uint64_t I, S;
  int mode;
  bool packed, loading = false, loaded = false;
  std::vector<double> indF, alpha, freq, pos;
  std::vector<uint8_t> seen;  // per site: loaded exactly once
  nghmm_handle* parent = nullptr;
  int replicas = 0, g_n = 0;
  uint64_t checksum = 0;
};

static thread_local std::string g_err;
// position-weighted byte sum: the same data gives the same total however the host cuts it into
// blocks, a byte in another place does not
static uint64_t touch(const void* p, size_t n, uint64_t at = 0) {
  const unsigned char* c = static_cast<const unsigned char*>(p);
  uint64_t s = 0;
  for (size_t k = 0; k < n; ++k) s += c[k] * (at + k + 1) * 0x9e3779b97f4a7c15ull;
  return s;
}

extern "C" {
const char* nghmm_last_error(void) { return g_err.c_str(); }
const char* nghmm_strerror(int code) { return code == 0 ? "ok" : "stub error"; }
int nghmm_has_hip(void) { return 0; }

int nghmm_create(nghmm_t** out, uint64_t n_ind, uint64_t n_sites, int device, int mode) {
  if (!out || !n_ind || !n_sites || device < 0) return NGHMM_ERR_ARG;
  nghmm_t* h = new nghmm_handle;
  h->I = n_ind;
  h->S = n_sites;
  h->packed = (mode & NGHMM_GENO_PACKED) != 0;
  h->mode = mode & ~NGHMM_GENO_PACKED;
  h->indF.assign(n_ind, 0);
  h->alpha.assign(n_ind, 0);
  h->freq.assign(n_sites, 0);
  *out = h;
  return NGHMM_OK;
}
int nghmm_create_replica(nghmm_t** out, nghmm_t* parent) {
  if (!out || !parent || !parent->loaded || parent->parent) return NGHMM_ERR_ARG;
  nghmm_t* h = new nghmm_handle(*parent);
  h->parent = parent;
  h->replicas = 0;
  parent->replicas++;
  *out = h;
  return NGHMM_OK;
}
int nghmm_destroy(nghmm_t* h) {
  if (!h) return NGHMM_OK;
  if (h->replicas > 0) return NGHMM_ERR_ARG;
  if (h->parent) h->parent->replicas--;
  delete h;
  return NGHMM_OK;
}
int nghmm_load_begin(nghmm_t* h, const double* pos) {
  if (!h || !pos) return NGHMM_ERR_ARG;
  h->pos.assign(pos, pos + h->S);
  h->seen.assign(h->S, 0);
  h->loading = true;
  h->loaded = false;
  return NGHMM_OK;
}
static int mark(nghmm_t* h, uint64_t s0, uint64_t n) {
  if (!h || !h->loading || s0 + n > h->S) return NGHMM_ERR_ARG;
  for (uint64_t s = s0; s < s0 + n; ++s) {
    if (h->seen[s]) {
      g_err = "site loaded twice";
      return NGHMM_ERR_ARG;
    }
    h->seen[s] = 1;
  }
  return NGHMM_OK;
}
int nghmm_load_gl_raw_sites(nghmm_t* h, uint64_t s0, uint64_t n, const double* gl, int space,
                            int call_geno, int check_nan) {
  (void)check_nan;
  int rc = mark(h, s0, n);
  if (rc) return rc;
  if (!gl || space < 0 || space > 2) return NGHMM_ERR_ARG;
  h->checksum += touch(gl, n * h->I * 3 * sizeof(double), s0 * h->I * 3 * sizeof(double));
  if (h->packed && !call_geno) {  // likelihoods into a packed handle: the host must fall back
    g_err = "not a called genotype (stub)";
    return NGHMM_ERR_NOT_PACKABLE;
  }
  return NGHMM_OK;
}
int nghmm_load_geno_sites(nghmm_t* h, uint64_t s0, uint64_t n, const int8_t* geno) {
  int rc = mark(h, s0, n);
  if (rc) return rc;
  if (!geno) return NGHMM_ERR_ARG;
  for (uint64_t k = 0; k < n * h->I; ++k)
    if (geno[k] < -1 || geno[k] > 2) return NGHMM_ERR_ARG;
  h->checksum += touch(geno, n * h->I, s0 * h->I);
  return NGHMM_OK;
}
int nghmm_load_end(nghmm_t* h) {
  if (!h || !h->loading) return NGHMM_ERR_ARG;
  for (uint64_t s = 0; s < h->S; ++s)
    if (!h->seen[s]) {
      g_err = "a site was never loaded";
      return NGHMM_ERR_ARG;
    }
  h->loading = false;
  h->loaded = true;
  if (getenv("NGHMM_STUB_CHECKSUM")) fprintf(stderr, "stub checksum %016llx\n", (unsigned long long)h->checksum);
  return NGHMM_OK;
}
int nghmm_set_params(nghmm_t* h, const double* F, const double* A, const double* f) {
  if (!h) return NGHMM_ERR_ARG;
  if (F) h->indF.assign(F, F + h->I);
  if (A) h->alpha.assign(A, A + h->I);
  if (f) h->freq.assign(f, f + h->S);
  return NGHMM_OK;
}
int nghmm_get_params(nghmm_t* h, double* F, double* A, double* f) {
  if (!h) return NGHMM_ERR_ARG;
  if (F) std::memcpy(F, h->indF.data(), h->I * sizeof(double));
  if (A) std::memcpy(A, h->alpha.data(), h->I * sizeof(double));
  if (f) std::memcpy(f, h->freq.data(), h->S * sizeof(double));
  return NGHMM_OK;
}
int nghmm_emission(nghmm_t* h) { return h && h->loaded ? NGHMM_OK : NGHMM_ERR_ARG; }
int nghmm_mstep_freq(nghmm_t* h, int) { return h && h->loaded ? NGHMM_OK : NGHMM_ERR_ARG; }
// a chain of site shards (what the host's --n_gpus uses): every handle all individuals, the
// sites one range after the other
int nghmm_chain_setup(nghmm_t** hs, int n) {
  if (!hs || n < 1) return NGHMM_ERR_ARG;
  for (int r = 0; r < n; ++r) {
    if (!hs[r] || !hs[r]->loaded || hs[r]->I != hs[0]->I) return NGHMM_ERR_ARG;
    hs[r]->g_n = n;
  }
  return NGHMM_OK;
}
int nghmm_chain_mstep_freq(nghmm_t** hs, int n, int) {
  return hs && n >= 1 && hs[0]->g_n == n ? NGHMM_OK : NGHMM_ERR_ARG;
}
int nghmm_chain_iter_em(nghmm_t** hs, int n, int, int, int, double* ind_lkl, nghmm_mstep_stats* st) {
  if (!hs || n < 1 || hs[0]->g_n != n) return NGHMM_ERR_ARG;
  if (st) std::memset(st, 0, sizeof *st);
  for (int r = 0; r < n; ++r)
    for (uint64_t i = 0; i < hs[r]->I; ++i) {
      hs[r]->indF[i] = 0.25 + 0.5 * hs[r]->indF[i];  // something that converges
      if (ind_lkl && r == 0) ind_lkl[i] = -1000.0 - hs[r]->indF[i];
    }
  return NGHMM_OK;
}
int nghmm_chain_viterbi(nghmm_t** hs, int n, uint8_t* path) {
  if (!hs || n < 1 || hs[0]->g_n != n || !path) return NGHMM_ERR_ARG;
  uint64_t S = 0;
  for (int r = 0; r < n; ++r) S += hs[r]->S;
  for (uint64_t k = 0; k < hs[0]->I * S; ++k) path[k] = (uint8_t)(k & 1);
  return NGHMM_OK;
}
void* nghmm_alloc_host(uint64_t bytes) { return bytes ? std::malloc(bytes) : nullptr; }
void nghmm_free_host(void* p) { std::free(p); }
int nghmm_viterbi(nghmm_t* h, uint8_t* path) {
  if (!h || !path) return NGHMM_ERR_ARG;
  for (uint64_t k = 0; k < h->I * h->S; ++k) path[k] = (uint8_t)(k & 1);
  return NGHMM_OK;
}
int nghmm_format_posteriors(nghmm_t* h, uint64_t i0, uint64_t n, char* out) {
  if (!h || !out || i0 + n > h->I) return NGHMM_ERR_ARG;
  for (uint64_t i = 0; i < n; ++i)
    for (uint64_t s = 0; s < h->S; ++s) {
      std::memcpy(out + (i * h->S + s) * 9, "0.500000", 8);
      out[(i * h->S + s) * 9 + 8] = (s + 1 == h->S) ? '\n' : '\t';
    }
  return NGHMM_OK;
}
int nghmm_geno_posteriors(nghmm_t* h, uint64_t s0, uint64_t n, double* out) {
  if (!h || !out || s0 + n > h->S) return NGHMM_ERR_ARG;
  for (uint64_t k = 0; k < n * h->I * 3; ++k) out[k] = 1.0 / 3;
  return NGHMM_OK;
}
}  // extern "C"

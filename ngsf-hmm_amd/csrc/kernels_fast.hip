// kernels_fast.hip -- placeholder until the fast-mode kernels land: every entry
// point reports failure, so NGHMM_MODE_FAST handles cannot be created.
#include "kernels_fast.hpp"

namespace nghmm {

bool fast_create(FastState&, uint64_t, uint64_t) { return false; }
void fast_destroy(FastState&) {}
bool fast_load(FastState&, hipStream_t, const double*, const double*) { return false; }
bool fast_refresh_site_tables(FastState&, hipStream_t, const double*, int*) { return false; }
bool fast_lkl_batch(FastState&, hipStream_t, uint32_t, const uint32_t*, const double*,
                    const double*, double*, int*) { return false; }
bool fast_estep(FastState&, hipStream_t, const double*, const double*, double*, double*, int*) {
  return false;
}
bool fast_estmaf(FastState&, hipStream_t, const double*, const double*, uint64_t, uint64_t,
                 uint64_t, double*) { return false; }
bool fast_viterbi(FastState&, hipStream_t, const double*, const double*, uint8_t*, uint8_t*) {
  return false;
}
bool fast_export_emissions(FastState&, hipStream_t, double*) { return false; }

}  // namespace nghmm

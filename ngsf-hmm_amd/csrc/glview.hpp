// glview.hpp -- how a kernel sees the genotype likelihoods of a cell (site, individual).
//
// Dense: three doubles per cell, site-major [cells][3] (24 B).  Packed: called genotypes
// (--call_geno, or a called-genotype input file: shared/read_data.cpp:88-98,
// ngsF-HMM.cpp:101-117, shared/gen_func.cpp:886-914) take one of four values per cell --
// genotype 0, 1, 2 or missing -- so a cell is a 2-bit code into a 4 x 3 table of the
// prepared (normalised natural-log, or linear) likelihoods: 0.25 B per cell, 16 cells per
// 32-bit word, cell k at bits 2 (k & 15) of word k >> 4.
#pragma once

#include <cstdint>

namespace nghmm {

struct GlView {
  const double* dense = nullptr;    // [cells][3], or nullptr when packed
  const uint32_t* codes = nullptr;  // packed: 2-bit codes, 16 cells per word
  const double* table = nullptr;    // packed: [4][3] likelihoods of the four classes
  uint64_t cell0 = 0;               // index of the view's first cell in `dense` / `codes`
};

#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t gl_code(const uint32_t* __restrict__ codes, uint64_t k) {
  return (codes[k >> 4] >> ((uint32_t)(k & 15) * 2)) & 3u;
}

// likelihoods of cell c of the view (the branch is uniform across the grid)
__device__ __forceinline__ void gl_fetch(const GlView& v, uint64_t c, double& g0, double& g1,
                                         double& g2) {
  const uint64_t k = v.cell0 + c;
  if (v.dense) {
    const double* g = v.dense + k * 3;
    g0 = g[0];
    g1 = g[1];
    g2 = g[2];
  } else {
    const double* t = v.table + gl_code(v.codes, k) * 3;
    g0 = t[0];
    g1 = t[1];
    g2 = t[2];
  }
}
#endif

inline GlView gl_dense(const double* p, uint64_t cell0 = 0) {
  GlView v;
  v.dense = p;
  v.cell0 = cell0;
  return v;
}

inline GlView gl_packed(const uint32_t* codes, const double* table, uint64_t cell0 = 0) {
  GlView v;
  v.codes = codes;
  v.table = table;
  v.cell0 = cell0;
  return v;
}

}  // namespace nghmm

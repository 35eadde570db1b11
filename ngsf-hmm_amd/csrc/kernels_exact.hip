// kernels_exact.hip -- exact-mode HIP kernels for gfx950.
//
// "Exact" = the reference's log-space formulation in the reference's operation
// order (shared/HMM.cpp, shared/gen_func.cpp:856-1009), with exp/log taken from
// detmath.h.  IEEE add/sub/mul/div are correctly rounded on CDNA4 and this file
// is compiled with -ffp-contract=off, so every value equals, bit for bit, what
// the oracle's `det` build computes on the host.  That turns "GPU == CPU" into a
// testable statement despite the chaotic finite-difference M-step (SURVEY.md
// finding 4).  These kernels are the correctness anchor; kernels_fast.hip holds
// the throughput path.
//
// Parallel decomposition: the recursions are sequential in the site index, so a
// lane owns one (individual, parameter point) chain and walks the sites; lanes of
// a wave own consecutive individuals, which makes every site-major load a
// contiguous 1 KiB (e_prob) or 512 B (marg) segment.  Loads are software-
// prefetched one group of sites ahead because a chain has no other work to hide
// HBM latency behind.  est_maf gives a wave to a site and strides lanes over the
// individuals, then accumulates the per-individual terms in individual order so
// the sums round exactly like the reference's serial loop.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "detmath.h"
#include "glview.hpp"
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace nghmm {

namespace {

#include "exact_dev.hpp"

constexpr int U = 4;  // sites per prefetch group

// ------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_emission_exact(const GlView gl, const double* __restrict__ freq,
                 double* __restrict__ eprob, uint64_t S, uint64_t I, int* __restrict__ flags) {
  const uint64_t n = S * I;
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n;
       c += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t s = c / I;
    const double maf = freq[s];
    double g0, g1, g2;
    gl_fetch(gl, c, g0, g1, g2);
    double e0, e1;
    if (maf < 0 || maf > 1) {
      flags[FLAG_INVALID_MAF] = 1;
      e0 = e1 = __builtin_nan("");
    } else {
      e0 = emission_log(g0, g1, g2, maf, 0);
      e1 = emission_log(g0, g1, g2, maf, 1);
    }
    eprob[c * 2] = e0;
    eprob[c * 2 + 1] = e1;
  }
}

// ------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_forward_exact(const double* __restrict__ eprob, const double* __restrict__ pos, uint64_t S,
                uint64_t I, uint32_t n_pts, const uint32_t* __restrict__ ind,
                const double* __restrict__ Fv, const double* __restrict__ Av,
                double* __restrict__ lkl_out, double* __restrict__ fw, int* __restrict__ flags) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pts) return;
  const uint64_t i = ind ? ind[p] : p;
  const double f = Fv[p], a = Av[p];
  const double q0 = 1 - f, q1 = f;  // EM.cpp:415
  double prev0 = det_log(q0), prev1 = det_log(q1);
  if (fw) {
    fw[i * 2] = prev0;
    fw[i * 2 + 1] = prev1;
  }
  bool bad = false;
  const double2* e2 = reinterpret_cast<const double2*>(eprob);

  double2 ecur[U], enxt[U];
  double dcur[U], dnxt[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint64_t s = u;
    const bool v = s < S;
    ecur[u] = v ? e2[s * I + i] : double2{0, 0};
    dcur[u] = v ? pos[s] : 0.0;
  }
  for (uint64_t s0 = 0; s0 < S; s0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t s = s0 + U + u;
      const bool v = s < S;
      enxt[u] = v ? e2[s * I + i] : double2{0, 0};
      dnxt[u] = v ? pos[s] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t s = s0 + u;
      if (s < S) {
        const Trans t = calc_trans_all(q0, q1, a, dcur[u]);
        double tmp0 = prev0 + t.t00, tmp1 = prev1 + t.t10;
        bad |= (tmp0 != tmp0) | (tmp1 != tmp1);
        const double cur0 = logsum2(tmp0, tmp1) + ecur[u].x;
        tmp0 = prev0 + t.t01;
        tmp1 = prev1 + t.t11;
        bad |= (tmp0 != tmp0) | (tmp1 != tmp1);
        const double cur1 = logsum2(tmp0, tmp1) + ecur[u].y;
        prev0 = cur0;
        prev1 = cur1;
        if (fw) {
          fw[((s + 1) * I + i) * 2] = cur0;
          fw[((s + 1) * I + i) * 2 + 1] = cur1;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ecur[u] = enxt[u];
      dcur[u] = dnxt[u];
    }
  }
  lkl_out[p] = logsum2(prev0, prev1);
  if (bad) flags[FLAG_INVALID_LKL] = 1;
}

// ------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_backward_exact(const double* __restrict__ eprob, const double* __restrict__ pos,
                 const double* __restrict__ fw, uint64_t S, uint64_t I,
                 const double* __restrict__ indF, const double* __restrict__ alpha,
                 const double* __restrict__ ind_lkl, double* __restrict__ marg,
                 int* __restrict__ flags) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= I) return;
  const double f = indF[i], a = alpha[i];
  const double q0 = 1 - f, q1 = f;
  const double lkl = ind_lkl[i];
  double b0 = det_log(1.0), b1 = det_log(1.0);  // HMM.cpp:37
  bool bad = false, nanflag = false;
  const double2* e2 = reinterpret_cast<const double2*>(eprob);
  const double2* f2 = reinterpret_cast<const double2*>(fw);

  // walk s = S .. 1 (reference numbering); site index here is s-1
  double2 ecur[U], enxt[U], fcur[U], fnxt[U];
  double dcur[U], dnxt[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool v = (uint64_t)u < S;
    const uint64_t s = S - (v ? u : 0);  // reference site number
    ecur[u] = v ? e2[(s - 1) * I + i] : double2{0, 0};
    fcur[u] = v ? f2[s * I + i] : double2{0, 0};
    dcur[u] = v ? pos[s - 1] : 0.0;
  }
  for (uint64_t r0 = 0; r0 < S; r0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t r = r0 + U + u;
      const bool v = r < S;
      const uint64_t s = S - (v ? r : 0);
      enxt[u] = v ? e2[(s - 1) * I + i] : double2{0, 0};
      fnxt[u] = v ? f2[s * I + i] : double2{0, 0};
      dnxt[u] = v ? pos[s - 1] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t r = r0 + u;
      if (r < S) {
        const uint64_t s = S - r;
        // posterior of site s (EM.cpp:184); k = 0 is computed for its NaN check only
        const double m0 = check_interv(det_exp(b0 + fcur[u].x - lkl), nanflag);
        const double m1 = check_interv(det_exp(b1 + fcur[u].y - lkl), nanflag);
        (void)m0;
        marg[(s - 1) * I + i] = m1;
        // Bw[s-1] (HMM.cpp:40-52)
        const Trans t = calc_trans_all(q0, q1, a, dcur[u]);
        double tmp0 = t.t00 + ecur[u].x + b0;
        double tmp1 = t.t01 + ecur[u].y + b1;
        bad |= (tmp0 != tmp0) | (tmp1 != tmp1);
        const double nb0 = logsum2(tmp0, tmp1);
        tmp0 = t.t10 + ecur[u].x + b0;
        tmp1 = t.t11 + ecur[u].y + b1;
        bad |= (tmp0 != tmp0) | (tmp1 != tmp1);
        const double nb1 = logsum2(tmp0, tmp1);
        b0 = nb0;
        b1 = nb1;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ecur[u] = enxt[u];
      fcur[u] = fnxt[u];
      dcur[u] = dnxt[u];
    }
  }
  b0 += det_log(q0);  // HMM.cpp:55-56
  b1 += det_log(q1);
  const double bl = logsum2(b0, b1);
  const double2 fS = f2[S * I + i];
  const double fl = logsum2(fS.x, fS.y);
  const double diff = fl - bl;
  const double adiff = (diff >= 0) ? diff : -diff;  // the reference's abs macro
  if (adiff > 0.001) flags[FLAG_FW_BW] = 1;         // EM.cpp:167
  if (bad) flags[FLAG_INVALID_LKL] = 1;
  if (nanflag) flags[FLAG_NAN] = 1;
}

// ------------------------------------------------------------------
__device__ __forceinline__ double bcast_lane(double v, int lane) {
  const uint64_t bits = ngh_bits(v);
  const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)(bits & 0xffffffffu), lane);
  const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(bits >> 32), lane);
  return ngh_from_bits(((uint64_t)hi << 32) | lo);
}

// One wave per site.  Per pass every lane evaluates the posterior genotype terms
// of its individuals, then the 64 terms of a chunk are added to (num, den) in
// individual order by all lanes redundantly, so the running sums are wave-uniform
// and round exactly as the reference's `for i` loop (gen_func.cpp:984-1003).
// BG_WAVES > 0: the version that runs on the second stream UNDERNEATH the objective rounds of a
// fused iteration.  At its natural 64 VGPRs the kernel holds all eight wave slots of every
// SIMD, and a round's workgroups (three waves + 60 KB of LDS that must land on one CU together)
// then wait for slots behind the quarter of a million small workgroups of this launch: measured
// at 1000 x 1M, the rounds made no progress at all while est_maf ran (45 rounds 17.6 s against
// 13.5 s alone, est_maf 4.2 s against 2.4 s alone).  Claiming 512 / BG_WAVES registers caps the
// kernel at BG_WAVES waves per SIMD and leaves the other slots to the chains, whose waves
// raise their issue priority.
template <int BG_WAVES>
__global__ void __launch_bounds__(256)
k_estmaf_exact(const GlView gl, const double* __restrict__ marg, uint64_t S_own,
               uint64_t I, double* __restrict__ freq_out, uint32_t* __restrict__ passes_out) {
  if constexpr (BG_WAVES == 4) asm volatile("; occupancy cap" ::: "v127");
  if constexpr (BG_WAVES == 2) asm volatile("; occupancy cap" ::: "v255");
  if constexpr (BG_WAVES == 3) asm volatile("; occupancy cap" ::: "v167");
  __shared__ double2 terms[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t site = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (site >= S_own) return;
  const double* ms = marg + site * I;

  int iters = 0;
  uint32_t passes = 0;
  double num = 0, den = 0;
  double prev_freq, freq = 0.01;
  bool again;
  do {
    prev_freq = freq;
    ++passes;
    for (uint64_t base = 0; base < I; base += 64) {
      const uint64_t i = base + lane;
      double tn = 0, td = 0;
      if (i < I) {
        const double F = ms[i];
        double g0, g1, g2;
        gl_fetch(gl, site * I + i, g0, g1, g2);
        double h0, h1, h2;
        hwe_log(freq, F, h0, h1, h2);
        double p0 = g0 + h0, p1 = g1 + h1, p2 = g2 + h2;  // post_prob, gen_func.cpp:920-932
        const double norm = logsum3(p0, p1, p2);
        p0 -= norm;
        p1 -= norm;
        p2 -= norm;
        p0 = det_exp(p0);
        p1 = det_exp(p1);
        p2 = det_exp(p2);
        tn = p1 + p2 * (2 - F);
        td = 2 * p1 + (p0 + p2) * (2 - F);
      }
      const int cnt = (I - base) < 64 ? (int)(I - base) : 64;
      // the 64 terms meet in LDS and every lane reads them back in individual order (one
      // broadcast ds_read_b128 per individual, where two v_readlane pairs and the hazards
      // between SGPR writes and the adds that use them cost twice the instructions)
      terms[wv][lane] = double2{tn, td};
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (cnt == 64) {
#pragma unroll 16
        for (int j = 0; j < 64; ++j) {
          const double2 t = terms[wv][j];
          num += t.x;
          den += t.y;
        }
      } else {
        for (int j = 0; j < cnt; ++j) {
          const double2 t = terms[wv][j];
          num += t.x;
          den += t.y;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // (the next chunk overwrites the terms)
    }
    freq = num / den;
    const double dlt = prev_freq - freq;
    const double adl = (dlt >= 0) ? dlt : -dlt;
    again = (adl > kEPS) && (iters++ < 100);
  } while (again);
  if (lane == 0) {
    freq_out[site] = freq;
    if (passes_out) passes_out[site] = passes;
  }
}

// ------------------------------------------------------------------
// Viterbi, two kernels per chunk of sites.  The four log transition probabilities of
// a site (one exp + four logs, shared/HMM.cpp:130-139) do not depend on the recursion
// state, so they are computed for a whole chunk in parallel first; the sequential
// sweep is then only the additions and comparisons of HMM.cpp:104-117, in the same
// order on the same values -- the path stays the reference's bit for bit, and the
// sweep no longer waits on ~70 dependent transcendental steps per site.
__global__ void __launch_bounds__(256)
k_trans_log_exact(const double* __restrict__ pos, const double* __restrict__ indF,
                  const double* __restrict__ alpha, uint64_t s0, uint64_t n_s, uint64_t I,
                  double* __restrict__ tl) {
  const uint64_t n = n_s * I;
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n;
       c += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t s = c / I, i = c % I;
    const double f = indF[i];
    const Trans t = calc_trans_all(1 - f, f, alpha[i], pos[s0 + s]);
    double* o = tl + c * 4;
    o[0] = t.t00;
    o[1] = t.t10;
    o[2] = t.t01;
    o[3] = t.t11;
  }
}

// v_max_f64 of two values known not to be signalling NaNs (the compiler's fmax first
// canonicalises both operands); a quiet NaN operand is dropped (IEEE maxNum)
__device__ __forceinline__ double vmax_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// state [I][2] carries (Vi_prob[0], Vi_prob[1]) between chunks
__global__ void __launch_bounds__(64)
k_viterbi_fwd_exact(const double* __restrict__ eprob, const double* __restrict__ tl, uint64_t s0,
                    uint64_t n_s, uint64_t S, uint64_t I, const double* __restrict__ indF,
                    double* __restrict__ state, uint8_t* __restrict__ bp,
                    uint8_t* __restrict__ last_state, int chain_start) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= I) return;
  double v0, v1;
  // chain_start == 0: the sites continue another handle's (a site shard): `state` holds the
  // scores it ended with
  if (s0 == 0 && chain_start) {
    const double f = indF[i];
    v0 = det_log(1 - f);  // HMM.cpp:101-102
    v1 = det_log(f);
  } else {
    v0 = state[i * 2];
    v1 = state[i * 2 + 1];
  }
  const double2* e2 = reinterpret_cast<const double2*>(eprob);
  const double4* t4 = reinterpret_cast<const double4*>(tl);

  // 16 sites per group and two groups in flight: with only I/64 waves on the whole GPU
  // the sweep is bound by (bytes in flight) / (HBM latency), so it prefetches as deep as
  // the register file allows (one wave per SIMD is all there is anyway)
  constexpr int UV = 16;
  double2 ecur[UV], enxt[UV];
  double4 tcur[UV], tnxt[UV];
  // (loads past the chunk's end re-read its last site: an unconditional load with a clamped
  // address stays in flight across the loop, a conditional one is waited for on the spot)
#pragma unroll
  for (int u = 0; u < UV; ++u) {
    const uint64_t r = (uint64_t)u < n_s ? (uint64_t)u : n_s - 1;
    ecur[u] = e2[(s0 + r) * I + i];
    tcur[u] = t4[r * I + i];
  }
  uint32_t bpw[4] = {0, 0, 0, 0};
  for (uint64_t r0 = 0; r0 < n_s; r0 += UV) {
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      const uint64_t rr = r0 + UV + u;
      const uint64_t r = rr < n_s ? rr : n_s - 1;
      enxt[u] = e2[(s0 + r) * I + i];
      tnxt[u] = t4[r * I + i];
    }
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      const uint64_t r = r0 + u;
      if (r < n_s) {
        // HMM.cpp:105-116 per state: vmax = -INF; for k: pval = Vi_prob[k] + T(k,l); if (vmax <
        // pval) { vmax = pval; best = k; }.  From vmax = -INF the first test leaves max(-INF,
        // pval) -- a NaN pval fails the test and is dropped by v_max_f64 alike -- with best = 0
        // either way; the second leaves max(vmax, pval) (equal values: the same number) and best =
        // 1 exactly when the strict test holds.  Two dependent v_max instead of two compare +
        // select pairs on the chain that every site waits for; the tests for the back-pointers
        // are off it.
        const double c00 = v0 + tcur[u].x;  // k = 0 -> l = 0
        const double c10 = v1 + tcur[u].y;  // k = 1 -> l = 0
        const double m0 = vmax_f64(-kINF, c00);
        const int k0 = m0 < c10;
        v0 = vmax_f64(m0, c10) + ecur[u].x;  // in place: l = 1 below reads the NEW v0 (reference behaviour)
        const double c01 = v0 + tcur[u].z;  // k = 0 -> l = 1
        const double c11 = v1 + tcur[u].w;  // k = 1 -> l = 1
        const double m1 = vmax_f64(-kINF, c01);
        const int k1 = m1 < c11;
        v1 = vmax_f64(m1, c11) + ecur[u].y;
        bpw[u >> 2] |= (uint32_t)(k0 | (k1 << 1)) << (8 * (u & 3));
      }
    }
    // back-pointers of 16 consecutive sites of one individual = one 16-byte store;
    // layout [site/16][individual][16]  (s0 and r0 are multiples of 16)
    *reinterpret_cast<uint4*>(bp + (((s0 + r0) >> 4) * I + i) * 16) =
        uint4{bpw[0], bpw[1], bpw[2], bpw[3]};
    bpw[0] = bpw[1] = bpw[2] = bpw[3] = 0;
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      ecur[u] = enxt[u];
      tcur[u] = tnxt[u];
    }
  }
  state[i * 2] = v0;
  state[i * 2 + 1] = v1;
  if (s0 + n_s >= S) {
    // array_max_pos (gen_func.cpp:73-84): strict >, starting from -inf
    int res = 0;
    double mx = NEG_INFINITY;
    if (v0 > mx) { res = 0; mx = v0; }
    if (v1 > mx) { res = 1; mx = v1; }
    last_state[i] = (uint8_t)res;
  }
}

// The same sweep with the loads taken off the chain.  k_viterbi_fwd_exact is one lane per
// individual -- 16 waves on the whole chip for a cohort of 1000 -- and a wave can keep only the
// 768 B per lane in flight that its registers hold, so it runs at the 0.36 TB/s that HBM latency
// allows (133 ns per site).  Here a workgroup is ONE consumer wave for 64 individuals and VNL
// loader waves: loader w owns LDS slot w and the groups of VG sites g = w (mod VNL); it asks for
// group g + VNL while the consumer walks group g out of its slot, holds the data in registers
// for VNL - 1 steps and writes it to the slot in the step before it is needed -- VNL - 1 groups
// in flight per workgroup (5 x 24 KB) on top of what the slots hold.  One barrier per step.  The consumer's
// arithmetic is the serial kernel's, operation for operation: the same path.
constexpr int VG = 8;    // sites per group: a block of 16 back-pointers is two groups
constexpr int VNL = 6;   // loader waves = LDS slots (even: a block's two groups sit in slots 2k, 2k+1)
struct VitSlot {
  double2 e[VG][64];
  double4 t[VG][64];
};
constexpr size_t kVitLds = sizeof(VitSlot) * VNL;  // 144 KB of the CU's 160

__global__ void __launch_bounds__(64 * (1 + VNL)) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_viterbi_fwd_pc(const double* __restrict__ eprob, const double* __restrict__ tl, uint64_t s0,
                 uint64_t n_s, uint64_t S, uint64_t I, const double* __restrict__ indF,
                 double* __restrict__ state, uint8_t* __restrict__ bp,
                 uint8_t* __restrict__ last_state, int chain_start) {
  extern __shared__ __align__(16) unsigned char vit_lds[];
  VitSlot* slot = reinterpret_cast<VitSlot*>(vit_lds);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint64_t i_raw = (uint64_t)blockIdx.x * 64 + lane;
  const bool valid = i_raw < I;
  const uint64_t i = valid ? i_raw : I - 1;  // (idle lanes repeat the last individual, write nothing)
  const uint64_t ngroups = (n_s + VG - 1) / VG;
  const double2* e2 = reinterpret_cast<const double2*>(eprob);
  const double4* t4 = reinterpret_cast<const double4*>(tl);

  if (wv == 0) {
    double v0, v1;
    if (s0 == 0 && chain_start) {
      const double f = indF[i];
      v0 = det_log(1 - f);  // HMM.cpp:101-102
      v1 = det_log(f);
    } else {
      v0 = state[i * 2];
      v1 = state[i * 2 + 1];
    }
    __syncthreads();  // the loaders have filled slots 0 .. VNL-1 with groups 0 .. VNL-1
    // half a block of 16 back-pointers per step; `whole`: every site of the group exists (all
    // groups but possibly the chunk's last), so the walk carries no per-site test; HALF: which
    // eight bytes of the block's sixteen (static, so the bytes land in fixed registers)
    uint32_t bpw[4] = {0, 0, 0, 0};
    auto walk = [&](const VitSlot& sl, uint64_t g, auto whole, auto half) {
      constexpr int H = decltype(half)::value;
#pragma unroll
      for (int u = 0; u < VG; ++u) {
        const double2 ec = sl.e[u][lane];
        const double4 tc = sl.t[u][lane];
        if (decltype(whole)::value || g * VG + u < n_s) {  // (see k_viterbi_fwd_exact for the two v_max per state)
          const double c00 = v0 + tc.x;
          const double c10 = v1 + tc.y;
          const double m0 = vmax_f64(-kINF, c00);
          const int k0 = m0 < c10;
          v0 = vmax_f64(m0, c10) + ec.x;
          const double c01 = v0 + tc.z;
          const double c11 = v1 + tc.w;
          const double m1 = vmax_f64(-kINF, c01);
          const int k1 = m1 < c11;
          v1 = vmax_f64(m1, c11) + ec.y;
          constexpr int HB = H * 8;
          bpw[(HB + u) >> 2] |= (uint32_t)(k0 | (k1 << 1)) << (8 * ((HB + u) & 3));
        }
      }
    };
    auto flush = [&](uint64_t g) {  // the block that group g belongs to
      if (valid)
        *reinterpret_cast<uint4*>(bp + (((s0 + (g & ~1ull) * VG) >> 4) * I + i) * 16) =
            uint4{bpw[0], bpw[1], bpw[2], bpw[3]};
      bpw[0] = bpw[1] = bpw[2] = bpw[3] = 0;
    };
    uint32_t sidx = 0;  // g % VNL
    for (uint64_t g = 0; g < ngroups; g += 2) {
      if ((g + 1) * VG <= n_s) walk(slot[sidx], g, std::true_type{}, std::integral_constant<int, 0>{});
      else walk(slot[sidx], g, std::false_type{}, std::integral_constant<int, 0>{});
      __syncthreads();  // end of step g
      if (g + 1 < ngroups) {
        if ((g + 2) * VG <= n_s) walk(slot[sidx + 1], g + 1, std::true_type{}, std::integral_constant<int, 1>{});
        else walk(slot[sidx + 1], g + 1, std::false_type{}, std::integral_constant<int, 1>{});
        flush(g);
        __syncthreads();  // end of step g + 1
      } else {
        flush(g);
      }
      sidx = sidx + 2 == VNL ? 0 : sidx + 2;
    }
    if (valid) {
      state[i * 2] = v0;
      state[i * 2 + 1] = v1;
      if (s0 + n_s >= S) {
        // array_max_pos (gen_func.cpp:73-84): strict >, starting from -inf
        int res = 0;
        double mx = NEG_INFINITY;
        if (v0 > mx) { res = 0; mx = v0; }
        if (v1 > mx) { res = 1; mx = v1; }
        last_state[i] = (uint8_t)res;
      }
    }
  } else {
    const uint64_t w = (uint64_t)(wv - 1);
    VitSlot& mine = slot[w];
    double2 re[VG];
    double4 rt[VG];
    {  // group w into slot w
      const uint64_t g_ = w < ngroups ? w : ngroups - 1;
#pragma unroll
      for (int u = 0; u < VG; ++u) {
        const uint64_t rr = g_ * VG + u;
        const uint64_t r = rr < n_s ? rr : n_s - 1;  // past the chunk: its last site again
        re[u] = e2[(s0 + r) * I + i];
        rt[u] = t4[r * I + i];
      }
#pragma unroll
      for (int u = 0; u < VG; ++u) {
        mine.e[u][lane] = double2{re[u].x, re[u].y};  // (component-wise: a whole-struct copy keeps
        mine.t[u][lane] = double4{rt[u].x, rt[u].y, rt[u].z, rt[u].w};  // the arrays in scratch)
      }
    }
    __syncthreads();
    // Steps 0 .. ngroups-1, one barrier each.  This loader asks for a group in the steps g = w
    // (mod VNL) and writes it to its slot VNL - 1 steps later; the requests sit in straight-line
    // code (a load under a condition is waited for on the spot, and a loader that waits holds
    // up the step's barrier for everybody).
    for (uint64_t g = 0; g < w && g < ngroups; ++g) __syncthreads();
    for (uint64_t g = w; g < ngroups; g += VNL) {
      const uint64_t gn = g + VNL;
      const uint64_t g_ = gn < ngroups ? gn : ngroups - 1;
#pragma unroll
      for (int u = 0; u < VG; ++u) {  // step g: the consumer walks this loader's slot
        const uint64_t rr = g_ * VG + u;
        const uint64_t r = rr < n_s ? rr : n_s - 1;
        re[u] = e2[(s0 + r) * I + i];
        rt[u] = t4[r * I + i];
      }
      __syncthreads();
#pragma unroll
      for (int k = 1; k <= VNL - 2; ++k)
        if (g + k < ngroups) __syncthreads();     // steps g+1 .. g+VNL-2: the loads are in flight
      if (g + VNL - 1 < ngroups) {
#pragma unroll
        for (int u = 0; u < VG; ++u) {  // step g+VNL-1: group g+VNL, needed in the next step
          mine.e[u][lane] = double2{re[u].x, re[u].y};
          mine.t[u][lane] = double4{rt[u].x, rt[u].y, rt[u].z, rt[u].w};
        }
        __syncthreads();
      }
    }
  }
}

// Trace back (HMM.cpp:119-122): path of reference site s (1-based) = state at s.
// Back-pointers and the path are blocked [site/16][individual][16], so a lane moves 16
// sites per load/store; only the 1-bit select chain is sequential.
__global__ void __launch_bounds__(64)
k_viterbi_back(const uint8_t* __restrict__ bp, const uint8_t* __restrict__ last_state, uint64_t S,
               uint64_t I, uint8_t* __restrict__ path16, uint8_t* __restrict__ state_before) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= I || S == 0) return;
  int st = last_state[i];
  const uint64_t nblk = (S + 15) / 16;
  // The select chain of a block of 16 sites is ~200 cycles of work, a load from HBM several
  // times that: the PFB blocks of the NEXT group are requested before the current group is
  // walked.  Loads are unconditional with clamped block numbers (a conditional load is waited
  // for on the spot); blocks below 0 are masked out of the walk.
  constexpr int PFB = 16;
  uint4 cur[PFB], nxt[PFB];
  const int64_t top0 = (int64_t)nblk - 1;
#pragma unroll
  for (int k = 0; k < PFB; ++k) {
    const int64_t b = top0 - k;
    cur[k] = *reinterpret_cast<const uint4*>(bp + ((uint64_t)(b >= 0 ? b : 0) * I + i) * 16);
  }
  for (int64_t top = top0; top >= 0; top -= PFB) {
#pragma unroll
    for (int k = 0; k < PFB; ++k) {
      const int64_t b = top - PFB - k;
      nxt[k] = *reinterpret_cast<const uint4*>(bp + ((uint64_t)(b >= 0 ? b : 0) * I + i) * 16);
    }
#pragma unroll
    for (int k = 0; k < PFB; ++k) {
      const int64_t blk = top - k;
      const bool live = blk >= 0;
      const uint32_t w[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w};
      uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
      for (int u = 15; u >= 0; --u) {
        const uint64_t s = (uint64_t)(live ? blk : 0) * 16 + u;  // 0-based site = reference site s+1
        const bool on = live && s < S;
        o[u >> 2] |= (uint32_t)st << (8 * (u & 3));              // path[s+1]
        const uint32_t bq = (w[u >> 2] >> (8 * (u & 3))) & 0xff;  // Vi[s+1][.]
        const int prev = (bq >> st) & 1;                           // path[s] = Vi[s+1][path[s+1]]
        st = on ? prev : st;
        if (!on) o[u >> 2] &= ~(0xffu << (8 * (u & 3)));           // bytes past S stay 0
      }
      if (live) *reinterpret_cast<uint4*>(path16 + ((uint64_t)blk * I + i) * 16) = uint4{o[0], o[1], o[2], o[3]};
    }
#pragma unroll
    for (int k = 0; k < PFB; ++k) cur[k] = nxt[k];
  }
  // path[0] of the reference; for a site shard the state at the last site of the range before
  if (state_before) state_before[i] = (uint8_t)st;
}

// blocked [site/16][individual][16] -> [individual][site]
__global__ void __launch_bounds__(256)
k_unblock_path(const uint8_t* __restrict__ path16, uint64_t S, uint64_t I,
               uint8_t* __restrict__ out) {
  const uint64_t nblk = (S + 15) / 16;
  const uint64_t n = nblk * I;
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n;
       c += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t blk = c / I, i = c % I;
    const uint4 v = *reinterpret_cast<const uint4*>(path16 + c * 16);
    const uint64_t s = blk * 16;
    uint8_t* dst = out + i * S + s;
    if (s + 16 <= S && (((uintptr_t)dst) & 15) == 0) {
      *reinterpret_cast<uint4*>(dst) = v;
    } else {
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      for (int u = 0; u < 16 && s + u < S; ++u) dst[u] = (uint8_t)(w[u >> 2] >> (8 * (u & 3)));
    }
  }
}

// Input preparation of one cell, in place, in the reference's operation order: conversion
// to log space and normalisation (shared/read_data.cpp:36-40,89-98: conv_space + post_prob
// with no prior, gen_func.cpp:123-130,920-932), optional genotype call with the defaults of
// ngsF-HMM.cpp:105 (gen_func.cpp:886-914; array_max_pos / array_min_pos :73-98), second
// normalisation (ngsF-HMM.cpp:117).  A NaN after the first normalisation is the reference's
// "NaN found! Is the file format correct?" (read_data.cpp:42-45).
__global__ void __launch_bounds__(256)
k_prepare_gl(double* __restrict__ gl, uint64_t n_cells, int space, int call_geno,
             int* __restrict__ flags) {
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cells;
       c += (uint64_t)gridDim.x * blockDim.x) {
    double g[3] = {gl[c * 3], gl[c * 3 + 1], gl[c * 3 + 2]};
    // a cell the reader never filled (an empty text line still consumes a site,
    // read_data.cpp:60-61): it keeps the initial -1e15 and sees only the second normalisation
    const bool unread = ngh_bits(g[0]) == kUnreadBits;
    if (unread) g[0] = g[1] = g[2] = -kINF;
    if (unread) {
    } else if (space == 1) {  // binary file: conv_space(log), log 0 -> -1e15 (read_data.cpp:36-37)
#pragma unroll
      for (int k = 0; k < 3; ++k) g[k] = log_or_minf(g[k]);
    } else if (space == 2) {  // text file: plain log (read_data.cpp:89)
#pragma unroll
      for (int k = 0; k < 3; ++k) g[k] = det_log(g[k]);
    }
    double norm = unread ? 0.0 : logsum3(g[0], g[1], g[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] -= norm;
    if (g[0] != g[0] || g[1] != g[1] || g[2] != g[2]) flags[FLAG_NAN] = 1;
    if (call_geno) {
      int max_pos = 0, min_pos = 0;
      double mx = NEG_INFINITY, mn = __builtin_huge_val();
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (g[k] > mx) {
          max_pos = k;
          mx = g[k];
        }
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (g[k] < mn) {
          min_pos = k;
          mn = g[k];
        }
      double max_pp = det_exp(g[max_pos]);
      if (g[min_pos] == g[max_pos]) max_pp = -1;  // missing data
      if (max_pp < 0) {
        const double u = det_log((double)1 / 3);
        g[0] = g[1] = g[2] = u;
      }
      if (max_pp >= 0) {
        g[0] = g[1] = g[2] = -kINF;
        g[max_pos] = det_log(1.0);
      }
    }
    norm = logsum3(g[0], g[1], g[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) gl[c * 3 + k] = g[k] - norm;
  }
}

// Genotype posteriors of the .geno output (EM.cpp:367-376): prior = calc_HWE(freq[s], F)
// with F the decoded state of the cell, post_prob, back to normal space.  path16 is the
// blocked Viterbi path [site/16][individual][16]; out [n_s][I][3] for sites s0 .. s0 + n_s.
__global__ void __launch_bounds__(256)
k_geno_post_exact(const GlView gl, const double* __restrict__ freq,
                  const uint8_t* __restrict__ path16, uint64_t I, uint64_t s0, uint64_t n_s,
                  double* __restrict__ out) {
  const uint64_t n = n_s * I;
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n;
       c += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t s = s0 + c / I, i = c % I;
    const double F = (double)path16[((s >> 4) * I + i) * 16 + (s & 15)];
    double h0, h1, h2;
    hwe_log(freq[s], F, h0, h1, h2);
    double g0, g1, g2;
    gl_fetch(gl, s * I + i, g0, g1, g2);
    const double p0 = g0 + h0, p1 = g1 + h1, p2 = g2 + h2;
    const double norm = logsum3(p0, p1, p2);
    out[c * 3 + 0] = det_exp(p0 - norm);
    out[c * 3 + 1] = det_exp(p1 - norm);
    out[c * 3 + 2] = det_exp(p2 - norm);
  }
}


// ---- called genotypes as 2-bit codes (glview.hpp) ---------------------------------------
// Classifies prepared cells: class g = the cell equals, bit for bit, the prepared likelihoods
// of called genotype g (table rows 0..2: what k_prepare_gl makes of a one-hot cell); class 3 =
// three equal values (missing data, or a cell --call_geno leaves uniform).  All class-3 cells
// of a data set must carry the same value: the first one seen is recorded in *uniform_bits
// (initially ~0) and every other is compared with it.  Anything else raises flags[FLAG_NAN + 1]
// ("not packable").  codes must be zeroed before the first chunk (atomicOr: chunks of sites
// need not start on a word boundary).
__global__ void __launch_bounds__(256)
k_pack_cells(const double* __restrict__ gl, uint64_t n_cells, uint64_t cell0,
             const double* __restrict__ table, uint32_t* __restrict__ codes,
             unsigned long long* __restrict__ uniform_bits, int* __restrict__ flags) {
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cells;
       c += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t b0 = ngh_bits(gl[c * 3]), b1 = ngh_bits(gl[c * 3 + 1]),
                   b2 = ngh_bits(gl[c * 3 + 2]);
    uint32_t code = 4;
#pragma unroll
    for (int g = 0; g < 3; ++g)
      if (b0 == ngh_bits(table[g * 3]) && b1 == ngh_bits(table[g * 3 + 1]) &&
          b2 == ngh_bits(table[g * 3 + 2]))
        code = g;
    if (code == 4 && b0 == b1 && b1 == b2 && gl[c * 3] == gl[c * 3]) {
      const unsigned long long old = atomicCAS(uniform_bits, ~0ull, (unsigned long long)b0);
      if (old == ~0ull || old == b0) code = 3;
    }
    if (code == 4) {
      flags[FLAG_NOT_PACKABLE] = 1;
      code = 3;
    }
    const uint64_t k = cell0 + c;
    atomicOr(&codes[k >> 4], code << ((uint32_t)(k & 15) * 2));
  }
}

// genotype codes as the reader sees them (-1 missing, 0, 1, 2) -> 2-bit codes
__global__ void __launch_bounds__(256)
k_pack_geno(const int8_t* __restrict__ geno, uint64_t n_cells, uint64_t cell0,
            uint32_t* __restrict__ codes, int* __restrict__ flags) {
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cells;
       c += (uint64_t)gridDim.x * blockDim.x) {
    const int g = geno[c];
    if (g > 2) flags[FLAG_BAD_GENO] = 1;  // "Genotypes must be coded as {-1,0,1,2} !"
    const uint32_t code = (g < 0 || g > 2) ? 3u : (uint32_t)g;
    const uint64_t k = cell0 + c;
    atomicOr(&codes[k >> 4], code << ((uint32_t)(k & 15) * 2));
  }
}

// the raw likelihoods the reference's reader builds from a called genotype
// (shared/read_data.cpp:21,88-98): log(1) at the genotype, -1e15 elsewhere; missing: the
// caller's log(1/3) three times
__global__ void __launch_bounds__(256)
k_expand_geno(const int8_t* __restrict__ geno, uint64_t n_cells, double log_third,
              double* __restrict__ gl, int* __restrict__ flags) {
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cells;
       c += (uint64_t)gridDim.x * blockDim.x) {
    const int g = geno[c];
    if (g > 2) flags[FLAG_BAD_GENO] = 1;
    double v0 = -kINF, v1 = -kINF, v2 = -kINF;
    if (g < 0 || g > 2) v0 = v1 = v2 = log_third;
    else if (g == 0) v0 = 0.0;
    else if (g == 1) v1 = 0.0;
    else v2 = 0.0;
    gl[c * 3] = v0;
    gl[c * 3 + 1] = v1;
    gl[c * 3 + 2] = v2;
  }
}

// any view -> dense [cells][3] (host read-back of the prepared likelihoods) or int8 codes
__global__ void __launch_bounds__(256)
k_unpack_cells(const GlView gl, uint64_t n_cells, double* __restrict__ out) {
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cells;
       c += (uint64_t)gridDim.x * blockDim.x) {
    double g0, g1, g2;
    gl_fetch(gl, c, g0, g1, g2);
    out[c * 3] = g0;
    out[c * 3 + 1] = g1;
    out[c * 3 + 2] = g2;
  }
}

__global__ void __launch_bounds__(256)
k_codes_to_bytes(const uint32_t* __restrict__ codes, uint64_t cell0, uint64_t n_cells,
                 uint8_t* __restrict__ out) {
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cells;
       c += (uint64_t)gridDim.x * blockDim.x)
    out[c] = (uint8_t)gl_code(codes, cell0 + c);
}

__global__ void __launch_bounds__(256)
k_bytes_to_codes(const uint8_t* __restrict__ in, uint64_t n_cells, uint32_t* __restrict__ codes) {
  // one thread per output word (the destination starts at cell 0)
  const uint64_t n_words = (n_cells + 15) / 16;
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words;
       w += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint64_t c = w * 16 + j;
      if (c < n_cells) v |= (uint32_t)(in[c] & 3) << (2 * j);
    }
    codes[w] = v;
  }
}

}  // namespace

void launch_prepare_gl(hipStream_t st, double* gl, uint64_t n_cells, int space, int call_geno,
                       int* flags) {
  if (n_cells == 0) return;
  uint64_t blocks = (n_cells + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(k_prepare_gl, dim3((unsigned)blocks), dim3(256), 0, st, gl, n_cells, space,
                     call_geno, flags);
}


static unsigned grid_for(uint64_t n) {
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  return (unsigned)(blocks ? blocks : 1);
}

void launch_pack_cells(hipStream_t st, const double* gl, uint64_t n_cells, uint64_t cell0,
                       const double* table, uint32_t* codes, unsigned long long* uniform_bits,
                       int* flags) {
  if (n_cells == 0) return;
  hipLaunchKernelGGL(k_pack_cells, dim3(grid_for(n_cells)), dim3(256), 0, st, gl, n_cells, cell0,
                     table, codes, uniform_bits, flags);
}

void launch_pack_geno(hipStream_t st, const int8_t* geno, uint64_t n_cells, uint64_t cell0,
                      uint32_t* codes, int* flags) {
  if (n_cells == 0) return;
  hipLaunchKernelGGL(k_pack_geno, dim3(grid_for(n_cells)), dim3(256), 0, st, geno, n_cells, cell0,
                     codes, flags);
}

void launch_expand_geno(hipStream_t st, const int8_t* geno, uint64_t n_cells, double log_third,
                        double* gl, int* flags) {
  if (n_cells == 0) return;
  hipLaunchKernelGGL(k_expand_geno, dim3(grid_for(n_cells)), dim3(256), 0, st, geno, n_cells,
                     log_third, gl, flags);
}

void launch_unpack_cells(hipStream_t st, const GlView& gl, uint64_t n_cells, double* out) {
  if (n_cells == 0) return;
  hipLaunchKernelGGL(k_unpack_cells, dim3(grid_for(n_cells)), dim3(256), 0, st, gl, n_cells, out);
}

void launch_codes_to_bytes(hipStream_t st, const uint32_t* codes, uint64_t cell0, uint64_t n_cells,
                           uint8_t* out) {
  if (n_cells == 0) return;
  hipLaunchKernelGGL(k_codes_to_bytes, dim3(grid_for(n_cells)), dim3(256), 0, st, codes, cell0,
                     n_cells, out);
}

void launch_bytes_to_codes(hipStream_t st, const uint8_t* in, uint64_t n_cells, uint32_t* codes) {
  if (n_cells == 0) return;
  hipLaunchKernelGGL(k_bytes_to_codes, dim3(grid_for((n_cells + 15) / 16)), dim3(256), 0, st, in,
                     n_cells, codes);
}

void launch_geno_post_exact(hipStream_t st, const GlView& gl, const double* freq,
                            const uint8_t* path16, uint64_t I, uint64_t s0, uint64_t n_s,
                            double* out) {
  if (n_s == 0 || I == 0) return;
  uint64_t blocks = (n_s * I + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(k_geno_post_exact, dim3((unsigned)blocks), dim3(256), 0, st, gl, freq, path16,
                     I, s0, n_s, out);
}

void launch_emission_exact(hipStream_t st, const GlView& gl, const double* freq, double* eprob,
                           uint64_t S, uint64_t I, int* flags) {
  const uint64_t n = S * I;
  if (n == 0) return;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(k_emission_exact, dim3((unsigned)blocks), dim3(256), 0, st, gl, freq, eprob, S,
                     I, flags);
}

void launch_forward_exact(hipStream_t st, const double* eprob, const double* pos, uint64_t S,
                          uint64_t I, uint32_t n_pts, const uint32_t* ind, const double* F,
                          const double* alpha, double* lkl_out, double* fw, int* flags) {
  if (n_pts == 0) return;
  hipLaunchKernelGGL(k_forward_exact, dim3((n_pts + 63) / 64), dim3(64), 0, st, eprob, pos, S, I,
                     n_pts, ind, F, alpha, lkl_out, fw, flags);
}

void launch_backward_exact(hipStream_t st, const double* eprob, const double* pos, const double* fw,
                           uint64_t S, uint64_t I, const double* indF, const double* alpha,
                           const double* ind_lkl, double* marg, int* flags) {
  if (I == 0) return;
  hipLaunchKernelGGL(k_backward_exact, dim3((unsigned)((I + 63) / 64)), dim3(64), 0, st, eprob, pos,
                     fw, S, I, indF, alpha, ind_lkl, marg, flags);
}

void launch_estmaf_exact(hipStream_t st, const GlView& gl_sites, const double* marg_sites,
                         uint64_t S_own, uint64_t I_tot, double* freq_out, uint32_t* passes_out,
                         int bg_waves) {
  if (S_own == 0) return;
  const dim3 grid((unsigned)((S_own + 3) / 4)), block(256);
  if (bg_waves == kExactBgWaves)
    hipLaunchKernelGGL((k_estmaf_exact<kExactBgWaves>), grid, block, 0, st, gl_sites, marg_sites, S_own, I_tot,
                       freq_out, passes_out);
  else
    hipLaunchKernelGGL((k_estmaf_exact<0>), grid, block, 0, st, gl_sites, marg_sites, S_own, I_tot, freq_out,
                       passes_out);
}

void launch_viterbi_fwd_exact(hipStream_t st, const double* eprob, const double* pos, uint64_t S,
                             uint64_t I, const double* indF, const double* alpha, uint8_t* bp,
                             double* scratch, uint64_t chunk_sites, bool chain_start, bool serial) {
  if (I == 0 || S == 0) return;
  // bp holds ceil(S/16)*16*I back-pointer bytes (blocked) followed by I last-state
  // bytes; scratch holds chunk_sites*I*4 transition logs followed by I*2 state doubles
  uint8_t* last_state = bp + ((S + 15) / 16) * 16 * I;
  double* tl = scratch;
  double* state = scratch + chunk_sites * I * 4;
  static const bool big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(k_viterbi_fwd_pc),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)kVitLds) == hipSuccess;
  if (!big_lds) (void)hipGetLastError();
  for (uint64_t s0 = 0; s0 < S; s0 += chunk_sites) {
    const uint64_t n_s = (S - s0) < chunk_sites ? (S - s0) : chunk_sites;
    uint64_t blocks = (n_s * I + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_trans_log_exact, dim3((unsigned)blocks), dim3(256), 0, st, pos, indF,
                       alpha, s0, n_s, I, tl);
    if (serial || !big_lds)
      hipLaunchKernelGGL(k_viterbi_fwd_exact, dim3((unsigned)((I + 63) / 64)), dim3(64), 0, st, eprob,
                         tl, s0, n_s, S, I, indF, state, bp, last_state, chain_start ? 1 : 0);
    else
      hipLaunchKernelGGL(k_viterbi_fwd_pc, dim3((unsigned)((I + 63) / 64)), dim3(64 * (1 + VNL)), kVitLds,
                         st, eprob, tl, s0, n_s, S, I, indF, state, bp, last_state, chain_start ? 1 : 0);
  }
}

void launch_viterbi_back_exact(hipStream_t st, uint8_t* bp, uint64_t S, uint64_t I,
                               uint8_t* path_sites, uint8_t* state_before) {
  if (I == 0 || S == 0) return;
  uint8_t* last_state = bp + ((S + 15) / 16) * 16 * I;
  hipLaunchKernelGGL(k_viterbi_back, dim3((unsigned)((I + 63) / 64)), dim3(64), 0, st, bp,
                     last_state, S, I, path_sites, state_before);
}

void launch_viterbi_exact(hipStream_t st, const double* eprob, const double* pos, uint64_t S,
                          uint64_t I, const double* indF, const double* alpha, uint8_t* bp,
                          uint8_t* path_sites, double* scratch, uint64_t chunk_sites, bool serial) {
  launch_viterbi_fwd_exact(st, eprob, pos, S, I, indF, alpha, bp, scratch, chunk_sites, true, serial);
  launch_viterbi_back_exact(st, bp, S, I, path_sites, nullptr);
}

void launch_unblock_path(hipStream_t st, const uint8_t* path16, uint64_t S, uint64_t I,
                         uint8_t* out) {
  hipLaunchKernelGGL(k_unblock_path, dim3(4096), dim3(256), 0, st, path16, S, I, out);
}

uint64_t viterbi_chunk_sites(uint64_t S, uint64_t I) {
  // about 2 GiB of transition logs per chunk; a multiple of 16 (blocked back-pointers)
  uint64_t ch = (2ull << 30) / (I * 32);
  ch &= ~15ull;
  if (ch < 64) ch = 64;
  const uint64_t Sp = (S + 15) & ~15ull;
  return ch < Sp ? ch : Sp;
}

uint64_t viterbi_blocked_bytes(uint64_t S, uint64_t I) { return ((S + 15) / 16) * 16 * I; }

}  // namespace nghmm

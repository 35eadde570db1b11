// kernels_fast_estmaf.hip -- the allele-frequency step (est_maf, gen_func.cpp:974-1009): register kernels with checked
// Chebyshev interpolation of the per-pass sums, four sites per wave for small cohorts, the closed
// form for called genotypes, the streaming / log-space kernel
// (fast mode, gfx950; kernels_fast.hip's header comment has the design, DESIGN.md section 4 the
// measurements.)
#include "fast_dev.hpp"

#include <mutex>

namespace nghmm {

namespace {


// ---- est_maf ----------------------------------------------------------------

// reference-order log-space term for a cell whose linear weights all vanish
// (e.g. a called heterozygote with posterior IBD = 1): gen_func.cpp:984-1000
__device__ double2 estmaf_term_logspace(const double* g, double freq, double F) {
  double h[3];
  h[0] = (1 - freq) * (1 - freq) + (1 - freq) * freq * F;
  h[1] = 2 * (1 - freq) * freq - 2 * (1 - freq) * freq * F;
  h[2] = freq * freq + (1 - freq) * freq * F;
  double pp[3];
  for (int k = 0; k < 3; ++k) {
    double l = log(h[k]);
    if (l == -__builtin_huge_val()) l = -kINF;
    h[k] = l;
  }
  if (F == 1) h[1] = -kINF;
  double M = g[0] + h[0];
  for (int k = 0; k < 3; ++k) {
    pp[k] = g[k] + h[k];
    M = (pp[k] >= M) ? pp[k] : M;
  }
  double sum = 0;
  for (int k = 0; k < 3; ++k) sum += exp(pp[k] - M);
  const double norm = log(sum) + M;
  for (int k = 0; k < 3; ++k) pp[k] = exp(pp[k] - norm);
  return double2{pp[1] + pp[2] * (2 - F), 2 * pp[1] + (pp[0] + pp[2]) * (2 - F)};
}

// ---- wave-wide sum that ends in a wave-uniform value -----------------------
// DPP moves stay inside the SIMD (no LDS round trip as with ds_bpermute), which
// matters here: est_maf has one dependent reduction per pass and ~100 passes.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  // full row mask and in-row permutations: every lane is written, so the "old" operand
  // is irrelevant (mov_dpp leaves it undefined and saves the two zeroing moves)
  const uint64_t b = ngh_bits(v);
  const int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)b, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(b >> 32), CTRL, 0xf, 0xf, true);
  return ngh_from_bits(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

__device__ __forceinline__ double lane_value(double v, int lane) {
  const uint64_t b = ngh_bits(v);
  const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)b, lane);
  const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
  return ngh_from_bits(((uint64_t)hi << 32) | lo);
}

// total in the lanes of the last row (48..63); other lanes hold partial sums
__device__ __forceinline__ double wave_sum_lastrow(double v) {
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror: every lane holds its row total
  {  // row_bcast15 into rows 1 and 3, then row_bcast31 into rows 2 and 3
    const uint64_t b = ngh_bits(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x142, 0xa, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x142, 0xa, 0xf, false);
    v += ngh_from_bits(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
  }
  {
    const uint64_t b = ngh_bits(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x143, 0xc, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x143, 0xc, 0xf, false);
    v += ngh_from_bits(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
  }
  return v;
}

__device__ __forceinline__ double wave_sum_uniform(double v) {
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror: every lane now holds its 16-lane row total
  return ((lane_value(v, 0) + lane_value(v, 16)) + lane_value(v, 32)) + lane_value(v, 48);
}

// Sums of two per-lane values over the wave in ONE reduction tree: the first step swaps
// the upper half of pn with the lower half of pd (v_permlane32_swap, gfx950), so lanes
// 0..31 carry pn partials and lanes 32..63 pd partials; four in-row DPP steps and one
// row_bcast15 finish both.  Returns the value whose lane 31 holds sum(pn) and lane 63
// sum(pd).
__device__ __forceinline__ double wave_sum_pair(double pn, double pd) {
  const uint64_t bn = ngh_bits(pn), bd = ngh_bits(pd);
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)bn, (unsigned)bd, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(bn >> 32), (unsigned)(bd >> 32),
                                                   false, false);
  double v = ngh_from_bits(((uint64_t)hi[0] << 32) | lo[0]) +
             ngh_from_bits(((uint64_t)hi[1] << 32) | lo[1]);
  v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror: every lane holds its row total
  {  // row_bcast15 into rows 1 and 3
    const uint64_t b = ngh_bits(v);
    const int l = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, 0x142, 0xa, 0xf, false);
    const int h = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), 0x142, 0xa, 0xf, false);
    v += ngh_from_bits(((uint64_t)(uint32_t)h << 32) | (uint32_t)l);
  }
  return v;
}

// ---- est_maf: certified interpolation of the per-pass sums -------------------
// The reference's loop (gen_func.cpp:981-1006) is a running average: pass k evaluates
// two sums over all individuals at the odds r_k of the current frequency and adds them
// to num/den; r_k creeps towards its limit like 1/k, so nearly every site runs into the
// 100-pass cap.  Both sums are rational functions of r whose poles all lie in Re r <= 0
// (their denominators sA + r sb + r^2 sC have non-negative coefficients), hence analytic
// in a disc of radius >= r around any r > 0.  In the Moebius variable of the build below an
// interval of ratio hi / lo = 2 has Bernstein-ellipse parameter 11.7: EN = 12 Chebyshev nodes
// reproduce the sums to 11.7^-12 = 1.5e-13 before the constant (measured against all-exact
// passes at five full-size shapes: frequencies within 6.7e-13; 14 nodes, the previous
// default, 8e-15 -- three and a half orders inside the 1e-9 the frequencies are held to,
// for two evaluations of all individuals fewer per site: est_maf 8.8 -> 8.1 ms at 1000 x 1M).
// So after a few exact passes the kernel evaluates the sums exactly at the EN Chebyshev nodes
// of an interval ahead of r_k (as expensive as EN passes), CHECKS the interpolant against the
// next exact pass (relative EST_TOL = 1e-11, else the site stays on exact passes), and hands
// the site to k_fast_estmaf_interp, where one LANE per site runs the
// remaining passes on the barycentric formula: the same recursion, same pass count,
// same stopping rule, at ~1/60 of the cost per pass.  A pass whose stopping decision
// would be closer than 1e-9 (relative) to the threshold, or whose r leaves the
// interval, goes back to exact evaluation (one more build is allowed per site).
constexpr int EN = 12;                  // Chebyshev nodes per interval
constexpr int EST_SCALARS = 8;          // num, den, pnum, pden, iters, mid, half, tF
constexpr int EST_FIELDS = EST_SCALARS + 2 * EN;
enum : uint8_t { EST_DONE = 0, EST_INTERP = 1, EST_EXACT = 2 };
constexpr int EST_K0 = 2;               // exact passes before the first interval
constexpr int EST_MIN_GAIN = 24;        // build only if about this many passes remain
constexpr double EST_DMAX = 0.85;       // interval length <= EST_DMAX * r ahead ...
constexpr double EST_BACK = 0.1;        // ... plus this fraction of it behind
constexpr double EST_MULT = 32.0;       // ... and about this many current steps
constexpr double EST_FIT = 0.72;        // build once k * step <= EST_FIT * EST_DMAX * r ...
constexpr int EST_KMAX = 32;            // ... or after this many passes at the latest
constexpr double EST_TOL = 1e-11;       // interpolant vs exact pass, relative
constexpr double EST_GUARD = 1e-9;      // stopping decisions this close go back to exact
// cos((2j+1) pi/(2 EN)) and (-1)^j sin((2j+1) pi/(2 EN)): first-kind Chebyshev nodes and
// their barycentric weights
__constant__ double kChebC[EN] = {0.9914448613738104, 0.9238795325112867, 0.7933533402912352, 0.6087614290087207, 0.38268343236508984, 0.1305261922200517, -0.1305261922200516, -0.3826834323650895, -0.6087614290087207, -0.793353340291235, -0.9238795325112867, -0.9914448613738104};
__constant__ double kChebW[EN] = {0.13052619222005157, -0.3826834323650898, 0.6087614290087207, -0.7933533402912352, 0.9238795325112867, -0.9914448613738104, 0.9914448613738104, -0.9238795325112868, 0.7933533402912352, -0.6087614290087209, 0.3826834323650899, -0.130526192220052};

// W = BLOCK/64 waves per site, NI individuals per lane held in registers.  With
//   A = (1-f)^2, b = (1-f) f, C = f^2
// the weights w_g = p_g * HWE_g(f, F) of calc_HWE/post_prob (gen_func.cpp:920-957) are
// linear in (A, b, C):  w0 = p0 (A + bF), w1 = b c1 with c1 = 2 p1 (1-F), w2 = p2 (C + bF),
// and the reference's per-individual terms (gen_func.cpp:999-1000) become
//   num-term = (w1 + (2-F) w2) / sum
//   den-term = (2 w1 + (2-F)(w0 + w2)) / sum = (2-F) + F w1 / sum
// Dividing every weight by (1-f)^2 leaves, in the odds r = f/(1-f),
//   sum' = sA + r sb + r^2 sC,  num-term = r (u0 + r nC) / sum',  den-term = (2-F) + r fc / sum'
// six constants per individual, 8 FP64 instructions per individual and evaluation with
// the reciprocals taken four at a time, and the (2-F) part of the denominator a per-site
// constant.  Nothing is read from memory again after the constants are formed.  A site
// with a cell whose weights all vanish (a called heterozygote at posterior IBD = 1, ...)
// ends with a non-finite frequency, is flagged and redone by k_fast_estmaf_stream, which
// takes the reference-order log-space route for such cells.
//
// Up to 1024 individuals one wave holds the whole site (NI <= 16: 192 VGPRs of constants,
// two waves per SIMD) and an evaluation is 128 + ~45 instructions; beyond that W waves
// share a site (e.g. the site-sharded frequency step of a multi-GPU run), their partial
// sums meet in LDS once per evaluation (double-buffered, one barrier) and are added in
// wave order, so the result does not depend on scheduling.
//
// fresh != 0: every site starts the loop; else only sites whose status is EST_EXACT
// resume from `state`.  n_exact passes are evaluated exactly, then (allow_build) the
// interval is built and checked; a site that ends here writes freq_out/redo.
constexpr int ESTMAF_MAXW = 16;

// is site (c*64 + l)*T + t in the tile rows c*T + t of [row0, row1)?  (tile_T == 0: no tiles,
// every site is)
__device__ __forceinline__ bool in_tile_rows(uint64_t site, uint64_t tile_T, uint64_t row0,
                                             uint64_t row1) {
  if (tile_T == 0) return true;
  const uint64_t j = site / tile_T, t = site - j * tile_T;
  const uint64_t row = (j >> 6) * tile_T + t;
  return row >= row0 && row < row1;
}
// TILE: the posteriors are read from the E-step's tile-major layout (post_lane_off /
// post_ind_off; site (c*64 + l)*T + t), one wave per site.  With kPost8 a wave-load of 64
// consecutive individuals is eight fully used 64 B sectors.  (Without: a lane's 8-byte loads
// are 512 B apart and the sector around each holds the eight sites l0..l0+7 of one individual;
// workgroups go round-robin to the 8 XCDs, each with its own L2, so the blockIdx -> site map
// gives XCD x the sites l = 8x..8x+7 of every tile row in eight consecutive workgroups, for
// the sector to be fetched once and hit in that L2 seven times.  The map is kept.)
// one site on the W = BLOCK / 64 waves of a workgroup (see above); the shared arrays are the
// calling kernel's
// The size dispatch of fast_estmaf gives the variant (NI, BLOCK) only to cohorts larger than
// the previous variant holds, so its first slots are full for every thread: no masking there.
__host__ __device__ constexpr int estmaf_full_slots(int NI, int BLOCK) {
  return BLOCK == 64 ? (NI == 16 ? 12 : NI == 12 ? 8 : NI == 8 ? 4 : NI == 4 ? 2 : NI == 2 ? 1 : 0)
                     : (NI == 16 ? 8 : 0);  // 128: > 1024 = 8 x 128; 256: > 2048; 512: > 4096
}

template <int NI, int BLOCK, bool TILE>
__device__ __forceinline__ void estmaf_site(
    const GlView& gl, const double* __restrict__ marg_blocks, uint64_t S_own, uint64_t I_tot,
    uint64_t I_blk, double* __restrict__ freq_out, uint8_t* __restrict__ redo,
    uint8_t* __restrict__ status, double* __restrict__ state, uint64_t state_stride, int fresh,
    int n_exact, int allow_build, uint64_t site, const double* __restrict__ tile_col,
    uint32_t* __restrict__ cnt, double (&xch)[2][ESTMAF_MAXW][2],
    double2 (&nodebuf)[(BLOCK == 64 && NI >= 8) ? EN : 1][(BLOCK == 64 && NI >= 8) ? 65 : 1],
    double2 (&xnode)[(BLOCK == 64 && NI >= 8) ? 1 : EN][(BLOCK == 64 && NI >= 8) ? 1 : BLOCK / 64],
    double2* park = nullptr) {
  constexpr int W = BLOCK / 64;
  constexpr bool PARK = (W == 1 && NI >= 8);
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const uint32_t tix = threadIdx.x;  // index among the site's threads
  constexpr uint64_t stride = BLOCK;
  const uint64_t cell_s = site * I_tot;  // first cell of the site's row

  double tF_lane_out;
  // The loads of eight slots (32 per lane) are issued before anything waits on them
  // (out-of-range slots re-read the last individual and are masked afterwards): a wave
  // has two memory round trips here, not NI of them.
  double sA[NI], sb[NI], sC[NI], u0[NI], nC[NI], fc[NI];
  {
    // slots per batch of loads: eight; NI = 12: two batches of six
    constexpr int NB = NI < 8 ? NI : (NI % 8 ? NI / 2 : 8);
    static_assert(NI % NB == 0, "whole batches");
    const bool one_block = (I_blk == I_tot);
    const uint32_t ib = (uint32_t)I_blk;
    double tF_acc = 0;
#pragma unroll
    for (int k0 = 0; k0 < NI; k0 += NB) {
      double r0[NB], r1[NB], r2[NB], rF[NB];
      uint64_t ic[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const uint64_t i = (uint64_t)tix + stride * (k0 + j);
        // (a slot that is full for every thread needs no clamp: its addresses are the lane's
        // plus a constant)
        ic[j] = (k0 + j < estmaf_full_slots(NI, BLOCK) || i < I_tot) ? i : I_tot - 1;
        gl_fetch(gl, cell_s + ic[j], r0[j], r1[j], r2[j]);
      }
      if constexpr (TILE) {
#pragma unroll
        for (int j = 0; j < NB; ++j)  // (post_ind_off is additive over multiples of 8)
          rF[j] = tile_col[k0 + j < estmaf_full_slots(NI, BLOCK)
                               ? post_ind_off(tix) + (uint64_t)(k0 + j) * post_ind_off(stride)
                               : post_ind_off(ic[j])];
      } else if (one_block) {
#pragma unroll
        for (int j = 0; j < NB; ++j) rF[j] = marg_blocks[site * I_blk + ic[j]];
      } else {  // posteriors arrive in rank blocks [I_tot / I_blk][S_own][I_blk]
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const uint32_t q = (uint32_t)ic[j] / ib;
          rF[j] = marg_blocks[((uint64_t)q * S_own + site) * I_blk + ((uint32_t)ic[j] - q * ib)];
        }
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int k = k0 + j;
        const bool valid = k < estmaf_full_slots(NI, BLOCK) || (uint64_t)tix + stride * k < I_tot;
        // empty slot: likelihoods (1, 0, 0) at posterior 0 give sum' = 1 and numerators 0 --
        // it contributes nothing (four selects on the inputs instead of six on the results)
        const double p0 = valid ? r0[j] : 1.0, p1 = valid ? r1[j] : 0.0;  // linear GL
        const double p2 = valid ? r2[j] : 0.0, F = valid ? rF[j] : 0.0;
        // (at F = 1 the heterozygote's weight is the reference's exp(-1e15) = 0: so is the
        // product, p1 being finite)
        const double cc = p1 * fma(-2.0, F, 2.0);  // = 2 p1 (1 - F), the same bits, one op fewer
        const double n2 = (2 - F) * p2;
        sA[k] = p0;
        sb[k] = fma(F, p0 + p2, cc);
        sC[k] = p2;
        u0[k] = fma(n2, F, cc);
        nC[k] = n2;
        fc[k] = F * cc;
        tF_acc += valid ? 2 - F : 0.0;
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the next batch's loads out of this one
    }
    tF_lane_out = tF_acc;
  }
  double tF_sum = wave_sum_uniform(tF_lane_out);
  if constexpr (W > 1) {
    if (lane == 0) xch[1][wv][0] = tF_sum;
    __syncthreads();
    tF_sum = xch[1][0][0];
#pragma unroll
    for (int w = 1; w < W; ++w) tF_sum += xch[1][w][0];
    __syncthreads();
  }

  // The loop carries num and den only.  The odds of freq = num/den are num/(den - num):
  // one reciprocal on the serial path instead of two, and the reference's stopping rule
  // |prev - freq| > EPSILON (gen_func.cpp:1006) is tested cross-multiplied,
  // |num_prev den - num den_prev| > EPSILON den den_prev, which needs no quotient.
  int iters = 0;
  int buf = 0;
  double num = 0, den = 0;
  double pnum = 0.01, pden = 1.0;  // freq = 0.01 (gen_func.cpp:976)
  if (!fresh) {
    num = state[0 * state_stride + site];
    den = state[1 * state_stride + site];
    pnum = state[2 * state_stride + site];
    pden = state[3 * state_stride + site];
    iters = (int)state[4 * state_stride + site];
  }
  bool built = !allow_build;  // at most one interval per launch
  int n_before = n_exact;     // exact passes before deciding on it
  bool check = false, interp_ok = false;
  double mid = 0, half = 0, my_gn = 0, my_gd = 0, rprev = 0;  // mid: centre a, half: h (see the build)
  // this lane's part of the two per-pass sums at odds r
  auto lane_sums = [&](double r, double& pn, double& pd) {
    pn = 0;
    pd = 0;
    if constexpr (NI >= 4) {
      // reciprocals four at a time (Montgomery's trick): one v_rcp_f64 + Newton step and
      // 9 multiplies instead of four reciprocals; v_rcp_f64 is the slow instruction
#pragma unroll
      for (int k0 = 0; k0 + 4 <= NI; k0 += 4) {
        double sm[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sm[j] = fma(r, fma(r, sC[k0 + j], sb[k0 + j]), sA[k0 + j]);
        const double p01 = sm[0] * sm[1], p23 = sm[2] * sm[3];
        // a vanishing sum makes R infinite and the site's freq non-finite, which ends the
        // loop (the comparison below is false for NaN) and flags the site after it
        const double R = rcp_nr(p01 * p23);
        const double r01 = R * p23, r23 = R * p01;
        const double inv0 = r01 * sm[1], inv1 = r01 * sm[0];
        const double inv2 = r23 * sm[3], inv3 = r23 * sm[2];
        pn = fma(fma(nC[k0], r, u0[k0]), inv0, pn);
        pd = fma(fc[k0], inv0, pd);
        pn = fma(fma(nC[k0 + 1], r, u0[k0 + 1]), inv1, pn);
        pd = fma(fc[k0 + 1], inv1, pd);
        pn = fma(fma(nC[k0 + 2], r, u0[k0 + 2]), inv2, pn);
        pd = fma(fc[k0 + 2], inv2, pd);
        pn = fma(fma(nC[k0 + 3], r, u0[k0 + 3]), inv3, pn);
        pd = fma(fc[k0 + 3], inv3, pd);
      }
      if constexpr (NI % 4 == 2) {  // (10 or 14 individuals per lane: the last two share a reciprocal)
        constexpr int k0 = NI - 2;
        const double s0 = fma(r, fma(r, sC[k0], sb[k0]), sA[k0]);
        const double s1 = fma(r, fma(r, sC[k0 + 1], sb[k0 + 1]), sA[k0 + 1]);
        const double R = rcp_nr(s0 * s1);
        const double inv0 = R * s1, inv1 = R * s0;
        pn = fma(fma(nC[k0], r, u0[k0]), inv0, pn);
        pd = fma(fc[k0], inv0, pd);
        pn = fma(fma(nC[k0 + 1], r, u0[k0 + 1]), inv1, pn);
        pd = fma(fc[k0 + 1], inv1, pd);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const double inv = rcp_nr(fma(r, fma(r, sC[k], sb[k]), sA[k]));
        pn = fma(fma(nC[k], r, u0[k]), inv, pn);
        pd = fma(fc[k], inv, pd);
      }
    }
  };
  for (;;) {
    const double r = pnum * rcp_nr2(pden - pnum);
    double pn, pd;
    lane_sums(r, pn, pd);
    const double v = wave_sum_pair(pn, pd);
    double sn = lane_value(v, 31), sd = lane_value(v, 63);
    if constexpr (W > 1) {
      if (lane == 0) {
        xch[buf][wv][0] = sn;
        xch[buf][wv][1] = sd;
      }
      __syncthreads();
      sn = xch[buf][0][0];
      sd = xch[buf][0][1];
#pragma unroll
      for (int w = 1; w < W; ++w) {
        sn += xch[buf][w][0];
        sd += xch[buf][w][1];
      }
      buf ^= 1;
    }
    if (check) {  // the pass after a build: exact sums in hand, compare the interpolant
      const int nj = lane < EN ? lane : 0;
      const double t = (r - mid) / (r + mid) - half * kChebC[nj];
      const double q = (lane < EN) ? kChebW[nj] / t : 0.0;
      const double Sq = wave_sum_uniform(q);
      const double bn = wave_sum_uniform(q * my_gn) / Sq, bd = wave_sum_uniform(q * my_gd) / Sq;
      interp_ok = fabs(bn - sn) <= EST_TOL * fabs(sn) && fabs(bd - sd) <= EST_TOL * fabs(sd);
      check = false;
      if (!interp_ok && tix == 0) atomicAdd(cnt + EST_CNT_CHECK_FAILED, 1u);  // (rare)
    }
    num = fma(r, sn, num);
    den = fma(r, sd, den + tF_sum);
    const double lhs = fabs(fma(pnum, den, -(num * pden))), thr = kEPS * (den * pden);
    const bool again = (lhs > thr) && (iters++ < 100);
    rprev = r;
    pnum = num;
    pden = den;
    if (!again) break;
    if (interp_ok) {  // hand the site to k_fast_estmaf_interp
      if (wv == 0) {
        if (lane < EN) {
          state[(EST_SCALARS + lane) * state_stride + site] = my_gn;
          state[(EST_SCALARS + EN + lane) * state_stride + site] = my_gd;
        }
        if (lane == 0) {
          state[0 * state_stride + site] = num;
          state[1 * state_stride + site] = den;
          state[2 * state_stride + site] = pnum;
          state[3 * state_stride + site] = pden;
          state[4 * state_stride + site] = (double)iters;
          state[5 * state_stride + site] = mid;
          state[6 * state_stride + site] = half;
          state[7 * state_stride + site] = tF_sum;
          status[site] = EST_INTERP;
        }
      }
      return;
    }
    if (!built && --n_before <= 0) {
      // |delta freq| shrinks roughly like 1/k^2: about k (sqrt(|delta|/EPSILON) - 1)
      // passes remain, and the odds still travel about k times their last step.  An
      // interval costs EN evaluations, so short tails stay exact; and a site whose
      // remaining travel does not fit into one interval yet (a frequency far from the
      // 0.01 every site starts at) takes a few more exact passes first, rather than
      // leaving its interval half way and paying for a second one.
      const double m_est = (double)iters * (sqrt(lhs / thr) - 1.0);
      const double rn = pnum * rcp_nr2(pden - pnum);
      const double step = fabs(rn - rprev);
      // (an interval reaches EST_DMAX * r ahead when r grows, down to r / (1 + EST_DMAX) when
      // it shrinks: the same ratio both ways)
      const double reach = (rn >= rprev) ? EST_DMAX * rn : EST_DMAX / (1 + EST_DMAX) * rn;
      const bool fits = (double)iters * step <= EST_FIT * reach;
      if (!fits && iters < EST_KMAX && m_est >= EST_MIN_GAIN) {
        n_before = 1;  // look again after the next exact pass
      } else {
        built = true;
      }
      if (built && m_est >= EST_MIN_GAIN && 100 - iters >= EST_MIN_GAIN) {
        const double g = fmin(EST_DMAX, fmax(EST_MULT * step / rn, 1e-3));  // relative length
        double lo, hi;
        if (rn >= rprev) {
          lo = rn * (1 - EST_BACK * g);
          hi = rn * (1 + g);
        } else {
          lo = rn / (1 + g);
          hi = rn * (1 + EST_BACK * g);
        }
        // Interpolation variable t = (r - a) / (r + a), a = sqrt(lo hi): the half plane
        // Re r <= 0 that holds every pole of the sums is the OUTSIDE of the unit disc in t, and
        // [lo, hi] becomes [-h, h] around 0 -- far from everything, so the Chebyshev
        // interpolant on EN nodes converges like rho^-EN with (rho + 1/rho) / 2 = 1/h: an
        // interval of ratio hi / lo = 2 has rho = 11.7, where the same nodes in r itself
        // (nearest pole at distance >= lo from an interval of length lo) would have rho = 5.8.
        mid = sqrt(lo * hi);
        half = (hi - mid) / (hi + mid);
        // a degenerate interval (rn not finite or not positive) keeps the site exact
        if (half > 0 && lo > 0 && hi < 1e300) {
          if constexpr (PARK) {
            // One wave holds the site: the node evaluations do not depend on each other,
            // so every lane parks its partial sums in LDS and the 16 x 64 partials are
            // added up once at the end -- no reduction tree (and its latency) per node.
            // Lane q*16 + j adds quarter q of node j's partials; two shuffles join the
            // quarters.
            // the nodes' odds: lane nd forms node nd's once, the loop reads them lane by lane
            // (a reciprocal and its Newton steps per node otherwise)
            const double tnl = half * kChebC[lane < EN ? lane : 0];
            const double r_nodes = mid * (1 + tnl) * rcp_nr2(1 - tnl);
#pragma unroll 1
            for (int nd = 0; nd < EN; ++nd) {
              double pn, pd;
              lane_sums(lane_value(r_nodes, nd), pn, pd);
              nodebuf[nd][lane] = double2{pn, pd};
            }
            __syncthreads();  // one wave: orders the LDS writes before the reads
            const int j = lane & 15, q4 = lane >> 4;
            double an = 0, ad = 0;
            if (j < EN) {
#pragma unroll
              for (int l = 0; l < 16; ++l) {
                const double2 t2 = nodebuf[j][q4 * 16 + l];
                an += t2.x;
                ad += t2.y;
              }
            }
            an += __shfl_xor(an, 16);
            ad += __shfl_xor(ad, 16);
            an += __shfl_xor(an, 32);
            ad += __shfl_xor(ad, 32);
            my_gn = an;  // lanes 0..EN-1 hold node `lane` (every quarter has the total)
            my_gd = ad;
            check = true;
          } else {
            const double tnl = half * kChebC[lane < EN ? lane : 0];
            const double r_nodes = mid * (1 + tnl) * rcp_nr2(1 - tnl);
            double an = 0, ad = 0;
            if constexpr (W > 1) {
              // several waves per site: as in the one-wave kernel no node is reduced on its own --
              // every thread parks its partial sums of the EN nodes in LDS (park[node][thread], the
              // kernel's dynamic shared memory), and wave w then adds up the nodes w, w + W, ...:
              // a lane the W waves' partials of its lane index in wave order, one reduction tree per
              // node instead of one per node AND wave (round 6: 360 -> 50-160 instructions per wave
              // and site)
#pragma unroll 1
              for (int nd = 0; nd < EN; ++nd) {
                double pn, pd;
                lane_sums(lane_value(r_nodes, nd), pn, pd);
                park[nd * BLOCK + (int)tix] = double2{pn, pd};
              }
              __syncthreads();
              for (int nd = wv; nd < EN; nd += W) {
                double sn = 0, sd = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                  const double2 t2 = park[nd * BLOCK + w * 64 + lane];
                  sn += t2.x;
                  sd += t2.y;
                }
                const double v = wave_sum_pair(sn, sd);
                const double tn = lane_value(v, 31), td = lane_value(v, 63);
                if (lane == 0) xnode[nd][0] = double2{tn, td};
              }
              __syncthreads();
              if (lane < EN) {
                an = xnode[lane][0].x;
                ad = xnode[lane][0].y;
              }
              __syncthreads();  // (the shared buffers serve the passes that follow)
            } else {
              // one wave, few individuals per lane (NI < 8): every node reduced in registers
#pragma unroll 1
              for (int nd = 0; nd < EN; ++nd) {
                double pn, pd;
                lane_sums(lane_value(r_nodes, nd), pn, pd);
                const double v = wave_sum_pair(pn, pd);
                const double sn = lane_value(v, 31), sd = lane_value(v, 63);
                if (lane == nd) {
                  an = sn;
                  ad = sd;
                }
              }
            }
            my_gn = an;
            my_gd = ad;
            check = true;
          }
        }
      }
    }
  }
  if (tix == 0) {
    // non-finite or out-of-range result: a cell with vanishing weights (or f reaching 1);
    // the careful kernel redoes the site in the reference's log-space order
    const double freq = num / den;
    const bool ok = freq >= 0 && freq < 1;
    freq_out[site] = freq;
    redo[site] = ok ? 0 : 1;
    status[site] = EST_DONE;
  }
}

// tile-major posteriors: where site (c*64 + l)*T + t finds individual i at col[i * 64]
__device__ __forceinline__ const double* estmaf_tile_col(const double* marg_blocks, uint64_t site,
                                                         uint64_t tile_T, uint64_t I_tot) {
  const uint64_t j = site / tile_T, t = site - j * tile_T;
  return marg_blocks + post_lane_off((j >> 6) * tile_T + t, j & 63, I_tot);
}

#define ESTMAF_SHARED(NI, BLOCK)                                                              \
  constexpr bool PARK_ = ((BLOCK) == 64 && (NI) >= 8);                                         \
  __shared__ double xch[2][ESTMAF_MAXW][2];                      /* [buffer][wave][num, den] */ \
  __shared__ double2 nodebuf[PARK_ ? EN : 1][PARK_ ? 65 : 1];                                  \
  __shared__ double2 xnode[PARK_ ? 1 : EN][PARK_ ? 1 : (BLOCK) / 64] /* !PARK: per-wave node sums */

// every site from the start (freq = 0.01): one workgroup per site
template <int NI, int BLOCK, bool TILE>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2)))
k_fast_estmaf(const GlView gl, const double* __restrict__ marg_blocks,
              uint64_t S_own, uint64_t I_tot, uint64_t I_blk, uint64_t tile_T,
              double* __restrict__ freq_out, uint8_t* __restrict__ redo,
              uint8_t* __restrict__ status, double* __restrict__ state, uint64_t state_stride,
              int n_exact, int allow_build, uint64_t blk0, uint32_t* __restrict__ cnt) {
  // W == 1: per-lane partial sums of the interval's nodes (see the build); the pad makes lane
  // j's reads of row j conflict-free (few individuals per lane, NI < 8: the 16 KB would cap
  // the waves per CU for nothing -- those kernels reduce every node in registers like the
  // multi-wave ones)
  ESTMAF_SHARED(NI, BLOCK);
  uint64_t site;
  const double* tile_col = nullptr;  // TILE: posterior of individual i at tile_col[post_ind_off(i)]
  if constexpr (TILE) {
    // blk0: first block of the launch's part of the grid (whole tile rows: blk0 % 64 == 0)
    const uint64_t b = blockIdx.x + blk0, x = b & 7, k = b >> 3;
    const uint64_t q = ((k >> 3) << 6) + (x << 3) + (k & 7);
    const uint64_t tile_row = q >> 6, l = q & 63;  // tile_row = c * T + t
    const uint64_t c = tile_row / tile_T, t = tile_row - c * tile_T;
    site = (c * 64 + l) * tile_T + t;
    if (site >= S_own) return;  // padding of the interleaved layout
    tile_col = marg_blocks + post_lane_off(tile_row, l, I_tot);
  } else {
    site = blockIdx.x;
  }
  extern __shared__ double2 park_mem[];  // BLOCK > 64: [EN][BLOCK] partial node sums (estmaf_site)
  estmaf_site<NI, BLOCK, TILE>(gl, marg_blocks, S_own, I_tot, I_blk, freq_out, redo, status, state,
                               state_stride, 1, n_exact, allow_build, site, tile_col, cnt, xch, nodebuf,
                               xnode, park_mem);
}

// The sites k_fast_estmaf_interp handed back (status EST_EXACT) resume from `state`.  They are
// few: instead of a workgroup per site that finds nothing to do, a workgroup reads the status
// of 64 sites at a time and takes the flagged ones in turn (tile rows [row0, row1) only).
template <int NI, int BLOCK, bool TILE>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2)))
k_fast_estmaf_resume(const GlView gl, const double* __restrict__ marg_blocks,
                     uint64_t S_own, uint64_t I_tot, uint64_t I_blk, uint64_t tile_T,
                     double* __restrict__ freq_out, uint8_t* __restrict__ redo,
                     uint8_t* __restrict__ status, double* __restrict__ state,
                     uint64_t state_stride, int n_exact, int allow_build, uint64_t row0,
                     uint64_t row1, uint32_t* __restrict__ cnt, int cnt_slot) {
  ESTMAF_SHARED(NI, BLOCK);
  extern __shared__ double2 park_mem[];
  const int lane = threadIdx.x & 63;
  for (uint64_t base = (uint64_t)blockIdx.x * 64; base < S_own; base += (uint64_t)gridDim.x * 64) {
    const uint64_t s = base + lane;
    const bool need = s < S_own && in_tile_rows(s, TILE ? tile_T : 0, row0, row1) &&
                      status[s] == EST_EXACT;
    uint64_t mask = __ballot(need);  // the same in every wave of the workgroup
    if (mask && threadIdx.x == 0) atomicAdd(cnt + cnt_slot, (uint32_t)__popcll(mask));
    if constexpr (BLOCK > 64) __syncthreads();  // ... all have read before anyone writes a status
    while (mask) {
      const uint64_t site = base + (uint64_t)__builtin_ctzll(mask);
      mask &= mask - 1;
      const double* tile_col = TILE ? estmaf_tile_col(marg_blocks, site, tile_T, I_tot) : nullptr;
      estmaf_site<NI, BLOCK, TILE>(gl, marg_blocks, S_own, I_tot, I_blk, freq_out, redo, status,
                                   state, state_stride, 0, n_exact, allow_build, site, tile_col, cnt,
                                   xch, nodebuf, xnode, park_mem);
      if constexpr (BLOCK > 64) __syncthreads();  // the shared buffers serve the next site
    }
  }
}

// Small cohorts: a 64-lane wave per site leaves most lanes empty below ~128 individuals and
// pays the per-pass bookkeeping for one site only.  Here a wave holds FOUR sites, one per
// 16-lane DPP row (individual i of the site in lane i % 16, slot i / 16; up to 16 NI = 128
// individuals), and every reduction is the four in-row DPP steps, which leave the row's total
// in all of its lanes -- no cross-lane reads, no LDS.  Same recursion, same interval logic,
// same hand-over to k_fast_estmaf_interp as k_fast_estmaf; the rows of a wave run their own
// sites independently (a row whose site is finished idles).
// The lanes of a row take their decisions (stop, hand over, build) each for itself from these
// totals, so the totals must be the SAME BITS in all 16 lanes: every step adds the two
// partners' values, a + b in one lane and b + a in the other.  That only holds for plain
// additions -- were the first one contracted with a multiplication that produced the argument
// (fma(x, y, partner's rounded x'y') here, fma(x', y', rounded xy) there) the partners would
// differ in the last bit, and a row's lanes would part ways at a threshold.  __dadd_rn is
// never contracted.
__device__ __forceinline__ double row_sum(double v) {
  v = __dadd_rn(v, dpp_move<0xB1>(v));   // quad_perm [1,0,3,2]
  v = __dadd_rn(v, dpp_move<0x4E>(v));   // quad_perm [2,3,0,1]
  v = __dadd_rn(v, dpp_move<0x141>(v));  // row_half_mirror
  v = __dadd_rn(v, dpp_move<0x140>(v));  // row_mirror: every lane holds its 16-lane row total
  return v;
}

// four sites on the four 16-lane rows of a wave; a row with done = true idles (site and
// tile_col must still be readable)
template <int NI, bool TILE>
__device__ __forceinline__ void estmaf_rows_sites(
    const GlView& gl, const double* __restrict__ marg_blocks, uint64_t S_own, uint64_t I_tot,
    uint64_t I_blk, double* __restrict__ freq_out, uint8_t* __restrict__ redo,
    uint8_t* __restrict__ status, double* __restrict__ state, uint64_t state_stride, int fresh,
    int n_exact, int allow_build, uint64_t site, const double* __restrict__ tile_col, bool done,
    uint32_t* __restrict__ cnt) {
  static_assert(EN <= 16, "a row's lanes hold the interval's node sums");
  // Control flow is kept WAVE-UNIFORM: the rows of a wave are at different points of their
  // recursions (one hands its site over while another still needs exact passes), but every
  // DPP reduction runs with all 64 lanes enabled -- a finished row computes along on its stale
  // values and ignores the results -- and the per-row decisions are applied under `!done`.
  const int lane = threadIdx.x, j = lane & 15;
  const uint64_t cell_s = site * I_tot;

  double sA[NI], sb[NI], sC[NI], u0[NI], nC[NI], fc[NI];
  double tF_acc = 0;
  {
    const bool one_block = (I_blk == I_tot);
    const uint32_t ib = (uint32_t)I_blk;
    double r0[NI], r1[NI], r2[NI], rF[NI];
    uint64_t ic[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const uint64_t i = (uint64_t)j + 16 * k;
      ic[k] = i < I_tot ? i : I_tot - 1;
      gl_fetch(gl, cell_s + ic[k], r0[k], r1[k], r2[k]);
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      if constexpr (TILE) {
        rF[k] = tile_col[post_ind_off(ic[k])];
      } else if (one_block) {
        rF[k] = marg_blocks[site * I_blk + ic[k]];
      } else {
        const uint32_t q = (uint32_t)ic[k] / ib;
        rF[k] = marg_blocks[((uint64_t)q * S_own + site) * I_blk + ((uint32_t)ic[k] - q * ib)];
      }
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const bool valid = (uint64_t)j + 16 * k < I_tot;
      const double p0 = valid ? r0[k] : 1.0, p1 = valid ? r1[k] : 0.0;  // (see estmaf_site)
      const double p2 = valid ? r2[k] : 0.0, F = valid ? rF[k] : 0.0;
      const double cc = p1 * fma(-2.0, F, 2.0);  // 2 p1 (1 - F): 0 at F = 1, as the reference's exp(-1e15)
      const double n2 = (2 - F) * p2;
      sA[k] = p0;
      sb[k] = fma(F, p0 + p2, cc);
      sC[k] = p2;
      u0[k] = fma(n2, F, cc);
      nC[k] = n2;
      fc[k] = F * cc;
      tF_acc += valid ? 2 - F : 0.0;
    }
  }
  const double tF_sum = row_sum(tF_acc);

  int iters = 0;
  double num = 0, den = 0;
  double pnum = 0.01, pden = 1.0;  // freq = 0.01 (gen_func.cpp:976)
  if (!fresh && !done) {
    num = state[0 * state_stride + site];
    den = state[1 * state_stride + site];
    pnum = state[2 * state_stride + site];
    pden = state[3 * state_stride + site];
    iters = (int)state[4 * state_stride + site];
  }
  bool built = !allow_build;
  int n_before = n_exact;
  bool check = false;
  double mid = 1, half = 0.5, my_gn = 0, my_gd = 0, rprev = 0;
  auto lane_sums = [&](double r, double& pn, double& pd) {
    pn = 0;
    pd = 0;
    if constexpr (NI >= 4) {
#pragma unroll
      for (int k0 = 0; k0 < NI; k0 += 4) {
        double sm[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sm[k] = fma(r, fma(r, sC[k0 + k], sb[k0 + k]), sA[k0 + k]);
        const double p01 = sm[0] * sm[1], p23 = sm[2] * sm[3];
        const double R = rcp_nr(p01 * p23);
        const double r01 = R * p23, r23 = R * p01;
        const double inv0 = r01 * sm[1], inv1 = r01 * sm[0];
        const double inv2 = r23 * sm[3], inv3 = r23 * sm[2];
        pn = fma(fma(nC[k0], r, u0[k0]), inv0, pn);
        pd = fma(fc[k0], inv0, pd);
        pn = fma(fma(nC[k0 + 1], r, u0[k0 + 1]), inv1, pn);
        pd = fma(fc[k0 + 1], inv1, pd);
        pn = fma(fma(nC[k0 + 2], r, u0[k0 + 2]), inv2, pn);
        pd = fma(fc[k0 + 2], inv2, pd);
        pn = fma(fma(nC[k0 + 3], r, u0[k0 + 3]), inv3, pn);
        pd = fma(fc[k0 + 3], inv3, pd);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NI; ++k) {
        const double inv = rcp_nr(fma(r, fma(r, sC[k], sb[k]), sA[k]));
        pn = fma(fma(nC[k], r, u0[k]), inv, pn);
        pd = fma(fc[k], inv, pd);
      }
    }
  };
  for (;;) {
    // ---- one exact pass of every row (all lanes) ----
    const double r = pnum * rcp_nr2(pden - pnum);
    double pn, pd;
    lane_sums(r, pn, pd);
    const double sn = row_sum(pn), sd = row_sum(pd);
    bool interp_ok = false;
    if (__builtin_amdgcn_ballot_w64(check && !done) != 0) {
      // the pass after a build: exact sums in hand, compare the interpolant (rows that did
      // not just build compute along and ignore the outcome)
      const int nj = j < EN ? j : 0;
      const double t = (r - mid) / (r + mid) - half * kChebC[nj];
      const double q = (j < EN) ? kChebW[nj] / t : 0.0;
      const double Sq = row_sum(q);
      const double bn = row_sum(q * my_gn) / Sq, bd = row_sum(q * my_gd) / Sq;
      if (check && !done) {
        interp_ok = fabs(bn - sn) <= EST_TOL * fabs(sn) && fabs(bd - sd) <= EST_TOL * fabs(sd);
        if (!interp_ok && j == 0) atomicAdd(cnt + EST_CNT_CHECK_FAILED, 1u);  // (rare)
      }
      check = false;
    }
    // ---- the recursion and the row's decisions ----
    bool want_build = false;
    if (!done) {
      num = fma(r, sn, num);
      den = fma(r, sd, den + tF_sum);
      const double lhs = fabs(fma(pnum, den, -(num * pden))), thr = kEPS * (den * pden);
      const bool again = (lhs > thr) && (iters++ < 100);
      rprev = r;
      pnum = num;
      pden = den;
      if (!again) {
        if (j == 0) {
          const double freq = num / den;
          const bool ok = freq >= 0 && freq < 1;
          freq_out[site] = freq;
          redo[site] = ok ? 0 : 1;
          status[site] = EST_DONE;
        }
        done = true;
      } else if (interp_ok) {  // hand the site to k_fast_estmaf_interp
        if (j < EN) {
          state[(EST_SCALARS + j) * state_stride + site] = my_gn;
          state[(EST_SCALARS + EN + j) * state_stride + site] = my_gd;
        }
        if (j == 0) {
          state[0 * state_stride + site] = num;
          state[1 * state_stride + site] = den;
          state[2 * state_stride + site] = pnum;
          state[3 * state_stride + site] = pden;
          state[4 * state_stride + site] = (double)iters;
          state[5 * state_stride + site] = mid;
          state[6 * state_stride + site] = half;
          state[7 * state_stride + site] = tF_sum;
          status[site] = EST_INTERP;
        }
        done = true;
      } else if (!built && --n_before <= 0) {  // see k_fast_estmaf for the reasoning
        const double m_est = (double)iters * (sqrt(lhs / thr) - 1.0);
        const double rn = pnum * rcp_nr2(pden - pnum);
        const double step = fabs(rn - rprev);
        const double reach = (rn >= rprev) ? EST_DMAX * rn : EST_DMAX / (1 + EST_DMAX) * rn;
        const bool fits = (double)iters * step <= EST_FIT * reach;
        if (!fits && iters < EST_KMAX && m_est >= EST_MIN_GAIN) {
          n_before = 1;
        } else {
          built = true;
        }
        if (built && m_est >= EST_MIN_GAIN && 100 - iters >= EST_MIN_GAIN) {
          const double g = fmin(EST_DMAX, fmax(EST_MULT * step / rn, 1e-3));
          double lo, hi;
          if (rn >= rprev) {
            lo = rn * (1 - EST_BACK * g);
            hi = rn * (1 + g);
          } else {
            lo = rn / (1 + g);
            hi = rn * (1 + EST_BACK * g);
          }
          const double a = sqrt(lo * hi), h = (hi - a) / (hi + a);
          if (h > 0 && lo > 0 && hi < 1e300) {
            mid = a;
            half = h;
            want_build = true;
          }
        }
      }
    }
    // ---- the interval's node sums, for the rows that build (all lanes compute) ----
    if (__builtin_amdgcn_ballot_w64(want_build) != 0) {
#pragma unroll 1
      for (int nd = 0; nd < EN; ++nd) {
        const double tn = half * kChebC[nd];
        double qn, qd;
        lane_sums(mid * (1 + tn) * rcp_nr2(1 - tn), qn, qd);
        const double gn = row_sum(qn), gd = row_sum(qd);
        if (want_build && j == nd) {  // lane nd of the row keeps node nd
          my_gn = gn;
          my_gd = gd;
        }
      }
      if (want_build) check = true;
    }
    if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
  }
}

template <int NI, bool TILE>
__global__ void __launch_bounds__(64)
k_fast_estmaf_rows(const GlView gl, const double* __restrict__ marg_blocks, uint64_t S_own,
                   uint64_t I_tot, uint64_t I_blk, uint64_t tile_T, double* __restrict__ freq_out,
                   uint8_t* __restrict__ redo, uint8_t* __restrict__ status,
                   double* __restrict__ state, uint64_t state_stride, int n_exact,
                   int allow_build, uint32_t* __restrict__ cnt) {
  const int lane = threadIdx.x, row = lane >> 4;
  uint64_t site;
  const double* tile_col = nullptr;
  if constexpr (TILE) {
    // as in k_fast_estmaf, XCD x gets the sites l = 8x..8x+7 of a tile row -- here in two
    // consecutive workgroups of four sites each
    const uint64_t b = blockIdx.x, x = b & 7, k = b >> 3;
    const uint64_t q = ((k >> 1) << 6) + (x << 3) + ((k & 1) << 2) + row;
    const uint64_t tile_row = q >> 6, l = q & 63;
    const uint64_t c = tile_row / tile_T, t = tile_row - c * tile_T;
    site = (c * 64 + l) * tile_T + t;
    tile_col = marg_blocks + post_lane_off(tile_row, l, I_tot);
  } else {
    site = (uint64_t)blockIdx.x * 4 + row;
  }
  bool done = site >= S_own;          // padding of the layout / past the end
  if (done) {                         // read something valid, write nothing
    site = 0;
    if constexpr (TILE) tile_col = marg_blocks;
  }
  if (__builtin_amdgcn_ballot_w64(!done) == 0) return;  // nothing to do in this wave
  estmaf_rows_sites<NI, TILE>(gl, marg_blocks, S_own, I_tot, I_blk, freq_out, redo, status, state,
                              state_stride, 1, n_exact, allow_build, site, tile_col, done, cnt);
}

// resuming sites (see k_fast_estmaf_resume): the wave reads 64 statuses at a time and gives the
// flagged sites to its rows four at a time
template <int NI, bool TILE>
__global__ void __launch_bounds__(64)
k_fast_estmaf_rows_resume(const GlView gl, const double* __restrict__ marg_blocks, uint64_t S_own,
                          uint64_t I_tot, uint64_t I_blk, uint64_t tile_T,
                          double* __restrict__ freq_out, uint8_t* __restrict__ redo,
                          uint8_t* __restrict__ status, double* __restrict__ state,
                          uint64_t state_stride, int n_exact, int allow_build,
                          uint32_t* __restrict__ cnt, int cnt_slot) {
  const int lane = threadIdx.x, row = lane >> 4;
  for (uint64_t base = (uint64_t)blockIdx.x * 64; base < S_own; base += (uint64_t)gridDim.x * 64) {
    const uint64_t s = base + lane;
    uint64_t mask = __ballot(s < S_own && status[s] == EST_EXACT);
    if (mask && lane == 0) atomicAdd(cnt + cnt_slot, (uint32_t)__popcll(mask));
    while (mask) {
      uint64_t site = 0;
      bool done = true;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (mask) {
          const uint64_t cand = base + (uint64_t)__builtin_ctzll(mask);
          mask &= mask - 1;
          if (r == row) {
            site = cand;
            done = false;
          }
        }
      }
      const double* tile_col =
          TILE ? (done ? marg_blocks : estmaf_tile_col(marg_blocks, site, tile_T, I_tot)) : nullptr;
      estmaf_rows_sites<NI, TILE>(gl, marg_blocks, S_own, I_tot, I_blk, freq_out, redo, status,
                                  state, state_stride, 0, n_exact, allow_build, site, tile_col, done,
                                  cnt);
    }
  }
}

// The passes between a checked interval and either the end of the loop or the point
// (a launch may cover only the tile rows [row0, row1) of the E-step's layout: see fast_estmaf)
// where exact evaluation is needed again: one lane per site.  The EN node values of each sum
// become Chebyshev coefficients once (c_k = 2/EN sum_j f_j T_k(x_j), the T_k by their
// recurrence: 3 EN^2 instructions), and a pass is then two Clenshaw recurrences of EN steps
// -- no divisions, where the barycentric formula spends one per node and pass (the ~80 passes
// of a site cost 3x less; the checked interpolant is the same polynomial).
__global__ void __launch_bounds__(256)
k_fast_estmaf_interp(uint64_t S_own, double* __restrict__ freq_out, uint8_t* __restrict__ redo,
                     uint8_t* __restrict__ status, double* __restrict__ state,
                     uint64_t state_stride, uint64_t tile_T, uint64_t row0, uint64_t row1) {
  const uint64_t site = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (site >= S_own || !in_tile_rows(site, tile_T, row0, row1) || status[site] != EST_INTERP)
    return;
  double num = state[0 * state_stride + site], den = state[1 * state_stride + site];
  double pnum = state[2 * state_stride + site], pden = state[3 * state_stride + site];
  int iters = (int)state[4 * state_stride + site];
  const double mid = state[5 * state_stride + site], half = state[6 * state_stride + site];
  const double tF_sum = state[7 * state_stride + site];
  double gn[EN], gd[EN];
#pragma unroll
  for (int j = 0; j < EN; ++j) {
    gn[j] = state[(EST_SCALARS + j) * state_stride + site];
    gd[j] = state[(EST_SCALARS + EN + j) * state_stride + site];
  }
  // node values -> Chebyshev coefficients (in place would need a second array anyway)
  double cn[EN], cd[EN];
#pragma unroll
  for (int k = 0; k < EN; ++k) cn[k] = cd[k] = 0;
#pragma unroll
  for (int j = 0; j < EN; ++j) {
    const double x = kChebC[j], fn = gn[j] * (2.0 / EN), fd = gd[j] * (2.0 / EN);
    double t0 = 1.0, t1 = x;
    cn[0] += fn;
    cd[0] += fd;
#pragma unroll
    for (int k = 1; k < EN; ++k) {
      cn[k] = fma(fn, t1, cn[k]);
      cd[k] = fma(fd, t1, cd[k]);
      const double t2 = fma(2 * x, t1, -t0);
      t0 = t1;
      t1 = t2;
    }
  }
  const double inv_half = 1.0 / half;
  uint8_t st = EST_EXACT;
  for (;;) {
    const double r = pnum * rcp_nr2(pden - pnum);  // the expression of k_fast_estmaf
    const double tt = (r - mid) * rcp_nr2(r + mid);  // the interpolation variable (see the build)
    if (!(fabs(tt) <= half)) break;                // left the interval (or not finite)
    // Clenshaw: p(x) = c_0 / 2 + sum_{k >= 1} c_k T_k(x),  x = tt / half in [-1, 1]
    const double x2 = 2 * (tt * inv_half);
    double bn1 = 0, bn2 = 0, bd1 = 0, bd2 = 0;
#pragma unroll
    for (int k = EN - 1; k >= 1; --k) {
      const double bn0 = fma(x2, bn1, cn[k] - bn2), bd0 = fma(x2, bd1, cd[k] - bd2);
      bn2 = bn1;
      bn1 = bn0;
      bd2 = bd1;
      bd1 = bd0;
    }
    const double sn = fma(0.5 * x2, bn1, 0.5 * cn[0] - bn2);
    const double sd = fma(0.5 * x2, bd1, 0.5 * cd[0] - bd2);
    const double num2 = fma(r, sn, num), den2 = fma(r, sd, den + tF_sum);
    const double lhs = fabs(fma(pnum, den2, -(num2 * pden))), thr = kEPS * (den2 * pden);
    if (!(fabs(lhs - thr) >= EST_GUARD * thr)) break;  // too close to call (or not finite)
    num = num2;
    den = den2;
    const bool again = (lhs > thr) && (iters++ < 100);
    pnum = num;
    pden = den;
    if (!again) {
      st = EST_DONE;
      break;
    }
  }
  if (st == EST_DONE) {
    const double freq = num / den;
    const bool ok = freq >= 0 && freq < 1;
    freq_out[site] = freq;
    redo[site] = ok ? 0 : 1;
  } else {
    state[0 * state_stride + site] = num;
    state[1 * state_stride + site] = den;
    state[2 * state_stride + site] = pnum;
    state[3 * state_stride + site] = pden;
    state[4 * state_stride + site] = (double)iters;
  }
  status[site] = st;
}

// ---- est_maf for CALLED genotypes (packed handles): the per-pass sums in closed form ----
// A called genotype (--call_geno, called-genotype input: gen_func.cpp:886-914,
// read_data.cpp:88-98) has linear likelihoods (1,0,0), (0,1,0), (0,0,1) or -- missing -- (u,u,u),
// so the genotype posterior of est_maf's pass (calc_HWE + post_prob, gen_func.cpp:984-1000) is
// a unit vector whatever the frequency, except for missing cells, where it is HWE itself:
//   genotype 0:  num += 0         den += 2 - F
//   genotype 1:  num += 1         den += 2                 (F < 1; at F = 1 the reference's weights
//                                                           all vanish: the site is redone in its
//                                                           log-space order, k_fast_estmaf_stream)
//   genotype 2:  num += 2 - F     den += 2 - F
//   missing:     num += h1 + h2 (2 - F),  den += 2 h1 + (h0 + h2)(2 - F),  with
//                h0 = (1-f)^2 + f(1-f)F, h1 = 2 f(1-f)(1-F), h2 = f^2 + f(1-f)F -- polynomials
//                in f whose coefficients are sums over the site's missing individuals of
//                (1-F), (2-F) and F(2-F).
// The whole <= 101-pass recursion of a site is therefore a SCALAR recursion on five sums over
// its individuals: one sweep over the 2-bit codes and the posteriors (8.25 B per cell, bound by
// HBM) leaves the sums, and one LANE per site runs the reference's passes -- same recursion,
// same pass count, same stopping rule as k_fast_estmaf, nothing interpolated.  (The general
// kernel spends 17 evaluations of every individual per site, 17.5 ps per cell at 5000
// individuals; this is one load of every cell.)
constexpr int ESTC_A0 = 0, ESTC_B0 = 1, ESTC_M0 = 2, ESTC_M1 = 3, ESTC_M2 = 4;

template <bool TILE>
__global__ void __launch_bounds__(64)
k_fast_estmaf_called_sums(const GlView gl, const double* __restrict__ marg_blocks, uint64_t S_own,
                          uint64_t I_tot, uint64_t I_blk, uint64_t tile_T,
                          uint8_t* __restrict__ redo, double* __restrict__ state,
                          uint64_t state_stride, uint64_t blk0) {
  const int lane = threadIdx.x;
  uint64_t site;
  const double* tile_col = nullptr;
  if constexpr (TILE) {  // the blockIdx -> site map of k_fast_estmaf<.., TILE>
    const uint64_t b = blockIdx.x + blk0, x = b & 7, k = b >> 3;
    const uint64_t q = ((k >> 3) << 6) + (x << 3) + (k & 7);
    const uint64_t tile_row = q >> 6, l = q & 63;
    const uint64_t c = tile_row / tile_T, t = tile_row - c * tile_T;
    site = (c * 64 + l) * tile_T + t;
    if (site >= S_own) return;
    tile_col = marg_blocks + post_lane_off(tile_row, l, I_tot);
  } else {
    site = blockIdx.x;
  }
  const uint64_t cell_s = gl.cell0 + site * I_tot;
  const bool one_block = (I_blk == I_tot);
  double n1 = 0, s2 = 0, s02 = 0, m0 = 0, m1 = 0, m2 = 0;
  bool bad = false;
  auto posterior = [&](uint64_t i) -> double {
    if constexpr (TILE) return tile_col[post_ind_off(i)];
    if (one_block) return marg_blocks[site * I_blk + i];
    const uint64_t q = i / I_blk;  // rank blocks [I_tot / I_blk][S_own][I_blk]
    return marg_blocks[(q * S_own + site) * I_blk + (i - q * I_blk)];
  };
  auto take = [&](uint32_t code, double F) {
    const double tF = 2 - F;
    const bool g1 = code == 1, g2 = code == 2, g3 = code == 3;
    n1 += g1 ? 1.0 : 0.0;
    bad |= g1 && !(F < 1);
    s2 += g2 ? tF : 0.0;
    s02 += (code == 0 || g2) ? tF : 0.0;
    m0 += g3 ? 1 - F : 0.0;
    m1 += g3 ? tF : 0.0;
    m2 += g3 ? F * tF : 0.0;
  };
  uint64_t i = lane;
  for (; i + 192 < I_tot; i += 256) {  // four loads of each kind in flight
    double F[4];
    uint32_t cd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      F[j] = posterior(i + 64 * j);
      cd[j] = gl_code(gl.codes, cell_s + i + 64 * j);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) take(cd[j], F[j]);
  }
  for (; i < I_tot; i += 64) take(gl_code(gl.codes, cell_s + i), posterior(i));
  const double A0 = wave_sum(n1 + s2), B0 = wave_sum(2 * n1 + s02);
  m0 = wave_sum(m0);
  m1 = wave_sum(m1);
  m2 = wave_sum(m2);
  const bool any_bad = __ballot(bad) != 0;
  if (lane == 0) {
    state[ESTC_A0 * state_stride + site] = A0;
    state[ESTC_B0 * state_stride + site] = B0;
    state[ESTC_M0 * state_stride + site] = m0;
    state[ESTC_M1 * state_stride + site] = m1;
    state[ESTC_M2 * state_stride + site] = m2;
    redo[site] = any_bad ? 1 : 0;
  }
}

// the passes themselves (gen_func.cpp:976-1006): one lane per site
__global__ void __launch_bounds__(256)
k_fast_estmaf_called_passes(uint64_t S_own, double* __restrict__ freq_out, uint8_t* __restrict__ redo,
                            uint8_t* __restrict__ status, const double* __restrict__ state,
                            uint64_t state_stride, uint64_t tile_T, uint64_t row0, uint64_t row1) {
  const uint64_t site = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (site >= S_own || !in_tile_rows(site, tile_T, row0, row1)) return;
  status[site] = EST_DONE;
  if (redo[site]) return;
  const double A0 = state[ESTC_A0 * state_stride + site], B0 = state[ESTC_B0 * state_stride + site];
  const double m0 = state[ESTC_M0 * state_stride + site], m1 = state[ESTC_M1 * state_stride + site];
  const double m2 = state[ESTC_M2 * state_stride + site];
  int iters = 0;
  double num = 0, den = 0, pnum = 0.01, pden = 1.0;  // freq = 0.01 (gen_func.cpp:976)
  for (;;) {
    const double f = pnum / pden, om = 1 - f;
    const double b = f * om, ff = f * f;
    const double miss_n = fma(2 * b, m0, fma(ff, m1, b * m2));
    const double miss_d = fma(4 * b, m0, fma(fma(om, om, ff), m1, 2 * b * m2));
    num += A0 + miss_n;
    den += B0 + miss_d;
    // |prev - freq| > EPSILON (gen_func.cpp:1006), cross-multiplied as in k_fast_estmaf
    const double lhs = fabs(fma(pnum, den, -(num * pden))), thr = kEPS * (den * pden);
    const bool again = (lhs > thr) && (iters++ < 100);
    pnum = num;
    pden = den;
    if (!again) break;
  }
  const double freq = num / den;
  const bool ok = freq >= 0 && freq < 1;
  freq_out[site] = freq;
  redo[site] = ok ? 0 : 1;
}

// any number of individuals: re-reads the (L2-resident) site row every pass.  One wave per
// site; a wave looks at the flags of 64 sites at a time (normally none is set: the launch is
// then a few thousand waves reading a cache line each, whatever the number of sites) and
// takes the flagged ones in turn.
__device__ void estmaf_stream_site(const GlView& gl, const double* __restrict__ marg_blocks,
                                   uint64_t S_own, uint64_t I_tot, uint64_t I_blk, uint64_t tile_T,
                                   double* __restrict__ freq_out, uint64_t site, int lane) {
  const uint64_t cell_s = site * I_tot;
  // tile_T != 0: posteriors in the tile-major layout (see k_fast_estmaf<.., TILE>)
  const uint64_t tj = tile_T ? site / tile_T : 0;  // lane-chunk c*64 + l; t = site - tj*T
  const double* trow =
      tile_T ? marg_blocks + post_lane_off((tj >> 6) * tile_T + (site - tj * tile_T), tj & 63, I_tot)
             : nullptr;
  int iters = 0;
  double num = 0, den = 0, freq = 0.01, prev;
  bool again;
  do {
    prev = freq;
    const double om = 1 - freq;
    const double b = om * freq;
    const double A = om * om, Cq = freq * freq;
    double pn = 0, pd = 0;
    for (uint64_t i = lane; i < I_tot; i += 64) {
      const double F = trow ? trow[post_ind_off(i)]
                            : marg_blocks[((i / I_blk) * S_own + site) * I_blk + (i % I_blk)];
      const double bF = b * F;
      const double h0 = A + bF, h2 = Cq + bF;
      const double h1 = (F == 1) ? 0.0 : (2 * b - 2 * bF);
      double p0, p1, p2;
      gl_fetch(gl, cell_s + i, p0, p1, p2);
      const double w0 = p0 * h0, w1 = p1 * h1, w2 = p2 * h2;
      const double sum = w0 + w1 + w2;
      const double tF = 2 - F;
      if (sum > 0) {
        const double inv = 1.0 / sum;
        pn += fma(w2, tF, w1) * inv;
        pd += fma(w0 + w2, tF, 2 * w1) * inv;
      } else {
        // a called genotype's impossible classes are -1e15 in the reference (read_data.cpp:21),
        // not -inf: the packed view knows it holds such cells
        double lg[3] = {log(p0), log(p1), log(p2)};
        if (!gl.dense)
          for (int k = 0; k < 3; ++k)
            if (lg[k] == -__builtin_huge_val()) lg[k] = -kINF;
        const double2 tt = estmaf_term_logspace(lg, freq, F);
        pn += tt.x;
        pd += tt.y;
      }
    }
    num += wave_sum(pn);
    den += wave_sum(pd);
    freq = num / den;
    again = (fabs(prev - freq) > kEPS) && (iters++ < 100);
  } while (again);
  if (lane == 0) freq_out[site] = freq;
}

__global__ void __launch_bounds__(256)
k_fast_estmaf_stream(const GlView gl, const double* __restrict__ marg_blocks,
                     uint64_t S_own, uint64_t I_tot, uint64_t I_blk, uint64_t tile_T,
                     double* __restrict__ freq_out, const uint8_t* __restrict__ redo,
                     uint64_t row0, uint64_t row1, uint32_t* __restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
  if (!redo) {  // every site (more individuals than the register kernels hold): a wave each
    for (uint64_t site = wave; site < S_own; site += n_waves)
      if (in_tile_rows(site, tile_T, row0, row1))
        estmaf_stream_site(gl, marg_blocks, S_own, I_tot, I_blk, tile_T, freq_out, site, lane);
    return;
  }
  for (uint64_t base = wave * 64; base < S_own; base += n_waves * 64) {
    const uint64_t s = base + lane;
    const bool need = s < S_own && in_tile_rows(s, tile_T, row0, row1) && redo[s];
    uint64_t mask = __ballot(need);
    if (mask && lane == 0) atomicAdd(cnt + EST_CNT_LOGSPACE, (uint32_t)__popcll(mask));
    while (mask) {
      const int b = __builtin_ctzll(mask);
      mask &= mask - 1;
      estmaf_stream_site(gl, marg_blocks, S_own, I_tot, I_blk, tile_T, freq_out, base + b, lane);
    }
  }
}

}  // namespace


// called genotypes (a packed handle whose class table is the four unit / uniform rows): est_maf's
// per-pass sums exist in closed form
bool fast_estmaf_called(const FastState& fs, const GlView& gl) {
  return !gl.dense && gl.codes && fs.called_table && !fs.sw.estmaf_no_called;
}

bool fast_estmaf_in_place(const FastState& fs, uint64_t I_tot) {
  // est_maf on the E-step's tile-major posteriors, without the site-major copy: every cohort the
  // register kernels hold (round 6, with the posteriors' groups of eight individuals adjacent:
  // est_maf 8.4 -> 7.4 ms at 5000 x 100k, 6.7 -> 6.0 at 8000 x 60k, no difference at 3000; rounds
  // 3-4 measured the copy faster above 4096 on the earlier layout); the called-genotype sweep
  // reads every cell once, whole sectors, at any size
  return I_tot <= 8192 || (fs.packed && fs.called_table && !fs.sw.estmaf_no_called);
}

bool fast_estmaf_splittable(const FastState& fs, uint64_t I_tot, bool tile_major) {
  // the wave-per-site kernels on the E-step's tile-major posteriors: a part is a range of
  // tile rows, i.e. of workgroups
  if (tile_major && fs.packed && fs.called_table && !fs.sw.estmaf_no_called) return true;
  return tile_major && I_tot > 128 && I_tot <= 8192;
}

// per-site state of the frequency step (flags, status, the interpolant's node values)
bool fast_estmaf_reserve(FastState& fs, uint64_t S_own) {
  if (!fs.est_counts) {
    if (hipMalloc((void**)&fs.est_counts, EST_COUNTS * sizeof(uint32_t)) != hipSuccess) return false;
    if (hipMemset(fs.est_counts, 0, EST_COUNTS * sizeof(uint32_t)) != hipSuccess) return false;
    if (hipDeviceSynchronize() != hipSuccess) return false;  // (the handle's stream waits for no other)
  }
  if (S_own <= fs.redo_cap) return true;
  if (fs.redo) (void)hipFree(fs.redo);
  if (fs.est_status) (void)hipFree(fs.est_status);
  if (fs.est_state) (void)hipFree(fs.est_state);
  fs.redo = fs.est_status = nullptr;
  fs.est_state = nullptr;
  fs.redo_cap = 0;
  if (hipMalloc((void**)&fs.redo, S_own) != hipSuccess) return false;
  if (hipMalloc((void**)&fs.est_status, S_own) != hipSuccess) return false;
  if (hipMalloc((void**)&fs.est_state, S_own * EST_FIELDS * sizeof(double)) != hipSuccess) return false;
  fs.redo_cap = S_own;
  return true;
}

// dynamic LDS of an est_maf kernel; beyond 64 KB the kernel has to be told once
static size_t estmaf_dyn_lds(const void* kernel, size_t bytes) {
  if (bytes > 65536) {
    // (per device: the handles of a chain sit on several; under a lock: replicas launch from
    // several host threads)
    static std::vector<std::pair<const void*, int>> told;
    static std::mutex mu;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    const std::pair<const void*, int> key{kernel, dev};
    if (std::find(told.begin(), told.end(), key) == told.end()) {
      if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        (void)hipGetLastError();
      told.push_back(key);
    }
  }
  return bytes;
}

bool fast_estmaf(FastState& fs, hipStream_t st, const GlView& d_gl_sites,
                 const double* d_marg_blocks, uint64_t S_own, uint64_t I_tot, uint64_t I_blk,
                 double* d_freq_out, bool tile_major, uint32_t part, uint32_t n_parts) {
  if (S_own == 0) return true;
  if (n_parts == 0 || part >= n_parts) return false;
  if (n_parts > 1 && !fast_estmaf_splittable(fs, I_tot, tile_major)) return false;
  // tile rows [row0, row1) of this call (everything when n_parts == 1)
  const uint64_t n_rows = (uint64_t)fs.C * fs.T;
  const uint64_t row0 = n_parts > 1 ? n_rows * part / n_parts : 0;
  const uint64_t row1 = n_parts > 1 ? n_rows * (part + 1) / n_parts : n_rows;
  const uint64_t blk0 = row0 * 64, nblk = (row1 - row0) * 64;
  if (nblk == 0) return true;
  // tile-major posteriors (the E-step's own layout) only for the handle's whole site range
  // and individuals that fit the registers of one workgroup
  if (tile_major && !((I_tot <= 8192 || fast_estmaf_called(fs, d_gl_sites)) && I_blk == I_tot &&
                      S_own == fs.S))
    return false;
  const uint64_t tile_T = tile_major ? fs.T : 0;
  // k_fast_estmaf_stream, 4 waves per workgroup: a wave per site when it streams every site,
  // else 64 flags per wave and turn
  const bool stream_all = I_tot > 8192 && !tile_major;
  const uint64_t stream_wgs = stream_all ? (S_own + 3) / 4 : (S_own + 255) / 256;
  const dim3 grid((unsigned)(stream_wgs < 65536 ? stream_wgs : 65536)), block(256);
  if (!fast_estmaf_reserve(fs, S_own)) return false;
  if (fast_estmaf_called(fs, d_gl_sites)) {
    // called genotypes: the per-pass sums in closed form (k_fast_estmaf_called_sums)
    if (tile_major)
      hipLaunchKernelGGL((k_fast_estmaf_called_sums<true>), dim3((unsigned)nblk), dim3(64), 0, st,
                         d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, tile_T, fs.redo, fs.est_state,
                         fs.redo_cap, blk0);
    else
      hipLaunchKernelGGL((k_fast_estmaf_called_sums<false>), dim3((unsigned)S_own), dim3(64), 0, st,
                         d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, tile_T, fs.redo, fs.est_state,
                         fs.redo_cap, (uint64_t)0);
    hipLaunchKernelGGL(k_fast_estmaf_called_passes, dim3((unsigned)((S_own + 255) / 256)), dim3(256), 0,
                       st, S_own, d_freq_out, fs.redo, fs.est_status, fs.est_state, fs.redo_cap, tile_T,
                       row0, row1);
    // a called heterozygote at posterior IBD = 1 (the reference keeps a finite -1e15 there)
    hipLaunchKernelGGL(k_fast_estmaf_stream, grid, block, 0, st, d_gl_sites, d_marg_blocks, S_own,
                       I_tot, I_blk, tile_T, d_freq_out, fs.redo, row0, row1, fs.est_counts);
    return hipGetLastError() == hipSuccess;
  }
  // Interpolated passes (see k_fast_estmaf) unless NGHMM_ESTMAF_INTERP=0, which runs
  // every pass exactly.
  bool interp = true;
  interp = fs.sw.estmaf_interp != 0;
  // waves per site (W) and individuals per lane (NI): 16 per lane at two waves per SIMD;
  // a workgroup must fit one CU
  // resuming launches: 64 statuses per workgroup and turn
  const uint64_t scan_wgs_all = (S_own + 63) / 64;
  const unsigned scan_wgs = (unsigned)(scan_wgs_all < 16384 ? scan_wgs_all : 16384);
  // (kernels of several waves per site park the interval's partial node sums in dynamic LDS,
  // EN * B double2: 24 / 48 / 96 KB at 2 / 4 / 8 waves -- four, two, one workgroup per CU either way)
#define ESTMAF_DYN(K, B)                                                                        \
  ((B) > 64 ? estmaf_dyn_lds(reinterpret_cast<const void*>(K), (size_t)EN * (B) * sizeof(double2)) : (size_t)0)
#define LAUNCH_NI(N, B)                                                                         \
  do {                                                                                          \
    if (fresh)                                                                                  \
      hipLaunchKernelGGL((k_fast_estmaf<N, B, false>), dim3((unsigned)S_own), dim3(B),           \
                         ESTMAF_DYN((k_fast_estmaf<N, B, false>), B), st,                        \
                         d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, (uint64_t)0,           \
                         d_freq_out, fs.redo, fs.est_status, fs.est_state, fs.redo_cap,         \
                         n_exact, allow_build, (uint64_t)0, fs.est_counts);                     \
    else                                                                                        \
      hipLaunchKernelGGL((k_fast_estmaf_resume<N, B, false>), dim3(scan_wgs), dim3(B),          \
                         ESTMAF_DYN((k_fast_estmaf_resume<N, B, false>), B), st,                 \
                         d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, (uint64_t)0,           \
                         d_freq_out, fs.redo, fs.est_status, fs.est_state, fs.redo_cap,         \
                         n_exact, allow_build, row0, row1, fs.est_counts, cnt_slot);            \
  } while (0)
#define LAUNCH_TILE(N, B)                                                                       \
  do {                                                                                          \
    if (fresh)                                                                                  \
      hipLaunchKernelGGL((k_fast_estmaf<N, B, true>), dim3((unsigned)nblk), dim3(B),            \
                         ESTMAF_DYN((k_fast_estmaf<N, B, true>), B), st,                         \
                         d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, tile_T, d_freq_out,    \
                         fs.redo, fs.est_status, fs.est_state, fs.redo_cap, n_exact,            \
                         allow_build, blk0, fs.est_counts);                                     \
    else                                                                                        \
      hipLaunchKernelGGL((k_fast_estmaf_resume<N, B, true>), dim3(scan_wgs), dim3(B),           \
                         ESTMAF_DYN((k_fast_estmaf_resume<N, B, true>), B), st,                  \
                         d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, tile_T, d_freq_out,    \
                         fs.redo, fs.est_status, fs.est_state, fs.redo_cap, n_exact,            \
                         allow_build, row0, row1, fs.est_counts, cnt_slot);                     \
  } while (0)
#define LAUNCH_ROWS(N, TL)                                                                      \
  do {                                                                                          \
    if (fresh)                                                                                  \
      hipLaunchKernelGGL((k_fast_estmaf_rows<N, TL>),                                            \
                         dim3((unsigned)((TL) ? fs.Spad / 4 : (S_own + 3) / 4)), dim3(64), 0,   \
                         st, d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, tile_T, d_freq_out, \
                         fs.redo, fs.est_status, fs.est_state, fs.redo_cap, n_exact,            \
                         allow_build, fs.est_counts);                                           \
    else                                                                                        \
      hipLaunchKernelGGL((k_fast_estmaf_rows_resume<N, TL>), dim3(scan_wgs), dim3(64), 0, st,    \
                         d_gl_sites, d_marg_blocks, S_own, I_tot, I_blk, tile_T, d_freq_out,    \
                         fs.redo, fs.est_status, fs.est_state, fs.redo_cap, n_exact,            \
                         allow_build, fs.est_counts, cnt_slot);                                 \
  } while (0)
  // up to 128 individuals: four sites per wave (k_fast_estmaf_rows)
  const bool rows = I_tot <= 128 && !fs.sw.estmaf_no_rows;
  auto launch = [&](int fresh, int n_exact, int allow_build, int cnt_slot = 0) -> bool {
    if (rows) {
      if (tile_major) {
        if (I_tot <= 16) LAUNCH_ROWS(1, true);
        else if (I_tot <= 32) LAUNCH_ROWS(2, true);
        else if (I_tot <= 64) LAUNCH_ROWS(4, true);
        else LAUNCH_ROWS(8, true);
      } else {
        if (I_tot <= 16) LAUNCH_ROWS(1, false);
        else if (I_tot <= 32) LAUNCH_ROWS(2, false);
        else if (I_tot <= 64) LAUNCH_ROWS(4, false);
        else LAUNCH_ROWS(8, false);
      }
      return true;
    }
    if (tile_major) {
      if (I_tot <= 64) LAUNCH_TILE(1, 64);
      else if (I_tot <= 128) LAUNCH_TILE(2, 64);
      else if (I_tot <= 256) LAUNCH_TILE(4, 64);
      else if (I_tot <= 512) LAUNCH_TILE(8, 64);
      else if (I_tot <= 768) LAUNCH_TILE(12, 64);
      else if (I_tot <= 1024) LAUNCH_TILE(16, 64);
      // several waves per site: a lane's slot k holds individual thread + BLOCK k, so a cohort
      // in the lower half of a size class leaves the upper slots of EVERY lane empty -- 12
      // instead of 16 slots there (5000 individuals on 512 threads: 9.8 slots in use)
      else if (I_tot <= 1536) LAUNCH_TILE(12, 128);
      else if (I_tot <= 2048) LAUNCH_TILE(16, 128);
      else if (I_tot <= 3072) LAUNCH_TILE(12, 256);
      else if (I_tot <= 4096) LAUNCH_TILE(16, 256);
      // (above 4096 a workgroup is a CU's eight waves and a site's time goes with the slots per
      // lane: 10 and 14 where they hold the cohort)
      else if (I_tot <= 5120) LAUNCH_TILE(10, 512);
      else if (I_tot <= 6144) LAUNCH_TILE(12, 512);
      else if (I_tot <= 7168) LAUNCH_TILE(14, 512);
      else LAUNCH_TILE(16, 512);
    } else if (I_tot <= 64) LAUNCH_NI(1, 64);
    else if (I_tot <= 128) LAUNCH_NI(2, 64);
    else if (I_tot <= 256) LAUNCH_NI(4, 64);
    else if (I_tot <= 512) LAUNCH_NI(8, 64);
    else if (I_tot <= 768) LAUNCH_NI(12, 64);
    else if (I_tot <= 1024) LAUNCH_NI(16, 64);
    else if (I_tot <= 1536) LAUNCH_NI(12, 128);
    else if (I_tot <= 2048) LAUNCH_NI(16, 128);
    else if (I_tot <= 3072) LAUNCH_NI(12, 256);
    else if (I_tot <= 4096) LAUNCH_NI(16, 256);
    else if (I_tot <= 5120) LAUNCH_NI(10, 512);
    else if (I_tot <= 6144) LAUNCH_NI(12, 512);
    else if (I_tot <= 7168) LAUNCH_NI(14, 512);
    else LAUNCH_NI(16, 512);
    return true;
  };
  const uint8_t* redo = fs.redo;
  if (I_tot > 8192 && !tile_major) {
    redo = nullptr;  // more individuals than registers hold: stream every site
  } else if (!interp) {
    if (!launch(1, 0, 0)) return false;
  } else {
    const dim3 igrid((unsigned)((S_own + 255) / 256));
    if (!launch(1, EST_K0, 1)) return false;
    hipLaunchKernelGGL(k_fast_estmaf_interp, igrid, dim3(256), 0, st, S_own, d_freq_out, fs.redo,
                       fs.est_status, fs.est_state, fs.redo_cap, tile_T, row0, row1);
    // sites that left their interval -- a frequency far from the 0.01 every site starts at: the
    // odds then travel several times their value (gen_func.cpp:976) -- get a second and a third
    for (int again = 0; again < 2; ++again) {
      if (!launch(0, 1, 1, again ? EST_CNT_RESUMED_AGAIN : EST_CNT_RESUMED)) return false;
      hipLaunchKernelGGL(k_fast_estmaf_interp, igrid, dim3(256), 0, st, S_own, d_freq_out, fs.redo,
                         fs.est_status, fs.est_state, fs.redo_cap, tile_T, row0, row1);
    }
    if (!launch(0, 0, 0, EST_CNT_EXACT_TAIL)) return false;  // whatever is left finishes on exact passes
  }
  hipLaunchKernelGGL(k_fast_estmaf_stream, grid, block, 0, st, d_gl_sites, d_marg_blocks, S_own,
                     I_tot, I_blk, tile_T, d_freq_out, redo, row0, row1, fs.est_counts);
#undef LAUNCH_NI
#undef ESTMAF_DYN
#undef LAUNCH_TILE
#undef LAUNCH_ROWS
  return hipGetLastError() == hipSuccess;
}

}  // namespace nghmm

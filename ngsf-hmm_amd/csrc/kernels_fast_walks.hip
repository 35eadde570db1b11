// kernels_fast_walks.hip -- the objective rounds of the indF / alpha M-step: forward walks over <= 5 probe points per
// individual (one kernel per finite-difference pattern), their finish, and what a site shard
// exchanges per round
// (fast mode, gfx950; kernels_fast.hip's header comment has the design, DESIGN.md section 4 the
// measurements.)
#include "fast_dev.hpp"

namespace nghmm {

namespace {

// The main loop of one wave for the finite-difference pattern.  Per site and lane: one
// exp(-alpha d) (SMALL, alpha * d_max <= 2^-6: a polynomial), the products shared by all points,
// and per point the two rows' update; one exponent (point 0's) rescales all points, which are
// perturbations of each other.
//
// SMALL versions run in the kappa form (fast_dev.hpp: op_step_k): the operators are kept divided
// by c_s = exp(-alpha d_s), whose product over the
// lane-chunk the end of the walk puts back -- 10 instructions for point 0 (the two rows 8), 10 per
// F probe, 15 per alpha probe (kappa_probe = kappa m + (m - 1), m = exp((alpha_probe - alpha_0) d)
// tiny-argument), 11 shared: 71 per site for the five points where the c form spends 87 (+ 3).
template <int NF, int NA, bool SMALL, bool EMIT, int XDEG, bool OWNEX, typename Src>
__device__ __forceinline__ void lkl_run_fd(Src& src, uint64_t T, const GroupDesc& G,
                                           Op (&R)[MAXP], EmitPtrs emit, uint64_t wave,
                                           int lane, const double2* __restrict__ chunk_scale,
                                           uint32_t chunk) {
  static_assert(NB * UG == CK && RENORM == CK, "checkpoints are stored right after a rescale");
  constexpr int NPT = 1 + NF + NA;
  constexpr bool KF = SMALL;
  const uint64_t nblk = T / CK;
  const double al0 = G.A[0];
  const double q1 = G.F[0], q0 = 1 - q1;
  double rho0[NF > 0 ? NF : 1], rho1[NF > 0 ? NF : 1], dal[NA > 0 ? NA : 1];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    rho0[f] = (1 - G.F[1 + f]) / q0;
    rho1[f] = G.F[1 + f] / q1;
  }
#pragma unroll
  for (int a = 0; a < NA; ++a) dal[a] = KF ? G.A[1 + NF + a] - al0 : al0 - G.A[1 + NF + a];
  int exc = 0;
  // Software pipeline: NB buffers of UG sites.  A buffer is refilled right after it
  // has been consumed, i.e. (NB-1) groups = 6 sites before it is needed again, which
  // covers HBM latency at two to three waves per SIMD.  T is a multiple of NB*UG and the
  // arrays carry one group of slack at the end, so neither the prologue nor the refills
  // need bound checks (values read past T are never used); sites past S are identity
  // operators (e = 1, d = 0).
  typename Src::Buf buf[NB][UG];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int u = 0; u < UG; ++u) buf[b][u] = src.load((uint64_t)b * UG + u);
  }
  for (uint64_t t0 = 0; t0 < T; t0 += NB * UG) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        double rho, d;
        src.get(buf[b][u], t0 + (uint64_t)b * UG + u, rho, d);
        if constexpr (KF) {
          // (chromosome starts, d = kDStart: the clamps give every point its constant kappa,
          // fast_dev.hpp; the loop body stays branch- and select-free)
          const double x = min_num(al0 * d, KAPPA_XMAX);
          const double kap = expm1_over_x_tiny(x) * x;
          const double eq1 = q1 * rho;
          const double g0 = kap * q0, g1 = kap * eq1;
          op_step_k(R[0], rho, g0, g1);
#pragma unroll
          for (int f = 0; f < NF; ++f) op_step_k(R[1 + f], rho, g0 * rho0[f], g1 * rho1[f]);
          const double dcl = min_num(d, KAPPA_DCLAMP);
#pragma unroll
          for (int a = 0; a < NA; ++a) {
            // |x| <= 1e-3 on every finite distance (checked by the host)
            const double mm1 = expm1_small<XDEG>(dal[a] * dcl);
            const double ka = fma(kap, 1.0 + mm1, mm1);
            op_step_k(R[1 + NF + a], rho, ka * q0, ka * eq1);
          }
        } else {
          const double c0 = coanc(al0, d);
          const double a0 = 1 - c0;
          const double ce0 = c0, ce1 = c0 * rho;  // emissions (1, rho)
          const double eq0 = q0, eq1 = rho * q1;
          const double g0 = a0 * eq0, g1 = a0 * eq1;
          op_step(R[0], ce0, ce1, g0, g1);
#pragma unroll
          for (int f = 0; f < NF; ++f) op_step(R[1 + f], ce0, ce1, g0 * rho0[f], g1 * rho1[f]);
#pragma unroll
          for (int a = 0; a < NA; ++a) {
            // |x| <= 1e-3 on every finite distance (checked by the host); at d = kDStart the
            // polynomial is huge but finite and multiplies c0 = 0
            const double m = exp_small<XDEG>(dal[a] * d);
            const double am = fma(-c0, m, 1.0);
            op_step(R[1 + NF + a], ce0 * m, ce1 * m, am * eq0, am * eq1);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UG; ++u) buf[b][u] = src.load(t0 + (uint64_t)(b + NB) * UG + u);
    }
    if constexpr (OWNEX) {  // every point by its own exponent (kept in R[p].ex)
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        const double mx = fmax(fmax(R[p].a00, R[p].a01), fmax(R[p].a10, R[p].a11));
        const int e = exp_of(mx);
        const double sc = __builtin_ldexp(1.0, -e);
        R[p].ex += e;
        R[p].a00 *= sc;
        R[p].a01 *= sc;
        R[p].a10 *= sc;
        R[p].a11 *= sc;
      }
      src.rescale();
    } else {  // rescale every point by point 0's exponent
      const double mx = fmax(fmax(R[0].a00, R[0].a01), fmax(R[0].a10, R[0].a11));
      const int e = exp_of(mx);
      const double sc = __builtin_ldexp(1.0, -e);
      exc += e;
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        R[p].a00 *= sc;
        R[p].a01 *= sc;
        R[p].a10 *= sc;
        R[p].a11 *= sc;
      }
      src.rescale();
    }
    if constexpr (EMIT) {  // first round of an M-step: point 0 is the E-step's forward walk
      // (no bound check, to keep the loop one basic block: the store after the last block
      // lands in the unused slot 0 of the next wave, or in the array's slack)
      // (kappa form: a checkpoint is the prefix operator times a positive factor, and the
      // backward sweep normalises every site's posterior by its own sum)
      emit_checkpoint(emit.ckpt, wave, nblk, t0 / CK + 1, lane, R[0]);
    }
  }
  if constexpr (!OWNEX) {
#pragma unroll
    for (int p = 0; p < NPT; ++p) R[p].ex = exc;
  }
  if constexpr (KF) {
    // back to the operators themselves: prod_s c_s = exp(-alpha sum_s d_s) over the lane-chunk's
    // finite distances, and per chromosome start the constant kappa the point had there
    const double2 cs = chunk_scale[(uint64_t)chunk * 64 + lane];
    const double s0 = exp_nonpos(-al0 * cs.x);
#pragma unroll
    for (int p = 0; p < NPT; ++p) {
      const double sp = p < 1 + NF ? s0 : exp_nonpos(-G.A[p] * cs.x);
      R[p].a00 *= sp;
      R[p].a01 *= sp;
      R[p].a10 *= sp;
      R[p].a11 *= sp;
    }
    if (cs.y > 0.0) {  // (a few lane-chunks of a data set)
      const double k0 = expm1_over_x_tiny(KAPPA_XMAX) * KAPPA_XMAX;
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        double kp = k0;
        if (p >= 1 + NF) {
          const double mm1 = expm1_small<XDEG>(dal[p - 1 - NF < NA ? p - 1 - NF : 0] * KAPPA_DCLAMP);
          kp = fma(k0, 1.0 + mm1, mm1);
        }
        const double inv = 1.0 / kp;
        for (int n = (int)cs.y; n > 0; --n) {
          R[p].a00 *= inv;
          R[p].a01 *= inv;
          R[p].a10 *= inv;
          R[p].a11 *= inv;
          renorm(R[p]);
        }
      }
    }
  }
}

// ordered product of the 64 lanes' operators; lane 0 stores the wave's operator
__device__ __forceinline__ void lkl_store_wave_op(Op r, int lane, double* __restrict__ out) {
  renorm(r);
  Op m = r;
  for (int off = 1; off < 64; off <<= 1) {
    const Op o = op_shfl_down(m, off);
    if ((lane & (2 * off - 1)) == 0) m = op_mul(m, o);
  }
  if (lane == 0) {
    out[0] = m.a00;
    out[1] = m.a01;
    out[2] = m.a10;
    out[3] = m.a11;
    out[4] = (double)m.ex;
  }
}

// The same ordered products for the NP points of a finite-difference group AT ONCE.  A shuffle
// tree per point keeps 2^-k of the lanes busy at level k and does that NP times: 6 op_mul levels
// x 5 points = 30 per lane, ~900 instructions per wave -- a dozen sites' worth, 2 % of a wave at
// 320 sites per lane and 20 % at the 40 sites per lane of an eight-GPU site shard (or of 100 x
// 100k).  Here the first level runs on shuffles for all points, the 32 products per point go to
// LDS, and every further level takes ALL points' pairs side by side, one pair per lane: 80, 40,
// 20, 10, 5 products in 2 + 1 + 1 + 1 + 1 passes -- 5 + 6 op_mul per lane instead of 30.  The
// pairs and their order are the shuffle tree's (level k multiplies the products of lanes
// 2^k m .. and 2^k m + 2^(k-1) ..), so every result has the same bits.  One wave per workgroup:
// in-place updates are safe (a wave's LDS reads are issued before its writes), passes of one
// level touch different points.
struct TreeLds {
  double v[MAXP * 32][6];   // a00, a01, a10, a11, (double) ex, pad: 48 B per operator
};

__device__ __forceinline__ Op tree_load(const TreeLds& L, int idx) {
  const double2* q = reinterpret_cast<const double2*>(L.v[idx]);
  const double2 x = q[0], y = q[1], z = q[2];
  return Op{x.x, x.y, y.x, y.y, (int)z.x};
}

__device__ __forceinline__ void tree_store(TreeLds& L, int idx, const Op& m) {
  double2* q = reinterpret_cast<double2*>(L.v[idx]);
  q[0] = double2{m.a00, m.a01};
  q[1] = double2{m.a10, m.a11};
  q[2] = double2{(double)m.ex, 0.0};
}

template <int NP>
__device__ __forceinline__ void lkl_store_wave_ops(Op (&R)[MAXP], int lane, double* __restrict__ out,
                                                   TreeLds& L) {
  static_assert(NP >= 1 && NP <= MAXP, "points of one group");
#pragma unroll
  for (int p = 0; p < NP; ++p) {  // level 1, as lkl_store_wave_op does it
    renorm(R[p]);
    const Op o = op_shfl_down(R[p], 1);
    if ((lane & 1) == 0) {
      R[p] = op_mul(R[p], o);
      tree_store(L, p * 32 + (lane >> 1), R[p]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int lg = 4; lg >= 0; --lg) {  // n = 2^lg products per point at this level
    const int n = 1 << lg;
    const int items = NP * n;
#pragma unroll
    for (int t0 = 0; t0 < NP * 16; t0 += 64) {
      if (t0 < items) {
        const int t = t0 + lane;
        const bool on = t < items;
        const int tt = on ? t : 0;
        const int p = tt >> lg, k = tt & (n - 1);
        const int src = p * 32 + 2 * k;
        const Op m = op_mul(tree_load(L, src), tree_load(L, src + 1));
        __syncthreads();  // (single wave: orders this pass's reads before its writes for the compiler)
        if (on) tree_store(L, p * 32 + k, m);
        __syncthreads();
      }
    }
  }
  if (lane < NP) {
    const Op m = tree_load(L, lane * 32);
    double* o = out + lane * 5;
    o[0] = m.a00;
    o[1] = m.a01;
    o[2] = m.a10;
    o[3] = m.a11;
    o[4] = (double)m.ex;
  }
}

// One wave's part of a group's objective: the lanes' operators over the T sites of lane-chunk c, the
// by-products of an emitting round, the wave's ordered product to part[g][c].
template <int NF, int NA, bool SMALL, bool EMIT, int SRC, int XDEG, bool OWNEX>
__device__ __forceinline__ void lkl_fd_wave(const LklArrays& arr, uint64_t T, uint32_t C,
                                            const GroupDesc* __restrict__ groups, uint32_t g, uint32_t c,
                                            double* __restrict__ part, const EmitPtrs& emit, TreeLds& tree) {
  static_assert(EMIT || SRC == SRC_PLAIN, "the fresh walk is the first round of an M-step");
  const int lane = threadIdx.x;
  const GroupDesc& G = groups[g];
  const uint64_t i = G.ind;
  Op R[MAXP];
#pragma unroll
  for (int p = 0; p < MAXP; ++p) R[p] = Op{1.0, 0.0, 0.0, 1.0, 0};
  const uint64_t wave_base = ((i * C + c) * T) * 64 + lane;
  const uint64_t pos_base = ((uint64_t)c * T) * 64 + lane;
  using Src = SrcOf<SRC>;
  Src src(arr, wave_base, pos_base);
  static_assert(!(OWNEX && EMIT), "an emitting round's checkpoints assume point 0's scale");
  lkl_run_fd<NF, NA, SMALL, EMIT, XDEG, OWNEX>(src, T, G, R, emit, i * C + c, lane, arr.chunk_scale, c);
  if constexpr (SRC != SRC_PLAIN) {  // fresh walk: the wave's part of sum log e0
    const double bl = wave_sum(src.base.log_value());
    if (lane == 0) arr.base_c[i * C + c] = bl + (arr.gl_scale_c ? arr.gl_scale_c[i * C + c] : 0.0);
  }
  if constexpr (EMIT) {
    Op r0 = R[0];
    renorm(r0);
    emit_lane_op(emit.lane_ops, i * C + c, lane, r0);
  }
  lkl_store_wave_ops<1 + NF + NA>(R, lane, part + ((uint64_t)g * C + c) * MAXP * 5, tree);
}

// One kernel per loop-body version (each gets its own register allocation); the host
// sorts the groups of a round by mode and launches every version on its range
// [g_begin, g_begin + gridDim.x / C).
template <int NF, int NA, bool SMALL, bool EMIT, int SRC, int XDEG, bool OWNEX = false>
__global__ void __launch_bounds__(64)
k_fast_lkl_fd(LklArrays arr, uint64_t T, uint32_t C, const GroupDesc* __restrict__ groups,
              uint32_t g_begin, double* __restrict__ part, EmitPtrs emit,
              const uint32_t* __restrict__ worklist = nullptr) {
  // chunk-major: the waves resident at a time walk the same few slices of the shared
  // distance / frequency tables, which then stay in L2
  const uint32_t n_g = gridDim.x / C;
  // (rounds planned on the device, kernels_bfgs.hip: the mode's worklist names the groups,
  // which lie -- descriptors and partial operators -- by individual)
  const uint32_t g = worklist ? worklist[blockIdx.x % n_g] : g_begin + blockIdx.x % n_g;
  const uint32_t c = blockIdx.x / n_g;
  __shared__ TreeLds tree;
  lkl_fd_wave<NF, NA, SMALL, EMIT, SRC, XDEG, OWNEX>(arr, T, C, groups, g, c, part, emit, tree);
}

// A SMALL round planned on the device whose individuals are of several versions (small and general
// alpha, degree 4 and 2): launched one version after the other, each launch is as long as one wave's
// walk -- a few thousand waves do not fill the chip -- and the round as long as their sum.  Here
// the versions share ONE launch: the block picks its version's worklist and loop body (the same
// code as k_fast_lkl_fd's, the same bits; the registers of the largest).
struct MixVersions {
  const uint32_t* worklist[8];
  uint32_t end[8];      // running count of groups up to and including version k
  uint32_t version[8];  // bit 0: small alpha, bit 1: degree 2, bit 2: an exponent per point
  uint32_t n;
};
__global__ void __launch_bounds__(64)
k_fast_lkl_fd_mix(LklArrays arr, uint64_t T, uint32_t C, const GroupDesc* __restrict__ groups,
                  double* __restrict__ part, MixVersions mix) {
  const uint32_t n_g = gridDim.x / C;
  const uint32_t idx = blockIdx.x % n_g, c = blockIdx.x / n_g;
  // (the block's version by compares on the arguments' scalars: no indexed copy of them)
  uint32_t first = 0, version = mix.version[0];
  const uint32_t* wl = mix.worklist[0];
#pragma unroll
  for (int j = 1; j < 8; ++j) {
    if ((uint32_t)j < mix.n && idx >= mix.end[j - 1]) {
      first = mix.end[j - 1];
      version = mix.version[j];
      wl = mix.worklist[j];
    }
  }
  const uint32_t g = wl[idx - first];
  const EmitPtrs emit{nullptr, nullptr};
  __shared__ TreeLds tree;
  switch (version) {
#define MIX_CASE(V, SM, XD, OX)                                                                       \
  case V:                                                                                             \
    lkl_fd_wave<2, 2, SM, false, SRC_PLAIN, XD, OX>(arr, T, C, groups, g, c, part, emit, tree);      \
    break;
    MIX_CASE(0, false, 4, false)
    MIX_CASE(1, true, 4, false)
    MIX_CASE(2, false, 2, false)
    MIX_CASE(3, true, 2, false)
    MIX_CASE(4, false, 4, true)
    MIX_CASE(5, true, 4, true)
    MIX_CASE(6, false, 2, true)
    MIX_CASE(7, true, 2, true)
#undef MIX_CASE
  }
}

template <int NP_MAX, int SRC>
__global__ void __launch_bounds__(64)
k_fast_lkl_chunks(LklArrays arr, uint64_t T, uint32_t C, const GroupDesc* __restrict__ groups,
                  uint32_t g_begin, double* __restrict__ part, EmitPtrs emit,
                  const uint32_t* __restrict__ worklist = nullptr) {
  // chunk-major: the waves resident at a time walk the same few slices of the shared
  // distance / frequency tables, which then stay in L2
  const uint32_t n_g = gridDim.x / C;
  const uint32_t g = worklist ? worklist[blockIdx.x % n_g] : g_begin + blockIdx.x % n_g;
  const uint32_t c = blockIdx.x / n_g;
  const int lane = threadIdx.x;
  const GroupDesc& G = groups[g];
  const uint32_t np = G.np;
  const uint64_t i = G.ind;

  Op R[NP_MAX];
  double q0[NP_MAX], q1[NP_MAX], al[NP_MAX];
#pragma unroll
  for (int p = 0; p < NP_MAX; ++p) {
    const double f = (p < (int)np) ? G.F[p] : 0.5;
    q1[p] = f;
    q0[p] = 1 - f;
    al[p] = (p < (int)np) ? G.A[p] : 1.0;
    R[p] = Op{1.0, 0.0, 0.0, 1.0, 0};
  }

  using Src = SrcOf<SRC>;
  Src src(arr, ((i * C + c) * T) * 64 + lane, ((uint64_t)c * T) * 64 + lane);
  typename Src::Buf buf[NB][UG];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int u = 0; u < UG; ++u) buf[b][u] = src.load((uint64_t)b * UG + u);
  }
  for (uint64_t t0 = 0; t0 < T; t0 += NB * UG) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        double rho, d;
        src.get(buf[b][u], t0 + (uint64_t)b * UG + u, rho, d);
#pragma unroll
        for (int p = 0; p < NP_MAX; ++p) {
          if (p < (int)np) {
            const double cc = coanc(al[p], d);
            const double a = 1 - cc;
            op_step(R[p], cc, cc * rho, a * q0[p], a * rho * q1[p]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UG; ++u) buf[b][u] = src.load(t0 + (uint64_t)(b + NB) * UG + u);
    }
#pragma unroll
    for (int p = 0; p < NP_MAX; ++p)
      if (p < (int)np) renorm(R[p]);
    src.rescale();
    if (emit.ckpt) {
      const uint64_t b = t0 / CK + 1;
      if (b < T / CK) emit_checkpoint(emit.ckpt, i * C + c, T / CK, b, lane, R[0]);
    }
  }
  if constexpr (SRC != SRC_PLAIN) {
    const double bl = wave_sum(src.base.log_value());
    if (lane == 0) arr.base_c[i * C + c] = bl + (arr.gl_scale_c ? arr.gl_scale_c[i * C + c] : 0.0);
  }
  if (emit.lane_ops) {
    Op r0 = R[0];
    renorm(r0);
    emit_lane_op(emit.lane_ops, i * C + c, lane, r0);
  }
#pragma unroll
  for (int p = 0; p < NP_MAX; ++p)
    if (p < (int)np) lkl_store_wave_op(R[p], lane, part + (((uint64_t)g * C + c) * MAXP + p) * 5);
}

// lkl = log( q . prod_c R_c . 1 ): one workgroup per group, one wave per point; lane l holds the
// ordered product of the operators of chunks l*K .. l*K + K - 1 of the wave's point (K =
// ceil(C / 64): one chunk per lane up to 64 chunks) and an ordered shuffle tree multiplies them.
// SHARD (a handle that holds a site range of a larger data set, SiteShard): the product and
// the sum of log e0 of this range go to the send buffer instead, six doubles per point in point
// order; k_fast_shard_combine finishes the value once every range's part has arrived.
template <bool SHARD>
__global__ void __launch_bounds__(64 * MAXP)
k_fast_lkl_finish(const GroupDesc* __restrict__ groups, uint32_t n_groups, uint32_t C,
                  const double* __restrict__ part, const double* __restrict__ base_c,
                  double* __restrict__ lkl_out, int* __restrict__ flags,
                  const uint32_t* __restrict__ worklist = nullptr) {
  const uint32_t g = worklist ? worklist[blockIdx.x] : blockIdx.x;
  const int lane = threadIdx.x & 63;
  const GroupDesc& G = groups[g];
  const uint32_t p = threadIdx.x >> 6;
  if (p >= G.np) return;
  // sum of log e0 over the individual's sites: the same for every point
  const double base = base_sum(base_c + (uint64_t)G.ind * C, C, lane);
  const Op m = lkl_point_product(part + (uint64_t)g * C * MAXP * 5, C, p, lane);
  if (lane == 0) {
    if constexpr (SHARD) {
      double* o = lkl_out + (uint64_t)G.out_idx[p] * 6;
      o[0] = m.a00;
      o[1] = m.a01;
      o[2] = m.a10;
      o[3] = m.a11;
      o[4] = (double)m.ex;
      o[5] = base;
    } else {
      const double l = lkl_point_value(m, G.F[p], base);
      lkl_out[G.out_idx[p]] = l;
      // NaN or +-inf: overflow of a probe against point 0's scale, or no probability mass left
      // in linear space; the host re-evaluates such points with the general kernel
      // (device-planned rounds pass no flags: k_bfgs_advance looks at the values itself)
      if (flags && !(fabs(l) < __builtin_huge_val())) flags[FLAG_INVALID_LKL] = 1;
    }
  }
}

// ---- site shards ------------------------------------------------------------
// A run of sites is the product of its operators, so the SITE axis can be cut between GPUs as
// it is cut between lane-chunks: every handle holds all individuals for a contiguous site
// range, walks it as if it were a data set of its own, and what the ranges owe each other per
// individual is one 2x2 operator (+ exponent, + the range's sum of log e0): six doubles.  They
// travel by an all-gather the caller provides (SiteShard::allgather, stream-ordered); every
// handle then multiplies the ranges' operators in rank order, so all of them see the same
// bits and run the same L-BFGS-B steps.  est_maf has every individual of its sites at hand:
// the frequency step needs no exchange at all.
//
// recv = [world][n][6]; one thread per point: lkl = sum_r base_r + log(q . prod_r M_r . 1)
__global__ void __launch_bounds__(256)
k_fast_shard_combine(const GroupDesc* __restrict__ groups, uint32_t n_groups,
                     const double* __restrict__ recv, uint32_t world, uint64_t n,
                     double* __restrict__ lkl_out, int* __restrict__ flags,
                     const uint32_t* __restrict__ worklist = nullptr) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t g = (uint32_t)(t / MAXP), p = (uint32_t)(t % MAXP);
  if (g >= n_groups) return;
  const GroupDesc& G = worklist ? groups[worklist[g]] : groups[g];
  if (p >= G.np) return;
  const uint64_t idx = G.out_idx[p];
  Op m = op_load(recv + idx * 6);
  double base = recv[idx * 6 + 5];
  for (uint32_t r = 1; r < world; ++r) {
    const double* o = recv + ((uint64_t)r * n + idx) * 6;
    m = op_mul(m, op_load(o));
    base += o[5];
  }
  const double q0 = 1 - G.F[p], q1 = G.F[p];
  const double v0 = fma(q0, m.a00, q1 * m.a10), v1 = fma(q0, m.a01, q1 * m.a11);
  const double l = base + (log(v0 + v1) + (double)m.ex * 0.6931471805599453094);
  lkl_out[idx] = l;
  if (flags && !(fabs(l) < __builtin_huge_val())) flags[FLAG_INVALID_LKL] = 1;
}

// The first objective round of an M-step carries every individual's current parameters as its
// point 0: that point's gathered operators ARE the ranges' operators the E-step needs, so the
// E-step's own all-gather is saved (stride 6 doubles per point, n points per rank)
__global__ void __launch_bounds__(256)
k_fast_shard_edges_from_round(const GroupDesc* __restrict__ groups, uint32_t n_groups,
                              const double* __restrict__ recv, uint32_t world, uint32_t rank,
                              uint64_t n, double* __restrict__ edges,
                              const uint32_t* __restrict__ worklist = nullptr) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const GroupDesc& G = worklist ? groups[worklist[g]] : groups[g];
  const uint64_t idx = G.out_idx[0];
  const double f = G.F[0];
  double u0 = 1 - f, u1 = f;
  int uex = 0;
  double base = 0.0;
  for (uint32_t r = 0; r < world; ++r) {
    const double* o = recv + ((uint64_t)r * n + idx) * 6;
    base += o[5];
    if (r < rank) {
      const Op m = op_load(o);
      const double n0 = fma(u0, m.a00, u1 * m.a10), n1 = fma(u0, m.a01, u1 * m.a11);
      u0 = n0;
      u1 = n1;
      uex += m.ex;
      renorm2(u0, u1, uex);
    }
  }
  double x0 = 1.0, x1 = 1.0;
  int xex = 0;
  for (uint32_t r = world; r-- > rank + 1;) {
    const Op m = op_load(recv + ((uint64_t)r * n + idx) * 6);
    const double n0 = fma(m.a00, x0, m.a01 * x1), n1 = fma(m.a10, x0, m.a11 * x1);
    x0 = n0;
    x1 = n1;
    xex += m.ex;
    renorm2(x0, x1, xex);
  }
  double* e = edges + (uint64_t)G.ind * 8;
  e[0] = u0;
  e[1] = u1;
  e[2] = (double)uex;
  e[3] = x0;
  e[4] = x1;
  e[5] = (double)xex;
  e[6] = base;
  e[7] = 0.0;
}

}  // namespace


bool fast_lkl_prepare(FastState& fs, hipStream_t st, uint32_t n_pts, const uint32_t* h_ind,
                      const double* h_F, const double* h_A, bool force_general) {
  FastState::LklLane& L = fs.lanes[fs.cur_lane];
  L.n_groups = 0;
  L.n_pts = n_pts;
  if (n_pts == 0) return true;
  // group the points by individual (<= MAXP per group): stable counting sort on the
  // individual index (the caller has checked ind < I)
  std::vector<uint32_t> order(n_pts), start(fs.I + 1, 0);
  for (uint32_t p = 0; p < n_pts; ++p) ++start[h_ind[p] + 1];
  for (uint64_t i = 0; i < fs.I; ++i) start[i + 1] += start[i];
  for (uint32_t p = 0; p < n_pts; ++p) order[start[h_ind[p]]++] = p;
  std::vector<GroupDesc> groups;
  groups.reserve(n_pts / 3 + 1);
  for (uint32_t k = 0; k < n_pts;) {
    GroupDesc G;
    std::memset(&G, 0, sizeof G);
    G.ind = h_ind[order[k]];
    uint32_t np = 0;
    while (k < n_pts && np < (uint32_t)MAXP && h_ind[order[k]] == G.ind) {
      G.F[np] = h_F[order[k]];
      G.A[np] = h_A[order[k]];
      G.out_idx[np] = order[k];
      ++np;
      ++k;
    }
    G.np = np;
    G.mode = force_general ? 0u : fd_pattern(G, fs.dmax_finite, fs.T, fs.packed, fs.alpha_small_min);
    fd_pad(G);
    groups.push_back(G);
  }
  // one kernel per loop-body version: sort the groups by mode (stable, so still in
  // individual order inside a mode) and remember the ranges
  std::stable_sort(groups.begin(), groups.end(),
                   [](const GroupDesc& a, const GroupDesc& b) { return a.mode < b.mode; });
  L.mode_ranges.clear();
  for (uint32_t k = 0; k < groups.size();) {
    uint32_t e = k;
    while (e < groups.size() && groups[e].mode == groups[k].mode) ++e;
    L.mode_ranges.push_back({groups[k].mode, k, e - k});
    k = e;
  }
  for (const auto& r : L.mode_ranges) fs.mode_ind_rounds[r.mode] += r.count;
  const uint32_t ng = (uint32_t)groups.size();
  const size_t gbytes = (size_t)ng * sizeof(GroupDesc);
  if (gbytes > L.grp_cap) {
    if (L.grp_dev) (void)hipFree(L.grp_dev);
    L.grp_dev = nullptr;
    L.grp_cap = 0;
    const size_t cap = gbytes + gbytes / 4 + 4096;
    if (hipMalloc(&L.grp_dev, cap) != hipSuccess) return false;
    L.grp_cap = cap;
  }
  const size_t pdoubles = (size_t)ng * fs.C * MAXP * 5;
  if (pdoubles > L.part_cap) {
    if (L.part) (void)hipFree(L.part);
    L.part = nullptr;
    L.part_cap = 0;
    const size_t cap = pdoubles + pdoubles / 4 + 1024;
    if (!dalloc(&L.part, cap)) return false;
    L.part_cap = cap;
  }
  // the descriptors must outlive the async copy
  L.grp_host.assign(reinterpret_cast<unsigned char*>(groups.data()),
                     reinterpret_cast<unsigned char*>(groups.data()) + gbytes);
  if (hipMemcpyAsync(L.grp_dev, L.grp_host.data(), gbytes, hipMemcpyHostToDevice, st) !=
      hipSuccess)
    return false;
  L.n_groups = ng;
  return true;
}

// One round: every loop-body version on its range of the sorted descriptors, then the finish.
// Device-planned rounds (d_worklists != null, kernels_bfgs.hip): the descriptors lie by
// individual, range r = the first r.count entries of mode r.mode's worklist
// (d_worklists + mode_slot(mode) * wl_stride), d_all = the ng groups of the round in one list.
// (waves of a round up to which its versions share a launch: four times what the chip holds at once)
constexpr uint64_t kMixMaxWaves = 16384;
constexpr uint32_t kModeMixedRounds = 0xffffffffu;  // nghmm_debug_mode_counts: the rounds that did
static bool lkl_launch_groups(FastState& fs, hipStream_t st, const GroupDesc* dg,
                              const std::vector<FastState::ModeRange>& mode_ranges, uint32_t ng,
                              uint32_t n_pts, double* part, double* d_lkl, int* d_flags,
                              bool emit_estep, const uint32_t* d_worklists = nullptr,
                              uint64_t wl_stride = 0, const uint32_t* d_all = nullptr,
                              bool finish_elsewhere = false) {
  if (ng == 0) return true;
  const LklArrays arr = lkl_arrays(fs);
  // first round of an M-step inside nghmm_estep_mstep: point 0 of every individual is the
  // E-step's forward walk, whose lane operators and checkpoints it leaves behind; if the
  // emissions are stale (frequencies just updated) the same walk recomputes and stores them
  const EmitPtrs emit = emit_estep ? EmitPtrs{fs.lane_ops, reinterpret_cast<double2*>(fs.ckpt)}
                                   : EmitPtrs{nullptr, nullptr};
  const bool fresh = emit_estep && fs.e_stale;
  if (fs.e_stale && !fresh) return false;  // the caller refreshes the emissions first
  // (the kernel versions of one round side by side on helper streams, so that they share one
  // partly filled last wave batch: measured, no gain -- 26.6-26.9 vs 27.0-27.1 ms per iteration
  // at 1000 x 1M)
  // a small device-planned round of several versions of the padded pattern: one launch (k_fast_lkl_fd_mix)
  bool mixed = false;
  if (d_worklists && !emit_estep && mode_ranges.size() > 1 && mode_ranges.size() <= 8 &&
      (uint64_t)ng * fs.C <= kMixMaxWaves) {
    MixVersions mix{};
    uint32_t end = 0;
    bool all_padded = true;
    for (const auto& r : mode_ranges) {
      const uint32_t full = fd_mode(2, 2, (r.mode & FD_SMALL) != 0, (r.mode & FD_XDEG2) != 0);
      all_padded = all_padded && (r.mode & ~FD_OWNEX) == full;
      end += r.count;
      mix.worklist[mix.n] = d_worklists + (uint64_t)mode_slot(r.mode) * wl_stride;
      mix.end[mix.n] = end;
      mix.version[mix.n] = ((r.mode & FD_SMALL) ? 1u : 0u) | ((r.mode & FD_XDEG2) ? 2u : 0u) | ((r.mode & FD_OWNEX) ? 4u : 0u);
      ++mix.n;
    }
    if (all_padded && end == ng) {
      hipLaunchKernelGGL(k_fast_lkl_fd_mix, dim3(ng * fs.C), dim3(64), 0, st, arr, fs.T, fs.C, dg, part, mix);
      mixed = true;
      ++fs.mode_ind_rounds[kModeMixedRounds];
    }
  }
  for (const auto& r : mode_ranges) {
    if (mixed) break;
    const dim3 grid(r.count * fs.C), block(64);
    const uint32_t* wl = d_worklists ? d_worklists + (uint64_t)mode_slot(r.mode) * wl_stride : nullptr;
    // a group that needs an exponent per point (FD_OWNEX) in a round that also emits the
    // E-step's by-products goes to the general kernel as before
    const uint32_t mode = ((r.mode & FD_OWNEX) && emit_estep) ? 0u : r.mode;
    switch (mode) {
#define FD_LAUNCH(NF, NA, SM, EM, FR, XD)                                                    \
  hipLaunchKernelGGL((k_fast_lkl_fd<NF, NA, SM, EM, FR, XD>), grid, block, 0, st, arr, fs.T,    \
                     fs.C, dg, r.begin, part, emit, wl)
#define FD_CASE1(NF, NA, SM, XD)                                              \
  case fd_mode(NF, NA, SM, XD == 2):                                          \
    if (fresh && fs.packed) FD_LAUNCH(NF, NA, SM, true, SRC_FRESH_PACKED, XD); \
    else if (fresh) FD_LAUNCH(NF, NA, SM, true, SRC_FRESH, XD);               \
    else if (emit_estep) FD_LAUNCH(NF, NA, SM, true, SRC_PLAIN, XD);          \
    else FD_LAUNCH(NF, NA, SM, false, SRC_PLAIN, XD);                         \
    break;                                                                    \
  case fd_mode(NF, NA, SM, XD == 2) | FD_OWNEX:                               \
    hipLaunchKernelGGL((k_fast_lkl_fd<NF, NA, SM, false, SRC_PLAIN, XD, true>), grid, block, 0, st, arr, \
                       fs.T, fs.C, dg, r.begin, part, emit, wl);                \
    break;
      // (every recognised pattern is padded to the full one: fast_dev.hpp, fd_pad)
      FD_CASE1(2, 2, false, 4)
      FD_CASE1(2, 2, true, 4)
      FD_CASE1(2, 2, false, 2)
      FD_CASE1(2, 2, true, 2)
#undef FD_CASE1
#undef FD_LAUNCH
      default:
        if (fresh && fs.packed)
          hipLaunchKernelGGL((k_fast_lkl_chunks<MAXP, SRC_FRESH_PACKED>), grid, block, 0, st, arr,
                             fs.T, fs.C, dg, r.begin, part, emit, wl);
        else if (fresh)
          hipLaunchKernelGGL((k_fast_lkl_chunks<MAXP, SRC_FRESH>), grid, block, 0, st, arr, fs.T,
                             fs.C, dg, r.begin, part, emit, wl);
        else
          hipLaunchKernelGGL((k_fast_lkl_chunks<MAXP, SRC_PLAIN>), grid, block, 0, st, arr, fs.T,
                             fs.C, dg, r.begin, part, emit, wl);
    }
  }
  if (fresh) fs.e_stale = false;
  if (fs.shard.world > 1) {
    // this handle's sites are a range of the data set's: its operators to everybody, theirs back
    SiteShard& sh = fs.shard;
    if ((uint64_t)n_pts * 6 > sh.cap) return false;
    hipLaunchKernelGGL(k_fast_lkl_finish<true>, dim3(ng), dim3(64 * MAXP), 0, st, dg, ng, fs.C, part,
                       fs.base_c, sh.send, d_flags, d_all);
    if (hipGetLastError() != hipSuccess) return false;
    if (sh.allgather(sh.user, (uint64_t)n_pts * 6 * sizeof(double)) != 0) return false;
    ++sh.n_gathers;
    hipLaunchKernelGGL(k_fast_shard_combine, dim3(((unsigned)ng * MAXP + 255) / 256), dim3(256), 0, st, dg,
                       ng, sh.recv, sh.world, (uint64_t)n_pts, d_lkl, d_flags, d_all);
    sh.edges_from_round = false;
    if (emit_estep) {  // every individual is in the batch: the E-step's edges too
      hipLaunchKernelGGL(k_fast_shard_edges_from_round, dim3((ng + 255) / 256), dim3(256), 0, st, dg, ng,
                         sh.recv, sh.world, sh.rank, (uint64_t)n_pts, sh.edges, d_all);
      sh.edges_from_round = true;
    }
    return hipGetLastError() == hipSuccess;
  }
  // (device-planned rounds: the planning kernel behind the round finishes its own individual's
  // points -- one launch and one kernel boundary fewer per round)
  if (!finish_elsewhere)
    hipLaunchKernelGGL(k_fast_lkl_finish<false>, dim3(ng), dim3(64 * MAXP), 0, st, dg, ng, fs.C, part,
                       fs.base_c, d_lkl, d_flags, d_all);
  return hipGetLastError() == hipSuccess;
}

bool fast_lkl_launch(FastState& fs, hipStream_t st, double* d_lkl, int* d_flags, bool emit_estep) {
  FastState::LklLane& L = fs.lanes[fs.cur_lane];
  return lkl_launch_groups(fs, st, reinterpret_cast<const GroupDesc*>(L.grp_dev), L.mode_ranges,
                           L.n_groups, L.n_pts, L.part, d_lkl, d_flags, emit_estep);
}

// a round planned on the device (kernels_bfgs.hip): `ranges` = the modes present and their
// counts (begin unused), values to d_lkl[individual * 5 + slot]
bool fast_lkl_launch_planned(FastState& fs, hipStream_t st, const void* d_groups_by_ind,
                             const std::vector<FastState::ModeRange>& ranges, uint32_t n_active,
                             const uint32_t* d_worklists, const uint32_t* d_all, double* part,
                             double* d_lkl, bool emit_estep) {
  // (a site shard exchanges the points' operators at their fixed positions, individual * 5 +
  // slot: every rank plans the same groups, but lists them in an order of its own)
  return lkl_launch_groups(fs, st, reinterpret_cast<const GroupDesc*>(d_groups_by_ind), ranges,
                           n_active, (uint32_t)fs.I * (uint32_t)MAXP, part, d_lkl, nullptr, emit_estep,
                           d_worklists, fs.I, d_all, /*finish_elsewhere=*/fs.shard.world <= 1);
}

bool fast_lkl_covers_everyone(const FastState& fs) {
  const FastState::LklLane& L = fs.lanes[fs.cur_lane];
  // one group per individual (points are grouped by individual, <= MAXP each; an M-step's
  // first round has <= 5 points per individual, so groups == individuals iff all are there)
  return L.n_groups == fs.I;
}

}  // namespace nghmm

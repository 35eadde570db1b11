// kernels_fast_estep.hip -- the E-step: lane-chunk operators and checkpoints, boundary vectors (ordered shuffle scan),
// backward sweep with block-wise forward recomputation, tile-major posteriors; the edges a site
// shard's E-step needs; the lazy emission refresh
// (fast mode, gfx950; kernels_fast.hip's header comment has the design, DESIGN.md section 4 the
// measurements.)
#include "fast_dev.hpp"

namespace nghmm {

namespace {


// E-step: the operator of the handle's whole site range per individual (ordered product of its
// lane-chunk operators) and the range's sum of log e0 -> send[i][6]; one wave per individual
__global__ void __launch_bounds__(64)
k_fast_shard_reduce(const double* __restrict__ lane_ops, uint64_t J, uint32_t C,
                    const double* __restrict__ base_c, double* __restrict__ send) {
  const uint64_t i = blockIdx.x;
  const int lane = threadIdx.x;
  const double* ops = lane_ops + (i * J + (uint64_t)lane * C) * 5;
  constexpr uint32_t PF = 8;
  Op L{1.0, 0.0, 0.0, 1.0, 0};
  for (uint32_t k0 = 0; k0 < C; k0 += PF) {
    Op o[PF];
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u)
      o[u] = op_load(ops + (uint64_t)(k0 + u < C ? k0 + u : C - 1) * 5);
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u)
      if (k0 + u < C) L = op_mul(L, o[u]);
  }
  for (int off = 1; off < 64; off <<= 1) {
    const Op o = op_shfl_down(L, off);
    if ((lane & (2 * off - 1)) == 0) L = op_mul(L, o);
  }
  const double base = base_sum(base_c + i * C, C, lane);
  if (lane == 0) {
    double* o = send + i * 6;
    o[0] = L.a00;
    o[1] = L.a01;
    o[2] = L.a10;
    o[3] = L.a11;
    o[4] = (double)L.ex;
    o[5] = base;
  }
}

// edges[i][8] = the row vector entering this range from the left (u0, u1, exponent), the column
// vector entering it from the right (x0, x1, exponent), the sum of log e0 over all ranges
__global__ void __launch_bounds__(256)
k_fast_shard_edges(const double* __restrict__ recv, uint32_t world, uint32_t rank, uint64_t I,
                   const double* __restrict__ indF, double* __restrict__ edges) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= I) return;
  const double f = indF[i];
  double u0 = 1 - f, u1 = f;
  int uex = 0;
  double base = 0.0;
  for (uint32_t r = 0; r < world; ++r) {
    const double* o = recv + ((uint64_t)r * I + i) * 6;
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
    const Op m = op_load(recv + ((uint64_t)r * I + i) * 6);
    const double n0 = fma(m.a00, x0, m.a01 * x1), n1 = fma(m.a10, x0, m.a11 * x1);
    x0 = n0;
    x1 = n1;
    xex += m.ex;
    renorm2(x0, x1, xex);
  }
  double* e = edges + i * 8;
  e[0] = u0;
  e[1] = u1;
  e[2] = (double)uex;
  e[3] = x0;
  e[4] = x1;
  e[5] = (double)xex;
  e[6] = base;
  e[7] = 0.0;
}

// ---- E-step ---------------------------------------------------------------
// The E-step of a lane-chunk needs the forward vector at every site and the backward
// vector at every site.  Storing either per site costs 8 B written and 8 B read back per
// site and individual; instead the forward pass leaves a CHECKPOINT every CK sites -- the
// 2x2 prefix operator of the lane-chunk up to there, 32 B per CK sites -- and the backward
// sweep recomputes the forward vectors of a block of CK sites from its checkpoint, in
// registers, before walking the block backwards (k_fast_bwd_recompute).  The prefix
// operators do not depend on the vector entering the lane-chunk, so they are produced by
// whichever kernel walks the chunk forward first: k_fast_chunk_ops for a stand-alone
// E-step, or the first objective round of the M-step (same parameters, same emissions:
// lkl_run_fd / k_fast_lkl_chunks with `emit`), which then replaces phase A altogether.
//
// checkpoint layout: ck[((i*C + c)*NBLK + b)*2 + h][64] of double2 = row h of the prefix
// operator of sites [0, b*CK) of lane-chunk (c, lane); b = 0 (identity) is not stored.
// phase A: the operator of every lane-chunk and its checkpoints, one point per individual
__global__ void __launch_bounds__(64)
k_fast_chunk_ops(const double* __restrict__ e_il, const double* __restrict__ pos_il, uint64_t T,
                 uint32_t C, const double* __restrict__ indF, const double* __restrict__ alpha,
                 EmitPtrs out) {
  const uint64_t i = blockIdx.x / C;
  const uint32_t c = blockIdx.x % C;
  const int lane = threadIdx.x;
  const double f = indF[i], al = alpha[i];
  const double q0 = 1 - f, q1 = f;
  Op R{1.0, 0.0, 0.0, 1.0, 0};
  const double* ep = e_il + ((i * C + c) * T) * 64 + lane;
  const double* dp = pos_il + ((uint64_t)c * T) * 64 + lane;
  const uint64_t nblk = T / CK;
  double ecur[UF], enxt[UF];  // emission ratios: the emissions are (1, rho)
  double dcur[UF], dnxt[UF];
#pragma unroll
  for (int u = 0; u < UF; ++u) {
    ecur[u] = ep[(uint64_t)u * 64];  // T is a multiple of 8; arrays carry a group of slack
    dcur[u] = dp[(uint64_t)u * 64];
  }
  for (uint64_t t0 = 0; t0 < T; t0 += UF) {
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const uint64_t t = t0 + UF + u;
      enxt[u] = ep[t * 64];
      dnxt[u] = dp[t * 64];
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const double cc = coanc(al, dcur[u]);
      const double a = 1 - cc;
      op_step(R, cc, cc * ecur[u], a * q0, a * ecur[u] * q1);
    }
    if (((t0 / UF) % (RENORM / UF)) == (RENORM / UF - 1)) {
      renorm(R);
      const uint64_t b = (t0 + UF) / CK;  // checkpoint in front of block b
      if (b < nblk) emit_checkpoint(out.ckpt, i * C + c, nblk, b, lane, R);
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      ecur[u] = enxt[u];
      dcur[u] = dnxt[u];
    }
  }
  renorm(R);
  emit_lane_op(out.lane_ops, i * C + c, lane, R);
}

// phase B: per individual, the vector entering every lane-chunk from the left
// (forward) and from the right (backward), the log-likelihood and the Fw/Bw check.
// One wave per individual: lane l owns the C consecutive lane-chunks l*C .. l*C + C-1,
// multiplies their operators, an ordered shuffle scan over the 64 lanes gives every lane
// the product of everything to its left (right), and the lane then walks its own chunks.
__device__ __forceinline__ Op op_shfl_up(const Op& m, int off) {
  Op o;
  o.a00 = __shfl_up(m.a00, off);
  o.a01 = __shfl_up(m.a01, off);
  o.a10 = __shfl_up(m.a10, off);
  o.a11 = __shfl_up(m.a11, off);
  o.ex = __shfl_up(m.ex, off);
  return o;
}

__global__ void __launch_bounds__(64)
k_fast_bounds(const double* __restrict__ lane_ops, uint64_t J, uint32_t C,
              const double* __restrict__ indF, const double* __restrict__ base_c,
              double* __restrict__ bound, double* __restrict__ ind_lkl, int* __restrict__ flags,
              const double* __restrict__ edges) {
  const uint64_t i = blockIdx.x;
  const int lane = threadIdx.x;
  const double f = indF[i];
  // what enters the handle's sites from the left and from the right: the initial distribution
  // and (1, 1), or -- a site shard -- the other ranges' products (k_fast_shard_edges)
  double q0 = 1 - f, q1 = f, x0 = 1.0, x1 = 1.0;
  int uex = 0, xex = 0;
  if (edges) {
    const double* e = edges + i * 8;
    q0 = e[0];
    q1 = e[1];
    uex = (int)e[2];
    x0 = e[3];
    x1 = e[4];
    xex = (int)e[5];
  }
  const double LN2 = 0.6931471805599453094;
  const double* ops = lane_ops + (i * J + (uint64_t)lane * C) * 5;
  double* bd = bound + (i * J + (uint64_t)lane * C) * 4;

  // A lane's C operators are read three times below, each time as a chain of dependent
  // multiplications: the loads of PF operators are issued together ahead of their chain link,
  // or every link would wait a memory round trip (100 us per call at C = 51, whatever I is).
  constexpr uint32_t PF = 8;
  Op L{1.0, 0.0, 0.0, 1.0, 0};
  for (uint32_t k0 = 0; k0 < C; k0 += PF) {
    Op o[PF];
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u)
      o[u] = op_load(ops + (uint64_t)(k0 + u < C ? k0 + u : C - 1) * 5);
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u)
      if (k0 + u < C) L = op_mul(L, o[u]);
  }

  // forward: product of the lanes to the left
  Op P = L;
  for (int off = 1; off < 64; off <<= 1) {
    const Op o = op_shfl_up(P, off);
    if (lane >= off) P = op_mul(o, P);
  }
  Op E = op_shfl_up(P, 1);
  if (lane == 0) E = Op{1.0, 0.0, 0.0, 1.0, 0};
  double v0 = fma(q0, E.a00, q1 * E.a10), v1 = fma(q0, E.a01, q1 * E.a11);
  int ex = E.ex + uex;
  renorm2(v0, v1, ex);
  for (uint32_t k0 = 0; k0 < C; k0 += PF) {
    Op o[PF];
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u)
      o[u] = op_load(ops + (uint64_t)(k0 + u < C ? k0 + u : C - 1) * 5);
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u) {
      const uint32_t k = k0 + u;
      if (k < C) {
        bd[(uint64_t)k * 4 + 0] = v0;
        bd[(uint64_t)k * 4 + 1] = v1;
        const double n0 = fma(v0, o[u].a00, v1 * o[u].a10);
        const double n1 = fma(v0, o[u].a01, v1 * o[u].a11);
        v0 = n0;
        v1 = n1;
        ex += o[u].ex;
        renorm2(v0, v1, ex);
      }
    }
  }
  // (x = (1, 1): v0 + v1 exactly)
  const double lf = __shfl(log(fma(v0, x0, v1 * x1)) + (double)(ex + xex) * LN2, 63);  // lane 63 has walked it all

  // backward: product of the lanes to the right
  Op Sx = L;
  for (int off = 1; off < 64; off <<= 1) {
    const Op o = op_shfl_down(Sx, off);
    if (lane + off < 64) Sx = op_mul(Sx, o);
  }
  Op X = op_shfl_down(Sx, 1);
  if (lane == 63) X = Op{1.0, 0.0, 0.0, 1.0, 0};
  double w0 = fma(X.a00, x0, X.a01 * x1), w1 = fma(X.a10, x0, X.a11 * x1);
  int exb = X.ex + xex;
  renorm2(w0, w1, exb);
  for (uint32_t kk0 = C; kk0 > 0; kk0 = kk0 > PF ? kk0 - PF : 0) {
    Op o[PF];  // operators kk0-1, kk0-2, ...
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u)
      o[u] = op_load(ops + (uint64_t)(kk0 > u ? kk0 - 1 - u : 0) * 5);
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u) {
      if (kk0 > u) {
        const uint32_t k = kk0 - 1 - u;
        bd[(uint64_t)k * 4 + 2] = w0;
        bd[(uint64_t)k * 4 + 3] = w1;
        const double n0 = fma(o[u].a00, w0, o[u].a01 * w1);
        const double n1 = fma(o[u].a10, w0, o[u].a11 * w1);
        w0 = n0;
        w1 = n1;
        exb += o[u].ex;
        renorm2(w0, w1, exb);
      }
    }
  }
  const double lb = __shfl(log(fma(q0, w0, q1 * w1)) + (double)(exb + uex) * LN2, 0);
  // the walks ran on the emissions (1, rho): add sum log e0 (as k_fast_lkl_finish does)
  const double base = edges ? edges[i * 8 + 6] : base_sum(base_c + i * C, C, lane);
  if (lane == 0) {
    ind_lkl[i] = base + lf;
    if (lf != lf || lb != lb || base != base) flags[FLAG_INVALID_LKL] = 1;
    if (fabs(lf - lb) > 0.001) flags[FLAG_FW_BW] = 1;  // EM.cpp:167
  }
}

// phase C: backward sweep with block-wise forward recomputation.  Posterior of the IBD
// state, snapped like check_interv (gen_func.cpp:55-70).  This version (kPost8 == false)
// writes the TILE-MAJOR layout
//   post[(c*T + t)*I + i][lane]      (site (c*64 + lane)*T + t)
// i.e. every wave-store is one contiguous 512 B segment and no transposition pass is
// needed: est_maf reads the layout directly (k_fast_estmaf<.., TILE>); the version in use,
// k_fast_bwd_recompute8, follows.
__global__ void __launch_bounds__(64)
k_fast_bwd_recompute(const double* __restrict__ e_il, const double* __restrict__ pos_il,
                     uint64_t T, uint32_t C, uint64_t S, uint64_t I,
                     const double* __restrict__ indF, const double* __restrict__ alpha,
                     const double* __restrict__ bound, const double2* __restrict__ ckpt,
                     double* __restrict__ post, int* __restrict__ flags) {
  const uint64_t i = blockIdx.x / C;
  const uint32_t c = blockIdx.x % C;
  const int lane = threadIdx.x;
  const double f = indF[i], al = alpha[i];
  const double q0 = 1 - f, q1 = f;
  const uint64_t J = (uint64_t)C * 64;
  const uint64_t j = (uint64_t)c * 64 + lane;
  const double* bd = bound + (i * J + j) * 4;
  const double vin0 = bd[0], vin1 = bd[1];
  double w0 = bd[2], w1 = bd[3];
  const double* ep = e_il + ((i * C + c) * T) * 64 + lane;
  const double* dp = pos_il + ((uint64_t)c * T) * 64 + lane;
  double* pp = post + ((uint64_t)c * T * I + i) * 64 + lane;  // site step t at pp[t * I * 64]
  const uint64_t nblk = T / CK;
  const double2* ck = ckpt + ((i * C + c) * nblk * 2) * 64 + lane;
  const uint64_t tstride = I * 64;
  bool nanflag = false;
  int exd = 0;

  double ecur[CK], enxt[CK];  // emission ratios: the emissions are (1, rho)
  double dcur[CK], dnxt[CK];
  double2 r0c, r1c, r0n, r1n;  // rows of the prefix operator in front of the block
  {
    const uint64_t b = nblk - 1;
#pragma unroll
    for (int u = 0; u < CK; ++u) {
      ecur[u] = ep[(b * CK + u) * 64];
      dcur[u] = dp[(b * CK + u) * 64];
    }
    r0c = b ? ck[(b * 2) * 64] : double2{1.0, 0.0};
    r1c = b ? ck[(b * 2 + 1) * 64] : double2{0.0, 1.0};
  }
  for (uint64_t b = nblk;;) {
    --b;
    if (b > 0) {  // the block in front: in flight while this one is computed
      const uint64_t bn = b - 1;
#pragma unroll
      for (int u = 0; u < CK; ++u) {
        enxt[u] = ep[(bn * CK + u) * 64];
        dnxt[u] = dp[(bn * CK + u) * 64];
      }
      r0n = bn ? ck[(bn * 2) * 64] : double2{1.0, 0.0};
      r1n = bn ? ck[(bn * 2 + 1) * 64] : double2{0.0, 1.0};
    }
    // forward vectors of the block's sites, from the checkpoint
    double v0 = fma(vin0, r0c.x, vin1 * r1c.x);
    double v1 = fma(vin0, r0c.y, vin1 * r1c.y);
    double f0[CK], f1[CK], cc[CK];
#pragma unroll
    for (int u = 0; u < CK; ++u) {
      cc[u] = coanc(al, dcur[u]);
      const double a = 1 - cc[u];
      const double sm = v0 + v1;
      v0 = fma(a * q0, sm, cc[u] * v0);
      v1 = fma(a * q1, sm, cc[u] * v1) * ecur[u];
      if (u == CK / 2 - 1) {  // the scale of (v0, v1) does not matter: keep it in range
        int dummy = 0;
        renorm2(v0, v1, dummy);
      }
      f0[u] = v0;
      f1[u] = v1;
    }
    // backward through the block: posterior, then the beta step
#pragma unroll
    for (int u = CK - 1; u >= 0; --u) {
      const uint64_t t = b * CK + u;
      const double x0 = f0[u] * w0, x1 = f1[u] * w1;
      double g1 = x1 * rcp_nr2(x0 + x1);  // 0/0 (no probability mass) stays NaN
      if (j * T + t < S) {
        if (g1 != g1) nanflag = true;
        // check_interv (gen_func.cpp:55-70)
        if (g1 < kEPS) g1 = 0;
        else if (g1 > 1 - kEPS) g1 = 1;
        pp[t * tstride] = g1;
      }
      // beta step: w'_k = c u_k + a (q . u),  u = e * w
      const double a = 1 - cc[u];
      const double u0 = w0, u1 = ecur[u] * w1;
      const double sq = a * fma(q0, u0, q1 * u1);
      w0 = fma(cc[u], u0, sq);
      w1 = fma(cc[u], u1, sq);
    }
    renorm2(w0, w1, exd);
    if (b == 0) break;
#pragma unroll
    for (int u = 0; u < CK; ++u) {
      ecur[u] = enxt[u];
      dcur[u] = dnxt[u];
    }
    r0c = r0n;
    r1c = r1n;
  }
  if (nanflag) flags[FLAG_NAN] = 1;
}

// The same sweep for the [tile row][i / 8][l][i % 8] layout (kPost8).  A workgroup of four waves
// is the eight individuals of a group x one half (32) of the lane-chunks of chunk c: a wave
// walks 32 lane-chunks of two individuals (every load two 256 B segments), the posteriors of a
// block of CK sites are staged in LDS and leave as contiguous 2 KB stores.  Four waves, because
// three such workgroups fit a CU at this kernel's three waves per SIMD (eight-wave workgroups
// -- a whole tile row block per store -- fit once: 4.5 instead of 4.1 ms).
__global__ void __launch_bounds__(256)
k_fast_bwd_recompute8(const double* __restrict__ e_il, const double* __restrict__ pos_il,
                     uint64_t T, uint32_t C, uint64_t S, uint64_t I,
                     const double* __restrict__ indF, const double* __restrict__ alpha,
                     const double* __restrict__ bound, const double2* __restrict__ ckpt,
                     double* __restrict__ post, int* __restrict__ flags) {
  // (a partial last group repeats its last individual: the copies land in the layout's padding)
  __shared__ double stage[2][CK][8][40];  // [buffer][site of the block][individual][chunk, padded]
  const uint64_t grp = blockIdx.x / ((uint64_t)C * 2);
  const uint32_t c = (uint32_t)((blockIdx.x >> 1) % C);
  const int half = blockIdx.x & 1;
  const int lp = threadIdx.x & 31;                                    // chunk within the half
  const int m8 = (threadIdx.x >> 6) * 2 + ((threadIdx.x >> 5) & 1);  // individual within the group
  const int lane = half * 32 + lp;                                    // the lane-chunk, 0..63
  const uint64_t i = (grp * 8 + m8 < I) ? grp * 8 + m8 : I - 1;
  const double f = indF[i], al = alpha[i];
  const double q0 = 1 - f, q1 = f;
  const uint64_t J = (uint64_t)C * 64;
  const uint64_t j = (uint64_t)c * 64 + lane;
  const double* bd = bound + (i * J + j) * 4;
  const double vin0 = bd[0], vin1 = bd[1];
  double w0 = bd[2], w1 = bd[3];
  const double* ep = e_il + ((i * C + c) * T) * 64 + lane;
  const double* dp = pos_il + ((uint64_t)c * T) * 64 + lane;
  // the half's 32 x 8 block of tile row c*T + t: 256 contiguous doubles, thread k's at [k]
  double* pp = post + (uint64_t)c * T * post_tile_doubles(I) + grp * 512 + half * 256 + threadIdx.x;
  const uint64_t nblk = T / CK;
  const double2* ck = ckpt + ((i * C + c) * nblk * 2) * 64 + lane;
  const uint64_t tstride = post_tile_doubles(I);
  bool nanflag = false;
  int exd = 0;

  double ecur[CK], enxt[CK];  // emission ratios: the emissions are (1, rho)
  double dcur[CK], dnxt[CK];
  double2 r0c, r1c, r0n, r1n;  // rows of the prefix operator in front of the block
  {
    const uint64_t b = nblk - 1;
#pragma unroll
    for (int u = 0; u < CK; ++u) {
      ecur[u] = ep[(b * CK + u) * 64];
      dcur[u] = dp[(b * CK + u) * 64];
    }
    r0c = b ? ck[(b * 2) * 64] : double2{1.0, 0.0};
    r1c = b ? ck[(b * 2 + 1) * 64] : double2{0.0, 1.0};
  }
  for (uint64_t b = nblk;;) {
    --b;
    if (b > 0) {  // the block in front: in flight while this one is computed
      const uint64_t bn = b - 1;
#pragma unroll
      for (int u = 0; u < CK; ++u) {
        enxt[u] = ep[(bn * CK + u) * 64];
        dnxt[u] = dp[(bn * CK + u) * 64];
      }
      r0n = bn ? ck[(bn * 2) * 64] : double2{1.0, 0.0};
      r1n = bn ? ck[(bn * 2 + 1) * 64] : double2{0.0, 1.0};
    }
    // forward vectors of the block's sites, from the checkpoint
    double v0 = fma(vin0, r0c.x, vin1 * r1c.x);
    double v1 = fma(vin0, r0c.y, vin1 * r1c.y);
    double f0[CK], f1[CK], cc[CK];
#pragma unroll
    for (int u = 0; u < CK; ++u) {
      cc[u] = coanc(al, dcur[u]);
      const double a = 1 - cc[u];
      const double sm = v0 + v1;
      v0 = fma(a * q0, sm, cc[u] * v0);
      v1 = fma(a * q1, sm, cc[u] * v1) * ecur[u];
      if (u == CK / 2 - 1) {  // the scale of (v0, v1) does not matter: keep it in range
        int dummy = 0;
        renorm2(v0, v1, dummy);
      }
      f0[u] = v0;
      f1[u] = v1;
    }
    // backward through the block: posterior, then the beta step
#pragma unroll
    for (int u = CK - 1; u >= 0; --u) {
      const uint64_t t = b * CK + u;
      const double x0 = f0[u] * w0, x1 = f1[u] * w1;
      double g1 = x1 * rcp_nr2(x0 + x1);  // 0/0 (no probability mass) stays NaN
      if (j * T + t < S && g1 != g1) nanflag = true;
      // check_interv (gen_func.cpp:55-70)
      if (g1 < kEPS) g1 = 0;
      else if (g1 > 1 - kEPS) g1 = 1;
      stage[b & 1][u][m8][lp] = g1;
      // beta step: w'_k = c u_k + a (q . u),  u = e * w
      const double a = 1 - cc[u];
      const double u0 = w0, u1 = ecur[u] * w1;
      const double sq = a * fma(q0, u0, q1 * u1);
      w0 = fma(cc[u], u0, sq);
      w1 = fma(cc[u], u1, sq);
    }
    renorm2(w0, w1, exd);
    // the block's 8 x 32 x 8 posteriors leave through LDS: thread k writes chunk k / 8 of
    // individual k % 8, i.e. the 256 threads store 2 KB contiguous per site (the buffers
    // alternate: the barrier of block b orders its reads before the writes of block b - 2)
    __syncthreads();
    {
      const int rl = threadIdx.x >> 3, rm = threadIdx.x & 7;
#pragma unroll
      for (int u = 0; u < CK; ++u) pp[(b * CK + u) * tstride] = stage[b & 1][u][rm][rl];
    }
    if (b == 0) break;
#pragma unroll
    for (int u = 0; u < CK; ++u) {
      ecur[u] = enxt[u];
      dcur[u] = dnxt[u];
    }
    r0c = r0n;
    r1c = r1n;
  }
  if (nanflag) flags[FLAG_NAN] = 1;
}

// tile-major posteriors -> site-major [S][I] (multi-GPU packing, host read-back, est_maf
// with more individuals than one wave holds); tile = (c, t) x 64 lanes x 64 individuals,
// 16-byte accesses on both sides: a thread reads two lanes of one individual and writes
// two individuals of one site (I even; odd I takes the 8-byte path)
template <bool PAIRS>
__global__ void __launch_bounds__(256)
k_fast_post_to_site_major(const double* __restrict__ post, uint64_t I, uint64_t S, uint64_t T,
                          uint32_t C, double* __restrict__ marg) {
  constexpr int TI = 64;
  __shared__ double tile[64][TI + 2];  // [lane][individual]
  const uint64_t n_it = (I + TI - 1) / TI;
  const uint64_t ct = blockIdx.x / n_it;  // c * T + t
  const uint64_t i0 = (blockIdx.x % n_it) * TI;
  const uint64_t c = ct / T, t = ct % T;
  if constexpr (PAIRS) {
    const int lp = threadIdx.x & 31, ty = threadIdx.x >> 5;  // lane pair, 8 individuals a pass
    for (int ii = ty; ii < TI; ii += 8) {
      const uint64_t i = i0 + ii;
      if (i < I) {
        if constexpr (kPost8) {  // lanes are 8 doubles apart: two 8-byte reads
          tile[2 * lp][ii] = post[post_lane_off(ct, 2 * lp, I) + post_ind_off(i)];
          tile[2 * lp + 1][ii] = post[post_lane_off(ct, 2 * lp + 1, I) + post_ind_off(i)];
        } else {
          const double2 v = *reinterpret_cast<const double2*>(post + (ct * I + i) * 64 + 2 * lp);
          tile[2 * lp][ii] = v.x;
          tile[2 * lp + 1][ii] = v.y;
        }
      }
    }
    __syncthreads();
    const int ip = threadIdx.x & 31, tz = threadIdx.x >> 5;  // individual pair, 8 sites a pass
    for (int ll = tz; ll < 64; ll += 8) {
      const uint64_t s = (c * 64 + ll) * T + t;
      const uint64_t i = i0 + 2 * ip;
      if (s < S && i < I)  // I even: i + 1 < I as well, and s * I + i is even
        *reinterpret_cast<double2*>(marg + s * I + i) = double2{tile[ll][2 * ip], tile[ll][2 * ip + 1]};
    }
  } else {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int ii = ty; ii < TI; ii += 4) {
      const uint64_t i = i0 + ii;
      if (i < I) tile[tx][ii] = post[post_lane_off(ct, tx, I) + post_ind_off(i)];
    }
    __syncthreads();
    for (int ll = ty; ll < 64; ll += 4) {
      const uint64_t s = (c * 64 + ll) * T + t;
      const uint64_t i = i0 + tx;
      if (s < S && i < I) marg[s * I + i] = tile[ll][tx];
    }
  }
}

// ---- emissions --------------------------------------------------------------
// Stand-alone refresh of the emission ratios and of sum log e0 (an E-step or an objective
// call that no fresh forward walk precedes): one wave per (individual, chunk) runs the fresh
// walk's source over its sites -- the interleaved likelihoods (or codes) and frequencies in,
// the ratios out, every access one contiguous segment per wave-instruction.
template <int SRC>
__global__ void __launch_bounds__(64)
k_fast_refresh(LklArrays arr, uint64_t T, uint32_t C) {
  static_assert(SRC != SRC_PLAIN, "a refresh computes the emissions");
  const uint64_t w = blockIdx.x;  // i * C + c
  const uint32_t c = (uint32_t)(w % C);
  const int lane = threadIdx.x;
  using Src = SrcOf<SRC>;
  Src src(arr, (w * T) * 64 + lane, ((uint64_t)c * T) * 64 + lane);
  for (uint64_t t0 = 0; t0 < T; t0 += RENORM) {  // T is a multiple of RENORM
    typename Src::Buf buf[RENORM];
#pragma unroll
    for (int u = 0; u < RENORM; ++u) buf[u] = src.load(t0 + u);
#pragma unroll
    for (int u = 0; u < RENORM; ++u) {
      double rho, d;
      src.get(buf[u], t0 + u, rho, d);
    }
    src.rescale();
  }
  const double bl = wave_sum(src.base.log_value());
  if (lane == 0) arr.base_c[w] = bl + (arr.gl_scale_c ? arr.gl_scale_c[w] : 0.0);
}

}  // namespace


bool fast_refresh_emissions(FastState& fs, hipStream_t st, const double* d_freq, int* d_flags) {
  // from the frequencies as they are now (the table may predate a parameter upload)
  if (!fast_refresh_freq_table(fs, st, d_freq, d_flags)) return false;
  const dim3 grid((unsigned)(fs.I * fs.C)), block(64);
  if (fs.packed)
    hipLaunchKernelGGL((k_fast_refresh<SRC_FRESH_PACKED>), grid, block, 0, st, lkl_arrays(fs), fs.T,
                       fs.C);
  else
    hipLaunchKernelGGL((k_fast_refresh<SRC_FRESH>), grid, block, 0, st, lkl_arrays(fs), fs.T, fs.C);
  fs.e_stale = false;
  return hipGetLastError() == hipSuccess;
}

bool fast_estep(FastState& fs, hipStream_t st, const double* d_indF, const double* d_alpha,
                double* d_ind_lkl, int* d_flags, bool have_forward_walk) {
  const double* e2 = fs.e_il;
  double2* ck = reinterpret_cast<double2*>(fs.ckpt);
  const unsigned waves = (unsigned)(fs.I * fs.C);
  if (!have_forward_walk)
    hipLaunchKernelGGL(k_fast_chunk_ops, dim3(waves), dim3(64), 0, st, e2, fs.pos_il, fs.T, fs.C,
                       d_indF, d_alpha, EmitPtrs{fs.lane_ops, ck});
  const double* edges = nullptr;
  if (fs.shard.world > 1) {
    SiteShard& sh = fs.shard;
    if (!(have_forward_walk && sh.edges_from_round)) {
      if (fs.I * 6 > sh.cap) return false;
      hipLaunchKernelGGL(k_fast_shard_reduce, dim3((unsigned)fs.I), dim3(64), 0, st, fs.lane_ops, fs.J,
                         fs.C, fs.base_c, sh.send);
      if (hipGetLastError() != hipSuccess) return false;
      if (sh.allgather(sh.user, fs.I * 6 * sizeof(double)) != 0) return false;
      ++sh.n_gathers;
      hipLaunchKernelGGL(k_fast_shard_edges, dim3((unsigned)((fs.I + 255) / 256)), dim3(256), 0, st,
                         sh.recv, sh.world, sh.rank, fs.I, d_indF, sh.edges);
    }
    sh.edges_from_round = false;
    edges = sh.edges;
  }
  hipLaunchKernelGGL(k_fast_bounds, dim3((unsigned)fs.I), dim3(64), 0, st, fs.lane_ops, fs.J, fs.C,
                     d_indF, fs.base_c, fs.bound, d_ind_lkl, d_flags, edges);
  if (kPost8)
    hipLaunchKernelGGL(k_fast_bwd_recompute8, dim3((unsigned)(((fs.I + 7) / 8) * fs.C * 2)), dim3(256),
                       0, st, e2, fs.pos_il, fs.T, fs.C, fs.S, fs.I, d_indF, d_alpha, fs.bound, ck,
                       fs.post, d_flags);
  else
    hipLaunchKernelGGL(k_fast_bwd_recompute, dim3(waves), dim3(64), 0, st, e2, fs.pos_il, fs.T, fs.C,
                       fs.S, fs.I, d_indF, d_alpha, fs.bound, ck, fs.post, d_flags);
  return hipGetLastError() == hipSuccess;
}

bool fast_post_to_site_major(FastState& fs, hipStream_t st, double* d_marg) {
  const uint64_t n_it = (fs.I + 63) / 64;
  const dim3 grid((unsigned)((uint64_t)fs.C * fs.T * n_it)), block(256);
  // 16-byte stores need s * I + i even for even i, i.e. an even number of individuals, and
  // a 16-byte aligned destination
  if (fs.I % 2 == 0 && (reinterpret_cast<uintptr_t>(d_marg) & 15) == 0)
    hipLaunchKernelGGL((k_fast_post_to_site_major<true>), grid, block, 0, st, fs.post, fs.I, fs.S,
                       fs.T, fs.C, d_marg);
  else
    hipLaunchKernelGGL((k_fast_post_to_site_major<false>), grid, block, 0, st, fs.post, fs.I, fs.S,
                       fs.T, fs.C, d_marg);
  return hipGetLastError() == hipSuccess;
}

}  // namespace nghmm

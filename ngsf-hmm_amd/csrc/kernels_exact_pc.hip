// kernels_exact_pc.hip -- exact-mode forward / backward recursions as producer-consumer
// workgroups (gfx950).
//
// What a recursion step costs in the reference's formulation (shared/HMM.cpp:6-60,130-139):
// one exp and four logs for the transition probabilities of the site, and two logsums of two
// terms (shared/gen_func.cpp:135-151) for the two states.  Only the logsums depend on the
// recursion state.  kernels_exact.hip's first version ran everything in the one lane that owns
// a chain -- 11 exp/log calls in sequence per site, 79 waves on a 1024-SIMD chip for a round of
// 5 x 1000 probe points: 142 s per EM iteration at 1000 x 1M.
//
// Here a workgroup is one CONSUMER wave and two PRODUCER waves for 32 chains:
//   * the producers compute, for a block of sites ahead of the consumer, everything that does
//     not depend on the recursion state -- det_exp(-alpha d), the four det_log of calc_trans,
//     for the backward sweep also their sums with the site's emissions -- in parallel over
//     (chain, site), and leave it in an LDS ring (two blocks, one barrier per block);
//   * the consumer gives TWO lanes to a chain, one per state: each lane computes its state's
//     logsum (one exp, one log: det_logsum2_chain) and the pair swaps results through DPP.
// The chain's dependent work per site drops from ~11 to 2 transcendental calls.  Every value
// is produced by the same operations on the same operands as before (detmath.h's select-form
// variants return det_exp / det_log's bits), so every result is bit-identical to the serial
// kernels and to the oracle's det build -- tests/test_gpu_parity.py holds both to it.
#include <hip/hip_runtime.h>

#include <cmath>
#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "detmath.h"
#include "glview.hpp"
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace nghmm {

namespace {

#include "exact_dev.hpp"

constexpr int PC_CH = 32;                            // chains per workgroup
constexpr int PC_B = 20;                             // sites per block of the ring (2 blocks: 60 KB of LDS)
constexpr int PC_NPROD = 2;                          // producer waves
constexpr int PC_THREADS = 64 * (1 + PC_NPROD);
constexpr int PC_PLANES = 3;                         // doubles per consumer lane and site
constexpr int PC_SSTRIDE = 64 * PC_NPROD / PC_CH;    // sites a producer pass covers
constexpr int PC_ITEMS = PC_B / PC_SSTRIDE;          // (chain, site) items per producer lane and block
static_assert(PC_B % PC_SSTRIDE == 0, "a block is a whole number of producer passes");

// calc_trans for the four (k, l) of a site (shared/HMM.cpp:130-139), select-form exp/log
__device__ __forceinline__ Trans calc_trans_sel(double q0, double q1, double alpha, double d) {
  Trans t;
  const double c = det_exp_sel(-alpha * d);
  const double b0 = (1 - c) * q0;
  const double b1 = (1 - c) * q1;
  t.t10 = det_log_pos(b0);
  t.t00 = det_log_pos(b0 + c);
  t.t01 = det_log_pos(b1);
  t.t11 = det_log_pos(b1 + c);
  return t;
}

// the value of the other lane of a chain's pair (lanes 2c and 2c + 1)
__device__ __forceinline__ double pair_swap(double v) {
  const uint64_t b = ngh_bits(v);
  // quad_perm [1, 0, 3, 2]
  const int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)b, 0xb1, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(b >> 32), 0xb1, 0xf, 0xf, true);
  return ngh_from_bits(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// v_max_f64 / v_min_f64 of two values known not to be signalling NaNs (the compiler's
// fmax / fmin first canonicalise both operands)
__device__ __forceinline__ double vmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// the reference's logsum with its operands in state order, out of line (rare route)
__device__ __attribute__((noinline)) double chain_logsum_cold(double to, double tx, int l) {
  return det_logsum2(l ? tx : to, l ? to : tx);
}

// One site of a chain for the lane that owns state l of it: `to` is the term through the
// lane's own state, `tx` the term through the other one; the reference's logsum takes them
// in state order, (to, tx) for l = 0 and (tx, to) for l = 1 (shared/HMM.cpp:12-24,40-52,
// shared/gen_func.cpp:135-151).  The larger term is the same value whichever way a tie is
// broken, so for two finite terms max / min stand for the reference's `a >= b ? a : b`, the
// larger term's own exponential is exp(0) = 1, and what is left is log(1 + exp(m - M)) + M
// with m - M in det_exp_chain_fast's domain.  Anything else (a NaN, an infinite term, terms
// closer than 2^-28, a difference in (-745.13, -708)) sends the whole wave -- a vote, so the
// branch is uniform -- through the loop as the reference writes it.
__device__ __forceinline__ double chain_logsum(double to, double tx, int l, int& bad) {
  const double M = vmax(to, tx);
  const double d = vmin(to, tx) - M;  // NaN if either term is
  const int under = d < -7.45133219101941108420e+02;
  if (NGH_ANY(!det_exp_chain_ok(d, under) | (M == __builtin_huge_val()))) {
    bad |= (to != to) | (tx != tx);  // the recursions' NaN test (HMM.cpp:18-21,45-48)
    return chain_logsum_cold(to, tx, l);
  }
  return det_log_chain_fast(det_exp_chain_fast(d, under) + 1.0) + M;
}

// one ring slot: [site of the block][plane][consumer lane]
using Ring = double[2][PC_B][PC_PLANES][64];

// ---------------------------------------------------------------------------------------
// forward (shared/HMM.cpp:6-28) for n_pts (individual, F, alpha) points; STORE_FW: the E-step's
// pass, which keeps Fw [S+1][I][2]
template <bool STORE_FW>
__global__ void __launch_bounds__(PC_THREADS)
k_forward_exact_pc(const double* __restrict__ eprob, const double* __restrict__ pos, uint64_t S,
                   uint64_t I, uint32_t n_pts, const uint32_t* __restrict__ ind,
                   const double* __restrict__ Fv, const double* __restrict__ Av,
                   double* __restrict__ lkl_out, double* __restrict__ fw,
                   int* __restrict__ flags) {
  __shared__ __attribute__((aligned(16))) Ring ring;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const uint64_t nblk = (S + PC_B - 1) / PC_B;
  const double2* e2 = reinterpret_cast<const double2*>(eprob);

  if (tid < 64) {
    // ---- consumer: lane 2c + l is state l of chain c ----
    __builtin_amdgcn_s_setprio(3);  // the chain is what the launch waits for
    const int l = lane & 1;
    uint32_t p = blockIdx.x * PC_CH + (lane >> 1);
    const bool valid = p < n_pts;
    if (!valid) p = n_pts - 1;  // a duplicate of the last chain: computed, never stored
    const uint64_t i = ind ? ind[p] : p;
    const double f = Fv[p];
    // own = Fw[s][l], oth = Fw[s][1 - l]; HMM.cpp:9-10 with q = (1 - F, F), EM.cpp:415
    double own = det_log(l ? f : 1 - f), oth = det_log(l ? 1 - f : f);
    if (STORE_FW && valid) fw[i * 2 + l] = own;
    int bad = 0;
    double A, B, E;  // the site's ring entries, read one site ahead
    // HMM.cpp:12-24 for state l: logsum_k(Fw[s-1][k] + log T_s(k, l)) + e[s][l]
    auto site = [&](const double* rn, uint64_t s) {
      const double a = A, b = B, e = E;
      A = rn[0];
      B = rn[64];
      E = rn[128];
      const double cur = chain_logsum(own + a, oth + b, l, bad) + e;
      own = cur;
      oth = pair_swap(cur);
      if (STORE_FW && valid) fw[((s + 1) * I + i) * 2 + l] = cur;
    };
    for (uint64_t step = 0; step <= nblk; ++step) {
      if (step >= 1) {
        const uint64_t s0 = (step - 1) * PC_B;
        const double* rb = &ring[(step - 1) & 1][0][0][lane];
        A = rb[0];
        B = rb[64];
        E = rb[128];
        // (the last site of a block reads its own entries again instead of the next block's)
        if (S - s0 >= (uint64_t)PC_B) {
#pragma unroll 4
          for (int u = 0; u < PC_B; ++u)
            site(rb + (u + 1 < PC_B ? u + 1 : u) * (PC_PLANES * 64), s0 + u);
        } else {
          const int ns = (int)(S - s0);
          for (int u = 0; u < ns; ++u)
            site(rb + (u + 1 < ns ? u + 1 : u) * (PC_PLANES * 64), s0 + u);
        }
      }
      __syncthreads();
    }
    if (valid && l == 0) lkl_out[p] = det_logsum2(own, oth);
    if (bad) flags[FLAG_INVALID_LKL] = 1;
  } else {
    // ---- producers: lane handles chain c at the sites so, so + PC_SSTRIDE, ... of a block ----
    const int pl = tid - 64;
    const int c = pl % PC_CH, so = pl / PC_CH;
    uint32_t p = blockIdx.x * PC_CH + c;
    if (p >= n_pts) p = n_pts - 1;
    const uint64_t i = ind ? ind[p] : p;
    const double f = Fv[p], a = Av[p];
    const double q0 = 1 - f, q1 = f;
    double dn[PC_ITEMS];
    double2 en[PC_ITEMS];
    auto fetch = [&](uint64_t blk) {
#pragma unroll
      for (int k = 0; k < PC_ITEMS; ++k) {
        const uint64_t s = blk * PC_B + so + k * PC_SSTRIDE;
        const bool v = s < S;
        dn[k] = v ? pos[s] : 0.0;
        en[k] = v ? e2[s * I + i] : double2{0, 0};
      }
    };
    fetch(0);
    for (uint64_t step = 0; step <= nblk; ++step) {
      if (step < nblk) {
        double dc[PC_ITEMS];
        double2 ec[PC_ITEMS];
#pragma unroll
        for (int k = 0; k < PC_ITEMS; ++k) {
          dc[k] = dn[k];
          ec[k] = en[k];
        }
        if (step + 1 < nblk) fetch(step + 1);
#pragma unroll
        for (int k = 0; k < PC_ITEMS; ++k) {
          const int u = so + k * PC_SSTRIDE;
          if (step * PC_B + u < S) {
            const Trans t = calc_trans_sel(q0, q1, a, dc[k]);
            double* w = &ring[step & 1][u][0][2 * c];
            *reinterpret_cast<double2*>(w) = double2{t.t00, t.t11};       // l -> l
            *reinterpret_cast<double2*>(w + 64) = double2{t.t10, t.t01};  // 1 - l -> l
            *reinterpret_cast<double2*>(w + 128) = ec[k];
          }
        }
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward + posteriors + Fw/Bw check (shared/HMM.cpp:33-60, EM.cpp:166-185), one chain per
// individual.  Block b of the ring holds the reference sites s = S - (b PC_B + u), u = 0 ..
// LDS of the backward kernel: the consumer's two inputs per site and lane, and the Bw values it
// leaves behind for the posteriors (2 x 20 x (2 + 1) x 64 doubles = 60 KB, as before)
struct BwRing {
  double in[2][PC_B][2][64];
  double out[2][PC_B][64];
};

__global__ void __launch_bounds__(PC_THREADS)
k_backward_exact_pc(const double* __restrict__ eprob, const double* __restrict__ pos,
                    const double* __restrict__ fw, uint64_t S, uint64_t I,
                    const double* __restrict__ indF, const double* __restrict__ alpha,
                    const double* __restrict__ ind_lkl, double* __restrict__ marg,
                    int* __restrict__ flags) {
  __shared__ __attribute__((aligned(16))) BwRing ring;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const uint64_t nblk = (S + PC_B - 1) / PC_B;
  const double2* e2 = reinterpret_cast<const double2*>(eprob);
  const double2* f2 = reinterpret_cast<const double2*>(fw);

  // Three roles over the steps 0 .. nblk + 1 (one barrier per step): the producers fill block
  // `step`, the consumer walks block step - 1, and the producers -- who have time to spare --
  // turn the Bw values the consumer left of block step - 2 into posteriors.  (Round 3's consumer
  // formed the posterior itself: a det_exp that is not on the recursion's dependency chain but
  // shares the one wave's issue slots with it: 0.60 us per site against the forward sweep's 0.34.)
  if (tid < 64) {
    // ---- consumer: lane 2c + k is state k of individual c's Bw ----
    __builtin_amdgcn_s_setprio(3);
    const int k = lane & 1;
    uint64_t i = (uint64_t)blockIdx.x * PC_CH + (lane >> 1);
    if (i >= I) i = I - 1;
    const double f = indF[i];
    double own = det_log(1.0), oth = det_log(1.0);  // Bw[S][k], Bw[S][1 - k]   (HMM.cpp:37)
    int bad = 0;
    double A, B;  // the site's ring entries, read one site ahead
    auto site = [&](const double* rn, double* wo) {
      const double a = A, b = B;
      A = rn[0];
      B = rn[64];
      *wo = own;  // Bw[s][k]: the posterior of site s is the producers' (EM.cpp:184)
      // Bw[s-1][k] = logsum_l(log T_s(k, l) + e[s][l] + Bw[s][l])   (HMM.cpp:40-52)
      const double nb = chain_logsum(a + own, b + oth, k, bad);
      own = nb;
      oth = pair_swap(nb);
    };
    for (uint64_t step = 0; step <= nblk + 1; ++step) {
      if (step >= 1 && step <= nblk) {
        const uint64_t r0 = (step - 1) * PC_B;
        const double* rb = &ring.in[(step - 1) & 1][0][0][lane];
        double* wb = &ring.out[(step - 1) & 1][0][lane];
        A = rb[0];
        B = rb[64];
        if (S - r0 >= (uint64_t)PC_B) {
#pragma unroll 4
          for (int u = 0; u < PC_B; ++u) site(rb + (u + 1 < PC_B ? u + 1 : u) * (2 * 64), wb + u * 64);
        } else {
          const int ns = (int)(S - r0);
          for (int u = 0; u < ns; ++u) site(rb + (u + 1 < ns ? u + 1 : u) * (2 * 64), wb + u * 64);
        }
      }
      __syncthreads();
    }
    double b0 = k ? oth : own, b1 = k ? own : oth;
    b0 += det_log(1 - f);  // HMM.cpp:55-56
    b1 += det_log(f);
    const double bl = det_logsum2(b0, b1);
    const double2 fS = f2[S * I + i];
    const double fl = det_logsum2(fS.x, fS.y);
    const double diff = fl - bl;
    const double adiff = (diff >= 0) ? diff : -diff;  // the reference's abs macro
    if (adiff > 0.001) flags[FLAG_FW_BW] = 1;         // EM.cpp:167
    if (bad) flags[FLAG_INVALID_LKL] = 1;
  } else {
    // ---- producers ----
    const int pl = tid - 64;
    const int c = pl % PC_CH, so = pl / PC_CH;
    uint64_t i = (uint64_t)blockIdx.x * PC_CH + c;
    const bool vi = i < I;
    if (!vi) i = I - 1;
    const double f = indF[i], a = alpha[i], lkl = ind_lkl[i];
    const double q0 = 1 - f, q1 = f;
    double dn[PC_ITEMS];
    double2 en[PC_ITEMS], fn[PC_ITEMS];
    double2 fh[2][PC_ITEMS];  // Fw of the items of the last two blocks produced (by step parity)
    bool nanflag = false;
    auto fetch = [&](uint64_t blk) {
#pragma unroll
      for (int k = 0; k < PC_ITEMS; ++k) {
        const uint64_t r = blk * PC_B + so + k * PC_SSTRIDE;
        const bool v = r < S;
        const uint64_t s = S - (v ? r : 0);
        dn[k] = v ? pos[s - 1] : 0.0;
        en[k] = v ? e2[(s - 1) * I + i] : double2{0, 0};
        fn[k] = v ? f2[s * I + i] : double2{0, 0};
      }
    };
    // posteriors of block `blk` (produced two steps ago: its Fw values are in fh[blk & 1])
    auto finish = [&](uint64_t blk, const double2 (&fw_of)[PC_ITEMS]) {
#pragma unroll
      for (int k = 0; k < PC_ITEMS; ++k) {
        const int u = so + k * PC_SSTRIDE;
        const uint64_t r = blk * PC_B + u;
        if (r < S) {
          const uint64_t s = S - r;
          const double2 bw = *reinterpret_cast<const double2*>(&ring.out[blk & 1][u][2 * c]);
          // marg_prob[i][s][k] = check_interv(exp(Fw + Bw - lkl)) (EM.cpp:184); state 0 only
          // for its NaN check
          (void)check_interv(det_exp_sel(bw.x + fw_of[k].x - lkl), nanflag);
          const double m1 = check_interv(det_exp_sel(bw.y + fw_of[k].y - lkl), nanflag);
          if (vi) marg[(s - 1) * I + i] = m1;
        }
      }
    };
    fetch(0);
    for (uint64_t step = 0; step <= nblk + 1; ++step) {
      if (step >= 2) {
        if (step & 1) finish(step - 2, fh[1]);
        else finish(step - 2, fh[0]);
      }
      if (step < nblk) {
        double dc[PC_ITEMS];
        double2 ec[PC_ITEMS];
#pragma unroll
        for (int k = 0; k < PC_ITEMS; ++k) {
          dc[k] = dn[k];
          ec[k] = en[k];
          if (step & 1) fh[1][k] = fn[k];
          else fh[0][k] = fn[k];
        }
        if (step + 1 < nblk) fetch(step + 1);
#pragma unroll
        for (int k = 0; k < PC_ITEMS; ++k) {
          const int u = so + k * PC_SSTRIDE;
          if (step * PC_B + u < S) {
            const Trans t = calc_trans_sel(q0, q1, a, dc[k]);
            double* w = &ring.in[step & 1][u][0][2 * c];
            // lane k's term through its own state, t[k][k] + e_k, and through the other one,
            // t[k][1-k] + e_(1-k) (HMM.cpp:44-45: the reference adds the transition and the
            // emission first, then Bw)
            *reinterpret_cast<double2*>(w) = double2{t.t00 + ec[k].x, t.t11 + ec[k].y};
            *reinterpret_cast<double2*>(w + 64) = double2{t.t01 + ec[k].y, t.t10 + ec[k].x};
          }
        }
      }
      __syncthreads();
    }
    if (nanflag) flags[FLAG_NAN] = 1;
  }
}

}  // namespace

// Dynamic LDS on top of the 60 KB ring, so that a workgroup has its CU (160 KB) to itself when the
// launch is small enough for every workgroup to get one: two chain workgroups on one CU share its
// SIMDs between their producer waves and the consumers that are the critical path -- measured at
// 1000 x 1M with the E-step's 32 workgroups next to a round's 157: forward 0.50 instead of 0.35 s,
// the round 0.39 instead of 0.31 s; with 24 KB of padding neither notices the other.  A launch of
// more workgroups than that (cohorts beyond ~1250 individuals) keeps two per CU: there the idle
// issue slots of one consumer are the other's throughput.  NGHMM_PC_PAD_KB overrides (0: never).
static unsigned pc_pad_bytes(unsigned n_workgroups) {
  static const unsigned pad = [] {
    unsigned kb = 24;
    if (const char* env = std::getenv("NGHMM_PC_PAD_KB")) kb = (unsigned)std::atoi(env);
    return (kb > 96 ? 96u : kb) * 1024u;
  }();
  if (pad == 0 || n_workgroups > 200) return 0u;
  // more than 64 KB of LDS per workgroup must be allowed per kernel AND per device (a process may
  // drive several: nghmm_group_*)
  static std::atomic<uint64_t> allowed{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0u;
  if (!(allowed.load() >> dev & 1)) {
    bool ok = true;
    for (const void* f : {reinterpret_cast<const void*>(k_forward_exact_pc<true>),
                          reinterpret_cast<const void*>(k_forward_exact_pc<false>),
                          reinterpret_cast<const void*>(k_backward_exact_pc)})
      ok = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad) == hipSuccess && ok;
    (void)hipGetLastError();
    if (!ok) return 0u;
    allowed.fetch_or(1ull << dev);
  }
  return pad;
}

void launch_forward_exact_pc(hipStream_t st, const double* eprob, const double* pos, uint64_t S,
                             uint64_t I, uint32_t n_pts, const uint32_t* ind, const double* F,
                             const double* alpha, double* lkl_out, double* fw, int* flags) {
  if (n_pts == 0 || S == 0) return;
  const dim3 grid((n_pts + PC_CH - 1) / PC_CH), block(PC_THREADS);
  if (fw)
    hipLaunchKernelGGL(k_forward_exact_pc<true>, grid, block, pc_pad_bytes(grid.x), st, eprob, pos, S, I, n_pts, ind,
                       F, alpha, lkl_out, fw, flags);
  else
    hipLaunchKernelGGL(k_forward_exact_pc<false>, grid, block, pc_pad_bytes(grid.x), st, eprob, pos, S, I, n_pts, ind,
                       F, alpha, lkl_out, fw, flags);
}

void launch_backward_exact_pc(hipStream_t st, const double* eprob, const double* pos,
                              const double* fw, uint64_t S, uint64_t I, const double* indF,
                              const double* alpha, const double* ind_lkl, double* marg,
                              int* flags) {
  if (I == 0 || S == 0) return;
  hipLaunchKernelGGL(k_backward_exact_pc, dim3((unsigned)((I + PC_CH - 1) / PC_CH)),
                     dim3(PC_THREADS), pc_pad_bytes((unsigned)((I + PC_CH - 1) / PC_CH)), st, eprob, pos, fw, S, I, indF, alpha, ind_lkl, marg,
                     flags);
}

}  // namespace nghmm

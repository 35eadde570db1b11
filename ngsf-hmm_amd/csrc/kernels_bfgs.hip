// kernels_bfgs.hip -- the indF / alpha M-step's optimizer ON THE DEVICE (fast mode).
//
// The reference runs one blocking findmax_bfgs per individual (EM.cpp:198-201,423-440;
// shared/bfgs.cpp:83-138): a 2-parameter L-BFGS-B whose every objective value is a forward pass.
// Rounds 1-4 of this project batched the objective on the GPU and kept the state machines on the
// host: every lock-step round was kernel -> copy down -> I host machines -> copy up -> kernel,
// ~100 us of idle device per round at 100 individuals, 0.1-0.15 ms of host arithmetic per round at
// 1000 -- a third of configs[1]'s iteration, an eighth of an 8-GPU rank's.  Here the machines
// live in device memory and one small kernel per round (k_bfgs_advance) does everything the host
// did between two rounds:
//
//   * one WAVE per individual, four per workgroup, lane 0 walking: the solver's work arrays (1238
//     doubles for n = 2, m = 10: ws, wy, sy, ss, wt, wn, snd, wa, ...) are staged in LDS --
//     the routines are chains of dependent little loops over them, and LDS latency is a sixth of
//     L2's -- by all 64 lanes of the wave, walked by lane 0, and written back; the matrices
//     (1220 of the doubles) only when a step ends an L-BFGS-B iteration: a line-search step
//     touches 18 doubles;
//   * the round's values (d_lkl[individual * 5 + slot], left there by k_fast_lkl_finish) become
//     objective + finite-difference gradient (bfgs.cpp:22-65), the solver (lbfgsb_core.hpp, the
//     same LbfgsbT<> the host runs) advances until it wants another evaluation or ends
//     (bfgs_problem.hpp: the code BfgsBatch runs on the host), the next points are planned, the
//     individual's group descriptor is written and its index appended to the worklist of the
//     loop-body version its points need (fd_pattern) and to the list of everybody;
//   * the last workgroup to finish publishes (round, active individuals, modes present and their
//     counts) in pinned host memory.  The host polls that word -- no copy, no event, no host
//     arithmetic -- and launches the per-mode objective kernels over exactly the planned groups.
//
// A value that comes back non-finite from a pattern kernel (its points share one scale) is not
// consumed: the individual's same points go to the general kernel in the next round (the host
// path's redo_nonfinite); non-finite from the general kernel is the reference's "invalid Lkl
// found!".
//
// IEEE add / sub / mul / div / sqrt only, contraction off: for the same objective values the
// device takes the host machine's steps bit for bit (tests/test_gpu_devbfgs.py).
#include "fast_dev.hpp"

#pragma clang fp contract(off)

#include "bfgs_problem.hpp"

#include <thread>

namespace nghmm {

namespace {

constexpr int kN = 2, kM = 10;                                   // MVAL, shared/bfgs.h:23
constexpr int kArr = (int)LbfgsbPtrs::doubles(kN, kM);           // 1238 doubles per individual
constexpr int kVec = 7 * kN;                                     // x l u z r d t: the doubles [0, 14)
constexpr int kMat = (int)LbfgsbPtrs::matrices(kN, kM);          // ws .. wa: the doubles [14, 1234)
constexpr int kStage = (kMat + 63) / 64;                         // ... = 20 per lane
static_assert(kVec + kMat + (4 * kN + 1) / 2 == kArr, "vectors | matrices | index arrays");
using DevSolver = LbfgsbT<PtrStore>;
using DevBfgs = FastState::DevBfgs;
constexpr uint32_t kCntAll = kModeSlots, kCntTicket = kModeSlots + 1, kCntStride = kModeSlots + 3;
constexpr uint32_t kTblStats = 4 + 2 * kModeSlots;               // table: 6 x u64 of statistics behind the pairs
static_assert(DevBfgs::kTableWords >= kTblStats + 12, "pairs + statistics");
// statistics published with an M-step's last (empty) plan: 0 points, 1 the reference's forward
// passes, 2 individual-rounds, 3 rounds, 4 rounds repeated by the general kernel, 5 "invalid Lkl
// found!"

struct DevPtrs {
  BfgsProblem* prob;
  DevSolver* solver;
  double* arrays;
  GroupDesc* groups;
  uint32_t* last_mode;
  uint32_t* worklists;
  uint32_t* all;
  uint32_t* counts;
  double* lkl;
  // the round's partial operators [I][C][MAXP][5] and sums of log e0 [I][C]: a handle that holds
  // whole chains finishes its individuals' points here (fast_dev.hpp: lkl_point_product); a site
  // shard's values come combined over the ranks (finish == 0)
  const double* part;
  const double* base_c;
  uint32_t C;
  int finish;
  double *d_F, *d_A;              // the handle's parameters: a finished individual's go there ...
  double *h_F, *h_A;              // ... and, when the M-step ends, all of them to their pinned host mirror
  double *snap_F, *snap_A;        // the parameters the M-step started from (the E-step's, which runs next to it)
  uint32_t* h_table;
  uint32_t I;
  // fd_pattern's view of the data
  double dmax, alpha_small_min;
  uint64_t T;
  int packed;
};

__host__ __device__ constexpr uint32_t slot_mode(uint32_t slot) {
  return slot == 0 ? 0u : (FD_FLAG | (((slot - 1) / 16u) << 9) | ((slot - 1) % 16u));
}
static_assert(mode_slot(slot_mode(37)) == 37 && mode_slot(fd_mode(2, 2, true, true) | FD_OWNEX) < kModeSlots,
              "mode <-> worklist slot");

// the planned points of p as the group descriptor of individual i
__device__ inline void build_group(const BfgsProblem& p, uint32_t i, const DevPtrs& D, GroupDesc& G) {
  G.ind = i;
  G.pad = 0;
  G.pad2 = 0;
  uint32_t np = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    G.F[k] = 0;
    G.A[k] = 0;
    G.out_idx[k] = 0;
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    if (!p.slot_used[k] || p.slot_nonfinite[k]) continue;
    G.F[np] = p.pt[k][0];
    G.A[np] = p.pt[k][1];
    G.out_idx[np] = i * 5 + (uint32_t)k;
    ++np;
  }
  G.np = np;
  G.mode = fd_pattern(G, D.dmax, D.T, D.packed != 0, D.alpha_small_min);
}

// (the number of points the optimizer asked for: G.np before fd_pad fills the pattern up)
__device__ inline uint32_t build_group_padded(const BfgsProblem& p, uint32_t i, const DevPtrs& D, GroupDesc& G) {
  build_group(p, i, D, G);
  const uint32_t asked = G.np;
  fd_pad(G);
  return asked;
}

// One individual per WAVE, kWg waves (individuals) per workgroup: the machines of different
// individuals are at different places of the algorithm -- lanes of one wave would take their
// branches one after the other -- so lane 0 of a wave walks its individual's machine while all 64
// lanes move that individual's work arrays between memory and the wave's part of LDS.  What the
// workgroup's individuals add to the plan goes through ONE atomic per counter and workgroup (a
// thousand waves adding one by one to the same three words would wait for each other longer than
// a machine's step takes).
//
// FIRST: plan P_out from the current parameters (no values yet).  Else: the values of plan
// P_out - 1 into the machines of that plan's individuals, P_out planned.  Plans are numbered
// through the handle's life (slot = P mod kRing, worklists by parity), so every kernel finds its
// counters zeroed by the one two before it.
constexpr uint32_t kRedone = 0xffffffffu;  // last_mode: the round was a site shard's repeat by the general kernel
constexpr int kWg = 4;  // 4 x 10 KB of LDS: three workgroups per CU
struct WaveLds {
  double arr[kArr];
  BfgsProblem p;
  GroupDesc g;
};
static_assert(kWg * sizeof(WaveLds) <= 65536, "one workgroup's LDS");

template <bool FIRST>
__global__ void __launch_bounds__(64 * kWg)
k_bfgs_advance(DevPtrs D, uint32_t P_out, uint32_t n_in, uint32_t round, int F_fixed, int alpha_fixed) {
  __shared__ WaveLds wl[kWg];
  __shared__ uint32_t wg_slot[kWg];   // mode slot + 1 of the wave's individual in the next plan, 0: none
  __shared__ uint32_t wg_pos[kWg], wg_all;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t* cnt_out = D.counts + (P_out % DevBfgs::kRing) * kCntStride;
  if (blockIdx.x == 0) {  // the slot after next is nobody's at the moment
    for (uint32_t k = threadIdx.x; k < kCntStride; k += 64 * kWg)
      D.counts[((P_out + 1) % DevBfgs::kRing) * kCntStride + k] = 0;
  }
  const uint32_t w_idx = blockIdx.x * kWg + wv;
  const bool have = w_idx < n_in;  // (wave-uniform)
  const uint32_t i = !have ? 0u : FIRST ? w_idx : D.all[(uint64_t)((P_out - 1) & 1u) * D.I + w_idx];
  double* lds = wl[wv].arr;
  BfgsProblem& p = wl[wv].p;
  GroupDesc& G = wl[wv].g;
  bool started = false;
  if (lane == 0) wg_slot[wv] = 0;
  if constexpr (!FIRST) {
    if (have && lane == 0) p = D.prob[i];
    __syncthreads();
    started = have && p.started != 0;
    if (started) {  // (else start_bound zeroes them)
      // the machine's vectors and index arrays: 18 doubles.  Its matrices (ws .. wa, 1220
      // doubles) only when a step needs them -- below
      const double* src = D.arrays + (uint64_t)i * kArr;
      if (lane < kVec) lds[lane] = src[lane];
      else if (lane < kVec + kArr - kVec - kMat) lds[kMat + lane] = src[kMat + lane];
    }
    __syncthreads();
  }

  if constexpr (!FIRST) {
    // the round's values of this individual, from the partial operators its objective waves left:
    // what k_fast_lkl_finish does with a workgroup per individual (the same operations in the same
    // order) -- the whole wave takes part, lane 0 keeps the values
    if (have && D.finish) {
      const GroupDesc& Gd = D.groups[i];
      const uint32_t np = Gd.np;
      const double base = base_sum(D.base_c + (uint64_t)i * D.C, D.C, lane);
      const double* part_g = D.part + (uint64_t)i * D.C * MAXP * 5;
      for (uint32_t q = 0; q < np && q < (uint32_t)MAXP; ++q) {
        const Op m = lkl_point_product(part_g, D.C, q, lane);
        if (lane == 0) D.lkl[Gd.out_idx[q]] = lkl_point_value(m, Gd.F[q], base);
      }
    }
  }
  bool keep = false;        // the solver's arrays go back to memory
  bool walk = false;        // (lane 0) the round's values are good: the machine takes its step
  if (have && lane == 0) {
    if constexpr (FIRST) {
      bfgs_problem_begin(p, D.d_F[i], D.d_A[i], F_fixed != 0, alpha_fixed != 0);
      D.snap_F[i] = p.x[0];
      D.snap_A[i] = p.x[1];
      bfgs_plan<DetPow>(p);
      p.n_rounds = 1;
      p.acc_points = build_group_padded(p, i, D, G);
      D.groups[i] = G;
      D.last_mode[i] = G.mode;
      D.prob[i] = p;
      wg_slot[wv] = mode_slot(G.mode) + 1;
    } else {
      double lklv[5] = {0, 0, 0, 0, 0};
      bool bad = false;
#pragma unroll
      for (int k = 0; k < 5; ++k)
        if (p.slot_used[k] && !p.slot_nonfinite[k]) {
          lklv[k] = D.lkl[(uint64_t)i * 5 + k];
          bad = bad || bfgs_nonfinite(lklv[k]);
        }
      // A handle that holds whole chains repeats a pattern kernel's non-finite values with the
      // general kernel, and takes the general kernel's for final.  Site shards: every rank sees the
      // same (combined) values but chose its kernel by its OWN sites' distances (fd_pattern), so
      // "was it a pattern kernel" differs between ranks, and ranks that disagree about a repeat
      // disagree about the number of exchanges.  There the rule is the same for everybody: ONE
      // repeat by the general kernel whatever ran before (kRedone marks it), then final.
      const bool may_repeat = D.finish ? D.last_mode[i] != 0 : D.last_mode[i] != kRedone;
      if (bad && may_repeat) {
        // a probe left the pattern kernel's shared scale: the same points by the general kernel
        D.groups[i].mode = 0;
        D.last_mode[i] = D.finish ? 0u : kRedone;
        wg_slot[wv] = mode_slot(0) + 1;
        ++p.acc_redone;  // (not a round of the optimizer's: n_rounds counts evaluations it asked for)
        D.prob[i] = p;
      } else if (bad) {
        p.acc_invalid = 1;  // EM.cpp:400-410: "invalid Lkl found!"
        p.active = 0;
        D.prob[i] = p;
      } else {
        // (the walk itself is further down: every lane of the wave takes part in its staging)
        walk = true;
      }
    }
  }

  // ---- the machine's walk (bfgs_consume, bfgs_problem.hpp, with the matrices staged lazily) ----
  // A step that continues a line search, or the first step of a machine, touches the vectors
  // only; the matrices are needed once an L-BFGS-B iteration has ended (NEW_X: update, Cholesky
  // factors, Cauchy point, subspace step).  So the wave brings them in -- from memory, or as the
  // zeros a new machine starts from -- when lane 0's solver is about to take such a step, and
  // writes them back only if it did.
  bool mats_here = false;
  if constexpr (!FIRST) {
    DevSolver s;
    uint64_t calls = 0, rc = 0;
    int state = 2;  // 0: call setulb again, 1: wants another round, 2: finished / nothing to walk
    walk = __shfl((int)walk, 0) != 0;
    if (walk) {
      if (lane == 0) {
        double lklv[5];
#pragma unroll
        for (int k = 0; k < 5; ++k)
          lklv[k] = (p.slot_used[k] && !p.slot_nonfinite[k]) ? D.lkl[(uint64_t)i * 5 + k] : 0.0;
        if (started) s = D.solver[i];
        s.st_.bind(lds, kN, kM);
        calls = bfgs_gradient(p, lklv);
        rc = calls;
        if (!p.started) {
          const int nbd[2] = {2, 2};
          s.start_bound(kN, kM, p.x, p.lb, p.ub, nbd, 1.0e6, 1.0e-3, false);  // FACTR, PGTOL: bfgs.h:24-25
          p.started = 1;
          p.big_valid = 0;
        }
      }
      state = 0;
      while (state == 0) {
        int need = 0;  // 1: load the matrices, 2: they are zeros
        if (lane == 0 && !mats_here && s.phase_ == LbfgsbPhase::NewX) need = p.big_valid ? 1 : 2;
        need = __shfl(need, 0);
        if (need) {
          const double* src = D.arrays + (uint64_t)i * kArr + kVec;
          double v[kStage];
#pragma unroll
          for (int t = 0; t < kStage; ++t) {
            const int j = lane + 64 * t;
            v[t] = (need == 1 && j < kMat) ? src[j] : 0.0;
          }
#pragma unroll
          for (int t = 0; t < kStage; ++t) {
            const int j = lane + 64 * t;
            if (j < kMat) lds[kVec + j] = v[t];
          }
          mats_here = true;
          __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) state = bfgs_step(p, s, calls, rc);
        state = __shfl(state, 0);
      }
    }
    if (walk && lane == 0) {
      {
        const bool again = state == 1;
        p.acc_ref_calls += (uint32_t)rc;
        if (again) {
          bfgs_plan<DetPow>(p);
          ++p.n_rounds;
          p.acc_points += build_group_padded(p, i, D, G);
          D.groups[i] = G;
          D.last_mode[i] = G.mode;
          wg_slot[wv] = mode_slot(G.mode) + 1;
          D.solver[i] = s;
          keep = true;
        } else {
          D.d_F[i] = p.x[0];
          D.d_A[i] = p.x[1];
        }
        if (keep && mats_here) p.big_valid = 1;
        D.prob[i] = p;
      }
    }
    // vectors and index arrays back if the machine goes on; the matrices if this step wrote them
    __builtin_amdgcn_wave_barrier();
    if (__shfl((int)keep, 0)) {
      double* dst = D.arrays + (uint64_t)i * kArr;
      if (lane < kVec) dst[lane] = lds[lane];
      else if (lane < kVec + kArr - kVec - kMat) dst[kMat + lane] = lds[kMat + lane];
      if (mats_here) {
#pragma unroll 4
        for (int j = lane; j < kMat; j += 64) dst[kVec + j] = lds[kVec + j];
      }
    }
  }

  // the workgroup's individuals into plan P_out: one atomic per mode present among them (its
  // worklist), one for the list of everybody, one ticket
  __syncthreads();
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    uint32_t n_new = 0;
    for (int a = 0; a < kWg; ++a) {
      const uint32_t sl = wg_slot[a];
      if (sl == 0) continue;
      ++n_new;
      bool first = true;   // first wave of the workgroup with this mode: it reserves for all of them
      for (int b2 = 0; b2 < a; ++b2) first = first && wg_slot[b2] != sl;
      if (!first) continue;
      uint32_t n_same = 0;
      for (int b2 = a; b2 < kWg; ++b2) n_same += wg_slot[b2] == sl ? 1u : 0u;
      uint32_t pos = atomicAdd(&cnt_out[sl - 1], n_same);
      for (int b2 = a; b2 < kWg; ++b2)
        if (wg_slot[b2] == sl) wg_pos[b2] = pos++;
    }
    wg_all = n_new ? atomicAdd(&cnt_out[kCntAll], n_new) : 0u;
  }
  __syncthreads();
  if (lane == 0 && wg_slot[wv] != 0) {
    const uint32_t par = P_out & 1u, sl = wg_slot[wv] - 1;
    D.worklists[((uint64_t)par * kModeSlots + sl) * D.I + wg_pos[wv]] = i;
    uint32_t before = 0;
    for (int a = 0; a < wv; ++a) before += wg_slot[a] != 0 ? 1u : 0u;
    D.all[(uint64_t)par * D.I + wg_all + before] = i;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) is_last = atomicAdd(&cnt_out[kCntTicket], 1u) == gridDim.x - 1;
  __syncthreads();

  // The last workgroup publishes plan P_out to the host: the modes present and their counts
  // (wave 0 reads the counters, a ballot compacts them) -- and, when the plan is empty, i.e.
  // the M-step is over, every individual's parameters and the accounting summed over them.
  if (is_last && wv == 0) {
    __threadfence();
    // ONE lane then writes the whole entry to host memory and, behind a system-scope fence, the
    // plan's number
    __shared__ uint32_t pub_mode[kModeSlots], pub_count[kModeSlots];
    uint32_t* t = D.h_table + (P_out % DevBfgs::kRing) * DevBfgs::kTableWords;
    const uint32_t n_all = __hip_atomic_load(&cnt_out[kCntAll], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t n_modes = 0;
    for (uint32_t s0 = 0; s0 < kModeSlots; s0 += 64) {
      const uint32_t sl = s0 + lane;
      const uint32_t c = sl < kModeSlots
                             ? __hip_atomic_load(&cnt_out[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : 0u;
      const uint64_t have_m = __ballot(c != 0);
      if (c != 0) {
        const uint32_t k = n_modes + (uint32_t)__popcll(have_m & ((1ull << lane) - 1));
        pub_mode[k] = slot_mode(sl);
        pub_count[k] = c;
      }
      n_modes += (uint32_t)__popcll(have_m);
    }
    unsigned long long pts = 0, calls = 0, indr = 0, redone = 0, invalid = 0;
    uint32_t rmax = 0;
    if (n_all == 0) {  // the M-step is over: every individual's parameters to the host, the accounting
      for (uint32_t k = lane; k < D.I; k += 64) {
        D.h_F[k] = D.d_F[k];
        D.h_A[k] = D.d_A[k];
        const BfgsProblem& q = D.prob[k];
        pts += q.acc_points;
        calls += q.acc_ref_calls;
        indr += q.n_rounds;
        redone += q.acc_redone;
        invalid += q.acc_invalid;
        rmax = q.n_rounds > rmax ? q.n_rounds : rmax;
      }
      for (int off = 32; off > 0; off >>= 1) {
        pts += __shfl_down(pts, off);
        calls += __shfl_down(calls, off);
        indr += __shfl_down(indr, off);
        redone += __shfl_down(redone, off);
        invalid += __shfl_down(invalid, off);
        const uint32_t o = __shfl_down(rmax, off);
        rmax = o > rmax ? o : rmax;
      }
    }
    __threadfence_system();  // (the parameters above: a wave's stores, complete before its lane 0 goes on)
    if (lane == 0) {
      t[3] = P_out;  // (the entry's second stamp: the host checks both)
      for (uint32_t k = 0; k < n_modes; ++k) {
        t[4 + 2 * k] = pub_mode[k];
        t[5 + 2 * k] = pub_count[k];
      }
      const unsigned long long v[6] = {pts, calls, indr, rmax, redone, invalid};
      for (int k = 0; k < 6; ++k) {
        t[kTblStats + 2 * k] = (uint32_t)v[k];
        t[kTblStats + 2 * k + 1] = (uint32_t)(v[k] >> 32);
      }
      t[1] = n_all;
      t[2] = n_modes;
      __threadfence_system();
      __hip_atomic_store(&t[0], P_out, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// the end of a fused iteration (FastState::DevBfgs::h_epi): see dbfgs_epilogue
__global__ void __launch_bounds__(256)
k_iter_epilogue(int* __restrict__ d_flags, uint32_t n_flags, const double* __restrict__ d_lkl, uint32_t I,
                double* __restrict__ h_lkl, uint32_t* __restrict__ h_epi, uint32_t seq) {
  if (d_lkl)
    for (uint32_t k = threadIdx.x; k < I; k += blockDim.x) h_lkl[k] = d_lkl[k];
  if (threadIdx.x < n_flags) {
    h_epi[threadIdx.x] = (uint32_t)d_flags[threadIdx.x];
    d_flags[threadIdx.x] = 0;  // the next iteration's background work starts from clear flags
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&h_epi[kEpiFlags], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

DevPtrs dev_ptrs(const FastState& fs, double* d_F, double* d_A) {
  const DevBfgs& d = fs.dev;
  DevPtrs D;
  D.prob = reinterpret_cast<BfgsProblem*>(d.prob);
  D.solver = reinterpret_cast<DevSolver*>(d.solver);
  D.arrays = d.arrays;
  D.groups = reinterpret_cast<GroupDesc*>(d.groups);
  D.last_mode = d.last_mode;
  D.worklists = d.worklists;
  D.all = d.all;
  D.counts = d.counts;
  D.lkl = d.lkl;
  D.part = d.part;
  D.base_c = fs.base_c;
  D.C = fs.C;
  D.finish = fs.shard.world <= 1 ? 1 : 0;
  D.d_F = d_F;
  D.d_A = d_A;
  D.h_F = d.h_F;
  D.h_A = d.h_A;
  D.snap_F = d.snap_F + (uint64_t)d.snap_set * fs.I;
  D.snap_A = d.snap_A + (uint64_t)d.snap_set * fs.I;
  D.h_table = const_cast<uint32_t*>(d.h_table);
  D.I = (uint32_t)fs.I;
  D.dmax = fs.dmax_finite;
  D.alpha_small_min = fs.alpha_small_min;
  D.T = fs.T;
  D.packed = fs.packed ? 1 : 0;
  return D;
}

template <typename T>
bool dmalloc(T** p, size_t n) {
  return hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)) == hipSuccess;
}

}  // namespace

bool dbfgs_available(const FastState& fs) {
  if (fs.I == 0 || fs.I > 0x0fffffffu) return false;
  // the largest step of an alpha probe (alpha <= 10, EM.cpp:427) inside exp_small's range on
  // every finite distance: else every group is a general one
  return DetPow::eh(10.0) * fs.dmax_finite <= 1e-3;
}

bool dbfgs_reserve(FastState& fs) {
  DevBfgs& d = fs.dev;
  if (d.cap_I == fs.I) return true;
  dbfgs_destroy(fs);
  const size_t I = fs.I;
  BfgsProblem* prob = nullptr;
  DevSolver* solver = nullptr;
  GroupDesc* groups = nullptr;
  bool ok = dmalloc(&prob, I) && dmalloc(&solver, I) && dmalloc(&d.arrays, I * kArr) &&
            dmalloc(&groups, I) && dmalloc(&d.last_mode, I) &&
            dmalloc(&d.worklists, (size_t)2 * kModeSlots * I) && dmalloc(&d.all, 2 * I) &&
            dmalloc(&d.counts, (size_t)DevBfgs::kRing * kCntStride) &&
            dmalloc(&d.lkl, 5 * I) && dmalloc(&d.part, I * fs.C * MAXP * 5) && dmalloc(&d.snap_F, 2 * I) &&
            dmalloc(&d.snap_A, 2 * I);
  d.prob = prob;
  d.solver = solver;
  d.groups = groups;
  void *t = nullptr, *hf = nullptr, *ha = nullptr, *el = nullptr, *ep = nullptr;
  const unsigned flags = hipHostMallocCoherent | hipHostMallocMapped;
  ok = ok && hipHostMalloc(&t, DevBfgs::kRing * DevBfgs::kTableWords * sizeof(uint32_t), flags) == hipSuccess &&
       hipHostMalloc(&hf, I * sizeof(double), flags) == hipSuccess &&
       hipHostMalloc(&ha, I * sizeof(double), flags) == hipSuccess &&
       hipHostMalloc(&el, I * sizeof(double), flags) == hipSuccess &&
       hipHostMalloc(&ep, (kEpiFlags + 1) * sizeof(uint32_t), flags) == hipSuccess;
  d.h_table = static_cast<volatile uint32_t*>(t);
  d.h_F = static_cast<double*>(hf);
  d.h_A = static_cast<double*>(ha);
  d.h_epi_lkl = static_cast<double*>(el);
  d.h_epi = static_cast<volatile uint32_t*>(ep);
  if (ep) std::memset(ep, 0, (kEpiFlags + 1) * sizeof(uint32_t));
  d.epi_seq = 0;
  ok = ok && hipMemset(d.counts, 0, (size_t)DevBfgs::kRing * kCntStride * sizeof(uint32_t)) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    dbfgs_destroy(fs);
    return false;
  }
  std::memset(t, 0, DevBfgs::kRing * DevBfgs::kTableWords * sizeof(uint32_t));
  // (hipMemset of device memory need not have finished when it returns, and the handle's stream
  // does not wait for the null stream)
  if (hipDeviceSynchronize() != hipSuccess) {
    dbfgs_destroy(fs);
    return false;
  }
  d.seq_base = 0;
  d.mstep_no = 0;
  d.cap_I = I;
  return true;
}

void dbfgs_destroy(FastState& fs) {
  DevBfgs& d = fs.dev;
  void* dev[] = {d.prob, d.solver, d.arrays, d.groups, d.last_mode, d.worklists, d.all, d.counts,
                 d.lkl, d.part, d.snap_F, d.snap_A};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  if (d.h_table) (void)hipHostFree(const_cast<uint32_t*>(d.h_table));
  if (d.h_F) (void)hipHostFree(d.h_F);
  if (d.h_A) (void)hipHostFree(d.h_A);
  if (d.h_epi_lkl) (void)hipHostFree(d.h_epi_lkl);
  if (d.h_epi) (void)hipHostFree(const_cast<uint32_t*>(d.h_epi));
  d = DevBfgs();
}

void dbfgs_invalidate(FastState& fs) {
  DevBfgs& d = fs.dev;
  if (!d.preplanned) return;
  // a plan has been published and its counters are set: the next dbfgs_begin starts over as after
  // an M-step that did not reach its end
  d.preplanned = false;
  d.clean = false;
}

bool dbfgs_epilogue(FastState& fs, hipStream_t st, int* d_flags, uint32_t n_flags, const double* d_lkl) {
  DevBfgs& d = fs.dev;
  if (!d.h_epi || n_flags > kEpiFlags) return false;
  ++d.epi_seq;
  if (d.epi_seq == 0) ++d.epi_seq;
  hipLaunchKernelGGL(k_iter_epilogue, dim3(1), dim3(256), 0, st, d_flags, n_flags, d_lkl, (uint32_t)fs.I,
                     d.h_epi_lkl, const_cast<uint32_t*>(d.h_epi), d.epi_seq);
  return hipGetLastError() == hipSuccess;
}

bool dbfgs_wait_epilogue(FastState& fs, hipStream_t st, int* flags_out, uint32_t n_flags, bool yield) {
  DevBfgs& d = fs.dev;
  const volatile uint32_t* w = d.h_epi + kEpiFlags;
  uint32_t spins = 0;
  bool drained = false;
  for (;;) {
    if (__atomic_load_n(const_cast<const uint32_t*>(w), __ATOMIC_ACQUIRE) == d.epi_seq) break;
    if (yield && (spins & 0x3fu) == 0x3fu) std::this_thread::yield();
    if ((++spins & 0x3fffu) == 0) {
      if (drained) return false;
      const hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) drained = true;
      else if (q != hipErrorNotReady) return false;
      (void)hipGetLastError();
    }
  }
  for (uint32_t k = 0; k < n_flags; ++k) flags_out[k] = (int)d.h_epi[k];
  return true;
}

bool dbfgs_begin(FastState& fs, hipStream_t st, double* d_indF, double* d_alpha, bool F_fixed,
                 bool alpha_fixed) {
  DevBfgs& d = fs.dev;
  if (d.cap_I != fs.I || fs.I == 0) return false;
  const uint32_t n = (uint32_t)fs.I;
  if (d.preplanned) {
    // round 1 of this M-step was planned when the last one ended, from the parameters the handle
    // still holds: its plan is in the table (or on its way)
    if (d.pre_F_fixed == F_fixed && d.pre_alpha_fixed == alpha_fixed && d.d_F == d_indF && d.d_A == d_alpha) {
      d.preplanned = false;
      return true;
    }
    dbfgs_invalidate(fs);
  }
  if (!d.clean) {
    // the M-step before this one did not reach its end (an error on its way): plans of it may
    // have been published and counters left behind.  Clear the counters (on the stream: the
    // kernel below comes after) and number on from well past anything that can be in the table.
    if (hipMemsetAsync(d.counts, 0, (size_t)DevBfgs::kRing * kCntStride * sizeof(uint32_t), st) != hipSuccess)
      return false;
    d.seq_base += 4096;
  }
  d.clean = false;
  d.d_F = d_indF;
  d.d_A = d_alpha;
  d.snap_set ^= 1u;
  hipLaunchKernelGGL(k_bfgs_advance<true>, dim3((n + kWg - 1) / kWg), dim3(64 * kWg), 0, st,
                     dev_ptrs(fs, d_indF, d_alpha),
                     d.seq_base + 1, n, 1u, F_fixed ? 1 : 0, alpha_fixed ? 1 : 0);
  return hipGetLastError() == hipSuccess;
}

const double* dbfgs_start_F(const FastState& fs) { return fs.dev.snap_F + (uint64_t)fs.dev.snap_set * fs.I; }
const double* dbfgs_start_A(const FastState& fs) { return fs.dev.snap_A + (uint64_t)fs.dev.snap_set * fs.I; }

bool dbfgs_advance(FastState& fs, hipStream_t st, uint32_t round, uint32_t n_in) {
  DevBfgs& d = fs.dev;
  if (n_in == 0) return false;
  hipLaunchKernelGGL(k_bfgs_advance<false>, dim3((n_in + kWg - 1) / kWg), dim3(64 * kWg), 0, st,
                     dev_ptrs(fs, d.d_F, d.d_A),
                     d.seq_base + round + 1, n_in, round + 1, 0, 0);
  return hipGetLastError() == hipSuccess;
}

bool dbfgs_wait_plan(FastState& fs, hipStream_t st, uint32_t round, uint32_t* n_active,
                     std::vector<FastState::ModeRange>* ranges, bool yield) {
  DevBfgs& d = fs.dev;
  const uint32_t P = d.seq_base + round;
  const volatile uint32_t* t = d.h_table + (P % DevBfgs::kRing) * DevBfgs::kTableWords;
  // the planning kernel is on the stream: its last workgroup stores the plan's number with
  // system scope.  Should the stream run dry without it (a failed launch), give up.
  uint32_t spins = 0;
  bool drained = false;
  for (;;) {
    if (__atomic_load_n(const_cast<const uint32_t*>(t), __ATOMIC_ACQUIRE) == P) break;
    if (yield && (spins & 0x3fu) == 0x3fu) std::this_thread::yield();
    if ((++spins & 0x3fffu) == 0) {
      if (drained) return false;
      const hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) drained = true;       // one more look at the word, then give up
      else if (q != hipErrorNotReady) return false;
      (void)hipGetLastError();
    }
  }
  if (t[3] != P) return false;  // (one lane writes the entry, its number last: both stamps agree)
  *n_active = t[1];
  const uint32_t nm = t[2];
  ranges->clear();
  uint32_t begin = 0;
  for (uint32_t k = 0; k < nm && k < kModeSlots; ++k) {
    ranges->push_back({t[4 + 2 * k], begin, t[5 + 2 * k]});
    begin += t[5 + 2 * k];
  }
  for (int k = 0; k < 6; ++k)
    d.stats_host[k] = (unsigned long long)t[kTblStats + 2 * k] | ((unsigned long long)t[kTblStats + 2 * k + 1] << 32);
  return begin == *n_active;
}

bool dbfgs_launch_round(FastState& fs, hipStream_t st, uint32_t round, uint32_t n_active,
                        const std::vector<FastState::ModeRange>& ranges, bool emit_estep) {
  DevBfgs& d = fs.dev;
  const uint32_t par = (d.seq_base + round) & 1u;
  return fast_lkl_launch_planned(fs, st, d.groups, ranges, n_active,
                                 d.worklists + (uint64_t)par * kModeSlots * fs.I, d.all + (uint64_t)par * fs.I,
                                 d.part, d.lkl, emit_estep);
}

// the M-step is over (the plan of round `last_round` came out empty): the parameters of every
// individual are in d.h_F / d.h_A (and on the device), the statistics in d.stats_host
void dbfgs_end(FastState& fs, uint32_t last_round) {
  DevBfgs& d = fs.dev;
  d.seq_base += last_round;
  ++d.mstep_no;
  d.clean = true;
}

// the next M-step's round 1 planned now (see FastState::DevBfgs::preplanned); nothing waits
bool dbfgs_preplan(FastState& fs, hipStream_t st, bool F_fixed, bool alpha_fixed) {
  DevBfgs& d = fs.dev;
  if (!d.clean || d.preplanned) return false;
  if (!dbfgs_begin(fs, st, d.d_F, d.d_A, F_fixed, alpha_fixed)) return false;
  d.preplanned = true;
  d.pre_F_fixed = F_fixed;
  d.pre_alpha_fixed = alpha_fixed;
  return true;
}

}  // namespace nghmm

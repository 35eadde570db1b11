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
//   * one LANE per individual, six individuals per workgroup: the solver's work arrays (1238
//     doubles for n = 2, m = 10: ws, wy, sy, ss, wt, wn, snd, wa, ...) are staged in LDS --
//     the routines are chains of dependent little loops over them, and LDS latency is a sixth of
//     L2's -- by all 64 lanes of the workgroup, walked by the individual's lane, and written back;
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
constexpr int kArrPad = kArr + 3;                                // LDS stride: no two lanes on a bank pair
constexpr int kPerWg = 6;                                        // individuals per workgroup (59.6 KB LDS)
using DevSolver = LbfgsbT<PtrStore>;
using DevBfgs = FastState::DevBfgs;
constexpr uint32_t kCntAll = kModeSlots, kCntTicket = kModeSlots + 1, kCntStride = kModeSlots + 3;
static_assert(DevBfgs::kTableWords >= 4 + 2 * kModeSlots, "a pair per mode");
static_assert(kPerWg * kArrPad * sizeof(double) <= 65536, "one workgroup's LDS");

struct DevPtrs {
  BfgsProblem* prob;
  DevSolver* solver;
  double* arrays;
  GroupDesc* groups;
  uint32_t* last_mode;
  uint32_t* worklists;
  uint32_t* all;
  uint32_t* counts;
  unsigned long long* stats;
  const double* lkl;
  double *new_F, *new_A;
  int* flags;
  uint32_t* h_table;
  uint32_t I;
  uint32_t seq_base;
  // fd_pattern's view of the data
  double dmax;
  uint64_t T;
  int packed, allow_xdeg2;
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
  for (int k = 0; k < 5; ++k) {
    G.F[k] = 0;
    G.A[k] = 0;
    G.out_idx[k] = 0;
  }
  for (int k = 0; k < 5; ++k) {
    if (!p.slot_used[k] || p.slot_nonfinite[k]) continue;
    G.F[np] = p.pt[k][0];
    G.A[np] = p.pt[k][1];
    G.out_idx[np] = i * 5 + (uint32_t)k;
    ++np;
  }
  G.np = np;
  G.mode = fd_pattern(G, D.dmax, D.T, D.packed != 0, D.allow_xdeg2 != 0);
}

// individual i into the plan of round p_next
__device__ inline void enlist(const DevPtrs& D, uint32_t p_next, uint32_t i, uint32_t mode) {
  uint32_t* cnt = D.counts + (p_next % DevBfgs::kRing) * kCntStride;
  const uint32_t par = p_next & 1u, slot = mode_slot(mode);
  const uint32_t pos = atomicAdd(&cnt[slot], 1u);
  D.worklists[((uint64_t)par * kModeSlots + slot) * D.I + pos] = i;
  const uint32_t pa = atomicAdd(&cnt[kCntAll], 1u);
  D.all[(uint64_t)par * D.I + pa] = i;
}

// FIRST: plan round 1 from the current parameters (no values yet); else: the values of round
// `round` into the machines of the n_in individuals of that round, round + 1 planned.
template <bool FIRST>
__global__ void __launch_bounds__(64)
k_bfgs_advance(DevPtrs D, uint32_t round, uint32_t n_in, const double* __restrict__ indF,
               const double* __restrict__ alpha, int F_fixed, int alpha_fixed) {
  __shared__ double lds[kPerWg * kArrPad];
  const int lane = threadIdx.x;
  const uint32_t p_next = round + 1;
  uint32_t* cnt_next = D.counts + (p_next % DevBfgs::kRing) * kCntStride;
  if (blockIdx.x == 0)  // the slot after next is nobody's at the moment
    for (uint32_t k = lane; k < kCntStride; k += 64)
      D.counts[((p_next + 1) % DevBfgs::kRing) * kCntStride + k] = 0;

  const uint32_t base = blockIdx.x * kPerWg;
  uint32_t my_i = ~0u;
  if (lane < kPerWg && base + lane < n_in)
    my_i = FIRST ? base + lane : D.all[(uint64_t)(round & 1u) * D.I + base + lane];

  if constexpr (!FIRST) {  // the started solvers' work arrays into LDS, by everybody
    for (int k = 0; k < kPerWg; ++k) {
      const uint32_t ik = __shfl(my_i, k);
      if (ik == ~0u) continue;
      if (!D.prob[ik].started) continue;  // (start_bound zeroes its block)
      const double* src = D.arrays + (uint64_t)ik * kArr;
      double* dst = lds + k * kArrPad;
      for (int j = lane; j < kArr; j += 64) dst[j] = src[j];
    }
    __syncthreads();
  }

  bool keep = false;  // the solver's arrays go back to memory
  if (my_i != ~0u) {
    const uint32_t i = my_i;
    BfgsProblem p;
    if constexpr (FIRST) {
      bfgs_problem_begin(p, indF[i], alpha[i], F_fixed != 0, alpha_fixed != 0);
      D.new_F[i] = p.x[0];
      D.new_A[i] = p.x[1];
      bfgs_plan<DetPow>(p);
      p.n_rounds = 1;
      GroupDesc G;
      build_group(p, i, D, G);
      D.groups[i] = G;
      D.last_mode[i] = G.mode;
      D.prob[i] = p;
      enlist(D, p_next, i, G.mode);
      atomicAdd(&D.stats[0], (unsigned long long)G.np);
      atomicAdd(&D.stats[2], 1ull);
      atomicMax(&D.stats[3], 1ull);
    } else {
      p = D.prob[i];
      double lklv[5] = {0, 0, 0, 0, 0};
      bool bad = false;
      for (int k = 0; k < 5; ++k)
        if (p.slot_used[k] && !p.slot_nonfinite[k]) {
          lklv[k] = D.lkl[(uint64_t)i * 5 + k];
          bad = bad || bfgs_nonfinite(lklv[k]);
        }
      if (bad && D.last_mode[i] != 0) {
        // a probe left the pattern kernel's shared scale: the same points by the general kernel
        D.groups[i].mode = 0;
        D.last_mode[i] = 0;
        enlist(D, p_next, i, 0);
        atomicAdd(&D.stats[4], 1ull);
      } else if (bad) {
        D.flags[FLAG_INVALID_LKL] = 1;  // EM.cpp:400-410: "invalid Lkl found!"
        p.active = 0;
        D.prob[i] = p;
      } else {
        DevSolver s;
        if (p.started) s = D.solver[i];
        s.st_.bind(lds + lane * kArrPad, kN, kM);
        unsigned long long ref_calls = 0;
        uint64_t rc = 0;
        const bool again = bfgs_consume(p, s, lklv, rc, [&](BfgsProblem& q) {
          const int nbd[2] = {2, 2};
          s.start_bound(kN, kM, q.x, q.lb, q.ub, nbd, 1.0e6, 1.0e-3);  // FACTR, PGTOL: bfgs.h:24-25
        });
        ref_calls = rc;
        atomicAdd(&D.stats[1], ref_calls);
        if (again) {
          bfgs_plan<DetPow>(p);
          ++p.n_rounds;
          GroupDesc G;
          build_group(p, i, D, G);
          D.groups[i] = G;
          D.last_mode[i] = G.mode;
          enlist(D, p_next, i, G.mode);
          atomicAdd(&D.stats[0], (unsigned long long)G.np);
          atomicAdd(&D.stats[2], 1ull);
          atomicMax(&D.stats[3], (unsigned long long)p.n_rounds);
          D.solver[i] = s;
          keep = true;
        } else {
          D.new_F[i] = p.x[0];
          D.new_A[i] = p.x[1];
        }
        D.prob[i] = p;
      }
    }
  }

  if constexpr (!FIRST) {
    __syncthreads();
    for (int k = 0; k < kPerWg; ++k) {
      const uint32_t ik = __shfl(my_i, k);
      const int kk = __shfl((int)keep, k);
      if (ik == ~0u || !kk) continue;
      double* dst = D.arrays + (uint64_t)ik * kArr;
      const double* src = lds + k * kArrPad;
      for (int j = lane; j < kArr; j += 64) dst[j] = src[j];
    }
  }

  // the last workgroup publishes the plan of round p_next to the host
  __threadfence();
  __shared__ int is_last;
  if (lane == 0) is_last = atomicAdd(&cnt_next[kCntTicket], 1u) == gridDim.x - 1;
  __syncthreads();
  if (is_last && lane == 0) {
    __threadfence();
    uint32_t* t = D.h_table + (p_next % DevBfgs::kRing) * DevBfgs::kTableWords;
    const uint32_t n_all = __hip_atomic_load(&cnt_next[kCntAll], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t n_modes = 0;
    for (uint32_t sl = 0; sl < kModeSlots; ++sl) {
      const uint32_t c = __hip_atomic_load(&cnt_next[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (c == 0) continue;
      t[4 + 2 * n_modes] = slot_mode(sl);
      t[5 + 2 * n_modes] = c;
      ++n_modes;
    }
    t[1] = n_all;
    t[2] = n_modes;
    t[3] = 0;
    __threadfence_system();
    __hip_atomic_store(&t[0], D.seq_base + p_next, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// nothing left to do, but the host waits for a plan: an M-step whose round came out empty
// never happens (advance is launched with n_in >= 1); begin with I = 0 is refused on the host.

DevPtrs dev_ptrs(const FastState& fs) {
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
  D.stats = d.stats;
  D.lkl = d.lkl;
  D.new_F = d.new_F;
  D.new_A = d.new_A;
  D.flags = d.flags;
  D.h_table = const_cast<uint32_t*>(d.h_table);
  D.I = (uint32_t)fs.I;
  D.seq_base = d.seq_base;
  D.dmax = fs.dmax_finite;
  D.T = fs.T;
  D.packed = fs.packed ? 1 : 0;
  D.allow_xdeg2 = fs.sw.no_xdeg2 ? 0 : 1;
  return D;
}

template <typename T>
bool dmalloc(T** p, size_t n) {
  return hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)) == hipSuccess;
}

}  // namespace

bool dbfgs_available(const FastState& fs) {
  if (fs.shard.world > 1 || fs.I == 0 || fs.I > 0x0fffffffu) return false;
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
            dmalloc(&d.counts, (size_t)DevBfgs::kRing * kCntStride) && dmalloc(&d.stats, (size_t)8) &&
            dmalloc(&d.lkl, 5 * I) && dmalloc(&d.part, I * fs.C * MAXP * 5) && dmalloc(&d.new_F, I) &&
            dmalloc(&d.new_A, I) && dmalloc(&d.flags, (size_t)NFLAGS);
  d.prob = prob;
  d.solver = solver;
  d.groups = groups;
  void *t = nullptr, *hs = nullptr, *hf = nullptr;
  ok = ok &&
       hipHostMalloc(&t, DevBfgs::kRing * DevBfgs::kTableWords * sizeof(uint32_t),
                     hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess &&
       hipHostMalloc(&hs, 8 * sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess &&
       hipHostMalloc(&hf, NFLAGS * sizeof(int), hipHostMallocDefault) == hipSuccess;
  d.h_table = static_cast<volatile uint32_t*>(t);
  d.h_stats = static_cast<unsigned long long*>(hs);
  d.h_flags = static_cast<int*>(hf);
  if (!ok) {
    (void)hipGetLastError();
    dbfgs_destroy(fs);
    return false;
  }
  std::memset(t, 0, DevBfgs::kRing * DevBfgs::kTableWords * sizeof(uint32_t));
  d.seq_base = 0;
  d.cap_I = I;
  return true;
}

void dbfgs_destroy(FastState& fs) {
  DevBfgs& d = fs.dev;
  void* dev[] = {d.prob, d.solver, d.arrays, d.groups, d.last_mode, d.worklists, d.all, d.counts,
                 d.stats, d.lkl, d.part, d.new_F, d.new_A, d.flags};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  if (d.h_table) (void)hipHostFree(const_cast<uint32_t*>(d.h_table));
  if (d.h_stats) (void)hipHostFree(d.h_stats);
  if (d.h_flags) (void)hipHostFree(d.h_flags);
  d = DevBfgs();
}

bool dbfgs_begin(FastState& fs, hipStream_t st, const double* d_indF, const double* d_alpha,
                 bool F_fixed, bool alpha_fixed) {
  DevBfgs& d = fs.dev;
  if (d.cap_I != fs.I || fs.I == 0) return false;
  if (hipMemsetAsync(d.counts, 0, (size_t)DevBfgs::kRing * kCntStride * sizeof(uint32_t), st) != hipSuccess ||
      hipMemsetAsync(d.stats, 0, 8 * sizeof(unsigned long long), st) != hipSuccess ||
      hipMemsetAsync(d.flags, 0, NFLAGS * sizeof(int), st) != hipSuccess)
    return false;
  const uint32_t n = (uint32_t)fs.I;
  hipLaunchKernelGGL(k_bfgs_advance<true>, dim3((n + kPerWg - 1) / kPerWg), dim3(64), 0, st, dev_ptrs(fs),
                     0u, n, d_indF, d_alpha, F_fixed ? 1 : 0, alpha_fixed ? 1 : 0);
  return hipGetLastError() == hipSuccess;
}

bool dbfgs_advance(FastState& fs, hipStream_t st, uint32_t round, uint32_t n_in) {
  if (n_in == 0) return false;
  hipLaunchKernelGGL(k_bfgs_advance<false>, dim3((n_in + kPerWg - 1) / kPerWg), dim3(64), 0, st,
                     dev_ptrs(fs), round, n_in, (const double*)nullptr, (const double*)nullptr, 0, 0);
  return hipGetLastError() == hipSuccess;
}

bool dbfgs_wait_plan(FastState& fs, hipStream_t st, uint32_t round, uint32_t* n_active,
                     std::vector<FastState::ModeRange>* ranges, bool yield) {
  DevBfgs& d = fs.dev;
  const volatile uint32_t* t = d.h_table + (round % DevBfgs::kRing) * DevBfgs::kTableWords;
  const uint32_t want = d.seq_base + round;
  // the planning kernel is on the stream: its last workgroup stores the sequence number with
  // system scope.  Should the stream run dry without it (a failed launch), give up.
  uint32_t spins = 0;
  bool drained = false;
  for (;;) {
    if (__atomic_load_n(const_cast<const uint32_t*>(t), __ATOMIC_ACQUIRE) == want) break;
    if (yield && (spins & 0x3fu) == 0x3fu) std::this_thread::yield();
    if ((++spins & 0x3fffu) == 0) {
      if (drained) return false;
      const hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) drained = true;       // one more look at the word, then give up
      else if (q != hipErrorNotReady) return false;
      (void)hipGetLastError();
    }
  }
  *n_active = t[1];
  const uint32_t nm = t[2];
  ranges->clear();
  uint32_t begin = 0;
  for (uint32_t k = 0; k < nm && k < kModeSlots; ++k) {
    ranges->push_back({t[4 + 2 * k], begin, t[5 + 2 * k]});
    begin += t[5 + 2 * k];
  }
  return begin == *n_active;
}

bool dbfgs_launch_round(FastState& fs, hipStream_t st, uint32_t round, uint32_t n_active,
                        const std::vector<FastState::ModeRange>& ranges, bool emit_estep) {
  DevBfgs& d = fs.dev;
  const uint32_t par = round & 1u;
  return fast_lkl_launch_planned(fs, st, d.groups, ranges, n_active,
                                 d.worklists + (uint64_t)par * kModeSlots * fs.I, d.all + (uint64_t)par * fs.I,
                                 d.part, d.lkl, emit_estep);
}

bool dbfgs_end(FastState& fs, hipStream_t st, double* d_indF, double* d_alpha, uint32_t rounds_used) {
  DevBfgs& d = fs.dev;
  d.seq_base += rounds_used + 1;
  return hipMemcpyAsync(d_indF, d.new_F, fs.I * sizeof(double), hipMemcpyDeviceToDevice, st) == hipSuccess &&
         hipMemcpyAsync(d_alpha, d.new_A, fs.I * sizeof(double), hipMemcpyDeviceToDevice, st) == hipSuccess &&
         hipMemcpyAsync(d.h_stats, d.stats, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st) ==
             hipSuccess &&
         hipMemcpyAsync(d.h_flags, d.flags, NFLAGS * sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess;
}

}  // namespace nghmm

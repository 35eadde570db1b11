// bfgs_batch.cpp -- see bfgs_batch.hpp.  Compile with -ffp-contract=off.
#include "bfgs_batch.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>

#include <omp.h>

#include <chrono>

namespace nghmm {

namespace {

// Host threads for the per-individual state machines (independent problems).  Bounded:
// an 8-GPU node runs eight of these processes side by side.  NGHMM_HOST_THREADS overrides.
int host_threads(uint64_t n_items) {
  static const int cap = [] {
    if (const char* env = std::getenv("NGHMM_HOST_THREADS")) {
      const int v = std::atoi(env);
      if (v >= 1) return v;
    }
    const int hw = omp_get_max_threads();
    return hw < 8 ? hw : 8;
  }();
  // measured: at 1000 problems a round's host work (~0.1 ms) is not worth waking threads
  // for; at 8000 it is 5 ms -> 1 ms on 8 threads
  static const uint64_t per_thread = [] {
    if (const char* env = std::getenv("NGHMM_HOST_WORK")) {
      const long v = std::atol(env);
      if (v >= 1) return (uint64_t)v;
    }
    return (uint64_t)512;
  }();
  const uint64_t by_work = n_items / per_thread;
  return (int)(by_work < 1 ? 1 : (by_work < (uint64_t)cap ? by_work : (uint64_t)cap));
}
}  // namespace

void BfgsBatch::reserve(uint64_t n_ind) {
  if (probs_.size() != n_ind) {
    probs_.clear();
    probs_.resize(n_ind);
  }
  for (auto& p : probs_) p.solver.configure(2, 10);
  // (the first parallel region of a process starts the OpenMP runtime: not inside an M-step)
  if (max_threads_ > 1) {
    int sink = 0;
#pragma omp parallel for num_threads(host_threads(4 * 512)) reduction(+ : sink)
    for (int i = 0; i < 64; ++i) sink += i;
    (void)sink;
  }
}

void BfgsBatch::begin(uint64_t n_ind, const double* indF, const double* alpha, bool F_fixed,
                      bool alpha_fixed) {
  // the problems (and their solvers' work arrays) are reused from call to call
  if (probs_.size() != n_ind) {
    probs_.clear();
    probs_.resize(n_ind);
  }
  rounds_ = 0;
  points_ = 0;
  ref_calls_ = 0;
  ind_rounds_ = 0;
  n_active_ = n_ind;
  for (uint64_t i = 0; i < n_ind; ++i) {
    Problem& p = probs_[i];
    p.solver.configure(2, 10);  // MVAL, shared/bfgs.h:23; start() zeroes the work arrays
    bfgs_problem_begin(p.p, indF[i], alpha[i], F_fixed, alpha_fixed);
  }
}

// The points one objective + gradient evaluation needs (bfgs.cpp:22-43,54).
void BfgsBatch::plan(Problem& p) {
  if (det_pow_) bfgs_plan<DetPow>(p.p);
  else bfgs_plan<LibmPow>(p.p);
}

uint64_t BfgsBatch::active_in(uint64_t lo, uint64_t hi) const {
  if (hi > probs_.size()) hi = probs_.size();
  uint64_t n = 0;
  for (uint64_t i = lo; i < hi; ++i) n += probs_[i].p.active ? 1 : 0;
  return n;
}

size_t BfgsBatch::gather(std::vector<uint32_t>& ind, std::vector<double>& F,
                         std::vector<double>& alpha, uint64_t lo, uint64_t hi) {
  ind.clear();
  F.clear();
  alpha.clear();
  if (n_active_ == 0) return 0;
  if (hi > probs_.size()) hi = probs_.size();
  const bool whole = (lo == 0 && hi == probs_.size());
  const uint64_t n_act = whole ? n_active_ : active_in(lo, hi);
  if (n_act == 0) return 0;
#pragma omp parallel for schedule(static) num_threads(host_threads(n_act))
  for (int64_t i = (int64_t)lo; i < (int64_t)hi; ++i)
    if (probs_[i].p.active) plan(probs_[i]);
  for (int k = 0; k < 5; ++k) {
    for (size_t i = lo; i < hi; ++i) {
      BfgsProblem& p = probs_[i].p;
      if (!p.active || !p.slot_used[k] || p.slot_nonfinite[k]) continue;
      p.slot_pos[k] = (uint32_t)ind.size();
      ind.push_back((uint32_t)i);
      F.push_back(p.pt[k][0]);
      alpha.push_back(p.pt[k][1]);
    }
  }
  // rounds = the longest sequence of evaluations any individual has needed: the number of
  // lock-step rounds when every gather covers everybody, and the same figure when the
  // individuals are gathered in parts (two-lane M-step), whatever the number of launches
  for (size_t i = lo; i < hi; ++i) {
    BfgsProblem& p = probs_[i].p;
    if (!p.active) continue;
    if (++p.n_rounds > rounds_) rounds_ = p.n_rounds;
  }
  points_ += ind.size();
  ind_rounds_ += n_act;
  return ind.size();
}

void BfgsBatch::consume(Problem& p, const double* lkl, uint64_t& ref_calls, uint64_t& finished) {
  double lklv[5] = {0, 0, 0, 0, 0};
  for (int k = 0; k < 5; ++k)
    if (p.p.slot_used[k] && !p.p.slot_nonfinite[k]) lklv[k] = lkl[p.p.slot_pos[k]];
  const bool again = bfgs_consume(p.p, p.solver, lklv, ref_calls, [&](BfgsProblem& q) {
    const int nbd[2] = {2, 2};
    p.solver.start(q.x, q.lb, q.ub, nbd, 1.0e6, 1.0e-3);  // FACTR, PGTOL: bfgs.h:24-25
  });
  if (!again) ++finished;
}

void BfgsBatch::scatter(const double* lkl, uint64_t lo, uint64_t hi) {
  if (hi > probs_.size()) hi = probs_.size();
  uint64_t ref_calls = 0, finished = 0;
  // What a machine does with its values varies by a factor of five: far from the optimum (the
  // first EM iterations of a run: 18 and 11 rounds) nearly every round ends a line search and
  // starts a new L-BFGS-B iteration -- matrix updates, Cauchy point, subspace step: 0.5 us per
  // machine, 0.5 ms per round at 1000 individuals, 9 ms of a first iteration's 76 -- near it (4-5
  // rounds) most rounds only continue one (0.1 us).  An M-step that is still going after five
  // rounds is of the first kind: from there on a few host threads share the machines even when
  // their number alone would not call for it.  (Never where several handles of one process run
  // their M-steps side by side -- replicas, groups, chains: set_max_threads.)
  int nt = host_threads(n_active_);
  if (nt == 1 && max_threads_ > 1 && rounds_ > 5 && n_active_ >= 256) nt = host_threads(4 * 512);
  if (nt > max_threads_) nt = max_threads_;
#pragma omp parallel for schedule(static) reduction(+ : ref_calls, finished) num_threads(nt)
  for (int64_t i = (int64_t)lo; i < (int64_t)hi; ++i)
    if (probs_[i].p.active) consume(probs_[i], lkl, ref_calls, finished);
  ref_calls_ += ref_calls;
  n_active_ -= finished;
}

void BfgsBatch::result(double* indF, double* alpha) const {
  for (size_t i = 0; i < probs_.size(); ++i) {
    indF[i] = probs_[i].p.x[0];
    alpha[i] = probs_[i].p.x[1];
  }
}

}  // namespace nghmm

// bfgs_batch.hpp -- the reference's per-individual indF/alpha optimisation
// (thread_slave type 4, EM.cpp:423-440; findmax_bfgs + getgradient/Yanggradient,
// shared/bfgs.cpp:22-138) for ALL individuals at once.
//
// The reference runs one blocking findmax_bfgs per pool task; every objective
// call is a full forward pass on the CPU.  Here each individual is a resumable
// Lbfgsb state machine; a "round" collects the (F, alpha) points all still-active
// machines need next (f(x) plus the finite-difference probes: at most five
// distinct points per individual), the caller evaluates them in one GPU launch,
// and the values are scattered back.  Per individual the sequence of points and
// every arithmetic step is the reference's, so the final (indF, alpha) are
// bit-identical to running findmax_bfgs on the same objective values.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "bfgs_problem.hpp"
#include "lbfgsb.hpp"

namespace nghmm {

class BfgsBatch {
 public:
  // Bounds as in EM.cpp:424-436.
  void begin(uint64_t n_ind, const double* indF, const double* alpha, bool F_fixed,
             bool alpha_fixed);

  // Points wanted this round, ordered probe-slot-major so that lanes of a wave
  // read consecutive individuals.  Returns the count (0 when all are done).
  // [lo, hi): only the individuals of that range (two halves can then be in flight at once:
  // the host digests one half's values while the GPU evaluates the other's points).
  size_t gather(std::vector<uint32_t>& ind, std::vector<double>& F, std::vector<double>& alpha,
                uint64_t lo = 0, uint64_t hi = ~0ull);

  // lkl[p] = forward log-likelihood of point p of the last gather() over the same range.
  void scatter(const double* lkl, uint64_t lo = 0, uint64_t hi = ~0ull);

  // the solvers' storage (and the OpenMP runtime) ahead of the first M-step: a run's first
  // iteration should not pay 15 ms for them
  void reserve(uint64_t n_ind);
  // host threads one M-step may use (1: a handle that runs next to others of its process)
  void set_max_threads(int n) { max_threads_ = n < 1 ? 1 : n; }
  // getgradient's step size by detmath's exp / log instead of libm's pow (bfgs_problem.hpp):
  // fast mode, where the device advances the same machines (kernels_bfgs.hip)
  void set_det_pow(bool on) { det_pow_ = on; }
  bool done() const { return n_active_ == 0; }
  uint64_t active_in(uint64_t lo, uint64_t hi) const;
  void result(double* indF, double* alpha) const;

  uint32_t rounds() const { return rounds_; }
  uint64_t points() const { return points_; }
  uint64_t ref_forward_calls() const { return ref_calls_; }
  uint64_t ind_rounds() const { return ind_rounds_; }

 private:
  struct Problem {
    Lbfgsb solver;
    BfgsProblem p;   // bfgs_problem.hpp: the part the device runs too
  };
  std::vector<Problem> probs_;
  uint64_t n_active_ = 0;
  int max_threads_ = 64;      // set_max_threads
  bool det_pow_ = false;      // set_det_pow
  uint32_t rounds_ = 0;
  uint64_t points_ = 0, ref_calls_ = 0, ind_rounds_ = 0;

  void plan(Problem& p);
  void consume(Problem& p, const double* lkl, uint64_t& ref_calls, uint64_t& finished);
};

}  // namespace nghmm

// lbfgsb.hpp -- bound-constrained limited-memory BFGS (L-BFGS-B 2.1 semantics),
// written as a resumable solver object so that many independent problems can be
// advanced in lock-step while their objective values come from one batched GPU
// launch per round.
//
// What it replaces in the reference: shared/bfgs.cpp:173-5915 (setulb_ and the
// routines below it, an f2c translation of Zhu/Byrd/Lu/Nocedal's L-BFGS-B 2.1)
// as driven by findmax_bfgs (shared/bfgs.cpp:83-138).  The arithmetic of every
// routine follows the published algorithm in the same operation order, because
// the reference's finite-difference M-step makes the EM trajectory depend on the
// last bit of every quantity here (SURVEY.md finding 4, section 8 rows a8/a9).
// tests/test_lbfgsb_ref.py pins it bit for bit against the reference's own
// object code (oracle/_ref/libref_bfgs.so) on the same objective.
//
// The routines themselves live in lbfgsb_core.hpp (LbfgsbT<Store>: one implementation for
// host and device); this class is that solver over std::vector storage of any (n, m).
#pragma once

#include <cstdint>
#include <vector>

#include "lbfgsb_core.hpp"

namespace nghmm {

// the work arrays of one problem in one std::vector (layout: LbfgsbPtrs::bind)
struct VecStore : LbfgsbPtrs {
  std::vector<double> buf;
  // dpmeps (bfgs.cpp:5166): smallest power of the radix with 1 + eps != 1; on
  // IEEE binary64 with round-to-nearest this evaluates to 2^-52.
  static double machine_eps();
};

class Lbfgsb : private LbfgsbT<VecStore> {
  using Core = LbfgsbT<VecStore>;

 public:
  using Task = LbfgsbTask;

  Lbfgsb() = default;
  Lbfgsb(int n, int m) { reset(n, m); }
  // (the store's pointers aim into its own vector: re-bind after a copy or move)
  Lbfgsb(const Lbfgsb& o) : Core(o) { rebind(); }
  Lbfgsb(Lbfgsb&& o) noexcept : Core(std::move(o)) { rebind(); }
  Lbfgsb& operator=(const Lbfgsb& o) {
    Core::operator=(o);
    rebind();
    return *this;
  }
  Lbfgsb& operator=(Lbfgsb&& o) noexcept {
    Core::operator=(std::move(o));
    rebind();
    return *this;
  }

  void reset(int n, int m);
  // reuse this object for another problem of size (n, m): scalars as newly constructed,
  // storage kept; follow with start()
  void configure(int n, int m);

  // Begin a minimisation.  x0 is copied; bounds as in the reference:
  // nbd[i] = 0 none, 1 lower, 2 both, 3 upper (bfgs.h:27-33).
  void start(const double* x0, const double* l, const double* u, const int* nbd,
             double factr, double pgtol);

  // One call of the reference's setulb_.  `f` and `g[n]` are read when the
  // previous task was EvalFG (and are overwritten with the restored values when
  // a failed line search rolls back, exactly as the reference writes through
  // its f/g pointers).
  Task advance(double* f, double* g) { return Core::advance(f, g); }

  const double* x() const { return st_.x; }
  double* x_mut() { return st_.x; }
  int n() const { return n_; }
  int iterations() const { return iter_; }
  int evaluations() const { return nfgv_; }
  int info() const { return info_; }

 private:
  void rebind() {
    if (!st_.buf.empty()) st_.bind(st_.buf.data(), n_, m_);
  }
};

}  // namespace nghmm

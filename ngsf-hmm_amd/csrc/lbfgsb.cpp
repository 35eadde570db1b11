// lbfgsb.cpp -- the host solver's storage (see lbfgsb.hpp; the routines are in
// lbfgsb_core.hpp).  Compile with -ffp-contract=off.
#include "lbfgsb.hpp"

#include <utility>

namespace nghmm {

double VecStore::machine_eps() {
  volatile double a = 1.0;
  for (;;) {
    volatile double t = 1.0 + a * 0.5;
    if (t == 1.0) break;
    a = a * 0.5;
  }
  return a;
}

// A solver object reused for another minimisation: every scalar back to its initial value
// (as a newly constructed object), the work arrays keep their storage.  start() sizes and
// zeroes them.
void Lbfgsb::configure(int n, int m) {
  clear_scalars();
  factr_ = pgtol_ = 0;
  n_ = n;
  m_ = m;
}

void Lbfgsb::reset(int n, int m) {
  n_ = n;
  m_ = m;
  // The reference calloc()s its work arrays once per findmax_bfgs call
  // (bfgs.cpp:103-105); start() re-zeroes them.
  st_.buf.assign(LbfgsbPtrs::doubles(n, m), 0.0);
  st_.bind(st_.buf.data(), n, m);
  phase_ = LbfgsbPhase::Start;
}

void Lbfgsb::start(const double* x0, const double* l, const double* u, const int* nbd,
                   double factr, double pgtol) {
  const int n = n_, m = m_;
  reset(n, m);
  for (int i = 0; i < n; ++i) {
    st_.x[i] = x0[i];
    st_.l[i] = l[i];
    st_.u[i] = u[i];
    st_.nbd[i] = nbd ? nbd[i] : 2;
  }
  factr_ = factr;
  pgtol_ = pgtol;
  ls_ = LbfgsbLsState();
  ls_task_ = LbfgsbLs::Start;
  phase_ = LbfgsbPhase::Start;
}

}  // namespace nghmm

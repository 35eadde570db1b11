// bfgs_shim.cpp -- the reference's findmax_bfgs driver (shared/bfgs.cpp:83-138)
// and its finite-difference gradient (shared/bfgs.cpp:22-65), restated on top of
// the product's L-BFGS-B core (ngsf-hmm_amd/csrc/lbfgsb.cpp).  TEST
// INFRASTRUCTURE ONLY (see ngsfhmm_oracle.h).
//
// The scalar, one-problem-at-a-time control flow is kept exactly, including the
// redundant evaluations (the objective is called again for f0 inside the
// gradient, and the START call re-requests f and g at the same x), so that the
// number of forward passes measured on the CPU baseline is the reference's.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "../ngsf-hmm_amd/csrc/lbfgsb.hpp"
#include "ngsfhmm_oracle.h"

namespace {

typedef double (*objective_fn)(const double x[], const void*);

// shared/bfgs.cpp:22-43 (Yanggradient)
void yang_gradient(int n, const double* x, double f0, double* g, const void* dats,
                   objective_fn fun, double* space, const double* lowbound,
                   const double* upbound) {
  double* x0 = space;
  double* x1 = space + n;
  const double eh01 = 1.e-8;
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) x0[j] = x1[j] = x[j];
    double eh = std::pow(eh01 * (std::fabs(x[i]) + 1), 0.67);
    x0[i] -= eh;
    x1[i] += eh;
    if (x0[i] < lowbound[i]) {
      x1[i] += eh;
      g[i] = (fun(x1, dats) - f0) / (eh * 2.0);
    } else if (x1[i] > upbound[i]) {
      x0[i] -= eh;
      g[i] = (f0 - fun(x0, dats)) / (eh * 2.0);
    } else {
      g[i] = (fun(x1, dats) - fun(x0, dats)) / (eh * 2.0);
    }
  }
}

// shared/bfgs.cpp:45-65 (getgradient)
void get_gradient(int npar, const double* invec, double* outvec, const void* dats,
                  objective_fn func, const double* lowbound, const double* upbound) {
  std::vector<double> space(2 * npar + 10, 0.0);
  double f0 = func(invec, dats);
  yang_gradient(npar, invec, f0, outvec, dats, func, space.data(), lowbound, upbound);
  for (int i = 0; i < npar; i++) {
    if (invec[i] <= lowbound[i] && outvec[i] > 0.0) outvec[i] = 0.0;
    if (invec[i] >= upbound[i] && outvec[i] < 0.0) outvec[i] = 0.0;
  }
}

}  // namespace

// shared/bfgs.cpp:83-138; MVAL/FACTR/PGTOL from shared/bfgs.h:23-25.
extern "C" double orc_findmax_bfgs(int numpars, double* invec, const void* dats,
                                   double (*fun)(const double x[], const void*),
                                   void (*dfun)(const double x[], double y[]), double* lowbound,
                                   double* upbound, int* nbd, int noisy) {
  (void)noisy;
  const int m = 10;
  const double factr = 1.0e6, pgtol = 1.0e-3;
  std::vector<double> grad(numpars, 0.0);
  nghmm::Lbfgsb solver(numpars, m);

  double like = fun(invec, dats);
  if (dfun)
    dfun(invec, grad.data());
  else
    get_gradient(numpars, invec, grad.data(), dats, fun, lowbound, upbound);

  solver.start(invec, lowbound, upbound, nbd, factr, pgtol);
  for (;;) {
    nghmm::Lbfgsb::Task task = solver.advance(&like, grad.data());
    for (int i = 0; i < numpars; i++) invec[i] = solver.x()[i];
    if (task == nghmm::Lbfgsb::Task::EvalFG) {
      like = fun(invec, dats);
      if (dfun)
        dfun(invec, grad.data());
      else
        get_gradient(numpars, invec, grad.data(), dats, fun, lowbound, upbound);
      continue;
    }
    if (task == nghmm::Lbfgsb::Task::NewX) continue;
    break;
  }
  return -like;
}

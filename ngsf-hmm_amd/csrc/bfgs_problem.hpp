// bfgs_problem.hpp -- one individual's (indF, alpha) optimisation between two objective rounds:
// the reference's findmax_bfgs + getgradient (shared/bfgs.cpp:22-65,83-138) around the solver of
// lbfgsb_core.hpp, as host/device code.  The host (BfgsBatch, bfgs_batch.cpp) and the device
// (k_bfgs_advance, kernels_bfgs.hip) run THESE functions, so that for the same objective values
// both ask for the same points and take the same steps, bit for bit.
//
// The one libm call of the path is getgradient's step size eh = pow(1e-8 (|x| + 1), 0.67)
// (bfgs.cpp:33).  Exact mode keeps the host's pow (the oracle's det build calls it too); fast mode
// -- whose objective is not the reference's bits anyway -- uses exp(0.67 log(.)) from detmath.h
// on both sides (DetPow), at most a few ulp from pow: the probe moves by 1e-21, the gradient by
// 1e-16 of itself.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#include "detmath.h"
#include "lbfgsb_core.hpp"

namespace nghmm {

constexpr double kBfgsINF = 1e15;  // shared/gen_func.hpp:15

// What one problem carries from round to round (plain data: it lives in a std::vector on the
// host and in device memory on the GPU).
struct BfgsProblem {
  double x[2], lb[2], ub[2];
  double like, grad[2];
  double eval_x[2];  // where (like, grad) were last evaluated
  double eh[2];
  double pt[5][2];   // slot 0 = x; slots 1,2 = param 0 probes; 3,4 = param 1 probes
  uint32_t slot_pos[5];
  uint32_t n_rounds;  // rounds this problem has had points in
  // accounting of this M-step (the device sums these over the individuals when the M-step ends;
  // the host's BfgsBatch keeps totals of its own)
  uint32_t acc_points;      // points evaluated
  uint32_t acc_ref_calls;   // forward passes the reference would have spent (bfgs.cpp:54,114-121)
  uint32_t acc_redone;      // rounds repeated by the general kernel
  uint32_t acc_invalid;     // "invalid Lkl found!"
  // plan of the current round
  int8_t probe_kind[2];  // 0 central, 1 forward (x + 2eh), 2 backward (x - 2eh), 3 fixed (skipped)
  uint8_t slot_used[5];
  uint8_t slot_nonfinite[5];
  uint8_t have_eval, started, active;
  uint8_t big_valid;  // device: the machine's matrices (ws .. wa) have been written to memory (else: zeros)
};

NGHMM_HD inline bool bfgs_nonfinite(double v) { return !(v - v == 0.0); }  // NaN or +-inf
NGHMM_HD inline bool bfgs_same_bits(const double a[2], const double b[2]) {
  uint64_t ua[2], ub[2];
  __builtin_memcpy(ua, a, 16);
  __builtin_memcpy(ub, b, 16);
  return ua[0] == ub[0] && ua[1] == ub[1];
}

// Bounds as in EM.cpp:424-436.
NGHMM_HD inline void bfgs_problem_begin(BfgsProblem& p, double F, double alpha, bool F_fixed,
                                        bool alpha_fixed) {
  p.x[0] = F;
  p.x[1] = alpha;
  p.lb[0] = 1 / kBfgsINF;
  p.lb[1] = 1 / kBfgsINF;
  p.ub[0] = 1 - p.lb[0];
  p.ub[1] = 10;
  if (F_fixed) p.lb[0] = p.ub[0] = F;
  if (alpha_fixed) p.lb[1] = p.ub[1] = alpha;
  p.like = 0;
  p.n_rounds = 0;
  p.acc_points = p.acc_ref_calls = p.acc_redone = p.acc_invalid = 0;
  p.grad[0] = p.grad[1] = 0;
  p.eval_x[0] = p.eval_x[1] = 0;
  p.have_eval = 0;
  p.started = 0;
  p.active = 1;
  p.big_valid = 0;
}

struct LibmPow {
  static double eh(double ax) { return std::pow(1.e-8 * (ax + 1), 0.67); }
};
struct DetPow {
  NGHMM_HD static double eh(double ax) { return det_exp(0.67 * det_log(1.e-8 * (ax + 1))); }
};

// The points one objective + gradient evaluation needs (bfgs.cpp:22-43,54).
template <class Pow>
NGHMM_HD inline void bfgs_plan(BfgsProblem& p) {
  for (int k = 0; k < 5; ++k) p.slot_used[k] = 0;
  p.pt[0][0] = p.x[0];
  p.pt[0][1] = p.x[1];
  p.slot_used[0] = 1;
  for (int i = 0; i < 2; ++i) {
    const int sa = 1 + 2 * i, sb = 2 + 2 * i;
    if (p.lb[i] == p.ub[i]) {
      // Fixed parameter: the reference still spends one probe on it, but the
      // bound check (bfgs.cpp:58-63) then forces the component to zero.
      p.probe_kind[i] = 3;
      continue;
    }
    const double eh = Pow::eh(p.x[i] >= 0 ? p.x[i] : -p.x[i]);
    p.eh[i] = eh;
    double x0 = p.x[i], x1 = p.x[i];
    x0 -= eh;
    x1 += eh;
    for (int k = sa; k <= sb; ++k) {
      p.pt[k][0] = p.x[0];
      p.pt[k][1] = p.x[1];
    }
    if (x0 < p.lb[i]) {
      x1 += eh;
      p.probe_kind[i] = 1;
      p.pt[sa][i] = x1;
      p.slot_used[sa] = 1;
    } else if (x1 > p.ub[i]) {
      x0 -= eh;
      p.probe_kind[i] = 2;
      p.pt[sa][i] = x0;
      p.slot_used[sa] = 1;
    } else {
      p.probe_kind[i] = 0;
      p.pt[sa][i] = x1;
      p.pt[sb][i] = x0;
      p.slot_used[sa] = p.slot_used[sb] = 1;
    }
  }
  for (int k = 0; k < 5; ++k)
    p.slot_nonfinite[k] =
        p.slot_used[k] && (bfgs_nonfinite(p.pt[k][0]) || bfgs_nonfinite(p.pt[k][1]));
}

// The round's values as objective and finite-difference gradient (bfgs.cpp:22-65) at p.x.
// lklv[k] = forward log-likelihood of slot k (read where the slot is used and finite).  Returns
// the objective calls the reference spends on one such evaluation.
NGHMM_HD inline uint64_t bfgs_gradient(BfgsProblem& p, const double lklv[5]) {
  // objective = -forward log-likelihood; non-finite parameters give -INF... i.e.
  // lkl = INF and the function returns -lkl (EM.cpp:454-463)
  double fv[5] = {0, 0, 0, 0, 0};
  for (int k = 0; k < 5; ++k) {
    if (!p.slot_used[k]) continue;
    fv[k] = p.slot_nonfinite[k] ? -kBfgsINF : -lklv[k];
  }
  const double f0 = fv[0];
  p.like = fv[0];
  uint64_t calls = 2;  // fun(x) in findmax_bfgs + fun(x) again inside getgradient
  for (int i = 0; i < 2; ++i) {
    const int sa = 1 + 2 * i, sb = 2 + 2 * i;
    double g;
    switch (p.probe_kind[i]) {
      case 0:
        g = (fv[sa] - fv[sb]) / (p.eh[i] * 2.0);
        calls += 2;
        break;
      case 1:
        g = (fv[sa] - f0) / (p.eh[i] * 2.0);
        calls += 1;
        break;
      case 2:
        g = (f0 - fv[sa]) / (p.eh[i] * 2.0);
        calls += 1;
        break;
      default:
        g = 0.0;
        calls += 1;
        break;
    }
    if (p.x[i] <= p.lb[i] && g > 0.0) g = 0.0;  // bfgs.cpp:58-63
    if (p.x[i] >= p.ub[i] && g < 0.0) g = 0.0;
    p.grad[i] = g;
  }
  p.eval_x[0] = p.x[0];
  p.eval_x[1] = p.x[1];
  p.have_eval = 1;
  return calls;
}

// One setulb_ call (bfgs.cpp:108-133) and what findmax_bfgs does with its answer:
// 0 = call again (NEW_X, or the START call asking for f and g at the point just evaluated:
// same x, same values, bfgs.cpp:901,114-121), 1 = wants another round, 2 = finished
// (p.active = 0).  `calls`: what bfgs_gradient returned for the evaluation in hand.
template <class Solver>
NGHMM_HD inline int bfgs_step(BfgsProblem& p, Solver& solver, uint64_t calls, uint64_t& ref_calls) {
  const LbfgsbTask task = solver.advance(&p.like, p.grad);
  p.x[0] = solver.x()[0];
  p.x[1] = solver.x()[1];
  if (task == LbfgsbTask::EvalFG) {
    if (p.have_eval && bfgs_same_bits(p.x, p.eval_x)) {
      ref_calls += calls;
      return 0;
    }
    return 1;
  }
  if (task == LbfgsbTask::NewX) return 0;
  p.active = 0;
  return 2;
}

// The round's values into the machine: gradient, then setulb_ calls until it wants another
// evaluation or ends.  `start(p)` begins the solver at p.x with p's bounds (nbd 2,2; FACTR,
// PGTOL: bfgs.h:24-25).  ref_calls += the objective calls the reference would have made.
// Returns true when the problem wants another round, false when it has finished (p.active = 0).
template <class Solver, class Start>
NGHMM_HD inline bool bfgs_consume(BfgsProblem& p, Solver& solver, const double lklv[5],
                                  uint64_t& ref_calls, Start start) {
  const uint64_t calls = bfgs_gradient(p, lklv);
  ref_calls += calls;
  if (!p.started) {
    start(p);
    p.started = 1;
  }
  for (;;) {
    const int r = bfgs_step(p, solver, calls, ref_calls);
    if (r) return r == 1;
  }
}

}  // namespace nghmm

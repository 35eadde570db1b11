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
#pragma once

#include <cstdint>
#include <vector>

namespace nghmm {

class Lbfgsb {
 public:
  // What the caller has to do next.
  enum class Task {
    EvalFG,       // evaluate f and g at x(), then call advance(f, g)
    NewX,         // an iteration finished; call advance() again (f, g ignored)
    ConvergedPG,  // |projected gradient|_inf <= pgtol            (bfgs.cpp:925,1136)
    ConvergedF,   // relative reduction of f <= factr * epsmch    (bfgs.cpp:1142)
    Abnormal,     // line search failed with empty memory         (bfgs.cpp:1080)
    Error         // invalid input (n, m, factr, bounds)          (bfgs.cpp:2309)
  };

  Lbfgsb() = default;
  Lbfgsb(int n, int m) { reset(n, m); }

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
  Task advance(double* f, double* g);

  const double* x() const { return x_.data(); }
  double* x_mut() { return x_.data(); }
  int n() const { return n_; }
  int iterations() const { return iter_; }
  int evaluations() const { return nfgv_; }
  int info() const { return info_; }

 private:
  // --- problem ---
  int n_ = 0, m_ = 0;
  std::vector<double> x_, l_, u_;
  std::vector<int> nbd_;
  double factr_ = 0, pgtol_ = 0;

  // --- limited-memory matrices (column-major, 1-based accessors in the .cpp) ---
  std::vector<double> ws_, wy_, sy_, ss_, wt_, wn_, snd_;
  std::vector<double> z_, r_, d_, t_, wa_;
  std::vector<int> index_, iwhere_, indx2_;

  // --- saved scalars (the reference's lsave/isave/dsave) ---
  enum class Phase { Start, FgStart, FgLnsrch, NewX, Done } phase_ = Phase::Start;
  bool prjctd_ = false, cnstnd_ = false, boxed_ = false, updatd_ = false;
  int nintol_ = 0, iback_ = 0, nskip_ = 0, head_ = 1, col_ = 0, itail_ = 0, iter_ = 0,
      iupdat_ = 0, nint_ = 0, nfgv_ = 0, info_ = 0, ifun_ = 0, iword_ = 0, nfree_ = 0,
      nact_ = 0, ileave_ = 0, nenter_ = 0;
  double theta_ = 1, fold_ = 0, tol_ = 0, dnorm_ = 0, epsmch_ = 0, gd_ = 0, stpmx_ = 0,
         sbgnrm_ = 0, stp_ = 0, gdold_ = 0, dtd_ = 0, xstep_ = 0;

  // --- More'-Thuente line search state (dcsrch's isave/dsave + its task word) ---
  enum class Ls { Start, FG, Convergence, Warning, Error } ls_task_ = Ls::Start;
  struct LsState {
    bool brackt = false;
    int stage = 0;
    double ginit = 0, gtest = 0, gx = 0, gy = 0, finit = 0, fx = 0, fy = 0, stx = 0, sty = 0,
           stmin = 0, stmax = 0, width = 0, width1 = 0;
  } ls_;

  // --- routines (names follow the published code) ---
  bool errclb();
  void active();
  void projgr(const double* g);
  void cauchy(const double* g, bool& ok);
  void freev(bool& wrk);
  void formk(bool& ok);
  void cmprlb(const double* g, bool& ok);
  void subsm(bool& ok);
  // returns true when an evaluation is requested, false when the search ended (NEW_X) or failed
  bool lnsrlb(double* f, double* g, bool fresh);
  void matupd(double rr, double dr);
  void formt(bool& ok);
  void bmv(const double* v, double* p, bool& ok);
  void dcsrch(double f, double g, double& stp, double stpmax);
  static void dcstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy,
                     double& stp, double fp, double dp, bool& brackt, double stpmin,
                     double stpmax);
  void refresh_memory();
};

}  // namespace nghmm

// lbfgsb_core.hpp -- the bound-constrained limited-memory BFGS solver (L-BFGS-B 2.1
// semantics; what it replaces in the reference: shared/bfgs.cpp:173-5915 as driven by
// findmax_bfgs, shared/bfgs.cpp:83-138) as ONE implementation that runs on the host and on
// the device.
//
// LbfgsbT<Store> holds the solver's saved scalars (the reference's lsave / isave / dsave and
// the line search's own) as plain members and reaches its work arrays through `Store`, a
// struct of raw pointers:
//   * host: VecStore (lbfgsb.hpp) owns std::vector storage of any (n, m) -- class Lbfgsb, the
//     solver tests/test_lbfgsb_ref.py pins bit for bit against the reference's own object code;
//   * device: PtrStore points into a block of LDS (kernels_bfgs.hip): one lane = one individual's
//     (F, alpha) problem, the per-individual M-step of EM.cpp:423-440 advanced on the GPU
//     between two objective rounds without a host round trip.
// Every routine uses IEEE add / sub / mul / div / sqrt only, in the operation order of the
// published code (summations in index order, products left to right); with -ffp-contract=off on
// both sides the device takes the host's steps bit for bit (tests/test_gpu_devbfgs.py, and on
// the CPU tests/test_host_logic.py through nghmm_bfgs_batch_core).
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define NGHMM_HD __host__ __device__
#else
#define NGHMM_HD
#endif

namespace nghmm {

// What the caller has to do next.
enum class LbfgsbTask {
  EvalFG,       // evaluate f and g at x(), then call advance(f, g)
  NewX,         // an iteration finished; call advance() again (f, g ignored)
  ConvergedPG,  // |projected gradient|_inf <= pgtol            (bfgs.cpp:925,1136)
  ConvergedF,   // relative reduction of f <= factr * epsmch    (bfgs.cpp:1142)
  Abnormal,     // line search failed with empty memory         (bfgs.cpp:1080)
  Error         // invalid input (n, m, factr, bounds)          (bfgs.cpp:2309)
};
enum class LbfgsbPhase { Start, FgStart, FgLnsrch, NewX, Done };
enum class LbfgsbLs { Start, FG, Convergence, Warning, Error };

// More'-Thuente line search state (dcsrch's isave/dsave)
struct LbfgsbLsState {
  bool brackt = false;
  int stage = 0;
  double ginit = 0, gtest = 0, gx = 0, gy = 0, finit = 0, fx = 0, fy = 0, stx = 0, sty = 0,
         stmin = 0, stmax = 0, width = 0, width1 = 0;
};

// The work arrays as raw pointers (column-major, 1-based accessors below):
//   x l u z r d t [n]; ws wy [n x m]; sy ss wt [m x m]; wn snd [2m x 2m]; wa [8m];
//   nbd index iwhere indx2 [n] (int)
struct LbfgsbPtrs {
  double *x = nullptr, *l = nullptr, *u = nullptr, *z = nullptr, *r = nullptr, *d = nullptr,
         *t = nullptr, *ws = nullptr, *wy = nullptr, *sy = nullptr, *ss = nullptr, *wt = nullptr,
         *wn = nullptr, *snd = nullptr, *wa = nullptr;
  int *nbd = nullptr, *index = nullptr, *iwhere = nullptr, *indx2 = nullptr;
  // doubles one problem of size (n, m) needs behind those pointers (the four int arrays
  // packed two to a double at the end)
  // ... of which the matrices ws .. wa, behind the 7n doubles of the vectors
  NGHMM_HD static constexpr size_t matrices(int n, int m) {
    return (size_t)2 * n * m + (size_t)3 * m * m + (size_t)8 * m * m + (size_t)8 * m;
  }
  NGHMM_HD static constexpr size_t doubles(int n, int m) {
    return (size_t)7 * n + (size_t)2 * n * m + (size_t)3 * m * m + (size_t)8 * m * m +
           (size_t)8 * m + (size_t)(4 * n + 1) / 2;
  }
  // carve the arrays out of one block
  NGHMM_HD void bind(double* base, int n, int m) {
    double* p = base;
    x = p; p += n;
    l = p; p += n;
    u = p; p += n;
    z = p; p += n;
    r = p; p += n;
    d = p; p += n;
    t = p; p += n;
    ws = p; p += (size_t)n * m;
    wy = p; p += (size_t)n * m;
    sy = p; p += (size_t)m * m;
    ss = p; p += (size_t)m * m;
    wt = p; p += (size_t)m * m;
    wn = p; p += (size_t)4 * m * m;
    snd = p; p += (size_t)4 * m * m;
    wa = p; p += (size_t)8 * m;
    int* q = reinterpret_cast<int*>(p);
    nbd = q; q += n;
    index = q; q += n;
    iwhere = q; q += n;
    indx2 = q;
  }
};

// pointers into a block the caller owns (device: LDS; host tests: a plain buffer)
struct PtrStore : LbfgsbPtrs {
  // dpmeps (bfgs.cpp:5166) evaluates to 2^-52 on IEEE binary64 with round-to-nearest
  NGHMM_HD static double machine_eps() { return 2.220446049250313080847263336181640625e-16; }
};

namespace lbfgsb_detail {

NGHMM_HD inline double absd(double v) { return v >= 0 ? v : -v; }        // bfgs.cpp:147 macro
NGHMM_HD inline double maxd(double a, double b) { return a >= b ? a : b; }  // bfgs.cpp:150 macro
NGHMM_HD inline double mind(double a, double b) { return a <= b ? a : b; }  // bfgs.cpp:149 macro

// BLAS-1 pieces the algorithm uses (bfgs.cpp:5200-5470): plain index-order loops;
// the reference's unrolled forms accumulate in the same order.
NGHMM_HD inline double dot(int n, const double* a, const double* b) {
  double acc = 0.0;
  for (int i = 0; i < n; ++i) acc += a[i] * b[i];
  return acc;
}
NGHMM_HD inline void axpy(int n, double da, const double* x, double* y) {
  if (n <= 0 || da == 0.0) return;
  for (int i = 0; i < n; ++i) y[i] += da * x[i];
}

// LINPACK dpofa: Cholesky factor of a symmetric positive definite matrix stored
// in the upper triangle (bfgs.cpp:5560-5610).  Returns 0 or the failing order.
NGHMM_HD inline int cholesky_upper(double* a, int lda, int n) {
  auto A = [&](int i, int j) -> double& { return a[(i - 1) + (size_t)(j - 1) * lda]; };
  for (int j = 1; j <= n; ++j) {
    double s = 0.0;
    for (int k = 1; k <= j - 1; ++k) {
      double t = A(k, j) - dot(k - 1, &A(1, k), &A(1, j));
      t /= A(k, k);
      A(k, j) = t;
      s += t * t;
    }
    s = A(j, j) - s;
    if (s <= 0.0) return j;
    A(j, j) = std::sqrt(s);
  }
  return 0;
}

// LINPACK dtrsl: triangular solves (bfgs.cpp:5700-5915).  job 00: T x = b, T
// lower; 01: T x = b, T upper; 10: T' x = b, T lower; 11: T' x = b, T upper.
NGHMM_HD inline int tri_solve(const double* t, int ldt, int n, double* b, int job) {
  auto T = [&](int i, int j) -> const double& { return t[(i - 1) + (size_t)(j - 1) * ldt]; };
  for (int d = 1; d <= n; ++d)
    if (T(d, d) == 0.0) return d;
  int kind = (job % 10 != 0) ? 2 : 1;
  if ((job % 100) / 10 != 0) kind += 2;
  double* B = b - 1;  // 1-based view
  switch (kind) {
    case 1:
      B[1] /= T(1, 1);
      for (int j = 2; j <= n; ++j) {
        double temp = -B[j - 1];
        axpy(n - j + 1, temp, &T(j, j - 1), &B[j]);
        B[j] /= T(j, j);
      }
      break;
    case 2:
      B[n] /= T(n, n);
      for (int jj = 2; jj <= n; ++jj) {
        int j = n - jj + 1;
        double temp = -B[j + 1];
        axpy(j, temp, &T(1, j + 1), &B[1]);
        B[j] /= T(j, j);
      }
      break;
    case 3:
      B[n] /= T(n, n);
      for (int jj = 2; jj <= n; ++jj) {
        int j = n - jj + 1;
        B[j] -= dot(jj - 1, &T(j + 1, j), &B[j + 1]);
        B[j] /= T(j, j);
      }
      break;
    default:
      B[1] /= T(1, 1);
      for (int j = 2; j <= n; ++j) {
        B[j] -= dot(j - 1, &T(1, j), &B[1]);
        B[j] /= T(j, j);
      }
      break;
  }
  return 0;
}

// Heap step of the breakpoint sort (hpsolb, bfgs.cpp:3020-3130).
NGHMM_HD inline void heap_pop_min(int n, double* t1, int* iorder1, bool heap_built) {
  double* t = t1 - 1;
  int* iorder = iorder1 - 1;
  if (!heap_built) {
    for (int k = 2; k <= n; ++k) {
      double ddum = t[k];
      int indxin = iorder[k];
      int i = k;
      while (i > 1) {
        int j = i / 2;
        if (ddum < t[j]) {
          t[i] = t[j];
          iorder[i] = iorder[j];
          i = j;
        } else {
          break;
        }
      }
      t[i] = ddum;
      iorder[i] = indxin;
    }
  }
  if (n > 1) {
    int i = 1;
    double out = t[1];
    int indxou = iorder[1];
    double ddum = t[n];
    int indxin = iorder[n];
    for (;;) {
      int j = i + i;
      if (j <= n - 1) {
        if (t[j + 1] < t[j]) ++j;
        if (t[j] < ddum) {
          t[i] = t[j];
          iorder[i] = iorder[j];
          i = j;
          continue;
        }
      }
      break;
    }
    t[i] = ddum;
    iorder[i] = indxin;
    t[n] = out;
    iorder[n] = indxou;
  }
}

}  // namespace lbfgsb_detail

// Store: a struct with the members of LbfgsbPtrs and `static double machine_eps()`.
template <class Store>
struct LbfgsbT {
  using Task = LbfgsbTask;
  Store st_;

  // --- problem ---
  int n_ = 0, m_ = 0;
  double factr_ = 0, pgtol_ = 0;

  // --- saved scalars (the reference's lsave/isave/dsave) ---
  LbfgsbPhase phase_ = LbfgsbPhase::Start;
  bool prjctd_ = false, cnstnd_ = false, boxed_ = false, updatd_ = false;
  int nintol_ = 0, iback_ = 0, nskip_ = 0, head_ = 1, col_ = 0, itail_ = 0, iter_ = 0,
      iupdat_ = 0, nint_ = 0, nfgv_ = 0, info_ = 0, ifun_ = 0, iword_ = 0, nfree_ = 0,
      nact_ = 0, ileave_ = 0, nenter_ = 0;
  double theta_ = 1, fold_ = 0, tol_ = 0, dnorm_ = 0, epsmch_ = 0, gd_ = 0, stpmx_ = 0,
         sbgnrm_ = 0, stp_ = 0, gdold_ = 0, dtd_ = 0, xstep_ = 0;

  // --- More'-Thuente line search state (dcsrch's isave/dsave + its task word) ---
  LbfgsbLs ls_task_ = LbfgsbLs::Start;
  LbfgsbLsState ls_;

  // the scalars of a solver about to start() (the arrays are the caller's to zero)
  NGHMM_HD void clear_scalars() {
    phase_ = LbfgsbPhase::Start;
    prjctd_ = cnstnd_ = boxed_ = updatd_ = false;
    nintol_ = iback_ = nskip_ = 0;
    head_ = 1;
    col_ = itail_ = iter_ = iupdat_ = nint_ = nfgv_ = info_ = ifun_ = iword_ = nfree_ = nact_ =
        ileave_ = nenter_ = 0;
    theta_ = 1;
    fold_ = tol_ = dnorm_ = epsmch_ = gd_ = stpmx_ = sbgnrm_ = stp_ = gdold_ = dtd_ = xstep_ = 0;
    ls_task_ = LbfgsbLs::Start;
    ls_ = LbfgsbLsState();
  }

  NGHMM_HD const double* x() const { return st_.x; }

  // Begin a minimisation of size (n, m) on a store whose pointers are bound: the work arrays
  // zeroed as the reference calloc()s them per findmax_bfgs call (bfgs.cpp:103-105), x0 and
  // the bounds copied in (nbd[i] = 0 none, 1 lower, 2 both, 3 upper: bfgs.h:27-33).
  // zero_matrices = false: only the vectors and index arrays are zeroed here -- the caller zeroes
  // ws .. wa (the doubles [7n, 7n + matrices(n, m)) of the block) before the solver first touches
  // them, which is not before its first iteration has ended (kernels_bfgs.hip stages them lazily)
  NGHMM_HD void start_bound(int n, int m, const double* x0, const double* l, const double* u,
                            const int* nbd, double factr, double pgtol, bool zero_matrices = true) {
    clear_scalars();
    n_ = n;
    m_ = m;
    if (zero_matrices) {
      const size_t nd = LbfgsbPtrs::doubles(n, m);
      for (size_t k = 0; k < nd; ++k) st_.x[k] = 0.0;  // one block from x on (LbfgsbPtrs::bind)
    } else {
      for (int k = 0; k < 7 * n; ++k) st_.x[k] = 0.0;
      for (int k = 0; k < 4 * n; ++k) st_.nbd[k] = 0;    // nbd, index, iwhere, indx2: one run of ints
    }
    for (int i = 0; i < n; ++i) {
      st_.x[i] = x0[i];
      st_.l[i] = l[i];
      st_.u[i] = u[i];
      st_.nbd[i] = nbd ? nbd[i] : 2;
    }
    factr_ = factr;
    pgtol_ = pgtol;
  }

  // One call of the reference's setulb_.  `f` and `g[n]` are read when the previous task was
  // EvalFG (and are overwritten with the restored values when a failed line search rolls back,
  // exactly as the reference writes through its f/g pointers).
  NGHMM_HD LbfgsbTask advance(double* f, double* g);

  // --- routines (names follow the published code) ---
  NGHMM_HD bool errclb();
  NGHMM_HD void active();
  NGHMM_HD void projgr(const double* g);
  NGHMM_HD void cauchy(const double* g, bool& ok);
  NGHMM_HD void freev(bool& wrk);
  NGHMM_HD void formk(bool& ok);
  NGHMM_HD void cmprlb(const double* g, bool& ok);
  NGHMM_HD void subsm(bool& ok);
  // returns true when an evaluation is requested, false when the search ended (NEW_X) or failed
  NGHMM_HD bool lnsrlb(double* f, double* g, bool fresh);
  NGHMM_HD void matupd(double rr, double dr);
  NGHMM_HD void formt(bool& ok);
  NGHMM_HD void bmv(const double* v, double* p, bool& ok);
  NGHMM_HD void dcsrch(double f, double g, double& stp, double stpmax);
  NGHMM_HD static void dcstep(double& stx, double& fx, double& dx, double& sty, double& fy,
                              double& dy, double& stp, double fp, double dp, bool& brackt,
                              double stpmin, double stpmax);
  NGHMM_HD void refresh_memory();
};

#define X(i) st_.x[(i) - 1]
#define L(i) st_.l[(i) - 1]
#define U(i) st_.u[(i) - 1]
#define NBD(i) st_.nbd[(i) - 1]
#define Z(i) st_.z[(i) - 1]
#define R(i) st_.r[(i) - 1]
#define D(i) st_.d[(i) - 1]
#define TT(i) st_.t[(i) - 1]
#define INDEX(i) st_.index[(i) - 1]
#define IWHERE(i) st_.iwhere[(i) - 1]
#define INDX2(i) st_.indx2[(i) - 1]
#define WS(i, j) st_.ws[((i) - 1) + (size_t)((j) - 1) * n_]
#define WY(i, j) st_.wy[((i) - 1) + (size_t)((j) - 1) * n_]
#define SY(i, j) st_.sy[((i) - 1) + (size_t)((j) - 1) * m_]
#define SS(i, j) st_.ss[((i) - 1) + (size_t)((j) - 1) * m_]
#define WT(i, j) st_.wt[((i) - 1) + (size_t)((j) - 1) * m_]
#define WN(i, j) st_.wn[((i) - 1) + (size_t)((j) - 1) * 2 * m_]
#define WN1(i, j) st_.snd[((i) - 1) + (size_t)((j) - 1) * 2 * m_]

template <class Store>
NGHMM_HD void LbfgsbT<Store>::refresh_memory() {  // bfgs.cpp: the repeated "refresh the lbfgs memory" blocks
  info_ = 0;
  col_ = 0;
  head_ = 1;
  theta_ = 1.0;
  iupdat_ = 0;
  updatd_ = false;
}

// errclb (bfgs.cpp:2309-2380): input checks.  Returns false on error.
template <class Store>
NGHMM_HD bool LbfgsbT<Store>::errclb() {
  bool ok = true;
  if (n_ <= 0) ok = false;
  if (m_ <= 0) ok = false;
  if (factr_ < 0.0) ok = false;
  for (int i = 1; i <= n_; ++i) {
    if (NBD(i) < 0 || NBD(i) > 3) {
      ok = false;
      info_ = -6;
    }
    if (NBD(i) == 2 && L(i) > U(i)) {
      ok = false;
      info_ = -7;
    }
  }
  return ok;
}

// active (bfgs.cpp:1269-1400): project x into the box, classify variables.
template <class Store>
NGHMM_HD void LbfgsbT<Store>::active() {
  prjctd_ = false;
  cnstnd_ = false;
  boxed_ = true;
  for (int i = 1; i <= n_; ++i) {
    if (NBD(i) > 0) {
      if (NBD(i) <= 2 && X(i) <= L(i)) {
        if (X(i) < L(i)) {
          prjctd_ = true;
          X(i) = L(i);
        }
      } else if (NBD(i) >= 2 && X(i) >= U(i)) {
        if (X(i) > U(i)) {
          prjctd_ = true;
          X(i) = U(i);
        }
      }
    }
  }
  for (int i = 1; i <= n_; ++i) {
    if (NBD(i) != 2) boxed_ = false;
    if (NBD(i) == 0) {
      IWHERE(i) = -1;
    } else {
      cnstnd_ = true;
      if (NBD(i) == 2 && U(i) - L(i) <= 0.0)
        IWHERE(i) = 3;
      else
        IWHERE(i) = 0;
    }
  }
}

// projgr (bfgs.cpp:3999-4060): infinity norm of the projected gradient.
template <class Store>
NGHMM_HD void LbfgsbT<Store>::projgr(const double* g) {
  sbgnrm_ = 0.0;
  for (int i = 1; i <= n_; ++i) {
    double gi = g[i - 1];
    if (NBD(i) != 0) {
      if (gi < 0.0) {
        if (NBD(i) >= 2) gi = lbfgsb_detail::maxd(X(i) - U(i), gi);
      } else {
        if (NBD(i) <= 2) gi = lbfgsb_detail::mind(X(i) - L(i), gi);
      }
    }
    sbgnrm_ = lbfgsb_detail::maxd(sbgnrm_, lbfgsb_detail::absd(gi));
  }
}

// bmv (bfgs.cpp:1402-1550): product of the 2m x 2m middle matrix with a vector.
template <class Store>
NGHMM_HD void LbfgsbT<Store>::bmv(const double* v1, double* p1, bool& ok) {
  ok = true;
  const int col = col_;
  if (col == 0) return;
  const double* v = v1 - 1;
  double* p = p1 - 1;
  p[col + 1] = v[col + 1];
  for (int i = 2; i <= col; ++i) {
    int i2 = col + i;
    double sum = 0.0;
    for (int k = 1; k <= i - 1; ++k) sum += SY(i, k) * v[k] / SY(k, k);
    p[i2] = v[i2] + sum;
  }
  if (lbfgsb_detail::tri_solve(&WT(1, 1), m_, col, &p[col + 1], 11) != 0) {
    ok = false;
    return;
  }
  for (int i = 1; i <= col; ++i) p[i] = v[i] / std::sqrt(SY(i, i));
  if (lbfgsb_detail::tri_solve(&WT(1, 1), m_, col, &p[col + 1], 1) != 0) {
    ok = false;
    return;
  }
  for (int i = 1; i <= col; ++i) p[i] = -p[i] / std::sqrt(SY(i, i));
  for (int i = 1; i <= col; ++i) {
    double sum = 0.0;
    for (int k = i + 1; k <= col; ++k) sum += SY(k, i) * p[col + k] / SY(i, i);
    p[i] += sum;
  }
}

// cauchy (bfgs.cpp:1553-2200): generalized Cauchy point along the projected
// steepest-descent path.  z_ receives the point, wa_[2m..4m) the vector c.
template <class Store>
NGHMM_HD void LbfgsbT<Store>::cauchy(const double* g1, bool& ok) {
  ok = true;
  const double* g = g1 - 1;
  const int n = n_, m = m_, col = col_;
  double* p = st_.wa - 1;            // wa(1 .. 2m)
  double* c = st_.wa + 2 * m - 1;    // wa(2m+1 .. 4m)
  double* wbp = st_.wa + 4 * m - 1;  // wa(4m+1 .. 6m)
  double* v = st_.wa + 6 * m - 1;    // wa(6m+1 .. 8m)
  double* xcp = st_.z - 1;
  int* iorder = st_.indx2 - 1;

  if (sbgnrm_ <= 0.0) {
    for (int i = 1; i <= n; ++i) xcp[i] = X(i);
    return;
  }
  bool bnded = true;
  int nfree = n + 1;
  int nbreak = 0;
  int ibkmin = 0;
  double bkmin = 0.0;
  const int col2 = 2 * col;
  double f1 = 0.0;
  double tl = 0.0, tu = 0.0;
  for (int i = 1; i <= col2; ++i) p[i] = 0.0;

  for (int i = 1; i <= n; ++i) {
    double neggi = -g[i];
    if (IWHERE(i) != 3 && IWHERE(i) != -1) {
      if (NBD(i) <= 2) tl = X(i) - L(i);
      if (NBD(i) >= 2) tu = U(i) - X(i);
      bool xlower = NBD(i) <= 2 && tl <= 0.0;
      bool xupper = NBD(i) >= 2 && tu <= 0.0;
      IWHERE(i) = 0;
      if (xlower) {
        if (neggi <= 0.0) IWHERE(i) = 1;
      } else if (xupper) {
        if (neggi >= 0.0) IWHERE(i) = 2;
      } else {
        if (lbfgsb_detail::absd(neggi) <= 0.0) IWHERE(i) = -3;
      }
    }
    int pointr = head_;
    if (IWHERE(i) != 0 && IWHERE(i) != -1) {
      D(i) = 0.0;
    } else {
      D(i) = neggi;
      f1 -= neggi * neggi;
      for (int j = 1; j <= col; ++j) {
        p[j] += WY(i, pointr) * neggi;
        p[col + j] += WS(i, pointr) * neggi;
        pointr = pointr % m + 1;
      }
      if (NBD(i) <= 2 && NBD(i) != 0 && neggi < 0.0) {
        ++nbreak;
        iorder[nbreak] = i;
        TT(nbreak) = tl / (-neggi);
        if (nbreak == 1 || TT(nbreak) < bkmin) {
          bkmin = TT(nbreak);
          ibkmin = nbreak;
        }
      } else if (NBD(i) >= 2 && neggi > 0.0) {
        ++nbreak;
        iorder[nbreak] = i;
        TT(nbreak) = tu / neggi;
        if (nbreak == 1 || TT(nbreak) < bkmin) {
          bkmin = TT(nbreak);
          ibkmin = nbreak;
        }
      } else {
        --nfree;
        iorder[nfree] = i;
        if (lbfgsb_detail::absd(neggi) > 0.0) bnded = false;
      }
    }
  }

  if (theta_ != 1.0)
    for (int j = 1; j <= col; ++j) p[col + j] = theta_ * p[col + j];

  for (int i = 1; i <= n; ++i) xcp[i] = X(i);
  if (nbreak == 0 && nfree == n + 1) return;

  for (int j = 1; j <= col2; ++j) c[j] = 0.0;

  double f2 = -theta_ * f1;
  if (col > 0) {
    bmv(&p[1], &v[1], ok);
    if (!ok) return;
    f2 -= lbfgsb_detail::dot(col2, &v[1], &p[1]);
  }
  double dtm = -f1 / f2;
  double tsum = 0.0;
  nint_ = 1;

  bool finish_segment = true;  // L888 unless the all-variables-fixed exit (L999) is taken
  if (nbreak != 0) {
    int nleft = nbreak;
    int iter = 1;
    double tj = 0.0;
    for (;;) {  // L777
      double tj0 = tj;
      int ibp;
      if (iter == 1) {
        tj = bkmin;
        ibp = iorder[ibkmin];
      } else {
        if (iter == 2) {
          if (ibkmin != nbreak) {
            TT(ibkmin) = TT(nbreak);
            iorder[ibkmin] = iorder[nbreak];
          }
        }
        lbfgsb_detail::heap_pop_min(nleft, &TT(1), &iorder[1], iter - 2 != 0);
        tj = TT(nleft);
        ibp = iorder[nleft];
      }
      double dt = tj - tj0;
      if (dtm < dt) break;  // -> L888
      tsum += dt;
      --nleft;
      ++iter;
      double dibp = D(ibp);
      D(ibp) = 0.0;
      double zibp;
      if (dibp > 0.0) {
        zibp = U(ibp) - X(ibp);
        xcp[ibp] = U(ibp);
        IWHERE(ibp) = 2;
      } else {
        zibp = L(ibp) - X(ibp);
        xcp[ibp] = L(ibp);
        IWHERE(ibp) = 1;
      }
      if (nleft == 0 && nbreak == n) {
        dtm = dt;
        finish_segment = false;  // -> L999
        break;
      }
      ++nint_;
      double dibp2 = dibp * dibp;
      f1 = f1 + dt * f2 + dibp2 - theta_ * dibp * zibp;
      f2 -= theta_ * dibp2;
      if (col > 0) {
        lbfgsb_detail::axpy(col2, dt, &p[1], &c[1]);
        int pointr = head_;
        for (int j = 1; j <= col; ++j) {
          wbp[j] = WY(ibp, pointr);
          wbp[col + j] = theta_ * WS(ibp, pointr);
          pointr = pointr % m + 1;
        }
        bmv(&wbp[1], &v[1], ok);
        if (!ok) return;
        double wmc = lbfgsb_detail::dot(col2, &c[1], &v[1]);
        double wmp = lbfgsb_detail::dot(col2, &p[1], &v[1]);
        double wmw = lbfgsb_detail::dot(col2, &wbp[1], &v[1]);
        lbfgsb_detail::axpy(col2, -dibp, &wbp[1], &p[1]);
        f1 += dibp * wmc;
        f2 = f2 + dibp * 2.0 * wmp - dibp2 * wmw;
      }
      if (nleft > 0) {
        dtm = -f1 / f2;
        continue;
      } else if (bnded) {
        f1 = 0.0;
        f2 = 0.0;
        dtm = 0.0;
      } else {
        dtm = -f1 / f2;
      }
      break;  // -> L888
    }
  }
  if (finish_segment) {  // L888
    if (dtm <= 0.0) dtm = 0.0;
    tsum += dtm;
    lbfgsb_detail::axpy(n, tsum, &D(1), &xcp[1]);
  }
  // L999
  if (col > 0) lbfgsb_detail::axpy(col2, dtm, &p[1], &c[1]);
}

// freev (bfgs.cpp:2871-3015): entering/leaving variables and the free set at the GCP.
template <class Store>
NGHMM_HD void LbfgsbT<Store>::freev(bool& wrk) {
  const int n = n_;
  nenter_ = 0;
  ileave_ = n + 1;
  if (iter_ > 0 && cnstnd_) {
    for (int i = 1; i <= nfree_; ++i) {
      int k = INDEX(i);
      if (IWHERE(k) > 0) {
        --ileave_;
        INDX2(ileave_) = k;
      }
    }
    for (int i = nfree_ + 1; i <= n; ++i) {
      int k = INDEX(i);
      if (IWHERE(k) <= 0) {
        ++nenter_;
        INDX2(nenter_) = k;
      }
    }
  }
  wrk = ileave_ < n + 1 || nenter_ > 0 || updatd_;
  nfree_ = 0;
  int iact = n + 1;
  for (int i = 1; i <= n; ++i) {
    if (IWHERE(i) <= 0) {
      ++nfree_;
      INDEX(nfree_) = i;
    } else {
      --iact;
      INDEX(iact) = i;
    }
  }
}

// formk (bfgs.cpp:2389-2780): LEL^T factorisation of the indefinite matrix K
// of the subspace problem.
template <class Store>
NGHMM_HD void LbfgsbT<Store>::formk(bool& ok) {
  ok = true;
  const int n = n_, m = m_, col = col_, nsub = nfree_;
  int upcl;
  if (updatd_) {
    if (iupdat_ > m) {
      for (int jy = 1; jy <= m - 1; ++jy) {
        int js = m + jy;
        for (int q = 0; q < m - jy; ++q) WN1(jy + q, jy) = WN1(jy + 1 + q, jy + 1);
        for (int q = 0; q < m - jy; ++q) WN1(js + q, js) = WN1(js + 1 + q, js + 1);
        for (int q = 0; q < m - 1; ++q) WN1(m + 1 + q, jy) = WN1(m + 2 + q, jy + 1);
      }
    }
    const int pbegin = 1, pend = nsub, dbegin = nsub + 1, dend = n;
    int iy = col;
    int is = m + col;
    int ipntr = head_ + col - 1;
    if (ipntr > m) ipntr -= m;
    int jpntr = head_;
    for (int jy = 1; jy <= col; ++jy) {
      int js = m + jy;
      double temp1 = 0.0, temp2 = 0.0, temp3 = 0.0;
      for (int k = pbegin; k <= pend; ++k) {
        int k1 = INDEX(k);
        temp1 += WY(k1, ipntr) * WY(k1, jpntr);
      }
      for (int k = dbegin; k <= dend; ++k) {
        int k1 = INDEX(k);
        temp2 += WS(k1, ipntr) * WS(k1, jpntr);
        temp3 += WS(k1, ipntr) * WY(k1, jpntr);
      }
      WN1(iy, jy) = temp1;
      WN1(is, js) = temp2;
      WN1(is, jy) = temp3;
      jpntr = jpntr % m + 1;
    }
    int jy = col;
    jpntr = head_ + col - 1;
    if (jpntr > m) jpntr -= m;
    ipntr = head_;
    for (int i = 1; i <= col; ++i) {
      is = m + i;
      double temp3 = 0.0;
      for (int k = pbegin; k <= pend; ++k) {
        int k1 = INDEX(k);
        temp3 += WS(k1, ipntr) * WY(k1, jpntr);
      }
      ipntr = ipntr % m + 1;
      WN1(is, jy) = temp3;
    }
    upcl = col - 1;
  } else {
    upcl = col;
  }

  int ipntr = head_;
  for (int iy = 1; iy <= upcl; ++iy) {
    int is = m + iy;
    int jpntr = head_;
    for (int jy = 1; jy <= iy; ++jy) {
      int js = m + jy;
      double temp1 = 0.0, temp2 = 0.0, temp3 = 0.0, temp4 = 0.0;
      for (int k = 1; k <= nenter_; ++k) {
        int k1 = INDX2(k);
        temp1 += WY(k1, ipntr) * WY(k1, jpntr);
        temp2 += WS(k1, ipntr) * WS(k1, jpntr);
      }
      for (int k = ileave_; k <= n; ++k) {
        int k1 = INDX2(k);
        temp3 += WY(k1, ipntr) * WY(k1, jpntr);
        temp4 += WS(k1, ipntr) * WS(k1, jpntr);
      }
      WN1(iy, jy) = WN1(iy, jy) + temp1 - temp3;
      WN1(is, js) = WN1(is, js) - temp2 + temp4;
      jpntr = jpntr % m + 1;
    }
    ipntr = ipntr % m + 1;
  }
  ipntr = head_;
  for (int is = m + 1; is <= m + upcl; ++is) {
    int jpntr = head_;
    for (int jy = 1; jy <= upcl; ++jy) {
      double temp1 = 0.0, temp3 = 0.0;
      for (int k = 1; k <= nenter_; ++k) {
        int k1 = INDX2(k);
        temp1 += WS(k1, ipntr) * WY(k1, jpntr);
      }
      for (int k = ileave_; k <= n; ++k) {
        int k1 = INDX2(k);
        temp3 += WS(k1, ipntr) * WY(k1, jpntr);
      }
      if (is <= jy + m)
        WN1(is, jy) = WN1(is, jy) + temp1 - temp3;
      else
        WN1(is, jy) = WN1(is, jy) - temp1 + temp3;
      jpntr = jpntr % m + 1;
    }
    ipntr = ipntr % m + 1;
  }

  const int m2 = 2 * m;
  for (int iy = 1; iy <= col; ++iy) {
    int is = col + iy;
    int is1 = m + iy;
    for (int jy = 1; jy <= iy; ++jy) {
      int js = col + jy;
      int js1 = m + jy;
      WN(jy, iy) = WN1(iy, jy) / theta_;
      WN(js, is) = WN1(is1, js1) * theta_;
    }
    for (int jy = 1; jy <= iy - 1; ++jy) WN(jy, is) = -WN1(is1, jy);
    for (int jy = iy; jy <= col; ++jy) WN(jy, is) = WN1(is1, jy);
    WN(iy, iy) += SY(iy, iy);
  }
  if (lbfgsb_detail::cholesky_upper(&WN(1, 1), m2, col) != 0) {
    info_ = -1;
    ok = false;
    return;
  }
  const int col2 = 2 * col;
  for (int js = col + 1; js <= col2; ++js) lbfgsb_detail::tri_solve(&WN(1, 1), m2, col, &WN(1, js), 11);
  for (int is = col + 1; is <= col2; ++is)
    for (int js = is; js <= col2; ++js) WN(is, js) += lbfgsb_detail::dot(col, &WN(1, is), &WN(1, js));
  if (lbfgsb_detail::cholesky_upper(&WN(col + 1, col + 1), m2, col) != 0) {
    info_ = -2;
    ok = false;
    return;
  }
}

// cmprlb (bfgs.cpp:2206-2305): r = -Z'(B(xcp - x) + g).
template <class Store>
NGHMM_HD void LbfgsbT<Store>::cmprlb(const double* g1, bool& ok) {
  ok = true;
  const double* g = g1 - 1;
  const int n = n_, m = m_, col = col_;
  double* wa = st_.wa - 1;
  if (!cnstnd_ && col > 0) {
    for (int i = 1; i <= n; ++i) R(i) = -g[i];
  } else {
    for (int i = 1; i <= nfree_; ++i) {
      int k = INDEX(i);
      R(i) = -theta_ * (Z(k) - X(k)) - g[k];
    }
    bmv(&wa[2 * m + 1], &wa[1], ok);
    if (!ok) {
      info_ = -8;
      return;
    }
    int pointr = head_;
    for (int j = 1; j <= col; ++j) {
      double a1 = wa[j];
      double a2 = theta_ * wa[col + j];
      for (int i = 1; i <= nfree_; ++i) {
        int k = INDEX(i);
        R(i) = R(i) + WY(k, pointr) * a1 + WS(k, pointr) * a2;
      }
      pointr = pointr % m + 1;
    }
  }
}

// subsm (bfgs.cpp:4068-4425): subspace minimisation over the free variables,
// then backtrack into the box.  Works on z_ (the Cauchy point) and r_.
template <class Store>
NGHMM_HD void LbfgsbT<Store>::subsm(bool& ok) {
  ok = true;
  const int m = m_, col = col_, nsub = nfree_;
  if (nsub <= 0) return;
  double* wv = st_.wa - 1;
  int pointr = head_;
  for (int i = 1; i <= col; ++i) {
    double temp1 = 0.0, temp2 = 0.0;
    for (int j = 1; j <= nsub; ++j) {
      int k = INDEX(j);
      temp1 += WY(k, pointr) * R(j);
      temp2 += WS(k, pointr) * R(j);
    }
    wv[i] = temp1;
    wv[col + i] = theta_ * temp2;
    pointr = pointr % m + 1;
  }
  const int m2 = 2 * m, col2 = 2 * col;
  if (lbfgsb_detail::tri_solve(&WN(1, 1), m2, col2, &wv[1], 11) != 0) {
    info_ = 1;
    ok = false;
    return;
  }
  for (int i = 1; i <= col; ++i) wv[i] = -wv[i];
  if (lbfgsb_detail::tri_solve(&WN(1, 1), m2, col2, &wv[1], 1) != 0) {
    info_ = 1;
    ok = false;
    return;
  }
  pointr = head_;
  for (int jy = 1; jy <= col; ++jy) {
    int js = col + jy;
    for (int i = 1; i <= nsub; ++i) {
      int k = INDEX(i);
      R(i) = R(i) + WY(k, pointr) * wv[jy] / theta_ + WS(k, pointr) * wv[js];
    }
    pointr = pointr % m + 1;
  }
  for (int i = 1; i <= nsub; ++i) R(i) /= theta_;

  double alpha = 1.0;
  double temp1 = alpha;
  int ibd = 0;
  for (int i = 1; i <= nsub; ++i) {
    int k = INDEX(i);
    double dk = R(i);
    if (NBD(k) != 0) {
      if (dk < 0.0 && NBD(k) <= 2) {
        double temp2 = L(k) - Z(k);
        if (temp2 >= 0.0)
          temp1 = 0.0;
        else if (dk * alpha < temp2)
          temp1 = temp2 / dk;
      } else if (dk > 0.0 && NBD(k) >= 2) {
        double temp2 = U(k) - Z(k);
        if (temp2 <= 0.0)
          temp1 = 0.0;
        else if (dk * alpha > temp2)
          temp1 = temp2 / dk;
      }
      if (temp1 < alpha) {
        alpha = temp1;
        ibd = i;
      }
    }
  }
  if (alpha < 1.0) {
    double dk = R(ibd);
    int k = INDEX(ibd);
    if (dk > 0.0) {
      Z(k) = U(k);
      R(ibd) = 0.0;
    } else if (dk < 0.0) {
      Z(k) = L(k);
      R(ibd) = 0.0;
    }
  }
  for (int i = 1; i <= nsub; ++i) {
    int k = INDEX(i);
    Z(k) += alpha * R(i);
  }
  iword_ = alpha < 1.0 ? 1 : 0;
}

// dcstep (bfgs.cpp:4772-5050): safeguarded cubic/quadratic step of More'-Thuente.
template <class Store>
NGHMM_HD void LbfgsbT<Store>::dcstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy,
                    double& stp, double fp, double dp, bool& brackt, double stpmin,
                    double stpmax) {
  double stpf, stpc, stpq, theta, s, gamma, p, q, r;
  const double sgnd = dp * (dx / lbfgsb_detail::absd(dx));
  if (fp > fx) {
    theta = (fx - fp) * 3.0 / (stp - stx) + dx + dp;
    s = lbfgsb_detail::maxd(lbfgsb_detail::maxd(lbfgsb_detail::absd(theta), lbfgsb_detail::absd(dx)), lbfgsb_detail::absd(dp));
    double a = theta / s;
    gamma = s * std::sqrt(a * a - dx / s * (dp / s));
    if (stp < stx) gamma = -gamma;
    p = gamma - dx + theta;
    q = gamma - dx + gamma + dp;
    r = p / q;
    stpc = stx + r * (stp - stx);
    stpq = stx + dx / ((fx - fp) / (stp - stx) + dx) / 2.0 * (stp - stx);
    if (lbfgsb_detail::absd(stpc - stx) < lbfgsb_detail::absd(stpq - stx))
      stpf = stpc;
    else
      stpf = stpc + (stpq - stpc) / 2.0;
    brackt = true;
  } else if (sgnd < 0.0) {
    theta = (fx - fp) * 3.0 / (stp - stx) + dx + dp;
    s = lbfgsb_detail::maxd(lbfgsb_detail::maxd(lbfgsb_detail::absd(theta), lbfgsb_detail::absd(dx)), lbfgsb_detail::absd(dp));
    double a = theta / s;
    gamma = s * std::sqrt(a * a - dx / s * (dp / s));
    if (stp > stx) gamma = -gamma;
    p = gamma - dp + theta;
    q = gamma - dp + gamma + dx;
    r = p / q;
    stpc = stp + r * (stx - stp);
    stpq = stp + dp / (dp - dx) * (stx - stp);
    if (lbfgsb_detail::absd(stpc - stp) > lbfgsb_detail::absd(stpq - stp))
      stpf = stpc;
    else
      stpf = stpq;
    brackt = true;
  } else if (lbfgsb_detail::absd(dp) < lbfgsb_detail::absd(dx)) {
    theta = (fx - fp) * 3.0 / (stp - stx) + dx + dp;
    s = lbfgsb_detail::maxd(lbfgsb_detail::maxd(lbfgsb_detail::absd(theta), lbfgsb_detail::absd(dx)), lbfgsb_detail::absd(dp));
    double a = theta / s;
    gamma = s * std::sqrt(lbfgsb_detail::maxd(0.0, a * a - dx / s * (dp / s)));
    if (stp > stx) gamma = -gamma;
    p = gamma - dp + theta;
    q = gamma + (dx - dp) + gamma;
    r = p / q;
    if (r < 0.0 && gamma != 0.0)
      stpc = stp + r * (stx - stp);
    else if (stp > stx)
      stpc = stpmax;
    else
      stpc = stpmin;
    stpq = stp + dp / (dp - dx) * (stx - stp);
    if (brackt) {
      if (lbfgsb_detail::absd(stpc - stp) < lbfgsb_detail::absd(stpq - stp))
        stpf = stpc;
      else
        stpf = stpq;
      if (stp > stx)
        stpf = lbfgsb_detail::mind(stp + (sty - stp) * 0.66, stpf);
      else
        stpf = lbfgsb_detail::maxd(stp + (sty - stp) * 0.66, stpf);
    } else {
      if (lbfgsb_detail::absd(stpc - stp) > lbfgsb_detail::absd(stpq - stp))
        stpf = stpc;
      else
        stpf = stpq;
      stpf = lbfgsb_detail::mind(stpmax, stpf);
      stpf = lbfgsb_detail::maxd(stpmin, stpf);
    }
  } else {
    if (brackt) {
      theta = (fp - fy) * 3.0 / (sty - stp) + dy + dp;
      s = lbfgsb_detail::maxd(lbfgsb_detail::maxd(lbfgsb_detail::absd(theta), lbfgsb_detail::absd(dy)), lbfgsb_detail::absd(dp));
      double a = theta / s;
      gamma = s * std::sqrt(a * a - dy / s * (dp / s));
      if (stp > sty) gamma = -gamma;
      p = gamma - dp + theta;
      q = gamma - dp + gamma + dy;
      r = p / q;
      stpc = stp + r * (sty - stp);
      stpf = stpc;
    } else if (stp > stx) {
      stpf = stpmax;
    } else {
      stpf = stpmin;
    }
  }
  if (fp > fx) {
    sty = stp;
    fy = fp;
    dy = dp;
  } else {
    if (sgnd < 0.0) {
      sty = stx;
      fy = fx;
      dy = dx;
    }
    stx = stp;
    fx = fp;
    dx = dp;
  }
  stp = stpf;
}

// dcsrch (bfgs.cpp:4429-4770): More'-Thuente line search, reverse communication.
// ftol 1e-3, gtol 0.9, xtol 0.1, stpmin 0 (bfgs.cpp:165-167, lnsrlb's call).
template <class Store>
NGHMM_HD void LbfgsbT<Store>::dcsrch(double f, double g, double& stp, double stpmax) {
  const double ftol = 1e-3, gtol = 0.9, xtol = 0.1, stpmin = 0.0;
  LbfgsbLsState s;
  if (ls_task_ == LbfgsbLs::Start) {
    bool err = false;
    if (stp < stpmin) err = true;
    if (stp > stpmax) err = true;
    if (g >= 0.0) err = true;
    if (stpmax < stpmin) err = true;
    if (err) {  // returns before anything is saved (bfgs.cpp: "ERROR" early return)
      ls_task_ = LbfgsbLs::Error;
      return;
    }
    s.brackt = false;
    s.stage = 1;
    s.finit = f;
    s.ginit = g;
    s.gtest = ftol * s.ginit;
    s.width = stpmax - stpmin;
    s.width1 = s.width / 0.5;
    s.stx = 0.0;
    s.fx = s.finit;
    s.gx = s.ginit;
    s.sty = 0.0;
    s.fy = s.finit;
    s.gy = s.ginit;
    s.stmin = 0.0;
    s.stmax = stp + stp * 4.0;
    ls_task_ = LbfgsbLs::FG;
    ls_ = s;
    return;
  }
  s = ls_;
  const double ftest = s.finit + stp * s.gtest;
  if (s.stage == 1 && f <= ftest && g >= 0.0) s.stage = 2;
  if (s.brackt && (stp <= s.stmin || stp >= s.stmax)) ls_task_ = LbfgsbLs::Warning;
  if (s.brackt && s.stmax - s.stmin <= xtol * s.stmax) ls_task_ = LbfgsbLs::Warning;
  if (stp == stpmax && f <= ftest && g <= s.gtest) ls_task_ = LbfgsbLs::Warning;
  if (stp == stpmin && (f > ftest || g >= s.gtest)) ls_task_ = LbfgsbLs::Warning;
  if (f <= ftest && lbfgsb_detail::absd(g) <= gtol * (-s.ginit)) ls_task_ = LbfgsbLs::Convergence;
  if (ls_task_ == LbfgsbLs::Warning || ls_task_ == LbfgsbLs::Convergence) {
    ls_ = s;
    return;
  }
  if (s.stage == 1 && f <= s.fx && f > ftest) {
    double fm = f - stp * s.gtest;
    double fxm = s.fx - s.stx * s.gtest;
    double fym = s.fy - s.sty * s.gtest;
    double gm = g - s.gtest;
    double gxm = s.gx - s.gtest;
    double gym = s.gy - s.gtest;
    dcstep(s.stx, fxm, gxm, s.sty, fym, gym, stp, fm, gm, s.brackt, s.stmin, s.stmax);
    s.fx = fxm + s.stx * s.gtest;
    s.fy = fym + s.sty * s.gtest;
    s.gx = gxm + s.gtest;
    s.gy = gym + s.gtest;
  } else {
    dcstep(s.stx, s.fx, s.gx, s.sty, s.fy, s.gy, stp, f, g, s.brackt, s.stmin, s.stmax);
  }
  if (s.brackt) {
    if (lbfgsb_detail::absd(s.sty - s.stx) >= s.width1 * 0.66) stp = s.stx + (s.sty - s.stx) * 0.5;
    s.width1 = s.width;
    s.width = lbfgsb_detail::absd(s.sty - s.stx);
  }
  if (s.brackt) {
    s.stmin = lbfgsb_detail::mind(s.stx, s.sty);
    s.stmax = lbfgsb_detail::maxd(s.stx, s.sty);
  } else {
    s.stmin = stp + (stp - s.stx) * 1.1;
    s.stmax = stp + (stp - s.stx) * 4.0;
  }
  stp = lbfgsb_detail::maxd(stp, stpmin);
  stp = lbfgsb_detail::mind(stp, stpmax);
  if ((s.brackt && (stp <= s.stmin || stp >= s.stmax)) ||
      (s.brackt && s.stmax - s.stmin <= xtol * s.stmax))
    stp = s.stx;
  ls_task_ = LbfgsbLs::FG;
  ls_ = s;
}

// lnsrlb (bfgs.cpp:3135-3290).  Returns true when f,g are wanted at the new x.
template <class Store>
NGHMM_HD bool LbfgsbT<Store>::lnsrlb(double* f, double* g1, bool fresh) {
  const int n = n_;
  double* g = g1 - 1;
  if (fresh) {
    dtd_ = lbfgsb_detail::dot(n, &D(1), &D(1));
    dnorm_ = std::sqrt(dtd_);
    stpmx_ = 1e10;
    if (cnstnd_) {
      if (iter_ == 0) {
        stpmx_ = 1.0;
      } else {
        for (int i = 1; i <= n; ++i) {
          double a1 = D(i);
          if (NBD(i) != 0) {
            if (a1 < 0.0 && NBD(i) <= 2) {
              double a2 = L(i) - X(i);
              if (a2 >= 0.0)
                stpmx_ = 0.0;
              else if (a1 * stpmx_ < a2)
                stpmx_ = a2 / a1;
            } else if (a1 > 0.0 && NBD(i) >= 2) {
              double a2 = U(i) - X(i);
              if (a2 <= 0.0)
                stpmx_ = 0.0;
              else if (a1 * stpmx_ > a2)
                stpmx_ = a2 / a1;
            }
          }
        }
      }
    }
    if (iter_ == 0 && !boxed_)
      stp_ = lbfgsb_detail::mind(1.0 / dnorm_, stpmx_);
    else
      stp_ = 1.0;
    for (int i = 1; i <= n; ++i) TT(i) = X(i);
    for (int i = 1; i <= n; ++i) R(i) = g[i];
    fold_ = *f;
    ifun_ = 0;
    iback_ = 0;
    ls_task_ = LbfgsbLs::Start;
  }
  gd_ = lbfgsb_detail::dot(n, &g[1], &D(1));
  if (ifun_ == 0) {
    gdold_ = gd_;
    if (gd_ >= 0.0) {
      info_ = -4;
      return false;
    }
  }
  dcsrch(*f, gd_, stp_, stpmx_);
  xstep_ = stp_ * dnorm_;
  if (ls_task_ != LbfgsbLs::Convergence && ls_task_ != LbfgsbLs::Warning) {
    ++ifun_;
    ++nfgv_;
    iback_ = ifun_ - 1;
    if (stp_ == 1.0) {
      for (int i = 1; i <= n; ++i) X(i) = Z(i);
    } else {
      for (int i = 1; i <= n; ++i) X(i) = stp_ * D(i) + TT(i);
    }
    return true;
  }
  return false;
}

// matupd (bfgs.cpp:3294-3400): append the new (s, y) pair and update SY, SS.
template <class Store>
NGHMM_HD void LbfgsbT<Store>::matupd(double rr, double dr) {
  const int n = n_, m = m_;
  if (iupdat_ <= m) {
    col_ = iupdat_;
    itail_ = (head_ + iupdat_ - 2) % m + 1;
  } else {
    itail_ = itail_ % m + 1;
    head_ = head_ % m + 1;
  }
  for (int i = 1; i <= n; ++i) WS(i, itail_) = D(i);
  for (int i = 1; i <= n; ++i) WY(i, itail_) = R(i);
  theta_ = rr / dr;
  const int col = col_;
  if (iupdat_ > m) {
    for (int j = 1; j <= col - 1; ++j) {
      for (int q = 0; q < j; ++q) SS(1 + q, j) = SS(2 + q, j + 1);
      for (int q = 0; q < col - j; ++q) SY(j + q, j) = SY(j + 1 + q, j + 1);
    }
  }
  int pointr = head_;
  for (int j = 1; j <= col - 1; ++j) {
    SY(col, j) = lbfgsb_detail::dot(n, &D(1), &WY(1, pointr));
    SS(j, col) = lbfgsb_detail::dot(n, &WS(1, pointr), &D(1));
    pointr = pointr % m + 1;
  }
  if (stp_ == 1.0)
    SS(col, col) = dtd_;
  else
    SS(col, col) = stp_ * stp_ * dtd_;
  SY(col, col) = dr;
}

// formt (bfgs.cpp:2782-2868): T = theta*SS + L*D^-1*L', Cholesky-factored.
template <class Store>
NGHMM_HD void LbfgsbT<Store>::formt(bool& ok) {
  ok = true;
  const int col = col_;
  for (int j = 1; j <= col; ++j) WT(1, j) = theta_ * SS(1, j);
  for (int i = 2; i <= col; ++i) {
    for (int j = i; j <= col; ++j) {
      int k1 = (i <= j ? i : j) - 1;
      double ddum = 0.0;
      for (int k = 1; k <= k1; ++k) ddum += SY(i, k) * SY(j, k) / SY(k, k);
      WT(i, j) = ddum + theta_ * SS(i, j);
    }
  }
  if (lbfgsb_detail::cholesky_upper(&WT(1, 1), m_, col) != 0) {
    info_ = -3;
    ok = false;
  }
}

// mainlb (bfgs.cpp:440-1265) as a resumable state machine.
template <class Store>
NGHMM_HD LbfgsbTask LbfgsbT<Store>::advance(double* f, double* g) {
  enum { L222, L333, L555, L666_FRESH, L666_RESUME, L777 } at;
  bool wrk = false;
  bool ok = true;
  const int n = n_;

  switch (phase_) {
    case LbfgsbPhase::Start: {
      epsmch_ = Store::machine_eps();
      col_ = 0;
      head_ = 1;
      theta_ = 1.0;
      iupdat_ = 0;
      updatd_ = false;
      iter_ = 0;
      nfgv_ = 0;
      nint_ = 0;
      nintol_ = 0;
      nskip_ = 0;
      nfree_ = n;
      tol_ = factr_ * epsmch_;
      info_ = 0;
      if (!errclb()) {
        phase_ = LbfgsbPhase::Done;
        return LbfgsbTask::Error;
      }
      active();
      phase_ = LbfgsbPhase::FgStart;
      return LbfgsbTask::EvalFG;
    }
    case LbfgsbPhase::FgStart:
      nfgv_ = 1;
      projgr(g);
      if (sbgnrm_ <= pgtol_) {
        phase_ = LbfgsbPhase::Done;
        return LbfgsbTask::ConvergedPG;
      }
      at = L222;
      break;
    case LbfgsbPhase::FgLnsrch:
      at = L666_RESUME;
      break;
    case LbfgsbPhase::NewX:
      at = L777;
      break;
    default:
      return LbfgsbTask::Error;
  }

  for (;;) {
    switch (at) {
      case L222: {
        iword_ = -1;
        if (!cnstnd_ && col_ > 0) {
          for (int i = 1; i <= n; ++i) Z(i) = X(i);
          wrk = updatd_;
          nint_ = 0;
          at = L333;
          break;
        }
        cauchy(g, ok);
        if (!ok) {
          refresh_memory();
          at = L222;
          break;
        }
        nintol_ += nint_;
        freev(wrk);
        nact_ = n - nfree_;
        at = L333;
        break;
      }
      case L333: {
        if (nfree_ == 0 || col_ == 0) {
          at = L555;
          break;
        }
        if (wrk) formk(ok);
        if (info_ != 0) {
          refresh_memory();
          at = L222;
          break;
        }
        cmprlb(g, ok);
        if (info_ == 0) subsm(ok);
        if (info_ != 0) {
          refresh_memory();
          at = L222;
          break;
        }
        at = L555;
        break;
      }
      case L555:
        for (int i = 1; i <= n; ++i) D(i) = Z(i) - X(i);
        at = L666_FRESH;
        break;
      case L666_FRESH:
      case L666_RESUME: {
        bool want_fg = lnsrlb(f, g, at == L666_FRESH);
        if (info_ != 0 || iback_ >= 20) {
          for (int i = 1; i <= n; ++i) X(i) = TT(i);
          for (int i = 1; i <= n; ++i) g[i - 1] = R(i);
          *f = fold_;
          if (col_ == 0) {
            if (info_ == 0) {
              info_ = -9;
              --nfgv_;
              --ifun_;
              --iback_;
            }
            ++iter_;
            phase_ = LbfgsbPhase::Done;
            return LbfgsbTask::Abnormal;
          }
          if (info_ == 0) --nfgv_;
          refresh_memory();
          at = L222;
          break;
        }
        if (want_fg) {
          phase_ = LbfgsbPhase::FgLnsrch;
          return LbfgsbTask::EvalFG;
        }
        ++iter_;
        projgr(g);
        phase_ = LbfgsbPhase::NewX;
        return LbfgsbTask::NewX;
      }
      case L777: {
        if (sbgnrm_ <= pgtol_) {
          phase_ = LbfgsbPhase::Done;
          return LbfgsbTask::ConvergedPG;
        }
        double ddum = lbfgsb_detail::maxd(lbfgsb_detail::maxd(lbfgsb_detail::absd(fold_), lbfgsb_detail::absd(*f)), 1.0);
        if (fold_ - *f <= tol_ * ddum) {
          if (iback_ >= 10) info_ = -5;
          phase_ = LbfgsbPhase::Done;
          return LbfgsbTask::ConvergedF;
        }
        for (int i = 1; i <= n; ++i) R(i) = g[i - 1] - R(i);
        double rr = lbfgsb_detail::dot(n, &R(1), &R(1));
        double dr;
        if (stp_ == 1.0) {
          dr = gd_ - gdold_;
          ddum = -gdold_;
        } else {
          dr = (gd_ - gdold_) * stp_;
          for (int i = 1; i <= n; ++i) D(i) = stp_ * D(i);
          ddum = -gdold_ * stp_;
        }
        if (dr <= epsmch_ * ddum) {
          ++nskip_;
          updatd_ = false;
          at = L222;
          break;
        }
        updatd_ = true;
        ++iupdat_;
        matupd(rr, dr);
        formt(ok);
        if (!ok) refresh_memory();
        at = L222;
        break;
      }
    }
  }
}


#undef X
#undef L
#undef U
#undef NBD
#undef Z
#undef R
#undef D
#undef TT
#undef INDEX
#undef IWHERE
#undef INDX2
#undef WS
#undef WY
#undef SY
#undef SS
#undef WT
#undef WN
#undef WN1

}  // namespace nghmm

// lbfgsb.cpp -- see lbfgsb.hpp.  Every routine states the published L-BFGS-B 2.1
// routine it restates and where the reference keeps its f2c translation
// (shared/bfgs.cpp).  Summations run in index order and products associate left
// to right exactly as there; compile with -ffp-contract=off.
#include "lbfgsb.hpp"

#include <utility>

#include <cmath>
#include <cstring>

namespace nghmm {

namespace {

inline double absd(double v) { return v >= 0 ? v : -v; }        // bfgs.cpp:147 macro
inline double maxd(double a, double b) { return a >= b ? a : b; }  // bfgs.cpp:150 macro
inline double mind(double a, double b) { return a <= b ? a : b; }  // bfgs.cpp:149 macro

// BLAS-1 pieces the algorithm uses (bfgs.cpp:5200-5470): plain index-order loops;
// the reference's unrolled forms accumulate in the same order.
inline double dot(int n, const double* a, const double* b) {
  double acc = 0.0;
  for (int i = 0; i < n; ++i) acc += a[i] * b[i];
  return acc;
}
inline void axpy(int n, double da, const double* x, double* y) {
  if (n <= 0 || da == 0.0) return;
  for (int i = 0; i < n; ++i) y[i] += da * x[i];
}

// LINPACK dpofa: Cholesky factor of a symmetric positive definite matrix stored
// in the upper triangle (bfgs.cpp:5560-5610).  Returns 0 or the failing order.
int cholesky_upper(double* a, int lda, int n) {
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
int tri_solve(const double* t, int ldt, int n, double* b, int job) {
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
void heap_pop_min(int n, double* t1, int* iorder1, bool heap_built) {
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

// dpmeps (bfgs.cpp:5166): smallest power of the radix with 1 + eps != 1; on
// IEEE binary64 with round-to-nearest this evaluates to 2^-52.
double machine_eps() {
  volatile double a = 1.0;
  for (;;) {
    volatile double t = 1.0 + a * 0.5;
    if (t == 1.0) break;
    a = a * 0.5;
  }
  return a;
}

}  // namespace

#define X(i) x_[(i) - 1]
#define L(i) l_[(i) - 1]
#define U(i) u_[(i) - 1]
#define NBD(i) nbd_[(i) - 1]
#define Z(i) z_[(i) - 1]
#define R(i) r_[(i) - 1]
#define D(i) d_[(i) - 1]
#define TT(i) t_[(i) - 1]
#define INDEX(i) index_[(i) - 1]
#define IWHERE(i) iwhere_[(i) - 1]
#define INDX2(i) indx2_[(i) - 1]
#define WS(i, j) ws_[((i) - 1) + (size_t)((j) - 1) * n_]
#define WY(i, j) wy_[((i) - 1) + (size_t)((j) - 1) * n_]
#define SY(i, j) sy_[((i) - 1) + (size_t)((j) - 1) * m_]
#define SS(i, j) ss_[((i) - 1) + (size_t)((j) - 1) * m_]
#define WT(i, j) wt_[((i) - 1) + (size_t)((j) - 1) * m_]
#define WN(i, j) wn_[((i) - 1) + (size_t)((j) - 1) * 2 * m_]
#define WN1(i, j) snd_[((i) - 1) + (size_t)((j) - 1) * 2 * m_]

// A solver object reused for another minimisation: every scalar back to its initial value
// (as a newly constructed object), the work arrays keep their storage.  start() sizes and
// zeroes them.
void Lbfgsb::configure(int n, int m) {
  Lbfgsb fresh;
  fresh.x_.swap(x_);
  fresh.l_.swap(l_);
  fresh.u_.swap(u_);
  fresh.nbd_.swap(nbd_);
  fresh.ws_.swap(ws_);
  fresh.wy_.swap(wy_);
  fresh.sy_.swap(sy_);
  fresh.ss_.swap(ss_);
  fresh.wt_.swap(wt_);
  fresh.wn_.swap(wn_);
  fresh.snd_.swap(snd_);
  fresh.z_.swap(z_);
  fresh.r_.swap(r_);
  fresh.d_.swap(d_);
  fresh.t_.swap(t_);
  fresh.wa_.swap(wa_);
  fresh.index_.swap(index_);
  fresh.iwhere_.swap(iwhere_);
  fresh.indx2_.swap(indx2_);
  *this = std::move(fresh);
  n_ = n;
  m_ = m;
}

void Lbfgsb::reset(int n, int m) {
  n_ = n;
  m_ = m;
  x_.assign(n, 0.0);
  l_.assign(n, 0.0);
  u_.assign(n, 0.0);
  nbd_.assign(n, 0);
  // The reference calloc()s its work arrays once per findmax_bfgs call
  // (bfgs.cpp:103-105); start() re-zeroes them.
  ws_.assign((size_t)n * m, 0.0);
  wy_.assign((size_t)n * m, 0.0);
  sy_.assign((size_t)m * m, 0.0);
  ss_.assign((size_t)m * m, 0.0);
  wt_.assign((size_t)m * m, 0.0);
  wn_.assign((size_t)4 * m * m, 0.0);
  snd_.assign((size_t)4 * m * m, 0.0);
  z_.assign(n, 0.0);
  r_.assign(n, 0.0);
  d_.assign(n, 0.0);
  t_.assign(n, 0.0);
  wa_.assign((size_t)8 * m, 0.0);
  index_.assign(n, 0);
  iwhere_.assign(n, 0);
  indx2_.assign(n, 0);
  phase_ = Phase::Start;
}

void Lbfgsb::start(const double* x0, const double* l, const double* u, const int* nbd,
                   double factr, double pgtol) {
  const int n = n_, m = m_;
  reset(n, m);
  for (int i = 0; i < n; ++i) {
    x_[i] = x0[i];
    l_[i] = l[i];
    u_[i] = u[i];
    nbd_[i] = nbd ? nbd[i] : 2;
  }
  factr_ = factr;
  pgtol_ = pgtol;
  ls_ = LsState();
  ls_task_ = Ls::Start;
  phase_ = Phase::Start;
}

void Lbfgsb::refresh_memory() {  // bfgs.cpp: the repeated "refresh the lbfgs memory" blocks
  info_ = 0;
  col_ = 0;
  head_ = 1;
  theta_ = 1.0;
  iupdat_ = 0;
  updatd_ = false;
}

// errclb (bfgs.cpp:2309-2380): input checks.  Returns false on error.
bool Lbfgsb::errclb() {
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
void Lbfgsb::active() {
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
void Lbfgsb::projgr(const double* g) {
  sbgnrm_ = 0.0;
  for (int i = 1; i <= n_; ++i) {
    double gi = g[i - 1];
    if (NBD(i) != 0) {
      if (gi < 0.0) {
        if (NBD(i) >= 2) gi = maxd(X(i) - U(i), gi);
      } else {
        if (NBD(i) <= 2) gi = mind(X(i) - L(i), gi);
      }
    }
    sbgnrm_ = maxd(sbgnrm_, absd(gi));
  }
}

// bmv (bfgs.cpp:1402-1550): product of the 2m x 2m middle matrix with a vector.
void Lbfgsb::bmv(const double* v1, double* p1, bool& ok) {
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
  if (tri_solve(&WT(1, 1), m_, col, &p[col + 1], 11) != 0) {
    ok = false;
    return;
  }
  for (int i = 1; i <= col; ++i) p[i] = v[i] / std::sqrt(SY(i, i));
  if (tri_solve(&WT(1, 1), m_, col, &p[col + 1], 1) != 0) {
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
void Lbfgsb::cauchy(const double* g1, bool& ok) {
  ok = true;
  const double* g = g1 - 1;
  const int n = n_, m = m_, col = col_;
  double* p = wa_.data() - 1;            // wa(1 .. 2m)
  double* c = wa_.data() + 2 * m - 1;    // wa(2m+1 .. 4m)
  double* wbp = wa_.data() + 4 * m - 1;  // wa(4m+1 .. 6m)
  double* v = wa_.data() + 6 * m - 1;    // wa(6m+1 .. 8m)
  double* xcp = z_.data() - 1;
  int* iorder = indx2_.data() - 1;

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
        if (absd(neggi) <= 0.0) IWHERE(i) = -3;
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
        if (absd(neggi) > 0.0) bnded = false;
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
    f2 -= dot(col2, &v[1], &p[1]);
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
        heap_pop_min(nleft, &TT(1), &iorder[1], iter - 2 != 0);
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
        axpy(col2, dt, &p[1], &c[1]);
        int pointr = head_;
        for (int j = 1; j <= col; ++j) {
          wbp[j] = WY(ibp, pointr);
          wbp[col + j] = theta_ * WS(ibp, pointr);
          pointr = pointr % m + 1;
        }
        bmv(&wbp[1], &v[1], ok);
        if (!ok) return;
        double wmc = dot(col2, &c[1], &v[1]);
        double wmp = dot(col2, &p[1], &v[1]);
        double wmw = dot(col2, &wbp[1], &v[1]);
        axpy(col2, -dibp, &wbp[1], &p[1]);
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
    axpy(n, tsum, &D(1), &xcp[1]);
  }
  // L999
  if (col > 0) axpy(col2, dtm, &p[1], &c[1]);
}

// freev (bfgs.cpp:2871-3015): entering/leaving variables and the free set at the GCP.
void Lbfgsb::freev(bool& wrk) {
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
void Lbfgsb::formk(bool& ok) {
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
  if (cholesky_upper(&WN(1, 1), m2, col) != 0) {
    info_ = -1;
    ok = false;
    return;
  }
  const int col2 = 2 * col;
  for (int js = col + 1; js <= col2; ++js) tri_solve(&WN(1, 1), m2, col, &WN(1, js), 11);
  for (int is = col + 1; is <= col2; ++is)
    for (int js = is; js <= col2; ++js) WN(is, js) += dot(col, &WN(1, is), &WN(1, js));
  if (cholesky_upper(&WN(col + 1, col + 1), m2, col) != 0) {
    info_ = -2;
    ok = false;
    return;
  }
}

// cmprlb (bfgs.cpp:2206-2305): r = -Z'(B(xcp - x) + g).
void Lbfgsb::cmprlb(const double* g1, bool& ok) {
  ok = true;
  const double* g = g1 - 1;
  const int n = n_, m = m_, col = col_;
  double* wa = wa_.data() - 1;
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
void Lbfgsb::subsm(bool& ok) {
  ok = true;
  const int m = m_, col = col_, nsub = nfree_;
  if (nsub <= 0) return;
  double* wv = wa_.data() - 1;
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
  if (tri_solve(&WN(1, 1), m2, col2, &wv[1], 11) != 0) {
    info_ = 1;
    ok = false;
    return;
  }
  for (int i = 1; i <= col; ++i) wv[i] = -wv[i];
  if (tri_solve(&WN(1, 1), m2, col2, &wv[1], 1) != 0) {
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
void Lbfgsb::dcstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy,
                    double& stp, double fp, double dp, bool& brackt, double stpmin,
                    double stpmax) {
  double stpf, stpc, stpq, theta, s, gamma, p, q, r;
  const double sgnd = dp * (dx / absd(dx));
  if (fp > fx) {
    theta = (fx - fp) * 3.0 / (stp - stx) + dx + dp;
    s = maxd(maxd(absd(theta), absd(dx)), absd(dp));
    double a = theta / s;
    gamma = s * std::sqrt(a * a - dx / s * (dp / s));
    if (stp < stx) gamma = -gamma;
    p = gamma - dx + theta;
    q = gamma - dx + gamma + dp;
    r = p / q;
    stpc = stx + r * (stp - stx);
    stpq = stx + dx / ((fx - fp) / (stp - stx) + dx) / 2.0 * (stp - stx);
    if (absd(stpc - stx) < absd(stpq - stx))
      stpf = stpc;
    else
      stpf = stpc + (stpq - stpc) / 2.0;
    brackt = true;
  } else if (sgnd < 0.0) {
    theta = (fx - fp) * 3.0 / (stp - stx) + dx + dp;
    s = maxd(maxd(absd(theta), absd(dx)), absd(dp));
    double a = theta / s;
    gamma = s * std::sqrt(a * a - dx / s * (dp / s));
    if (stp > stx) gamma = -gamma;
    p = gamma - dp + theta;
    q = gamma - dp + gamma + dx;
    r = p / q;
    stpc = stp + r * (stx - stp);
    stpq = stp + dp / (dp - dx) * (stx - stp);
    if (absd(stpc - stp) > absd(stpq - stp))
      stpf = stpc;
    else
      stpf = stpq;
    brackt = true;
  } else if (absd(dp) < absd(dx)) {
    theta = (fx - fp) * 3.0 / (stp - stx) + dx + dp;
    s = maxd(maxd(absd(theta), absd(dx)), absd(dp));
    double a = theta / s;
    gamma = s * std::sqrt(maxd(0.0, a * a - dx / s * (dp / s)));
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
      if (absd(stpc - stp) < absd(stpq - stp))
        stpf = stpc;
      else
        stpf = stpq;
      if (stp > stx)
        stpf = mind(stp + (sty - stp) * 0.66, stpf);
      else
        stpf = maxd(stp + (sty - stp) * 0.66, stpf);
    } else {
      if (absd(stpc - stp) > absd(stpq - stp))
        stpf = stpc;
      else
        stpf = stpq;
      stpf = mind(stpmax, stpf);
      stpf = maxd(stpmin, stpf);
    }
  } else {
    if (brackt) {
      theta = (fp - fy) * 3.0 / (sty - stp) + dy + dp;
      s = maxd(maxd(absd(theta), absd(dy)), absd(dp));
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
void Lbfgsb::dcsrch(double f, double g, double& stp, double stpmax) {
  const double ftol = 1e-3, gtol = 0.9, xtol = 0.1, stpmin = 0.0;
  LsState s;
  if (ls_task_ == Ls::Start) {
    bool err = false;
    if (stp < stpmin) err = true;
    if (stp > stpmax) err = true;
    if (g >= 0.0) err = true;
    if (stpmax < stpmin) err = true;
    if (err) {  // returns before anything is saved (bfgs.cpp: "ERROR" early return)
      ls_task_ = Ls::Error;
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
    ls_task_ = Ls::FG;
    ls_ = s;
    return;
  }
  s = ls_;
  const double ftest = s.finit + stp * s.gtest;
  if (s.stage == 1 && f <= ftest && g >= 0.0) s.stage = 2;
  if (s.brackt && (stp <= s.stmin || stp >= s.stmax)) ls_task_ = Ls::Warning;
  if (s.brackt && s.stmax - s.stmin <= xtol * s.stmax) ls_task_ = Ls::Warning;
  if (stp == stpmax && f <= ftest && g <= s.gtest) ls_task_ = Ls::Warning;
  if (stp == stpmin && (f > ftest || g >= s.gtest)) ls_task_ = Ls::Warning;
  if (f <= ftest && absd(g) <= gtol * (-s.ginit)) ls_task_ = Ls::Convergence;
  if (ls_task_ == Ls::Warning || ls_task_ == Ls::Convergence) {
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
    if (absd(s.sty - s.stx) >= s.width1 * 0.66) stp = s.stx + (s.sty - s.stx) * 0.5;
    s.width1 = s.width;
    s.width = absd(s.sty - s.stx);
  }
  if (s.brackt) {
    s.stmin = mind(s.stx, s.sty);
    s.stmax = maxd(s.stx, s.sty);
  } else {
    s.stmin = stp + (stp - s.stx) * 1.1;
    s.stmax = stp + (stp - s.stx) * 4.0;
  }
  stp = maxd(stp, stpmin);
  stp = mind(stp, stpmax);
  if ((s.brackt && (stp <= s.stmin || stp >= s.stmax)) ||
      (s.brackt && s.stmax - s.stmin <= xtol * s.stmax))
    stp = s.stx;
  ls_task_ = Ls::FG;
  ls_ = s;
}

// lnsrlb (bfgs.cpp:3135-3290).  Returns true when f,g are wanted at the new x.
bool Lbfgsb::lnsrlb(double* f, double* g1, bool fresh) {
  const int n = n_;
  double* g = g1 - 1;
  if (fresh) {
    dtd_ = dot(n, &D(1), &D(1));
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
      stp_ = mind(1.0 / dnorm_, stpmx_);
    else
      stp_ = 1.0;
    for (int i = 1; i <= n; ++i) TT(i) = X(i);
    for (int i = 1; i <= n; ++i) R(i) = g[i];
    fold_ = *f;
    ifun_ = 0;
    iback_ = 0;
    ls_task_ = Ls::Start;
  }
  gd_ = dot(n, &g[1], &D(1));
  if (ifun_ == 0) {
    gdold_ = gd_;
    if (gd_ >= 0.0) {
      info_ = -4;
      return false;
    }
  }
  dcsrch(*f, gd_, stp_, stpmx_);
  xstep_ = stp_ * dnorm_;
  if (ls_task_ != Ls::Convergence && ls_task_ != Ls::Warning) {
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
void Lbfgsb::matupd(double rr, double dr) {
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
    SY(col, j) = dot(n, &D(1), &WY(1, pointr));
    SS(j, col) = dot(n, &WS(1, pointr), &D(1));
    pointr = pointr % m + 1;
  }
  if (stp_ == 1.0)
    SS(col, col) = dtd_;
  else
    SS(col, col) = stp_ * stp_ * dtd_;
  SY(col, col) = dr;
}

// formt (bfgs.cpp:2782-2868): T = theta*SS + L*D^-1*L', Cholesky-factored.
void Lbfgsb::formt(bool& ok) {
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
  if (cholesky_upper(&WT(1, 1), m_, col) != 0) {
    info_ = -3;
    ok = false;
  }
}

// mainlb (bfgs.cpp:440-1265) as a resumable state machine.
Lbfgsb::Task Lbfgsb::advance(double* f, double* g) {
  enum { L222, L333, L555, L666_FRESH, L666_RESUME, L777 } at;
  bool wrk = false;
  bool ok = true;
  const int n = n_;

  switch (phase_) {
    case Phase::Start: {
      epsmch_ = machine_eps();
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
        phase_ = Phase::Done;
        return Task::Error;
      }
      active();
      phase_ = Phase::FgStart;
      return Task::EvalFG;
    }
    case Phase::FgStart:
      nfgv_ = 1;
      projgr(g);
      if (sbgnrm_ <= pgtol_) {
        phase_ = Phase::Done;
        return Task::ConvergedPG;
      }
      at = L222;
      break;
    case Phase::FgLnsrch:
      at = L666_RESUME;
      break;
    case Phase::NewX:
      at = L777;
      break;
    default:
      return Task::Error;
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
            phase_ = Phase::Done;
            return Task::Abnormal;
          }
          if (info_ == 0) --nfgv_;
          refresh_memory();
          at = L222;
          break;
        }
        if (want_fg) {
          phase_ = Phase::FgLnsrch;
          return Task::EvalFG;
        }
        ++iter_;
        projgr(g);
        phase_ = Phase::NewX;
        return Task::NewX;
      }
      case L777: {
        if (sbgnrm_ <= pgtol_) {
          phase_ = Phase::Done;
          return Task::ConvergedPG;
        }
        double ddum = maxd(maxd(absd(fold_), absd(*f)), 1.0);
        if (fold_ - *f <= tol_ * ddum) {
          if (iback_ >= 10) info_ = -5;
          phase_ = Phase::Done;
          return Task::ConvergedF;
        }
        for (int i = 1; i <= n; ++i) R(i) = g[i - 1] - R(i);
        double rr = dot(n, &R(1), &R(1));
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

}  // namespace nghmm

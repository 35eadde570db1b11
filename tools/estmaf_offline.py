#!/usr/bin/env python3
"""Offline study of est_maf's interval rule on the sample tools/estmaf_traj_dump.py saved
(gpurun_out/estmaf_sample.npz): numpy restatement of the kernel's recursion in the odds r
(gen_func.cpp:974-1009), the current rule (k_fast_estmaf: wait until k * step fits, interval
g = min(DMAX, MULT * step / r)) against rules that PREDICT the end of the travel from the first
passes, priced with the kernel's instruction model (profiles/rNN_isa_summary.txt).
  python tools/estmaf_offline.py [post6|post1|post2]"""
import sys
import numpy as np

which = sys.argv[1] if len(sys.argv) > 1 else "post6"
d = np.load(sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/estmaf_sample.npz")
p = np.exp(d["gl"].astype(np.float64))
F = d[which]
p0, p1, p2 = p[..., 0], p[..., 1], p[..., 2]
cc = 2 * p1 * (1 - F); n2 = (2 - F) * p2
sA, sb, sC = p0, F * (p0 + p2) + cc, p2
u0, nC, fc = n2 * F + cc, n2, F * cc
tF = (2 - F).sum(1)
n = F.shape[0]
EPS = 1e-5


def sums(r):
    r = r[:, None]
    sm = sA + r * (sb + r * sC)
    return ((u0 + r * nC) / sm).sum(1), (fc / sm).sum(1)


# ---- the exact recursion: r_k, the map's values M_k = n(r_k) / d(r_k), pass counts -------------
num = np.zeros(n); den = np.zeros(n); pnum = np.full(n, 0.01); pden = np.ones(n)
R, NUM, DEN, LHS, THR = [], [], [], [], []
active = np.ones(n, bool); iters = np.zeros(n, int)
for k in range(101):
    r = pnum / (pden - pnum)
    R.append(r.copy())
    sn, sd = sums(r)
    num2 = num + r * sn; den2 = den + r * sd + tF
    lhs = np.abs(pnum * den2 - num2 * pden); thr = EPS * den2 * pden
    go = active & (lhs > thr) & (iters < 100)
    NUM.append(np.where(active, num2, num)); DEN.append(np.where(active, den2, den))
    LHS.append(lhs); THR.append(thr)
    num = np.where(active, num2, num); den = np.where(active, den2, den)
    pnum, pden = num, den
    iters += go
    active = go
R = np.array(R); NUM = np.array(NUM); DEN = np.array(DEN); LHS = np.array(LHS); THR = np.array(THR)
passes = iters + 1                      # evaluations of a site
print(f"{which}: sites {n}, passes median {np.median(passes)}, at cap {np.mean(passes >= 101):.3f}, "
      f"min {passes.min()}")
ar = np.arange(n)


def hull(k):          # [lo, hi] of the odds the passes k .. end evaluate at
    idx = np.arange(101)[:, None]
    m = (idx >= k) & (idx < passes[None, :])
    lo = np.where(m, R, np.inf).min(0); hi = np.where(m, R, -np.inf).max(0)
    return lo, hi


# ---- instruction model (VALU wave-instructions per site) ----------------------------------------
SETUP, PASS, REC, CHECK, NODE, TAIL = 283, 159, 169, 86, 131, 32


def price(n_exact, nodes, extra=0.0):
    return SETUP + (PASS + REC) * n_exact + CHECK + NODE * nodes + TAIL + extra


def rho_of(ratio):
    h = (np.sqrt(ratio) - 1) / (np.sqrt(ratio) + 1)
    x = 1 / h
    return x + np.sqrt(x * x - 1)


# ---- current rule ------------------------------------------------------------------------------
def current_rule(K0=2, DMAX=0.85, MULT=32.0, FIT=0.72, KMAX=32, BACK=0.1, MIN_GAIN=24):
    built_at = np.full(n, -1); lo_i = np.zeros(n); hi_i = np.zeros(n)
    for s in range(n):
        nb = K0
        for k in range(passes[s] - 1):          # pass k done, `again` true
            it = k + 1
            nb -= 1
            if nb > 0:
                continue
            m_est = it * (np.sqrt(LHS[k, s] / THR[k, s]) - 1)
            rn, rp = R[k + 1, s], R[k, s]
            step = abs(rn - rp)
            reach = DMAX * rn if rn >= rp else DMAX / (1 + DMAX) * rn
            fits = it * step <= FIT * reach
            if (not fits) and it < KMAX and m_est >= MIN_GAIN:
                nb = 1
                continue
            if m_est >= MIN_GAIN and 100 - it >= MIN_GAIN:
                g = min(DMAX, max(MULT * step / rn, 1e-3))
                if rn >= rp:
                    lo, hi = rn * (1 - BACK * g), rn * (1 + g)
                else:
                    lo, hi = rn / (1 + g), rn * (1 + BACK * g)
                built_at[s] = it; lo_i[s] = lo; hi_i[s] = hi
            break
    return built_at, lo_i, hi_i


def report(name, built_at, lo_i, hi_i, nodes, extra=0.0):
    b = built_at >= 0
    if not b.any():
        print(f"{name}: no site builds")
        return None
    lo_t = np.zeros(n); hi_t = np.zeros(n)
    for s in np.where(b)[0]:
        seg = R[built_at[s]:passes[s], s]
        lo_t[s], hi_t[s] = seg.min(), seg.max()
    inside = b & (lo_t >= lo_i) & (hi_t <= hi_i)
    ratio = np.where(b, hi_i / np.maximum(lo_i, 1e-300), 1.0)
    # exact passes of a site that stays inside: built_at + 1 (the check); one that leaves pays a
    # second interval (counted as nodes + 3 passes more)
    n_ex = np.where(b, built_at + 1, passes)
    cost = price(n_ex, np.where(b, nodes, 0), extra) + np.where(b & ~inside, NODE * nodes + 3 * (PASS + REC), 0)
    rho = rho_of(np.maximum(ratio, 1.0001))
    print(f"{name}: built {b.mean():.3f}, built after pass median {np.median(built_at[b]):.0f} "
          f"mean {built_at[b].mean():.2f}, stays inside {inside[b].mean():.4f}, ratio hi/lo p50/p99/max "
          f"{np.percentile(ratio[b], 50):.3f}/{np.percentile(ratio[b], 99):.3f}/{ratio[b].max():.3f}, "
          f"rho^-nodes worst {np.max(rho[b] ** -float(nodes)):.1e}, VALU per site {cost.mean():.0f}")
    return cost.mean()


ba, lo_c, hi_c = current_rule()
report("current rule (12 nodes)", ba, lo_c, hi_c, 12)
for k in (1, 2, 3, 4):
    lo, hi = hull(k)
    print(f"odds of the passes {k} .. end: hi/lo percentiles 50/90/99/100 "
          f"{np.round(np.percentile(hi / lo, [50, 90, 99, 100]), 3)}, rising at {np.mean(R[passes - 1, ar] > R[k]):.3f} of the sites")


# ---- predicted end of the travel ----------------------------------------------------------------
def predicted_rule(kb, margin_hi, margin_lo, nodes, max_ratio, back=0.02):
    """Build after pass kb (kb >= 2) for every site that still runs: the map M(f) = n/d of the
    last two passes gives a secant, its fixed point f* the end the running average creeps towards."""
    built_at = np.full(n, -1); lo_i = np.zeros(n); hi_i = np.zeros(n)
    f = NUM / DEN                                       # f after pass k
    fprev = np.vstack([np.full((1, n), 0.01), f[:-1]])  # f pass k evaluated at
    dN = np.diff(np.vstack([np.zeros((1, n)), NUM]), axis=0)
    dD = np.diff(np.vstack([np.zeros((1, n)), DEN]), axis=0)
    M = dN / dD                                         # the map's value at fprev[k]
    for s in range(n):
        if passes[s] - 1 < kb:
            continue
        k = kb - 1                                      # last pass done
        fa, fb = fprev[k - 1, s], fprev[k, s]
        Ma, Mb = M[k - 1, s], M[k, s]
        lam = (Mb - Ma) / (fb - fa) if fb != fa else 0.0
        lam = min(max(lam, 0.0), 0.95)
        fstar = (Mb - lam * fb) / (1 - lam)
        fnow = f[k, s]
        rnow = fnow / (1 - fnow)
        fstar = min(max(fstar, 1e-9), 1 - 1e-9)
        rstar = fstar / (1 - fstar)
        if rstar >= rnow:
            lo, hi = rnow * (1 - back), rnow + (rstar - rnow) * (1 + margin_hi)
            hi = max(hi, rnow * (1 + margin_lo))
        else:
            hi, lo = rnow * (1 + back), rnow + (rstar - rnow) * (1 + margin_hi)
            lo = min(lo, rnow / (1 + margin_lo))
            if lo <= 0:
                lo = rnow / max_ratio
        if hi / lo > max_ratio:
            continue                                    # (would wait: priced as not built here)
        built_at[s] = kb; lo_i[s] = lo; hi_i[s] = hi
    return built_at, lo_i, hi_i


for kb in (2, 3):
    for nodes, max_ratio in ((12, 2.1), (11, 1.8), (10, 1.6)):
        for mh in (0.05, 0.15, 0.3):
            ba, lo_p, hi_p = predicted_rule(kb, mh, 0.02, nodes, max_ratio)
            report(f"secant after pass {kb}, margin {mh}, {nodes} nodes, ratio <= {max_ratio}", ba, lo_p, hi_p, nodes)


# ---- what a perfect knowledge of the travel would buy (round 6: the simulator's r regimes) -------
def oracle_policy(max_ratio=2.0, nodes=12, wide_ratio=None, wide_nodes=24, back=1.02):
    """Every site builds at the first pass from which the odds' hull to the end fits an interval of
    ratio max_ratio (or, when a wide interval is allowed and cheaper than waiting, wide_ratio)."""
    cost = np.zeros(n)
    built = np.zeros(n, int)
    for s in range(n):
        best = price(passes[s], 0)             # never build: every pass exact
        for k in range(2, min(passes[s] - 1, 40)):
            seg = R[k:passes[s], s]
            ratio = seg.max() / seg.min() * back
            left = passes[s] - k
            if left < 24:
                break
            c = None
            if ratio <= max_ratio:
                c = price(k + 1, nodes)
            elif wide_ratio and ratio <= wide_ratio:
                c = price(k + 1, wide_nodes)
            if c is not None and c < best:
                best, built[s] = c, k
                if ratio <= max_ratio:
                    break
        cost[s] = best
    return cost.mean(), np.mean(built > 0), np.mean(built[built > 0])


c_now = report("current rule again", *current_rule(), 12)
for label, kw in (("perfect knowledge, narrow only", {}),
                  ("perfect knowledge, narrow or wide (7.6, 24 nodes)", dict(wide_ratio=7.6)),
                  ("perfect knowledge, narrow or medium (3.2, 16 nodes)", dict(wide_ratio=3.2, wide_nodes=16))):
    c, b, kb = oracle_policy(**kw)
    print(f"{label}: VALU per site {c:.0f} ({c / c_now:.3f} of the current rule's), builds {b:.3f} at pass {kb:.2f} on average")

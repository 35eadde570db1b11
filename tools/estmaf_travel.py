#!/usr/bin/env python3
"""How far est_maf's running average travels after the interval is built, against the length the
kernel's rule gives the interval (k_fast_estmaf: g = min(EST_DMAX, EST_MULT * step / r)) -- on the
benchmark's data set, with the posteriors of the GPU's own E-step after a few EM iterations.
numpy restatement of the kernel's recursion (gen_func.cpp:974-1009 in the odds r), sample of sites.
  python tools/estmaf_travel.py [n_sites=20000] [iterations=6]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("ngsf-hmm_amd")
import torch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
its = int(sys.argv[2]) if len(sys.argv) > 2 else 6
I = 1000
dev = torch.device("cuda", 0)
sim = pkg.simulate.IndexedSim(1000, 1_000_000, dev, seed=12345)
gl_d, pos_d = sim.gl((0, I), (0, S)), sim.pos_dist(0, S)
torch.cuda.synchronize()
with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as fa:
    fa.load_device(gl_d.data_ptr(), pos_d.data_ptr())
    fa.set_params(0.1, 0.2, 0.1)
    fa.init_emission()
    for _ in range(its):
        fa.iter_EM()
    fa.estep()
    post = fa.marg_prob.copy()            # [I][S]
sites = np.arange(0, S, 10)
gl = gl_d.cpu().numpy()[sites]            # [n][I][3] log
p = np.exp(gl)
F = post[:, sites].T                      # [n][I]
p0, p1, p2 = p[..., 0], p[..., 1], p[..., 2]
cc = 2 * p1 * (1 - F); n2 = (2 - F) * p2
sA, sb, sC = p0, F * (p0 + p2) + cc, p2
u0, nC, fc = n2 * F + cc, n2, F * cc
tF = (2 - F).sum(1)
n = len(sites)
num = np.zeros(n); den = np.zeros(n); pnum = np.full(n, 0.01); pden = np.ones(n)
traj = []
active = np.ones(n, bool); iters = np.zeros(n, int)
for k in range(101):
    r = pnum / (pden - pnum)
    traj.append(r.copy())
    sm = sA + r[:, None] * (sb + r[:, None] * sC)
    sn = ((u0 + r[:, None] * nC) / sm).sum(1); sd = (fc / sm).sum(1)
    num2 = num + r * sn; den2 = den + r * sd + tF
    f_old = pnum / pden; f_new = num2 / den2
    go = active & (np.abs(f_old - f_new) > 1e-5) & (iters < 100)
    upd = active
    num = np.where(upd, num2, num); den = np.where(upd, den2, den)
    pnum = np.where(upd, num2, pnum); pden = np.where(upd, den2, pden)
    iters += go
    active = go
traj = np.array(traj)                     # [pass][site]: odds at which pass k evaluates
passes = iters + 1
print("sites", n, "passes: median", np.median(passes), "at cap", np.mean(passes >= 100))
for kb in (3, 4, 6):
    # interval built after pass kb (evaluations kb+1 ... follow): what the rule gives, what is used
    rn, rprev = traj[kb + 1], traj[kb]
    step = np.abs(rn - rprev)
    last = traj[np.minimum(passes, 100), np.arange(n)]
    lo = np.minimum(traj[kb + 1:].min(0), rn); hi = np.maximum(traj[kb + 1:].max(0), rn)
    need = hi / lo - 1
    g = np.minimum(0.85, np.maximum(32 * step / rn, 1e-3))
    q = [50, 90, 99, 99.9, 100]
    print(f"build after pass {kb}: needed hi/lo - 1 percentiles {q}: {np.percentile(need, q)}")
    print(f"   rule's g: {np.percentile(g, q)};  g/need median {np.median(g / np.maximum(need, 1e-12)):.1f};"
          f" k*step/r: {np.percentile((kb + 1) * step / rn, q)}")
    for cap in (0.1, 0.2, 0.3, 0.5):
        print(f"   sites whose remaining travel fits a relative length {cap}: {np.mean(need * 1.3 <= cap):.3f}")

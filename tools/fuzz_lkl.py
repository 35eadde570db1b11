#!/usr/bin/env python3
"""Stress run of the fast-mode objective kernels against the exact-mode forward recursion:
finite-difference groups (x, F +- eh, alpha +- eh and the one-sided patterns at the bounds of
EM.cpp:425-427), general points, on likelihood data and on called genotypes (packed), several
chromosomes.  Prints the worst relative difference per case; 1e-9 is the stated tolerance.
   python tools/fuzz_lkl.py      (needs an MI355X)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ngsf-hmm_amd")


def fd_group(F, A, lbF=1e-15, ubF=1 - 1e-15, lbA=1e-15, ubA=10.0):
    """the points findmax_bfgs's getgradient asks for at x = (F, A) (bfgs.cpp:22-43)"""
    pts = [(F, A)]
    for k, (x, lb, ub) in enumerate(((F, lbF, ubF), (A, lbA, ubA))):
        eh = (1e-8 * (abs(x) + 1)) ** 0.67
        if x - eh < lb:
            probes = [x + 2 * eh]
        elif x + eh > ub:
            probes = [x - 2 * eh]
        else:
            probes = [x + eh, x - eh]
        for p in probes:
            pts.append((p, A) if k == 0 else (F, p))
    return pts


def main():
    rng = np.random.default_rng(7)
    worst = 0.0
    for I, S, nchr, call in ((40, 30_000, 1, False), (40, 30_000, 7, False), (40, 30_000, 3, True),
                             (7, 200_000, 2, False), (7, 200_000, 2, True)):
        d = pkg.simulate.simulate(I, S, seed=S + nchr, n_chrom=nchr, missing_rate=0.05, indF="r",
                                  freq="r", alpha=0.5)
        hs = {}
        for name, mode in (("fast", pkg.MODE_FAST), ("exact", pkg.MODE_EXACT)):
            h = pkg.NgsFHMM(I, S, mode=mode | (pkg.GENO_PACKED if call else 0))
            h.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=call)
            h.set_params(0.3, 0.1, np.clip(d.freq, 0.02, 0.98))
            h.init_emission()
            hs[name] = h
        Fs = [1e-15, 1e-9, 1e-6, 1e-3, 0.3, 0.7, 1 - 1e-3, 1 - 1e-6, 1 - 1e-9, 1 - 1e-15]
        As = [1e-15, 1e-8, 1e-3, 0.05, 1.0, 9.99, 10.0]
        ind, F, A = [], [], []
        for i in range(I):
            for _ in range(3):
                f0 = Fs[rng.integers(len(Fs))] if rng.random() < 0.6 else rng.uniform(0, 1)
                a0 = As[rng.integers(len(As))] if rng.random() < 0.6 else 10 ** rng.uniform(-6, 1)
                for f, a in fd_group(f0, a0):
                    ind.append(i); F.append(f); A.append(a)
        ind, F, A = np.array(ind, dtype=np.uint32), np.array(F), np.array(A)
        got = hs["fast"].lkl(ind, F, A)      # grouped by individual (<= 5 points per group) inside
        want = hs["exact"].lkl(ind, F, A)
        fin = np.isfinite(want)
        rel = np.abs(got[fin] - want[fin]) / np.abs(want[fin])
        bad = int((~np.isfinite(got[fin])).sum())
        worst = max(worst, np.nanmax(rel))
        print(f"I={I} S={S} chr={nchr} call_geno={int(call)}: {len(ind)} points, max rel {np.nanmax(rel):.2e}, "
              f"non-finite in fast where exact is finite: {bad}, exact non-finite: {int((~fin).sum())}",
              flush=True)
        k = np.nanargmax(rel)
        print("   worst at F=%.17g alpha=%.6g  fast %.10f exact %.10f" % (F[fin][k], A[fin][k], got[fin][k], want[fin][k]))
        for h in hs.values():
            h.close()
    print("worst", worst)


if __name__ == "__main__":
    main()

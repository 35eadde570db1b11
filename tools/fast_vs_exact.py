#!/usr/bin/env python3
"""End-to-end comparison of the two arithmetic modes at a moderate size: the same EM run
(reference loop control, EM.cpp:27-103) in --mode fast and --mode exact semantics through
the Python mirror; prints the total log-likelihood per iteration for both and the final
differences in indF / alpha / freq / Viterbi paths.

  python tools/fast_vs_exact.py [n_ind n_sites iters]      (needs an MI355X)
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ngsf-hmm_amd")


def main():
    I = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    d = pkg.simulate.simulate(I, S, seed=4242, n_chrom=4, missing_rate=0.02, indF="r", freq="r")
    gl = pkg.simulate.normalise_log_gl(d.gl)
    res = {}
    for mode, name in ((pkg.MODE_FAST, "fast"), (pkg.MODE_EXACT, "exact")):
        with pkg.NgsFHMM(I, S, mode=mode) as hmm:
            hmm.load(gl, d.pos_dist_mb)
            hmm.set_params(0.1, 0.2, 0.1)
            hmm.init_emission()
            lk = []
            t0 = time.time()
            for _ in range(iters):
                hmm.iter_EM()
                lk.append(float(np.sum(hmm.ind_lkl)))
            dt = time.time() - t0
            res[name] = dict(lk=np.array(lk), indF=hmm.indF.copy(), alpha=hmm.alpha.copy(),
                             freq=hmm.freq.copy(), path=hmm.viterbi(), dt=dt)
        print(f"{name}: {iters} iterations in {dt:.2f} s, tot_lkl {lk[0]:.6f} -> {lk[-1]:.6f}", flush=True)
    a, b = res["fast"], res["exact"]
    rel = np.abs(a["lk"] - b["lk"]) / np.abs(b["lk"])
    print("tot_lkl relative difference per iteration: first %.2e, max %.2e, last %.2e"
          % (rel[0], rel.max(), rel[-1]))
    print("final indF: max |diff| %.2e, median %.2e;  alpha: median rel %.2e;  freq: max |diff| %.2e"
          % (np.abs(a["indF"] - b["indF"]).max(), np.median(np.abs(a["indF"] - b["indF"])),
             np.median(np.abs(a["alpha"] - b["alpha"]) / np.abs(b["alpha"])),
             np.abs(a["freq"] - b["freq"]).max()))
    print("Viterbi paths: %.4f %% of cells differ" % (100 * (a["path"] != b["path"]).mean()))


if __name__ == "__main__":
    main()

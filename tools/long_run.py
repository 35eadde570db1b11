#!/usr/bin/env python3
"""Stability run at scale: N EM iterations of the fused fast-mode iteration on a synthetic
cohort (default: one GPU's share of BASELINE config 5, 625 x 5M called genotypes in 25
chromosomes, 100 iterations as the config asks), printing the total log-likelihood as it goes
and checking that every array stays finite and in range.
   python tools/long_run.py [n_ind n_sites n_chrom n_iter call_geno]      (needs an MI355X)"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ngsf-hmm_amd")
import torch  # noqa: E402

I = int(sys.argv[1]) if len(sys.argv) > 1 else 625
S = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
NCHR = int(sys.argv[3]) if len(sys.argv) > 3 else 25
NIT = int(sys.argv[4]) if len(sys.argv) > 4 else 100
CALL = int(sys.argv[5]) if len(sys.argv) > 5 else 1

dev = torch.device("cuda", 0)
mode = pkg.MODE_FAST | (pkg.GENO_PACKED if CALL else 0)
h = pkg.NgsFHMM(I, S, mode=mode)
pos, chunks = pkg.simulate.simulate_torch_chunks(I, S, dev, seed=11, n_chrom=NCHR,
                                                 chunk_sites=50_000)


def feed():
    for s0, c in chunks:
        torch.cuda.synchronize()
        yield s0, c.shape[0], c.data_ptr()


h.load_chunks_device(pos.data_ptr(), feed(), space=0, call_geno=bool(CALL))
h.set_params(0.1, 0.2, 0.1)
h.init_emission()
t0 = time.time()
prev = None
worst_drop = 0.0
for it in range(NIT):
    h.iter_EM()
    tot = float(h.ind_lkl.sum())
    assert np.isfinite(tot), (it, tot)
    if prev is not None:
        worst_drop = min(worst_drop, tot - prev)
    if it < 5 or it % 10 == 9:
        print(f"iteration {it + 1:3d}: total lkl {tot:.6f}", flush=True)
    prev = tot
dt = time.time() - t0
f = h.freq
assert np.isfinite(f).all() and (f >= 0).all() and (f < 1).all()
assert np.isfinite(h.indF).all() and np.isfinite(h.alpha).all()
print(f"{NIT} iterations of {I} x {S} in {dt:.1f} s = {dt / NIT * 1e3:.1f} ms each; "
      f"largest decrease of the total lkl between iterations {worst_drop:.3g}; "
      f"indF {h.indF.min():.4g}..{h.indF.max():.4g}, alpha {h.alpha.min():.4g}..{h.alpha.max():.4g}, "
      f"freq {f.min():.4g}..{f.max():.4g}")

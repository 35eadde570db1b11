#!/usr/bin/env python3
"""One objective round (5 finite-difference points for every individual) through nghmm_lkl_batch,
kernel milliseconds per call.   python tools/lkl_timing.py [n_ind n_sites reps]   (needs an MI355X)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ngsf-hmm_amd")
import torch
I = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
gl, pos = pkg.simulate.simulate_torch(I, S, torch.device("cuda", 0), seed=5)
torch.cuda.synchronize()
with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as h:
    h.load_device(gl.data_ptr(), pos.data_ptr())
    del gl
    h.set_params(0.1, 0.2, 0.1)
    h.init_emission()
    h.estep()
    eh = 4e-6
    ind = np.repeat(np.arange(I), 5).astype(np.uint32)
    F = np.tile([0.1, 0.1 + eh, 0.1 - eh, 0.1, 0.1], I)
    A = np.tile([0.2, 0.2, 0.2, 0.2 + eh, 0.2 - eh], I)
    ms = []
    for k in range(reps):
        v = h.lkl(ind, F, A)
        ms.append(h.kernel_ms("lkl_batch")[0])
    print("%d x %d: lkl round of 5 points, kernel ms min %.3f median %.3f  (value[0] %.6f)" %
          (I, S, min(ms), sorted(ms)[len(ms) // 2], v[0]))

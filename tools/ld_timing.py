#!/usr/bin/env python3
"""The opt-in intended --freq_est 2 (nghmm_mstep_freq with NGHMM_LD_INTENDED): seconds per call.
   python tools/ld_timing.py [n_ind n_sites mode(fast|exact)]      (needs an MI355X)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ngsf-hmm_amd")
import torch
I = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
mode = pkg.MODE_EXACT if (len(sys.argv) > 3 and sys.argv[3] == "exact") else pkg.MODE_FAST
gl, pos = pkg.simulate.simulate_torch(I, S, torch.device("cuda", 0), seed=5)
torch.cuda.synchronize()
with pkg.NgsFHMM(I, S, mode=mode) as h:
    h.load_device(gl.data_ptr(), pos.data_ptr())
    del gl
    h.set_params(0.1, 0.2, 0.1)
    h.init_emission()
    h.estep()
    for fe, label in ((1, "freq_est 1"), (2 | pkg.LD_INTENDED, "intended freq_est 2"),
                      (2 | pkg.LD_INTENDED | pkg.EPROB_LD, "intended freq_est 2 + e_prob 2")):
        t0 = time.time()
        h.mstep_freq(fe)
        h.synchronize()
        print("%d x %d %s: %-32s %.3f s" % (I, S, "exact" if mode == pkg.MODE_EXACT else "fast", label, time.time() - t0), flush=True)

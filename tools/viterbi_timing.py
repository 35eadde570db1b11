#!/usr/bin/env python3
"""Viterbi decoding at size (fast-mode handle: emissions recomputed in log space, transition
logs per chunk, the two sweeps): seconds per call.   python tools/viterbi_timing.py [n_ind n_sites]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ngsf-hmm_amd")
import torch
I = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
gl, pos = pkg.simulate.simulate_torch(I, S, torch.device("cuda", 0), seed=5)
torch.cuda.synchronize()
with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as h:
    h.load_device(gl.data_ptr(), pos.data_ptr())
    del gl
    h.set_params(0.1, 0.2, 0.1)
    h.init_emission()
    h.iter_EM()
    for k in range(3):
        t0 = time.time()
        p = h.viterbi()
        dt = time.time() - t0
        print("viterbi call %d: %.3f s (kernels %.1f ms), checksum %d" % (k, dt, h.kernel_ms("viterbi")[0], int(p[:, ::997].sum())), flush=True)

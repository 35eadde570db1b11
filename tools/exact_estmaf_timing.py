#!/usr/bin/env python3
"""Exact-mode est_maf alone (nothing underneath it): the wave-per-site kernel against the
lane-per-site kernel, kernel milliseconds and bit-identity of the frequencies.
   python tools/exact_estmaf_timing.py [n_ind n_sites]      (needs an MI355X)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("ngsf-hmm_amd")
I = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
sim = pkg.simulate.IndexedSim(I, S, torch.device("cuda", 0), seed=12345)
gl, pos = sim.gl(), sim.pos_dist(0, S)
torch.cuda.synchronize()
with pkg.NgsFHMM(I, S, mode=pkg.MODE_EXACT) as h:
    h.load_device(gl.data_ptr(), pos.data_ptr())
    del gl
    h.set_params(0.1, 0.2, 0.1)
    h.init_emission()
    h.estep()
    res = {}
    for lanes, sel in ((0, 0), (0, 1), (1, 0), (0, 0), (0, 1)):
        h.set_switch("estmaf_exact_lanes", lanes)
        h.set_switch("estmaf_exact_sel", sel)
        h.set_params(None, None, 0.1)
        h.mstep_freq(1)
        ms = h.kernel_ms("est_maf")[0]
        f = h.freq
        print(f"{I} x {S}: lanes={lanes} select forms={sel}: est_maf {ms:.1f} ms", flush=True)
        if (lanes, sel) in res:
            assert np.array_equal(res[(lanes, sel)], f)
        res[(lanes, sel)] = f
    assert all(np.array_equal(res[(0, 0)], v) for v in res.values())
    print("frequencies bit-identical")

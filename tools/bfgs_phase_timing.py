#!/usr/bin/env python3
"""Phase times of k_bfgs_advance's slowest workgroup (a build with -DNGHMM_BFGS_TIMING:
make -C ngsf-hmm_amd/csrc EXTRA=-DNGHMM_BFGS_TIMING).  python tools/bfgs_phase_timing.py [I S]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ngsf-hmm_amd")
import torch  # noqa: E402

I, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 100_000)
lib = pkg.load_library()
dev = torch.device("cuda", 0)
sim = pkg.simulate.IndexedSim(I, S, dev, seed=12345)
gl, pos = sim.gl(), sim.pos_dist(0, S)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
names = ["stage in", "consume (solver)", "plan + out", "arrays back", "fence + publish"]
with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as h:
    h.set_switch("spans", 1)    # kernel times of the fused iterations (off by default)
    h.load_device(gl.data_ptr(), pos.data_ptr())
    h.set_params(0.1, 0.2, 0.1)
    h.init_emission()
    for it in range(8):
        lib.nghmm_debug_bfgs_phases(buf, 1)
        st = h.iter_EM()
        lib.nghmm_debug_bfgs_phases(buf, 0)
        # wall_clock64: 100 MHz
        print(f"iteration {it + 1}: {st.rounds} rounds; slowest workgroup per phase (us):",
              ", ".join(f"{n} {buf[k] / 100.0:.1f}" for k, n in enumerate(names)),
              f"| bfgs kernels {h.kernel_ms('bfgs')[0] * 1e3:.0f} us total")

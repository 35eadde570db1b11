#!/usr/bin/env python3
"""Exact mode at size: one EM iteration through the C ABI with the producer-consumer
recursions (kernels_exact_pc.hip, the default) and, optionally, with the one-lane-per-chain
kernels they replace (NGHMM_EXACT_SERIAL=1): per-phase kernel times, and every array of the two
runs compared bit for bit.

  python tools/exact_timing.py [n_ind n_sites iters serial(0/1)]      (needs an MI355X)
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ngsf-hmm_amd")


def run(I, S, iters, gl, pos, serial):
    if serial:
        os.environ["NGHMM_EXACT_SERIAL"] = "1"
    else:
        os.environ.pop("NGHMM_EXACT_SERIAL", None)
    out = {"phases": []}
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_EXACT) as hmm:
        hmm.load_device(gl.data_ptr(), pos.data_ptr())
        hmm.set_params(0.1, 0.2, 0.1)
        hmm.init_emission()
        for it in range(iters):
            t0 = time.time()
            st = hmm.iter_EM()
            dt = time.time() - t0
            ph = dict(iteration=it, seconds=round(dt, 3), rounds=int(st.rounds), points=int(st.points),
                      forward_ms=round(hmm.kernel_ms("forward")[0], 2), backward_ms=round(hmm.kernel_ms("backward")[0], 2),
                      lkl_ms=round(hmm.kernel_ms("lkl_batch")[0], 2), est_maf_ms=round(hmm.kernel_ms("est_maf")[0], 2),
                      emission_ms=round(hmm.kernel_ms("emission")[0], 2))
            out["phases"].append(ph)
            print(("serial" if serial else "pc"), json.dumps(ph), flush=True)
        out["ind_lkl"] = hmm.ind_lkl.copy()
        out["indF"] = hmm.indF.copy()
        out["alpha"] = hmm.alpha.copy()
        out["freq"] = hmm.freq.copy()
        out["post"] = hmm.marg_prob[: min(I, 64)].copy()
        t0 = time.time()
        out["path"] = hmm.viterbi()[: min(I, 64)].copy()
        print("viterbi %.2f s" % (time.time() - t0), flush=True)
    return out


def main():
    import torch
    I = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    with_serial = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    gl, pos = pkg.simulate.simulate_torch(I, S, torch.device("cuda", 0), seed=5)
    pos[S // 3] = float("inf")
    torch.cuda.synchronize()
    a = run(I, S, iters, gl, pos, False)
    if with_serial:
        b = run(I, S, iters, gl, pos, True)
        for k in ("ind_lkl", "indF", "alpha", "freq", "post", "path"):
            same = np.array_equal(a[k], b[k])
            print(f"{k}: {'bit-identical' if same else 'DIFFERENT'}")
            assert same, k
        ta = sum(p["seconds"] for p in a["phases"])
        tb = sum(p["seconds"] for p in b["phases"])
        print(f"{I} x {S}, {iters} iterations: producer-consumer {ta:.2f} s, serial {tb:.2f} s "
              f"({tb / ta:.1f}x)")


if __name__ == "__main__":
    main()

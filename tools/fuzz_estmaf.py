#!/usr/bin/env python3
"""Stress run of est_maf's interpolated passes against all-exact passes (switch estmaf_interp = 0)
over cohort sizes (every kernel variant: rows, 1..16 individuals per lane, several waves per
site, the streaming kernel) and data regimes (depth, uniform / extreme site frequencies,
called genotypes).  Prints the worst relative difference per case; anything above 1e-10 is
a bug.   python tools/fuzz_estmaf.py      (needs an MI355X)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ngsf-hmm_amd")


def main():
    import torch
    dev = torch.device("cuda", 0)
    worst = 0.0
    cases = []
    for I in (8, 30, 64, 100, 128, 129, 300, 600, 700, 1024, 1500, 4100):
        S = max(2_000, min(150_000, 30_000_000 // I))
        for kw in (dict(freq="r"), dict(freq="r", depth=0.5), dict(freq="r", depth=30.0, error=0.001),
                   dict(freq=0.003), dict(freq=0.995), dict(freq="r", call_geno=True)):
            cases.append((I, S, kw))
    for I, S, kw in cases:
        kw = dict(kw)
        call = kw.pop("call_geno", False)
        gl, pos = pkg.simulate.simulate_torch(I, S, dev, seed=I + 17, **kw)
        torch.cuda.synchronize()
        hmm = pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST | (pkg.GENO_PACKED if call else 0))
        hmm.load_chunks_device(pos.data_ptr(), [(0, S, gl.data_ptr())], space=0, call_geno=call)
        del gl, pos
        res = {}
        try:
            for interp in ("0", "1"):
                hmm.set_switch("estmaf_interp", int(interp))
                hmm.set_params(np.full(I, 0.2), np.full(I, 0.05), np.full(S, 0.1))
                hmm.init_emission()
                out = []
                for _ in range(2):
                    hmm.estep()
                    hmm.mstep_freq(1)
                    out.append(hmm.freq.copy())
                res[interp] = out
        except pkg.NgsFHMMError as e:
            print(f"I={I} S={S} {kw} call_geno={call}: {e}")
            hmm.close()
            continue
        hmm.close()
        torch.cuda.empty_cache()
        f0, f1 = res["0"][0], res["1"][0]     # first step: both start from the same posteriors
        ok = np.isfinite(f0) & (f0 > 0)
        rel = np.abs(f1[ok] - f0[ok]) / f0[ok]
        worst = max(worst, rel.max())
        flag = "  <-- BUG" if rel.max() > 1e-10 else ""
        print(f"I={I:5d} S={S:6d} {str(kw):45s} call_geno={int(call)}: max rel {rel.max():.2e}, "
              f"sites > 1e-12: {int((rel > 1e-12).sum())}{flag}", flush=True)
    print("worst", worst)


if __name__ == "__main__":
    main()

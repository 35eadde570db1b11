#!/usr/bin/env python3
"""Time one EM iteration (fast mode) on data regimes other than the benchmark's, to catch
performance cliffs: sequencing depth (sharper likelihoods -> more snapped posteriors, more
sites on est_maf's careful route), random site frequencies, many chromosomes, fast
recombination (alpha * d no longer tiny: general exp in the objective kernel).

  python tools/perf_variants.py [n_ind n_sites [substring of the regime name]]   (needs an MI355X)
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
    import torch
    I = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
    dev = torch.device("cuda", 0)
    regimes = [
        ("benchmark model (depth 2, freq 0.2, alpha 0.01)", {}, {}),
        ("depth 10", dict(depth=10.0), {}),
        ("depth 30, error 0.001", dict(depth=30.0, error=0.001), {}),
        ("uniform site frequencies", dict(freq="r"), {}),
        ("alpha 1.0 (short tracts)", dict(alpha=1.0), dict(alpha0=1.0)),
        ("indF 0.05 (few tracts)", dict(indF=0.05), dict(indF0=0.02)),
        ("22 chromosomes", {}, dict(n_chrom=22)),
    ]
    only = sys.argv[3] if len(sys.argv) > 3 else ""
    for name, kw, opt in regimes:
        if only not in name:
            continue
        gl, pos = pkg.simulate.simulate_torch(I, S, dev, seed=7, **kw)
        n_chrom = opt.get("n_chrom", 1)
        if n_chrom > 1:
            for k in range(1, n_chrom):
                pos[k * (S // n_chrom)] = float("inf")
        torch.cuda.synchronize()
        hmm = pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST)
        hmm.set_switch("spans", 1)    # kernel times of the fused iterations (off by default)
        hmm.load_device(gl.data_ptr(), pos.data_ptr())
        del gl, pos
        hmm.set_params(opt.get("indF0", 0.1), opt.get("alpha0", 0.2), 0.1)
        hmm.init_emission()
        times, fam = [], {}
        for it in range(7):
            t0 = time.perf_counter()
            st = hmm.iter_EM()
            times.append((time.perf_counter() - t0) * 1e3)
            if it >= 4:
                for k in ("forward", "lkl_batch", "est_maf"):
                    fam[k] = fam.get(k, 0.0) + hmm.kernel_ms(k)[0] / 3
        print(f"{name:48s} {np.mean(times[4:]):7.2f} ms/iter  "
              f"(rounds {st.rounds}, " + ", ".join(f"{k} {v:.2f}" for k, v in fam.items()) + ")",
              flush=True)
        hmm.close()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

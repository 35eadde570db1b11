#!/usr/bin/env python3
"""Random cohort shapes through the fast path against the CPU oracle (per call, 1e-9):
E-step log-likelihoods and posteriors, allele-frequency step, then two whole iterations for
finiteness -- likelihood data and called genotypes (packed), 1..5 chromosomes, group of 2
handles where the shape allows.   python tools/fuzz_shapes.py [n_cases]   (needs an MI355X)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("ngsf-hmm_amd")
import orclib  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(2026)
    orc = orclib.Oracle("libm")
    worst = dict(lkl=0.0, post=0.0, freq=0.0)
    for case in range(n_cases):
        I = int(rng.choice([1, 2, 3, 5, 15, 16, 17, 33, 64, 65, 127, 129, 200, 513, 700]))
        S = int(rng.integers(1, 40)) if rng.random() < 0.2 else int(rng.integers(40, 6000))
        nchr = int(rng.integers(1, 6)) if S > 20 else 1
        call = bool(rng.random() < 0.4)
        d = pkg.simulate.simulate(I, S, seed=1000 + case, n_chrom=nchr, missing_rate=0.1, indF="r",
                                  freq=float(rng.uniform(0.05, 0.6)), alpha=float(10 ** rng.uniform(-2, 0.5)))
        gl = orc.prepare_gl(d.gl, 0, call_geno=call)
        F0, A0, f0 = rng.uniform(0.01, 0.9, I), 10 ** rng.uniform(-2, 0.5, I), rng.uniform(0.05, 0.6, S)
        em = orclib.OracleEM(orc, gl, d.pos_dist_mb)
        em.set_params(F0, A0, f0)
        assert em.init_emission() == 0
        if em.estep() != 0:
            print(f"case {case}: oracle E-step fails (reference behaviour), skipped")
            continue
        h = pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST | (pkg.GENO_PACKED if call else 0))
        h.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=call)
        h.set_params(F0, A0, f0)
        h.init_emission()
        lk = h.estep().copy()
        e_l = np.max(np.abs(lk - em.ind_lkl) / np.abs(em.ind_lkl))
        m = h.marg_prob
        e_p = np.max(np.abs(m - em.marg) / np.maximum(em.marg, 1e-5))
        em.mstep_freq(1)
        h.mstep_freq(1)
        e_f = np.max(np.abs(h.freq - em.freq) / np.maximum(em.freq, 1e-3))   # relative above 1e-3
        ok2 = True
        try:
            h.set_params(F0, A0, f0)
            h.init_emission()
            for _ in range(2):
                h.iter_EM()
            ok2 = bool(np.isfinite(h.ind_lkl).all() and np.isfinite(h.freq).all())
        except pkg.NgsFHMMError as e:
            ok2 = f"iter_EM: {e}"
        h.close()
        grp = ""
        if I % 2 == 0 and S % 2 == 0 and I >= 2:
            parts = [pkg.NgsFHMM(I // 2, S, mode=pkg.MODE_FAST | (pkg.GENO_PACKED if call else 0)) for _ in range(2)]
            for r, p in enumerate(parts):
                p.load_raw(np.ascontiguousarray(d.gl[:, r * (I // 2):(r + 1) * (I // 2)]), d.pos_dist_mb,
                           space=0, call_geno=call)
                p.set_params(F0[r * (I // 2):(r + 1) * (I // 2)], A0[r * (I // 2):(r + 1) * (I // 2)], f0)
                p.init_emission()
            g = pkg.Group(parts)
            g.iter_EM(1, True, True)                      # fixed parameters: E-step + frequency step
            e_g = np.max(np.abs(g.ind_lkl - em.ind_lkl) / np.abs(em.ind_lkl))
            e_gf = np.max(np.abs(parts[1].freq - em.freq) / np.maximum(em.freq, 1e-3))
            grp = f" group2: lkl {e_g:.1e} freq {e_gf:.1e}"
            for p in parts:
                p.close()
        for k, v in (("lkl", e_l), ("post", e_p), ("freq", e_f)):
            worst[k] = max(worst[k], v)
        # (a posterior within rounding of a snapping threshold may fall on either side: 1e-5)
        flag = "" if (e_l < 1e-9 and (e_p < 1e-9 or abs(e_p - 1e-5) < 1e-9) and e_f < 1e-7 and ok2 is True) else "   <-- CHECK"
        print(f"case {case:2d}: I={I:3d} S={S:4d} chr={nchr} call_geno={int(call)}: lkl {e_l:.1e} post {e_p:.1e} "
              f"freq {e_f:.1e} two iterations {ok2}{grp}{flag}", flush=True)
    print("worst", worst)


if __name__ == "__main__":
    main()

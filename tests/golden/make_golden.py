"""Generates tests/golden/*.npz: small seeded inputs and the outputs of the CPU oracle
(both builds; the libm build run with the REFERENCE'S OWN optimizer object from
oracle/_ref when it is available).  Run from the repo root:

    python tests/golden/make_golden.py

The reference itself cannot be executed for these routines in this image (GSL is
missing; see oracle/ngsfhmm_oracle.h), so the vectors are the oracle's."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib  # noqa: E402

pkg = importlib.import_module("ngsf-hmm_amd")


def em_fixture(name, n_ind, n_sites, iters, **sim):
    d = pkg.simulate.simulate(n_ind, n_sites, **sim)
    gl = pkg.simulate.normalise_log_gl(d.gl)
    out = dict(n_ind=n_ind, n_sites=n_sites, iters=iters, gl=gl, pos_dist=d.pos_dist_mb,
               indF0=np.full(n_ind, 0.1), alpha0=np.full(n_ind, 0.2), freq0=np.full(n_sites, 0.1),
               true_path=d.path)
    for kind in ("libm", "det"):
        orc = orclib.Oracle(kind)
        em = orclib.OracleEM(orc, gl, d.pos_dist_mb)
        em.set_params(out["indF0"], out["alpha0"], out["freq0"])
        if kind == "libm" and orclib.RefBfgs.available():
            em.use_reference_optimizer(orclib.RefBfgs())
        assert em.init_emission() == 0
        out[f"{kind}_eprob0"] = em.e_prob
        for _ in range(iters):
            assert em.iterate() == 0
        out[f"{kind}_indF"] = em.indF
        out[f"{kind}_alpha"] = em.alpha
        out[f"{kind}_freq"] = em.freq
        out[f"{kind}_marg"] = em.marg
        out[f"{kind}_ind_lkl"] = em.ind_lkl
        out[f"{kind}_path"] = em.viterbi()
    path = os.path.join(ROOT, "tests", "golden", name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    orclib.build_oracle()
    em_fixture("em_small.npz", 8, 300, 3, seed=2024, n_chrom=2, missing_rate=0.05)

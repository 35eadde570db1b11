"""The oracle against the committed golden vectors (tests/golden/make_golden.py): guards
the oracle itself against drift, on CPU."""
import os

import numpy as np
import pytest

import orclib


@pytest.mark.parametrize("kind", ["libm", "det"])
def test_oracle_reproduces_golden(kind):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "em_small.npz"))
    orc = orclib.Oracle(kind)
    em = orclib.OracleEM(orc, g["gl"], g["pos_dist"])
    em.set_params(g["indF0"], g["alpha0"], g["freq0"])
    assert em.init_emission() == 0
    assert np.array_equal(em.e_prob, g[f"{kind}_eprob0"])
    for _ in range(int(g["iters"])):
        assert em.iterate() == 0
    for name, val in (("indF", em.indF), ("alpha", em.alpha), ("freq", em.freq),
                      ("marg", em.marg), ("ind_lkl", em.ind_lkl), ("path", em.viterbi())):
        assert np.array_equal(val, g[f"{kind}_{name}"]), name


def test_golden_viterbi_recovers_most_of_the_truth():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "em_small.npz"))
    assert (g["libm_path"] == g["true_path"]).mean() > 0.8


def test_golden_builds_agree_on_the_fixture():
    """Rung 3 of the parity ladder on the committed fixture (8 x 300, 3 iterations): the
    oracle with detmath's exp/log against the oracle with libm's -- the same code, the two
    functions within 1 ulp of each other."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "em_small.npz"))
    for name, tol in (("indF", 1e-14), ("alpha", 1e-14), ("freq", 1e-14), ("marg", 1e-12),
                      ("ind_lkl", 1e-14)):
        a, b = g[f"det_{name}"], g[f"libm_{name}"]
        assert np.max(np.abs(a - b)) <= tol * max(1.0, np.max(np.abs(b))), name
    assert np.array_equal(g["det_path"], g["libm_path"])


def test_det_build_against_libm_build_end_to_end(pkg):
    """Rung 3 as a measured number: exact mode is bit-identical to the oracle's det build,
    the reference calls libm.  The same ten EM iterations with both (40 individuals x 3000
    sites, two chromosomes, uniform site frequencies, 3 % missing cells): exp/log differ in
    the last bit, the finite-difference L-BFGS-B turns that into ~1e-5 on indF (SURVEY finding
    4: the reference differs from itself by as much under another compiler flag), everything
    the optimizer does not touch stays at 1e-9.  Measured here: total log-likelihood 3.4e-10
    relative, indF max 5.1e-6 (median 1.2e-8), alpha median 2.4e-8 relative (one individual
    8.5e-4), freq 3.6e-7, posteriors 2.4e-5, decoded paths identical."""
    I, S, iters = 40, 3000, 10
    d = pkg.simulate.simulate(I, S, seed=1, n_chrom=2, missing_rate=0.03, freq="r")
    gl = pkg.simulate.normalise_log_gl(d.gl)
    res = {}
    for kind in ("det", "libm"):
        em = orclib.OracleEM(orclib.Oracle(kind), gl, d.pos_dist_mb)
        em.set_params(0.1, 0.2, 0.1)
        assert em.init_emission() == 0
        tot = []
        for _ in range(iters):
            assert em.iterate(n_threads=8) == 0
            tot.append(em.ind_lkl.sum())
        res[kind] = dict(tot=np.array(tot), indF=em.indF.copy(), alpha=em.alpha.copy(),
                         freq=em.freq.copy(), marg=em.marg.copy(), path=em.viterbi())
        em.close()
    a, b = res["det"], res["libm"]
    d_lkl = np.max(np.abs(a["tot"] - b["tot"]) / np.abs(b["tot"]))
    d_F = np.abs(a["indF"] - b["indF"])
    d_A = np.abs(a["alpha"] - b["alpha"]) / b["alpha"]
    d_f = np.max(np.abs(a["freq"] - b["freq"]))
    d_m = np.max(np.abs(a["marg"] - b["marg"]))
    print(f"det vs libm, {iters} iterations of {I} x {S}: tot lkl {d_lkl:.2e} rel, indF max "
          f"{d_F.max():.2e} median {np.median(d_F):.2e}, alpha max {d_A.max():.2e} median "
          f"{np.median(d_A):.2e} rel, freq {d_f:.2e}, posteriors {d_m:.2e}, paths differ in "
          f"{(a['path'] != b['path']).mean():.2e} of the cells")
    assert d_lkl < 1e-8
    assert d_F.max() < 1e-4 and np.median(d_F) < 1e-6
    assert np.median(d_A) < 1e-5 and np.quantile(d_A, 0.9) < 1e-3
    assert d_f < 1e-5 and d_m < 1e-3
    assert (a["path"] != b["path"]).mean() < 1e-4

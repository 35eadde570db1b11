"""The oracle's restatement of forward / backward / posteriors / est_maf against an anchor it
shares no code with: oracle/hp_anchor.c evaluates the same model (shared/HMM.cpp:6-60,130-154;
shared/gen_func.cpp:974-1009) in binary128, linear space, Rabiner scaling.

Why this exists: those reference routines cannot be compiled in this image (GSL headers), so
the restatement is pinned to no reference binary.  Agreement with a 113-bit evaluation of the
model shows the restatement computes what the reference's formulas say, up to the rounding
noise of the double-precision log-space formulation -- which these tests also MEASURE, so that
the GPU tests (tests/test_gpu_fast.py::test_fast_mode_against_the_binary128_anchor) can check
that the linear-space kernels are no further from the truth than the reference's own
arithmetic."""
import numpy as np
import pytest

import orclib


@pytest.fixture(scope="module")
def hp():
    orclib.build_oracle()
    return orclib.HpAnchor()


@pytest.fixture(scope="module")
def sim(pkg):
    d = pkg.simulate.simulate(12, 10_000, seed=31, n_chrom=3, missing_rate=0.03, indF="r",
                              freq="r", alpha=0.3)
    return d, pkg.simulate.normalise_log_gl(d.gl)


def test_oracle_estep_within_its_rounding_envelope(orc_libm, hp, sim):
    """10 000 sites: the log-space recursion rounds a quantity of size |Fw| ~ 1e4 at every
    site, so its log-likelihood carries ~S * ulp(|lkl|) of noise: allow 1e-11 relative, and
    1e-9 relative on the (unsnapped-range) posteriors, which inherit exp(Fw + Bw - lkl)."""
    d, gl = sim
    em = orclib.OracleEM(orc_libm, gl, d.pos_dist_mb)
    indF = np.linspace(0.02, 0.9, d.n_ind)
    alpha = np.linspace(0.01, 5.0, d.n_ind)
    freq = np.clip(d.freq, 0.02, 0.98)
    em.set_params(indF, alpha, freq)
    assert em.init_emission() == 0 and em.estep() == 0
    worst_l = worst_p = 0.0
    for i in range(d.n_ind):
        lk, post = hp.forward_backward(gl[:, i], freq, d.pos_dist_mb, indF[i], alpha[i])
        worst_l = max(worst_l, abs(em.ind_lkl[i] - lk) / abs(lk))
        snapped = np.where(post < 1e-5, 0.0, np.where(post > 1 - 1e-5, 1.0, post))
        # cells within 1e-9 of a snapping threshold may legitimately fall on either side
        near = (np.abs(post - 1e-5) < 1e-9) | (np.abs(post - (1 - 1e-5)) < 1e-9)
        err = np.abs(em.marg[i] - snapped)[~near] / np.maximum(snapped[~near], 1e-5)
        worst_p = max(worst_p, err.max())
    print("oracle (libm, log space) vs binary128: lkl rel %.2e, posteriors rel %.2e" %
          (worst_l, worst_p))
    assert worst_l < 1e-11 and worst_p < 1e-9


def test_oracle_est_maf_against_binary128(orc_libm, hp, sim):
    d, gl = sim
    rng = np.random.default_rng(5)
    worst = 0.0
    for s in range(0, d.n_sites, 97):
        F = rng.uniform(0, 1, d.n_ind)
        F[rng.uniform(size=d.n_ind) < 0.3] = 0.0
        F[rng.uniform(size=d.n_ind) < 0.2] = 1.0
        f_o, n_o = orc_libm.est_maf(gl[s], F)
        f_h, n_h = hp.est_maf(gl[s], F)
        assert n_o == n_h                       # same number of passes
        worst = max(worst, abs(f_o - f_h) / f_h)
    print("oracle est_maf vs binary128: max rel %.2e" % worst)
    assert worst < 1e-12


def test_brute_force_agrees_with_the_anchor(hp):
    """The anchor itself against explicit enumeration of all 2^S paths (S = 10)."""
    import itertools
    import math
    rng = np.random.default_rng(0)
    S = 10
    gl = np.log(rng.dirichlet([1, 1, 1], S))
    freq = rng.uniform(0.05, 0.6, S)
    d = rng.uniform(0.01, 2.0, S)
    d[4] = math.inf
    F, al = 0.3, 0.7
    q = [1 - F, F]
    e = np.empty((S, 2))
    for s in range(S):
        f = freq[s]
        p = np.exp(gl[s])
        e[s, 0] = p[0] * (1 - f) ** 2 + p[1] * 2 * f * (1 - f) + p[2] * f * f
        e[s, 1] = p[0] * (1 - f) + p[2] * f
    tot, w1 = 0.0, np.zeros(S)
    for path in itertools.product((0, 1), repeat=S + 1):      # state at virtual site 0 .. S
        pr = q[path[0]]
        for s in range(S):
            c = 0.0 if math.isinf(d[s]) else math.exp(-al * d[s])
            pr *= ((1 - c) * q[path[s + 1]] + (c if path[s] == path[s + 1] else 0.0)) * e[s, path[s + 1]]
        tot += pr
        w1 += pr * np.array(path[1:])
    lk, post = hp.forward_backward(gl, freq, d, F, al)
    assert abs(lk - math.log(tot)) < 1e-13 * abs(lk)
    np.testing.assert_allclose(post, w1 / tot, rtol=1e-12)

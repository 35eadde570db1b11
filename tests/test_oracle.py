"""The C oracle (oracle/ngsfhmm_oracle.c) against closed-form / brute-force answers
and an independent pure-Python restatement (tests/pyref.py).  The reference cannot
be compiled for these routines in this image (they include <gsl/gsl_rng.h>), and it
ships no golden vectors for them, so this is what anchors them (see
oracle/ngsfhmm_oracle.h, "Parity status")."""
import math

import numpy as np
import pytest

import pyref
from orclib import OracleEM


def _rand_case(rng, S, with_inf=True, called=False):
    q1 = rng.uniform(0.05, 0.95)
    q = [1 - q1, q1]
    alpha = rng.uniform(0.01, 3.0)
    pos = rng.uniform(0.01, 2.0, S)
    if with_inf and S > 3:
        pos[S // 2] = math.inf            # a chromosome start
    e = np.log(rng.uniform(1e-4, 1.0, (S, 2)))
    if called:
        e[rng.integers(0, S)] = [-0.3, -1e15]  # impossible emission as the reference encodes it
    return q, alpha, pos, e


@pytest.mark.parametrize("seed", range(6))
def test_forward_is_the_path_sum(orc_libm, seed):
    rng = np.random.default_rng(seed)
    S = int(rng.integers(1, 9))
    q, alpha, pos, e = _rand_case(rng, S)
    rc, lkl, Fw = orc_libm.forward(q, alpha, e, pos)
    assert rc == 0
    assert lkl == pytest.approx(pyref.brute_force_loglik(q, alpha, e.tolist(), pos.tolist()),
                                rel=1e-12, abs=1e-12)
    rcb, lklb, Bw = orc_libm.backward(q, alpha, e, pos)
    assert rcb == 0 and lklb == pytest.approx(lkl, rel=1e-12, abs=1e-11)


@pytest.mark.parametrize("seed", range(4))
def test_posterior_matches_enumeration(orc_libm, seed):
    rng = np.random.default_rng(100 + seed)
    S = 7
    q, alpha, pos, e = _rand_case(rng, S)
    rc, lkl, Fw = orc_libm.forward(q, alpha, e, pos)
    rcb, _, Bw = orc_libm.backward(q, alpha, e, pos)
    post = np.exp(Fw[1:, 1] + Bw[1:, 1] - lkl)
    want = pyref.brute_force_posterior(q, alpha, e.tolist(), pos.tolist())
    np.testing.assert_allclose(post, want, rtol=1e-11)


@pytest.mark.parametrize("seed", range(8))
def test_c_oracle_equals_python_restatement(orc_libm, seed):
    """Same libm, same operation order: forward/backward/viterbi/emission/est_maf of the C
    oracle agree with the pure-Python restatement to the last bit."""
    rng = np.random.default_rng(200 + seed)
    S = int(rng.integers(2, 40))
    q, alpha, pos, e = _rand_case(rng, S, called=(seed % 2 == 1))
    rc, lkl, Fw = orc_libm.forward(q, alpha, e, pos)
    plkl, pFw = pyref.forward(q, alpha, e.tolist(), pos.tolist())
    assert lkl == plkl and Fw.tolist() == pFw
    rc, lklb, Bw = orc_libm.backward(q, alpha, e, pos)
    plklb, pBw = pyref.backward(q, alpha, e.tolist(), pos.tolist())
    assert lklb == plklb and Bw.tolist() == pBw
    _, path = orc_libm.viterbi(q, alpha, e, pos)
    assert path.tolist() == pyref.viterbi(q, alpha, e.tolist(), pos.tolist())


def test_viterbi_inplace_quirk_is_kept(orc_libm):
    """HMM.cpp:104-117 updates Vi_prob[0] before computing state 1 of the same site.
    A case where this changes the path relative to textbook Viterbi pins the quirk."""
    q, alpha = [0.5, 0.5], 50.0
    pos = np.array([1.0, 1.0, 1.0])
    e = np.log(np.array([[0.9, 0.1], [0.1, 0.9], [0.9, 0.1]]))
    _, path = orc_libm.viterbi(q, alpha, e, pos)
    assert path.tolist() == pyref.viterbi(q, alpha, e.tolist(), pos.tolist())


def test_hwe_and_emission(orc_libm):
    for maf in (0.0, 0.01, 0.2, 0.5, 1.0):
        for F in (0.0, 0.3, 1.0):
            np.testing.assert_array_equal(orc_libm.calc_hwe(maf, F), pyref.calc_hwe(maf, F))
            lin = orc_libm.calc_hwe(maf, F, log_scale=False)
            if F != 1.0:
                assert lin.sum() == pytest.approx(1.0, abs=1e-15)
    gl = np.log(np.array([0.7, 0.2, 0.1]))
    for k in (0, 1):
        v, bad = orc_libm.calc_emission(gl, 0.2, k)
        assert not bad and v == pyref.calc_emission(gl.tolist(), 0.2, k)
    h = pyref.calc_hwe(0.2, 0.0, log_scale=False)
    v, _ = orc_libm.calc_emission(gl, 0.2, 0)
    assert math.exp(v) == pytest.approx(0.7 * h[0] + 0.2 * h[1] + 0.1 * h[2], rel=1e-14)
    _, bad = orc_libm.calc_emission(gl, 1.5, 0)     # "invalid MAF!"
    assert bad
    # F == 1: a heterozygote is impossible but kept finite (-1e15), gen_func.cpp:951-956
    assert orc_libm.calc_hwe(0.3, 1.0)[1] == -1e15


@pytest.mark.parametrize("seed", range(4))
def test_est_maf(orc_libm, seed):
    rng = np.random.default_rng(300 + seed)
    I = int(rng.integers(3, 30))
    p = rng.dirichlet([1, 1, 1], I)
    gl = np.log(p)
    F = rng.uniform(0, 1, I)
    F[rng.integers(0, I)] = 1.0       # snapped posteriors occur all the time
    F[rng.integers(0, I)] = 0.0
    f, passes = orc_libm.est_maf(gl, F)
    pf, ppasses = pyref.est_maf(gl.tolist(), F.tolist())
    assert f == pf and passes == ppasses
    assert 1 <= passes <= 101 and 0.0 < f < 1.0


def test_est_maf_quirks(orc_libm):
    """The start value is 0.01 and num/den accumulate ACROSS passes
    (gen_func.cpp:975-1006).  All-missing data: the first pass returns the HWE
    expectation at 0.01, i.e. 0.01 again, and the loop stops after one pass.
    Informative data: the cumulative sums make the iterate a running average, so
    the result differs from the plain EM fixed point and needs many passes."""
    gl = np.log(np.full((1, 3), 1 / 3))
    f, passes = orc_libm.est_maf(gl, np.array([0.0]))
    assert passes == 1 and f == pytest.approx(0.01, abs=1e-15)
    rng = np.random.default_rng(5)
    geno = rng.integers(0, 3, 50)
    gl = np.log(np.full((50, 3), 0.01))
    gl[np.arange(50), geno] = np.log(0.98)
    f, passes = orc_libm.est_maf(gl, np.zeros(50))
    plain = geno.mean() / 2                      # where an ordinary EM would converge (about)
    assert passes > 20                           # slow: running average of the iterates
    assert abs(f - plain) > 1e-4 and abs(f - plain) < 0.05


def test_logsum_edge_cases(orc_libm):
    inf = math.inf
    assert orc_libm.logsum([-inf, -inf]) == -inf
    assert orc_libm.logsum([0.0, -inf]) == 0.0
    assert orc_libm.logsum([-1e15, -1e15]) == pytest.approx(-1e15 + math.log(2), abs=0.2)
    assert math.isnan(orc_libm.logsum([math.nan, 0.0]))


def test_em_iteration_runs_and_increases_likelihood(orc_libm, small_sim):
    d, gl = small_sim
    em = OracleEM(orc_libm, gl, d.pos_dist_mb)
    em.set_params(0.1, 0.2, 0.1)
    assert em.init_emission() == 0
    lk = []
    for _ in range(4):
        assert em.iterate() == 0
        lk.append(em.ind_lkl.sum())
    assert lk[-1] > lk[0]
    assert np.all((em.marg >= 0) & (em.marg <= 1))
    assert em.iterate(freq_est=2) == -5      # the reference aborts: "invalid allele frequencies"
    path = em.viterbi()
    assert path.shape == (d.n_ind, d.n_sites) and set(np.unique(path)) <= {0, 1}
    gp = em.geno_post(path)
    np.testing.assert_allclose(gp.sum(axis=2), 1.0, rtol=1e-12)


def test_threads_do_not_change_results(orc_libm, small_sim):
    """results do not depend on --n_threads."""
    d, gl = small_sim
    out = []
    for nt, tf in ((1, False), (4, False), (4, True)):
        em = OracleEM(orc_libm, gl, d.pos_dist_mb)
        em.set_params(0.1, 0.2, 0.1)
        em.init_emission()
        for _ in range(2):
            assert em.iterate(n_threads=nt, thread_freq=tf) == 0
        out.append((em.indF.tobytes(), em.alpha.tobytes(), em.freq.tobytes(), em.marg.tobytes()))
    assert out[0] == out[1] == out[2]


def test_em_loop_control(orc_libm, small_sim):
    """EM.cpp:56: runs at least min_iters, at most max_iters."""
    d, gl = small_sim
    em = OracleEM(orc_libm, gl, d.pos_dist_mb)
    em.set_params(0.1, 0.2, 0.1)
    em.init_emission()
    assert em.run(min_iters=3, max_iters=4, min_epsilon=1e-5) in (3, 4)
    em2 = OracleEM(orc_libm, gl, d.pos_dist_mb)
    em2.set_params(0.1, 0.2, 0.1)
    em2.init_emission()
    assert em2.run(min_iters=1, max_iters=2, min_epsilon=1e30) >= 1


def test_prepare_gl_against_a_numpy_restatement(orc_libm):
    """orc_prepare_gl (input preparation, read_data.cpp:36-40,89-98; ngsF-HMM.cpp:101-117)
    against the formulas written out in numpy: post_prob = g - logsum(g) twice, call_geno
    (gen_func.cpp:886-914 with its defaults) in between."""
    rng = np.random.default_rng(5)
    raw = np.log(rng.dirichlet([1, 1, 1], size=(40, 7))) + rng.normal(size=(40, 7, 1))
    raw[0, 0] = np.log(1 / 3)

    def post_prob(g):
        m = g.max(-1, keepdims=True)
        return g - (np.log(np.exp(g - m).sum(-1, keepdims=True)) + m)

    a = orc_libm.prepare_gl(raw, 0, False)
    np.testing.assert_allclose(a, post_prob(post_prob(raw)), rtol=0, atol=1e-14)
    b = orc_libm.prepare_gl(raw, 0, True)
    called = post_prob(raw)
    k = called.argmax(-1)
    want = np.full_like(called, -1e15)
    np.put_along_axis(want, k[..., None], 0.0, axis=-1)
    want[0, 0] = np.log(1 / 3)                       # all equal: missing data
    np.testing.assert_allclose(b, post_prob(want), rtol=0, atol=1e-14)
    c = orc_libm.prepare_gl(np.exp(raw), 2, False)   # text reader: plain log first
    np.testing.assert_allclose(c, a, rtol=0, atol=1e-14)
    z = orc_libm.prepare_gl(np.array([[0.0, 0.0, 1.0]]), 1, False)   # binary reader: log 0 -> -1e15
    assert z[0, 2] == 0.0 and z[0, 0] == -1e15

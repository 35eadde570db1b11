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


def test_intended_ld_frequency_step_against_binary128(pkg, orc_libm):
    """--freq_est 2 / --e_prob 2 as INTENDED (opt-in; PARITY UNPINNED: the reference aborts on
    both, SURVEY.md finding 3): the oracle's restatement of the normal-space pair iteration
    (gen_func.cpp:1076-1119 under haplo_freq, :1027-1063) and of calc_emissionLD (HMM.cpp:175-236)
    against the same MODEL in binary128 (oracle/hp_anchor.c: another arrangement of the sums, no
    shared code): haplotype frequencies within 1e-13 with the same iteration count, LD emissions
    within 1e-12, on pairs of sites in linkage disequilibrium and in equilibrium, with missing
    cells and with certain genotypes."""
    hp = orclib.HpAnchor()
    rng = np.random.default_rng(12)
    for n, ld in ((40, 0.0), (200, 0.8), (1000, 0.4), (7, 0.95)):
        # two sites with allele frequencies f1, f2 and a fraction `ld` of haplotypes carrying the
        # same allele at both; genotype probabilities from noisy reads
        f1, f2 = rng.uniform(0.05, 0.6, 2)
        hap = np.empty((n, 2, 2), dtype=int)
        hap[..., 0] = rng.random((n, 2)) < f1
        copy = rng.random((n, 2)) < ld
        hap[..., 1] = np.where(copy, hap[..., 0], rng.random((n, 2)) < f2)
        g = hap.sum(axis=1)                                      # [n][site] in 0..2
        p = np.full((2, n, 3), 0.02)
        for s in range(2):
            p[s, np.arange(n), g[:, s]] = 0.96
        p[:, : n // 10] = 1 / 3                                  # missing individuals
        if n >= 40:
            p[0, -1] = (0.0, 0.0, 1.0)                           # a certain genotype
        p /= p.sum(axis=2, keepdims=True)
        m1, m2 = g[:, 0].mean() / 2 + 1e-3, g[:, 1].mean() / 2 + 1e-3
        h_orc, it_orc = orc_libm.haplo_freq(p[0], p[1], m1, m2)
        h_hp, it_hp = hp.haplo_freq(p[0], p[1], m1, m2)
        assert it_orc == it_hp and 0 < it_orc < 100
        np.testing.assert_allclose(h_orc, h_hp, rtol=1e-13, atol=1e-15)
        assert abs(h_orc.sum() - 1) < 1e-14
        if ld > 0.5:     # the pair is in LD: D = P_ba - maf1 maf2 clearly positive
            assert h_orc[3] - (h_orc[2] + h_orc[3]) * (h_orc[1] + h_orc[3]) > 0.02
        gl = np.log(np.maximum(p, 1e-300))
        for i in (0, n // 2, n - 1):
            for F in (0, 1):
                e_orc = orc_libm.calc_emission_ld(h_orc, gl[0, i], gl[1, i], m1, m2, F)
                e_hp = hp.emission_ld(h_orc, gl[0, i], gl[1, i], m1, F)
                if np.isfinite(e_hp):
                    assert abs(e_orc - e_hp) <= 1e-12 * max(1.0, abs(e_hp)), (n, i, F)
    # no information (every genotype equally likely everywhere): the starting point, the
    # product of the marginal frequencies, is a fixed point and the loop ends at once
    p = np.full((50, 3), 1 / 3)
    h, it = orc_libm.haplo_freq(p, p, 0.3, 0.2)
    np.testing.assert_allclose(h, [0.56, 0.14, 0.24, 0.06], rtol=1e-12)
    assert it == 0


def test_intended_ld_m_step_of_the_oracle(pkg, orc_libm):
    """orc_em_mstep_freq_ld (EM.cpp:210-272 as intended): site 1 takes est_maf as the
    reference's own `freq_est == 1 || s == 1` says, every later site the haplotype route with
    the NEW frequency of its predecessor (the loop updates freq in place); with --e_prob 2 the
    emissions of s > 1 come from calc_emissionLD.  Checked against a step-by-step recomputation
    in Python from the oracle's scalar routines."""
    I, S = 9, 40
    d = pkg.simulate.simulate(I, S, seed=5, missing_rate=0.1)
    gl = pkg.simulate.normalise_log_gl(d.gl)
    for freq_est, e_prob in ((2, 1), (2, 2), (1, 2)):
        em = orclib.OracleEM(orc_libm, gl, d.pos_dist_mb)
        em.set_params(0.2, 0.1, 0.15)
        assert em.init_emission() == 0 and em.estep() == 0
        marg, freq0 = em.marg.copy(), em.freq.copy()
        assert em.mstep_freq_ld(freq_est, e_prob) == 0
        freq = freq0.copy()
        for s in range(S):
            post = np.empty((2, I, 3))
            for i in range(I):
                for j, ss in enumerate((s - 1, s)):
                    if ss < 0:
                        continue
                    prior = orc_libm.calc_hwe(freq[ss], marg[i, ss], True)
                    pp = orc_libm.post_prob(gl[ss, i], prior)
                    post[j, i] = [orc_libm.lib.orc_exp(float(v)) for v in pp]   # the C library's exp
            if s >= 1:
                hap, _ = orc_libm.haplo_freq(post[0], post[1], freq[s - 1], freq[s])
            if freq_est == 1 or s == 0:
                freq[s] = orc_libm.est_maf(gl[s], marg[:, s])[0]
            else:
                freq[s] = hap[1] + hap[3]
            for i in range(I):
                for k in range(2):
                    if e_prob == 1 or s == 0:
                        want = orc_libm.calc_emission(gl[s, i], freq[s], k)[0]
                    else:
                        want = orc_libm.calc_emission_ld(hap, gl[s - 1, i], gl[s, i], freq[s - 1],
                                                         freq[s], k)
                    assert em.e_prob[i, s, k] == want, (freq_est, e_prob, s, i, k)
        assert np.array_equal(em.freq, freq)
        assert np.all((freq > 0) & (freq < 1))
        em.close()

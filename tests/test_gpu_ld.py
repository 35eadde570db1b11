"""--freq_est 2 / --e_prob 2 AS INTENDED (opt-in; PARITY UNPINNED: the reference aborts on both
at the first site, EM.cpp:235-238 -> shared/gen_func.cpp:1030-1031, SURVEY.md finding 3).

There is no reference output for this path.  What is tested is that the GPU computes the
oracle's restatement of what the code evidently means (oracle/ngsfhmm_oracle.c:
orc_em_mstep_freq_ld -- the loop of EM.cpp:224-263 as written, minus its three defects; itself
checked against binary128 in tests/test_hp_anchor.py): exact mode bit for bit, fast mode within
1e-9, and that plain --freq_est 2 keeps returning the reference's abort."""
import numpy as np
import pytest

import orclib
from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _data(pkg, I, S, seed):
    d = pkg.simulate.simulate(I, S, seed=seed, n_chrom=2, missing_rate=0.05, freq="r")
    return d, pkg.simulate.normalise_log_gl(d.gl)


@pytest.mark.parametrize("I,S", [(9, 300), (70, 257), (1100, 60), (2100, 24)])
@pytest.mark.parametrize("freq_est,e_prob", [(2, 1), (2, 2), (1, 2)])
def test_exact_mode_is_the_oracle_bit_for_bit(pkg, orc_det, I, S, freq_est, e_prob):
    """One E-step, then the intended frequency step: frequencies and emissions equal the
    oracle's det build in every bit -- 9 .. 2100 individuals: one to three individuals per
    thread of the chain's workgroup, sums in individual order across its chunks."""
    d, gl = _data(pkg, I, S, seed=I)
    em = orclib.OracleEM(orc_det, gl, d.pos_dist_mb)
    em.set_params(0.2, 0.1, 0.15)
    assert em.init_emission() == 0 and em.estep() == 0
    assert em.mstep_freq_ld(freq_est, e_prob) == 0
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_EXACT) as h:
        h.load(gl, d.pos_dist_mb)
        h.set_params(0.2, 0.1, 0.15)
        h.init_emission()
        h.estep()
        h.mstep_freq(freq_est | pkg.LD_INTENDED | (pkg.EPROB_LD if e_prob == 2 else 0))
        assert np.array_equal(h.freq, em.freq)
        assert np.array_equal(h.e_prob, em.e_prob)
    assert np.all((em.freq > 0) & (em.freq < 1))
    em.close()


def test_exact_mode_whole_iterations(pkg, orc_det):
    """Three EM iterations with the intended --freq_est 2 --e_prob 2 through nghmm_iter_em: the
    whole trajectory equals the oracle's (E-step, L-BFGS-B M-step, intended frequency step)."""
    I, S = 12, 500
    d, gl = _data(pkg, I, S, seed=3)
    em = orclib.OracleEM(orc_det, gl, d.pos_dist_mb)
    em.set_params(0.1, 0.2, 0.1)
    assert em.init_emission() == 0
    flag = 2 | pkg.LD_INTENDED | pkg.EPROB_LD
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_EXACT) as h:
        h.load(gl, d.pos_dist_mb)
        h.set_params(0.1, 0.2, 0.1)
        h.init_emission()
        for it in range(3):
            assert em.estep() == 0 and em.mstep_indf() == 0 and em.mstep_freq_ld(2, 2) == 0
            h.iter_EM(flag)
            for name in ("ind_lkl", "indF", "alpha", "freq"):
                assert np.array_equal(getattr(h, name), getattr(em, name)), (it, name)
            assert np.array_equal(h.e_prob, em.e_prob), it
        assert np.array_equal(h.viterbi(), em.viterbi())
    em.close()


@pytest.mark.parametrize("I,S", [(40, 400), (1500, 50)])
def test_fast_mode_within_1e9(pkg, orc_libm, I, S):
    d, gl = _data(pkg, I, S, seed=7 + I)
    em = orclib.OracleEM(orc_libm, gl, d.pos_dist_mb)
    em.set_params(0.2, 0.1, 0.15)
    assert em.init_emission() == 0 and em.estep() == 0
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as h:
        h.load(gl, d.pos_dist_mb)
        h.set_params(0.2, 0.1, 0.15)
        h.init_emission()
        h.estep()
        post = h.marg_prob
        np.testing.assert_allclose(post, em.marg, rtol=1e-9, atol=1e-12)
        h.mstep_freq(2 | pkg.LD_INTENDED)
        assert em.mstep_freq_ld(2, 1) == 0
        # the chain feeds every site's result into the next: compare at 1e-9 all the same (a
        # frequency that the pair EM drives to ~1e-9 itself is compared absolutely)
        np.testing.assert_allclose(h.freq, em.freq, rtol=1e-9, atol=1e-15)
        np.testing.assert_allclose(np.exp(h.e_prob), np.exp(em.e_prob), rtol=1e-9, atol=1e-300)
        with pytest.raises(pkg.NgsFHMMError) as ei:       # LD emissions need exact mode
            h.mstep_freq(2 | pkg.LD_INTENDED | pkg.EPROB_LD)
        assert ei.value.code == -10
    em.close()


def test_plain_freq_est_2_still_aborts_like_the_reference(pkg):
    I, S = 6, 100
    d, gl = _data(pkg, I, S, seed=1)
    for mode in (pkg.MODE_EXACT, pkg.MODE_FAST):
        with pkg.NgsFHMM(I, S, mode=mode) as h:
            h.load(gl, d.pos_dist_mb)
            h.set_params(0.1, 0.2, 0.1)
            h.init_emission()
            h.estep()
            with pytest.raises(pkg.NgsFHMMError) as ei:
                h.mstep_freq(2)
            assert ei.value.code == -5 and "invalid allele frequencies" in str(ei.value)
            with pytest.raises(pkg.NgsFHMMError) as ei:
                h.mstep_freq(3 | pkg.LD_INTENDED)
            assert ei.value.code == -10
            # a frequency outside [0, 1] is haplo_freq's own abort (gen_func.cpp:1030-1031)
            f = np.full(S, 0.2)
            f[40] = 1.5
            h.set_params(None, None, f)
            with pytest.raises(pkg.NgsFHMMError) as ei:
                h.mstep_freq(2 | pkg.LD_INTENDED)
            assert ei.value.code == -5


def test_cli_ld_intended(pkg, orc_det, tmp_path):
    """The host binary: --freq_est 2 / --e_prob 2 abort like the reference unless --ld_intended is
    given; with it (exact mode) the output files are, byte for byte, those of the oracle's EM
    loop (EM.cpp:27-103) with the intended frequency step."""
    import math
    import cli_util
    I, S = 8, 400
    d = pkg.simulate.simulate(I, S, seed=2024, n_chrom=2)
    paths = cli_util.write_inputs(str(tmp_path), d, d.gl)
    common = ["--geno", paths["glf_bin"], "--loglkl", "--pos", paths["pos_gz"], "--n_ind", I,
              "--n_sites", S, "--freq", 0.1, "--indF", "0.1,0.2", "--min_iters", 3, "--max_iters", 4,
              "--mode", "exact", "--seed", 1, "--verbose", 1, "--freq_est", 2, "--e_prob", 2]
    r = cli_util.run_cli(common + ["--out", str(tmp_path / "abort")], check=False)
    assert r.returncode != 0 and "invalid allele frequencies" in (r.stdout + r.stderr)
    out = str(tmp_path / "ld")
    r = cli_util.run_cli(common + ["--ld_intended", "--out", out])
    assert "parity unpinned" in (r.stdout + r.stderr)

    gl = orc_det.prepare_gl(d.gl, 0)
    em = orclib.OracleEM(orc_det, gl, d.pos_dist_mb)
    em.set_params(0.1, 0.2, 0.1)
    assert em.init_emission() == 0
    tot = prev_tot = 0.0
    prev_ind = np.full(I, -math.inf)
    max_eps, it = -math.inf, 0
    while (prev_tot - tot > 1e-5 or max_eps > 1e-5 or it < 3) and it < 4:      # EM.cpp:33-35
        it += 1
        assert em.estep() == 0 and em.mstep_indf() == 0 and em.mstep_freq_ld(2, 2) == 0
        prev_tot, tot = tot, 0.0
        for v in em.ind_lkl:
            tot += float(v)
        with np.errstate(invalid="ignore"):
            eps = (em.ind_lkl - prev_ind) / np.abs(prev_ind)
        best, mx = 0, -math.inf
        for i, e in enumerate(eps):                  # array_max_pos, gen_func.cpp:73-84
            if e > mx:
                best, mx = i, e
        max_eps = eps[best]
        prev_ind = em.ind_lkl.copy()
    assert f"Iteration {it}:" in r.stdout and f"Iteration {it + 1}:" not in r.stdout
    path = em.viterbi()
    f_indF, f_ibd, f_geno = cli_util.expected_files(tot, em.indF, em.alpha, em.freq, em.ind_lkl,
                                                    path, em.marg, em.geno_post(path))
    assert open(out + ".indF", "rb").read() == f_indF
    assert open(out + ".ibd", "rb").read() == f_ibd
    assert open(out + ".geno", "rb").read() == f_geno

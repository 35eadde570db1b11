"""GPU parity of FAST mode (linear-space, chunk-parallel kernels) against the oracle.

Fast mode re-associates the recursions (products of 2x2 operators) and works in
linear space, so it cannot be bitwise; the bar is BASELINE.json's: per-call
log-likelihoods, posteriors and frequencies within 1e-9 relative of the reference
arithmetic (oracle, libm build), Viterbi paths identical for identical parameters.
End-to-end indF/alpha after several EM iterations are compared at the spread the
reference shows against itself under a different libm/FMA build (SURVEY.md
finding 4: ~1e-5 absolute), because the finite-difference L-BFGS-B amplifies
last-bit noise."""
import numpy as np
import pytest

import orclib
from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

RTOL = 1e-9


def _pair(pkg, orc, gl, pos, indF=0.1, alpha=0.2, freq=0.1):
    S, I = gl.shape[0], gl.shape[1]
    em = orclib.OracleEM(orc, gl, pos)
    em.set_params(indF, alpha, freq)
    hmm = pkg.NgsFHMM(I, S, device=0, mode=pkg.MODE_FAST)
    hmm.load(gl, pos)
    hmm.set_params(indF, alpha, freq)
    return hmm, em


@pytest.fixture(scope="module")
def mid_sim(pkg):
    d = pkg.simulate.simulate(37, 5000, seed=99, n_chrom=3, missing_rate=0.03, indF="r",
                              alpha=0.3, freq="r")
    return d, pkg.simulate.normalise_log_gl(d.gl)


def test_fast_emission(pkg, orc_libm, mid_sim):
    d, gl = mid_sim
    hmm, em = _pair(pkg, orc_libm, gl, d.pos_dist_mb, freq=np.linspace(0.0, 1.0, d.n_sites))
    em.init_emission(); hmm.init_emission()
    np.testing.assert_allclose(np.exp(hmm.e_prob), np.exp(em.e_prob), rtol=RTOL, atol=1e-300)
    hmm.close()


def test_fast_estep(pkg, orc_libm, mid_sim):
    d, gl = mid_sim
    hmm, em = _pair(pkg, orc_libm, gl, d.pos_dist_mb, indF=np.linspace(0.01, 0.95, d.n_ind),
                    alpha=np.linspace(0.01, 8, d.n_ind))
    em.init_emission(); hmm.init_emission()
    assert em.estep() == 0
    lk = hmm.estep()
    np.testing.assert_allclose(lk, em.ind_lkl, rtol=1e-12)
    np.testing.assert_allclose(hmm.marg_prob, em.marg, rtol=RTOL, atol=1e-12)
    hmm.close()


def test_fast_lkl_batch(pkg, orc_libm, mid_sim):
    d, gl = mid_sim
    hmm, em = _pair(pkg, orc_libm, gl, d.pos_dist_mb)
    em.init_emission(); hmm.init_emission()
    rng = np.random.default_rng(3)
    n = 500
    ind = rng.integers(0, d.n_ind, n)          # unsorted, > 5 points for some individuals
    F = rng.uniform(1e-15, 1 - 1e-15, n)
    A = rng.uniform(1e-15, 10, n)
    F[:4] = [1e-15, 1 - 1e-15, 0.5, 1e-6]; A[:4] = [1e-15, 10.0, 1e-15, 10.0]
    got = hmm.lkl(ind, F, A)
    e = em.e_prob
    want = np.array([-orc_libm.lkl([F[p], A[p]], e[ind[p]], d.pos_dist_mb) for p in range(n)])
    np.testing.assert_allclose(got, want, rtol=1e-12)
    # the finite-difference pattern: 5 points per individual, three sharing alpha
    ind5 = np.repeat(np.arange(d.n_ind), 5)
    eh = 4e-6
    F5 = np.tile([0.3, 0.3 + eh, 0.3 - eh, 0.3, 0.3], d.n_ind)
    A5 = np.tile([0.7, 0.7, 0.7, 0.7 + eh, 0.7 - eh], d.n_ind)
    got5 = hmm.lkl(ind5, F5, A5)
    want5 = np.array([-orc_libm.lkl([F5[p], A5[p]], e[ind5[p]], d.pos_dist_mb)
                      for p in range(len(ind5))])
    np.testing.assert_allclose(got5, want5, rtol=1e-12)
    # finite differences themselves (what the optimizer consumes) to 1e-5 relative
    g_got = (got5[1::5] - got5[2::5]) / (2 * eh)
    g_want = (want5[1::5] - want5[2::5]) / (2 * eh)
    np.testing.assert_allclose(g_got, g_want, rtol=1e-4, atol=1e-3)
    hmm.close()


def test_fast_mstep_freq(pkg, orc_libm, mid_sim):
    d, gl = mid_sim
    hmm, em = _pair(pkg, orc_libm, gl, d.pos_dist_mb)
    em.init_emission(); hmm.init_emission()
    em.estep(); hmm.estep()
    assert em.mstep_freq(1) == 0
    hmm.mstep_freq(1)
    np.testing.assert_allclose(hmm.freq, em.freq, rtol=RTOL)
    np.testing.assert_allclose(np.exp(hmm.e_prob), np.exp(em.e_prob), rtol=RTOL, atol=1e-300)
    hmm.close()


def test_fast_whole_em_vs_oracle(pkg, orc_libm, mid_sim):
    """Four free EM iterations of fast mode against the oracle's libm build -- the reference's
    arithmetic -- on 37 individuals x 5000 sites in three chromosomes.  From iteration 2 on the
    likelihoods are evaluated at optimizer outputs that already differ in their last digits
    (the finite-difference L-BFGS-B amplifies 1e-13 on the objective), so beyond the first
    iteration this is a trajectory check with the bars of
    test_fast_mode_end_to_end_against_exact_mode's kind, at what four iterations leave room
    for: total log-likelihood within 1e-9 relative at every iteration, indF within 1e-6
    (median 1e-7), alpha within 2e-6 relative for 90 % (median 5e-7), frequencies within 3e-7,
    posteriors within 1e-5, decoded paths identical."""
    d, gl = mid_sim
    hmm, em = _pair(pkg, orc_libm, gl, d.pos_dist_mb)
    em.init_emission(); hmm.init_emission()
    worst_tot = 0.0
    for it in range(4):
        assert em.iterate() == 0
        hmm.iter_EM()
        if it == 0:      # the same parameters on both sides: a per-call comparison
            np.testing.assert_allclose(hmm.ind_lkl, em.ind_lkl, rtol=1e-12)
        worst_tot = max(worst_tot, abs(hmm.ind_lkl.sum() - em.ind_lkl.sum()) / abs(em.ind_lkl.sum()))
    dF = np.abs(hmm.indF - em.indF)
    dA = np.abs(hmm.alpha - em.alpha) / np.abs(em.alpha)
    dfreq = np.abs(hmm.freq - em.freq).max()
    dm = np.abs(hmm.marg_prob - em.marg).max()
    vp, op = hmm.viterbi(), em.viterbi()
    print("fast vs oracle(libm), 4 iterations at %d x %d: tot_lkl rel max %.1e; indF max %.1e median "
          "%.1e; alpha rel p90 %.1e median %.1e; freq max %.1e; posteriors max %.1e; paths "
          "differing %d" % (d.n_ind, d.n_sites, worst_tot, dF.max(), np.median(dF),
                            np.quantile(dA, 0.9), np.median(dA), dfreq, dm, int((vp != op).sum())))
    # measured: tot_lkl 6.5e-10, indF max 5.1e-8 (median 3.3e-9), alpha p90 9.2e-8 (median
    # 1.8e-8), freq 1.3e-8, posteriors 6.8e-7, no path cell differs
    assert worst_tot < 1e-9
    assert dF.max() < 1e-6 and np.median(dF) < 1e-7
    assert np.quantile(dA, 0.9) < 2e-6 and np.median(dA) < 5e-7
    assert dfreq < 3e-7 and dm < 1e-5
    assert np.array_equal(vp, op)
    hmm.close()


def test_fast_viterbi_identical_for_identical_parameters(pkg, orc_det, mid_sim):
    """Decoding uses the exact-mode kernel: with the same parameters the path is the
    reference's path bit for bit, whatever mode the EM ran in."""
    d, gl = mid_sim
    hmm, em = _pair(pkg, orc_det, gl, d.pos_dist_mb, indF=0.4, alpha=0.05, freq=0.2)
    em.init_emission(); hmm.init_emission()
    assert np.array_equal(hmm.viterbi(), em.viterbi())
    hmm.close()


def test_fast_teacher_forced_iterations(pkg, orc_libm, mid_sim):
    """Feed the oracle's state of iteration t, compare the outputs of iteration t+1
    before the optimizer can amplify anything: E-step and frequency step at 1e-9."""
    d, gl = mid_sim
    hmm, em = _pair(pkg, orc_libm, gl, d.pos_dist_mb)
    em.init_emission()
    for it in range(3):
        hmm.set_params(em.indF, em.alpha, em.freq)
        hmm.init_emission()
        assert em.estep() == 0
        lk = hmm.estep()
        np.testing.assert_allclose(lk, em.ind_lkl, rtol=1e-12)
        np.testing.assert_allclose(hmm.marg_prob, em.marg, rtol=RTOL, atol=1e-12)
        assert em.mstep_indf() == 0
        assert em.mstep_freq(1) == 0
        hmm.mstep_freq(1)
        np.testing.assert_allclose(hmm.freq, em.freq, rtol=RTOL)
    hmm.close()


@pytest.mark.parametrize("shape", [(1, 1), (1, 70), (3, 17), (65, 1030), (130, 2049)])
def test_fast_ragged_shapes(pkg, orc_libm, shape):
    I, S = shape
    d = pkg.simulate.simulate(I, S, seed=I * 1000 + S, missing_rate=0.1, n_chrom=2 if S > 10 else 1)
    gl = pkg.simulate.normalise_log_gl(d.gl)
    hmm, em = _pair(pkg, orc_libm, gl, d.pos_dist_mb)
    em.init_emission(); hmm.init_emission()
    assert em.estep() == 0
    lk = hmm.estep()
    np.testing.assert_allclose(lk, em.ind_lkl, rtol=1e-11)
    np.testing.assert_allclose(hmm.marg_prob, em.marg, rtol=RTOL, atol=1e-12)
    em.mstep_freq(1); hmm.mstep_freq(1)
    np.testing.assert_allclose(hmm.freq, em.freq, rtol=RTOL)
    st = hmm.mstep_indf()
    assert st.rounds >= 1
    hmm.close()


def test_fast_called_genotypes(pkg, orc_libm):
    """One-hot GLs (-1e15 elsewhere): impossible emissions are exact zeros in linear space."""
    d = pkg.simulate.simulate(12, 900, seed=77, n_chrom=4)
    geno = d.geno.copy()
    geno[::17, ::3] = -1
    gl = pkg.simulate.called_genotype_gl(geno)
    hmm, em = _pair(pkg, orc_libm, gl, d.pos_dist_mb, indF=0.5, alpha=0.01, freq=0.2)
    em.init_emission(); hmm.init_emission()
    assert em.estep() == 0
    lk = hmm.estep()
    np.testing.assert_allclose(lk, em.ind_lkl, rtol=1e-11)
    np.testing.assert_allclose(hmm.marg_prob, em.marg, rtol=RTOL, atol=1e-12)
    em.mstep_freq(1); hmm.mstep_freq(1)
    np.testing.assert_allclose(hmm.freq, em.freq, rtol=RTOL)
    hmm.close()


def test_fast_reference_fatal_errors(pkg):
    """The reference's fatal conditions surface as the same messages in fast mode."""
    import math
    d = pkg.simulate.simulate(9, 300, seed=4)
    gl = pkg.simulate.normalise_log_gl(d.gl)
    hmm = pkg.NgsFHMM(9, 300, mode=pkg.MODE_FAST)
    hmm.load(gl, d.pos_dist_mb)
    hmm.set_params(0.1, 0.2, -0.2)
    with pytest.raises(pkg.NgsFHMMError) as ei:
        hmm.init_emission()
    assert ei.value.code == -3 and "invalid MAF!" in ei.value.message
    hmm.set_params(0.1, 0.2, 0.1)
    hmm.init_emission()
    hmm.estep()
    with pytest.raises(pkg.NgsFHMMError) as ei:
        hmm.mstep_freq(2)
    assert ei.value.code == -5
    bad = gl.copy()
    bad[7, 3, :] = math.nan
    hmm.load(bad, d.pos_dist_mb)
    hmm.init_emission()
    with pytest.raises(pkg.NgsFHMMError) as ei:
        hmm.estep()
    assert ei.value.code in (-1, -4)
    hmm.close()


@pytest.mark.parametrize("fixed", [(False, False), (True, False), (False, True), (True, True)])
def test_fast_estep_mstep_shares_the_forward_walk(pkg, orc_libm, mid_sim, fixed):
    """nghmm_estep_mstep runs the M-step's first objective round first and lets it leave
    the E-step's forward walk (lane operators, checkpoints) behind.  Its E-step must be
    the oracle's (1e-12 / 1e-9 as for nghmm_estep), its M-step that of nghmm_estep +
    nghmm_mstep_indf on a second handle, and the after-E-step hook runs exactly once,
    when the posteriors are final."""
    d, gl = mid_sim
    indF = np.linspace(0.01, 0.95, d.n_ind)
    alpha = np.linspace(0.01, 8, d.n_ind)
    a, em = _pair(pkg, orc_libm, gl, d.pos_dist_mb, indF=indF, alpha=alpha)
    b, _ = _pair(pkg, orc_libm, gl, d.pos_dist_mb, indF=indF, alpha=alpha)
    em.init_emission(); a.init_emission(); b.init_emission()
    assert em.estep() == 0
    seen = []
    st = a.estep_mstep(*fixed, after_estep=lambda: seen.append(a.marg_prob.copy()))
    assert len(seen) == 1
    np.testing.assert_allclose(a.ind_lkl, em.ind_lkl, rtol=1e-12)
    np.testing.assert_allclose(seen[0], em.marg, rtol=RTOL, atol=1e-12)
    np.testing.assert_array_equal(a.marg_prob, seen[0])      # later rounds leave them alone
    b.estep()
    st_b = b.mstep_indf(*fixed)
    np.testing.assert_allclose(a.ind_lkl, b.ind_lkl, rtol=1e-13)
    np.testing.assert_allclose(a.marg_prob, b.marg_prob, rtol=RTOL, atol=1e-10)
    # same optimiser, objective values equal to rounding: same path to within its own noise
    np.testing.assert_allclose(a.indF, b.indF, atol=2e-4)
    np.testing.assert_allclose(a.alpha, b.alpha, rtol=2e-2, atol=2e-4)
    if fixed[0]:
        np.testing.assert_array_equal(a.indF, indF)
    if fixed[1]:
        np.testing.assert_array_equal(a.alpha, alpha)
    assert (st.rounds == 0) == all(fixed)
    assert abs(int(st_b.rounds) - int(st.rounds)) <= 2
    # the frequency step reads the posteriors the fused E-step wrote
    a.mstep_freq(1); b.mstep_freq(1)
    np.testing.assert_allclose(a.freq, b.freq, rtol=1e-9)
    a.close(); b.close()


def test_fast_hook_exception_propagates(pkg, orc_libm, mid_sim):
    d, gl = mid_sim
    a, _ = _pair(pkg, orc_libm, gl, d.pos_dist_mb)
    a.init_emission()

    def boom():
        raise ValueError("from the hook")

    with pytest.raises(ValueError, match="from the hook"):
        a.estep_mstep(after_estep=boom)
    a.close()


@pytest.mark.parametrize("alpha0", [0.01, 0.7, 9.99])
@pytest.mark.parametrize("pattern", ["2F2A", "1F2A", "2F1A", "1F1A", "2F0A", "0F2A"])
def test_fast_objective_kernel_variants(pkg, orc_libm, mid_sim, pattern, alpha0):
    """One objective kernel exists per finite-difference pattern of shared/bfgs.cpp:22-43
    (two-sided / one-sided / absent probes of F and of alpha), each in a version for
    alpha * d_max <= 2^-6 (polynomial exp) and a general one.  Every one of them against the
    oracle's forward log-likelihood at 1e-12, and their finite differences -- what the
    optimizer consumes -- against the oracle's.  alpha0 = 0.01 takes the small-argument
    versions on this data (d_max ~ 0.3 Mb), 0.7 and 9.99 the general ones; 9.99 + 2 eh > 10
    is the one-sided pattern at the upper bound."""
    d, gl = mid_sim
    hmm, em = _pair(pkg, orc_libm, gl, d.pos_dist_mb)
    em.init_emission(); hmm.init_emission()
    nf, na = int(pattern[0]), int(pattern[2])
    F0 = 0.3
    ehF = (1e-8 * (abs(F0) + 1)) ** 0.67
    ehA = (1e-8 * (abs(alpha0) + 1)) ** 0.67
    pts = [(F0, alpha0)]
    pts += [(F0 + ehF, alpha0), (F0 - ehF, alpha0)][:nf] if nf == 2 else [(F0 + 2 * ehF, alpha0)][:nf]
    pts += [(F0, alpha0 + ehA), (F0, alpha0 - ehA)][:na] if na == 2 else [(F0, alpha0 - 2 * ehA)][:na]
    npt = len(pts)
    ind = np.repeat(np.arange(d.n_ind), npt)
    F = np.tile([p[0] for p in pts], d.n_ind)
    A = np.tile([p[1] for p in pts], d.n_ind)
    got = hmm.lkl(ind, F, A)
    e = em.e_prob
    want = np.array([-orc_libm.lkl([F[p], A[p]], e[ind[p]], d.pos_dist_mb) for p in range(len(ind))])
    np.testing.assert_allclose(got, want, rtol=1e-12)
    for k in range(1, npt):                       # every probe's difference from f(x)
        np.testing.assert_allclose(got[k::npt] - got[0::npt], want[k::npt] - want[0::npt],
                                   rtol=1e-4, atol=2e-8)
    hmm.close()


@pytest.mark.parametrize("alpha0", [1e-15, 1e-6, 2e-5, 0.01, 0.04])
def test_fast_kappa_form_at_chromosome_starts(pkg, orc_libm, mid_sim, alpha0):
    """The small-alpha objective kernels keep their operators divided by exp(-alpha d) (kappa
    form, fast_dev.hpp: op_step_k) and meet c = 0 at chromosome starts through two clamps instead
    of a select.  A data set full of starts -- every seventh site over a stretch, eight in a row
    (a whole block between two rescales), pairs, the very last site, next to the three the
    simulation has -- at alpha from its lower bound (EM.cpp:427) up to the small-argument limit.
    Below alpha * mean distance = 1e-6 (here alpha < ~1e-5) the general-exp kernel takes the points:
    there the reference's 1 - exp(-alpha d) is mostly the rounding of exp(-alpha d), which the kappa
    form's exact expm1 does not reproduce (0.015 of log-likelihood at alpha = 1e-15, measured;
    fast_dev.hpp: fd_pattern).  Every point against the oracle at 1e-12, the probes'
    differences against the oracle's, the fused round's E-step log-likelihood (its lane
    operators come out of the same walk) and posteriors too."""
    d, gl = mid_sim
    pos = d.pos_dist_mb.copy()
    starts = list(range(1000, 1500, 7)) + list(range(2000, 2008)) + [3000, 3001, 3500, 3502, d.n_sites - 1]
    pos[starts] = np.inf
    F0 = 0.3
    hmm, em = _pair(pkg, orc_libm, gl, pos, indF=F0, alpha=alpha0)
    em.init_emission(); hmm.init_emission()
    ehF = (1e-8 * (abs(F0) + 1)) ** 0.67
    ehA = (1e-8 * (abs(alpha0) + 1)) ** 0.67
    pts = [(F0, alpha0), (F0 + ehF, alpha0), (F0 - ehF, alpha0)]
    pts += [(F0, alpha0 + ehA), (F0, alpha0 - ehA)] if alpha0 - ehA > 1e-15 else [(F0, alpha0 + 2 * ehA)]
    npt = len(pts)
    ind = np.repeat(np.arange(d.n_ind), npt)
    F = np.tile([p[0] for p in pts], d.n_ind)
    A = np.tile([p[1] for p in pts], d.n_ind)
    got = hmm.lkl(ind, F, A)
    e = em.e_prob
    want = np.array([-orc_libm.lkl([F[p], A[p]], e[ind[p]], pos) for p in range(len(ind))])
    np.testing.assert_allclose(got, want, rtol=1e-12)
    for k in range(1, npt):
        np.testing.assert_allclose(got[k::npt] - got[0::npt], want[k::npt] - want[0::npt],
                                   rtol=1e-4, atol=2e-8)
    assert em.estep() == 0
    hmm.estep_mstep(True, True)          # the emitting walk alone: nothing to optimise
    np.testing.assert_allclose(hmm.ind_lkl, em.ind_lkl, rtol=1e-12)
    np.testing.assert_allclose(hmm.marg_prob, em.marg, rtol=RTOL, atol=1e-12)
    st = hmm.estep_mstep()               # and with the rounds behind it
    assert st.rounds >= 1 and np.all(np.isfinite(hmm.indF)) and np.all(np.isfinite(hmm.alpha))
    hmm.close()


def test_fast_fused_walk_through_the_general_kernel(pkg, orc_libm):
    """A data set with one very long finite distance (2000 Mb): the alpha probes' exp(-+ eh d)
    shortcut does not hold there (|eh d| > 1e-3), so every group takes the general objective
    kernel -- which must then also play the E-step's forward walk and refresh the emissions
    (nghmm_estep_mstep after a frequency update)."""
    d = pkg.simulate.simulate(21, 3000, seed=5, n_chrom=2, missing_rate=0.02)
    gl = pkg.simulate.normalise_log_gl(d.gl)
    pos = d.pos_dist_mb.copy()
    pos[1234] = 2000.0
    a, em = _pair(pkg, orc_libm, gl, pos, indF=0.2, alpha=0.5, freq=0.2)
    em.init_emission(); a.init_emission()
    for it in range(2):
        assert em.estep() == 0
        st = a.estep_mstep()
        # (5e-12: the log-space oracle's own rounding over 3000 sites is ~1e-12 here)
        np.testing.assert_allclose(a.ind_lkl, em.ind_lkl, rtol=5e-12)
        np.testing.assert_allclose(a.marg_prob, em.marg, rtol=RTOL, atol=1e-12)
        # continue the oracle from the GPU's parameters (teacher forcing), then both update
        # the frequencies and, lazily on the GPU, the emissions
        em.set_params(a.indF, a.alpha, None)
        assert em.mstep_freq(1) == 0
        a.mstep_freq(1)
        np.testing.assert_allclose(a.freq, em.freq, rtol=RTOL)
    a.close()


@pytest.mark.parametrize("shape", [(1, 1), (1, 70), (3, 17), (65, 1030), (130, 2049), (1025, 300),
                                   (4100, 40)])
def test_fast_ragged_shapes_fused_iteration(pkg, orc_libm, shape):
    """Two whole fast-mode EM iterations (shared forward walk, lazily refreshed emissions,
    tile-major posteriors read by est_maf, with one and with two waves per site) on
    shapes that do not fill waves, lanes or checkpoint blocks (4100 individuals: est_maf
    through the site-major copy): E-step and frequency step of each iteration against the
    oracle continued from the GPU's indF/alpha."""
    I, S = shape
    d = pkg.simulate.simulate(I, S, seed=I * 1000 + S + 1, missing_rate=0.1,
                              n_chrom=2 if S > 10 else 1)
    gl = pkg.simulate.normalise_log_gl(d.gl)
    hmm, em = _pair(pkg, orc_libm, gl, d.pos_dist_mb)
    em.init_emission(); hmm.init_emission()
    for it in range(2):
        assert em.estep() == 0
        hmm.iter_EM()
        np.testing.assert_allclose(hmm.ind_lkl, em.ind_lkl, rtol=1e-11)
        np.testing.assert_allclose(hmm.marg_prob, em.marg, rtol=RTOL, atol=1e-12)
        em.set_params(hmm.indF, hmm.alpha, None)
        assert em.mstep_freq(1) == 0
        np.testing.assert_allclose(hmm.freq, em.freq, rtol=RTOL)
    hmm.close()



def test_fast_more_than_64_waves_per_individual(pkg, orc_libm):
    """A small cohort over many sites gets more than 64 lane-chunk waves per individual (a rank's
    share of a strong-scaling run): the objective's finish kernel and the boundary scan then
    combine K > 1 chunk operators per lane.  3 x 300 000 (> 100 waves each), against the
    binary128 anchor (oracle/hp_anchor.c) -- over 300 000 sites the log-space double recursion
    of the reference, as the oracle restates it, is itself 3e-7 off in the posteriors and 6e-13
    (relative) in the log-likelihood; fast mode stays within 1e-14 of the anchor."""
    I, S = 3, 300_000
    d = pkg.simulate.simulate(I, S, seed=77, n_chrom=2, missing_rate=0.05, indF="r", freq="r")
    gl = pkg.simulate.normalise_log_gl(d.gl)
    hmm, em = _pair(pkg, orc_libm, gl, d.pos_dist_mb)
    assert hmm.layout()[0] > 64
    em.init_emission(); hmm.init_emission()
    hp = orclib.HpAnchor()
    lk = hmm.estep().copy()
    mp = hmm.marg_prob
    assert em.estep() == 0
    for i in range(I):
        lk_hp, post = hp.forward_backward(gl[:, i, :].copy(), em.freq.copy(), d.pos_dist_mb, 0.1, 0.2)
        post[post < 1e-5] = 0
        post[post > 1 - 1e-5] = 1
        assert abs(lk[i] - lk_hp) <= 1e-13 * abs(lk_hp)
        assert abs(em.ind_lkl[i] - lk_hp) <= 1e-11 * abs(lk_hp)      # the oracle's own distance
        snapped_differently = (mp[i] != post) & ((mp[i] == 0) | (mp[i] == 1) | (post == 0) | (post == 1))
        assert snapped_differently.sum() <= 2                        # a value next to 1e-5
        ok = ~snapped_differently
        np.testing.assert_allclose(mp[i][ok], post[ok], rtol=0, atol=1e-13)
    # objective values of arbitrary points (general kernel) and two fused iterations: finite,
    # consistent with the E-step's walk
    ind = np.arange(I, dtype=np.uint32)
    np.testing.assert_allclose(hmm.lkl(ind, np.full(I, 0.1), np.full(I, 0.2)), lk, rtol=1e-13)
    for _ in range(2):
        hmm.iter_EM()
        assert np.isfinite(hmm.ind_lkl).all() and np.isfinite(hmm.freq).all()
    hmm.close()


@pytest.mark.parametrize("shape", [(600, 6000), (90, 9000)])
def test_fast_background_pieces_change_nothing(pkg, monkeypatch, shape):
    """nghmm_iter_em puts the backward sweep and est_maf (in parts) onto the stream behind the
    objective rounds' kernels (nghmm_capi.hip: bg_*).  Same kernels on the same data in a
    different order: with the pieces after the rounds (NGHMM_BG_PARTS=0), est_maf cut differently
    (NGHMM_BG_PARTS), the rounds planned by the host (NGHMM_NO_DEV_BFGS: where the pieces go
    BETWEEN the rounds) or everything on one stream (NGHMM_NO_BG_STREAM), every array of three
    iterations must come out bit for bit the same -- 600 individuals: whole rounds, est_maf in
    tile-row parts; 90: the two-lane rounds."""
    I, S = shape
    d = pkg.simulate.simulate(I, S, seed=4242, n_chrom=3, missing_rate=0.05, indF="r", freq="r")
    gl = pkg.simulate.normalise_log_gl(d.gl)

    def run():
        with pkg.NgsFHMM(I, S, device=0, mode=pkg.MODE_FAST) as h:
            h.load(gl, d.pos_dist_mb)
            h.set_params(0.1, 0.2, 0.1)
            h.init_emission()
            out = []
            for _ in range(3):
                h.iter_EM()
                out.append((h.ind_lkl.copy(), h.indF.copy(), h.alpha.copy(), h.freq.copy(),
                            h.marg_prob.copy()))
            return out

    ref = run()
    for env in ({"NGHMM_BG_PARTS": "0"}, {"NGHMM_BG_PARTS": "1"}, {"NGHMM_BG_PARTS": "5"},
                {"NGHMM_NO_DEV_BFGS": "1"}, {"NGHMM_NO_DEV_BFGS": "1", "NGHMM_BG_PARTS": "0"},
                {"NGHMM_NO_DEV_BFGS": "1", "NGHMM_BG_PARTS": "5"}, {"NGHMM_NO_BG_STREAM": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = run()
        for k in env:
            monkeypatch.delenv(k)
        for a, b in zip(ref, got):
            for x, y in zip(a, b):
                assert np.array_equal(x, y), env


def test_fast_mode_end_to_end_against_exact_mode(pkg):
    """A whole EM run in both arithmetic modes (200 individuals x 50 000 sites, 4
    chromosomes, random indF and site frequencies, 2 % missing cells, 25 iterations +
    decoding): the END-TO-END spread of fast mode, as tested numbers.

    Exact mode is bit-identical to the oracle (tests/test_gpu_parity.py), so it stands for
    the reference here.  Per call the two modes agree to ~1e-13; the finite-difference
    L-BFGS-B then amplifies that last-bit noise (SURVEY.md finding 4: the reference itself,
    rebuilt with FMA contraction, moves its final indF by 1e-5 on 10 x 10 000 sites), so the
    trajectory check is: total log-likelihood within 1e-9 relative at EVERY iteration,
    frequencies within 1e-6, indF within 1e-4 (median within 1e-5), alpha within 1e-3
    relative for 99 % of the individuals, and -- the criterion BASELINE.json states without
    tolerance -- identical Viterbi paths in all 10^7 cells.  Measured: tot_lkl 6e-14 after the
    first iteration, at most 6.4e-10 on the way, 1.3e-11 at the end; indF max 3.5e-5, 99 %
    2.4e-5, median 3.5e-7; alpha 99 % 7.9e-5, median 1.1e-6 relative; freq max 1.1e-7; no
    path cell differs."""
    I, S, iters = 200, 50_000, 25
    d = pkg.simulate.simulate(I, S, seed=4242, n_chrom=4, missing_rate=0.02, indF="r", freq="r")
    gl = pkg.simulate.normalise_log_gl(d.gl)
    res = {}
    for mode, name in ((pkg.MODE_FAST, "fast"), (pkg.MODE_EXACT, "exact")):
        with pkg.NgsFHMM(I, S, mode=mode) as hmm:
            hmm.load(gl, d.pos_dist_mb)
            hmm.set_params(0.1, 0.2, 0.1)
            hmm.init_emission()
            lk = []
            for _ in range(iters):
                hmm.iter_EM()
                lk.append(float(np.sum(hmm.ind_lkl)))
            res[name] = dict(lk=np.array(lk), indF=hmm.indF.copy(), alpha=hmm.alpha.copy(),
                             freq=hmm.freq.copy(), path=hmm.viterbi())
    a, b = res["fast"], res["exact"]
    rel = np.abs(a["lk"] - b["lk"]) / np.abs(b["lk"])
    dF = np.abs(a["indF"] - b["indF"])
    dA = np.abs(a["alpha"] - b["alpha"]) / np.abs(b["alpha"])
    dfreq = np.abs(a["freq"] - b["freq"])
    print("fast vs exact, %d iterations at %d x %d: tot_lkl rel first %.1e max %.1e last %.1e; indF "
          "max %.1e p99 %.1e median %.1e; alpha rel p99 %.1e median %.1e; freq max %.1e; paths "
          "differing %d" % (iters, I, S, rel[0], rel.max(), rel[-1], dF.max(), np.quantile(dF, 0.99),
                            np.median(dF), np.quantile(dA, 0.99), np.median(dA), dfreq.max(),
                            int((a["path"] != b["path"]).sum())))
    assert rel[0] < 1e-12 and rel.max() < 1e-9
    assert dF.max() < 1e-4 and np.median(dF) < 1e-5
    assert np.quantile(dA, 0.99) < 1e-3 and np.median(dA) < 1e-5
    assert dfreq.max() < 1e-6
    assert np.array_equal(a["path"], b["path"])


def test_fast_mode_against_the_binary128_anchor(pkg, orc_libm):
    """Fast mode (linear space, per-block rescaling) and the oracle (the reference's log
    space) both measured against oracle/hp_anchor.c, the model in binary128 (no code shared
    with either; tests/test_hp_anchor.py): the linear-space kernels must be no further from
    the truth than the reference's own double-precision formulation -- DESIGN.md section 5's
    claim that fast mode's differences from the oracle are the ORACLE's rounding noise."""
    hp = orclib.HpAnchor()
    d = pkg.simulate.simulate(12, 10_000, seed=31, n_chrom=3, missing_rate=0.03, indF="r",
                              freq="r", alpha=0.3)
    gl = pkg.simulate.normalise_log_gl(d.gl)
    indF = np.linspace(0.02, 0.9, d.n_ind)
    alpha = np.linspace(0.01, 5.0, d.n_ind)
    freq = np.clip(d.freq, 0.02, 0.98)
    em = orclib.OracleEM(orc_libm, gl, d.pos_dist_mb)
    em.set_params(indF, alpha, freq)
    assert em.init_emission() == 0 and em.estep() == 0
    with pkg.NgsFHMM(d.n_ind, d.n_sites, mode=pkg.MODE_FAST) as hmm:
        hmm.load(gl, d.pos_dist_mb)
        hmm.set_params(indF, alpha, freq)
        hmm.init_emission()
        lk = hmm.estep().copy()
        marg = hmm.marg_prob
        hmm.mstep_freq(1)
        f_gpu = hmm.freq
    e_l = {"fast": 0.0, "oracle": 0.0}
    e_p = {"fast": 0.0, "oracle": 0.0}
    for i in range(d.n_ind):
        t_lk, post = hp.forward_backward(gl[:, i], freq, d.pos_dist_mb, indF[i], alpha[i])
        snapped = np.where(post < 1e-5, 0.0, np.where(post > 1 - 1e-5, 1.0, post))
        near = (np.abs(post - 1e-5) < 1e-9) | (np.abs(post - (1 - 1e-5)) < 1e-9)
        for name, l, m in (("fast", lk[i], marg[i]), ("oracle", em.ind_lkl[i], em.marg[i])):
            e_l[name] = max(e_l[name], abs(l - t_lk) / abs(t_lk))
            e_p[name] = max(e_p[name], (np.abs(m - snapped)[~near] / np.maximum(snapped[~near], 1e-5)).max())
    # est_maf on the GPU's own posteriors, a sample of sites
    e_f = 0.0
    for s in range(0, d.n_sites, 53):
        f_h, _ = hp.est_maf(gl[s], marg[:, s])
        e_f = max(e_f, abs(f_gpu[s] - f_h) / f_h)
    print("vs binary128 -- log-likelihood rel: fast %.1e, oracle %.1e; posteriors rel: fast %.1e, "
          "oracle %.1e; est_maf rel: fast %.1e" % (e_l["fast"], e_l["oracle"], e_p["fast"],
                                                   e_p["oracle"], e_f))
    assert e_l["fast"] < 1e-12 and e_p["fast"] < 1e-10 and e_f < 1e-12
    assert e_l["fast"] <= max(2 * e_l["oracle"], 2e-15)
    assert e_p["fast"] <= max(2 * e_p["oracle"], 1e-13)


def test_zero_linear_mass_is_where_the_two_modes_part(pkg, orc_det):
    """DESIGN.md section 5's known divergence, pinned: a site whose two emissions are both
    exactly zero in linear space -- a called genotype 2 under an allele frequency of exactly 0.
    The reference's log space carries the finite stand-in log 0 = -1e15 (conv_space,
    gen_func.cpp:123-130): its objective is a number, about -1e15, and its E-step dies one
    step later with "Fw and Bw lkl do not match!" (EM.cpp:166-170: at that magnitude a double
    resolves 0.125, the two likelihoods differ by more than 1e-3).  Exact mode does exactly
    that, bit for bit with the oracle.  Fast mode works in linear space, where the individual's
    probability mass is exactly zero: it reports "invalid Lkl found!" for the objective and
    the E-step alike; the other individuals are unaffected."""
    I, S, bad_i, bad_s = 6, 400, 3, 137
    d = pkg.simulate.simulate(I, S, seed=8, n_chrom=2)
    gl_called = orc_det.prepare_gl(d.gl, 0, call_geno=True)
    one_hot = orc_det.prepare_gl(np.array([[-1e15, -1e15, 0.0]]), 0, call_geno=True)[0]
    gl_called[bad_s, bad_i] = one_hot                         # genotype 2, called
    freq = np.full(S, 0.2)
    freq[bad_s] = 0.0                                          # ... where the allele does not exist
    em = orclib.OracleEM(orc_det, gl_called, d.pos_dist_mb)
    em.set_params(0.1, 0.2, freq)
    assert em.init_emission() == 0
    want = np.array([-orc_det.lkl([0.1, 0.2], em.e_prob[i], d.pos_dist_mb) for i in range(I)])
    assert want[bad_i] < -9e14 and np.all(np.delete(want, bad_i) > -1e6)
    assert em.estep() == -2                                    # "Fw and Bw lkl do not match!"
    ind = np.arange(I, dtype=np.uint32)
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_EXACT) as h:
        h.load(gl_called, d.pos_dist_mb)
        h.set_params(0.1, 0.2, freq)
        h.init_emission()
        assert np.array_equal(h.lkl(ind, np.full(I, 0.1), np.full(I, 0.2)), want)
        with pytest.raises(pkg.NgsFHMMError) as ei:
            h.estep()
        assert ei.value.code == -2 and "Fw and Bw lkl do not match!" in str(ei.value)
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as h:
        h.load(gl_called, d.pos_dist_mb)
        h.set_params(0.1, 0.2, freq)
        h.init_emission()
        with pytest.raises(pkg.NgsFHMMError) as ei:
            h.lkl(ind, np.full(I, 0.1), np.full(I, 0.2))
        assert ei.value.code == -1 and "invalid Lkl found!" in str(ei.value)
        with pytest.raises(pkg.NgsFHMMError) as ei:
            h.estep()
        assert ei.value.code == -1
        others = np.delete(ind, bad_i)
        np.testing.assert_allclose(h.lkl(others, np.full(I - 1, 0.1), np.full(I - 1, 0.2)),
                                   want[others], rtol=1e-12)


def test_fast_alpha_probes_on_data_with_a_distance_beyond_the_kappa_clamp(pkg, orc_libm, mid_sim):
    """The small-alpha (kappa form) kernels give their alpha probes the distance clamped to 1000 Mb
    (fast_dev.hpp: KAPPA_DCLAMP) -- right only while every finite distance is below that.  The
    M-step's own probes never meet such data (dbfgs_available: distances <= 46 Mb), but a caller of
    nghmm_lkl_batch chooses its points: (F, a), (F, a +- 5e-7) with a = 1e-5 on a data set with one
    gap of 1500 Mb passes the pattern's |da| d_max <= 1e-3 and alpha d_max <= 2^-6.  fd_pattern
    must then refuse the kappa form (round-5 advisory: it did not, and the probes' likelihoods were
    silently those of a 1000 Mb gap): every point against the oracle at 1e-12, and the probes'
    differences from f(x) against the oracle's."""
    d, gl = mid_sim
    pos = d.pos_dist_mb.copy()
    pos[2500] = 1500.0
    F0, a0, da = 0.3, 1e-5, 5e-7
    hmm, em = _pair(pkg, orc_libm, gl, pos, indF=F0, alpha=a0)
    em.init_emission(); hmm.init_emission()
    pts = [(F0, a0), (F0, a0 + da), (F0, a0 - da)]
    ind = np.repeat(np.arange(d.n_ind), 3)
    F = np.tile([p[0] for p in pts], d.n_ind)
    A = np.tile([p[1] for p in pts], d.n_ind)
    got = hmm.lkl(ind, F, A)
    e = em.e_prob
    want = np.array([-orc_libm.lkl([F[p], A[p]], e[ind[p]], pos) for p in range(len(ind))])
    np.testing.assert_allclose(got, want, rtol=1e-12)
    for k in (1, 2):
        np.testing.assert_allclose(got[k::3] - got[0::3], want[k::3] - want[0::3], rtol=1e-4, atol=2e-8)
    hmm.close()

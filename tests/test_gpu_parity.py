"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, through the C ABI,
against the oracle on the same seeded inputs.

Exact mode is compared with the oracle's det build BIT FOR BIT (same exp/log, same
operation order; see csrc/detmath.h) -- every kernel and whole EM trajectories,
which is stronger than BASELINE.json's 1e-9 relative and sidesteps the chaotic
finite-difference M-step (SURVEY.md finding 4).  Against the oracle's libm build
(the reference's arithmetic) the per-call tolerance is 1e-9 relative."""
import importlib
import math

import numpy as np
import pytest

import orclib
from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

RTOL = 1e-9   # BASELINE.json north_star: indF/alpha/freq/posteriors within 1e-9 relative


def _pair(pkg, orc, gl, pos, indF=0.1, alpha=0.2, freq=0.1, mode=None):
    S, I = gl.shape[0], gl.shape[1]
    em = orclib.OracleEM(orc, gl, pos)
    em.set_params(indF, alpha, freq)
    hmm = pkg.NgsFHMM(I, S, device=0, mode=pkg.MODE_EXACT if mode is None else mode)
    hmm.load(gl, pos)
    hmm.set_params(indF, alpha, freq)
    return hmm, em


def test_emission_bitwise(pkg, orc_det, small_sim):
    d, gl = small_sim
    hmm, em = _pair(pkg, orc_det, gl, d.pos_dist_mb, freq=np.linspace(0.01, 0.99, d.n_sites))
    assert em.init_emission() == 0
    hmm.init_emission()
    assert np.array_equal(hmm.e_prob, em.e_prob)
    hmm.close()


def test_estep_bitwise(pkg, orc_det, small_sim):
    d, gl = small_sim
    hmm, em = _pair(pkg, orc_det, gl, d.pos_dist_mb, indF=np.linspace(0.01, 0.9, d.n_ind),
                    alpha=np.linspace(0.01, 5, d.n_ind))
    em.init_emission(); hmm.init_emission()
    assert em.estep() == 0
    lk = hmm.estep()
    assert np.array_equal(lk, em.ind_lkl)
    assert np.array_equal(hmm.marg_prob, em.marg)
    hmm.close()


def test_lkl_batch_bitwise(pkg, orc_det, small_sim):
    d, gl = small_sim
    hmm, em = _pair(pkg, orc_det, gl, d.pos_dist_mb)
    em.init_emission(); hmm.init_emission()
    rng = np.random.default_rng(1)
    n = 333                                   # not a multiple of 64
    ind = rng.integers(0, d.n_ind, n)
    F = rng.uniform(1e-15, 1 - 1e-15, n)
    A = rng.uniform(1e-15, 10, n)
    F[:4] = [1e-15, 1 - 1e-15, 0.5, 1e-6]; A[:4] = [1e-15, 10.0, 1e-15, 10.0]   # the box corners
    got = hmm.lkl(ind, F, A)
    e = em.e_prob
    want = np.array([-orc_det.lkl([F[p], A[p]], e[ind[p]], d.pos_dist_mb) for p in range(n)])
    assert np.array_equal(got, want)
    hmm.close()


def test_mstep_freq_bitwise(pkg, orc_det, small_sim):
    d, gl = small_sim
    hmm, em = _pair(pkg, orc_det, gl, d.pos_dist_mb)
    em.init_emission(); hmm.init_emission()
    em.estep(); hmm.estep()
    assert em.mstep_freq(1) == 0
    hmm.mstep_freq(1)
    assert np.array_equal(hmm.freq, em.freq)
    assert np.array_equal(hmm.e_prob, em.e_prob)
    hmm.close()


@pytest.mark.parametrize("shape,packed", [((1, 1), False), ((37, 70), False), ((100, 130), False),
                                          ((9, 1000), False), ((70, 515), True), ((130, 64), True)])
def test_mstep_freq_exact_kernel_bitwise_on_ragged_shapes(pkg, orc_det, shape, packed):
    """est_maf in exact mode (k_estmaf_exact: a wave per site, the reference's serial sum of
    gen_func.cpp:984-1003 kept in individual order) against the oracle: the same frequencies bit
    for bit, dense and packed handles, ragged site and individual counts (partly filled waves),
    alone and inside a whole fused iteration (est_maf on the second stream underneath the
    rounds)."""
    import orclib
    I, S = shape
    d = pkg.simulate.simulate(I, S, seed=I * 1000 + S, n_chrom=2, missing_rate=0.07, indF="r", freq="r")
    gl = orc_det.prepare_gl(d.gl, 0, call_geno=packed)
    em = orclib.OracleEM(orc_det, gl, d.pos_dist_mb)
    em.set_params(0.3, 0.1, 0.15)
    assert em.init_emission() == 0 and em.estep() == 0 and em.mstep_freq(1) == 0
    got = {}
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_EXACT | (pkg.GENO_PACKED if packed else 0)) as hmm:
        hmm.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=packed)
        hmm.set_params(0.3, 0.1, 0.15)
        hmm.init_emission()
        hmm.estep()
        hmm.mstep_freq(1)
        got[0] = hmm.freq
        assert np.array_equal(got[0], em.freq)
        # a whole fused iteration (est_maf on the second stream underneath the rounds)
        hmm.set_params(0.3, 0.1, 0.15)
        hmm.init_emission()
        em.set_params(0.3, 0.1, 0.15)
        em.init_emission()
        assert em.iterate() == 0
        hmm.iter_EM()
        assert np.array_equal(hmm.freq, em.freq) and np.array_equal(hmm.indF, em.indF)


def test_fused_exact_iteration_gives_the_oracles_bits(pkg, orc_det):
    """Exact mode's fused iteration (nghmm_iter_em) runs its E-step on a second stream next to the
    first objective rounds and est_maf underneath the rest in capped pieces (three waves per SIMD,
    three pieces queued at a time: kernels.hpp) -- the same kernels on the same data in another
    order: three iterations are bit-identical to the oracle's, with the serial recursion kernels
    (exact_serial) as with the producer-consumer ones.  Sites enough for the 16 pieces to exist
    (>= 1024)."""
    import orclib
    I, S = 45, 1500
    d = pkg.simulate.simulate(I, S, seed=61, n_chrom=3, missing_rate=0.05, indF="r")
    gl = pkg.simulate.normalise_log_gl(d.gl)
    em = orclib.OracleEM(orc_det, gl, d.pos_dist_mb)
    em.set_params(0.1, 0.2, 0.1)
    em.init_emission()
    want = []
    for _ in range(3):
        assert em.iterate() == 0
        want.append((em.ind_lkl.copy(), em.indF.copy(), em.alpha.copy(), em.freq.copy(), em.marg.copy()))
    for sw in ({}, {"exact_serial": 1}):
        with pkg.NgsFHMM(I, S, mode=pkg.MODE_EXACT) as hmm:
            for k, v in sw.items():
                hmm.set_switch(k, v)
            hmm.load(gl, d.pos_dist_mb)
            hmm.set_params(0.1, 0.2, 0.1)
            hmm.init_emission()
            for it in range(3):
                hmm.iter_EM()
                got = (hmm.ind_lkl, hmm.indF, hmm.alpha, hmm.freq, hmm.marg_prob)
                for a, b in zip(got, want[it]):
                    assert np.array_equal(a, b), (sw, it)


@pytest.mark.parametrize("fixed", [(False, False), (True, False), (False, True), (True, True)])
def test_mstep_indf_bitwise(pkg, orc_det, small_sim, fixed):
    d, gl = small_sim
    hmm, em = _pair(pkg, orc_det, gl, d.pos_dist_mb)
    em.init_emission(); hmm.init_emission()
    assert em.mstep_indf(*fixed) == 0
    st = hmm.mstep_indf(*fixed)
    assert np.array_equal(hmm.indF, em.indF) and np.array_equal(hmm.alpha, em.alpha)
    if fixed == (True, True):
        assert st.rounds == 0
    else:
        assert st.ref_forward_calls == em.lkl_calls     # the reference's forward-pass count
    hmm.close()


def test_whole_em_bitwise_and_viterbi_identical(pkg, orc_det, small_sim):
    """Five EM iterations end to end + Viterbi: every output identical to the oracle."""
    d, gl = small_sim
    hmm, em = _pair(pkg, orc_det, gl, d.pos_dist_mb)
    em.init_emission(); hmm.init_emission()
    for it in range(5):
        assert em.iterate() == 0
        hmm.iter_EM()
        assert np.array_equal(hmm.ind_lkl, em.ind_lkl), f"iteration {it}"
        assert np.array_equal(hmm.indF, em.indF) and np.array_equal(hmm.alpha, em.alpha)
        assert np.array_equal(hmm.freq, em.freq)
    assert np.array_equal(hmm.marg_prob, em.marg)
    assert np.array_equal(hmm.viterbi(), em.viterbi())
    hmm.close()


def test_em_loop_matches_oracle_loop(pkg, orc_det, small_sim):
    d, gl = small_sim
    hmm, em = _pair(pkg, orc_det, gl, d.pos_dist_mb)
    em.init_emission(); hmm.init_emission()
    n_or = em.run(min_iters=3, max_iters=6)
    n_gpu = hmm.EM(min_iters=3, max_iters=6)
    assert n_or == n_gpu
    assert hmm.tot_lkl == em.tot_lkl
    assert np.array_equal(hmm.indF, em.indF)
    hmm.close()


def test_against_reference_arithmetic_tolerance(pkg, orc_libm, small_sim):
    """Same E-step / objective / freq step against the oracle built with libm (what the
    reference calls): per-call agreement within 1e-9 relative."""
    d, gl = small_sim
    hmm, em = _pair(pkg, orc_libm, gl, d.pos_dist_mb)
    em.init_emission(); hmm.init_emission()
    np.testing.assert_allclose(hmm.e_prob, em.e_prob, rtol=RTOL)
    em.estep(); lk = hmm.estep()
    np.testing.assert_allclose(lk, em.ind_lkl, rtol=RTOL)
    np.testing.assert_allclose(hmm.marg_prob, em.marg, rtol=RTOL, atol=1e-14)
    em.mstep_freq(1); hmm.mstep_freq(1)
    np.testing.assert_allclose(hmm.freq, em.freq, rtol=RTOL)
    hmm.close()


@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (3, 1), (65, 130), (129, 33)])
def test_ragged_shapes(pkg, orc_det, shape):
    """Sizes that are not multiples of the wave (64) or the prefetch group (4)."""
    I, S = shape
    d = pkg.simulate.simulate(I, S, seed=I * 1000 + S, missing_rate=0.1)
    gl = pkg.simulate.normalise_log_gl(d.gl)
    hmm, em = _pair(pkg, orc_det, gl, d.pos_dist_mb)
    em.init_emission(); hmm.init_emission()
    for _ in range(2):
        assert em.iterate() == 0
        hmm.iter_EM()
    assert np.array_equal(hmm.indF, em.indF) and np.array_equal(hmm.freq, em.freq)
    assert np.array_equal(hmm.marg_prob, em.marg)
    assert np.array_equal(hmm.viterbi(), em.viterbi())
    hmm.close()


def test_called_genotypes_and_chromosomes(pkg, orc_det):
    """One-hot GLs with the reference's -1e15 stand-in (shared/read_data.cpp:88-98),
    missing genotypes, several chromosomes (infinite distances)."""
    d = pkg.simulate.simulate(12, 400, seed=77, n_chrom=4)
    geno = d.geno.copy()
    geno[::17, ::3] = -1
    gl = pkg.simulate.called_genotype_gl(geno)
    hmm, em = _pair(pkg, orc_det, gl, d.pos_dist_mb, indF=0.5, alpha=0.01, freq=0.2)
    em.init_emission(); hmm.init_emission()
    for _ in range(3):
        assert em.iterate() == 0
        hmm.iter_EM()
    assert np.array_equal(hmm.indF, em.indF) and np.array_equal(hmm.alpha, em.alpha)
    assert np.array_equal(hmm.freq, em.freq)
    assert np.array_equal(hmm.marg_prob, em.marg)
    assert np.array_equal(hmm.viterbi(), em.viterbi())
    hmm.close()


def test_reference_fatal_errors(pkg, small_sim):
    d, gl = small_sim
    hmm = pkg.NgsFHMM(d.n_ind, d.n_sites)
    hmm.load(gl, d.pos_dist_mb)
    hmm.set_params(0.1, 0.2, 1.5)
    with pytest.raises(pkg.NgsFHMMError) as ei:
        hmm.init_emission()
    assert ei.value.code == -3 and "invalid MAF!" in ei.value.message
    hmm.set_params(0.1, 0.2, 0.1)
    hmm.init_emission()
    hmm.estep()
    with pytest.raises(pkg.NgsFHMMError) as ei:
        hmm.mstep_freq(2)                      # --freq_est 2 aborts in the reference
    assert ei.value.code == -5
    bad = gl.copy()
    bad[5, 2, :] = math.nan
    hmm.load(bad, d.pos_dist_mb)
    hmm.init_emission()
    with pytest.raises(pkg.NgsFHMMError) as ei:
        hmm.estep()
    assert ei.value.code in (-1, -4)           # "invalid Lkl found!" / "value is NaN!"
    hmm.close()


def test_golden_fixture(pkg):
    """Committed fixture (tests/golden/make_golden.py): inputs + oracle outputs."""
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "em_small.npz")
    g = np.load(path)
    hmm = pkg.NgsFHMM(int(g["n_ind"]), int(g["n_sites"]))
    hmm.load(g["gl"], g["pos_dist"])
    hmm.set_params(g["indF0"], g["alpha0"], g["freq0"])
    hmm.init_emission()
    for _ in range(int(g["iters"])):
        hmm.iter_EM()
    # det-build fields are bitwise; libm-build (reference arithmetic) fields to tolerance
    assert np.array_equal(hmm.indF, g["det_indF"]) and np.array_equal(hmm.alpha, g["det_alpha"])
    assert np.array_equal(hmm.freq, g["det_freq"])
    assert np.array_equal(hmm.marg_prob, g["det_marg"])
    assert np.array_equal(hmm.viterbi(), g["det_path"])
    assert np.array_equal(hmm.viterbi(), g["libm_path"])
    np.testing.assert_allclose(hmm.ind_lkl, g["libm_ind_lkl"], rtol=RTOL)
    hmm.close()


@pytest.mark.parametrize("space,call", [(0, False), (0, True), (1, False), (1, True), (2, False)])
def test_input_preparation_on_device_bitwise(pkg, orc_det, small_sim, space, call):
    """nghmm_load_gl_raw: conversion to log space, the two normalisations and the optional
    genotype call of shared/read_data.cpp:36-40,89-98 / ngsF-HMM.cpp:101-117 on the device,
    bit for bit the det oracle's, for log-scale input, normal-space input with the binary
    reader's log 0 -> -1e15 and with the text reader's plain log; zeros, ties (missing data)
    and cells the text reader never filled included."""
    d, _ = small_sim
    rng = np.random.default_rng(space * 2 + call)
    raw = d.gl + rng.normal(size=d.gl.shape[:2] + (1,))      # not normalised
    raw[3, 1] = np.log(1.0 / 3.0)                              # missing data: all equal
    raw[5, 2] = [0.0, -np.inf, -np.inf] if space == 0 else [0.0, -50.0, -60.0]
    if space:
        raw = np.exp(raw)
        raw[7, 0] = [0.0, 0.0, 1.0]                            # log 0
    unread = np.frombuffer(np.uint64(0x7ff8dead00000001).tobytes(), dtype=np.float64)[0]
    raw[9, 3] = unread                                         # an empty text line's cell
    want = orc_det.prepare_gl(raw, space, call)
    with pkg.NgsFHMM(d.n_ind, d.n_sites, mode=pkg.MODE_EXACT) as hmm:
        hmm.load_raw(raw, d.pos_dist_mb, space=space, call_geno=call)
        got = hmm.gl
    assert np.array_equal(got, want, equal_nan=True)
    assert np.all(np.abs(np.exp(want[:3]).sum(-1) - 1) < 1e-12)


def test_input_preparation_nan_check(pkg, small_sim):
    d, gl = small_sim
    raw = gl.copy()
    raw[2, 2, 1] = np.nan
    with pkg.NgsFHMM(d.n_ind, d.n_sites, mode=pkg.MODE_EXACT) as hmm:
        hmm.load_raw(raw, d.pos_dist_mb, check_nan=False)      # text reader: no check
        with pytest.raises(pkg.NgsFHMMError, match="NaN found"):
            hmm.load_raw(raw, d.pos_dist_mb, check_nan=True)   # binary reader: read_data.cpp:42-45


def test_geno_posteriors_bitwise(pkg, orc_det, small_sim):
    """nghmm_geno_posteriors = the .geno output of EM.cpp:367-376, bit for bit the det
    oracle's, whole range and a chunk; before any decoding the path counts as all zeros."""
    d, gl = small_sim
    em = orclib.OracleEM(orc_det, gl, d.pos_dist_mb)
    em.set_params(0.1, 0.2, 0.1)
    em.init_emission()
    assert em.iterate() == 0
    with pkg.NgsFHMM(d.n_ind, d.n_sites, mode=pkg.MODE_EXACT) as hmm:
        hmm.load(gl, d.pos_dist_mb)
        hmm.set_params(0.1, 0.2, 0.1)
        hmm.init_emission()
        hmm.iter_EM()
        zero = np.zeros((d.n_ind, d.n_sites), dtype=np.uint8)
        assert np.array_equal(hmm.geno_posteriors(), em.geno_post(zero))
        path = hmm.viterbi()
        want = em.geno_post(em.viterbi())
        assert np.array_equal(hmm.geno_posteriors(), want)
        assert np.array_equal(hmm.geno_posteriors(17, 40), want[17:57])


def test_device_formatting_is_printf(pkg, orc_det, small_sim):
    """The .ibd posterior lines formatted on the device are byte for byte what printf("%f")
    writes: on the E-step's posteriors, and on values chosen to stress the rounding -- exact
    ties (j/128 * ... = k + 0.5 in units of 1e-6: ties to even), neighbours of ties, values
    next to 0 and 1, the snapping thresholds, random values."""
    d, gl = small_sim
    with pkg.NgsFHMM(d.n_ind, d.n_sites, mode=pkg.MODE_EXACT) as hmm:
        hmm.load(gl, d.pos_dist_mb)
        hmm.set_params(0.1, 0.2, 0.1)
        hmm.init_emission()
        hmm.estep()
        post = hmm.marg_prob
        want = "".join("\t".join("%f" % v for v in row) + "\n" for row in post).encode()
        assert hmm.format_posteriors() == want
        assert hmm.format_posteriors(3, 2) == "".join(
            "\t".join("%f" % v for v in row) + "\n" for row in post[3:5]).encode()

        rng = np.random.default_rng(0)
        ties = np.arange(0, 129) / 128.0                   # j/128 = (j * 7812.5) e-6: exact ties
        near = np.concatenate([np.nextafter(ties, 0.0), np.nextafter(ties, 1.0)])
        edge = np.array([0.0, 1.0, 1e-5, 1 - 1e-5, 5e-7, 4.9999999e-7, 5.0000001e-7, 0.9999995,
                         0.99999949, 0.99999951, 1 - 2.0 ** -53, 2.0 ** -60, 0.5, 0.1234565,
                         0.1234575, 0.0000015, 0.0000025])
        dec = np.round(rng.uniform(0, 1, 20000), 6) + rng.choice([0.0, 5e-7, -5e-7], 20000)
        vals = np.clip(np.concatenate([ties, near, edge, dec, rng.uniform(0, 1, 50000),
                                       rng.uniform(0, 1, 5000) ** 8]), 0.0, 1.0)
        vals = vals[:(len(vals) // 7) * 7].reshape(-1, 7)
        got = hmm.format_fixed6(vals)
        want = "".join("\t".join("%f" % v for v in row) + "\n" for row in vals).encode()
        assert got == want
        with pytest.raises(pkg.NgsFHMMError):
            hmm.format_fixed6(np.array([[0.5, 1.5]]))


def test_producer_consumer_recursions_equal_the_serial_ones(pkg):
    """The exact-mode forward / backward recursions run as producer-consumer workgroups
    (kernels_exact_pc.hip: transition logs of a block of sites computed in parallel, two lanes
    per chain, detmath.h's chain forms of exp / log); NGHMM_EXACT_SERIAL=1 selects the one-lane-
    per-chain kernels they replaced.  Same operations on the same operands: every array of two
    EM iterations and the decoded paths must be the same bits, at a size the oracle needs
    minutes for (ragged: 333 individuals are 10.4 workgroups of 32 chains, 24 007 sites are
    1200.35 ring blocks of 20), over three chromosomes, with missing data and objective points
    at the parameter bounds.  The second data set (true indF uniform in (0, 1)) drives
    individuals onto the bounds F = 1 - 1e-15, alpha = 1e-15 and, in its second iteration, the
    recursion into the reference's "invalid Lkl found!" (shared/HMM.cpp:18-21; the oracle
    returns the same there): both kernel sets must fail the same way at the same point."""
    import os
    I, S = 333, 24_007
    rng = np.random.default_rng(5)
    ind = rng.integers(0, I, 500)
    F = np.concatenate([rng.uniform(0, 1, 490), [1e-15, 1 - 1e-15, 1e-15, 1 - 1e-15, 0.5,
                                                  0.5, 0.3, 0.7, 1e-6, 1 - 1e-6]])
    A = np.concatenate([rng.uniform(1e-3, 10, 490), [1e-15, 10, 10, 1e-15, 1e-15, 10, 1e-9,
                                                      5.0, 1e-6, 1e-6]])
    for kw, n_ok in ((dict(), 2), (dict(indF="r"), 1)):
        d = pkg.simulate.simulate(I, S, seed=77, n_chrom=3, missing_rate=0.03, freq="r", **kw)
        gl = pkg.simulate.normalise_log_gl(d.gl)
        res = {}
        try:
            for serial in (False, True):
                if serial:
                    os.environ["NGHMM_EXACT_SERIAL"] = "1"
                with pkg.NgsFHMM(I, S, mode=pkg.MODE_EXACT) as hmm:
                    hmm.load(gl, d.pos_dist_mb)
                    hmm.set_params(0.1, 0.2, 0.1)
                    hmm.init_emission()
                    out = {"obj": hmm.lkl(ind, F, A)}
                    for it in range(2):
                        try:
                            hmm.iter_EM()
                        except pkg.NgsFHMMError as e:
                            out[f"error{it}"] = np.array([e.code])
                            break
                        out[f"lkl{it}"] = hmm.ind_lkl.copy()
                        out[f"post{it}"] = hmm.marg_prob.copy()
                        out[f"indF{it}"] = hmm.indF.copy()
                        out[f"alpha{it}"] = hmm.alpha.copy()
                        out[f"freq{it}"] = hmm.freq.copy()
                    else:
                        out["path"] = hmm.viterbi()
                res[serial] = out
        finally:
            os.environ.pop("NGHMM_EXACT_SERIAL", None)
        assert np.all(np.isfinite(res[False]["obj"]))
        assert sorted(res[False]) == sorted(res[True])
        assert f"lkl{n_ok - 1}" in res[False] and (n_ok == 2) == ("path" in res[False])
        if n_ok == 1:
            assert res[False]["error1"][0] == -1          # NGHMM_ERR_INVALID_LKL
        for k in res[False]:
            assert np.array_equal(res[False][k], res[True][k]), k

"""BASELINE.json's full size (1000 individuals x 1,000,000 sites) on the GPU, checked
through properties that do not need the CPU oracle at that size:

* fast mode against EXACT mode (which is bit-identical to the oracle at small sizes).  At a
  million sites the reference's own log-space formulation has accumulated rounding: every
  forward step rounds a quantity of size |Fw| ~ 1e6 (1e-10 absolute per site), so its
  log-likelihoods carry ~1e-6 absolute (3e-12 relative) noise and its posteriors
  exp(Fw + Bw - lkl) ~1e-6 relative.  The linear-space kernels rescale and do not have
  this growth, so the tolerances here are the exact formulation's error bars, not fast
  mode's: 2e-11 relative on log-likelihoods, 2e-5 absolute on posteriors;
* chromosome-break additivity: with an infinite distance in the middle, the forward
  log-likelihood is the sum of the two halves' (the transition forgets everything there);
* independence from the chunking of the site axis (the C knob of the interleaved layout);
* est_maf: a fixed point of its own update within the reference's 1e-5 stopping rule, and
  deterministic.
"""
import importlib
import os

import numpy as np
import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

I, S = 1000, 1_000_000


@pytest.fixture(scope="module")
def big(pkg):
    import torch
    dev = torch.device("cuda", 0)
    gl, pos = pkg.simulate.simulate_torch(I, S, dev, seed=2024)
    pos[S // 2] = float("inf")              # a second chromosome
    torch.cuda.synchronize()
    yield gl, pos
    del gl, pos
    torch.cuda.empty_cache()


def _handle(pkg, gl, pos, mode, n_sites=S, site0=0):
    h = pkg.NgsFHMM(I, n_sites, mode=mode)
    h.load_device(gl[site0:site0 + n_sites].data_ptr(), pos[site0:site0 + n_sites].data_ptr())
    h.set_params(0.4, 0.02, 0.15)
    h.init_emission()
    return h


def test_fast_equals_exact_at_full_size(pkg, big):
    gl, pos = big
    fast = _handle(pkg, gl, pos, pkg.MODE_FAST)
    lk_fast = fast.estep().copy()
    rng = np.random.default_rng(0)
    ind = rng.integers(0, I, 300)
    F = rng.uniform(0.01, 0.99, 300)
    A = rng.uniform(1e-3, 5, 300)
    obj_fast = fast.lkl(ind, F, A)
    sub = slice(0, 250)   # 2.5 x 10^8 posteriors
    post_fast = fast.marg_prob[sub].copy()
    fast.mstep_freq(1)
    freq_fast = fast.freq
    fast.close()

    exact = _handle(pkg, gl, pos, pkg.MODE_EXACT)
    lk_exact = exact.estep().copy()
    np.testing.assert_allclose(lk_fast, lk_exact, rtol=2e-11)
    np.testing.assert_allclose(obj_fast, exact.lkl(ind, F, A), rtol=2e-11)
    post_exact = exact.marg_prob[sub]
    np.testing.assert_allclose(post_fast, post_exact, atol=2e-5)
    exact.mstep_freq(1)
    # est_maf sees posteriors that differ at the 1e-6 level (above); where one of them
    # crosses the 1e-5 snapping threshold of check_interv the input changes by 1e-5
    np.testing.assert_allclose(freq_fast, exact.freq, rtol=1e-5)
    rel = np.abs(freq_fast - exact.freq) / exact.freq
    print("freq rel diff: median %.3g  99%% %.3g  max %.3g; posterior max abs diff %.3g" % (
        np.median(rel), np.quantile(rel, 0.99), rel.max(), np.abs(post_fast - post_exact).max()))
    assert np.median(rel) < 1e-6
    exact.close()


def test_chromosome_break_additivity_and_chunking(pkg, big):
    gl, pos = big
    whole = _handle(pkg, gl, pos, pkg.MODE_FAST)
    lk = whole.estep().copy()
    whole.close()
    # the two chromosomes as separate data sets.  The second one starts with d = inf, which
    # is what a first site sees anyway (its transition is the stationary distribution).
    left = _handle(pkg, gl, pos, pkg.MODE_FAST, n_sites=S // 2, site0=0)
    lk_l = left.estep().copy()
    left.close()
    right = _handle(pkg, gl, pos, pkg.MODE_FAST, n_sites=S - S // 2, site0=S // 2)
    lk_r = right.estep().copy()
    right.close()
    np.testing.assert_allclose(lk, lk_l + lk_r, rtol=1e-12)
    # a different decomposition of the site axis
    os.environ["NGHMM_FAST_C"] = "20"
    try:
        other = _handle(pkg, gl, pos, pkg.MODE_FAST)
        np.testing.assert_allclose(other.estep(), lk, rtol=1e-12)
        other.close()
    finally:
        del os.environ["NGHMM_FAST_C"]


def test_est_maf_properties_at_full_size(pkg, big):
    gl, pos = big
    h = _handle(pkg, gl, pos, pkg.MODE_FAST)
    h.estep()
    h.mstep_freq(1)
    f1 = h.freq
    assert np.all((f1 > 0) & (f1 < 1))
    assert abs(f1.mean() - 0.2) < 0.02            # simulated at 0.2
    h.mstep_freq(1)                                # same posteriors: same frequencies, exactly
    assert np.array_equal(h.freq, f1)
    path = h.viterbi()
    post = h.marg_prob[:8]
    agree = ((post > 0.5) == (path[:8] == 1)).mean()
    assert agree > 0.98                            # decoding and posteriors tell the same story
    h.close()


@pytest.mark.parametrize("n_ind,n_sites,kw", [
    (1000, 200_000, {}),
    (1000, 200_000, dict(freq="r")),
    (1000, 100_000, dict(freq="r", depth=10.0)),
    (2000, 50_000, dict(freq="r")),
    (100, 200_000, dict(freq="r")),
    # several waves per site: 4 (3000), 8 with 10 / 14 / 16 individuals per lane (5000, 6500, 8000)
    (3000, 20_000, dict(freq="r")),
    (5000, 20_000, dict(freq="r")),
    (6500, 9_000, dict(freq="r")),
    (8000, 8_000, dict(freq="r")),
])
def test_est_maf_interpolated_passes_equal_exact_passes(pkg, n_ind, n_sites, kw):
    """est_maf runs most of its <= 101 passes per site on a checked Chebyshev interpolant of
    the per-pass sums (k_fast_estmaf / k_fast_estmaf_interp); the switch estmaf_interp = 0 evaluates
    every pass over all individuals.  Same recursion, same stopping rule: the frequencies
    must agree far inside the 1e-9 parity tolerance, on fixed and on uniform site
    frequencies, at low and high depth, and across the kernel's individuals-per-lane
    variants.  Two consecutive frequency steps, so the second starts from posteriors of
    refreshed emissions."""
    import torch
    gl, pos = pkg.simulate.simulate_torch(n_ind, n_sites, torch.device("cuda", 0), seed=1, **kw)
    torch.cuda.synchronize()
    hmm = pkg.NgsFHMM(n_ind, n_sites, mode=pkg.MODE_FAST)
    hmm.load_device(gl.data_ptr(), pos.data_ptr())
    del gl, pos
    res = {}
    try:
        for interp in ("0", "1"):
            hmm.set_switch("estmaf_interp", int(interp))
            hmm.set_params(np.full(n_ind, 0.1), np.full(n_ind, 0.01), np.full(n_sites, 0.1))
            hmm.init_emission()
            for _ in range(2):
                hmm.estep()
                hmm.mstep_freq(1)
            res[interp] = hmm.freq.copy()
    finally:
        hmm.close()
        torch.cuda.empty_cache()
    f0, f1 = res["0"], res["1"]
    rel = np.abs(f1 - f0) / np.abs(f0)
    print(f"est_maf interp vs exact passes {n_ind} x {n_sites} {kw}: max rel diff {rel.max():.3e}, "
          f"sites > 1e-12: {int((rel > 1e-12).sum())}")
    assert rel.max() < 1e-10

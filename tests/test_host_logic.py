"""Host-side logic of the product that needs no GPU: the lock-step batched L-BFGS-B
(ngsf-hmm_amd/csrc/bfgs_batch.cpp) must reproduce, per individual, exactly what the
reference's blocking findmax_bfgs does with the same objective -- same final
(indF, alpha) to the last bit and the same count of forward passes -- and the
simulator must produce well-formed inputs."""
import ctypes as C
import importlib
import math

import numpy as np
import pytest

import orclib
from orclib import LklData, _dp, run_findmax


def _scalar_findmax(orc, impl, e_i, pos, x0, fixed=(False, False)):
    lb, ub = [1 / 1e15, 1 / 1e15], [1 - 1 / 1e15, 10.0]
    for k in range(2):
        if fixed[k]:
            lb[k] = ub[k] = x0[k]
    ei = np.ascontiguousarray(e_i)
    data = LklData(_dp(ei), _dp(pos), len(pos), 0, 0)
    x, _ = run_findmax(impl, C.cast(orc.lib.orc_lkl, C.c_void_p), x0, lb, ub,
                       C.cast(C.byref(data), C.c_void_p))
    return x, data.n_calls


@pytest.mark.parametrize("fixed", [(False, False), (True, False), (False, True)])
def test_batched_bfgs_equals_blocking_findmax(pkg, orc_libm, small_sim, fixed):
    hm = importlib.import_module("ngsf-hmm_amd.hmm")
    d, gl = small_sim
    em = orclib.OracleEM(orc_libm, gl, d.pos_dist_mb)
    em.set_params(0.1, 0.2, 0.1)
    em.init_emission()
    e = em.e_prob
    pos = np.ascontiguousarray(d.pos_dist_mb)
    x0F = np.linspace(0.05, 0.9, d.n_ind)
    x0A = np.linspace(0.01, 2.0, d.n_ind)

    def objective(i, F, a):          # forward log-likelihood (EM.cpp:463 returns its negative)
        return -orc_libm.lkl([F, a], e[i], pos)

    F, A, st = hm.bfgs_batch_host(x0F, x0A, objective, indF_fixed=fixed[0], alpha_fixed=fixed[1])
    calls = 0
    for i in range(d.n_ind):
        x, n = _scalar_findmax(orc_libm, orc_libm.lib.orc_findmax_bfgs, e[i], pos,
                               [x0F[i], x0A[i]], fixed)
        assert x[0] == F[i] and x[1] == A[i], f"individual {i}"
        calls += n
    assert st.ref_forward_calls == calls
    assert st.points < calls          # duplicates and fixed-parameter probes are not re-evaluated
    assert st.rounds >= 2


def test_batched_bfgs_against_reference_object(pkg, orc_libm, ref_bfgs, small_sim):
    """Same, against the reference's own compiled findmax_bfgs."""
    hm = importlib.import_module("ngsf-hmm_amd.hmm")
    d, gl = small_sim
    em = orclib.OracleEM(orc_libm, gl, d.pos_dist_mb)
    em.set_params(0.3, 0.05, 0.2)
    em.init_emission()
    e = em.e_prob
    pos = np.ascontiguousarray(d.pos_dist_mb)
    x0F = np.full(d.n_ind, 0.3)
    x0A = np.full(d.n_ind, 0.05)
    F, A, st = hm.bfgs_batch_host(x0F, x0A, lambda i, f, a: -orc_libm.lkl([f, a], e[i], pos))
    for i in range(d.n_ind):
        x, _ = _scalar_findmax(orc_libm, ref_bfgs.findmax, e[i], pos, [0.3, 0.05])
        assert x[0] == F[i] and x[1] == A[i]


def test_nonfinite_parameters_take_the_reference_branch(pkg):
    """EM.cpp:454-456: NaN/Inf parameters make the objective -1e15 without a forward pass."""
    hm = importlib.import_module("ngsf-hmm_amd.hmm")
    seen = []

    def objective(i, F, a):
        seen.append((F, a))
        return -((F - 0.4) ** 2 + (a - 2.0) ** 2)
    F, A, st = hm.bfgs_batch_host([0.2], [1.0], objective)
    assert all(math.isfinite(f) and math.isfinite(a) for f, a in seen)
    assert abs(F[0] - 0.4) < 1e-3 and abs(A[0] - 2.0) < 1e-3


def test_both_fixed_is_a_no_op(pkg):
    hm = importlib.import_module("ngsf-hmm_amd.hmm")
    F, A, st = hm.bfgs_batch_host([0.2, 0.3], [1.0, 2.0], lambda i, f, a: 0.0, True, True)
    assert F.tolist() == [0.2, 0.3] and A.tolist() == [1.0, 2.0] and st.rounds == 0


def test_simulator_shapes_and_model(pkg):
    sim = pkg.simulate
    d = sim.simulate(7, 500, seed=3, n_chrom=3, missing_rate=0.1)
    assert d.gl.shape == (500, 7, 3) and d.path.shape == (7, 500) and d.geno.shape == (500, 7)
    np.testing.assert_allclose(np.exp(d.gl).sum(axis=2), 1.0, atol=1e-8)   # rounded to 10 decimals
    assert np.isinf(d.pos_dist_mb).sum() == 2                              # two chromosome changes
    assert d.pos_dist_mb[0] == d.pos[0] / 1e6                              # first site: absolute position
    assert np.all(d.pos_dist_mb[np.isfinite(d.pos_dist_mb)] >= 1e-6)
    gl = sim.normalise_log_gl(d.gl)
    np.testing.assert_allclose(np.exp(gl).sum(axis=2), 1.0, atol=1e-12)
    cg = sim.called_genotype_gl(d.geno)
    assert cg.shape == d.gl.shape and np.all(cg.max(axis=2) == 0.0)
    # IBD sites are homozygous
    assert np.all(d.geno.T[d.path == 1] != 1)
    # same seed, same data
    d2 = sim.simulate(7, 500, seed=3, n_chrom=3, missing_rate=0.1)
    assert np.array_equal(d.gl, d2.gl) and np.array_equal(d.path, d2.path)

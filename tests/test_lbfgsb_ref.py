"""The product's L-BFGS-B core (ngsf-hmm_amd/csrc/lbfgsb.cpp, driven by the oracle's
restatement of findmax_bfgs) against the REFERENCE'S OWN OBJECT CODE: shared/bfgs.cpp
compiled from /root/reference in place into oracle/_ref/libref_bfgs.so.

Both are handed the same objective; every point at which the objective is evaluated,
in order, and the final x must agree bit for bit.  This pins SURVEY.md section 8 rows
a8/a9 (findmax_bfgs, getgradient, setulb_ ... dtrsl_)."""
import ctypes as C

import numpy as np
import pytest

from orclib import OBJECTIVE, LklData, _dp, run_findmax


def _traced(fn):
    trace = []

    def cb(xp, _):
        n = cb.n
        x = [xp[i] for i in range(n)]
        trace.append(tuple(x))
        return fn(x)
    return cb, trace


def _run_both(ref, orc, fn, x0, lb, ub):
    out = []
    for impl in (ref.findmax, orc.lib.orc_findmax_bfgs):
        cb, trace = _traced(fn)
        cb.n = len(x0)
        x, r = run_findmax(impl, OBJECTIVE(cb), x0, lb, ub)
        out.append((x, r, trace))
    return out


def rosen(x):
    return sum(100.0 * (x[i + 1] - x[i] ** 2) ** 2 + (1 - x[i]) ** 2 for i in range(len(x) - 1))


CASES_2D = [
    (rosen, [0.1, 0.2]), (rosen, [0.9, 5.0]), (rosen, [1e-15, 1e-15]),
    (lambda x: np.sin(3 * x[0]) * np.cos(2 * x[1]) + 0.1 * (x[0] - 0.3) ** 2 + 0.05 * (x[1] - 2) ** 2,
     [0.5, 1.0]),
    (lambda x: -np.log(x[0] * (1 - x[0])) + (x[1] - 11) ** 2, [0.2, 3.0]),   # optimum on a bound
    (lambda x: (x[0] - 2) ** 2 + (x[1] + 1) ** 2, [0.5, 0.5]),                 # both bounds active
    (lambda x: abs(x[0] - 0.3) ** 1.5 + abs(x[1] - 0.7) ** 1.5, [0.9, 0.1]),   # non-smooth: long line searches
    (lambda x: 1e6 * (x[0] - 0.5) ** 2 + 1e-6 * (x[1] - 5) ** 2, [0.1, 0.1]),  # badly scaled
]


@pytest.mark.parametrize("k", range(len(CASES_2D)))
def test_box_2d_bitwise(ref_bfgs, orc_libm, k):
    fn, x0 = CASES_2D[k]
    lb, ub = [1e-15, 1e-15], [1 - 1e-15, 10.0]           # the bounds of EM.cpp:425-427
    (xr, rr, tr), (xo, ro, to) = _run_both(ref_bfgs, orc_libm, fn, x0, lb, ub)
    assert tr == to, "evaluation-point traces differ"
    assert xr.tobytes() == xo.tobytes()
    assert (rr == ro) or (np.isnan(rr) and np.isnan(ro))


@pytest.mark.parametrize("fixed", [0, 1])
def test_fixed_parameter(ref_bfgs, orc_libm, fixed):
    """--indF_fixed / --alpha_fixed: lower == upper bound (EM.cpp:429-436)."""
    fn, x0 = CASES_2D[3]
    lb, ub = [1e-15, 1e-15], [1 - 1e-15, 10.0]
    lb[fixed] = ub[fixed] = x0[fixed]
    (xr, rr, tr), (xo, ro, to) = _run_both(ref_bfgs, orc_libm, fn, x0, lb, ub)
    assert tr == to and xr.tobytes() == xo.tobytes()
    assert xo[fixed] == x0[fixed]


def test_more_variables_and_memory_wrap(ref_bfgs, orc_libm):
    """n = 6 Rosenbrock chain: > 10 BFGS updates, so the limited memory wraps
    (iupdat > m branches of matupd_/formk_) and several variables hit bounds."""
    x0 = [-1.2, 1.0, -0.5, 0.8, 0.3, -0.9]
    lb = [-2.0, -0.5, -1.0, 0.0, -2.0, -2.0]
    ub = [0.9, 2.0, 2.0, 0.7, 2.0, 0.95]
    (xr, rr, tr), (xo, ro, to) = _run_both(ref_bfgs, orc_libm, rosen, x0, lb, ub)
    assert len(tr) > 13 * 20
    assert tr == to and xr.tobytes() == xo.tobytes()


def test_infeasible_start_is_projected(ref_bfgs, orc_libm):
    (xr, rr, tr), (xo, ro, to) = _run_both(ref_bfgs, orc_libm, rosen, [3.0, -4.0], [0.0, 0.0],
                                           [2.0, 2.0])
    assert tr == to and xr.tobytes() == xo.tobytes()


def test_hmm_objective_bitwise(ref_bfgs, orc_libm, small_sim):
    """The real objective (EM.cpp:449-464) on simulated data: same forward-pass
    count and the same (indF, alpha) to the last bit, for every individual."""
    import orclib
    d, gl = small_sim
    em = orclib.OracleEM(orc_libm, gl, d.pos_dist_mb)
    em.set_params(0.1, 0.2, 0.1)
    assert em.init_emission() == 0
    e = em.e_prob
    pos = np.ascontiguousarray(d.pos_dist_mb)
    for i in range(d.n_ind):
        res = []
        for impl in (ref_bfgs.findmax, orc_libm.lib.orc_findmax_bfgs):
            ei = np.ascontiguousarray(e[i])
            data = LklData(_dp(ei), _dp(pos), d.n_sites, 0, 0)
            x, r = run_findmax(impl, C.cast(orc_libm.lib.orc_lkl, C.c_void_p), [0.1, 0.2],
                               [1e-15, 1e-15], [1 - 1e-15, 10.0], C.cast(C.byref(data), C.c_void_p))
            res.append((x.tobytes(), r, data.n_calls))
        assert res[0] == res[1]


def test_em_with_reference_optimizer_identical(ref_bfgs, orc_libm, small_sim):
    """Three whole EM iterations of the oracle with the reference's optimizer object
    plugged in give bit-identical state to the oracle with the restated optimizer."""
    import orclib
    d, gl = small_sim
    states = []
    for use_ref in (False, True):
        em = orclib.OracleEM(orc_libm, gl, d.pos_dist_mb)
        em.set_params(0.1, 0.2, 0.1)
        if use_ref:
            em.use_reference_optimizer(ref_bfgs)
        em.init_emission()
        for _ in range(3):
            assert em.iterate() == 0
        states.append((em.indF.tobytes(), em.alpha.tobytes(), em.freq.tobytes(),
                       em.marg.tobytes(), em.ind_lkl.tobytes(), em.lkl_calls))
    assert states[0] == states[1]

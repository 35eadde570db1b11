"""detmath.h exp/log (used by exact-mode kernels and the oracle's det build) against
libm: at most 1 ulp apart, identical special values."""
import numpy as np


def _ulps(a, b):
    a = np.asarray(a, dtype=np.float64).view(np.int64)
    b = np.asarray(b, dtype=np.float64).view(np.int64)
    return np.abs(a - b)


def test_exp_log_within_one_ulp(orc_det, orc_libm):
    assert orc_det.lib.orc_detmath() == 1 and orc_libm.lib.orc_detmath() == 0
    rng = np.random.default_rng(7)
    xs = np.concatenate([rng.uniform(-745, 709, 20000), rng.uniform(-2, 2, 20000),
                         -rng.uniform(0, 60, 20000), rng.uniform(-1e-9, 1e-9, 2000)])
    e_det = np.array([orc_det.lib.orc_exp(float(x)) for x in xs])
    e_ref = np.array([orc_libm.lib.orc_exp(float(x)) for x in xs])
    assert _ulps(e_det, e_ref).max() <= 1
    ys = np.concatenate([np.exp(rng.uniform(-700, 700, 20000)), rng.uniform(0, 2, 20000),
                         1 + rng.uniform(-1e-3, 1e-3, 5000), rng.uniform(0, 1e-300, 100)])
    l_det = np.array([orc_det.lib.orc_log(float(y)) for y in ys])
    l_ref = np.array([orc_libm.lib.orc_log(float(y)) for y in ys])
    assert _ulps(l_det, l_ref).max() <= 1


def test_special_values(orc_det):
    L = orc_det.lib
    assert L.orc_exp(float("-inf")) == 0.0
    assert L.orc_exp(-1e15) == 0.0            # the reference's -INF stand-in underflows to 0
    assert L.orc_exp(0.0) == 1.0
    assert L.orc_exp(float("inf")) == float("inf")
    assert np.isnan(L.orc_exp(float("nan")))
    assert L.orc_log(0.0) == float("-inf")
    assert L.orc_log(1.0) == 0.0
    assert np.isnan(L.orc_log(-1.0))
    assert L.orc_log(float("inf")) == float("inf")
    assert abs(L.orc_log(5e-324) - (-744.4400719213812)) < 1e-9

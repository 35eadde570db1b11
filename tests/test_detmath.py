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


def _probe(tmp_path_factory, which):
    """detmath.h compiled for the host with its select-form variants exposed over arrays."""
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path_factory.mktemp("detmath") / "libdetprobe.so")
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-std=gnu11", "-Wall",
                    "-I" + os.path.join(root, "ngsf-hmm_amd", "csrc"),
                    os.path.join(root, "tests", "stub", "detmath_probe.c"), "-o", out], check=True)
    L = C.CDLL(out)
    dp = C.POINTER(C.c_double)
    L.probe_exp.argtypes = [dp, C.c_size_t, dp, dp]
    L.probe_log.argtypes = [dp, C.c_size_t, dp, dp]
    L.probe_logsum2.argtypes = [dp, dp, C.c_size_t, dp, dp]
    L.probe_exp_chain.argtypes = [dp, C.c_size_t, dp, dp]
    L.probe_log_chain.argtypes = [dp, C.c_size_t, dp, dp]
    L.probe_logsum2_chain.argtypes = [dp, dp, C.c_size_t, dp, dp]

    def run1(fn, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        g, v = np.empty_like(x), np.empty_like(x)
        fn(x.ctypes.data_as(dp), x.size, g.ctypes.data_as(dp), v.ctypes.data_as(dp))
        return g, v

    def run2(fn, a0, a1):
        a0 = np.ascontiguousarray(a0, dtype=np.float64)
        a1 = np.ascontiguousarray(a1, dtype=np.float64)
        g, v = np.empty_like(a0), np.empty_like(a0)
        fn(a0.ctypes.data_as(dp), a1.ctypes.data_as(dp), a0.size,
           g.ctypes.data_as(dp), v.ctypes.data_as(dp))
        return g, v

    sel = ((lambda x: run1(L.probe_exp, x)), (lambda x: run1(L.probe_log, x)),
           (lambda a, b: run2(L.probe_logsum2, a, b)))
    chain = ((lambda x: run1(L.probe_exp_chain, x)), (lambda x: run1(L.probe_log_chain, x)),
             (lambda a, b: run2(L.probe_logsum2_chain, a, b)))
    return {"select": sel, "chain": chain}[which]


def _same_bits(a, b):
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


def _around(values, width=40):
    """every double within `width` ulps of each value"""
    v = np.asarray(values, dtype=np.float64).view(np.int64)
    return (v[:, None] + np.arange(-width, width + 1)[None, :]).ravel().view(np.float64)


import pytest


@pytest.mark.parametrize("which", ["select", "chain"])
def test_select_form_variants_return_the_same_bits(tmp_path_factory, which):
    """det_exp_sel / det_log_pos / det_logsum2 and det_exp_chain / det_log_chain /
    det_logsum2_chain (what the exact-mode recursion kernels call) against det_exp / det_log /
    the reference's logsum loop: bit-identical over 10^7 arguments, every case boundary of
    the two functions and the special values."""
    p_exp, p_log, p_ls = _probe(tmp_path_factory, which)
    rng = np.random.default_rng(11)
    ln2 = np.log(2.0)
    hi_word = lambda h: np.array([h << 32], dtype=np.uint64).view(np.float64)[0]
    exp_edges = [-0.5 * ln2, -1.5 * ln2, -2.0 ** -28, -709.782712893384, -745.1332191019411,
                 -hi_word(0x3fd62e42), -hi_word(0x3fd62e43), -hi_word(0x3FF0A2B2), -hi_word(0x40862E42),
                 -hi_word(0x3e300000), -708.3964185322641, -1021 * ln2, -1022 * ln2, -1000 * ln2]
    xs = np.concatenate([
        -rng.uniform(0, 760, 2_000_000), -rng.uniform(0, 3, 2_000_000),
        -np.exp(rng.uniform(-45, 7, 1_000_000)), -rng.uniform(700, 750, 200_000),
        -rng.uniform(0, 0.36, 3_000_000), -rng.uniform(0.3, 1.1, 1_000_000),
        _around(exp_edges), -np.arange(0, 1100) * ln2, -(np.arange(0, 1100) + 0.5) * ln2,
        [0.0, -0.0, -np.inf, np.nan, -np.nan, -1e15, -1e300, -5e-324, -2.2250738585072014e-308,
         1.0, 0.5, np.inf, 1e-300, 1e300, 709.782712893384, 709.7827128933841]])
    xs = np.concatenate([xs, -xs, rng.uniform(-1e-6, 1e-6, 500_000)])
    g, v = p_exp(xs)
    assert _same_bits(g, v), int((g.view(np.uint64) != v.view(np.uint64)).sum())

    log_edges = [1.0, 2.0, np.sqrt(2.0), 0.5, np.sqrt(0.5), 1 + 2.0 ** -20, 1 - 2.0 ** -21,
                 hi_word(0x3ff00000 + 0x6147a), hi_word(0x3ff00000 + 0x6b851), hi_word(0x3ff6a09e),
                 hi_word(0x3ff6a09f), 2.2250738585072014e-308, 1.7976931348623157e308, 4.0, 0.25]
    ys = np.concatenate([
        1 + rng.uniform(0, 1, 5_000_000), 1 + np.exp(rng.uniform(-60, 0, 1_000_000)),
        np.exp(rng.uniform(-700, 700, 1_000_000)), rng.uniform(0, 1, 1_000_000),
        1 - np.exp(rng.uniform(-40, -1, 300_000)), _around(log_edges),
        [0.0, -0.0, -1.0, np.inf, np.nan, 5e-324, 1e-310]])
    g, v = p_log(ys)
    assert _same_bits(g, v), int((g.view(np.uint64) != v.view(np.uint64)).sum())

    # logsum2 as the recursions feed it: two log quantities of any magnitude and distance,
    # equal terms, the reference's -1e15 stand-in, infinities and NaN on either side
    n = 2_000_000
    base = -np.exp(rng.uniform(-3, 35, n))
    gap = np.where(rng.random(n) < 0.5, rng.uniform(-60, 60, n), rng.normal(0, 1e-3, n))
    a0 = np.concatenate([base, base, [-np.inf, -np.inf, 0.0, -1e15, -1e15, np.nan, 1.0, np.inf,
                                      -np.inf, np.inf, -3.0, 5.0, np.nan]])
    a1 = np.concatenate([base + gap, base, [-np.inf, -2.0, -np.inf, -1e15, -7.0, 1.0, np.nan, 1.0,
                                           np.inf, np.inf, -3.0, -1e15, np.nan]])
    g, v = p_ls(a0, a1)
    assert _same_bits(g, v), int((g.view(np.uint64) != v.view(np.uint64)).sum())

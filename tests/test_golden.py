"""The oracle against the committed golden vectors (tests/golden/make_golden.py): guards
the oracle itself against drift, on CPU."""
import os

import numpy as np
import pytest

import orclib


@pytest.mark.parametrize("kind", ["libm", "det"])
def test_oracle_reproduces_golden(kind):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "em_small.npz"))
    orc = orclib.Oracle(kind)
    em = orclib.OracleEM(orc, g["gl"], g["pos_dist"])
    em.set_params(g["indF0"], g["alpha0"], g["freq0"])
    assert em.init_emission() == 0
    assert np.array_equal(em.e_prob, g[f"{kind}_eprob0"])
    for _ in range(int(g["iters"])):
        assert em.iterate() == 0
    for name, val in (("indF", em.indF), ("alpha", em.alpha), ("freq", em.freq),
                      ("marg", em.marg), ("ind_lkl", em.ind_lkl), ("path", em.viterbi())):
        assert np.array_equal(val, g[f"{kind}_{name}"]), name


def test_golden_viterbi_recovers_most_of_the_truth():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "em_small.npz"))
    assert (g["libm_path"] == g["true_path"]).mean() > 0.8

"""pytest configuration: the `gpu` marker and shared fixtures.

CPU tests (`-m "not gpu"`) cover the oracle, the host logic and the C-ABI surface;
GPU tests (`-m gpu`) are the parity tests proper and call through the C ABI.
Nothing here reads /root/reference at run time on the GPU box: the reference's
L-BFGS-B object (oracle/_ref/libref_bfgs.so) is prebuilt by __graft_entry__.build()
in the authoring container and travels with the snapshot; tests that need it skip
when it is absent.
"""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("ngsf-hmm_amd")


@pytest.fixture(scope="session")
def orc_libm():
    import orclib
    orclib.build_oracle()
    return orclib.Oracle("libm")


@pytest.fixture(scope="session")
def orc_det():
    import orclib
    orclib.build_oracle()
    return orclib.Oracle("det")


@pytest.fixture(scope="session")
def ref_bfgs():
    import orclib
    if not orclib.RefBfgs.available():
        pytest.skip("oracle/_ref/libref_bfgs.so not built (needs /root/reference at build time)")
    return orclib.RefBfgs()


@pytest.fixture(scope="session")
def small_sim(pkg):
    """10 individuals x 600 sites, two chromosomes, some missing data."""
    d = pkg.simulate.simulate(10, 600, seed=4242, n_chrom=2, missing_rate=0.05, indF="r",
                              alpha=0.5, freq="r")
    gl = pkg.simulate.normalise_log_gl(d.gl)
    return d, gl


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False

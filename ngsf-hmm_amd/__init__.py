"""MI355X-native EM hot path of ngsF-HMM behind a C ABI (include/nghmm.h).

The directory name contains a hyphen, so import it with
``importlib.import_module("ngsf-hmm_amd")``.

This package holds only what the hot path needs:

* ``csrc/``       hand-written HIP kernels for gfx950, the L-BFGS-B state machines and
                  the C-ABI implementation -> ``libnghmm.so`` (built in-tree);
* ``hmm.py``      host-side mirror of the reference's EM interface (EM, iter_EM,
                  forward/backward/viterbi semantics) over ctypes;
* ``simulate.py`` synthetic inputs restating scripts/ngsF-HMMsim.R.

There is no CPU fallback: if ``libnghmm.so`` or a HIP device is missing, creating
an :class:`NgsFHMM` raises.
"""
from .hmm import (NgsFHMM, NgsFHMMError, Group, Chain, MODE_EXACT, MODE_FAST, GENO_PACKED, LD_INTENDED, EPROB_LD, library_path,
                  load_library,
                  build_library)
from . import simulate

__all__ = ["NgsFHMM", "NgsFHMMError", "Group", "Chain", "MODE_EXACT", "MODE_FAST", "GENO_PACKED", "LD_INTENDED", "EPROB_LD", "library_path",
           "load_library", "build_library", "simulate"]

"""The C-ABI shared library: it loads, and exports every function include/nghmm.h
declares.  No compute call is made here (no GPU in the CPU suite)."""
import ctypes as C
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    # the drop-in boundary and the measurement / debugging entry points next to it
    text = "".join(open(os.path.join(ROOT, "include", f)).read() for f in ("nghmm.h", "nghmm_debug.h"))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(nghmm_[a-z_0-9]+)\s*\(", text))
    return sorted(names)


def test_header_and_python_binding_agree(pkg):
    hm = importlib.import_module("ngsf-hmm_amd.hmm")
    assert _declared_functions() == sorted(hm.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(pkg):
    path = pkg.library_path()
    if not os.path.exists(path):
        pkg.build_library()
    lib = C.CDLL(path)
    for name in _declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/nghmm.h but not exported"
    lib.nghmm_has_hip.restype = C.c_int
    assert lib.nghmm_has_hip() == 1


def test_error_strings_are_the_reference_messages(pkg):
    L = pkg.load_library()
    assert L.nghmm_strerror(-1) == b"invalid Lkl found!"            # shared/HMM.cpp:20
    assert L.nghmm_strerror(-2) == b"Fw and Bw lkl do not match!"   # EM.cpp:169
    assert L.nghmm_strerror(-3) == b"invalid MAF!"                  # shared/HMM.cpp:146
    assert L.nghmm_strerror(-5) == b"invalid allele frequencies"    # shared/gen_func.cpp:1031


def test_no_silent_cpu_fallback(pkg):
    """Without a HIP device the product must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.NgsFHMMError) as ei:
        pkg.NgsFHMM(4, 16)
    assert "no CPU fallback" in str(ei.value)


def test_product_never_touches_the_oracle():
    """Rule: only tests/, smoke() and bench.py's cpu_baseline may use oracle/."""
    pkgdir = os.path.join(ROOT, "ngsf-hmm_amd")
    for dirpath, _, files in os.walk(pkgdir):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle/" not in text.replace("the oracle's", "") or fn == "detmath.h" or \
                    "liboracle" not in text, f"{fn} references the oracle"
                assert "liboracle" not in text and "orclib" not in text, fn

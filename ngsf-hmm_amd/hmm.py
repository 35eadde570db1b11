"""Host-side mirror of the reference's EM interface over the C ABI (include/nghmm.h).

Names follow the reference: ``EM`` (EM.cpp:27-135), ``iter_EM`` (EM.cpp:139-289),
``viterbi`` (shared/HMM.cpp:98-125), ``lkl`` (EM.cpp:449-464), the ``params`` fields
``indF / alpha / freq / marg_prob / ind_lkl / e_prob / path / tot_lkl``
(ngsF-HMM.hpp:13-52).  Errors the reference reports through ``error()`` +
``exit(-1)`` are raised as :class:`NgsFHMMError` carrying the same message.

All arithmetic happens in ``libnghmm.so`` (HIP kernels).  This module never
computes a result itself and has no fallback.
"""
from __future__ import annotations

import ctypes as C
import weakref
import math
import os
import subprocess

import numpy as np

MODE_EXACT = 0
MODE_FAST = 1
GENO_PACKED = 0x10   # OR-ed into the mode: called genotypes kept as 2-bit codes
# OR-ed into freq_est: --freq_est 2 / --e_prob 2 AS INTENDED (opt-in, parity unpinned: the
# reference aborts on both; include/nghmm.h)
LD_INTENDED = 0x20
EPROB_LD = 0x40

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = C.POINTER(C.c_double)
HOOK_FN = C.CFUNCTYPE(None, C.c_void_p)   # nghmm_hook_fn
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64)   # nghmm_allgather_fn


class NgsFHMMError(RuntimeError):
    """A fatal condition of the reference (its message) or a runtime failure."""

    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code
        self.message = message


class ModeCount(C.Structure):        # nghmm_mode_count (include/nghmm_debug.h)
    _fields_ = [("mode", C.c_uint32), ("ind_rounds", C.c_uint64)]


class MstepStats(C.Structure):
    _fields_ = [("rounds", C.c_uint32), ("points", C.c_uint64),
                ("ref_forward_calls", C.c_uint64), ("ind_rounds", C.c_uint64)]


def library_path():
    # NGHMM_LIB: another build of the same library (kernel tuning experiments)
    return os.environ.get("NGHMM_LIB") or os.path.join(_HERE, "libnghmm.so")


def build_library(verbose=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], check=True, stdout=out)
    return library_path()


def load_library():
    """dlopen libnghmm.so and declare every entry point of include/nghmm.h."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise NgsFHMMError(-11, f"{path} is missing: build it with "
                                "`python -c 'import __graft_entry__ as g; g.build()'` "
                                "(the hot path has no CPU fallback)")
    L = C.CDLL(path)
    vp, u64, u32, i32, d = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_double
    dp = c_double_p
    sig = {
        "nghmm_last_error": (C.c_char_p, []),
        "nghmm_strerror": (C.c_char_p, [i32]),
        "nghmm_has_hip": (i32, []),
        "nghmm_create": (i32, [C.POINTER(vp), u64, u64, i32, i32]),
        "nghmm_destroy": (i32, [vp]),
        "nghmm_create_replica": (i32, [C.POINTER(vp), vp]),
        "nghmm_load_gl": (i32, [vp, dp, dp]),
        "nghmm_load_gl_raw": (i32, [vp, dp, i32, i32, i32, dp]),
        "nghmm_get_gl": (i32, [vp, dp]),
        "nghmm_geno_posteriors": (i32, [vp, u64, u64, dp]),
        "nghmm_format_posteriors": (i32, [vp, u64, u64, C.c_char_p]),
        "nghmm_format_fixed6": (i32, [vp, dp, u64, u64, C.c_char_p]),
        "nghmm_load_gl_device": (i32, [vp, vp, vp]),
        "nghmm_load_begin": (i32, [vp, dp]),
        "nghmm_load_begin_dev": (i32, [vp, vp]),
        "nghmm_load_gl_raw_sites": (i32, [vp, u64, u64, dp, i32, i32, i32]),
        "nghmm_load_gl_raw_sites_dev": (i32, [vp, u64, u64, vp, i32, i32, i32]),
        "nghmm_load_geno_sites": (i32, [vp, u64, u64, C.POINTER(C.c_int8)]),
        "nghmm_load_end": (i32, [vp]),
        "nghmm_get_geno_codes_dev": (i32, [vp, u64, u64, vp]),
        "nghmm_load_geno_site_shard_dev": (i32, [vp, vp]),
        "nghmm_set_params": (i32, [vp, dp, dp, dp]),
        "nghmm_get_params": (i32, [vp, dp, dp, dp]),
        "nghmm_emission": (i32, [vp]),
        "nghmm_estep": (i32, [vp, dp]),
        "nghmm_lkl_batch": (i32, [vp, u32, C.POINTER(u32), dp, dp, dp]),
        "nghmm_mstep_indf": (i32, [vp, i32, i32, C.POINTER(MstepStats)]),
        "nghmm_bfgs_batch_host": (i32, [u64, dp, dp, i32, i32, vp, vp, C.POINTER(MstepStats)]),
        "nghmm_bfgs_batch_host2": (i32, [u64, dp, dp, i32, i32, vp, vp, C.POINTER(MstepStats), i32]),
        "nghmm_mstep_freq": (i32, [vp, i32]),
        "nghmm_estep_mstep": (i32, [vp, i32, i32, dp, C.POINTER(MstepStats), HOOK_FN, vp]),
        "nghmm_iter_em": (i32, [vp, i32, i32, i32, dp, C.POINTER(MstepStats)]),
        "nghmm_viterbi": (i32, [vp, C.POINTER(C.c_uint8)]),
        "nghmm_get_posteriors": (i32, [vp, dp]),
        "nghmm_get_emissions": (i32, [vp, dp]),
        "nghmm_shard_config": (i32, [vp, u64, u64, u64, u64]),
        "nghmm_load_gl_site_shard": (i32, [vp, dp]),
        "nghmm_load_gl_site_shard_dev": (i32, [vp, vp]),
        "nghmm_pack_posteriors_dev": (i32, [vp, u64, u64, vp]),
        "nghmm_mstep_freq_sites_dev": (i32, [vp, vp, vp]),
        "nghmm_set_freq_dev": (i32, [vp, vp]),
        "nghmm_group_setup": (i32, [C.POINTER(vp), i32]),
        "nghmm_group_iter_em": (i32, [C.POINTER(vp), i32, i32, i32, i32, dp, C.POINTER(MstepStats)]),
        "nghmm_group_mstep_freq": (i32, [C.POINTER(vp), i32, i32]),
        "nghmm_fast_layout": (i32, [vp, C.POINTER(u32), C.POINTER(u64)]),
        "nghmm_stream": (vp, [vp]),
        "nghmm_synchronize": (i32, [vp]),
        "nghmm_kernel_ms": (i32, [vp, i32, dp, C.POINTER(u32)]),
        "nghmm_set_switch": (i32, [vp, C.c_char_p, C.c_long]),
        "nghmm_debug_mode_counts": (i32, [vp, C.POINTER(ModeCount), u32, C.POINTER(u32), i32]),
        "nghmm_debug_estmaf_counts": (i32, [vp, C.POINTER(u64), i32]),
        "nghmm_alloc_host": (vp, [u64]),
        "nghmm_free_host": (None, [vp]),
        "nghmm_site_shard_bytes": (u64, [vp]),
        "nghmm_site_shard_setup": (i32, [vp, i32, i32, vp, vp, u64, ALLGATHER_FN, vp]),
        "nghmm_chain_setup": (i32, [C.POINTER(vp), i32]),
        "nghmm_chain_iter_em": (i32, [C.POINTER(vp), i32, i32, i32, i32, dp, C.POINTER(MstepStats)]),
        "nghmm_chain_mstep_freq": (i32, [C.POINTER(vp), i32, i32]),
        "nghmm_chain_viterbi": (i32, [C.POINTER(vp), i32, C.POINTER(C.c_uint8)]),
        "nghmm_viterbi_shard_forward": (i32, [vp, dp, dp]),
        "nghmm_viterbi_shard_back": (i32, [vp, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8),
                                           C.POINTER(C.c_uint8)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def _dp(a):
    return a.ctypes.data_as(c_double_p)


EXPORTED_SYMBOLS = [
    "nghmm_last_error", "nghmm_strerror", "nghmm_has_hip", "nghmm_create", "nghmm_destroy",
    "nghmm_create_replica",
    "nghmm_load_gl", "nghmm_load_gl_raw", "nghmm_get_gl", "nghmm_geno_posteriors",
    "nghmm_format_posteriors", "nghmm_format_fixed6",
    "nghmm_load_gl_device", "nghmm_load_begin", "nghmm_load_begin_dev", "nghmm_load_gl_raw_sites",
    "nghmm_load_gl_raw_sites_dev", "nghmm_load_geno_sites", "nghmm_load_end",
    "nghmm_get_geno_codes_dev", "nghmm_load_geno_site_shard_dev",
    "nghmm_set_params", "nghmm_get_params",
    "nghmm_emission", "nghmm_estep", "nghmm_lkl_batch", "nghmm_mstep_indf",
    "nghmm_bfgs_batch_host", "nghmm_bfgs_batch_host2", "nghmm_mstep_freq", "nghmm_estep_mstep",
    "nghmm_iter_em", "nghmm_viterbi", "nghmm_get_posteriors", "nghmm_get_emissions",
    "nghmm_shard_config", "nghmm_load_gl_site_shard", "nghmm_load_gl_site_shard_dev",
    "nghmm_pack_posteriors_dev",
    "nghmm_mstep_freq_sites_dev", "nghmm_set_freq_dev", "nghmm_group_setup", "nghmm_group_iter_em",
    "nghmm_group_mstep_freq",
    "nghmm_fast_layout", "nghmm_stream",
    "nghmm_synchronize",
    "nghmm_kernel_ms", "nghmm_set_switch", "nghmm_debug_mode_counts", "nghmm_debug_estmaf_counts",
    "nghmm_site_shard_bytes", "nghmm_site_shard_setup", "nghmm_viterbi_shard_forward",
    "nghmm_viterbi_shard_back", "nghmm_chain_setup", "nghmm_chain_iter_em", "nghmm_chain_mstep_freq",
    "nghmm_chain_viterbi", "nghmm_alloc_host", "nghmm_free_host",
]

OBJECTIVE_FN = C.CFUNCTYPE(C.c_double, C.c_uint32, C.c_double, C.c_double, C.c_void_p)


def bfgs_batch_host(indF, alpha, objective, indF_fixed=False, alpha_fixed=False, det_pow=None,
                    device_solver=False):
    """Lock-step batched L-BFGS-B (the indF/alpha M-step's host half) with a Python
    objective ``objective(ind, F, alpha) -> forward log-likelihood``.  Returns
    (indF, alpha, stats).  det_pow: the finite-difference step by the library's own exp / log
    (fast mode's) instead of libm's pow; device_solver: the solver type k_bfgs_advance runs on
    the GPU, on the host (nghmm_bfgs_batch_host2)."""
    L = load_library()
    F = np.array(indF, dtype=np.float64)
    A = np.array(alpha, dtype=np.float64)
    st = MstepStats()
    cb = OBJECTIVE_FN(lambda i, f, a, _u: float(objective(i, f, a)))
    if det_pow is None and not device_solver:
        rc = L.nghmm_bfgs_batch_host(len(F), _dp(F), _dp(A), int(indF_fixed), int(alpha_fixed),
                                     C.cast(cb, C.c_void_p), None, C.byref(st))
    else:
        rc = L.nghmm_bfgs_batch_host2(len(F), _dp(F), _dp(A), int(indF_fixed), int(alpha_fixed),
                                      C.cast(cb, C.c_void_p), None, C.byref(st),
                                      (1 if det_pow else 0) | (2 if device_solver else 0))
    if rc != 0:
        raise NgsFHMMError(rc, L.nghmm_strerror(rc).decode())
    return F, A, st


KERNEL_SLOTS = {"emission": 0, "forward": 1, "backward": 2, "lkl_batch": 3, "est_maf": 4,
                "viterbi": 5, "lkl_first": 6, "bfgs": 7}


class NgsFHMM:
    """Device-resident EM state of one GPU (the reference's ``params`` + ``EM``)."""

    def __init__(self, n_ind, n_sites, device=0, mode=MODE_EXACT, _replica_of=None):
        self.lib = load_library()
        self.n_ind, self.n_sites = int(n_ind), int(n_sites)
        self.mode = mode
        self._h = C.c_void_p()
        self._parent = _replica_of          # keeps the parent alive
        self._replicas = weakref.WeakSet()  # a parent closes its live replicas before itself
        if _replica_of is not None:
            self._check(self.lib.nghmm_create_replica(C.byref(self._h), _replica_of._h))
            _replica_of._replicas.add(self)
        else:
            self._check(self.lib.nghmm_create(C.byref(self._h), self.n_ind, self.n_sites, device,
                                              mode))
        self.tot_lkl = 0.0        # parse_args.cpp:31-32
        self.prev_tot_lkl = 0.0
        self.ind_lkl = np.full(self.n_ind, -math.inf)  # parse_args.cpp:412
        self.last_stats = None
        self.iterations = 0

    # -- plumbing ---------------------------------------------------------
    def _check(self, rc):
        err = getattr(self, "_shard_err", None)
        if err:                      # raised inside the site-shard all-gather callback
            e = err[0]
            del err[:]
            raise e
        if rc != 0:
            msg = self.lib.nghmm_last_error().decode() or self.lib.nghmm_strerror(rc).decode()
            raise NgsFHMMError(rc, msg)

    def replica(self):
        """A handle that shares this one's (loaded) data and owns its own EM state and stream
        (nghmm_create_replica): multi-start runs, ngsF-HMM.sh:77-101."""
        return NgsFHMM(self.n_ind, self.n_sites, mode=self.mode, _replica_of=self)

    def close(self, _finalizing=False):
        """Destroys the handle.  A parent whose replicas are still open refuses, as the library
        does (they share its data): close them first.  Garbage collection (no order is
        guaranteed there) closes a parent's remaining replicas before the parent."""
        if getattr(self, "_h", None) is not None and self._h:
            live = [r for r in list(getattr(self, "_replicas", ())) if not r.closed]
            if live and not _finalizing:
                raise NgsFHMMError(-10, f"{len(live)} replica(s) of this handle are still open: "
                                        "close them first")
            for r in live:
                r.close(_finalizing=True)
            self._check(self.lib.nghmm_destroy(self._h))
            self._h = C.c_void_p()

    @property
    def closed(self):
        return not self._h

    def set_switch(self, name, value):
        """A measurement / debugging switch of this handle (nghmm_set_switch, include/nghmm_debug.h)."""
        self._check(self.lib.nghmm_set_switch(self._h, name.encode(), int(value)))

    def __del__(self):
        try:
            self.close(_finalizing=True)
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    # -- data --------------------------------------------------------------
    def load(self, gl, pos_dist):
        """gl: [S][I][3] normalised natural-log GLs (binary --geno order);
        pos_dist: [S] distances in Mb, inf at chromosome starts."""
        gl = np.ascontiguousarray(gl, dtype=np.float64)
        pos_dist = np.ascontiguousarray(pos_dist, dtype=np.float64)
        if gl.shape != (self.n_sites, self.n_ind, 3) or pos_dist.shape != (self.n_sites,):
            raise NgsFHMMError(-10, f"load: shapes {gl.shape} {pos_dist.shape} do not match "
                                    f"({self.n_sites}, {self.n_ind}, 3)")
        self._check(self.lib.nghmm_load_gl(self._h, _dp(gl), _dp(pos_dist)))

    def load_raw(self, gl_raw, pos_dist, space=0, call_geno=False, check_nan=False):
        """Raw genotype likelihoods as read from the input file; conversion to log space,
        normalisation and optional genotype calling happen on the device
        (nghmm_load_gl_raw).  space: 0 log, 1 normal space from a binary file, 2 normal
        space from a text file."""
        gl_raw = np.ascontiguousarray(gl_raw, dtype=np.float64)
        pos_dist = np.ascontiguousarray(pos_dist, dtype=np.float64)
        if gl_raw.shape != (self.n_sites, self.n_ind, 3) or pos_dist.shape != (self.n_sites,):
            raise NgsFHMMError(-10, f"load_raw: shapes {gl_raw.shape} {pos_dist.shape} do not "
                                    f"match ({self.n_sites}, {self.n_ind}, 3)")
        self._check(self.lib.nghmm_load_gl_raw(self._h, _dp(gl_raw), int(space), int(call_geno),
                                               int(check_nan), _dp(pos_dist)))

    @property
    def gl(self):
        """[S][I][3] prepared (natural-log, normalised) genotype likelihoods."""
        out = np.empty((self.n_sites, self.n_ind, 3))
        self._check(self.lib.nghmm_get_gl(self._h, _dp(out)))
        return out

    def geno_posteriors(self, site_begin=0, n_sites=None):
        """[n_sites][I][3] genotype posteriors of the .geno output (EM.cpp:367-376), from the
        path of the last viterbi() call."""
        n = self.n_sites - site_begin if n_sites is None else n_sites
        out = np.empty((n, self.n_ind, 3))
        self._check(self.lib.nghmm_geno_posteriors(self._h, int(site_begin), int(n), _dp(out)))
        return out

    def format_fixed6(self, values):
        """printf("%f") of a [rows][cols] array of values in [0, 1], on the device: tab-separated,
        one line per row, as bytes."""
        v = np.ascontiguousarray(values, dtype=np.float64)
        rows, cols = v.shape
        buf = C.create_string_buffer(rows * cols * 9)
        self._check(self.lib.nghmm_format_fixed6(self._h, _dp(v), rows, cols, buf))
        return buf.raw

    def format_posteriors(self, ind_begin=0, n_ind=None):
        """The .ibd file's posterior lines of individuals [ind_begin, ind_begin + n_ind) as
        bytes: "%f" values, tab-separated, one line per individual (EM.cpp:347-353)."""
        n = self.n_ind - ind_begin if n_ind is None else n_ind
        buf = C.create_string_buffer(n * 9 * self.n_sites)
        self._check(self.lib.nghmm_format_posteriors(self._h, int(ind_begin), int(n), buf))
        return buf.raw

    def load_chunks(self, pos_dist, chunks, space=0, call_geno=False, check_nan=False):
        """Chunked loading (nghmm_load_begin / _sites / _end): ``chunks`` yields
        ``(site_begin, array)`` with raw likelihoods [n][I][3] (float64) or reader genotypes
        [n][I] (int8: -1 missing, 0, 1, 2)."""
        pos_dist = np.ascontiguousarray(pos_dist, dtype=np.float64)
        self._check(self.lib.nghmm_load_begin(self._h, _dp(pos_dist)))
        for s0, a in chunks:
            if a.dtype == np.int8:
                a = np.ascontiguousarray(a)
                self._check(self.lib.nghmm_load_geno_sites(
                    self._h, int(s0), a.shape[0], a.ctypes.data_as(C.POINTER(C.c_int8))))
            else:
                a = np.ascontiguousarray(a, dtype=np.float64)
                self._check(self.lib.nghmm_load_gl_raw_sites(
                    self._h, int(s0), a.shape[0], _dp(a), int(space), int(call_geno),
                    int(check_nan)))
        self._check(self.lib.nghmm_load_end(self._h))

    def load_chunks_device(self, pos_ptr, chunks, space=0, call_geno=False):
        """The same from device buffers: ``chunks`` yields (site_begin, n_sites, data_ptr) of
        raw likelihoods [n][I][3]."""
        self._check(self.lib.nghmm_load_begin_dev(self._h, C.c_void_p(pos_ptr)))
        for s0, n, ptr in chunks:
            self._check(self.lib.nghmm_load_gl_raw_sites_dev(self._h, int(s0), int(n),
                                                             C.c_void_p(ptr), int(space),
                                                             int(call_geno), 0))
        self._check(self.lib.nghmm_load_end(self._h))

    def load_device(self, gl_ptr, pos_ptr):
        """Same as load() from raw device pointers (e.g. torch tensors' data_ptr())."""
        self._check(self.lib.nghmm_load_gl_device(self._h, C.c_void_p(gl_ptr), C.c_void_p(pos_ptr)))

    def set_params(self, indF=None, alpha=None, freq=None):
        def prep(a, n):
            if a is None:
                return None
            return np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (n,)))
        a, b, c = prep(indF, self.n_ind), prep(alpha, self.n_ind), prep(freq, self.n_sites)
        self._check(self.lib.nghmm_set_params(self._h, _dp(a) if a is not None else None,
                                              _dp(b) if b is not None else None,
                                              _dp(c) if c is not None else None))

    def _get(self, which):
        out = np.empty(self.n_sites if which == 2 else self.n_ind)
        args = [None, None, None]
        args[which] = _dp(out)
        self._check(self.lib.nghmm_get_params(self._h, *args))
        return out

    @property
    def indF(self):
        return self._get(0)

    @property
    def alpha(self):
        return self._get(1)

    @property
    def freq(self):
        return self._get(2)

    @property
    def marg_prob(self):
        """[I][S] posterior of the IBD state (marg_prob[i][s][1]) of the last E-step."""
        out = np.empty((self.n_ind, self.n_sites))
        self._check(self.lib.nghmm_get_posteriors(self._h, _dp(out)))
        return out

    @property
    def e_prob(self):
        """[I][S][2] log emission probabilities."""
        out = np.empty((self.n_ind, self.n_sites, 2))
        self._check(self.lib.nghmm_get_emissions(self._h, _dp(out)))
        return out

    # -- the hot path ------------------------------------------------------
    def init_emission(self):
        """parse_args.cpp:372-387."""
        self._check(self.lib.nghmm_emission(self._h))

    def estep(self):
        """EM.cpp:147-185; returns ind_lkl."""
        self._check(self.lib.nghmm_estep(self._h, _dp(self.ind_lkl)))
        return self.ind_lkl

    def lkl(self, ind, F, alpha):
        """Forward log-likelihood of (individual, F, alpha) points (EM.cpp:449-464 negated)."""
        ind = np.ascontiguousarray(ind, dtype=np.uint32)
        F = np.ascontiguousarray(F, dtype=np.float64)
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        out = np.empty(len(ind))
        self._check(self.lib.nghmm_lkl_batch(self._h, len(ind),
                                             ind.ctypes.data_as(C.POINTER(C.c_uint32)), _dp(F),
                                             _dp(alpha), _dp(out)))
        return out

    def mstep_indf(self, indF_fixed=False, alpha_fixed=False):
        st = MstepStats()
        self._check(self.lib.nghmm_mstep_indf(self._h, int(indF_fixed), int(alpha_fixed),
                                              C.byref(st)))
        self.last_stats = st
        return st

    def mstep_freq(self, freq_est=1):
        self._check(self.lib.nghmm_mstep_freq(self._h, int(freq_est)))

    def layout(self):
        """(waves per individual, sites per lane) of the fast-mode site layout; (0, 0) in
        exact mode."""
        c, t = C.c_uint32(0), C.c_uint64(0)
        self._check(self.lib.nghmm_fast_layout(self._h, C.byref(c), C.byref(t)))
        return int(c.value), int(t.value)

    def estep_mstep(self, indF_fixed=False, alpha_fixed=False, after_estep=None):
        """E-step and indF/alpha M-step of one iteration in one call (nghmm_estep_mstep);
        ``after_estep()`` runs as soon as the posteriors are final.  An exception raised
        in it is re-raised here after the call returns."""
        st = MstepStats()
        err = []

        def _hook(_user):
            try:
                after_estep()
            except BaseException as e:      # must not propagate through the C frame
                err.append(e)

        cb = HOOK_FN(_hook) if after_estep is not None else C.cast(None, HOOK_FN)
        rc = self.lib.nghmm_estep_mstep(self._h, int(indF_fixed), int(alpha_fixed),
                                        _dp(self.ind_lkl), C.byref(st), cb, None)
        if err:
            raise err[0]
        self._check(rc)
        self.last_stats = st
        return st

    def iter_EM(self, freq_est=1, indF_fixed=False, alpha_fixed=False):
        """One EM iteration (EM.cpp:139-289)."""
        st = MstepStats()
        self._check(self.lib.nghmm_iter_em(self._h, int(freq_est), int(indF_fixed),
                                           int(alpha_fixed), _dp(self.ind_lkl), C.byref(st)))
        self.last_stats = st
        return st

    def EM(self, freq_est=1, indF_fixed=False, alpha_fixed=False, min_iters=10, max_iters=100,
           min_epsilon=1e-5, callback=None):
        """The iteration loop and convergence test of EM.cpp:27-103 (host control only).
        Returns the number of iterations run."""
        it = 0
        max_lkl_epsilon = -math.inf
        prev_ind_lkl = np.full(self.n_ind, -math.inf)
        while ((self.prev_tot_lkl - self.tot_lkl > min_epsilon or max_lkl_epsilon > min_epsilon
                or it < min_iters) and it < max_iters):
            it += 1
            self.iter_EM(freq_est, indF_fixed, alpha_fixed)
            self.prev_tot_lkl = self.tot_lkl
            tot = 0.0
            for v in self.ind_lkl:          # EM.cpp:77-79: summation in individual order
                tot += float(v)
            self.tot_lkl = tot
            with np.errstate(invalid="ignore", divide="ignore"):
                eps = (self.ind_lkl - prev_ind_lkl) / np.abs(prev_ind_lkl)
            # array_max_pos (gen_func.cpp:73-84): strict '>' from -inf, NaN never wins
            best, mx = 0, -math.inf
            for i, v in enumerate(eps):
                if v > mx:
                    best, mx = i, v
            max_lkl_epsilon = float(eps[best])
            prev_ind_lkl = self.ind_lkl.copy()
            if callback:
                callback(it, self)
        self.iterations = it
        return it

    # -- site shards (include/nghmm.h, "shard the SITES") ----------------------
    def site_shard_bytes(self):
        return int(self.lib.nghmm_site_shard_bytes(self._h))

    def site_shard_setup(self, rank, world, send_ptr, recv_ptr, nbytes, allgather):
        """This handle's sites are range `rank` of `world`; ``allgather(n_bytes)`` gathers the
        first n_bytes of the send buffer of every range into the receive buffer, ordered on
        the handle's stream (nghmm_site_shard_setup).  An exception raised in it makes the
        library call that needed it fail; it is re-raised by _check."""
        self._shard_err = []

        def _cb(_user, n):
            try:
                allgather(int(n))
                return 0
            except BaseException as e:      # must not propagate through the C frame
                self._shard_err.append(e)
                return 1

        self._shard_cb = ALLGATHER_FN(_cb) if world > 1 else C.cast(None, ALLGATHER_FN)
        self._check(self.lib.nghmm_site_shard_setup(self._h, int(rank), int(world), C.c_void_p(send_ptr),
                                                    C.c_void_p(recv_ptr), int(nbytes), self._shard_cb,
                                                    None))

    def viterbi_shard_forward(self, scores_in=None):
        """Forward half of the decoding over a chain of site shards: scores_in = what the range
        before ended with ([I][2]; None on the first range); returns this range's."""
        out = np.empty((self.n_ind, 2))
        if scores_in is not None:
            scores_in = np.ascontiguousarray(scores_in, dtype=np.float64).reshape(self.n_ind, 2)
        self._check(self.lib.nghmm_viterbi_shard_forward(
            self._h, _dp(scores_in) if scores_in is not None else None, _dp(out)))
        return out

    def viterbi_shard_back(self, state_after=None):
        """Backward half: state_after = the state the range after found for this range's last
        site ([I]; None on the last range); returns (state at the site in front of this range's
        first [I], path [I][n_sites])."""
        u8p = C.POINTER(C.c_uint8)
        before = np.empty(self.n_ind, dtype=np.uint8)
        path = np.empty((self.n_ind, self.n_sites), dtype=np.uint8)
        if state_after is not None:
            state_after = np.ascontiguousarray(state_after, dtype=np.uint8).reshape(self.n_ind)
        self._check(self.lib.nghmm_viterbi_shard_back(
            self._h, state_after.ctypes.data_as(u8p) if state_after is not None else None,
            before.ctypes.data_as(u8p), path.ctypes.data_as(u8p)))
        return before, path

    def viterbi(self):
        """[I][S] most probable IBD path (EM.cpp:105-116)."""
        path = np.empty((self.n_ind, self.n_sites), dtype=np.uint8)
        self._check(self.lib.nghmm_viterbi(self._h, path.ctypes.data_as(C.POINTER(C.c_uint8))))
        return path

    # -- measurement -------------------------------------------------------
    def kernel_ms(self, name):
        """(milliseconds, launches) of a kernel family in the last call that ran it.  Fast mode's
        M-step and fused iteration time their kernels only after ``set_switch("spans", 1)`` (the
        events cost 0.05 ms per iteration) and report 0 without it; include/nghmm.h."""
        ms = C.c_double(0)
        n = C.c_uint32(0)
        self._check(self.lib.nghmm_kernel_ms(self._h, KERNEL_SLOTS[name], C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def synchronize(self):
        self._check(self.lib.nghmm_synchronize(self._h))

    @staticmethod
    def mode_name(mode):
        """A loop-body version of the objective kernels as debug_modes prints it: 2F2As2 = two F
        probes, two alpha probes, small-alpha (kappa) form, degree-2 alpha probes."""
        if mode == 0xffffffff:
            return "rounds_of_mixed_versions"
        if mode == 0:
            return "general"
        return (f"{(mode >> 2) & 3}F{mode & 3}A" + ("s" if mode & 0x200 else "") +
                ("2" if mode & 0x400 else "") + ("e" if mode & 0x800 else ""))

    def mode_counts(self, reset=False):
        """{kernel version: individual-rounds it evaluated} of the objective rounds so far
        (nghmm_debug_mode_counts)."""
        buf = (ModeCount * 160)()
        n = C.c_uint32(0)
        self._check(self.lib.nghmm_debug_mode_counts(self._h, buf, 160, C.byref(n), int(reset)))
        return {self.mode_name(buf[k].mode): int(buf[k].ind_rounds) for k in range(min(n.value, 160))}

    def estmaf_counts(self, reset=False):
        """Sites of the allele-frequency step that left its common route (nghmm_debug_estmaf_counts)."""
        v = (C.c_uint64 * 5)()
        self._check(self.lib.nghmm_debug_estmaf_counts(self._h, v, int(reset)))
        return {"check_failed": int(v[0]), "second_interval": int(v[1]), "third_interval": int(v[2]),
                "log_space": int(v[3]), "exact_tail": int(v[4])}


class Group:
    """n handles of one process as one cohort (nghmm_group_setup / nghmm_group_iter_em): handle
    r holds the individuals [r I, (r+1) I) for all sites."""

    def __init__(self, handles):
        self.handles = list(handles)
        self.lib = self.handles[0].lib
        n = len(self.handles)
        self._arr = (C.c_void_p * n)(*[h._h for h in self.handles])
        self.handles[0]._check(self.lib.nghmm_group_setup(self._arr, n))
        self.ind_lkl = np.full(n * self.handles[0].n_ind, -math.inf)

    def _members_open(self):
        # the group keeps its members alive; one that was closed by hand leaves a dangling
        # pointer in the array the library is given
        if any(h.closed for h in self.handles):
            raise NgsFHMMError(-10, "a member of this group has been closed")

    def mstep_freq(self, freq_est=1):
        self._members_open()
        self.handles[0]._check(self.lib.nghmm_group_mstep_freq(self._arr, len(self.handles),
                                                               int(freq_est)))

    def iter_EM(self, freq_est=1, indF_fixed=False, alpha_fixed=False):
        self._members_open()
        st = MstepStats()
        self.handles[0]._check(self.lib.nghmm_group_iter_em(
            self._arr, len(self.handles), int(freq_est), int(indF_fixed), int(alpha_fixed),
            _dp(self.ind_lkl), C.byref(st)))
        return st


class Chain:
    """n fast-mode handles of one process as one data set cut along the SITE axis
    (nghmm_chain_setup / nghmm_chain_iter_em): handle r holds all individuals for the r-th
    site range; log-likelihoods, indF and alpha are the chain's and equal on every handle."""

    def __init__(self, handles):
        self.handles = list(handles)
        self.lib = self.handles[0].lib
        n = len(self.handles)
        self._arr = (C.c_void_p * n)(*[h._h for h in self.handles])
        self.handles[0]._check(self.lib.nghmm_chain_setup(self._arr, n))
        self.n_ind = self.handles[0].n_ind
        self.n_sites = sum(h.n_sites for h in self.handles)
        self.ind_lkl = np.full(self.n_ind, -math.inf)

    def _members_open(self):
        if any(h.closed for h in self.handles):
            raise NgsFHMMError(-10, "a member of this chain has been closed")

    def mstep_freq(self, freq_est=1):
        self._members_open()
        self.handles[0]._check(self.lib.nghmm_chain_mstep_freq(self._arr, len(self.handles),
                                                               int(freq_est)))

    def iter_EM(self, freq_est=1, indF_fixed=False, alpha_fixed=False):
        self._members_open()
        st = MstepStats()
        self.handles[0]._check(self.lib.nghmm_chain_iter_em(
            self._arr, len(self.handles), int(freq_est), int(indF_fixed), int(alpha_fixed),
            _dp(self.ind_lkl), C.byref(st)))
        return st

    def viterbi(self):
        self._members_open()
        path = np.empty((self.n_ind, self.n_sites), dtype=np.uint8)
        self.handles[0]._check(self.lib.nghmm_chain_viterbi(
            self._arr, len(self.handles), path.ctypes.data_as(C.POINTER(C.c_uint8))))
        return path

    @property
    def freq(self):
        return np.concatenate([h.freq for h in self.handles])

    @property
    def marg_prob(self):
        return np.concatenate([h.marg_prob for h in self.handles], axis=1)

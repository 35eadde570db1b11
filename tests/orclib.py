"""ctypes bindings for the CPU oracle (oracle/liboracle_{libm,det}.so) and, when it
has been built, the reference's own L-BFGS-B object (oracle/_ref/libref_bfgs.so).

Test infrastructure: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

c_double_p = C.POINTER(C.c_double)
OBJECTIVE = C.CFUNCTYPE(C.c_double, c_double_p, C.c_void_p)
# mangled name of findmax_bfgs in the reference object (g++ ABI)
REF_FINDMAX_SYMBOL = "_Z12findmax_bfgsiPdPKvPFdPKdS1_EPFvS3_S_ES_S_Pii"


def build_oracle():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True, stdout=subprocess.DEVNULL)


def _dp(a):
    return a.ctypes.data_as(c_double_p)


class LklData(C.Structure):
    _fields_ = [("e_prob", c_double_p), ("pos_dist", c_double_p), ("S", C.c_uint64),
                ("n_calls", C.c_uint64), ("failed", C.c_int)]


class Oracle:
    """One of the two oracle builds ('libm' or 'det')."""

    def __init__(self, kind="libm"):
        path = os.path.join(ORACLE_DIR, f"liboracle_{kind}.so")
        if not os.path.exists(path):
            build_oracle()
        self.kind = kind
        L = self.lib = C.CDLL(path)
        d, dp, u64, i32 = C.c_double, c_double_p, C.c_uint64, C.c_int
        ip = C.POINTER(C.c_int)
        sig = {
            "orc_detmath": (i32, []),
            "orc_exp": (d, [d]), "orc_log": (d, [d]),
            "orc_logsum": (d, [dp, u64]),
            "orc_calc_trans": (d, [i32, i32, d, d, d]),
            "orc_calc_hwe": (None, [dp, d, d, i32]),
            "orc_post_prob": (None, [dp, dp, dp]),
            "orc_call_geno": (None, [dp]),
            "orc_prepare_gl": (None, [dp, u64, i32, i32]),
            "orc_check_interv": (d, [d, ip]),
            "orc_calc_emission": (d, [dp, d, i32, ip]),
            "orc_est_maf": (d, [u64, dp, dp, ip]),
            "orc_forward": (i32, [dp, dp, d, dp, dp, u64, dp]),
            "orc_backward": (i32, [dp, dp, d, dp, dp, u64, dp]),
            "orc_viterbi": (d, [dp, d, dp, dp, u64, C.c_char_p]),
            "orc_lkl": (d, [dp, C.c_void_p]),
            "orc_findmax_bfgs": (d, [i32, dp, C.c_void_p, C.c_void_p, C.c_void_p, dp, dp, ip, i32]),
            "orc_em_create": (C.c_void_p, [u64, u64, dp, dp]),
            "orc_em_destroy": (None, [C.c_void_p]),
            "orc_em_set_params": (None, [C.c_void_p, dp, dp, dp]),
            "orc_em_set_optimizer": (None, [C.c_void_p, C.c_void_p]),
            "orc_em_init_emission": (i32, [C.c_void_p]),
            "orc_em_iter": (i32, [C.c_void_p, i32, i32, i32, i32, i32]),
            "orc_em_estep": (i32, [C.c_void_p, i32]),
            "orc_em_mstep_indf": (i32, [C.c_void_p, i32, i32, i32]),
            "orc_em_mstep_freq": (i32, [C.c_void_p, i32, i32]),
            "orc_em_mstep_freq_ld": (i32, [C.c_void_p, i32, i32]),
            "orc_pair_freq_iter": (u64, [dp, dp, dp, u64]),
            "orc_haplo_freq": (i32, [dp, dp, dp, d, d, u64]),
            "orc_joint_geno_prob": (d, [dp, i32, i32, i32]),
            "orc_calc_emission_ld": (d, [dp, dp, dp, d, d, i32, ip]),
            "orc_em_run": (i32, [C.c_void_p, i32, i32, i32, i32, i32, d, i32]),
            "orc_em_viterbi": (i32, [C.c_void_p, C.POINTER(C.c_uint8), i32]),
            "orc_em_geno_post": (None, [C.c_void_p, C.POINTER(C.c_uint8), dp]),
            "orc_em_indF": (dp, [C.c_void_p]), "orc_em_alpha": (dp, [C.c_void_p]),
            "orc_em_freq": (dp, [C.c_void_p]), "orc_em_ind_lkl": (dp, [C.c_void_p]),
            "orc_em_marg": (dp, [C.c_void_p]), "orc_em_eprob": (dp, [C.c_void_p]),
            "orc_em_tot_lkl": (d, [C.c_void_p]),
            "orc_em_lkl_calls": (u64, [C.c_void_p]), "orc_em_maf_passes": (u64, [C.c_void_p]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args

    # ---- small kernels -------------------------------------------------
    def logsum(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return self.lib.orc_logsum(_dp(a), len(a))

    def calc_hwe(self, maf, F, log_scale=True):
        out = np.empty(3)
        self.lib.orc_calc_hwe(_dp(out), maf, F, int(log_scale))
        return out

    def post_prob(self, lkl, prior=None):
        """shared/gen_func.cpp:920-932: normalised log posterior of one cell."""
        lkl = np.ascontiguousarray(lkl, dtype=np.float64)
        pp = np.empty(3)
        pr = None if prior is None else np.ascontiguousarray(prior, dtype=np.float64)
        self.lib.orc_post_prob(_dp(pp), _dp(lkl), _dp(pr) if pr is not None else None)
        return pp

    def calc_emission(self, gl, maf, k):
        gl = np.ascontiguousarray(gl, dtype=np.float64)
        bad = C.c_int(0)
        v = self.lib.orc_calc_emission(_dp(gl), maf, k, C.byref(bad))
        return v, bool(bad.value)

    def prepare_gl(self, gl_raw, space=0, call_geno=False):
        """Input preparation of every cell (read_data.cpp:36-40,89-98; ngsF-HMM.cpp:101-117):
        raw [..., 3] values -> normalised natural-log likelihoods (a new array)."""
        out = np.ascontiguousarray(gl_raw, dtype=np.float64).copy()
        self.lib.orc_prepare_gl(_dp(out), out.size // 3, int(space), int(call_geno))
        return out

    def est_maf(self, gl_site, indF):
        gl_site = np.ascontiguousarray(gl_site, dtype=np.float64)
        indF = np.ascontiguousarray(indF, dtype=np.float64)
        n = C.c_int(0)
        f = self.lib.orc_est_maf(len(indF), _dp(gl_site), _dp(indF), C.byref(n))
        return f, n.value

    # --freq_est 2 / --e_prob 2 as intended (parity unpinned: the reference aborts)
    def haplo_freq(self, p1, p2, maf1, maf2):
        """p1, p2: [n][3] genotype probabilities (normal space) -> (hap_freq[4], iterations)."""
        p1 = np.ascontiguousarray(p1, dtype=np.float64)
        p2 = np.ascontiguousarray(p2, dtype=np.float64)
        hap = np.zeros(4)
        it = self.lib.orc_haplo_freq(_dp(hap), _dp(p1), _dp(p2), float(maf1), float(maf2), len(p1))
        return hap, it

    def calc_emission_ld(self, hap, gl_p, gl_c, maf_p, maf_c, F):
        hap = np.ascontiguousarray(hap, dtype=np.float64)
        gl_p = np.ascontiguousarray(gl_p, dtype=np.float64)
        gl_c = np.ascontiguousarray(gl_c, dtype=np.float64)
        bad = C.c_int(0)
        return self.lib.orc_calc_emission_ld(_dp(hap), _dp(gl_p), _dp(gl_c), float(maf_p),
                                             float(maf_c), int(F), C.byref(bad))

    def forward(self, q, alpha, e_prob, pos_dist, store=True):
        e_prob = np.ascontiguousarray(e_prob, dtype=np.float64)
        pos_dist = np.ascontiguousarray(pos_dist, dtype=np.float64)
        S = len(pos_dist)
        q = np.ascontiguousarray(q, dtype=np.float64)
        Fw = np.empty((S + 1, 2)) if store else None
        lkl = C.c_double(0)
        rc = self.lib.orc_forward(_dp(Fw) if store else None, _dp(q), alpha, _dp(e_prob),
                                  _dp(pos_dist), S, C.byref(lkl))
        return rc, lkl.value, Fw

    def backward(self, q, alpha, e_prob, pos_dist):
        e_prob = np.ascontiguousarray(e_prob, dtype=np.float64)
        pos_dist = np.ascontiguousarray(pos_dist, dtype=np.float64)
        S = len(pos_dist)
        q = np.ascontiguousarray(q, dtype=np.float64)
        Bw = np.empty((S + 1, 2))
        lkl = C.c_double(0)
        rc = self.lib.orc_backward(_dp(Bw), _dp(q), alpha, _dp(e_prob), _dp(pos_dist), S,
                                   C.byref(lkl))
        return rc, lkl.value, Bw

    def viterbi(self, q, alpha, e_prob, pos_dist):
        e_prob = np.ascontiguousarray(e_prob, dtype=np.float64)
        pos_dist = np.ascontiguousarray(pos_dist, dtype=np.float64)
        S = len(pos_dist)
        q = np.ascontiguousarray(q, dtype=np.float64)
        buf = C.create_string_buffer(S + 1)
        v = self.lib.orc_viterbi(_dp(q), alpha, _dp(e_prob), _dp(pos_dist), S, buf)
        return v, np.frombuffer(buf.raw, dtype=np.uint8)[: S + 1].copy()

    def lkl(self, x, e_prob, pos_dist):
        """-forward log-likelihood at x = (F, alpha) (EM.cpp:449-464)."""
        e_prob = np.ascontiguousarray(e_prob, dtype=np.float64)
        pos_dist = np.ascontiguousarray(pos_dist, dtype=np.float64)
        d = LklData(_dp(e_prob), _dp(pos_dist), len(pos_dist), 0, 0)
        x = np.ascontiguousarray(x, dtype=np.float64)
        return self.lib.orc_lkl(_dp(x), C.cast(C.byref(d), C.c_void_p))


class OracleEM:
    """Whole-EM state of one oracle build."""

    def __init__(self, orc: Oracle, gl, pos_dist):
        self.orc = orc
        gl = np.ascontiguousarray(gl, dtype=np.float64)
        pos_dist = np.ascontiguousarray(pos_dist, dtype=np.float64)
        self.S, self.I = gl.shape[0], gl.shape[1]
        self.h = orc.lib.orc_em_create(self.I, self.S, _dp(gl), _dp(pos_dist))
        self._keep = None

    def close(self):
        if self.h:
            self.orc.lib.orc_em_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_params(self, indF=None, alpha=None, freq=None):
        def prep(a, n):
            if a is None:
                return None
            a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (n,)))
            return a
        a, b, c = prep(indF, self.I), prep(alpha, self.I), prep(freq, self.S)
        self.orc.lib.orc_em_set_params(self.h, _dp(a) if a is not None else None,
                                       _dp(b) if b is not None else None,
                                       _dp(c) if c is not None else None)

    def use_reference_optimizer(self, ref):
        """Run the indF/alpha M-step with the reference's own findmax_bfgs."""
        self._keep = ref
        self.orc.lib.orc_em_set_optimizer(self.h, C.cast(ref.findmax, C.c_void_p))

    def init_emission(self):
        return self.orc.lib.orc_em_init_emission(self.h)

    def iterate(self, freq_est=1, indF_fixed=False, alpha_fixed=False, n_threads=1, thread_freq=False):
        return self.orc.lib.orc_em_iter(self.h, freq_est, int(indF_fixed), int(alpha_fixed),
                                        n_threads, int(thread_freq))

    def estep(self, n_threads=1):
        return self.orc.lib.orc_em_estep(self.h, n_threads)

    def mstep_indf(self, indF_fixed=False, alpha_fixed=False, n_threads=1):
        return self.orc.lib.orc_em_mstep_indf(self.h, int(indF_fixed), int(alpha_fixed), n_threads)

    def mstep_freq(self, freq_est=1, n_threads=1):
        return self.orc.lib.orc_em_mstep_freq(self.h, freq_est, n_threads)

    def mstep_freq_ld(self, freq_est=2, e_prob_calc=1):
        """--freq_est 2 / --e_prob 2 as intended (parity unpinned: the reference aborts)."""
        return self.orc.lib.orc_em_mstep_freq_ld(self.h, freq_est, e_prob_calc)

    def run(self, freq_est=1, indF_fixed=False, alpha_fixed=False, min_iters=10, max_iters=100,
            min_epsilon=1e-5, n_threads=1):
        return self.orc.lib.orc_em_run(self.h, freq_est, int(indF_fixed), int(alpha_fixed),
                                       min_iters, max_iters, min_epsilon, n_threads)

    def viterbi(self, n_threads=1):
        path = np.empty((self.I, self.S), dtype=np.uint8)
        self.orc.lib.orc_em_viterbi(self.h, path.ctypes.data_as(C.POINTER(C.c_uint8)), n_threads)
        return path

    def geno_post(self, path):
        path = np.ascontiguousarray(path, dtype=np.uint8)
        out = np.empty((self.S, self.I, 3))
        self.orc.lib.orc_em_geno_post(self.h, path.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(out))
        return out

    def _arr(self, fn, shape):
        p = fn(self.h)
        return np.ctypeslib.as_array(p, shape=shape).copy()

    @property
    def indF(self):
        return self._arr(self.orc.lib.orc_em_indF, (self.I,))

    @property
    def alpha(self):
        return self._arr(self.orc.lib.orc_em_alpha, (self.I,))

    @property
    def freq(self):
        return self._arr(self.orc.lib.orc_em_freq, (self.S,))

    @property
    def ind_lkl(self):
        return self._arr(self.orc.lib.orc_em_ind_lkl, (self.I,))

    @property
    def marg(self):
        """[I][S] posterior of the IBD state (marg_prob[i][s][1])."""
        return self._arr(self.orc.lib.orc_em_marg, (self.I, self.S, 2))[:, :, 1].copy()

    @property
    def marg_both(self):
        return self._arr(self.orc.lib.orc_em_marg, (self.I, self.S, 2))

    @property
    def e_prob(self):
        return self._arr(self.orc.lib.orc_em_eprob, (self.I, self.S, 2))

    @property
    def tot_lkl(self):
        return self.orc.lib.orc_em_tot_lkl(self.h)

    @property
    def lkl_calls(self):
        return self.orc.lib.orc_em_lkl_calls(self.h)

    @property
    def maf_passes(self):
        return self.orc.lib.orc_em_maf_passes(self.h)


class RefBfgs:
    """The reference's own L-BFGS-B (shared/bfgs.cpp) compiled in oracle/_ref."""

    PATH = os.path.join(ORACLE_DIR, "_ref", "libref_bfgs.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self):
        self.lib = C.CDLL(self.PATH)
        self.findmax = getattr(self.lib, REF_FINDMAX_SYMBOL)
        self.findmax.restype = C.c_double
        self.findmax.argtypes = [C.c_int, c_double_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 c_double_p, c_double_p, C.POINTER(C.c_int), C.c_int]


def run_findmax(fn, objective, x0, lb, ub, data=None):
    """Call a findmax_bfgs-shaped function with a Python or C objective.
    Returns (x_final, return_value)."""
    x = np.array(x0, dtype=np.float64)
    lb = np.array(lb, dtype=np.float64)
    ub = np.array(ub, dtype=np.float64)
    nbd = (C.c_int * len(x))(*([2] * len(x)))
    r = fn(len(x), _dp(x), data, objective, None, _dp(lb), _dp(ub), nbd, -1)
    return x, r


class HpAnchor:
    """oracle/hp_anchor.c: the model of forward / backward / posteriors / est_maf evaluated in
    binary128, independent of the oracle's restatement (no shared code, another formulation)."""

    def __init__(self):
        path = os.path.join(ORACLE_DIR, "liboracle_hp.so")
        if not os.path.exists(path):
            build_oracle()
        L = self.lib = C.CDLL(path)
        L.hp_forward_backward.restype = C.c_double
        L.hp_forward_backward.argtypes = [c_double_p, c_double_p, c_double_p, C.c_uint64,
                                          C.c_double, C.c_double, c_double_p]
        L.hp_est_maf.restype = C.c_double
        L.hp_est_maf.argtypes = [C.c_uint64, c_double_p, c_double_p, C.POINTER(C.c_int)]
        L.hp_haplo_freq.restype = C.c_int
        L.hp_haplo_freq.argtypes = [c_double_p, c_double_p, c_double_p, C.c_double, C.c_double,
                                    C.c_uint64]
        L.hp_emission_ld.restype = C.c_double
        L.hp_emission_ld.argtypes = [c_double_p, c_double_p, c_double_p, C.c_double, C.c_int]

    def forward_backward(self, gl_ind, freq, pos_dist, indF, alpha, want_post=True):
        """gl_ind [S][3] log GLs of one individual -> (log-likelihood, posteriors [S] or None)."""
        gl_ind = np.ascontiguousarray(gl_ind, dtype=np.float64)
        freq = np.ascontiguousarray(freq, dtype=np.float64)
        pos_dist = np.ascontiguousarray(pos_dist, dtype=np.float64)
        S = len(pos_dist)
        post = np.empty(S) if want_post else None
        lk = self.lib.hp_forward_backward(_dp(gl_ind), _dp(freq), _dp(pos_dist), S, float(indF),
                                          float(alpha), _dp(post) if want_post else None)
        return lk, post

    def est_maf(self, gl_site, indF):
        gl_site = np.ascontiguousarray(gl_site, dtype=np.float64)
        indF = np.ascontiguousarray(indF, dtype=np.float64)
        n = C.c_int(0)
        f = self.lib.hp_est_maf(len(indF), _dp(gl_site), _dp(indF), C.byref(n))
        return f, n.value

    def haplo_freq(self, p1, p2, maf1, maf2):
        p1 = np.ascontiguousarray(p1, dtype=np.float64)
        p2 = np.ascontiguousarray(p2, dtype=np.float64)
        hap = np.zeros(4)
        it = self.lib.hp_haplo_freq(_dp(hap), _dp(p1), _dp(p2), float(maf1), float(maf2), len(p1))
        return hap, it

    def emission_ld(self, hap, gl_p, gl_c, maf_p, F):
        hap = np.ascontiguousarray(hap, dtype=np.float64)
        gl_p = np.ascontiguousarray(gl_p, dtype=np.float64)
        gl_c = np.ascontiguousarray(gl_c, dtype=np.float64)
        return self.lib.hp_emission_ld(_dp(hap), _dp(gl_p), _dp(gl_c), float(maf_p), int(F))

"""The indF / alpha M-step's L-BFGS-B machines ON THE DEVICE (kernels_bfgs.hip) against the
same machines on the host (bfgs_batch.cpp, the round-1..4 path, still there behind the switch
`no_dev_bfgs`): the reference's per-individual optimizer (EM.cpp:198-201,423-440;
shared/bfgs.cpp:22-138) has no barrier between individuals, so WHO advances the machines between
two objective rounds must not show in any result.  Same handle, same data, same starting
values; the objective kernels are the same either way, so every array must agree bit for bit
and the optimizer's accounting (rounds, points, the reference's forward-pass count) too."""
import numpy as np
import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _stats(st):
    return (st.rounds, st.points, st.ref_forward_calls, st.ind_rounds)


def _two_paths(h, fn):
    out = []
    for host in (1, 0):
        h.set_switch("no_dev_bfgs", host)
        out.append(fn(h))
    return out


CASES = [
    # I, S, chromosomes, missing, call_geno, seed
    (7, 900, 1, 0.0, False, 1),
    (100, 20_000, 3, 0.03, False, 2),
    (333, 6_000, 2, 0.1, False, 3),
    (1000, 4_000, 1, 0.0, False, 4),
    (64, 12_000, 4, 0.05, True, 5),     # called genotypes: probes at F's upper bound, own exponents
    (13, 257, 1, 0.0, False, 6),
]


@pytest.mark.parametrize("I,S,nchr,miss,call,seed", CASES)
def test_mstep_on_the_device_equals_the_host_machines(pkg, I, S, nchr, miss, call, seed):
    d = pkg.simulate.simulate(I, S, seed=seed, n_chrom=nchr, missing_rate=miss, indF="r", freq="r", alpha=0.4)
    rng = np.random.default_rng(seed)
    F0, A0 = rng.uniform(0.02, 0.9, I), 10 ** rng.uniform(-2, 0.7, I)
    f0 = rng.uniform(0.05, 0.5, S)
    mode = pkg.MODE_FAST | (pkg.GENO_PACKED if call else 0)
    with pkg.NgsFHMM(I, S, mode=mode) as h:
        h.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=call)
        for fixed in ((False, False), (True, False), (False, True)):
            def run(h):
                h.set_params(F0, A0, f0)
                h.init_emission()
                st = h.mstep_indf(*fixed)
                return h.indF.copy(), h.alpha.copy(), _stats(st)
            h.mode_counts(reset=True)
            (Fh, Ah, sh), (Fd, Ad, sd) = _two_paths(h, run)
            mixed_rounds = h.mode_counts().get("rounds_of_mixed_versions", 0)
            if (I, S) == (100, 20_000) and not fixed[1]:
                # small and general alpha side by side in a small cohort: the device's rounds took their
                # versions in ONE launch (k_fast_lkl_fd_mix), the host's one launch per version
                assert mixed_rounds >= 1, h.mode_counts()
            assert np.array_equal(Fh, Fd), (fixed, np.abs(Fh - Fd).max())
            assert np.array_equal(Ah, Ad), (fixed, np.abs(Ah - Ad).max())
            assert sh == sd, (fixed, sh, sd)
            assert sd[0] >= 2 and np.all(np.isfinite(Fd)) and np.all(np.isfinite(Ad))
            if fixed[0]:
                assert np.array_equal(Fd, F0)
            if fixed[1]:
                assert np.array_equal(Ad, A0)


@pytest.mark.parametrize("I,S,nchr,call", [(100, 30_000, 2, False), (1000, 5_000, 1, False), (96, 16_000, 4, True)])
def test_whole_iterations_on_the_device_equal_the_host_machines(pkg, I, S, nchr, call):
    """Six fused EM iterations (E-step sharing the first round's walk, frequency step behind the
    rounds) from test.sh's starting values: every array after every iteration."""
    d = pkg.simulate.simulate(I, S, seed=77, n_chrom=nchr, missing_rate=0.02, indF="r", freq="r", alpha=0.3)
    mode = pkg.MODE_FAST | (pkg.GENO_PACKED if call else 0)
    with pkg.NgsFHMM(I, S, mode=mode) as h:
        h.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=call)

        def run(h):
            h.set_params(0.1, 0.2, 0.1)
            h.init_emission()
            out = []
            for _ in range(6):
                st = h.iter_EM()
                out.append((h.ind_lkl.copy(), h.indF.copy(), h.alpha.copy(), h.freq.copy(), _stats(st)))
            out.append(h.marg_prob)
            return out
        host, dev = _two_paths(h, run)
        for it, (a, b) in enumerate(zip(host[:-1], dev[:-1])):
            for name, x, y in zip(("ind_lkl", "indF", "alpha", "freq"), a, b):
                assert np.array_equal(x, y), (it, name, np.abs(x - y).max())
            assert a[4] == b[4], (it, a[4], b[4])
        assert np.array_equal(host[-1], dev[-1])
        assert dev[0][4][0] > dev[-2][4][0]        # the first iteration needs more rounds than the sixth
        # the kernels of a fused iteration are timed on request only (switch `spans`: its events
        # are packets between the kernels): nothing by default, the same bits with it
        assert h.kernel_ms("bfgs")[0] == 0 and h.kernel_ms("est_maf")[0] == 0
        h.set_switch("spans", 1)
        timed = run(h)
        h.set_switch("spans", 0)
        for a, b in zip(dev[:-1], timed[:-1]):
            for x, y in zip(a[:4], b[:4]):
                assert np.array_equal(x, y)
        bf, n = h.kernel_ms("bfgs")
        assert bf > 0 and n >= 1                   # the device's planning kernels ran (slot 7)
        assert h.kernel_ms("est_maf")[0] > 0 and h.kernel_ms("lkl_batch")[0] > 0


def test_device_mstep_of_a_replica_and_stats_of_the_reference(pkg):
    """A replica (multi-start, nghmm_create_replica) has machines of its own on the device; and
    the count of forward passes the reference would have spent is reported as before."""
    I, S = 50, 8_000
    d = pkg.simulate.simulate(I, S, seed=9, n_chrom=2, indF="r", freq=0.2, alpha=0.1)
    gl = pkg.simulate.normalise_log_gl(d.gl)
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as h:
        h.load(gl, d.pos_dist_mb)
        with h.replica() as r:
            for x in (h, r):
                x.set_params(0.1, 0.2, 0.1)
                x.init_emission()
            sa, sb = h.iter_EM(), r.iter_EM()
            assert _stats(sa) == _stats(sb)
            assert np.array_equal(h.indF, r.indF) and np.array_equal(h.freq, r.freq)
            assert sa.ref_forward_calls > sa.points > 0


def test_an_mstep_that_ends_early_leaves_the_handle_usable(pkg):
    """An M-step that returns on its way (a launch failure, say; here the test switch
    dbg_abort_round after round 2) has plans published in pinned memory and counters left behind;
    the next M-step starts over (dbfgs_begin: counters cleared, plan numbers far ahead) and gives
    what a fresh handle gives, bit for bit."""
    I, S = 40, 3000
    d = pkg.simulate.simulate(I, S, seed=21, n_chrom=2, indF="r", freq=0.25, alpha=0.2)
    gl = pkg.simulate.normalise_log_gl(d.gl)
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as h, pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as fresh:
        for x in (h, fresh):
            x.load(gl, d.pos_dist_mb)
        h.set_params(0.1, 0.2, 0.2)
        h.init_emission()
        h.iter_EM()                                 # (a clean M-step first)
        h.set_switch("dbg_abort_round", 2)
        with pytest.raises(pkg.NgsFHMMError) as ei:
            h.iter_EM()
        assert "dbg_abort_round" in str(ei.value)
        h.set_switch("dbg_abort_round", 0)
        h.synchronize()
        for x in (h, fresh):
            x.set_params(0.1, 0.2, 0.2)
            x.init_emission()
        for _ in range(3):
            sa, sb = h.iter_EM(), fresh.iter_EM()
            assert _stats(sa) == _stats(sb)
            assert np.array_equal(h.indF, fresh.indF) and np.array_equal(h.alpha, fresh.alpha)
            assert np.array_equal(h.ind_lkl, fresh.ind_lkl) and np.array_equal(h.freq, fresh.freq)


def test_a_round_planned_in_advance_never_outlives_its_parameters(pkg):
    """When an M-step ends the device plans the NEXT M-step's first round at once (dbfgs_preplan:
    the parameters are final), and an iteration ends by a word the second stream's epilogue kernel
    writes.  Whatever happens between two iterations -- new indF / alpha (nghmm_set_params), only
    new frequencies, a change of the fixed parameters, the machines moved to the host and back, a
    standalone M-step, a reload of the data -- the next iteration must be the one a handle without
    that history computes: every array bit for bit against a fresh handle given the same state."""
    I, S = 120, 9_000
    d = pkg.simulate.simulate(I, S, seed=77, n_chrom=2, missing_rate=0.04, indF="r", freq="r", alpha=0.3)
    gl = pkg.simulate.normalise_log_gl(d.gl)
    rng = np.random.default_rng(5)

    def state(h):
        return (h.ind_lkl.copy(), h.indF.copy(), h.alpha.copy(), h.freq.copy(), h.marg_prob.copy())

    def fresh(F, A, f, fixed=(False, False), n=1):
        with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as g:
            g.load(gl, d.pos_dist_mb)
            g.set_params(F, A, f)
            g.init_emission()
            for _ in range(n):
                g.iter_EM(1, *fixed)
            return state(g)

    def same(a, b, what):
        for x, y in zip(a, b):
            assert np.array_equal(x, y), what

    with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as h:
        h.load(gl, d.pos_dist_mb)
        h.set_params(0.1, 0.2, 0.1)
        h.init_emission()
        h.iter_EM()
        h.iter_EM()
        same(state(h), fresh(0.1, 0.2, 0.1, n=2), "two plain iterations")
        # new indF / alpha under the plan made in advance
        F1, A1 = rng.uniform(0.05, 0.8, I), rng.uniform(0.05, 2.0, I)
        f_now = h.freq
        h.set_params(F1, A1, None)
        h.init_emission()
        h.iter_EM()
        same(state(h), fresh(F1, A1, f_now), "after new indF / alpha")
        # only new frequencies: the plan stays valid, the emissions do not
        F2, A2 = h.indF, h.alpha
        f2 = rng.uniform(0.05, 0.5, S)
        h.set_params(None, None, f2)
        h.init_emission()
        h.iter_EM()
        same(state(h), fresh(F2, A2, f2), "after new frequencies")
        # another choice of fixed parameters than the plan was made for
        F3, A3, f3 = h.indF, h.alpha, h.freq
        h.iter_EM(1, True, False)
        same(state(h), fresh(F3, A3, f3, fixed=(True, False)), "indF fixed")
        # the machines on the host for an iteration, then back on the device
        F4, A4, f4 = h.indF, h.alpha, h.freq
        h.set_switch("no_dev_bfgs", 1)
        h.iter_EM()
        h.set_switch("no_dev_bfgs", 0)
        h.iter_EM()
        same(state(h), fresh(F4, A4, f4, n=2), "host machines in between")
        # a standalone M-step (no E-step, no frequency step), then an iteration
        F5, A5, f5 = h.indF, h.alpha, h.freq
        h.mstep_indf()
        F6, A6 = h.indF, h.alpha
        h.iter_EM()
        same(state(h), fresh(F6, A6, f5), "after a standalone M-step")
        # the same data loaded again
        F7, A7, f7 = h.indF, h.alpha, h.freq
        h.load(gl, d.pos_dist_mb)
        h.set_params(F7, A7, f7)
        h.init_emission()
        h.iter_EM()
        same(state(h), fresh(F7, A7, f7), "after a reload")
        # and the counters of the rare code paths answer (include/nghmm_debug.h)
        mc, ec = h.mode_counts(), h.estmaf_counts()
        mc.pop("rounds_of_mixed_versions", None)
        assert sum(mc.values()) > 0 and all(k == "general" or k.startswith("2F2A") for k in mc), mc
        assert set(ec) == {"check_failed", "second_interval", "third_interval", "log_space", "exact_tail"}


def test_asynchronous_iteration_soak():
    """tools/stress_async.py, short: 90 fused iterations with everything asynchronous on (second
    stream, epilogue word, pre-planned rounds) against the same iterations on one stream with timing
    events and a stream synchronisation at the end, and four replicas driven concurrently against the
    same runs one after the other -- every iteration's arrays bit for bit, on three cohort shapes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_async.py"), "90", "4"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "stress ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]

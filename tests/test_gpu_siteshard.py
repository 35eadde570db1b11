"""Site shards (include/nghmm.h, "shard the SITES instead") on ONE GPU: V handles of one process
play the V ranks -- each holds all individuals for a contiguous site range and runs in a thread
of its own; the all-gather the library asks for is done by the test (barrier + device copies) --
and the chain must reproduce one handle that holds every site."""
import importlib
import threading

import numpy as np
import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


class Chain:
    """V site-shard handles on cuda:0 with a thread-barrier all-gather."""

    def __init__(self, pkg, gl, pos, V, mode=None, raw=None):
        import torch
        dd = importlib.import_module("ngsf-hmm_amd.distributed")
        self.torch = torch
        self.V = V
        S, I = gl.shape[0] if gl is not None else raw.shape[0], (gl if gl is not None else raw).shape[1]
        self.ranges = dd.site_ranges_ragged(S, V)
        self.barrier = threading.Barrier(V)
        dev = torch.device("cuda", 0)
        self.h, self.send, self.recv = [], [], []
        self.gathers = [0] * V
        for r, (lo, hi) in enumerate(self.ranges):
            h = pkg.NgsFHMM(I, hi - lo, mode=pkg.MODE_FAST if mode is None else mode)
            if raw is not None:
                h.load_raw(np.ascontiguousarray(raw[lo:hi]), np.ascontiguousarray(pos[lo:hi]),
                           space=0, call_geno=True)
            else:
                h.load(np.ascontiguousarray(gl[lo:hi]), np.ascontiguousarray(pos[lo:hi]))
            n = h.site_shard_bytes()
            self.send.append(torch.zeros(n // 8, dtype=torch.float64, device=dev))
            self.recv.append(torch.zeros(V * (n // 8), dtype=torch.float64, device=dev))
            self.h.append(h)
        torch.cuda.synchronize()
        for r, h in enumerate(self.h):
            h.site_shard_setup(r, V, self.send[r].data_ptr(), self.recv[r].data_ptr(),
                               h.site_shard_bytes(), self._gather(r))

    def _gather(self, r):
        def cb(n):
            k = n // 8
            self.h[r].synchronize()            # this handle's stream has written send[r]
            self.barrier.wait()
            for q in range(self.V):
                self.recv[r][q * k:(q + 1) * k].copy_(self.send[q][:k])
            self.torch.cuda.synchronize()
            self.barrier.wait()                # nobody rewrites its send before all have read it
            self.gathers[r] += 1
        return cb

    def each(self, fn):
        """fn(r, handle) on every handle, one thread each; returns the results in rank order."""
        out, err = [None] * self.V, []

        def run(r):
            try:
                out[r] = fn(r, self.h[r])
            except BaseException as e:
                err.append(e)
                self.barrier.abort()
        th = [threading.Thread(target=run, args=(r,)) for r in range(self.V)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if err:
            raise err[0]
        return out

    def set_params(self, F, A, freq):
        for (lo, hi), h in zip(self.ranges, self.h):
            h.set_params(F, A, np.broadcast_to(freq, (self.ranges[-1][1],))[lo:hi])
            h.init_emission()

    def viterbi(self):
        scores = None
        for h in self.h:
            scores = h.viterbi_shard_forward(scores)
        state, parts = None, [None] * self.V
        for r in reversed(range(self.V)):
            state, parts[r] = self.h[r].viterbi_shard_back(state)
        return np.concatenate(parts, axis=1)

    def close(self):
        for h in self.h:
            h.close()


@pytest.mark.parametrize("V", [2, 3])
def test_site_shards_reproduce_one_handle(pkg, orc_libm, V):
    I, S = 37, 5000
    d = pkg.simulate.simulate(I, S, seed=11 + V, n_chrom=3, missing_rate=0.05, indF="r")
    gl = pkg.simulate.normalise_log_gl(d.gl)
    rng = np.random.default_rng(5)
    F, A = rng.uniform(0.02, 0.6, I), rng.uniform(0.01, 2.0, I)
    freq = rng.uniform(0.05, 0.45, S)

    whole = pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST)
    whole.load(gl, d.pos_dist_mb)
    whole.set_params(F, A, freq)
    whole.init_emission()
    ch = Chain(pkg, gl, d.pos_dist_mb, V)
    ch.set_params(F, A, freq)

    # E-step: the chain's log-likelihoods on every handle, the posteriors of the own sites
    whole.estep()
    lk = ch.each(lambda r, h: (h.estep(), h.ind_lkl.copy())[1])
    for r in range(1, V):
        assert np.array_equal(lk[r], lk[0])                    # the same bits everywhere
    np.testing.assert_allclose(lk[0], whole.ind_lkl, rtol=1e-12)
    post = np.concatenate([h.marg_prob for h in ch.h], axis=1)
    np.testing.assert_allclose(post, whole.marg_prob, atol=1e-10)
    assert min(ch.gathers) == max(ch.gathers) == 1

    # objective values at arbitrary points (several per individual, some alone)
    ind = np.concatenate([np.repeat(np.arange(I), 3), [4, 4, 9]]).astype(np.uint32)
    pF = rng.uniform(0.001, 0.9, ind.size)
    pA = rng.uniform(0.001, 5.0, ind.size)
    want = whole.lkl(ind, pF, pA)
    got = ch.each(lambda r, h: h.lkl(ind, pF, pA))
    for r in range(1, V):
        assert np.array_equal(got[r], got[0])
    np.testing.assert_allclose(got[0], want, rtol=1e-12)
    em = __import__("orclib").OracleEM(orc_libm, gl, d.pos_dist_mb)
    em.set_params(F, A, freq)
    assert em.init_emission() == 0
    e = em.e_prob
    np.testing.assert_allclose(got[0], [-orc_libm.lkl([f, a], e[int(i)], d.pos_dist_mb)
                                        for i, f, a in zip(ind, pF, pA)], rtol=1e-11)

    # whole iterations, teacher-forced (the optimizer amplifies last-bit noise; a shard cuts the
    # site axis into other lane-chunks than the whole does)
    for it in range(3):
        whole.set_params(F, A, freq)
        whole.init_emission()
        ch.set_params(F, A, freq)
        whole.iter_EM(1)
        g0 = ch.gathers[0]
        st = ch.each(lambda r, h: h.iter_EM(1))
        # one all-gather per objective round (a chain never splits its rounds into two half
        # launches by itself: that rule looks at the handle's own size); the E-step's rides on the
        # first round's (point 0 of every individual is its current parameters); a re-evaluated
        # round adds one
        assert st[0].rounds <= ch.gathers[0] - g0 <= st[0].rounds + 2 and len(set(ch.gathers)) == 1
        pars = [(h.indF, h.alpha, h.ind_lkl.copy()) for h in ch.h]
        for r in range(1, V):                                  # every handle took the same steps
            assert np.array_equal(pars[r][0], pars[0][0]) and np.array_equal(pars[r][1], pars[0][1])
            assert np.array_equal(pars[r][2], pars[0][2])
            assert st[r].rounds == st[0].rounds and st[r].points == st[0].points
        np.testing.assert_allclose(pars[0][2], whole.ind_lkl, rtol=1e-12)
        np.testing.assert_allclose(pars[0][0], whole.indF, atol=2e-6)
        np.testing.assert_allclose(pars[0][1], whole.alpha, rtol=2e-4, atol=2e-6)
        f_chain = np.concatenate([h.freq for h in ch.h])
        np.testing.assert_allclose(f_chain, whole.freq, rtol=1e-9, atol=1e-12)
        post = np.concatenate([h.marg_prob for h in ch.h], axis=1)
        np.testing.assert_allclose(post, whole.marg_prob, atol=1e-9)
        F, A, freq = whole.indF, whole.alpha, whole.freq

    # fixed indF / alpha: nothing to amplify
    whole.set_params(F, A, freq)
    whole.init_emission()
    ch.set_params(F, A, freq)
    for it in range(2):
        whole.iter_EM(1, True, True)
        ch.each(lambda r, h: h.iter_EM(1, True, True))
    np.testing.assert_allclose(np.concatenate([h.freq for h in ch.h]), whole.freq, rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(ch.h[0].ind_lkl, whole.ind_lkl, rtol=1e-12)

    # decoding over the chain: the same path, cell for cell
    whole.set_params(F, A, whole.freq)
    ch.set_params(F, A, whole.freq)
    assert np.array_equal(ch.viterbi(), whole.viterbi())
    ch.close()
    whole.close()


def test_site_shards_of_called_genotypes(pkg):
    """Packed handles (2-bit codes) as site shards: est_maf on the codes, the forward walk on
    the class table."""
    I, S, V = 150, 4000, 2
    d = pkg.simulate.simulate(I, S, seed=3, n_chrom=2, missing_rate=0.05, indF="r")
    whole = pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST | pkg.GENO_PACKED)
    whole.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=True)
    whole.set_params(0.1, 0.2, 0.1)
    whole.init_emission()
    ch = Chain(pkg, None, d.pos_dist_mb, V, mode=pkg.MODE_FAST | pkg.GENO_PACKED, raw=d.gl)
    ch.set_params(0.1, 0.2, 0.1)
    for it in range(2):
        whole.iter_EM(1, True, True)
        ch.each(lambda r, h: h.iter_EM(1, True, True))
    np.testing.assert_allclose(ch.h[1].ind_lkl, whole.ind_lkl, rtol=1e-12)
    np.testing.assert_allclose(np.concatenate([h.freq for h in ch.h]), whole.freq, rtol=1e-11, atol=1e-14)
    assert np.array_equal(ch.viterbi(), whole.viterbi())
    ch.close()
    whole.close()


def test_site_shards_are_a_fast_mode_layout(pkg):
    import torch
    h = pkg.NgsFHMM(4, 64, mode=pkg.MODE_EXACT)
    buf = torch.zeros(4096, dtype=torch.float64, device="cuda")
    with pytest.raises(pkg.NgsFHMMError, match="fast-mode layout"):
        h.site_shard_setup(0, 2, buf.data_ptr(), buf.data_ptr(), 8 * 1024, lambda n: None)
    h.close()
    # a buffer smaller than nghmm_site_shard_bytes() is refused too
    h = pkg.NgsFHMM(4, 64, mode=pkg.MODE_FAST)
    with pytest.raises(pkg.NgsFHMMError, match="nghmm_site_shard_bytes"):
        h.site_shard_setup(0, 2, buf.data_ptr(), buf.data_ptr(), 64, lambda n: None)
    h.close()


def test_a_failing_exchange_fails_the_call(pkg):
    """An exception inside the all-gather callback must surface from the library call that
    needed the exchange, not vanish in the C frame."""
    import torch
    I, S = 6, 800
    d = pkg.simulate.simulate(I, S, seed=2)
    h = pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST)
    h.load(pkg.simulate.normalise_log_gl(d.gl), d.pos_dist_mb)
    n = h.site_shard_bytes()
    send = torch.zeros(n // 8, dtype=torch.float64, device="cuda")
    recv = torch.zeros(2 * (n // 8), dtype=torch.float64, device="cuda")

    def boom(nbytes):
        raise RuntimeError("link down")
    h.site_shard_setup(0, 2, send.data_ptr(), recv.data_ptr(), n, boom)
    # one layout at a time: a site shard is not an individual shard, has no replicas, joins no group
    assert h.lib.nghmm_shard_config(h.handle, 2 * I, 0, 0, S // 2) != 0
    with pytest.raises(pkg.NgsFHMMError):
        h.replica()
    h.set_params(0.1, 0.2, 0.1)
    h.init_emission()
    with pytest.raises(RuntimeError, match="link down"):
        h.estep()
    h.close()


@pytest.mark.parametrize("packed", [False, True])
def test_in_process_chain_equals_one_handle(pkg, packed):
    """nghmm_chain_setup / _iter_em / _mstep_freq / _viterbi (what the C++ host's --n_gpus uses):
    four handles on one GPU, the library's own barrier + device-copy all-gather."""
    I, S, V = 60, 6400, 4
    d = pkg.simulate.simulate(I, S, seed=21, n_chrom=2, missing_rate=0.05, indF="r")
    gl = pkg.simulate.normalise_log_gl(d.gl)
    dd = importlib.import_module("ngsf-hmm_amd.distributed")
    mode = pkg.MODE_FAST | (pkg.GENO_PACKED if packed else 0)

    def make(lo, hi):
        h = pkg.NgsFHMM(I, hi - lo, mode=mode)
        if packed:
            h.load_raw(np.ascontiguousarray(d.gl[lo:hi]), np.ascontiguousarray(d.pos_dist_mb[lo:hi]),
                       space=0, call_geno=True)
        else:
            h.load(np.ascontiguousarray(gl[lo:hi]), np.ascontiguousarray(d.pos_dist_mb[lo:hi]))
        return h
    ranges = dd.site_ranges_ragged(S, V)
    # a chromosome that starts exactly at a cut (the handle's first distance is +inf), one that
    # starts on the last site of a range, one in the middle of a range
    d.pos_dist_mb[ranges[1][0]] = np.inf
    d.pos_dist_mb[ranges[2][1] - 1] = np.inf
    whole = make(0, S)
    hs = [make(lo, hi) for lo, hi in ranges]
    ch = pkg.Chain(hs)
    # --freq e: the frequency step before any E-step, every handle on its own sites
    for h in [whole] + hs:
        h.set_params(0.1, 0.2, 0.1)
    whole.mstep_freq(1)
    ch.mstep_freq(1)
    np.testing.assert_allclose(ch.freq, whole.freq, rtol=1e-12)
    for h in [whole] + hs:
        h.init_emission()
    for it in range(3):
        whole.iter_EM(1, True, True)
        st = ch.iter_EM(1, True, True)
    np.testing.assert_allclose(ch.ind_lkl, whole.ind_lkl, rtol=1e-12)
    np.testing.assert_allclose(ch.freq, whole.freq, rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(ch.marg_prob, whole.marg_prob, atol=1e-9)
    # free parameters: every handle takes the same steps
    st = ch.iter_EM(1)
    whole.iter_EM(1)
    assert st.rounds >= 2
    for h in hs[1:]:
        assert np.array_equal(h.indF, hs[0].indF) and np.array_equal(h.alpha, hs[0].alpha)
    np.testing.assert_allclose(hs[0].indF, whole.indF, atol=5e-6)
    whole.set_params(hs[0].indF, hs[0].alpha, ch.freq)
    assert np.array_equal(ch.viterbi(), whole.viterbi())
    # a closed member dissolves the chain
    hs[2].close()
    with pytest.raises(pkg.NgsFHMMError):
        ch.iter_EM(1)
    for h in hs[:2] + hs[3:]:
        h.close()
    whole.close()


@pytest.mark.parametrize("I,S,V", [(1, 97, 2), (3, 640, 4), (9, 2049, 3), (130, 333, 2), (64, 64, 2),
                                   (17, 5000, 5)])
def test_chain_shapes(pkg, I, S, V):
    """Odd shapes: one individual, ranges of a few sites (fewer than a wave has lanes), more
    handles than chromosomes, a cohort that takes the four-sites-per-wave est_maf."""
    d = pkg.simulate.simulate(I, S, seed=I + S, n_chrom=2, missing_rate=0.1, indF="r")
    gl = pkg.simulate.normalise_log_gl(d.gl)
    dd = importlib.import_module("ngsf-hmm_amd.distributed")
    ranges = dd.site_ranges_ragged(S, V)

    def make(lo, hi):
        h = pkg.NgsFHMM(I, hi - lo, mode=pkg.MODE_FAST)
        h.load(np.ascontiguousarray(gl[lo:hi]), np.ascontiguousarray(d.pos_dist_mb[lo:hi]))
        h.set_params(0.2, 0.3, 0.15)
        h.init_emission()
        return h
    whole = make(0, S)
    hs = [make(lo, hi) for lo, hi in ranges]
    ch = pkg.Chain(hs)
    for it in range(2):
        whole.iter_EM(1, True, True)
        ch.iter_EM(1, True, True)
    np.testing.assert_allclose(ch.ind_lkl, whole.ind_lkl, rtol=1e-12)
    np.testing.assert_allclose(ch.freq, whole.freq, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(ch.marg_prob, whole.marg_prob, atol=1e-9)
    ch.iter_EM(1)
    for h in hs[1:]:
        assert np.array_equal(h.indF, hs[0].indF) and np.array_equal(h.alpha, hs[0].alpha)
    assert np.isfinite(ch.ind_lkl).all()
    whole.set_params(hs[0].indF, hs[0].alpha, ch.freq)
    assert np.array_equal(ch.viterbi(), whole.viterbi())
    for h in hs + [whole]:
        h.close()


def test_chain_of_eight_packed_shards_hundred_iterations(pkg):
    """BASELINE configs[4]'s recipe (--call_geno, 25 chromosomes, 100 iterations, 8 GPUs) on a
    chain of eight packed site shards at 625 individuals x 320 000 sites: every handle takes the
    same steps in every one of the 100 iterations, everything stays finite and in range, the
    total log-likelihood settles."""
    import torch
    dd = importlib.import_module("ngsf-hmm_amd.distributed")
    I, S, V = 625, 320_000, 8
    dev = torch.device("cuda", 0)
    mode = pkg.MODE_FAST | pkg.GENO_PACKED
    ranges = dd.site_ranges_ragged(S, V)
    pos, chunks = pkg.simulate.simulate_torch_chunks(I, S, dev, seed=17, n_chrom=25, chunk_sites=40_000)
    chunks = list(chunks)                      # 8 blocks of 40 000 sites = the ranges
    assert [lo for lo, _ in chunks] == [lo for lo, _ in ranges]
    hs = []
    for (lo, hi), (_, c) in zip(ranges, chunks):
        h = pkg.NgsFHMM(I, hi - lo, mode=mode)
        torch.cuda.synchronize()
        h.load_chunks_device(pos[lo:hi].contiguous().data_ptr(), [(0, hi - lo, c.data_ptr())],
                             space=0, call_geno=True)
        h.set_params(0.1, 0.2, 0.1)
        h.init_emission()
        hs.append(h)
    del chunks
    ch = pkg.Chain(hs)
    tot = []
    for it in range(100):
        st = ch.iter_EM(1)
        tot.append(float(ch.ind_lkl.sum()))
        assert np.isfinite(ch.ind_lkl).all() and st.rounds >= 1
        if it % 25 == 24:
            for h in hs[1:]:
                assert np.array_equal(h.indF, hs[0].indF) and np.array_equal(h.alpha, hs[0].alpha)
    F, A, f = hs[0].indF, hs[0].alpha, ch.freq
    assert (F >= 0).all() and (F <= 1).all() and (A > 0).all() and (A <= 10).all()
    assert np.isfinite(f).all() and (f >= 0).all() and (f <= 1).all()
    tot = np.array(tot)
    assert np.diff(tot)[10:].min() > -1e-6 * abs(tot[-1])            # no step backwards
    assert abs(tot[-1] - tot[-2]) < 1e-8 * abs(tot[-1])              # settled
    path = ch.viterbi()
    assert path.shape == (I, S) and set(np.unique(path)) <= {0, 1}
    for h in hs:
        h.close()

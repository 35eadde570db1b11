"""Called genotypes as 2-bit codes (NGHMM_GENO_PACKED): --call_geno
and called-genotype input make every cell one of four (ngsF-HMM.cpp:101-117,
shared/read_data.cpp:88-98, shared/gen_func.cpp:886-914), so a packed handle keeps 0.25 B
instead of 24 B per site and individual.  It must give what an unpacked handle gives on the
same cells: bit for bit in exact mode (which is itself bit-identical to the oracle), within
1e-9 per call in fast mode.  Also here: the chunked loaders (a block of sites at a time, any
order) against the whole-matrix loaders, and BASELINE.json configs[4]'s shape (25
chromosomes, --call_geno) at the largest size one GPU holds, through properties."""
import importlib

import numpy as np
import pytest

import cli_util
import orclib
from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

I, S, NCHR = 40, 2500, 25


@pytest.fixture(scope="module")
def called(pkg):
    d = pkg.simulate.simulate(I, S, seed=99, n_chrom=NCHR, indF="r", freq="r", alpha=0.4)
    return d


def _pair(pkg, d, mode, how):
    """(unpacked, packed) handles loaded with the same called genotypes.
    how = 'call_geno': GL input + --call_geno; 'tg': called-genotype input."""
    a = pkg.NgsFHMM(I, S, mode=mode)
    b = pkg.NgsFHMM(I, S, mode=mode | pkg.GENO_PACKED)
    if how == "call_geno":
        a.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=True)
        b.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=True)
    else:
        # true genotypes of sites that are all polymorphic (a monomorphic site drives est_maf
        # to exactly 0 and the reference, like the oracle, then dies with "invalid Lkl found!")
        d = pkg.simulate.simulate(I, S, seed=98, n_chrom=NCHR, indF="r", freq=0.35, alpha=0.4)
        geno = d.geno.astype(np.int8).copy()
        geno[::7, ::3] = -1                                   # some missing genotypes
        a.load_chunks(d.pos_dist_mb, [(0, geno)])             # expanded to dense likelihoods
        b.load_chunks(d.pos_dist_mb, [(0, geno)])             # straight to codes
    for h in (a, b):
        h.set_params(0.1, 0.2, 0.1)
        h.init_emission()
    return a, b, d


@pytest.mark.parametrize("how", ["call_geno", "tg"])
def test_packed_equals_unpacked_bitwise_in_exact_mode(pkg, orc_det, called, how):
    d = called
    a, b, d = _pair(pkg, d, pkg.MODE_EXACT, how)
    assert np.array_equal(a.gl, b.gl)                         # the prepared likelihoods
    if how == "call_geno":                                    # ... which are the oracle's
        assert np.array_equal(b.gl, orc_det.prepare_gl(d.gl, 0, call_geno=True))
    assert np.array_equal(a.e_prob, b.e_prob)
    for it in range(3):
        a.iter_EM()
        b.iter_EM()
        assert np.array_equal(a.ind_lkl, b.ind_lkl)
        assert np.array_equal(a.marg_prob, b.marg_prob)
        assert np.array_equal(a.indF, b.indF) and np.array_equal(a.alpha, b.alpha)
        assert np.array_equal(a.freq, b.freq)
    assert np.array_equal(a.viterbi(), b.viterbi())
    assert np.array_equal(a.geno_posteriors(), b.geno_posteriors())
    a.close()
    b.close()


@pytest.mark.parametrize("how", ["call_geno", "tg"])
def test_packed_fast_mode_per_call(pkg, orc_libm, called, how):
    """Fast mode, packed against unpacked and against the oracle, call by call (1e-9, the
    tolerance BASELINE.json states; measured ~1e-13), then whole iterations: the two handles
    run the same arithmetic except for the emission of a cell, which the packed walk picks
    from four per-site values instead of three multiply-adds."""
    d = called
    a, b, d = _pair(pkg, d, pkg.MODE_FAST, how)
    gl = a.gl
    em = orclib.OracleEM(orc_libm, gl, d.pos_dist_mb)
    em.set_params(0.1, 0.2, 0.1)
    assert em.init_emission() == 0 and em.estep() == 0
    np.testing.assert_array_equal(b.gl, gl)
    np.testing.assert_allclose(b.e_prob, a.e_prob, rtol=1e-12, atol=0)   # log emissions (-inf = -inf)
    la, lb = a.estep().copy(), b.estep().copy()
    np.testing.assert_allclose(lb, la, rtol=1e-12)
    np.testing.assert_allclose(lb, em.ind_lkl, rtol=1e-9)
    np.testing.assert_allclose(b.marg_prob, a.marg_prob, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(b.marg_prob, em.marg, rtol=1e-9, atol=1e-12)
    rng = np.random.default_rng(1)
    ind = rng.integers(0, I, 200).astype(np.uint32)
    F, A = rng.uniform(0.01, 0.99, 200), rng.uniform(1e-3, 5, 200)
    np.testing.assert_allclose(b.lkl(ind, F, A), a.lkl(ind, F, A), rtol=1e-12)
    a.mstep_freq(1)
    b.mstep_freq(1)
    assert em.mstep_freq(1) == 0
    np.testing.assert_allclose(b.freq, a.freq, rtol=1e-12)
    np.testing.assert_allclose(b.freq, em.freq, rtol=1e-9)
    # whole iterations through the fused walk (the packed fresh walk reads the codes)
    for h in (a, b):
        h.set_params(0.1, 0.2, 0.1)
        h.init_emission()
    for it in range(3):
        a.iter_EM()
        b.iter_EM()
        np.testing.assert_allclose(b.ind_lkl, a.ind_lkl, rtol=1e-9)
    np.testing.assert_allclose(b.freq, a.freq, rtol=1e-5, atol=1e-7)     # after the optimizer
    np.testing.assert_allclose(b.indF, a.indF, atol=2e-4)
    pa, pb = a.viterbi(), b.viterbi()
    assert (pa != pb).mean() < 1e-3
    b.set_params(a.indF, a.alpha, a.freq)                                # identical parameters:
    b.init_emission()
    assert np.array_equal(b.viterbi(), pa)                               # ... identical paths
    assert np.array_equal(b.geno_posteriors(), a.geno_posteriors())
    a.close()
    b.close()


def test_packed_probes_at_the_upper_bound_of_F(pkg, orc_libm):
    """L-BFGS-B's first trial step of an M-step lands on a bound: F = 1 - 1e-15 with a one-sided
    probe at F - 2 eh.  On called genotypes a forced non-IBD site multiplies that probe's
    operator by (1 - F_p) / (1 - F_0) ~ 5e10 relative to point 0's, which a common exponent for
    all points cannot hold over a lane-chunk: such groups run the finite-difference kernel with
    an exponent PER POINT (FD_OWNEX; the general kernel before).  The groups against the same
    points sent one at a time (one point per group:
    the general kernel) and against the oracle."""
    I2, S2 = 24, 64 * 1200
    d = pkg.simulate.simulate(I2, S2, seed=5, n_chrom=4, indF=0.6, freq="r", alpha=0.4)
    h = pkg.NgsFHMM(I2, S2, mode=pkg.MODE_FAST | pkg.GENO_PACKED)
    h.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=True)
    h.set_params(0.5, 0.2, 0.2)
    h.init_emission()
    T = h.layout()[1]
    ub = 1 - 1e-15
    eh_F = (1e-8 * (ub + 1)) ** 0.67
    assert np.log(2 * eh_F / 1e-15) * T > 600          # beyond what one exponent holds
    al = 0.7
    eh_A = (1e-8 * (al + 1)) ** 0.67
    # f0, backward F probe (bfgs.cpp:38-40), the two alpha probes
    F5 = np.tile([ub, ub - 2 * eh_F, ub, ub], I2)
    A5 = np.tile([al, al, al + eh_A, al - eh_A], I2)
    ind5 = np.repeat(np.arange(I2), 4).astype(np.uint32)
    h.mode_counts(reset=True)
    got = h.lkl(ind5, F5, A5)
    assert any(k.endswith("e") for k in h.mode_counts()), h.mode_counts()   # an exponent per point (FD_OWNEX)
    assert np.isfinite(got).all()
    one_by_one = np.array([h.lkl(ind5[k:k + 1], F5[k:k + 1], A5[k:k + 1])[0] for k in range(len(ind5))])
    np.testing.assert_allclose(got, one_by_one, rtol=1e-12)
    em = orclib.OracleEM(orc_libm, h.gl, d.pos_dist_mb)
    em.set_params(0.5, 0.2, 0.2)
    assert em.init_emission() == 0
    e = em.e_prob
    want = np.array([-orc_libm.lkl([F5[k], A5[k]], e[ind5[k]], d.pos_dist_mb) for k in range(len(ind5))])
    np.testing.assert_allclose(got, want, rtol=1e-11)
    # the finite difference the optimizer forms from them
    g_got = (got[0::4] - got[1::4]) / (2 * eh_F)
    g_want = (want[0::4] - want[1::4]) / (2 * eh_F)
    np.testing.assert_allclose(g_got, g_want, rtol=1e-4, atol=1e-2)
    h.close()


def test_packed_handle_refuses_likelihoods(pkg, called):
    d = called
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST | pkg.GENO_PACKED) as h:
        with pytest.raises(pkg.NgsFHMMError) as e:
            h.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=False)
        assert "not a called genotype" in str(e.value)
        bad = d.geno.astype(np.int8).copy()
        bad[3, 4] = 3
        with pytest.raises(pkg.NgsFHMMError) as e:
            h.load_chunks(d.pos_dist_mb, [(0, bad)])
        assert "{-1,0,1,2}" in str(e.value)
        # and a handle survives a refused load
        h.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=True)
        h.set_params(0.1, 0.2, 0.1)
        h.init_emission()
        assert np.isfinite(h.estep()).all()


@pytest.mark.parametrize("packed", [False, True])
def test_chunked_loading_equals_whole_matrix(pkg, called, packed):
    """nghmm_load_begin / _sites / _end with blocks of sites in shuffled order, raw
    likelihoods and reader genotypes, host and device sources."""
    import torch
    d = called
    flag = pkg.GENO_PACKED if packed else 0
    whole = pkg.NgsFHMM(I, S, mode=pkg.MODE_EXACT | flag)
    whole.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=True)
    want = whole.gl
    whole.close()
    bounds = [0, 1, 17, 640, 641, 1500, 2499, S]
    blocks = [(lo, hi) for lo, hi in zip(bounds[:-1], bounds[1:])]
    order = np.random.default_rng(3).permutation(len(blocks))
    h = pkg.NgsFHMM(I, S, mode=pkg.MODE_EXACT | flag)
    h.load_chunks(d.pos_dist_mb, [(blocks[k][0], d.gl[blocks[k][0]:blocks[k][1]]) for k in order],
                  space=0, call_geno=True)
    assert np.array_equal(h.gl, want)
    # device source: the caller's buffer must come back untouched
    dev = torch.device("cuda", 0)
    t = torch.from_numpy(d.gl).to(dev)
    keep = t.clone()
    pos = torch.from_numpy(d.pos_dist_mb).to(dev)
    torch.cuda.synchronize()
    h.load_chunks_device(pos.data_ptr(),
                         [(lo, hi - lo, t[lo:hi].data_ptr()) for lo, hi in blocks], space=0,
                         call_geno=True)
    assert np.array_equal(h.gl, want) and torch.equal(t, keep)
    h.close()
    if not packed:                                  # reader genotypes into a dense handle
        geno = d.geno.astype(np.int8)
        a = pkg.NgsFHMM(I, S, mode=pkg.MODE_EXACT)
        a.load_raw(cli_util.raw_called_genotypes(geno), d.pos_dist_mb, space=0)
        b = pkg.NgsFHMM(I, S, mode=pkg.MODE_EXACT)
        b.load_chunks(d.pos_dist_mb, [(blocks[k][0], geno[blocks[k][0]:blocks[k][1]]) for k in order])
        assert np.array_equal(a.gl, b.gl)
        a.close()
        b.close()


def test_packed_two_shards_equal_one_handle(pkg, called):
    """The multi-GPU entry points with packed handles on one GPU: individuals split in two,
    the site shards built from exchanged code bytes (nghmm_get_geno_codes_dev ->
    nghmm_load_geno_site_shard_dev), posteriors and frequencies moved as the all-to-all /
    all-gather would."""
    import ctypes as C
    import torch
    dd = importlib.import_module("ngsf-hmm_amd.distributed")
    d = called
    world, I_loc = 2, I // 2
    dev = torch.device("cuda", 0)
    mode = pkg.MODE_FAST | pkg.GENO_PACKED
    whole = pkg.NgsFHMM(I, S, mode=mode)
    whole.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=True)
    whole.set_params(0.1, 0.2, 0.1)
    whole.init_emission()
    ranges = dd.site_ranges(S, world)
    S_own = S // world
    ranks = []
    for r in range(world):
        be = dd.GpuBackend(pkg, I_loc, S, 0, mode)
        be.hmm.load_raw(np.ascontiguousarray(d.gl[:, r * I_loc:(r + 1) * I_loc]), d.pos_dist_mb,
                        space=0, call_geno=True)
        be.hmm.set_params(0.1, 0.2, 0.1)
        be.hmm.init_emission()
        be.shard_config(I, r * I_loc, ranges[r][0], S_own)
        ranks.append(be)
    # one-off exchange of the codes: every rank's [S][I_loc] bytes -> [S_own][I] per rank
    codes = []
    for be in ranks:
        t = torch.empty((S, I_loc), device=dev, dtype=torch.uint8)
        be.hmm._check(be.hmm.lib.nghmm_get_geno_codes_dev(be.hmm.handle, 0, S,
                                                         C.c_void_p(t.data_ptr())))
        codes.append(t)
    for r, be in enumerate(ranks):
        lo, hi = ranges[r]
        shard = torch.cat([c[lo:hi] for c in codes], dim=1).contiguous()
        torch.cuda.synchronize()
        be.hmm._check(be.hmm.lib.nghmm_load_geno_site_shard_dev(be.hmm.handle,
                                                               C.c_void_p(shard.data_ptr())))
    for it in range(2):
        whole.estep(); whole.mstep_indf(); whole.mstep_freq(1)
        send = []
        for be in ranks:
            be.estep(); be.mstep_indf(False, False)
            buf = be.empty(world, S_own, I_loc)
            be.pack_posteriors(0, S, buf)
            send.append(buf)
        torch.cuda.synchronize()
        freq_all = torch.empty(S, device=dev, dtype=torch.float64)
        for r, be in enumerate(ranks):
            recv = torch.stack([send[q][r] for q in range(world)]).contiguous()
            own = be.empty(S_own)
            torch.cuda.synchronize()
            be.mstep_freq_sites(recv, own)
            freq_all[ranges[r][0]:ranges[r][1]] = own
        torch.cuda.synchronize()
        for be in ranks:
            be.set_freq(freq_all)
        np.testing.assert_allclose(ranks[0].hmm.freq, whole.freq, rtol=1e-12)
        for r, be in enumerate(ranks):
            sl = slice(r * I_loc, (r + 1) * I_loc)
            np.testing.assert_allclose(be.hmm.ind_lkl, whole.ind_lkl[sl], rtol=1e-12)
    for be in ranks:
        be.hmm.close()
    whole.close()


def test_config5_shape_at_one_gpu_share(pkg):
    """BASELINE.json configs[4] is 5000 individuals x 5 M sites, 25 chromosomes, --call_geno,
    on 8 GPUs: 625 individuals x 5 M sites per GPU.  This runs exactly one GPU's share --
    625 x 5,000,000 = 3.1e9 cells, generated on the device a block of sites at a time and
    packed on the way in (the dense matrix would be 75 GB) -- through two EM iterations and
    checks what does not need an oracle at that size: finite, increasing log-likelihoods,
    frequencies in (0, 1), posteriors in [0, 1], additivity of the log-likelihood over the
    25 chromosomes against 25 separate small handles for a few individuals, and the device
    memory the handle takes (the DESIGN.md section 3 table: <= 22 B per cell packed)."""
    import torch
    Ib, Sb, nchr = 625, 5_000_000, 25
    dev = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info(dev)[0]
    h = pkg.NgsFHMM(Ib, Sb, mode=pkg.MODE_FAST | pkg.GENO_PACKED)
    pos, chunks = pkg.simulate.simulate_torch_chunks(Ib, Sb, dev, seed=5, n_chrom=nchr,
                                                     chunk_sites=50_000)
    keep = {}                                     # chromosomes 0 and 7 of 3 individuals, dense
    per = Sb // nchr

    def feed():
        for s0, c in chunks:
            for ch in (0, 7):
                lo, hi = max(s0, ch * per), min(s0 + c.shape[0], (ch + 1) * per)
                if lo < hi:
                    keep.setdefault(ch, []).append(c[lo - s0:hi - s0, :3].clone())
            torch.cuda.synchronize()              # the library reads on its own stream
            yield s0, c.shape[0], c.data_ptr()
    h.load_chunks_device(pos.data_ptr(), feed(), space=0, call_geno=True)
    h.set_params(0.1, 0.2, 0.1)
    h.init_emission()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    used = free0 - torch.cuda.mem_get_info(dev)[0]
    per_cell = used / (Ib * Sb)
    print(f"packed handle: {used / 2**30:.1f} GiB = {per_cell:.2f} B per site and individual")
    assert per_cell <= 22.0
    lk0 = h.estep().copy()
    assert np.isfinite(lk0).all()
    # additivity over chromosomes: the E-step's per-individual log-likelihood is the sum of
    # the chromosomes' (infinite distance = the chain forgets); check two chromosomes of three
    # individuals against small dense handles with the same parameters
    for ch in (0, 7):
        sub = torch.cat(keep[ch]).contiguous()
        p = pos[ch * per:(ch + 1) * per].clone()
        p[0] = float("inf")
        with pkg.NgsFHMM(3, per, mode=pkg.MODE_FAST) as small:
            small.load_chunks_device(p.data_ptr(), [(0, per, sub.data_ptr())], space=0,
                                     call_geno=True)
            small.set_params(0.1, 0.2, 0.1)
            small.init_emission()
            keep[ch] = small.estep().copy()
    # (the full check needs all 25; two chromosomes bound the rest: lk0 <= their sum, both < 0)
    assert (lk0[:3] < keep[0] + keep[7]).all() and (keep[0] < 0).all()
    st = h.iter_EM()
    lk1 = h.ind_lkl.copy()
    h.set_switch("spans", 1)                                         # (kernel times below)
    h.iter_EM()
    lk2 = h.ind_lkl.copy()
    assert np.isfinite(lk2).all() and lk2.sum() > lk1.sum()          # EM ascends
    f = h.freq
    assert (f > 0).all() and (f < 1).all() and st.rounds > 0
    ms = {k: h.kernel_ms(k)[0] for k in ("lkl_batch", "forward", "est_maf")}
    print("config 5 share, one EM iteration kernels (ms):", ms)
    h.close()


@pytest.mark.parametrize("shape", [(1, 1), (1, 70), (3, 17), (5, 33), (17, 1000), (70, 515)])
def test_packed_ragged_shapes(pkg, orc_det, orc_libm, shape):
    """Sizes that are not multiples of anything (16 cells per code word, 16 sites per
    interleaved word, 64 lanes, 8-site blocks): a packed handle against the oracle, exact mode
    bit for bit and fast mode per call, one whole iteration + decoding."""
    Ir, Sr = shape
    d = pkg.simulate.simulate(Ir, Sr, seed=Ir * 1000 + Sr, missing_rate=0.1,
                              n_chrom=2 if Sr > 10 else 1, freq=0.3)
    gl = orc_det.prepare_gl(d.gl, 0, call_geno=True)
    for mode, orc in ((pkg.MODE_EXACT, orc_det), (pkg.MODE_FAST, orc_libm)):
        em = orclib.OracleEM(orc, gl, d.pos_dist_mb)
        em.set_params(0.2, 0.3, 0.25)
        assert em.init_emission() == 0 and em.estep() == 0
        with pkg.NgsFHMM(Ir, Sr, mode=mode | pkg.GENO_PACKED) as h:
            h.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=True)
            assert np.array_equal(h.gl, gl)
            h.set_params(0.2, 0.3, 0.25)
            h.init_emission()
            lk = h.estep().copy()
            if mode == pkg.MODE_EXACT:
                assert np.array_equal(lk, em.ind_lkl) and np.array_equal(h.marg_prob, em.marg)
                assert em.mstep_indf() == 0 and em.mstep_freq(1) == 0
                h.mstep_indf()
                h.mstep_freq(1)
                assert np.array_equal(h.indF, em.indF) and np.array_equal(h.freq, em.freq)
                assert np.array_equal(h.viterbi(), em.viterbi())
            else:
                np.testing.assert_allclose(lk, em.ind_lkl, rtol=1e-11)
                np.testing.assert_allclose(h.marg_prob, em.marg, rtol=1e-9, atol=1e-12)
                assert em.mstep_freq(1) == 0
                h.mstep_freq(1)
                np.testing.assert_allclose(h.freq, em.freq, rtol=1e-9)
                st = h.mstep_indf()
                assert st.rounds >= 1 and np.isfinite(h.indF).all()
                h.iter_EM()                                  # the fused walk on the codes
                assert np.isfinite(h.ind_lkl).all()


@pytest.mark.parametrize("packed", [False, True])
def test_chunked_loading_checks_every_site_exactly_once(pkg, packed):
    """nghmm_load_begin / _sites / _end: a chunk that overlaps sites already received is
    refused (a packed handle would OR two codes into one cell), nghmm_load_end before all
    sites have arrived is refused, and a clean load after either works."""
    I, S = 12, 300
    d = pkg.simulate.simulate(I, S, seed=3, missing_rate=0.1)
    mode = pkg.MODE_EXACT | (pkg.GENO_PACKED if packed else 0)
    ref = pkg.NgsFHMM(I, S, mode=mode)
    ref.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=packed)
    want = ref.gl
    ref.close()
    h = pkg.NgsFHMM(I, S, mode=mode)
    cut = lambda lo, hi: (lo, d.gl[lo:hi])
    with pytest.raises(pkg.NgsFHMMError) as ei:               # [90, 120) twice
        h.load_chunks(d.pos_dist_mb, [cut(0, 100), cut(100, 120), cut(90, 200), cut(200, S)],
                      space=0, call_geno=packed)
    assert ei.value.code == -10 and "overlap" in str(ei.value)
    with pytest.raises(pkg.NgsFHMMError) as ei:               # [100, 150) never arrives
        h.load_chunks(d.pos_dist_mb, [cut(150, S), cut(0, 100)], space=0, call_geno=packed)
    assert ei.value.code == -10 and "250 of 300 sites" in str(ei.value)
    with pytest.raises(pkg.NgsFHMMError):                     # the failed load is over
        h._check(h.lib.nghmm_load_end(h._h))
    h.load_chunks(d.pos_dist_mb, [cut(150, S), cut(100, 150), cut(0, 100)], space=0,
                  call_geno=packed)                            # any order, every site once
    assert np.array_equal(h.gl, want)
    h.close()


def test_switches_are_per_handle(pkg):
    """Measurement switches are read from the environment once, when a handle is created, and
    changed on a live handle by nghmm_set_switch only (include/nghmm.h)."""
    import os
    h = pkg.NgsFHMM(8, 200, mode=pkg.MODE_FAST)
    h.set_switch("estmaf_interp", 0)
    h.set_switch("bg_parts", 3)
    with pytest.raises(pkg.NgsFHMMError):
        h.set_switch("no_such_switch", 1)
    with pytest.raises(pkg.NgsFHMMError):
        h.set_switch("fast_c", 7)                              # layout: fixed at creation
    os.environ["NGHMM_FAST_C"] = "2"
    try:
        g = pkg.NgsFHMM(8, 2000, mode=pkg.MODE_FAST)
    finally:
        del os.environ["NGHMM_FAST_C"]
    assert g.layout()[0] == 2 and h.layout()[0] != 2
    h.close()
    g.close()

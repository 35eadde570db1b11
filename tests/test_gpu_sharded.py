"""The multi-GPU entry points of the C ABI on ONE GPU: two handles play two ranks
(individuals split in halves), the test moves the buffers between them exactly as the
all-to-all / all-gather of ngsf-hmm_amd/distributed.py would, and the result must equal
a single handle that owns every individual."""
import ctypes as C
import importlib

import numpy as np
import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def test_two_shards_equal_one_handle(pkg):
    import torch
    dd = importlib.import_module("ngsf-hmm_amd.distributed")
    world, I_loc, S = 2, 33, 1200
    d = pkg.simulate.simulate(I_loc * world, S, seed=8, n_chrom=2, missing_rate=0.05, indF="r")
    gl = pkg.simulate.normalise_log_gl(d.gl)
    dev = torch.device("cuda", 0)

    whole = pkg.NgsFHMM(I_loc * world, S, mode=pkg.MODE_FAST)
    whole.load(gl, d.pos_dist_mb)
    whole.set_params(0.1, 0.2, 0.1)
    whole.init_emission()

    ranges = dd.site_ranges(S, world)
    S_own = S // world
    ranks = []
    for r in range(world):
        be = dd.GpuBackend(pkg, I_loc, S, 0, pkg.MODE_FAST)
        be.hmm.load(np.ascontiguousarray(gl[:, r * I_loc:(r + 1) * I_loc]), d.pos_dist_mb)
        be.hmm.set_params(0.1, 0.2, 0.1)
        be.hmm.init_emission()
        be.shard_config(I_loc * world, r * I_loc, ranges[r][0], S_own)
        shard = torch.from_numpy(np.ascontiguousarray(gl[ranges[r][0]:ranges[r][1]])).to(dev)
        be.load_site_shard_device(shard)
        ranks.append(be)

    for it in range(2):
        whole.estep(); whole.mstep_indf(); whole.mstep_freq(1)
        send = []
        for be in ranks:
            be.estep(); be.mstep_indf(False, False)
            buf = be.empty(world, S_own, I_loc)
            for q, (lo, hi) in enumerate(ranges):
                be.pack_posteriors(lo, hi, buf[q])
            send.append(buf)
        torch.cuda.synchronize()
        freq_all = torch.empty(S, device=dev, dtype=torch.float64)
        for r, be in enumerate(ranks):
            recv = torch.stack([send[q][r] for q in range(world)]).contiguous()   # the all-to-all
            own = be.empty(S_own)
            torch.cuda.synchronize()
            be.mstep_freq_sites(recv, own)
            freq_all[ranges[r][0]:ranges[r][1]] = own                             # the all-gather
        torch.cuda.synchronize()
        for be in ranks:
            be.set_freq(freq_all)
        got_freq = ranks[0].hmm.freq
        np.testing.assert_allclose(got_freq, whole.freq, rtol=1e-12)
        for r, be in enumerate(ranks):
            sl = slice(r * I_loc, (r + 1) * I_loc)
            np.testing.assert_allclose(be.hmm.ind_lkl, whole.ind_lkl[sl], rtol=1e-12)
            np.testing.assert_allclose(be.hmm.indF, whole.indF[sl], atol=1e-6)
    for be in ranks:
        be.hmm.close()
    whole.close()


def test_bench_multi_rank_path_on_one_gpu(pkg):
    """`python bench.py --gpus 2` as the driver starts it (no torch.distributed environment):
    bench.py launches the two ranks itself before any GPU call; on this one-GPU box both
    ranks share cuda:0 and the collectives go through gloo staged on the host (the line says
    so).  Strong scaling is the default, and every rank holds a slice of the data set the N = 1
    job processes (simulate.IndexedSim): the lines' `check` objects -- two EM iterations from the
    starting values -- agree between one rank, two individual shards, two site shards and the
    site-shard run's embedded `alt_sharding`.  Functional check only; the measured multi-GPU configuration uses nccl (= RCCL) and is
    run by the driver on an 8-GPU node."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NGHMM_BENCH_BACKEND")}

    def run(*extra, workload="tiny"):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1",
               "--workload", workload, "--no_cpu_baseline", *extra]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])

    sys.path.insert(0, root)
    import bench

    def same(a, b):
        d = bench.compare_checks(a, b)
        assert d["tot_lkl_max_rel_diff"] <= 1e-12 and d["freq_probe_max_rel_diff"] <= 1e-9, d
        assert d["freq_weighted_sum_rel_diff"] <= 1e-9 and d["rounds_equal"], d

    one = run()
    two = run("--gpus", "2", "--shard", "individuals")
    assert one["n_gpus"] == 1 and one["ranks"] == 1 and one["config"]["n_ind_per_gpu"] == 64
    assert two["alt_sharding"] is None and one["vs_n1"] is None
    same(two["check"], one["check"])
    assert two["n_gpus"] == 2 and two["ranks"] == 2 and two["scaling"] == "strong"
    assert two["config"]["n_ind_total"] == 64 and two["config"]["n_ind_per_gpu"] == 32
    assert "gloo" in two["collectives"] and two["value"] > 0
    # the default in fast mode: SITE shards -- every rank all 64 individuals for half the sites,
    # one small all-gather per E-step and per objective round, nothing for the frequency step
    st = run("--gpus", "2")
    c = st["config"]
    assert st["n_gpus"] == 2 and st["scaling"] == "strong" and c["sharding"].startswith("sites:")
    assert c["n_ind_total"] == 64 and c["n_ind_per_gpu"] == 64
    assert c["n_sites"] == one["config"]["n_sites"] and c["n_sites_per_gpu"] * 2 == c["n_sites"]
    cb = st["collective_bytes_per_iter"]
    assert cb["all_to_all_out"] == 0 and cb["all_gathers"] >= 2 and 0 < cb["all_gather_out"] < 1e6
    assert [p["all_gathers_per_iter"] for p in st["per_rank"]] == [cb["all_gathers"]] * 2
    assert st["per_rank"][0]["rounds_per_iter"] == st["per_rank"][1]["rounds_per_iter"]
    same(st["check"], one["check"])
    alt = st["alt_sharding"]                 # the other layout from the same invocation
    assert alt["sharding"] == "individuals" and alt["n_ind_per_gpu"] == 32 and alt["ms_per_step"] > 0
    same(alt["check"], one["check"])
    assert alt["vs_main_sharding"]["ok"] is True
    # a layout that does not fit the job (63 individuals over two ranks) costs the line its
    # `alt_sharding` object, not the run
    odd = run("--gpus", "2", "--n_ind", "63")
    assert odd["value"] > 0 and "do not divide" in odd["alt_sharding"]["skipped"]
    sw = run("--gpus", "2", "--scaling", "weak")
    assert sw["config"]["n_sites"] == 2 * one["config"]["n_sites"] and sw["config"]["n_ind_total"] == 64
    # ... and weak scaling in the other layout: every rank its own 64 individuals
    swi = run("--gpus", "2", "--scaling", "weak", "--shard", "individuals")
    assert swi["config"]["n_ind_total"] == 128 and swi["config"]["n_ind_per_gpu"] == 64 and swi["value"] > 0
    assert swi["scaling"] == "weak" and swi["check"]["rounds"][0] >= 2
    # BASELINE configs[4]'s path at smoke size: --call_geno, 2-bit packed handles, the site
    # shards built from exchanged genotype codes
    cg = run("--gpus", "2", workload="tinycg")           # site shards, and individual shards as `alt_sharding`
    assert cg["n_gpus"] == 2 and cg["value"] > 0 and cg["config"]["sharding"].startswith("sites:")
    assert "packed" in cg["config"]["workload"] and cg["alt_sharding"]["vs_main_sharding"]["ok"] is True
    # a rank count that does not match --gpus is an error, not a silent N = 1 run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2",
                          "--workload", "tiny", "--no_cpu_baseline"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert bad.returncode != 0 and "WORLD_SIZE is 1" in bad.stderr


def test_bench_a_failing_rank_ends_the_job_with_its_message(pkg):
    """One rank of `bench.py --gpus 2` hits a fatal (injected after the warm-up: `invalid MAF!` is
    reachable from user data on ONE rank's site range) while the other goes on into its next
    all-gather: the job must end non-zero within seconds, the healthy rank with the failing
    rank's message (distributed.FailureBeacon), not after the process group's timeout."""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NGHMM_BENCH_BACKEND")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--workload", "tiny", "--no_cpu_baseline", "--no_alt",
                        "--timeout_s", "300"],
                       env=dict(env, NGHMM_BENCH_FAIL_RANK="1"), capture_output=True, text=True,
                       timeout=280, cwd=root)
    dt = time.time() - t0
    assert r.returncode != 0
    assert "invalid MAF! (injected" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]      # no line from a broken job
    assert dt < 120, dt
    # who ended the healthy rank: the beacon (its message is then on stderr) or torchrun's agent,
    # which also terminates the other workers when one fails -- whichever came first; a launcher
    # without that (mpirun, srun) leaves it to the beacon alone (tests/test_distributed_cpu.py
    # covers the beacon by itself)
    print("healthy rank ended by:", "the failure beacon" if "a peer failed -- rank 1" in r.stderr
          else "the launcher", f"after {dt:.1f} s")


@pytest.mark.parametrize("I", [40, 100, 200, 400, 600, 700, 1024, 1100, 1700, 2100, 3500, 4100, 7000, 8200])
def test_est_maf_register_and_stream_variants(pkg, I):
    """est_maf picks a kernel by the number of individuals (one wave per site with 1..16
    individuals per lane in registers up to 1024 (12 per lane from 513 to 768), then 2, 4 or 8 waves
    per site with 12 individuals per lane in the lower half of each size class and 16 in the upper; beyond 8192
    the streaming kernel), as the site-sharded frequency step of an N-GPU run needs: all
    must agree with the oracle."""
    import orclib
    S = 96 if I < 4000 else 24
    d = pkg.simulate.simulate(I, S, seed=I, missing_rate=0.05)
    gl = pkg.simulate.normalise_log_gl(d.gl)
    orc = orclib.Oracle("libm")
    em = orclib.OracleEM(orc, gl, d.pos_dist_mb)
    em.set_params(0.3, 0.05, 0.1)
    em.init_emission()
    assert em.estep() == 0 and em.mstep_freq(1) == 0
    hmm = pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST)
    hmm.load(gl, d.pos_dist_mb)
    hmm.set_params(0.3, 0.05, 0.1)
    hmm.init_emission()
    hmm.estep()
    hmm.mstep_freq(1)
    np.testing.assert_allclose(hmm.freq, em.freq, rtol=1e-9)
    hmm.close()


@pytest.mark.parametrize("I", [3, 40, 130, 700, 1024, 1100, 2100, 4100, 5000, 8200, 9000])
def test_est_maf_called_genotypes_closed_form(pkg, I):
    """Called genotypes (packed handles): a genotype's posterior in est_maf's pass is a unit
    vector whatever the frequency, a missing cell's is HWE itself, so the per-pass sums are a
    constant plus polynomials in f with three per-site coefficients: one sweep over codes and
    posteriors and a scalar recursion per site (k_fast_estmaf_called_sums / _passes), at any
    cohort size, in place on the tile-major posteriors.  Against the oracle (1e-9) and against
    the general kernels on the same handle (switch estmaf_no_called; 1e-12) -- including sites
    with a called heterozygote at posterior IBD exactly 1, where the reference's weights all
    vanish and both routes hand the site to the log-space kernel."""
    import orclib
    S = 96 if I < 4000 else 24
    d = pkg.simulate.simulate(I, S, seed=I + 1, missing_rate=0.1, n_chrom=2)
    orc = orclib.Oracle("libm")
    gl = orc.prepare_gl(d.gl, 0, call_geno=True)
    em = orclib.OracleEM(orc, gl, d.pos_dist_mb)
    # alpha small and F large: long IBD tracts, posteriors snapped to exactly 0 and 1
    em.set_params(0.6, 0.02, 0.2)
    em.init_emission()
    assert em.estep() == 0 and em.mstep_freq(1) == 0
    het_at_one = ((em.marg == 1.0).T & (np.argmax(gl, axis=2) == 1) & (gl.max(axis=2) == 0)).any(axis=1)
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST | pkg.GENO_PACKED) as hmm:
        hmm.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=True)
        got = {}
        for general in (0, 1):
            hmm.set_switch("estmaf_no_called", general)
            hmm.set_params(0.6, 0.02, 0.2)
            hmm.init_emission()
            hmm.estep()
            hmm.mstep_freq(1)
            got[general] = hmm.freq
            # the log-space route of a site with a vanishing cell rounds like the reference
            # (terms of magnitude 1e15): compared where it is not taken, finite everywhere
            assert np.isfinite(got[general]).all()
            ok = ~het_at_one
            np.testing.assert_allclose(got[general][ok], em.freq[ok], rtol=1e-9)
        np.testing.assert_allclose(got[0], got[1], rtol=1e-12)
        # --freq e: est_maf at F = 0 for everybody (parse_args.cpp:312-318)
        hmm.set_switch("estmaf_no_called", 0)
    print(f"I = {I}: {int(het_at_one.sum())} of {S} sites with a called heterozygote at posterior 1")


@pytest.mark.parametrize("n,packed", [(2, False), (4, False), (2, True)])
def test_group_of_handles_equals_one_handle(pkg, n, packed):
    """nghmm_group_setup / nghmm_group_iter_em (one process, several handles; on an 8-GPU node
    one per GPU with peer copies over xGMI, here all on cuda:0): the cohort's EM iterations do
    not depend on how the individuals are split."""
    I_loc, S = 24, 2400
    I = I_loc * n
    d = pkg.simulate.simulate(I, S, seed=21, n_chrom=3, missing_rate=0.04, indF="r")
    mode = pkg.MODE_FAST | (pkg.GENO_PACKED if packed else 0)

    def load(h, cols):
        h.load_raw(np.ascontiguousarray(d.gl[:, cols]), d.pos_dist_mb, space=0, call_geno=packed)
        h.set_params(0.1, 0.2, 0.1)
        h.init_emission()

    whole = pkg.NgsFHMM(I, S, mode=mode)
    load(whole, slice(0, I))
    parts = [pkg.NgsFHMM(I_loc, S, mode=mode) for _ in range(n)]
    for r, h in enumerate(parts):
        load(h, slice(r * I_loc, (r + 1) * I_loc))
    grp = pkg.Group(parts)
    for it in range(3):
        whole.iter_EM()
        st = grp.iter_EM()
        np.testing.assert_allclose(grp.ind_lkl, whole.ind_lkl, rtol=1e-12)
        np.testing.assert_allclose(parts[0].freq, whole.freq, rtol=1e-12)
        np.testing.assert_array_equal(parts[0].freq, parts[-1].freq)
        got_F = np.concatenate([h.indF for h in parts])
        np.testing.assert_allclose(got_F, whole.indF, atol=1e-6)
        assert st.rounds > 0
    post = np.concatenate([h.marg_prob for h in parts])
    np.testing.assert_allclose(post, whole.marg_prob, rtol=1e-9, atol=1e-12)
    # fixed parameters: nothing for the optimizer to amplify -> the split must not matter at all
    for h in parts + [whole]:
        h.set_params(0.3, 0.1, 0.15)
        h.init_emission()
    whole.iter_EM(1, True, True)
    grp.iter_EM(1, True, True)
    np.testing.assert_allclose(parts[1].freq, whole.freq, rtol=1e-13)
    for h in parts:
        h.close()
    whole.close()


@pytest.mark.parametrize("I", [16, 40, 100, 128])
def test_est_maf_rows_kernel_equals_wave_kernel(pkg, I):
    """Below 129 individuals est_maf runs four sites per wave, one per 16-lane DPP row
    (k_fast_estmaf_rows); the switch estmaf_no_rows forces the one-site-per-wave kernel.  Same
    recursion and interval logic: the frequencies must agree to summation-order rounding on
    data whose sites travel far (uniform site frequencies: odds from 0.01 up to ~20, several
    intervals per site).  The lanes of a row decide for themselves from row totals, which
    therefore must be the same bits in all 16 lanes -- a contracted first addition once
    made DPP partners differ in the last bit and split rows at the 1e-13 check."""
    import os
    import torch
    S = 60_000
    gl, pos = pkg.simulate.simulate_torch(I, S, torch.device("cuda", 0), seed=3, freq="r")
    torch.cuda.synchronize()
    hmm = pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST)
    hmm.load_device(gl.data_ptr(), pos.data_ptr())
    del gl, pos
    res = {}
    try:
        for rows in ("1", "0"):
            hmm.set_switch("estmaf_no_rows", 1 if rows == "0" else 0)
            hmm.set_params(np.full(I, 0.1), np.full(I, 0.01), np.full(S, 0.1))
            hmm.init_emission()
            hmm.estep()
            hmm.mstep_freq(1)
            res[rows] = hmm.freq.copy()
    finally:
        hmm.close()
    rel = np.abs(res["1"] - res["0"]) / res["0"]
    assert rel.max() < 1e-12, (rel.max(), int((rel > 1e-12).sum()))

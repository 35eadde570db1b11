"""End to end through the command line on a GPU: the C++ host with --mode exact must
write .indF / .ibd / .geno files that are BYTE-IDENTICAL to what the reference's
print_iter (EM.cpp:293-380) would write from the oracle's results, for the three input
encodings of examples/test.sh (called genotypes .geno.gz, log-GL text .glf.gz, binary
GL, and --call_geno) -- the drop-in criterion of BASELINE.json configs[0]."""
import os

import numpy as np
import pytest

import cli_util
import orclib
from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

I, S = 10, 700


@pytest.fixture(scope="module")
def data(pkg, tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("cli"))
    d = pkg.simulate.simulate(I, S, seed=12345, n_chrom=2)
    return d, cli_util.write_inputs(tmp, d, d.gl), tmp


def _oracle_outputs(orc_det, orc_libm, gl, d, freq0, indF0, alpha0, min_iters, max_iters,
                    freq_est=1, indF_fixed=False):
    em = orclib.OracleEM(orc_det, gl, d.pos_dist_mb)
    em.set_params(indF0, alpha0, freq0)
    assert em.init_emission() == 0
    n = em.run(freq_est=freq_est, indF_fixed=indF_fixed, min_iters=min_iters, max_iters=max_iters)
    path = em.viterbi()
    gp = em.geno_post(path)              # .geno posteriors (EM.cpp:367-376), on the device too
    return n, cli_util.expected_files(em.tot_lkl, em.indF, em.alpha, em.freq, em.ind_lkl, path,
                                      em.marg, gp)


CASES = [
    # GL_beagle_pieces: the text reader's pipeline in pieces of 777 bytes (every line cut across
    # reads) and device blocks of 13 sites
    ("GL_beagle_pieces", ["--lkl"], "beagle_gz", False, False),
    ("GL_text", ["--loglkl"], "glf_gz", False, False),
    ("GL_bin", ["--loglkl"], "glf_bin", False, False),
    ("GL_callgeno", ["--loglkl", "--call_geno"], "glf_gz", True, False),
    ("TG", [], "geno_gz", False, True),
    ("GL_beagle", ["--lkl"], "beagle_gz", False, False),
    # the same text as a BGZF file (bgzip / ANGSD) in blocks of 500 bytes: parallel inflate
    ("GL_beagle_bgzf", ["--lkl"], "beagle_gz", False, False),
]


@pytest.mark.parametrize("name,flags,key,call,called", CASES)
def test_cli_outputs_byte_identical(pkg, orc_det, orc_libm, data, name, flags, key, call, called):
    d, paths, tmp = data
    # the raw values as the host's reader sees them; their preparation (log, normalisation,
    # genotype calling) runs on the device with the exact-mode arithmetic = the det oracle's
    space = 0
    if called:
        raw = cli_util.raw_called_genotypes(d.geno)
    elif name.startswith("GL_beagle"):
        raw, space = np.exp(d.gl), 2      # normal-space values of a text file: plain log
    else:
        raw = d.gl
    gl = orc_det.prepare_gl(raw, space, call_geno=call)
    out = os.path.join(tmp, "out_" + name)
    env = None
    if name == "GL_beagle_pieces":
        env = dict(os.environ, NGHMM_HOST_CHUNK_BYTES="777", NGHMM_HOST_BLOCK_SITES="13")
    geno_path = paths[key]
    if name == "GL_beagle_bgzf":
        import gzip
        geno_path = os.path.join(tmp, "bgzf.beagle.gz")
        cli_util.write_bgzf(geno_path, gzip.open(paths[key], "rb").read(), 500)
    r = cli_util.run_cli(["--geno", geno_path, *flags, "--pos", paths["pos_gz"], "--n_ind", I,
                          "--n_sites", S, "--freq", 0.1, "--indF", "0.1,0.2", "--out", out,
                          "--min_iters", 3, "--max_iters", 5, "--mode", "exact", "--seed", 12345,
                          "--n_threads", 4 if name != "TG" else 1,
                          "--verbose", 2 if name == "GL_beagle_bgzf" else 1], env=env)
    n, (f_indF, f_ibd, f_geno) = _oracle_outputs(orc_det, orc_libm, gl, d, 0.1, 0.1, 0.2, 3, 5)
    assert f"Iteration {n}:" in r.stdout and f"Iteration {n + 1}:" not in r.stdout
    assert (": BGZF)" in r.stdout) == (name == "GL_beagle_bgzf")
    assert open(out + ".indF", "rb").read() == f_indF
    assert open(out + ".ibd", "rb").read() == f_ibd
    assert open(out + ".geno", "rb").read() == f_geno


def test_cli_fixed_parameters_and_fast_mode(pkg, orc_det, orc_libm, data):
    """examples/test.sh 'indF_fixed' configuration; fast mode must agree to print precision
    on freq/posteriors (no optimizer in the loop to amplify anything)."""
    d, paths, tmp = data
    for mode in ("exact", "fast"):
        out = os.path.join(tmp, "fixed_" + mode)
        cli_util.run_cli(["--geno", paths["glf_gz"], "--loglkl", "--pos", paths["pos_gz"],
                          "--n_ind", I, "--n_sites", S, "--freq", 0.1, "--indF", "0.5,0.01",
                          "--indF_fixed", "--alpha_fixed", "--out", out, "--min_iters", 2,
                          "--max_iters", 3, "--mode", mode, "--verbose", 0])
    a = open(os.path.join(tmp, "fixed_exact.indF")).read().split("\n")
    b = open(os.path.join(tmp, "fixed_fast.indF")).read().split("\n")
    assert a[1:] == b[1:]                              # indF/alpha/freq lines as printed
    assert abs(float(a[0]) - float(b[0])) < 1e-6 * abs(float(a[0]))
    ia = open(os.path.join(tmp, "fixed_exact.ibd")).read().split("\n")
    ib = open(os.path.join(tmp, "fixed_fast.ibd")).read().split("\n")
    assert ia[1:1 + I] == ib[1:1 + I]                  # Viterbi paths identical


def test_cli_freq_e_initialisation(pkg, orc_det, orc_libm, data):
    """--freq e: initial frequencies from est_maf with F = 0 (parse_args.cpp:312-318)."""
    d, paths, tmp = data
    gl = orc_det.prepare_gl(d.gl)
    out = os.path.join(tmp, "freq_e")
    cli_util.run_cli(["--geno", paths["glf_bin"], "--loglkl", "--pos", paths["pos_gz"], "--n_ind",
                      I, "--n_sites", S, "--freq", "e", "--freq_est", 0, "--indF", "0.1,0.2",
                      "--indF_fixed", "--alpha_fixed", "--out", out, "--min_iters", 1,
                      "--max_iters", 2, "--mode", "exact", "--verbose", 0])
    want = [orc_det.est_maf(gl[s], np.zeros(I))[0] for s in range(S)]
    got = [float(x) for x in open(out + ".indF").read().split("\n")[1 + I:1 + I + S]]
    assert got == [float("%f" % w) for w in want]


def test_cli_freq_e_and_log_in_fast_mode(pkg, orc_det, orc_libm, data):
    """Fast mode before its first E-step: the posteriors must be the reference's initial
    zeros (parse_args.cpp:403-405), not whatever the allocation held.  (1) --freq e =
    est_maf with F = 0; (2) --log N prints at iteration 0 (EM.cpp:59-63): posterior lines
    of 0.000000, paths of '0'."""
    d, paths, tmp = data
    gl = orc_libm.prepare_gl(d.gl)
    # a few handles first, so that the next allocation recycles used device memory
    for _ in range(2):
        with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as h:
            h.load(gl, d.pos_dist_mb); h.set_params(0.3, 0.2, 0.2); h.init_emission(); h.estep()
    out = os.path.join(tmp, "freq_e_fast")
    cli_util.run_cli(["--geno", paths["glf_bin"], "--loglkl", "--pos", paths["pos_gz"], "--n_ind",
                      I, "--n_sites", S, "--freq", "e", "--freq_est", 0, "--indF", "0.1,0.2",
                      "--indF_fixed", "--alpha_fixed", "--out", out, "--min_iters", 1,
                      "--max_iters", 2, "--mode", "fast", "--verbose", 0])
    want = np.array([orc_libm.est_maf(gl[s], np.zeros(I))[0] for s in range(S)])
    got = np.array([float(x) for x in open(out + ".indF").read().split("\n")[1 + I:1 + I + S]])
    assert np.abs(got - want).max() < 1e-6          # the file holds 6 decimals
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as h:
        h.load(gl, d.pos_dist_mb)
        h.set_params(0.1, 0.2, 0.1)
        h.mstep_freq(1)                              # posteriors still 0
        np.testing.assert_allclose(h.freq, want, rtol=1e-9)
        assert not h.marg_prob.any()
        assert h.format_posteriors(0, 2) == (b"\t".join([b"0.000000"] * S) + b"\n") * 2


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_cli_log_prints_initial_state(pkg, data, mode):
    """--log 5 with --max_iters 2: intermediate prints at iteration 0 (before any E-step:
    posteriors and paths are the initial zeros) and at iteration 1 (EM.cpp:59-63: iter == 1
    always prints); the run must succeed and leave complete final files."""
    d, paths, tmp = data
    out = os.path.join(tmp, "log_" + mode)
    r = cli_util.run_cli(["--geno", paths["glf_gz"], "--loglkl", "--pos", paths["pos_gz"], "--n_ind",
                          I, "--n_sites", S, "--freq", 0.1, "--indF", "0.1,0.2", "--out", out,
                          "--min_iters", 1, "--max_iters", 2, "--mode", mode, "--log", 5,
                          "--verbose", 1])
    assert r.stdout.count("==> Printing current iteration parameters") == 2
    ibd = open(out + ".ibd").read().split("\n")
    assert len(ibd) == 1 + 2 * I + 1 and all(len(l) == S for l in ibd[1:1 + I])


def test_cli_random_initial_values(pkg, data):
    """The reference's DEFAULT start (--freq r, and --indF r): gsl_rng_taus seeded with
    --seed, indF_i / alpha_i interleaved, then one frequency per site
    (parse_args.cpp:232-233,248-254,305-310).  With everything fixed the output files
    print the initial values; the generator itself has its known-answer test in
    tests/test_cli_cpu.py."""
    import pyref
    d, paths, tmp = data
    for seed in (12345, 1):
        out = os.path.join(tmp, f"rand_{seed}")
        cli_util.run_cli(["--geno", paths["glf_gz"], "--loglkl", "--pos", paths["pos_gz"],
                          "--n_ind", I, "--n_sites", S, "--freq", "r", "--freq_est", 0, "--indF",
                          "r", "--indF_fixed", "--alpha_fixed", "--out", out, "--min_iters", 1,
                          "--max_iters", 2, "--mode", "fast", "--seed", seed, "--verbose", 0])
        indF, alpha, freq = pyref.random_initial_values(seed, I, S)
        lines = open(out + ".indF").read().split("\n")
        assert lines[1:1 + I] == ["%.5f\t%f" % (f, a) for f, a in zip(indF, alpha)]
        assert lines[1 + I:1 + I + S] == ["%f" % f for f in freq]
    # --freq r alone (the reference's default): the generator is consumed by the
    # frequencies only
    out = os.path.join(tmp, "rand_freq_only")
    cli_util.run_cli(["--geno", paths["glf_gz"], "--loglkl", "--pos", paths["pos_gz"], "--n_ind",
                      I, "--n_sites", S, "--freq_est", 0, "--indF_fixed", "--alpha_fixed",
                      "--out", out, "--min_iters", 1, "--max_iters", 2, "--seed", 7,
                      "--verbose", 0])
    _, _, freq = pyref.random_initial_values(7, I, S, indF_random=False)
    lines = open(out + ".indF").read().split("\n")
    assert lines[1:1 + I] == ["%.5f\t%f" % (0.01, 0.001)] * I     # default --indF 0.01-0.001
    assert lines[1 + I:1 + I + S] == ["%f" % f for f in freq]


# ---- examples/test.sh part A: its five configurations x three input types at its size ----
TS_I, TS_S = 10, 10000
TS_CONFIGS = {   # examples/test.sh:40-53 (FREQ = 0.2, INDF = 0.5, ALPHA = 0.01)
    "TRUE": (["--freq", 0.2, "--freq_est", 0, "--indF", "0.5,0.01", "--indF_fixed"],
             dict(freq0=0.2, indF0=0.5, alpha0=0.01, freq_est=0, indF_fixed=True)),
    "BEST": (["--freq", 0.2, "--indF", "0.5,0.01"],
             dict(freq0=0.2, indF0=0.5, alpha0=0.01)),
    "freq_fixed": (["--freq", 0.2, "--freq_est", 0, "--indF", "0.1,0.2"],
                   dict(freq0=0.2, indF0=0.1, alpha0=0.2, freq_est=0)),
    "indF_fixed": (["--freq", 0.1, "--indF", "0.5,0.01", "--indF_fixed"],
                   dict(freq0=0.1, indF0=0.5, alpha0=0.01, indF_fixed=True)),
    "normal": (["--freq", 0.1, "--indF", "0.1,0.2"], dict(freq0=0.1, indF0=0.1, alpha0=0.2)),
}
TS_TYPES = {     # examples/test.sh:28-38
    "TG": ("geno_gz", [], False),
    "GL": ("glf_gz", ["--loglkl"], False),
    "GL_CG": ("glf_gz", ["--loglkl", "--call_geno"], True),
}


@pytest.fixture(scope="module")
def testsh_data(pkg, tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("testsh"))
    # test.sh:10-23: --freq 0.2 --site_pos r --indF 0.5 --alpha 0.01 --depth 2 --error 0.01
    d = pkg.simulate.simulate(TS_I, TS_S, seed=12345, freq=0.2, indF=0.5, alpha=0.01, depth=2.0,
                              error=0.01)
    return d, cli_util.write_inputs(tmp, d, d.gl), tmp


@pytest.mark.parametrize("typ", list(TS_TYPES))
@pytest.mark.parametrize("cfg", list(TS_CONFIGS))
def test_examples_test_sh_matrix(pkg, orc_det, orc_libm, testsh_data, cfg, typ):
    """BASELINE.json configs[0]: the reference's own regression matrix (examples/test.sh:
    TRUE, BEST, freq_fixed, indF_fixed, normal x TG, GL, GL_CG at 10 x 10 000, --log 1,
    --n_threads 10, --seed 12345).  Its md5s pin outputs of inputs that need R to
    regenerate, so the comparison is against the files print_iter would write from the
    oracle's results (byte for byte, exact mode).  Deviation, stated: the runs are capped
    at 6 iterations (--min_iters 3 --max_iters 6) to keep the suite short."""
    d, paths, tmp = testsh_data
    key, flags, call = TS_TYPES[typ]
    args, okw = TS_CONFIGS[cfg]
    raw = cli_util.raw_called_genotypes(d.geno) if typ == "TG" else d.gl
    gl = orc_det.prepare_gl(raw, 0, call_geno=call)
    out = os.path.join(tmp, f"testF-HMM.{cfg}.{typ}")
    r = cli_util.run_cli(["--verbose", 2, "--n_threads", TS_I, "--seed", 12345, "--geno",
                          paths[key], *flags, "--n_ind", TS_I, "--n_sites", TS_S, "--pos",
                          paths["pos_gz"], *args, "--out", out, "--log", 1, "--min_iters", 3,
                          "--max_iters", 6, "--mode", "exact"])
    em = orclib.OracleEM(orc_det, gl, d.pos_dist_mb)
    em.set_params(okw["indF0"], okw["alpha0"], okw["freq0"])
    assert em.init_emission() == 0
    n = em.run(freq_est=okw.get("freq_est", 1), indF_fixed=okw.get("indF_fixed", False),
               min_iters=3, max_iters=6, n_threads=TS_I)
    path = em.viterbi(n_threads=TS_I)
    f_indF, f_ibd, f_geno = cli_util.expected_files(em.tot_lkl, em.indF, em.alpha, em.freq,
                                                    em.ind_lkl, path, em.marg, em.geno_post(path))
    assert f"Iteration {n}:" in r.stdout and f"Iteration {n + 1}:" not in r.stdout
    assert r.stdout.count("==> Printing current iteration parameters") == n   # --log 1
    assert open(out + ".indF", "rb").read() == f_indF
    assert open(out + ".ibd", "rb").read() == f_ibd
    assert open(out + ".geno", "rb").read() == f_geno


def test_cli_packed_default_equals_no_pack(pkg, data):
    """The host keeps called genotypes (a called-genotype file, --call_geno) as 2-bit codes
    unless --no_pack: the files of the two runs must be identical byte for byte in exact mode;
    in fast mode (the packed handle's frequency step is the called genotypes' closed form) the
    same paths and the same printed values up to the optimizer's spread."""
    d, paths, tmp = data
    for name, key, flags in (("tg", "geno_gz", []), ("cg", "glf_bin", ["--loglkl", "--call_geno"])):
        for mode in ("exact", "fast"):
            outs = []
            for extra in ([], ["--no_pack"]):
                out = os.path.join(tmp, f"pk_{name}_{mode}_{len(extra)}")
                cli_util.run_cli(["--geno", paths[key], *flags, "--pos", paths["pos_gz"], "--n_ind", I,
                                  "--n_sites", S, "--freq", 0.1, "--indF", "0.1,0.2", "--out", out,
                                  "--min_iters", 2, "--max_iters", 3, "--mode", mode, "--verbose", 0,
                                  *extra])
                outs.append(out)
            if mode == "fast":
                # the packed handle on the GENERAL est_maf kernels (switch estmaf_no_called): then
                # packed and unpacked differ in nothing but where a cell's likelihoods come from
                # (2-bit code + class table, or three doubles), and every file must be the
                # unpacked run's byte for byte -- this pins the packed emission / est_maf reads
                out = os.path.join(tmp, f"pk_{name}_{mode}_nc")
                cli_util.run_cli(["--geno", paths[key], *flags, "--pos", paths["pos_gz"], "--n_ind", I,
                                  "--n_sites", S, "--freq", 0.1, "--indF", "0.1,0.2", "--out", out,
                                  "--min_iters", 2, "--max_iters", 3, "--mode", mode, "--verbose", 0],
                                 env=dict(os.environ, NGHMM_ESTMAF_NO_CALLED="1"))
                for ext in (".indF", ".ibd", ".geno"):
                    assert open(out + ext, "rb").read() == open(outs[1] + ext, "rb").read(), (name, ext)
            for ext in (".indF", ".ibd", ".geno"):
                a, b = open(outs[0] + ext, "rb").read(), open(outs[1] + ext, "rb").read()
                if mode == "exact":
                    assert a == b, (name, mode, ext)
                    continue
                if ext == ".indF":      # lkl, (indF, alpha) per individual, a frequency per site
                    ta, tb = a.decode().split(), b.decode().split()
                    assert len(ta) == len(tb)
                    assert [x == "NA" for x in ta] == [x == "NA" for x in tb]
                    va = np.array([float(x) for x in ta if x != "NA"])
                    vb = np.array([float(x) for x in tb if x != "NA"])
                    assert np.all(np.abs(va - vb) <= 2e-6 + 1e-6 * np.abs(vb)), np.abs(va - vb).max()
                    continue
                # fast mode: a packed handle's est_maf is the called genotypes' closed form
                # (k_fast_estmaf_called_sums), an unpacked one's the general kernel's interpolated
                # passes -- the same frequencies to ~1e-15, which three free L-BFGS-B iterations
                # turn into ~1e-8: the same decoded paths, every printed value to its last digits
                assert len(a) == len(b), (name, mode, ext)
                if ext == ".ibd":
                    la, lb = a.decode().split("\n"), b.decode().split("\n")
                    assert la[1:1 + I] == lb[1:1 + I]                     # Viterbi paths
                    for x, y in ((la[0], lb[0]), ("\t".join(la[1 + I:1 + 2 * I]), "\t".join(lb[1 + I:1 + 2 * I]))):
                        va = np.array([float(t) for t in x.split("\t") if t not in ("//", "")])
                        vb = np.array([float(t) for t in y.split("\t") if t not in ("//", "")])
                        assert va.shape == vb.shape
                        assert np.all(np.abs(va - vb) <= 2e-6 + 1e-7 * np.abs(vb)), np.abs(va - vb).max()
                elif ext == ".geno":
                    np.testing.assert_allclose(np.frombuffer(a, dtype=np.float64),
                                               np.frombuffer(b, dtype=np.float64), rtol=1e-6, atol=1e-300)


def test_cli_empty_line_in_called_genotype_file(pkg, orc_det, orc_libm, data):
    """An empty line in a text input still consumes a site (read_data.cpp:60-61): its cells
    keep the reader's initial -1e15 and see only the second normalisation.  2-bit codes
    cannot express that, so the host falls back to likelihoods; the files must be what the
    oracle gives for those cells."""
    import gzip
    d, paths, tmp = data
    lines = gzip.open(paths["geno_gz"], "rt").read().split("\n")[:S]
    lines[41] = ""
    path = os.path.join(tmp, "hole.geno.gz")
    with gzip.open(path, "wt") as fh:
        fh.write("\n".join(lines) + "\n")
    raw = cli_util.raw_called_genotypes(d.geno)
    raw[41, :, 0] = np.frombuffer(np.uint64(0x7ff8dead00000001).tobytes(), dtype=np.float64)[0]
    gl = orc_det.prepare_gl(raw, 0, call_geno=False)
    out = os.path.join(tmp, "hole")
    cli_util.run_cli(["--geno", path, "--pos", paths["pos_gz"], "--n_ind", I, "--n_sites", S,
                      "--freq", 0.1, "--indF", "0.1,0.2", "--out", out, "--min_iters", 2,
                      "--max_iters", 3, "--mode", "exact", "--verbose", 0])
    n, (f_indF, f_ibd, f_geno) = _oracle_outputs(orc_det, orc_libm, gl, d, 0.1, 0.1, 0.2, 2, 3)
    assert open(out + ".indF", "rb").read() == f_indF
    assert open(out + ".ibd", "rb").read() == f_ibd
    assert open(out + ".geno", "rb").read() == f_geno


@pytest.mark.parametrize("key,flags", [("glf_bin", ["--loglkl"]), ("geno_gz", []),
                                       ("glf_gz", ["--loglkl", "--call_geno"])])
def test_cli_n_gpus_splits_the_cohort(pkg, data, key, flags):
    """--n_gpus N: the SITES split over N handles (on an 8-GPU node one per GPU; here
    --devices 0,0 / 0,0,0: handles on the one GPU), every handle all individuals, the ranges'
    operators exchanged by device copies (include/nghmm.h, "a CHAIN of site shards"); blocks of
    the input go to the handle that owns their sites, output lines are stitched from the
    handles' pieces.  Fixed indF/alpha: the files must equal the single-handle run's except for
    last-bit differences of the frequencies; free parameters: same paths and printed values up
    to the optimizer's spread."""
    d, paths, tmp = data
    base = ["--geno", paths[key], *flags, "--pos", paths["pos_gz"], "--n_ind", I, "--n_sites", S,
            "--freq", 0.1, "--min_iters", 2, "--max_iters", 4, "--mode", "fast", "--verbose", 0]
    for name, extra in (("fixed", ["--indF", "0.4,0.05", "--indF_fixed", "--alpha_fixed"]),
                        ("free", ["--indF", "0.1,0.2"]), ("freq_e", ["--indF", "0.1,0.2", "--freq", "e"])):
        one, two = os.path.join(tmp, f"g1_{key}_{name}"), os.path.join(tmp, f"g2_{key}_{name}")
        cli_util.run_cli(base + extra + ["--out", one])
        n_dev = 3 if name == "fixed" else 2      # three ranges: none a multiple of the block size
        cli_util.run_cli(base + extra + ["--out", two, "--n_gpus", n_dev, "--devices",
                                         ",".join(["0"] * n_dev)])
        a, b = open(one + ".indF").read().split("\n"), open(two + ".indF").read().split("\n")
        assert abs(float(a[0]) - float(b[0])) <= 1e-9 * abs(float(a[0]))
        fa = np.array([float(x) for x in a[1 + I:1 + I + S]])
        fb = np.array([float(x) for x in b[1 + I:1 + I + S]])
        assert np.abs(fa - fb).max() <= 1.1e-6
        ia, ib = open(one + ".ibd").read().split("\n"), open(two + ".ibd").read().split("\n")
        assert ia[1:1 + I] == ib[1:1 + I]                          # Viterbi paths
        if name == "fixed":
            assert a[1:1 + I] == b[1:1 + I]
            pa = np.array([[float(x) for x in l.split("\t")] for l in ia[1 + I:1 + 2 * I]])
            pb = np.array([[float(x) for x in l.split("\t")] for l in ib[1 + I:1 + 2 * I]])
            assert np.abs(pa - pb).max() <= 1.1e-6
            ga = np.fromfile(one + ".geno")
            gb = np.fromfile(two + ".geno")
            np.testing.assert_allclose(gb, ga, rtol=1e-9, atol=1e-300)

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
    ("GL_text", ["--loglkl"], "glf_gz", False, False),
    ("GL_bin", ["--loglkl"], "glf_bin", False, False),
    ("GL_callgeno", ["--loglkl", "--call_geno"], "glf_gz", True, False),
    ("TG", [], "geno_gz", False, True),
    ("GL_beagle", ["--lkl"], "beagle_gz", False, False),
]


@pytest.mark.parametrize("name,flags,key,call,called", CASES)
def test_cli_outputs_byte_identical(pkg, orc_det, orc_libm, data, name, flags, key, call, called):
    d, paths, tmp = data
    # the raw values as the host's reader sees them; their preparation (log, normalisation,
    # genotype calling) runs on the device with the exact-mode arithmetic = the det oracle's
    space = 0
    if called:
        raw = cli_util.raw_called_genotypes(d.geno)
    elif name == "GL_beagle":
        raw, space = np.exp(d.gl), 2      # normal-space values of a text file: plain log
    else:
        raw = d.gl
    gl = orc_det.prepare_gl(raw, space, call_geno=call)
    out = os.path.join(tmp, "out_" + name)
    r = cli_util.run_cli(["--geno", paths[key], *flags, "--pos", paths["pos_gz"], "--n_ind", I,
                          "--n_sites", S, "--freq", 0.1, "--indF", "0.1,0.2", "--out", out,
                          "--min_iters", 3, "--max_iters", 5, "--mode", "exact", "--seed", 12345,
                          "--n_threads", 4 if name != "TG" else 1, "--verbose", 1])
    n, (f_indF, f_ibd, f_geno) = _oracle_outputs(orc_det, orc_libm, gl, d, 0.1, 0.1, 0.2, 3, 5)
    assert f"Iteration {n}:" in r.stdout and f"Iteration {n + 1}:" not in r.stdout
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

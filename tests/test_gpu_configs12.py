"""BASELINE.json configs[1] and configs[2] at their own sizes and encodings, inside the `-m gpu`
suite (round 3 had them only as tools/ runs with committed profiles):

* configs[1]: 100 individuals x 100 000 sites, binary log-likelihood doubles, --freq_est 1 --
  fast mode against exact mode (which is bit-identical to the oracle) per call, and the EM
  ascent of a whole run;
* configs[2]'s ENCODING through the C++ host: a BGZF BEAGLE-style file (header, three id
  columns, 3 x I normal-space likelihoods per line, `--lkl`) of 1000 individuals x 50 000 sites,
  `--n_gpus 1` against `--n_gpus 2` (two site-shard handles) and fast against exact mode.
"""
import importlib
import os
import subprocess
import time

import numpy as np
import pytest

import cli_util
from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config2_100_x_100k_binary_log_gl(pkg):
    """configs[1] in full: `ngsF-HMMsim.R synthetic: 100 ind x 100k sites, binary GL doubles,
    --freq_est 1`, starting values of examples/test.sh (--freq 0.1 --indF 0.1,0.2)."""
    import torch
    I, S = 100, 100_000
    dev = torch.device("cuda", 0)
    sim = pkg.simulate.IndexedSim(I, S, dev, seed=4321, freq="r", indF="r", alpha=0.05)
    gl, pos = sim.gl(), sim.pos_dist(0, S)
    torch.cuda.synchronize()
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_EXACT) as ex, pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as fa:
        for h in (ex, fa):
            h.load_device(gl.data_ptr(), pos.data_ptr())
            h.set_params(0.1, 0.2, 0.1)
            h.init_emission()
        # -- per call, same parameters: E-step, objective at arbitrary points, frequency step
        le, lf = ex.estep().copy(), fa.estep().copy()
        np.testing.assert_allclose(lf, le, rtol=1e-12)
        # (posteriors: exp(Fw + Bw - lkl) of log-space values ~1e5 carries ~1e-8 of its own noise)
        pf, pe = fa.marg_prob, ex.marg_prob
        d = np.abs(pf - pe)
        snapped = (pf == 0) | (pf == 1) | (pe == 0) | (pe == 1)   # check_interv, gen_func.cpp:55-70:
        assert d[~snapped].max() <= 2e-7                          # a value on the other side of a
        assert d[snapped].max() <= 1e-5 + 2e-7                    # threshold moves by up to 1e-5
        assert np.count_nonzero(d > 2e-7) <= 1e-5 * d.size
        rng = np.random.default_rng(3)
        ind = rng.integers(0, I, 400).astype(np.uint32)
        F, A = rng.uniform(1e-3, 0.999, 400), rng.uniform(1e-3, 5.0, 400)
        np.testing.assert_allclose(fa.lkl(ind, F, A), ex.lkl(ind, F, A), rtol=1e-12)
        ex.mstep_freq(1)
        fa.mstep_freq(1)
        np.testing.assert_allclose(fa.freq, ex.freq, rtol=0, atol=1e-7)
        # -- a whole run in fast mode (EM.cpp:56,75-86).  The reference's iteration is not a
        # strict EM -- est_maf takes the IBD posterior for a per-site inbreeding coefficient --
        # so the total log-likelihood climbs steeply, may give back ~1e-5 of itself on the way
        # (exact mode does the same: compared below) and settles
        fa.set_params(0.1, 0.2, 0.1)
        fa.init_emission()
        tot = []
        t0 = time.time()
        fa.EM(min_iters=10, max_iters=20, callback=lambda it, h: tot.append(h.tot_lkl))
        dt = time.time() - t0
        tot = np.array(tot)
        assert len(tot) >= 10 and np.all(np.isfinite(tot))
        assert tot[1] > tot[0] and tot[-1] > tot[0] + 0.3 * abs(tot[0])
        assert np.all(np.diff(tot)[1:] > -1e-4 * np.abs(tot[1:-1])), np.diff(tot)
        assert abs(tot[-1] - tot[-2]) <= 1e-8 * abs(tot[-1])
        # -- and the first iterations of the same run in exact mode: the same trajectory
        ex.set_params(0.1, 0.2, 0.1)
        ex.init_emission()
        tot_e = []
        ex.EM(min_iters=3, max_iters=3, callback=lambda it, h: tot_e.append(h.tot_lkl))
        np.testing.assert_allclose(tot[:3], tot_e, rtol=1e-9)
        print(f"100 x 100k: {len(tot)} fast-mode EM iterations in {dt:.2f} s, total log-likelihood "
              f"{tot[0]:.3f} -> {tot[-1]:.3f}; first three iterations vs exact mode "
              f"{np.max(np.abs(tot[:3] - np.array(tot_e)) / np.abs(tot_e)):.1e} relative")


def _cli(args, tag):
    t0 = time.time()
    r = subprocess.run([cli_util.BINARY] + [str(a) for a in args], capture_output=True, text=True)
    assert r.returncode == 0, (tag, r.stdout[-1500:], r.stderr[-1500:])
    return time.time() - t0


def test_config3_encoding_bgzf_beagle_1000_x_50k_through_the_host(pkg, tmp_path):
    """configs[2]'s input encoding at 1000 x 50 000 (1.35 GB of text as BGZF) through the C++
    host, `--mode fast`: one handle against two site-shard handles (`--n_gpus 2 --devices 0,0`:
    the same Viterbi paths, total log-likelihood to 1e-9, printed frequencies and posteriors to
    the last printed digit up to the shards' last-bit differences), and against `--mode exact`
    (paths: the decoding runs the exact kernel on parameters that differ by the optimizer's
    spread)."""
    I, S = 1000, 50_000
    tmp = str(tmp_path)
    gen = os.path.join(tmp, "make_beagle")
    subprocess.run(["g++", "-O2", "-fopenmp", os.path.join(ROOT, "tools", "make_beagle.cpp"), "-o", gen,
                    "-lz"], check=True)
    subprocess.run([gen, str(I), str(S), os.path.join(tmp, "sim"), "11"], check=True, capture_output=True)
    gz = os.path.join(tmp, "sim.beagle.gz")
    assert open(gz, "rb").read(16)[12:14] == b"BC"            # a BGZF file
    base = ["--geno", gz, "--lkl", "--pos", os.path.join(tmp, "sim.pos.gz"), "--n_ind", I, "--n_sites", S,
            "--freq", 0.1, "--indF", "0.1,0.2", "--min_iters", 4, "--max_iters", 5, "--verbose", 0,
            "--seed", 1]
    t = {}
    t["fast1"] = _cli(base + ["--mode", "fast", "--out", os.path.join(tmp, "f1")], "fast1")
    t["fast2"] = _cli(base + ["--mode", "fast", "--n_gpus", 2, "--devices", "0,0",
                              "--out", os.path.join(tmp, "f2")], "fast2")
    t["exact"] = _cli(base + ["--mode", "exact", "--out", os.path.join(tmp, "ex")], "exact")

    def read(prefix):
        a = open(os.path.join(tmp, prefix + ".indF")).read().split("\n")
        lkl = float(a[0])
        # (an indF below 1e-5 or above 1 - 1e-5 prints as 0 / 1 with alpha "NA", EM.cpp:306-312)
        par = np.array([[np.nan if x == "NA" else float(x) for x in l.split("\t")] for l in a[1:1 + I]])
        freq = np.array([float(x) for x in a[1 + I:1 + I + S]])
        with open(os.path.join(tmp, prefix + ".ibd"), "rb") as fh:
            fh.readline()
            paths = [fh.readline() for _ in range(I)]
            rest = fh.read()
        return lkl, par, freq, paths, rest

    l1, p1, f1, path1, post1 = read("f1")
    l2, p2, f2, path2, post2 = read("f2")
    le, pe, fe, pathe, _ = read("ex")
    # -- one handle against two site shards
    assert abs(l1 - l2) <= 1e-9 * abs(l1)
    assert path1 == path2
    assert np.abs(f1 - f2).max() <= 1.1e-6
    assert np.array_equal(np.isnan(p1), np.isnan(p2)) and np.nanmax(np.abs(p1 - p2), initial=0) <= 1e-5
    same_bytes = post1 == post2
    if not same_bytes:      # printed posteriors: "%f" values, a last-bit difference may move a digit
        a = np.array(post1.split(), dtype=np.float64)
        b = np.array(post2.split(), dtype=np.float64)
        assert a.shape == b.shape and np.abs(a - b).max() <= 1.1e-6
        assert np.count_nonzero(a != b) <= 1e-4 * a.size
    # -- fast against exact mode
    assert abs(l1 - le) <= 1e-9 * abs(le)
    assert np.abs(f1 - fe).max() <= 1e-5
    cells = sum(len(x) for x in pathe)
    diff = sum(sum(c1 != c2 for c1, c2 in zip(x, y)) for x, y in zip(path1, pathe) if x != y)
    assert diff <= 1e-5 * cells, diff
    print(f"1000 x 50k BGZF BEAGLE through the host: fast 1 handle {t['fast1']:.1f} s, 2 handles "
          f"{t['fast2']:.1f} s, exact {t['exact']:.1f} s; posterior lines of 1 vs 2 handles "
          f"{'byte-identical' if same_bytes else 'equal to the printed digit'}; "
          f"{diff} of {cells} path cells differ between fast and exact mode")

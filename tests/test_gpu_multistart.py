"""Multi-start (ngsF-HMM.sh:77-101 runs 20 replicates from random
starts and keeps the best likelihood): replicas share the likelihoods on the device
(nghmm_create_replica) and run concurrently, one host thread and one HIP stream each.
A replica must behave exactly like an independent handle loaded with the same data:
bit-identical results in exact AND fast mode (both are deterministic), whatever runs next
to it."""
import os
import threading

import numpy as np
import pytest

import cli_util
from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _run(h, start, iters):
    h.set_params(*start)
    h.init_emission()
    lk = []
    for _ in range(iters):
        h.iter_EM()
        lk.append(h.ind_lkl.sum())
    return dict(lk=np.array(lk), indF=h.indF.copy(), alpha=h.alpha.copy(), freq=h.freq.copy(),
                post=h.marg_prob.copy(), path=h.viterbi())


@pytest.mark.parametrize("mode_name,packed", [("exact", False), ("fast", False), ("fast", True)])
def test_replicas_equal_independent_handles(pkg, mode_name, packed):
    I, S, R, iters = 24, 3000, 4, 3
    d = pkg.simulate.simulate(I, S, seed=77, n_chrom=3, missing_rate=0.03, indF="r")
    mode = (pkg.MODE_EXACT if mode_name == "exact" else pkg.MODE_FAST) | (pkg.GENO_PACKED if packed else 0)
    rng = np.random.default_rng(1)
    starts = [(rng.uniform(0.01, 0.9, I), rng.uniform(0.01, 2.0, I), rng.uniform(0.02, 0.48, S))
              for _ in range(R)]

    def load(h):
        h.load_raw(d.gl, d.pos_dist_mb, space=0, call_geno=packed)

    want = []
    for r in range(R):                      # R independent handles, one after the other
        with pkg.NgsFHMM(I, S, mode=mode) as h:
            load(h)
            want.append(_run(h, starts[r], iters))

    parent = pkg.NgsFHMM(I, S, mode=mode)
    load(parent)
    hs = [parent] + [parent.replica() for _ in range(R - 1)]
    got = [None] * R
    err = []

    def work(r):
        try:
            got[r] = _run(hs[r], starts[r], iters)
        except BaseException as e:          # noqa: BLE001
            err.append(e)
    th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    for r in range(R):
        for k in want[r]:
            assert np.array_equal(got[r][k], want[r][k]), (r, k)
    # the parent cannot go before its replicas, nor be reloaded under them
    with pytest.raises(pkg.NgsFHMMError):
        parent.close()
    with pytest.raises(pkg.NgsFHMMError):
        load(parent)
    for h in hs[1:]:
        h.close()
    parent.close()


def test_cli_n_starts_keeps_the_best_replicate(pkg, tmp_path):
    """--n_starts 3 --seed 5 with random starts = the three runs --seed 5, 6, 7; the output
    files are those of the run with the largest final log-likelihood (ngsF-HMM.sh:93-101)."""
    I, S = 10, 1500
    d = pkg.simulate.simulate(I, S, seed=12345, n_chrom=2)
    paths = cli_util.write_inputs(str(tmp_path), d, d.gl)
    common = ["--geno", paths["glf_bin"], "--loglkl", "--pos", paths["pos_gz"], "--n_ind", I,
              "--n_sites", S, "--freq", "r", "--indF", "r", "--min_iters", 3, "--max_iters", 6,
              "--mode", "fast", "--verbose", 1]
    singles = []
    for seed in (5, 6, 7):
        out = str(tmp_path / f"single_{seed}")
        cli_util.run_cli(common + ["--seed", seed, "--out", out])
        singles.append((float(open(out + ".indF").readline()), out))
    best = max(singles, key=lambda t: t[0])
    out = str(tmp_path / "multi")
    r = cli_util.run_cli(common + ["--seed", 5, "--n_starts", 3, "--keep_starts", "--out", out])
    assert "Best replicate: %d (seed %d)" % (singles.index(best) + 1, 5 + singles.index(best)) in r.stdout
    for ext in (".indF", ".ibd", ".geno"):
        assert open(out + ext, "rb").read() == open(best[1] + ext, "rb").read()
        for k, (_, single) in enumerate(singles):       # --keep_starts: every replicate's files
            assert open(f"{out}.REP_{k + 1:02d}{ext}", "rb").read() == open(single + ext, "rb").read()
    out2 = str(tmp_path / "multi2")
    cli_util.run_cli(common + ["--seed", 5, "--n_starts", 3, "--out", out2])
    for ext in (".indF", ".ibd", ".geno"):
        assert open(out2 + ext, "rb").read() == open(best[1] + ext, "rb").read()
    assert not os.path.exists(out2 + ".REP_01.indF")

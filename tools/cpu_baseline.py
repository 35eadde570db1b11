#!/usr/bin/env python3
"""CPU side of BASELINE.md section 4, on this box's host cores: the oracle (the CPU
restatement of the reference's path, libm build; test infrastructure, only timed here)
with the per-individual phases on 1 thread and on all cores -- the allele-frequency loop is
serial in the reference (EM.cpp:224) and stays serial here -- on configs[1] in full
(100 x 100k) and on a 1000 x 2000 slice of configs[2].  Prints site-individual updates/s.

  python tools/cpu_baseline.py [--quick]
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("ngsf-hmm_amd")
import orclib  # noqa: E402


def run(n_ind, n_sites, threads, iters):
    d = pkg.simulate.simulate(n_ind, n_sites, seed=777)
    gl = pkg.simulate.normalise_log_gl(d.gl)
    em = orclib.OracleEM(orclib.Oracle("libm"), gl, d.pos_dist_mb)
    em.set_params(0.1, 0.2, 0.1)
    em.init_emission()
    t0 = time.time()
    for _ in range(iters):
        assert em.iterate(n_threads=threads) == 0
    dt = time.time() - t0
    print(f"{n_ind} x {n_sites}, {threads} thread(s), {iters} EM iteration(s): {dt:.1f} s = "
          f"{n_ind * n_sites * iters / dt:.3g} site-individual updates/s", flush=True)


def main():
    quick = "--quick" in sys.argv
    cores = os.cpu_count() or 1
    print(f"host: {cores} logical cores")
    if quick:
        run(100, 4000, 1, 1)
        run(100, 4000, min(cores, 100), 2)
        return
    run(100, 100_000, min(cores, 100), 1)      # configs[1] in full, all cores
    run(100, 10_000, 1, 1)                      # a tenth of it on one thread
    run(1000, 2000, min(cores, 1000), 1)        # a slice of configs[2], all cores


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""End-to-end run of the C++ host on BASELINE.json configs[1]: 100 individuals x 100k
sites, binary log-GL doubles, --freq_est 1, start values of examples/test.sh.  Prints the
wall time of the whole command (file reading, normalisation, EM, Viterbi, writing) for
--mode fast and --mode exact, and how far the two modes' outputs are apart.

  python tools/cli_c2.py [n_ind n_sites]      (needs an MI355X)
"""
import importlib
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ngsf-hmm_amd")
BINARY = os.path.join(ROOT, "ngsf-hmm_amd", "ngsF-HMM")


def main():
    I = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    tmp = tempfile.mkdtemp(prefix="nghmm_c2_")
    t0 = time.time()
    d = pkg.simulate.simulate(I, S, seed=12345)
    d.gl.astype("<f8").tofile(os.path.join(tmp, "sim.glf"))
    with open(os.path.join(tmp, "sim.pos"), "w") as fh:
        for s in range(S):
            fh.write(f"chr{int(d.chrom[s])}\t{int(d.pos[s])}\n")
    print(f"simulated and wrote {I} x {S} in {time.time() - t0:.1f} s "
          f"({os.path.getsize(os.path.join(tmp, 'sim.glf')) / 1e6:.0f} MB of GL)")
    res = {}
    for mode, iters in (("fast", 20), ("exact", 3)):
        out = os.path.join(tmp, "out_" + mode)
        cmd = [BINARY, "--geno", os.path.join(tmp, "sim.glf"), "--loglkl", "--pos",
               os.path.join(tmp, "sim.pos"), "--n_ind", str(I), "--n_sites", str(S), "--freq", "0.1",
               "--indF", "0.1,0.2", "--freq_est", "1", "--min_iters", str(iters - 1), "--max_iters",
               str(iters), "--out", out, "--mode", mode, "--n_threads", "16", "--verbose", "1"]
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True)
        dt = time.time() - t0
        if r.returncode != 0:
            print(mode, "FAILED", r.stdout[-500:], r.stderr[-500:])
            continue
        n_it = r.stdout.count("Iteration ")
        lines = open(out + ".indF").read().split("\n")
        res[mode] = (float(lines[0]), np.array([float(x.split("\t")[0]) for x in lines[1:1 + I]]))
        print(f"--mode {mode}: {n_it} EM iterations, whole command {dt:.2f} s "
              f"(.ibd {os.path.getsize(out + '.ibd') / 1e6:.0f} MB)")
    if len(res) == 2:
        print("tot_lkl after 20 fast iterations %.4f, after 3 exact iterations %.4f"
              % (res["fast"][0], res["exact"][0]))


if __name__ == "__main__":
    main()

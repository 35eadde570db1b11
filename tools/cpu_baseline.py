#!/usr/bin/env python3
"""CPU side of BASELINE.md section 4, on this box's host cores: the oracle (the CPU
restatement of the reference's path, libm build; test infrastructure, only timed here) on
BASELINE.json configs[1] in full (100 x 100k) and on a 1000 x 10k slice of configs[2] (cost is
linear in individuals x sites), with the per-individual phases on all cores and on one thread
-- the allele-frequency loop is serial in the reference (EM.cpp:224) and stays serial -- and,
labelled as such, the "improved CPU" with that loop threaded over sites too.  Timed region: EM
iterations only.  Prints one JSON object (committed as profiles/r04_cpu_baseline.json, which
bench.py's cpu_baseline cites).

  python tools/cpu_baseline.py [--quick]        (about four minutes on a 256-thread host)
"""
import importlib
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("ngsf-hmm_amd")
import orclib  # noqa: E402


def run(name, n_ind, n_sites, threads, iters, thread_freq=False):
    d = pkg.simulate.simulate(n_ind, n_sites, seed=777)
    gl = pkg.simulate.normalise_log_gl(d.gl)
    em = orclib.OracleEM(orclib.Oracle("libm"), gl, d.pos_dist_mb)
    em.set_params(0.1, 0.2, 0.1)
    em.init_emission()
    t0 = time.time()
    for _ in range(iters):
        assert em.iterate(n_threads=threads, thread_freq=thread_freq) == 0
    dt = time.time() - t0
    res = {"name": name, "n_ind": n_ind, "n_sites": n_sites, "threads": threads, "em_iterations": iters,
           "seconds": round(dt, 2), "site_ind_updates_per_s": n_ind * n_sites * iters / dt,
           "freq_loop": "threaded over sites (improved CPU: not the reference's behaviour)"
                        if thread_freq else "serial, as in the reference (EM.cpp:224)"}
    sys.stderr.write(json.dumps(res) + "\n")
    em.close()
    return res


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    quick = "--quick" in sys.argv
    cores = os.cpu_count() or 1
    runs = []
    if quick:
        runs.append(run("quick sample, 1 thread", 100, 4000, 1, 1))
        runs.append(run("quick sample, all cores", 100, 4000, min(cores, 100), 2))
    else:
        runs.append(run("configs[1] in full, all cores", 100, 100_000, min(cores, 100), 1))
        runs.append(run("configs[2] slice 1000 x 10k, all cores", 1000, 10_000, min(cores, 1000), 1))
        runs.append(run("a tenth of configs[1], 1 thread", 100, 10_000, 1, 1))
        runs.append(run("configs[1] in full, improved CPU", 100, 100_000, min(cores, 100), 1, True))
        runs.append(run("configs[2] slice 1000 x 10k, improved CPU", 1000, 10_000, cores, 1, True))
    print(json.dumps({"host": {"logical_cores": cores, "cpu": cpu_model()},
                      "what": "oracle (libm build = the reference's arithmetic), EM iterations only; "
                              "one run each; configs[2] at full size = the slice's rate (cost is "
                              "linear in individuals x sites)",
                      "runs": runs}, indent=1))


if __name__ == "__main__":
    main()

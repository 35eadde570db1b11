#!/usr/bin/env python3
"""CPU side of BASELINE.md section 4, on the GPU box's host cores: the oracle (the CPU restatement
of the reference's path, libm build; test infrastructure, only TIMED here) on the benchmark's own
data set (simulate.IndexedSim, seed 12345) --

  * BASELINE.json configs[1] in full (100 x 100k) and a 1000 x 10k slice of configs[2] (cost is
    linear in individuals x sites: the full size is the slice's rate, stated as extrapolated);
  * `--n_threads` = all host cores for the per-individual phases with the allele-frequency loop
    SERIAL as in the reference (EM.cpp:224), one thread (on a tenth of configs[1]), and, labelled,
    the "improved CPU" with the frequency loop threaded over sites too;
  * in the regime the GPU number stands in: started from the state after 5 GPU-computed EM
    iterations (the optimizer in its 4-round steady state, which bench.py's timed iterations are
    in) -- and, beside it, from the cold start (--freq 0.1 --indF 0.1,0.2: 18, 11, 6 rounds);
  * median of >= 3 runs of >= 3 iterations each (timed region: EM iterations only).

Prints one JSON object (committed as profiles/rNN_cpu_baseline.json, which bench.py cites).

  python tools/cpu_baseline.py [--part a|b|all] [--runs 3] [--iters 3]
      part a: the reference's behaviour on all cores (the long one: ~12 min on a 256-thread host)
      part b: one thread, improved CPU, cold starts
"""
import argparse
import importlib
import json
import os
import platform
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("ngsf-hmm_amd")
import orclib  # noqa: E402

WARM_ITERS = 5


def data(i_tot, s_tot, inds, sites):
    """The slice of the benchmark's data set on the host, and the parameters after WARM_ITERS
    fast-mode EM iterations on the GPU."""
    import torch
    dev = torch.device("cuda", 0)
    sim = pkg.simulate.IndexedSim(i_tot, s_tot, dev, seed=12345)
    gl_d, pos_d = sim.gl(inds, sites), sim.pos_dist(*sites)
    torch.cuda.synchronize()
    I, S = inds[1] - inds[0], sites[1] - sites[0]
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as h:
        h.load_device(gl_d.data_ptr(), pos_d.data_ptr())
        h.set_params(0.1, 0.2, 0.1)
        h.init_emission()
        rounds = [h.iter_EM().rounds for _ in range(WARM_ITERS)]
        warm = (h.indF.copy(), h.alpha.copy(), h.freq.copy())
    return gl_d.cpu().numpy(), pos_d.cpu().numpy(), warm, rounds


def timed(name, gl, pos, start, threads, thread_freq, runs, iters, start_name):
    orc = orclib.Oracle("libm")
    S, I = gl.shape[0], gl.shape[1]
    secs, passes = [], None
    for r in range(runs):
        em = orclib.OracleEM(orc, gl, pos)
        em.set_params(*start)
        assert em.init_emission() == 0
        t0 = time.time()
        for k in range(iters):
            assert em.iterate(1, False, False, threads, thread_freq) == 0
            # (a line per iteration: the GPU box ends a command that stays silent for 7 minutes)
            sys.stderr.write(f"  {name}: run {r + 1}/{runs}, iteration {k + 1}/{iters} at {time.time() - t0:.1f} s\n")
            sys.stderr.flush()
        secs.append(time.time() - t0)
        passes = em.lkl_calls / (I * iters)
        em.close()
    med = statistics.median(secs)
    res = {"name": name, "n_ind": I, "n_sites": S, "threads": threads, "start": start_name,
           "runs": runs, "em_iterations_per_run": iters, "seconds_per_run": [round(x, 2) for x in secs],
           "median_seconds_per_iteration": med / iters,
           "site_ind_updates_per_s": I * S * iters / med, "em_iterations_per_s": iters / med,
           "forward_passes_per_ind_iter": passes,
           "freq_loop": "threaded over sites (improved CPU: not the reference's behaviour)"
                        if thread_freq else "serial, as in the reference (EM.cpp:224)"}
    sys.stderr.write(json.dumps(res) + "\n")
    sys.stderr.flush()
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--part", default="all", choices=["a", "b", "all"])
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--quick", action="store_true", help="a small sample (smoke test of this tool)")
    a = ap.parse_args()
    cores = os.cpu_count() or 1
    cold = (0.1, 0.2, 0.1)
    runs = []
    if a.quick:
        gl, pos, warm, rd = data(100, 100_000, (0, 100), (0, 4000))
        runs.append(timed("quick sample, all cores, warm", gl, pos, warm, min(cores, 100), False, 2, 2, "warm"))
        runs.append(timed("quick sample, 1 thread, cold", gl, pos, cold, 1, False, 1, 1, "cold"))
        warm_rounds = {"quick": rd}
    else:
        c2 = data(100, 100_000, (0, 100), (0, 100_000))
        sl = data(1000, 1_000_000, (0, 1000), (0, 10_000))
        warm_rounds = {"configs[1]": c2[3], "configs[2] slice": sl[3]}
        warm_name = f"the state after {WARM_ITERS} GPU-computed EM iterations"
        if a.part in ("a", "all"):
            runs.append(timed("configs[1] in full, all cores", c2[0], c2[1], c2[2], min(cores, 100), False,
                              a.runs, a.iters, warm_name))
            runs.append(timed("configs[2] slice 1000 x 10k, all cores", sl[0], sl[1], sl[2], min(cores, 1000),
                              False, a.runs, a.iters, warm_name))
        if a.part in ("b", "all"):
            tenth = (c2[0][:10_000], c2[1][:10_000], (c2[2][0], c2[2][1], c2[2][2][:10_000]))
            runs.append(timed("a tenth of configs[1] (its first 10k sites), 1 thread", tenth[0], tenth[1], tenth[2],
                              1, False, a.runs, a.iters, warm_name))
            runs.append(timed("configs[1] in full, improved CPU", c2[0], c2[1], c2[2], min(cores, 100), True,
                              a.runs, a.iters, warm_name))
            runs.append(timed("configs[2] slice 1000 x 10k, improved CPU", sl[0], sl[1], sl[2], cores, True,
                              a.runs, a.iters, warm_name))
            # the cold start beside it: its three iterations take 18, 11 and 6 objective rounds,
            # one run each (the same computation every time: runs differ by timing noise only)
            runs.append(timed("configs[1] in full, improved CPU, cold start", c2[0], c2[1], cold, min(cores, 100),
                              True, 1, a.iters, "cold: --freq 0.1 --indF 0.1,0.2"))
            runs.append(timed("configs[2] slice 1000 x 10k, improved CPU, cold start", sl[0], sl[1], cold, cores,
                              True, 1, a.iters, "cold: --freq 0.1 --indF 0.1,0.2"))
            runs.append(timed("configs[1] in full, all cores, cold start", c2[0], c2[1], cold, min(cores, 100),
                              False, 1, 1, "cold: --freq 0.1 --indF 0.1,0.2 (first iteration only)"))
    print(json.dumps({"host": {"logical_cores": cores, "cpu": cpu_model()}, "part": a.part,
                      "what": "oracle (libm build = the reference's arithmetic) on the benchmark's data set "
                              "(IndexedSim seed 12345), EM iterations only, median over the runs; configs[2] "
                              "at full size = the slice's rate (extrapolated: cost is linear in individuals "
                              "x sites)",
                      "gpu_rounds_of_the_warm_up": warm_rounds,
                      "runs": runs}, indent=1))


if __name__ == "__main__":
    main()

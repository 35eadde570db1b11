#!/usr/bin/env python3
"""Benchmark of the EM hot path on MI355X.

A "step" is one EM iteration (iter_EM, EM.cpp:139-289) over one synthetic data set:
E-step, per-individual L-BFGS-B M-step for (indF, alpha), per-site allele-frequency
EM + emission refresh.  Default workload = BASELINE.json configs[2]: 1000 individuals
x 1,000,000 sites, --freq_est 1 (the reference aborts on --freq_est 2), starting
values of examples/test.sh "normal" (--freq 0.1 --indF 0.1,0.2), inputs resident in
HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c5|c5share|tiny|tinycg]
                   [--mode fast|exact]
                   [--scaling strong|weak]

N > 1 = one rank per GPU over RCCL.  Started by torch.distributed.run (RANK / WORLD_SIZE in
the environment) the process IS a rank; started plainly (`python bench.py --gpus N`) it
launches the N ranks itself as a child `python -m torch.distributed.run ...` BEFORE any
GPU call and relays the child's JSON line and exit code.  A rank exits non-zero when the
process group's size is not --gpus.

--scaling strong (the default; BASELINE.json configs[3], the 1000 x 1M job over N GPUs).  What a
rank holds (--shard):
  sites (fast mode's default)  all individuals for a contiguous range of S / N sites.  A run of
      sites is a product of 2x2 operators, so the ranges owe each other six doubles per
      individual and objective point: one all-gather of ~200 kB per objective round; the
      allele-frequency step has every individual of its sites at hand and exchanges nothing.
  individuals (exact mode; what BASELINE.json's wording describes)  1000 / N individuals for
      all sites, the frequency step on S / N sites x all individuals: every posterior moves
      once per iteration (all-to-all, 8 GB at 1000 x 1M -- over ONE xGMI link between two
      GPUs), frequencies by an all-gather.
  DESIGN.md section 6 has the link arithmetic and one rank's measured compute for both.
--scaling weak: every rank the workload's full number of sites (or individuals).
--emulate_ranks V [--emulate_rccl]: N = 1 only, one rank's compute of a V-rank run.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "c3": dict(n_ind=1000, n_sites=1_000_000,
               name="1000 ind x 1M sites, synthetic log-GL (ngsF-HMMsim model), --freq_est 1, "
                    "--freq 0.1 --indF 0.1,0.2"),
    "c2": dict(n_ind=100, n_sites=100_000,
               name="100 ind x 100k sites, synthetic log-GL (ngsF-HMMsim model), --freq_est 1, "
                    "--freq 0.1 --indF 0.1,0.2"),
    "tiny": dict(n_ind=64, n_sites=20_000, name="64 ind x 20k sites (smoke)"),
    # ---- the same sizes off examples/test.sh's operating point (F 0.5, alpha 0.01, freq 0.2, depth 2):
    # the simulator's `r` options (scripts/ngsF-HMMsim.R:108-110,127-129,146-148: indF, alpha ~ U(0,1)
    # per individual, freq ~ U(0,1) per site), its default depth 5 (:77) or test.sh's 2, 2 % of the
    # cells without a read; a transition rate far above the small-alpha kernels' range
    "c3r": dict(n_ind=1000, n_sites=1_000_000,
                sim=dict(indF="r", alpha="r", freq="r", depth=5.0, missing_rate=0.02),
                name="1000 ind x 1M sites, --indF r --alpha r --freq r (U(0,1) each), depth 5, 2 % missing "
                     "cells, --freq_est 1, --freq 0.1 --indF 0.1,0.2"),
    "c3r2": dict(n_ind=1000, n_sites=1_000_000,
                 sim=dict(indF="r", alpha="r", freq="r", depth=2.0, missing_rate=0.02),
                 name="1000 ind x 1M sites, --indF r --alpha r --freq r (U(0,1) each), depth 2, 2 % missing "
                      "cells, --freq_est 1, --freq 0.1 --indF 0.1,0.2"),
    "c2r": dict(n_ind=100, n_sites=100_000,
                sim=dict(indF="r", alpha="r", freq="r", depth=5.0, missing_rate=0.02),
                name="100 ind x 100k sites, --indF r --alpha r --freq r (U(0,1) each), depth 5, 2 % missing "
                     "cells, --freq_est 1, --freq 0.1 --indF 0.1,0.2"),
    "c3hi": dict(n_ind=1000, n_sites=1_000_000, sim=dict(alpha=2.0),
                 name="1000 ind x 1M sites, true alpha 2 (alpha d_max > 2^-6: general-exp objective kernels in "
                      "the steady state), F 0.5, freq 0.2, depth 2, --freq_est 1, --freq 0.1 --indF 0.1,0.2"),
    # likelihood cohorts above 1024 individuals: est_maf on several waves per site
    "c2k": dict(n_ind=2000, n_sites=200_000,
                name="2000 ind x 200k sites, synthetic log-GL (ngsF-HMMsim model), --freq_est 1, "
                     "--freq 0.1 --indF 0.1,0.2"),
    "c5k": dict(n_ind=5000, n_sites=100_000,
                name="5000 ind x 100k sites, synthetic log-GL (ngsF-HMMsim model), --freq_est 1, "
                     "--freq 0.1 --indF 0.1,0.2"),
    # BASELINE.json configs[4]: needs 8 GPUs (625 individuals x 5M sites each, 21 B per cell
    # packed); c5share is exactly one GPU's share of it, without the exchange
    "c5": dict(n_ind=5000, n_sites=5_000_000, n_chrom=25, call_geno=True,
               name="5000 ind x 5M sites, 25 chromosomes, --call_geno (2-bit packed), "
                    "synthetic log-GL, --freq_est 1, --freq 0.1 --indF 0.1,0.2"),
    "c5share": dict(n_ind=625, n_sites=5_000_000, n_chrom=25, call_geno=True,
                    name="625 ind x 5M sites (one GPU's share of 5000 x 5M over 8), 25 chromosomes, "
                         "--call_geno (2-bit packed), --freq_est 1, --freq 0.1 --indF 0.1,0.2"),
    "tinycg": dict(n_ind=64, n_sites=20_000, n_chrom=4, call_geno=True,
                   name="64 ind x 20k sites, 4 chromosomes, --call_geno (2-bit packed) (smoke)"),
}

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)


FP64_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4   # wave-instructions/s: 256 CUs x 4 SIMDs, an FP64 (or any
                                        # VALU) instruction of a 64-wide wave takes 4 cycles;
                                        # x 64 lanes x 2 flop = the 78.6 TFLOP/s FP64 vector peak

# The kernels of one fast-mode EM iteration and what bounds each.  Their
# instruction counts per site come from the device assembly of THIS build: profiles/rNN_isa_summary.txt
# (tools/isa_summary.sh: `hipcc -S` + tools/isa_report.py), parsed by isa_counts() below and used
# only when the file carries the build id of the sources this run loads (profiles/build_id.py).
KERNELS = {
    "lkl_later_rounds": dict(
        bound="fp64_valu", isa="later objective rounds",
        kernel="k_fast_lkl_fd<2,2,true,false,SRC_PLAIN,2>",
        note="objective rounds 2.. of the L-BFGS-B M-step: 5 probe points share one pass over "
             "the 8 B emission ratio of every still-active individual; kappa form (operators kept "
             "divided by exp(-alpha d): 4 instructions per row instead of 5), 74 FP64 instructions "
             "per site (c form: 87), 4 waves per SIMD; the chip runs it at ~1.7 GHz (power limit)"),
    "lkl_first_round": dict(
        bound="hbm", isa="fresh forward walk",
        kernel="k_fast_lkl_fd<2,2,true,true,SRC_FRESH,2>",
        note="round 1 = E-step's forward walk = emission refresh: reads the 16 B relative "
             "likelihoods, writes the 8 B emission ratio and 4 B of checkpoints; with the walk in "
             "the kappa form (113 VALU per site, 128 in the c form) the kernel sits on its memory "
             "side: 0.68 of the HBM roof = 87 % of the 6.29 TB/s a copy reaches; 3 waves per SIMD"),
    "est_maf": dict(
        bound="fp64_valu", isa="est_maf: k_fast_estmaf<16, 64, true>",
        kernel="k_fast_estmaf<16,64,true> + _interp + _resume",
        note="the reference's ~100 passes per site: 3 evaluated over all individuals, 12 "
             "Chebyshev nodes of a checked interpolant, the rest on the interpolant; 256 VGPRs, "
             "2 waves per SIMD, SQ_ACTIVE_INST_VALU 0.43 per wave; neither roof is reached: a site's "
             "load phase and its serial passes are only partly under the other wave's arithmetic"),
    "backward_sweep": dict(
        bound="hbm", kernel="k_fast_bounds + k_fast_bwd_recompute8",
        note="boundary vectors + backward sweep with block-wise forward recomputation: 8 B "
             "emission ratio + 4 B checkpoints read, 8 B posteriors written per site and "
             "individual; a copy kernel reaches 6.29 TB/s on this chip"),
}


def this_build_id():
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    from build_id import build_id
    return build_id()


def isa_counts():
    """(counts, source, reason): per kernel of KERNELS the (instructions, VALU, FP64) of every basic
    block of >= 100 instructions, in file order, from the latest profiles/rNN_isa_summary.txt --
    or (None, file, why) when there is none for THIS build."""
    import re
    path = _latest_profile("isa_summary.txt")
    if not path:
        return None, None, "no profiles/rNN_isa_summary.txt"
    text = open(os.path.join(ROOT, path)).read().split("\n")
    m = re.match(r"# build_id: (\w+)", text[0]) if text else None
    if not m:
        return None, path, f"{path} carries no build id (made before round 5)"
    if m.group(1) != this_build_id():
        return None, path, (f"{path} is of build {m.group(1)}, the sources here are {this_build_id()}: "
                            "re-run tools/isa_summary.sh")
    out, cur = {}, None
    for line in text:
        if line.startswith("== "):
            cur = next((k for k, v in KERNELS.items() if v.get("isa") and line[3:].startswith(v["isa"])), None)
            if cur:
                out[cur] = []
        b = re.match(r"\s+block \S+: (\d+) instructions = (\d+) VALU \((\d+) FP64\)", line)
        if b and cur and int(b.group(1)) >= 100:
            out[cur].append(tuple(int(x) for x in b.groups()))
        if cur and "est_maf model (" in line:
            out[cur + ":model"] = {n: (int(v), int(f)) for n, v, f in re.findall(r"(\w+) (\d+)/(\d+)", line)}
    return out, path, None


def walk_instr_per_site(blocks):
    """lkl kernels: the loop body (the largest block) covers 8 sites."""
    n, valu, fp64 = max(blocks)
    return valu / 8.0, fp64 / 8.0


def estmaf_instr_per_site(i_tot, model):
    """VALU / FP64 wave-instructions est_maf issues per site, from the parts tools/isa_report.py
    recognises in its assembly (`est_maf model`: set-up of the per-individual constants for 16
    individuals per lane; an exact pass = sums with their reduction + recursion; the interpolant's
    check, once; a node evaluation; the addition of the parked node sums): 3 exact passes and 12
    nodes per site, the per-individual parts scaled to the individuals per lane of the cohort.
    None without the model line.
    (Until round 5 the node loop was counted with the code it falls through to, 208 instructions
    for 139: 3 388 VALU per site where SQ_INSTS_VALU counts 2 830.)"""
    if not model:
        return None
    ni = min(16, -(-i_tot // 64))
    waves = max(1, -(-i_tot // 1024))
    sc = ni / 16.0
    out = []
    for k in (0, 1):
        g = lambda name: model[name][k]
        out.append(waves * (g("setup") * sc + 3 * (g("pass") * sc + g("recursion")) + g("check")
                            + 12 * g("node") * sc + g("node_tail")))
    return tuple(out)


def _latest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that has one (relative path), or None."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{suffix}")))
    return os.path.relpath(hits[-1], ROOT) if hits else None


PMC_SUMMARY = _latest_profile("pmc_summary.json")


def pmc_summary():
    """(summary, reason): the committed rocprofv3 --pmc passes of the default workload
    (profiles/collect.sh -> profiles/summarize_pmc.py: 2 x FETCH_SIZE + WRITE_SIZE, KiB units,
    gfx950 correction) -- if they are of THIS build (profiles/build_id.py), else (None, why)."""
    if not PMC_SUMMARY:
        return None, "no profiles/rNN_pmc_summary.json"
    summ = json.load(open(os.path.join(ROOT, PMC_SUMMARY)))
    have = summ.get("build_id")
    if have != this_build_id():
        return None, (f"{PMC_SUMMARY} is of build {have}, the sources here are {this_build_id()}: counters "
                      "of another build are not quoted (profiles/collect.sh)")
    return summ, None


def pmc_traffic(summ, I, C):
    """HBM bytes the PMC passes saw, per unit that does not depend on which iterations a pass
    happened to hold: a fresh first round per launch (all individuals), a later round per
    individual in it, est_maf and the backward sweep per EM iteration."""
    if not summ:
        return {}
    fresh_b = fresh_n = plain_b = plain_ind = 0.0
    for k, d in summ.items():
        if not k.startswith("k_fast_lkl_fd<") or "hbm_bytes_per_launch" not in d:
            continue
        n = d["launches_fetch_pass"]
        targs = [t.strip() for t in k[k.index("<") + 1:k.rindex(">")].split(",")]
        emit, src = targs[3], targs[4]      # <NF, NA, SMALL, EMIT, SRC, XDEG>: SRC 0 = stored emissions
        inds = n * d["avg_grid_threads"] / 64.0 / C
        if src != "0":
            fresh_b += d["hbm_bytes_per_launch"] * n
            fresh_n += inds / I
        elif emit == "false":
            plain_b += d["hbm_bytes_per_launch"] * n
            plain_ind += inds
    out = {}
    if fresh_n:
        out["lkl_first_round_per_launch"] = fresh_b / fresh_n
    if plain_ind:
        out["lkl_later_rounds_per_individual"] = plain_b / plain_ind
    fam = summ.get("families", {})
    for key, name in (("est_maf", "est_maf_per_iteration"), ("forward", "backward_sweep_per_iteration")):
        if fam.get(key, {}).get("hbm_bytes_per_em_iteration"):
            out[name] = fam[key]["hbm_bytes_per_em_iteration"]
    return out


def make_sim(pkg, wl, n_ind, n_sites, device):
    """The workload's data set (simulate.IndexedSim, seed 12345): any slice of it is bit for bit the
    slice of the whole, so every part of a run -- the ranks of a job, the CPU sample, the exact
    mode's slice -- sees the same cells."""
    return pkg.simulate.IndexedSim(n_ind, n_sites, device, seed=12345, n_chrom=wl.get("n_chrom", 1),
                                   **wl.get("sim", {}))


def cpu_baseline(pkg, seconds_budget=20.0, full_c2=False, wl=None):
    """The oracle (libm build = the reference's arithmetic; its L-BFGS-B core is pinned
    bit for bit to the reference object) timed on this box's host cores on a bounded
    sample of the same workload: per-individual phases threaded like the reference's
    pool, the frequency loop serial as in the reference (EM.cpp:224).  The sample is a
    slice of the benchmark's own data set (100 individuals x its first 4000 sites), and
    the timed iterations start from the state after 5 GPU-computed EM iterations -- the
    optimizer's 4-round steady state, which the GPU's timed iterations are in (`cold_start`:
    the same from --freq 0.1 --indF 0.1,0.2, whose first iterations take 18, 11, 6 rounds).
    `improved_cpu`: the frequency loop threaded over sites too (what a
    maintainer could get from the CPU without changing the arithmetic).
    full_c2: additionally one EM iteration of BASELINE.json configs[1] (100 x 100k) in
    full, about a minute on a 256-thread host."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    import torch
    cores = os.cpu_count() or 1
    orc = orclib.Oracle("libm")
    dev = torch.device("cuda", 0)

    def sample(n_ind, n_sites, s_tot):
        sim = make_sim(pkg, {"sim": (wl or {}).get("sim", {})}, n_ind, s_tot, dev)
        gl_d, pos_d = sim.gl((0, n_ind), (0, n_sites)), sim.pos_dist(0, n_sites)
        torch.cuda.synchronize()
        with pkg.NgsFHMM(n_ind, n_sites, mode=pkg.MODE_FAST) as h:
            h.load_device(gl_d.data_ptr(), pos_d.data_ptr())
            h.set_params(0.1, 0.2, 0.1)
            h.init_emission()
            for _ in range(5):
                h.iter_EM()
            warm = (h.indF.copy(), h.alpha.copy(), h.freq.copy())
        return gl_d.cpu().numpy(), pos_d.cpu().numpy(), warm

    def timed(gl, pos, start, max_iters, budget, thread_freq):
        n_sites, n_ind = gl.shape[0], gl.shape[1]
        em = orclib.OracleEM(orc, gl, pos)
        em.set_params(*start)
        em.init_emission()
        thr = min(cores, n_sites if thread_freq else n_ind)
        t0 = time.time()
        iters = 0
        while iters < max_iters and (time.time() - t0) < budget:
            rc = em.iterate(n_threads=thr, thread_freq=thread_freq)
            assert rc == 0
            iters += 1
        dt = time.time() - t0
        res = (n_ind * n_sites * iters / dt, iters, dt, thr, em.lkl_calls / (n_ind * iters),
               em.maf_passes / (n_sites * iters))
        em.close()
        return res

    n_ind, n_sites = 100, 4000
    gl, pos, warm = sample(n_ind, n_sites, 100_000)
    cold = (0.1, 0.2, 0.1)
    v, iters, dt, thr, fw, mp = timed(gl, pos, warm, 3, seconds_budget, False)
    out = {
        "value": v,
        "unit": "site-ind updates/s",
        "cores": thr,
        "kind": "port",
        "sample": f"{n_ind} ind x the first {n_sites} sites of the 100 x 100k data set"
                  + (f" of this workload's regime ({(wl or {}).get('sim')})" if (wl or {}).get("sim") else "")
                  + f", {iters} EM iterations in "
                  f"{dt:.1f} s from the state after 5 GPU-computed iterations (the steady state the GPU "
                  f"is timed in), oracle libm build, per-individual phases on {thr} threads, "
                  f"allele-frequency loop serial as in the reference (EM.cpp:224)",
        "forward_passes_per_ind_iter": fw,
        "est_maf_passes_per_site": mp,
    }
    vc, itc, dtc, _, fwc, _ = timed(gl, pos, cold, 3, seconds_budget / 2, False)
    out["cold_start"] = {"value": vc, "unit": "site-ind updates/s", "cores": thr,
                         "forward_passes_per_ind_iter": fwc,
                         "sample": f"the same sample from --freq 0.1 --indF 0.1,0.2, {itc} iterations in {dtc:.1f} s"}
    v2, it2, dt2, thr2, _, _ = timed(gl, pos, warm, 3, seconds_budget / 2, True)
    out["improved_cpu"] = {
        "value": v2, "unit": "site-ind updates/s", "cores": thr2,
        "sample": f"same sample and start, {it2} iterations in {dt2:.1f} s, allele-frequency "
                  f"loop threaded over sites as well ({thr2} threads): not the reference's "
                  f"behaviour (its loop is serial, EM.cpp:224), same arithmetic"}
    # BASELINE.md section 4's full-size CPU numbers (configs[1] in full, a 1000 x 10k slice of
    # configs[2], one thread, the "improved CPU"; median of 3 runs of 3 iterations, warm and
    # cold) take many minutes: measured once per round on the GPU box's host with
    # tools/cpu_baseline.py and committed; this run's sample above is the live cross-check of
    # the same code on this host
    committed = _latest_profile("cpu_baseline.json")
    if committed:
        out["baseline_md_section4"] = dict(json.load(open(os.path.join(ROOT, committed))),
                                           source=f"{committed} (tools/cpu_baseline.py)")
    if full_c2:
        gl2, pos2, warm2 = sample(100, 100_000, 100_000)
        v3, it3, dt3, thr3, _, _ = timed(gl2, pos2, warm2, 1, 1e9, False)
        out["configs1_full"] = {
            "value": v3, "unit": "site-ind updates/s", "cores": thr3,
            "sample": f"BASELINE configs[1] in full: 100 ind x 100k sites, {it3} EM iteration in "
                      f"{dt3:.1f} s from the warm state, frequency loop serial as in the reference"}
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` without a torch.distributed environment: start the N rank
    processes as a CHILD (never exec: this process may not have touched the GPU, and it has
    not -- device_count() does not initialise it) and relay its output and exit code."""
    import socket
    import subprocess
    import torch
    n_dev = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.dry_launch:
        # the launcher, the rendezvous, the collectives' preflight and every rank's shard arithmetic
        # WITHOUT a GPU call: what an N-rank run does before it touches its card, rehearsed on the CPU
        env["NGHMM_BENCH_BACKEND"] = "gloo"
    elif n_dev < args.gpus and "NGHMM_BENCH_BACKEND" not in env:
        # fewer GPUs than ranks (a one-GPU box): functional run only -- every rank on cuda:0,
        # collectives through gloo staged on the host; the line says so.  At most six processes may
        # have one card open on the build's GPU boxes, and a launch of N ranks is N + 1 of them
        # (measured: six ranks are refused): four, as the tests use.
        if n_dev < 1 or args.gpus > 4:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible")
        env["NGHMM_BENCH_BACKEND"] = "gloo"
        env["NGHMM_BENCH_ONE_GPU"] = "1"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
           str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env)
    raise SystemExit(r.returncode)


PARITY = {
    "oracle": "partial: the L-BFGS-B core is pinned bit for bit to the reference's own object code; "
              "HMM.cpp / gen_func.cpp / EM.cpp cannot be compiled here (GSL absent, no stand-in "
              "written): restated, anchored by a binary128 model and path enumeration",
    "exact_mode": "bit-identical to the oracle's det build end to end (every array, Viterbi paths, "
                  "output files)",
    "per_call": "fast mode vs the oracle per call and teacher-forced per iteration: log-likelihoods 1e-12 and "
                "Viterbi paths identical at every size; frequencies 1e-9 relative when both sides are fed the same "
                "posteriors; POSTERIORS within 1e-9 only on chains up to ~10^4 sites (5.8e-10 there, 3.2e-8 at 10^5, "
                "2.4e-7 at 3 x 10^5) -- the oracle (the reference's log-space doubles) itself drifts from a binary128 "
                "evaluation with the chain length, to 7.5e-6 at the 10^6 sites this line is timed on, where fast mode stays within 2e-14 of binary128 (`by_chain_length` "
                "below: tests/test_gpu_baseline_oracle.py::test_fast_mode_against_binary128_by_chain_length).  At the "
                "benchmarked size the timed mode does NOT match the reference's posteriors to 1e-9; exact mode "
                "(bit-identical to the oracle, ~500 x slower) does",
    "baseline_sizes_vs_oracle": "tests/test_gpu_baseline_oracle.py (-m gpu): configs[1] in full (100 x 100k "
                                "of this data set) and a 1000 x 10k slice of configs[2] -- exact mode "
                                "BITWISE against the oracle over a whole EM iteration + Viterbi; fast mode per "
                                "call: log-likelihoods 1e-12, posteriors within the ORACLE's own distance "
                                "from binary128 (measured there), frequencies 1e-12 when the oracle's est_maf "
                                "is fed the GPU's posteriors",
    "device_optimizer": "tests/test_gpu_devbfgs.py: the L-BFGS-B machines advanced on the device "
                        "(kernels_bfgs.hip) against the same machines on the host -- every array and the "
                        "optimizer's accounting bit for bit; on the CPU the device's solver type against "
                        "class Lbfgsb, which is pinned to the reference's object code",
    "end_to_end_indF": "3.5e-5 max (median 3.5e-7) vs exact mode after 25 iterations at 200 x 50k; the "
                       "reference differs from itself by 1e-5 under another compiler flag (SURVEY "
                       "finding 4): the finite-difference L-BFGS-B amplifies last-bit differences, so "
                       "1e-9 end to end holds for the likelihood trajectory, not for indF / alpha",
}

CHECK_REF = os.path.join(ROOT, "profiles", "check_n1.json")


def parity_object():
    """PARITY + the measured pairs of tests/test_gpu_baseline_oracle.py's 10^6-site test (committed
    as profiles/rNN_parity_1M.json by the round's collection): fast mode and the oracle (the
    reference's log-space doubles) both against the binary128 anchor at the benchmarked chain length."""
    out = dict(PARITY)
    path = _latest_profile("parity_by_length.json")
    if path:
        m = json.load(open(os.path.join(ROOT, path)))
        out["by_chain_length"] = {
            "source": path,
            "posteriors_abs_vs_binary128": {S: {"fast": v["post_fast"], "oracle": v["post_oracle"],
                                                "fast_vs_oracle": v["post_fast_vs_oracle"]}
                                            for S, v in m["by_length"].items()},
            "log_likelihood_rel_vs_binary128": {S: {"fast": v["lkl_fast"], "oracle": v["lkl_oracle"]}
                                                for S, v in m["by_length"].items()},
            "fast_vs_oracle_posteriors_within_1e-9_up_to_S": m.get("fast_vs_oracle_posteriors_within_1e-9_up_to_S")}
    path = _latest_profile("parity_1M.json")
    if path:
        m = json.load(open(os.path.join(ROOT, path)))
        c = m["chains"]
        out["one_million_sites_vs_binary128"] = {
            "source": path + " (tests/test_gpu_baseline_oracle.py::test_one_million_site_chains_against_binary128)",
            "log_likelihood_rel": {"fast": max(v["lkl_fast"] for v in c.values()),
                                   "oracle": max(v["lkl_oracle"] for v in c.values())},
            "posteriors_abs": {"fast": max(v["post_fast"] for v in c.values()),
                               "oracle": max(v["post_oracle"] for v in c.values())},
            "est_maf_1000_individuals_rel": {"fast": m["est_maf_1000_individuals"]["freq_fast"],
                                             "oracle": m["est_maf_1000_individuals"]["freq_oracle"]},
            "reading": "fast mode's distance from exact mode at 1000 x 1M (posteriors 2e-5, frequencies 1e-5: "
                       "tests/test_gpu_fullsize.py) is the REFERENCE formulation's own rounding at 10^6 "
                       "sites, not fast mode's; est_maf is the one routine where the oracle is the more "
                       "accurate side (interpolated passes, checked to 1e-13)"}
    return out


class Ctx:
    """What every part of a run needs: arguments, modules, this process's place in the job."""


class ShardSetupRefused(RuntimeError):
    """build_run: the ranks have agreed (one all-reduce) that at least one of them could not set
    up its shard of a layout; nobody has entered the load's collectives."""


def shard_shapes(ctx, shard):
    """What this rank's handle holds under `shard` ("sites" | "individuals" | None = one GPU):
    the slice of the ONE data set IndexedSim(I_tot, S_job) describes.  The N-rank job processes
    the data set the N = 1 job processes, whatever the sharding (EM.cpp:151-161: results do not
    depend on the number of workers)."""
    args, wl, world, rank = ctx.args, ctx.wl, ctx.world, ctx.rank
    I, S = wl["n_ind"], wl["n_sites"]
    V = args.emulate_ranks
    strong = args.scaling == "strong"
    sh = {"by_sites": shard == "sites", "shard": shard}
    if shard == "sites":
        S_job = S if strong else S * world            # weak: the chain N times as long
        lo, hi = ctx.dd.site_ranges_ragged(S_job, world * V)[rank]
        sh.update(I=I, I_tot=I, S=hi - lo, S_job=S_job, ind_range=(0, I), site_range=(lo, hi))
    elif shard == "individuals":
        I_tot = I if strong else I * world
        if I_tot % (world * V) or S % (world * V):
            raise ValueError(f"individual shards: {I_tot} individuals / {S} sites do not divide by "
                             f"{world * V}")
        Il = I_tot // (world * V)
        sh.update(I=Il, I_tot=I_tot, S=S, S_job=S, ind_range=(rank * Il, (rank + 1) * Il),
                  site_range=(0, S))
    else:
        sh.update(I=I, I_tot=I, S=S, S_job=S, ind_range=(0, I), site_range=(0, S))
    per_cell = 24.0 if ctx.call_geno else 90.0           # 21 / 84 B per cell + exchange
    if sh["I"] * sh["S"] * per_cell > 270e9:
        raise ValueError(f"workload {args.workload}: {sh['I']} x {sh['S']} per GPU does not fit one "
                         f"MI355X (use more ranks: --gpus 8)")
    return sh


def build_run(ctx, shard, agree=None):
    """Create the rank's handle(s) for `shard` and load its slice of the synthetic data set,
    generated on the device (same data model as scripts/ngsF-HMMsim.R, every element a hash of
    its global indices: simulate.IndexedSim).  agree(error_or_None) -> bool: called once, between
    the rank-local part (handle, exchange buffers, the generated likelihoods: where memory runs
    out if it does) and the part that contains collectives (individual shards exchange their site
    shards while loading); the ranks go on only if it returns True."""
    torch, pkg, dd, args = ctx.torch, ctx.pkg, ctx.dd, ctx.args
    sh = shard_shapes(ctx, shard)
    V = args.emulate_ranks
    em, gl, sim, pos, err = None, None, None, None, None
    try:
        if sh["by_sites"]:
            em = dd.SiteShardedEM(pkg, sh["I"], sh["S_job"], device_index=ctx.local_rank, mode=ctx.mode,
                                  rank=ctx.rank, world=ctx.world, emulate_ranks=V,
                                  emulate_through_group=args.emulate_rccl)
            assert em.S_own == sh["S"]
        else:
            em = dd.ShardedEM(pkg, sh["I"], sh["S"], device_index=ctx.local_rank, mode=ctx.mode,
                              rank=ctx.rank, world=ctx.world, emulate_ranks=V)
        sim = make_sim(pkg, ctx.wl, sh["I_tot"], sh["S_job"], ctx.device)
        pos = sim.pos_dist(*sh["site_range"])
        if not ctx.call_geno:
            gl = sim.gl(sh["ind_range"], sh["site_range"])
            torch.cuda.synchronize()
    except Exception as e:   # noqa: BLE001
        if agree is None:
            raise
        err = f"{type(e).__name__}: {e}"
    if agree is not None and not agree(err):
        if em is not None:
            em.close()
        del gl, sim, pos
        torch.cuda.empty_cache()
        raise ShardSetupRefused(err or "another rank could not set up its shard")
    if ctx.call_geno:   # a block of sites at a time, called and packed on the way in
        em.load_chunks_device(pos, sim.chunks(sh["ind_range"], sh["site_range"], chunk_sites=50_000),
                              space=0, call_geno=True)
    else:
        em.load_device(gl, pos)
        del gl
    del sim, pos
    torch.cuda.empty_cache()
    sh["em"] = em
    return sh


def reset_params(em):
    """examples/test.sh "normal": --freq 0.1 --indF 0.1,0.2."""
    em.set_params(0.1, 0.2, 0.1)
    em.init_emission()


def barrier(ctx):
    ctx.torch.cuda.synchronize()
    if ctx.world > 1:
        ctx.dist.barrier()
    ctx.torch.cuda.synchronize()


def allreduce(ctx, values, op="sum"):
    """numpy float64 array reduced over the ranks (identity on one rank)."""
    import numpy as np
    a = np.ascontiguousarray(values, dtype=np.float64)
    if ctx.world == 1:
        return a
    torch, dist = ctx.torch, ctx.dist
    t = torch.from_numpy(a.copy())
    if dist.get_backend() == "nccl":
        t = t.to(ctx.device)
    dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX,
                           "min": dist.ReduceOp.MIN}[op])
    return t.cpu().numpy()


FAMILIES = ("emission", "forward", "backward", "lkl_batch", "est_maf", "lkl_first", "bfgs")


def timed_loop(ctx, run, steps, warmup, replicas=()):
    """W untimed iterations from the starting values, then EXACTLY K timed ones between
    barriers; MAX over ranks.  Every iteration's own wall time is kept as well (iter_EM returns
    after its stream has drained, so the clock reads add no synchronisation): the first
    iterations of a run are its cold start (L-BFGS-B needs 18, 13, 6 objective rounds there)."""
    import threading
    em = run["em"]

    def iterate_all():
        if not replicas:
            return em.iter_EM()
        th = [threading.Thread(target=h.iter_EM) for h in replicas]
        for t in th:
            t.start()
        st = em.iter_EM()
        for t in th:
            t.join()
        return st

    each_ms, each_rounds = [], []

    def one():
        t = time.perf_counter()
        st = iterate_all()
        each_ms.append((time.perf_counter() - t) * 1e3)
        each_rounds.append(int(st.rounds))
        if os.environ.get("NGHMM_DEBUG_CHECK"):
            import numpy as np
            lk = np.asarray(em.ind_lkl if getattr(em, "ind_lkl", None) is not None else em.hmm.ind_lkl)
            sys.stderr.write(f"[loop r{ctx.rank}] rounds {st.rounds} lkl sum {lk.sum()!r}\n")
        return st

    for _ in range(warmup):
        one()
    if os.environ.get("NGHMM_BENCH_FAIL_RANK") == str(ctx.rank) and ctx.world > 1:
        # (tests: one rank hits a fatal of the reference between two collectives -- the others
        # must end with its message, not wait in their next all-gather)
        with em.beacon.guard():
            raise ctx.pkg.NgsFHMMError(-3, "invalid MAF! (injected by NGHMM_BENCH_FAIL_RANK)")
    em.reset_timing()
    # which kernel versions the objective rounds take and how many sites of the frequency step leave
    # its common route (include/nghmm_debug.h): counted over the timed iterations only
    fast_handles = handles_of(em) if ctx.args.mode == "fast" else []
    for h in fast_handles:
        h.mode_counts(reset=True)
        h.estmaf_counts(reset=True)
    ex = getattr(em, "exchange", None)
    if ex is not None and hasattr(ex, "take_log"):
        ctx.torch.cuda.synchronize()
        ex.take_log()
    fam = {k: 0.0 for k in FAMILIES}
    launches = dict.fromkeys(fam, 0)
    rounds = points = ind_rounds = ref_calls = 0
    # Kernel-family times (HIP events around the families, nghmm_kernel_ms): exact mode always
    # records them; fast mode only with the switch `spans` -- the events are packets the queue
    # works through between two kernels, 0.05 ms per iteration (8 % of configs[1]'s) -- so the
    # timed iterations run without them and a few more iterations of the same steady state
    # behind the timed region collect them (scaled to K iterations; not part of `value`).
    # (--serial_kernels, what profiles/collect.sh traces: events in the timed iterations themselves,
    # so that the trace and the line speak of the same iterations)
    in_loop = ctx.args.mode != "fast" or ctx.args.serial_kernels
    barrier(ctx)
    t0 = time.perf_counter()
    # (site shards under nccl: a pair of events around every all-gather -- the LAST timed
    # iteration's calls are set side by side over the ranks, the others carry none)
    ex_timed = ex is not None and hasattr(ex, "timed")
    if ex_timed:
        ex.timed = False
    for i_step in range(steps):
        if ex_timed and i_step == steps - 1:
            ex.timed = True
        st = one()
        rounds += st.rounds
        points += st.points
        ind_rounds += st.ind_rounds
        ref_calls += st.ref_forward_calls
        if in_loop:
            for k in fam:
                ms, n = em.hmm.kernel_ms(k)
                fam[k] += ms
                launches[k] += n
    barrier(ctx)
    dt = time.perf_counter() - t0
    if ctx.world > 1:
        dt = float(allreduce(ctx, [dt], "max")[0])
    exchange_log = ex.take_log() if ex is not None and hasattr(ex, "take_log") else None
    if ex_timed:
        ex.timed = False
    mode_counts, estmaf_counts = {}, {}
    for h in fast_handles:
        for k, v in h.mode_counts().items():
            mode_counts[k] = mode_counts.get(k, 0) + v
        for k, v in h.estmaf_counts().items():
            estmaf_counts[k] = estmaf_counts.get(k, 0) + v
    if not in_loop:
        hs = handles_of(em) + list(replicas)
        n_sp = max(1, min(5, steps))
        saved_timing = dict(em.timing) if isinstance(getattr(em, "timing", None), dict) else None
        saved_ex = (ex.calls, ex.bytes, ex.host_ms) if ex is not None and hasattr(ex, "calls") else None
        for h in hs:
            h.set_switch("spans", 1)
        try:
            for _ in range(n_sp):
                iterate_all()
                for k in fam:
                    ms, n = em.hmm.kernel_ms(k)
                    fam[k] += ms * steps / n_sp
                    launches[k] += n * steps / n_sp
        finally:
            for h in hs:
                h.set_switch("spans", 0)
        # (the accounting of the timed region is what the line reports)
        if saved_timing is not None:
            em.timing.clear()
            em.timing.update(saved_timing)
        if saved_ex is not None:
            ex.calls, ex.bytes, ex.host_ms = saved_ex
            if hasattr(ex, "take_log"):
                ctx.torch.cuda.synchronize()
                ex.take_log()
    return dict(dt=dt, fam=fam, launches=launches, rounds=rounds, points=points, ind_rounds=ind_rounds,
                ref_calls=ref_calls, each_ms=each_ms, each_rounds=each_rounds, steps=steps, warmup=warmup,
                exchange_log=exchange_log, mode_counts=mode_counts, estmaf_counts=estmaf_counts)


def handles_of(em):
    hm = getattr(em, "hmm", em)
    return list(getattr(hm, "handles", [hm]))


def serial_kernel_loop(ctx, run, n, replicas=()):
    """n more iterations (continuing the timed loop's run, i.e. its steady state) with the
    library's switch `no_bg_stream`: the E-step's sweep and est_maf between the objective rounds
    on the ONE stream instead of next to them on a second.  In the timed loop kernels of the two
    streams share the chip, and a kernel's event span then holds its neighbours' work too; here
    they run one after the other, so a span is that kernel's own duration -- what a roofline
    fraction needs, and what rocprofv3 --kernel-trace of `bench.py --serial_kernels` shows.
    Not part of `value`."""
    em = run["em"]
    hs = handles_of(em) + list(replicas)
    for h in hs:
        h.set_switch("no_bg_stream", 1)
        h.set_switch("spans", 1)
    try:
        fam = {k: 0.0 for k in FAMILIES}
        launches = dict.fromkeys(fam, 0)
        rounds = ind_rounds = 0
        barrier(ctx)
        t0 = time.perf_counter()
        for _ in range(n):
            st = em.iter_EM()
            rounds += st.rounds
            ind_rounds += st.ind_rounds
            for k in fam:
                ms, c = em.hmm.kernel_ms(k)
                fam[k] += ms
                launches[k] += c
        barrier(ctx)
        dt = time.perf_counter() - t0
    finally:
        for h in hs:
            h.set_switch("no_bg_stream", 0)
            h.set_switch("spans", 0)
    return dict(fam=fam, launches=launches, rounds=rounds, ind_rounds=ind_rounds, steps=n, dt=dt)


def cold_run(ctx, run, iterations=20):
    """The run a user gets: `iterations` EM iterations from the starting values of examples/test.sh
    (--freq 0.1 --indF 0.1,0.2), timed as a whole between barriers (MAX over ranks) and one by one --
    a run has at least min_iters = 10 iterations (parse_args.cpp:5-34, EM.cpp:56), and its first ones
    take several times the objective rounds of the steady state `value` is measured in."""
    em = run["em"]
    reset_params(em)
    each_ms, rounds, ind_rounds = [], [], 0
    barrier(ctx)
    t0 = time.perf_counter()
    for _ in range(iterations):
        t = time.perf_counter()
        st = em.iter_EM()
        each_ms.append((time.perf_counter() - t) * 1e3)
        rounds.append(int(st.rounds))
        ind_rounds += int(st.ind_rounds)
    barrier(ctx)
    dt = time.perf_counter() - t0
    if ctx.world > 1:
        dt = float(allreduce(ctx, [dt], "max")[0])
    return dict(dt=dt, each_ms=each_ms, rounds=rounds, ind_rounds=ind_rounds, iterations=iterations)


def result_check(ctx, run, iterations=2):
    """Two EM iterations from the starting values, reduced to numbers that do not depend on how
    the job is sharded: the N-rank line's `check` must equal the one-GPU line's (log-likelihood
    1e-12, frequencies 1e-9, the same L-BFGS-B rounds) -- the reference's "results do not depend
    on the number of workers" (EM.cpp:151-161,198-201), checked by the run that is timed."""
    import numpy as np
    em, hmm = run["em"], run["em"].hmm
    I_tot, S_job = run["I_tot"], run["S_job"]
    reset_params(em)
    dbg = os.environ.get("NGHMM_DEBUG_CHECK")
    if dbg:
        sys.stderr.write(f"[check r{ctx.rank}] start indF {hmm.indF[:3]} alpha {hmm.alpha[:3]} freq {hmm.freq[:3]}\n")
    out = {"iterations": iterations, "tot_lkl": [], "rounds": [], "ind_rounds": [], "points": []}
    rounds_agree = True
    for _ in range(iterations):
        st = em.iter_EM()
        lkl = np.asarray(em.ind_lkl if em.ind_lkl is not None else hmm.ind_lkl, dtype=np.float64)
        if dbg:
            sys.stderr.write(f"[check r{ctx.rank}] iter: rounds {st.rounds} points {st.points} lkl sum {lkl.sum()!r} "
                             f"lkl {lkl[:3]} indF {hmm.indF[:3]} alpha {hmm.alpha[:3]} freq {hmm.freq[:3]}\n")
        if run["by_sites"]:            # the chain's values, the same on every rank
            tot = float(lkl.sum())
            r = allreduce(ctx, [st.rounds, -float(st.rounds)], "max")
            rounds_agree = rounds_agree and r[0] == -r[1]
            out["rounds"].append(int(st.rounds))
            out["ind_rounds"].append(int(st.ind_rounds))
            out["points"].append(int(st.points))
        else:                          # every rank its own individuals
            tot = float(allreduce(ctx, [lkl.sum()])[0])
            out["rounds"].append(int(allreduce(ctx, [st.rounds], "max")[0]))
            c = allreduce(ctx, [st.ind_rounds, st.points])
            out["ind_rounds"].append(int(c[0]))
            out["points"].append(int(c[1]))
        out["tot_lkl"].append(tot)
    indF, alpha = hmm.indF, hmm.alpha
    if run["by_sites"]:
        out["indF_sum"], out["alpha_sum"] = float(indF.sum()), float(alpha.sum())
        out["rounds_equal_on_all_ranks"] = bool(rounds_agree)
    else:
        c = allreduce(ctx, [indF.sum(), alpha.sum()])
        out["indF_sum"], out["alpha_sum"] = float(c[0]), float(c[1])
    # frequencies: sum, an index-weighted sum and 16 probe sites, by GLOBAL site index
    freq = hmm.freq
    lo = run["site_range"][0] if run["by_sites"] else 0
    s_glob = lo + np.arange(len(freq), dtype=np.int64)
    w = ((s_glob * 2654435761) % 1000003) / 1000003.0
    probes = [(2 * k + 1) * S_job // 32 for k in range(16)]
    pv = np.array([freq[p - lo] if lo <= p < lo + len(freq) else 0.0 for p in probes])
    part = np.concatenate([[freq.sum(), (freq * w).sum()], pv])
    if run["by_sites"]:
        part = allreduce(ctx, part)
    out["freq_sum"], out["freq_weighted_sum"] = float(part[0]), float(part[1])
    out["freq_probe_sites"] = probes
    out["freq_probes"] = [float(v) for v in part[2:]]
    out["data"] = f"IndexedSim seed 12345, {I_tot} x {S_job}"
    return out


def compare_checks(got, ref):
    """Relative differences of two `check` objects (the N-rank line against the one-GPU line)."""
    def rel(a, b):
        return abs(a - b) / max(abs(b), 1e-300)
    d = {
        "tot_lkl_max_rel_diff": max(rel(a, b) for a, b in zip(got["tot_lkl"], ref["tot_lkl"])),
        "freq_probe_max_rel_diff": max(rel(a, b) for a, b in zip(got["freq_probes"], ref["freq_probes"])),
        "freq_sum_rel_diff": rel(got["freq_sum"], ref["freq_sum"]),
        "freq_weighted_sum_rel_diff": rel(got["freq_weighted_sum"], ref["freq_weighted_sum"]),
        "indF_sum_rel_diff": rel(got["indF_sum"], ref["indF_sum"]),
        "alpha_sum_rel_diff": rel(got["alpha_sum"], ref["alpha_sum"]),
        "rounds_equal": got["rounds"] == ref["rounds"],
        "rounds": got["rounds"], "rounds_n1": ref["rounds"],
    }
    d["ok"] = bool(d["tot_lkl_max_rel_diff"] <= 1e-12 and d["freq_probe_max_rel_diff"] <= 1e-9 and
                   d["freq_weighted_sum_rel_diff"] <= 1e-9 and d["rounds_equal"])
    return d


def check_key(ctx, run):
    return (f"{ctx.args.workload}:{ctx.args.mode}:{run['I_tot']}x{run['S_job']}"
            f":{'cg' if ctx.call_geno else 'gl'}")


def exact_mode_line(ctx, budget_s=5.0):
    """The bit-exact mode (the only one that equals the CPU oracle bit for bit end to end,
    HMM.cpp:6-60 order preserved) on the first 100 000 sites of the workload's data set, a few
    iterations within `budget_s` of wall time; and the fast mode's first iteration on the same
    slice against it."""
    import numpy as np
    torch, pkg = ctx.torch, ctx.pkg
    I, S = ctx.wl["n_ind"], min(100_000, ctx.wl["n_sites"])
    sim = make_sim(pkg, ctx.wl, I, ctx.wl["n_sites"], ctx.device)
    gl, pos = sim.gl((0, I), (0, S)), sim.pos_dist(0, S)
    torch.cuda.synchronize()
    res = {}
    out = {"workload": f"{I} x {S // 1000}k (the first {S} sites of the data set)", "bound": "latency"}
    for name, mode in (("exact", pkg.MODE_EXACT), ("fast", pkg.MODE_FAST)):
        with pkg.NgsFHMM(I, S, device=ctx.local_rank, mode=mode) as h:
            h.load_device(gl.data_ptr(), pos.data_ptr())
            reset_params(h)
            ms, rounds, fam = [], [], dict.fromkeys(("forward", "backward", "lkl_batch", "est_maf"), 0.0)
            t0 = time.perf_counter()
            while len(ms) < (3 if name == "exact" else 1) and (not ms or time.perf_counter() - t0 < budget_s - 1.5):
                t = time.perf_counter()
                st = h.iter_EM()
                ms.append((time.perf_counter() - t) * 1e3)
                rounds.append(int(st.rounds))
                for k in fam:
                    fam[k] += h.kernel_ms(k)[0]
                if len(ms) == 1:
                    res[name] = (h.ind_lkl.copy(), h.freq)
            if name == "exact":
                out.update(ms_per_step=sum(ms) / len(ms), iterations_ms=ms, rounds=rounds,
                           site_ind_updates_per_s=I * S * len(ms) / (sum(ms) * 1e-3),
                           kernel_ms_per_step={k: v / len(ms) for k, v in fam.items()})
    a, b = res["exact"], res["fast"]
    out["fast_vs_exact_first_iteration"] = {
        "ind_lkl_max_rel_diff": float(np.max(np.abs(a[0] - b[0]) / np.abs(a[0]))),
        "freq_max_abs_diff": float(np.max(np.abs(a[1] - b[1])))}
    del gl, pos
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: an EM run is 10-100 iterations (parse_args.cpp:17-18) and the first three of
    # a run are not typical (L-BFGS-B needs 18, 13, 6 objective rounds there, then 4-5)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="fast", choices=["fast", "exact"])
    ap.add_argument("--n_ind", type=int, default=None)
    ap.add_argument("--n_sites", type=int, default=None)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_full_c2", action="store_true",
                    help="cpu_baseline also times BASELINE configs[1] (100 x 100k) in full")
    ap.add_argument("--replicas", type=int, default=1,
                    help="N = 1 only: R concurrent EM runs over the same data (multi-start, "
                         "ngsF-HMM.sh: 20 replicates), one host thread and HIP stream each; "
                         "value counts all R runs")
    ap.add_argument("--emulate_ranks", type=int, default=1,
                    help="N = 1 only: time the COMPUTE of one rank of a V-rank strong-scaling run "
                         "of the workload (its individuals for all sites + the frequency step on "
                         "its S / V sites over all individuals; exchanges are local copies): the "
                         "line's `predicted` object, not a measurement of V GPUs")
    ap.add_argument("--emulate_rccl", action="store_true",
                    help="with --emulate_ranks V and site shards: every all-gather also goes through "
                         "all_gather_into_tensor of a ONE-rank nccl (= RCCL) process group, issued on "
                         "the handle's stream exactly as a multi-rank run issues it: the code path "
                         "and its per-call cost on a one-GPU box")
    ap.add_argument("--shard", default=None, choices=["sites", "individuals"],
                    help="N > 1 (or --emulate_ranks): what a rank holds -- a contiguous range of "
                         "SITES for all individuals (fast mode's default: the ranges exchange six "
                         "doubles per individual and objective point, the frequency step nothing) "
                         "or a range of INDIVIDUALS for all sites (exact mode's only choice; every "
                         "posterior crosses the links once per iteration)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: shard the workload's individuals over the ranks (strong, "
                         "BASELINE configs[3]) or give every rank the full number (weak)")
    ap.add_argument("--no_alt", action="store_true",
                    help="N > 1, site shards: skip the second, short loop over the same data set in "
                         "the OTHER layout (individual shards, BASELINE configs[3]'s wording) that "
                         "the line embeds as `alt_sharding`")
    ap.add_argument("--no_exact_line", action="store_true",
                    help="N = 1: skip the bit-exact mode's bounded run (`exact_mode` in the line)")
    ap.add_argument("--no_check", action="store_true", help="skip the cross-N result check")
    ap.add_argument("--dry_launch", action="store_true",
                    help="N > 1: everything an N-rank run does BEFORE it touches a GPU -- the child launcher, the "
                         "rendezvous, the known-answer preflight of the collectives (gloo, CPU tensors), every "
                         "rank's shard of both layouts -- and nothing after: one JSON line, no measurement")
    ap.add_argument("--no_cold", action="store_true",
                    help="skip the 20 iterations from the starting values behind the timed loop (`value_cold20`)")
    ap.add_argument("--serial_kernels", action="store_true",
                    help="fast mode: the whole run with the background work (backward sweep, est_maf) "
                         "between the objective rounds on one stream, not next to them on a second "
                         "(what profiles/collect.sh traces: every kernel's span is then its own; "
                         "`value` is lower than the default run's)")
    ap.add_argument("--write_check", action="store_true",
                    help="N = 1: store this line's `check` in profiles/check_n1.json, which N > 1 "
                         "lines compare themselves with (`vs_n1`)")
    ap.add_argument("--timeout_s", type=float,
                    default=float(os.environ.get("NGHMM_BENCH_TIMEOUT_S", "120")),
                    help="process-group timeout: a collective (or the rendezvous) that takes longer "
                         "ends the rank with a non-zero exit code")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)       # does not return
    if args.gpus == 1:
        return run_rank(args)
    # a rank of several: whatever goes wrong ends THIS process with a non-zero code (a fresh exit,
    # never an exec), and the failure beacon / the process group's timeout end the others
    try:
        run_rank(args)
    except SystemExit:
        raise
    except BaseException as e:   # noqa: BLE001
        import traceback
        traceback.print_exc()
        sys.stderr.write(f"bench.py: rank {os.environ.get('RANK', '?')}: {type(e).__name__}: {e}\n")
        sys.stderr.flush()
        os._exit(4)


def run_rank(args):
    from datetime import timedelta
    import torch
    import torch.distributed as dist

    ctx = Ctx()
    ctx.args, ctx.torch, ctx.dist = args, torch, dist
    pkg = ctx.pkg = importlib.import_module("ngsf-hmm_amd")
    dd = ctx.dd = importlib.import_module("ngsf-hmm_amd.distributed")
    rank = ctx.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = ctx.world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}")
    backend = None
    # RCCL prints a version banner on STDOUT when its first communicator comes up: keep the
    # line contract (one JSON line on stdout) by pointing fd 1 at stderr until the preflight
    # has run
    sys.stdout.flush()
    fd_stdout = os.dup(1)
    os.dup2(2, 1)

    def restore_stdout():
        nonlocal fd_stdout
        if fd_stdout is not None:
            sys.stdout.flush()
            os.dup2(fd_stdout, 1)
            os.close(fd_stdout)
            fd_stdout = None

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # NGHMM_BENCH_BACKEND=gloo + NGHMM_BENCH_ONE_GPU=1: functional test of the
        # multi-rank path on a single-GPU box (all ranks share cuda:0, collectives staged
        # through the host); the measured configuration is always nccl = RCCL.
        backend = os.environ.get("NGHMM_BENCH_BACKEND", "nccl")
        if os.environ.get("NGHMM_BENCH_ONE_GPU"):
            local_rank = 0
        # a collective that hangs must end the run inside the driver's window, not wait out
        # torch's default (10 min under the RCCL watchdog, 30 min under gloo)
        tmo = timedelta(seconds=args.timeout_s)
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank),
                                    timeout=tmo)
        else:
            dist.init_process_group(backend=backend, timeout=tmo)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, "
                             f"--gpus says {args.gpus}")
    if args.dry_launch:
        return dry_launch(ctx, restore_stdout, backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    ctx.local_rank = local_rank
    device = ctx.device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    preflight = None
    if world > 1:
        # known-answer all-to-all + all-gather on float64 device tensors through the very
        # calls the iteration makes, BEFORE any data is loaded: a backend that moves wrong
        # bytes must end the run with a message, not a plausible number
        try:
            preflight = dd.preflight(device)
        except Exception as e:   # noqa: BLE001 - any failure of the collectives ends the run
            sys.stderr.write(f"bench.py: rank {rank}: {type(e).__name__}: {e}\n")
            sys.stderr.flush()
            os._exit(3)
    if not (args.emulate_rccl and world == 1):
        restore_stdout()

    wl = ctx.wl = dict(WORKLOADS[args.workload])
    if args.n_ind:
        wl["n_ind"] = args.n_ind
    if args.n_sites:
        wl["n_sites"] = args.n_sites
    V = args.emulate_ranks
    sharded = world > 1 or V > 1
    shard = (args.shard or ("sites" if args.mode == "fast" else "individuals")) if sharded else None
    if shard == "sites" and args.mode != "fast":
        raise SystemExit("--shard sites is a fast-mode layout")
    if V > 1 and (world > 1 or args.replicas > 1):
        raise SystemExit("--emulate_ranks V needs --gpus 1 and no replicas")
    if args.replicas > 1 and world > 1:
        raise SystemExit("--replicas needs --gpus 1")
    ctx.mode = pkg.MODE_FAST if args.mode == "fast" else pkg.MODE_EXACT
    ctx.call_geno = bool(wl.get("call_geno"))
    if ctx.call_geno:
        ctx.mode |= pkg.GENO_PACKED
    if args.emulate_rccl:
        if V < 2 or shard != "sites":
            raise SystemExit("--emulate_rccl goes with --emulate_ranks V and site shards")
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=device)
        preflight = dd.preflight(device)
        torch.cuda.synchronize()
        restore_stdout()

    try:
        run = build_run(ctx, shard)
    except ValueError as e:
        raise SystemExit(f"bench.py: {e}")
    restore_stdout()
    em = run["em"]
    by_sites, I, S, I_tot, S_job = run["by_sites"], run["I"], run["S"], run["I_tot"], run["S_job"]
    reset_params(em)

    # multi-start: R - 1 replicas sharing the likelihoods on the device, each with its own
    # (perturbed) starting values, run next to the main handle from R host threads
    replicas = []
    if args.replicas > 1:
        import numpy as np
        rng = np.random.default_rng(1)
        for r in range(args.replicas - 1):
            h = em.hmm.replica()
            h.set_params(rng.uniform(0.05, 0.3, I), rng.uniform(0.05, 0.5, I), rng.uniform(0.05, 0.3, S))
            h.init_emission()
            replicas.append(h)

    if args.serial_kernels:
        for h in handles_of(em) + replicas:
            h.set_switch("no_bg_stream", 1)
            h.set_switch("spans", 1)
    tl = timed_loop(ctx, run, args.steps, args.warmup, replicas)
    dt, fam, launches = tl["dt"], tl["fam"], tl["launches"]
    rounds, points, ind_rounds, ref_calls = tl["rounds"], tl["points"], tl["ind_rounds"], tl["ref_calls"]

    per_rank = None
    if world > 1:
        # every rank's own clocks, so that the scaling line can be read: kernel families
        # (HIP events on the library's stream), the exchange (duration and the part of it the
        # host actually waited for), the frequency step on the own site range
        per_rank = [None] * world
        dist.all_gather_object(per_rank, rank_clocks(ctx, run, tl))

    # (the accounting of the timed loop, before the check's iterations add to it)
    collective_bytes = None if world == 1 else em.collective_bytes_per_iter()
    exch_calls = (em.exchange.calls, em.exchange.bytes, em.exchange.host_ms) if by_sites and em.exchange else None
    # per-kernel durations for the roofline: from iterations whose kernels run one after the
    # other (fast mode overlaps them on two streams in the timed loop); after the timed loop's
    # accounting has been taken, before the check resets the parameters
    kt = None
    if args.mode == "fast" and not args.serial_kernels and args.replicas == 1:
        kt = serial_kernel_loop(ctx, run, max(1, min(5, args.steps)))
    cold = None
    if args.mode == "fast" and args.replicas == 1 and V == 1 and not args.no_cold:
        cold = cold_run(ctx, run)
    check = None
    if not args.no_check and V == 1:
        check = result_check(ctx, run)
    C_waves = em.hmm.layout()[0]
    for h in replicas:
        h.close()
    em.close()
    run["em"] = em = None
    torch.cuda.empty_cache()

    # ---- N > 1, site shards: the same data set once more in the OTHER layout ----
    alt = None
    if world > 1 and by_sites and not args.no_alt and args.scaling == "strong":
        alt = alt_sharding(ctx, min(args.steps, 5), min(args.warmup, 2), check)

    exact_line = None
    if (world == 1 and V == 1 and args.replicas == 1 and args.mode == "fast" and not args.no_exact_line
            and args.workload == "c3" and rank == 0):
        exact_line = exact_mode_line(ctx)

    if rank == 0:
        K = max(args.steps, 1)
        fam_timed = dict(fam)
        if kt is not None:      # the roofline's kernel times: the sequential loop's
            fam, launches, ind_rounds_k, K_k = kt["fam"], kt["launches"], kt["ind_rounds"], kt["steps"]
        else:
            ind_rounds_k, K_k = ind_rounds, K
        # site shards: all individuals x the job's sites (an emulated rank: x its own range)
        units = float(I_tot) * (S_job if by_sites and V == 1 else S) * K * args.replicas
        if not by_sites and V > 1:
            units = float(I) * S * K
        units_per_iter = units / K
        est_sites = S if by_sites else S / (world * V)       # what one rank's est_maf covers
        est_inds = I if by_sites else I_tot
        call_geno = ctx.call_geno
        # dominant kernel family by measured time, and its algorithmic traffic per launch
        # (DESIGN.md section 4); est_maf = 24 B GL + 8 B posterior per site-individual;
        # fast mode: the first objective round of an iteration (all I individuals) reads the
        # 16 B GL and writes the 8 B emission ratio + 4 B checkpoints; later rounds read 8 B per
        # still-active individual; the E-step then reads 8 + 4 B and writes 8 B
        fast = args.mode == "fast"
        glb = 0.25 if call_geno else 24.0     # bytes of genotype likelihoods per cell (est_maf)
        glq = 0.25 if call_geno else 16.0     # ... in the interleaved copy the forward walk reads
        # ---- every kernel of the iteration against BOTH roofs; `bound` names its limiter ----
        # times: HIP events on the library's stream around each kernel family (nghmm_kernel_ms)
        pmc_summ, pmc_why = (pmc_summary() if (args.workload == "c3" and fast and world == 1 and V == 1 and
                                               not args.n_ind and not args.n_sites)
                             else (None, "PMC passes exist for the default workload on one GPU only"))
        pmc = pmc_traffic(pmc_summ, I, C_waves)
        isa, isa_path, isa_why = isa_counts()
        rows = {}
        n_first = launches["lkl_first"]
        later_ms = fam["lkl_batch"] - fam["lkl_first"]
        later_ind = max(ind_rounds_k - I * n_first, 0)
        later_launches = max(launches["lkl_batch"] - n_first, 0)
        if fast:
            if n_first and fam["lkl_first"] > 0:
                rows["lkl_first_round"] = dict(ms=fam["lkl_first"], launches=n_first,
                                               bytes=(glq + 12.0) * S * I * n_first, sites_ind=S * I * n_first,
                                               traffic=pmc.get("lkl_first_round_per_launch"))
            if later_launches and later_ms > 0:
                t = pmc.get("lkl_later_rounds_per_individual")
                rows["lkl_later_rounds"] = dict(ms=later_ms, launches=later_launches,
                                                bytes=8.0 * S * later_ind, sites_ind=S * later_ind,
                                                traffic=t * later_ind / later_launches if t else None)
            if launches["est_maf"] and fam["est_maf"] > 0:
                t = pmc.get("est_maf_per_iteration")
                rows["est_maf"] = dict(ms=fam["est_maf"], launches=launches["est_maf"],
                                       bytes=(glb + 8.0) * est_sites * est_inds * launches["est_maf"],
                                       sites=est_sites * launches["est_maf"], traffic=t)
            if launches["forward"] and fam["forward"] > 0:
                rows["backward_sweep"] = dict(ms=fam["forward"], launches=launches["forward"],
                                              bytes=20.0 * S * I * launches["forward"],
                                              traffic=pmc.get("backward_sweep_per_iteration"))
        roof_all = {}
        for name, r in rows.items():
            meta = KERNELS[name]
            secs = r["ms"] * 1e-3
            gbs = r["bytes"] / secs / 1e9
            e = {"bound": meta["bound"], "kernel": meta["kernel"],
                 "ms_per_em_iteration": r["ms"] / K_k, "launches": r["launches"],
                 "avg_launch_ms": r["ms"] / r["launches"],
                 "hbm": {"achieved_GBps": gbs, "peak_GBps": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": r["bytes"] / r["launches"],
                         "traffic_bytes_per_launch": r["traffic"]},
                 "note": meta["note"]}
            w_valu = w_fp64 = None
            blocks = (isa or {}).get(name)
            if name == "est_maf" and blocks:
                vf = estmaf_instr_per_site(est_inds, (isa or {}).get("est_maf:model"))
                if vf:
                    per = {"valu": vf[0], "fp64": vf[1],
                           "what": "from the assembly's parts: an upper bound (the build decision's "
                                   "block is counted with every exact pass)"}
                    # the counted instructions of this build's PMC pass, when there is one: the
                    # assembly model then only supplies the FP64 share
                    k = (pmc_summ or {}).get("k_fast_estmaf<16, 64, true>")
                    if k and k.get("insts_valu_per_launch") and k.get("avg_grid_threads") and est_inds == 1000:
                        v_pmc = k["insts_valu_per_launch"] / (k["avg_grid_threads"] / 64.0)
                        per = {"valu": v_pmc, "fp64": vf[1] * v_pmc / vf[0], "assembly_model_valu": vf[0],
                               "what": f"VALU: SQ_INSTS_VALU per site of this build's PMC pass ({PMC_SUMMARY}); "
                                       "FP64: that count times the FP64 share of the assembly's parts"}
                    w_valu, w_fp64 = per["valu"] * r["sites"], per["fp64"] * r["sites"]
            elif meta.get("isa") and blocks:
                v, f = walk_instr_per_site(blocks)
                w_valu, w_fp64, per = v * r["sites_ind"] / 64.0, f * r["sites_ind"] / 64.0, {"valu": v, "fp64": f}
            if w_valu:
                e["fp64_valu"] = {"achieved_wave_instr_per_s": w_fp64 / secs, "peak": FP64_ISSUE_PEAK,
                                  "frac": w_fp64 / secs / FP64_ISSUE_PEAK,
                                  "valu_issue_frac": w_valu / secs / FP64_ISSUE_PEAK,
                                  "instr_per_site": per,
                                  "source": f"{isa_path} (device assembly of this build, id {this_build_id()})"}
            elif meta["bound"] == "fp64_valu":
                e["fp64_valu"] = {"achieved_wave_instr_per_s": None, "peak": FP64_ISSUE_PEAK, "frac": None,
                                  "instr_per_site": None,
                                  "reason": isa_why or f"{isa_path}: no blocks for this kernel"}
            roof_all[name] = e
        # the iteration as a whole (north_star: "as fraction of HBM roofline"): algorithmic bytes and
        # FP64 wave-instructions of its four kernels per EM iteration against the TIMED loop's
        # iteration (wall clock, everything in it), next to the kernels' own sum
        roof_iter = None
        if fast and roof_all:
            it_bytes = sum(r["bytes"] for r in rows.values()) / K_k
            it_traffic = [roof_all[n]["hbm"]["traffic_bytes_per_launch"] for n in rows]
            it_traffic = (sum(t * rows[n]["launches"] for t, n in zip(it_traffic, rows)) / K_k
                          if all(t is not None for t in it_traffic) else None)
            it_fp64 = [roof_all[n].get("fp64_valu", {}).get("achieved_wave_instr_per_s") for n in rows
                       if roof_all[n]["bound"] == "fp64_valu" or "fp64_valu" in roof_all[n]]
            it_fp64 = (sum(roof_all[n]["fp64_valu"]["achieved_wave_instr_per_s"] * rows[n]["ms"] * 1e-3
                           for n in rows if roof_all[n].get("fp64_valu", {}).get("achieved_wave_instr_per_s")) / K_k
                       if it_fp64 and all(v is not None for v in it_fp64) else None)
            wall_s = dt / K
            roof_iter = {"ms_per_em_iteration": wall_s * 1e3,
                         "kernels_ms_per_em_iteration": sum(r["ms"] for r in rows.values()) / K_k,
                         "algorithmic_bytes_per_em_iteration": it_bytes,
                         "traffic_bytes_per_em_iteration": it_traffic,
                         "hbm": {"achieved_GBps": it_bytes / wall_s / 1e9, "peak_GBps": HBM_PEAK_GBS,
                                 "frac": it_bytes / wall_s / 1e9 / HBM_PEAK_GBS},
                         "fp64_valu": ({"achieved_wave_instr_per_s": it_fp64 / wall_s, "peak": FP64_ISSUE_PEAK,
                                        "frac": it_fp64 / wall_s / FP64_ISSUE_PEAK,
                                        "wave_instr_per_em_iteration": it_fp64}
                                       if it_fp64 else None),
                         "note": "the four kernels of roofline_all_kernels (objective rounds first / later, est_maf, "
                                 "backward sweep) per EM iteration of THIS line's timed loop; the backward sweep's "
                                 "instructions are not counted (HBM-bound: no assembly block model for it)"}
        if not fast:   # exact mode: latency-bound chains, not a roofline candidate
            for k in ("forward", "backward", "lkl_batch", "est_maf"):
                if launches[k] and fam[k] > 0:
                    roof_all[k] = {"bound": "latency", "ms_per_em_iteration": fam[k] / K,
                                   "launches": launches[k]}
        # the contract's object: the kernel that takes the most time, against ITS roof
        dom = max(roof_all, key=lambda k: roof_all[k]["ms_per_em_iteration"]) if roof_all else None
        if dom and roof_all[dom]["bound"] == "fp64_valu" and roof_all[dom]["fp64_valu"]["frac"] is None:
            # no instruction counts for this build: the dominant kernel against the HBM roof only
            d = roof_all[dom]
            roofline = {"bound": "hbm", "kernel": dom, "achieved": d["hbm"]["achieved_GBps"],
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["hbm"]["frac"],
                        "traffic": d["hbm"]["traffic_bytes_per_launch"],
                        "algorithmic_bytes_per_launch": d["hbm"]["algorithmic_bytes_per_launch"],
                        "avg_launch_ms": d["avg_launch_ms"], "launches": d["launches"],
                        "device_kernel": d["kernel"],
                        "note": "this kernel is FP64-issue bound (DESIGN.md section 4), but "
                                + d["fp64_valu"]["reason"] + ": only its HBM fraction can be stated"}
        elif dom and roof_all[dom]["bound"] == "fp64_valu":
            d = roof_all[dom]
            roofline = {"bound": "fp64_valu", "kernel": dom, "achieved": d["fp64_valu"]["achieved_wave_instr_per_s"],
                        "peak": FP64_ISSUE_PEAK, "unit": "FP64 wave-instructions/s",
                        "frac": d["fp64_valu"]["frac"],
                        "valu_issue_frac": d["fp64_valu"]["valu_issue_frac"],
                        "hbm_frac": d["hbm"]["frac"], "hbm_achieved_GBps": d["hbm"]["achieved_GBps"],
                        "traffic": d["hbm"]["traffic_bytes_per_launch"],
                        "algorithmic_bytes_per_launch": d["hbm"]["algorithmic_bytes_per_launch"],
                        "avg_launch_ms": d["avg_launch_ms"], "launches": d["launches"],
                        "device_kernel": d["kernel"], "note": d["note"]}
        elif dom and roof_all[dom]["bound"] == "hbm":
            d = roof_all[dom]
            roofline = {"bound": "hbm", "kernel": dom, "achieved": d["hbm"]["achieved_GBps"],
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["hbm"]["frac"],
                        "traffic": d["hbm"]["traffic_bytes_per_launch"],
                        "algorithmic_bytes_per_launch": d["hbm"]["algorithmic_bytes_per_launch"],
                        "avg_launch_ms": d["avg_launch_ms"], "launches": d["launches"],
                        "device_kernel": d["kernel"], "note": d["note"]}
        else:
            roofline = {"bound": "latency", "kernel": dom, "achieved": None, "peak": None, "unit": None,
                        "frac": None, "traffic": None,
                        "note": "exact mode: sequential log-space chains (DESIGN.md section 4)"}
        if roofline.get("traffic") is not None:
            roofline["traffic_source"] = (f"{PMC_SUMMARY}: rocprofv3 --pmc FETCH_SIZE / "
                                          f"WRITE_SIZE passes of this workload and THIS build (id "
                                          f"{this_build_id()}, profiles/collect.sh), not measured in this run")
        else:
            roofline["traffic_source"] = pmc_why
        roofline["kernel_times"] = (
            f"HIP events over {K_k} iterations that follow the timed loop with the library's switch "
            "no_bg_stream (every kernel alone on the chip; in the timed loop the backward sweep and "
            "est_maf run on a second stream next to the objective rounds and the spans overlap)"
            if kt is not None else
            ("HIP events over the timed loop" if (args.mode != "fast" or args.serial_kernels) else
             f"HIP events over {max(1, min(5, args.steps))} iterations that follow the timed loop (fast mode "
             "records them on request only, switch `spans`: the timed iterations carry none)"))
        roofline["timed_loop"] = ("no timing events inside the timed iterations (fast mode; switch `spans` off): "
                                  "`per_step_kernel_ms_timed_loop` comes from iterations behind them, two streams "
                                  "as in the timed loop" if (args.mode == "fast" and not args.serial_kernels) else
                                  "timing events inside the timed iterations")
        each = tl["each_ms"]
        n_run = min(20, len(each))
        vs_n1 = None
        if check is not None:
            key = check_key(ctx, run)
            ref = json.load(open(CHECK_REF)) if os.path.exists(CHECK_REF) else {}
            if world == 1 and args.write_check and args.replicas == 1:
                ref[key] = dict(check, build_id=this_build_id())
                json.dump(ref, open(CHECK_REF, "w"), indent=1, sort_keys=True)
            elif world > 1 and key in ref and ref[key].get("build_id") != this_build_id():
                vs_n1 = {"ok": None, "note": f"profiles/check_n1.json holds the one-GPU `check` of build "
                                             f"{ref[key].get('build_id')}, the sources here are "
                                             f"{this_build_id()}: not compared (bench.py --write_check on one GPU)"}
            elif world > 1 and key in ref:
                vs_n1 = dict(compare_checks(check, ref[key]),
                             source="profiles/check_n1.json: the `check` of a one-GPU run of this "
                                    "workload (bench.py --write_check)")
            elif world > 1:
                vs_n1 = {"ok": None, "note": f"no one-GPU `check` stored for {key}: compare this line's "
                                             "`check` with the --gpus 1 line's"}
        out = {
            "metric": "site-ind updates/sec (EM iterations x individuals x sites / s), 1M sites x 1k ind",
            "value": units / dt,
            "unit": "site-ind updates/s",
            "em_iters_per_sec": K / dt,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            # total work fixed as N grows (strong: this run is a point of the 1/2/4/8 series
            # over the same 1000 x 1M job) or per-GPU work fixed (weak)
            "scaling": args.scaling,
            "ranks": world,
            "collectives": (None if world == 1 else
                            ("rccl (nccl backend)" if backend == "nccl" else
                             f"{backend}, all ranks on one GPU: functional run, not a measurement")),
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "replicas": args.replicas,
            "config": {"workload": wl["name"] + (f", {args.replicas} concurrent multi-start "
                                                 f"replicas" if args.replicas > 1 else ""),
                       "n_ind_total": I_tot, "n_ind_per_gpu": I,
                       "n_sites": S_job if by_sites else S, "n_sites_per_gpu": S,
                       "mode": args.mode, "freq_est": 1,
                       "data_set": "simulate.IndexedSim(seed 12345): every element a hash of its global "
                                   "(individual, site) indices -- the N-rank job holds slices of the "
                                   "N = 1 job's data set, whatever the sharding",
                       "sharding": (None if world == 1 else
                                    (f"sites: {S} of {S_job} sites per GPU for all {I} individuals; "
                                     f"the ranges exchange six doubles per individual and E-step "
                                     f"and per objective point and round (one small all-gather "
                                     f"each), the allele-frequency step nothing") if by_sites else
                                    f"individuals: {I} of {I_tot} individuals per GPU for all sites; allele-"
                                    f"frequency step on {S // world} sites x all individuals per "
                                    f"GPU (all-to-all of posteriors, all-gather of frequencies)")},
            "roofline": roofline,
            "roofline_all_kernels": roof_all,
            "roofline_iteration": roof_iter,
            "per_step_kernel_ms": {k: fam[k] / K_k for k in fam if k != "lkl_first"},
            "per_step_kernel_ms_timed_loop": {k: fam_timed[k] / K for k in fam_timed if k != "lkl_first"},
            # the metric is EM iterations/s and a run is >= 10 iterations from the starting values
            # (parse_args.cpp:5-34 min_iters 10): `ms_per_step` is the steady state after the
            # warm-up iterations, these are the run's first iterations themselves
            "first_iterations_ms": each[:args.warmup] if args.warmup else each[:min(3, len(each))],
            "first_iterations_rounds": tl["each_rounds"][:max(args.warmup, min(3, len(each)))],
            "run_of_20_ms_per_iter": (sum(each[:20]) / 20 if len(each) >= 20 else None),
            # THE RUN A USER GETS next to the steady state: 20 iterations from --freq 0.1 --indF 0.1,0.2
            # (a run is >= 10, parse_args.cpp:5-34), same units, same accounting of algorithmic bytes
            "value_cold20": (None if cold is None else units_per_iter * cold["iterations"] / cold["dt"]),
            "cold20": (None if cold is None else cold20_object(cold, units_per_iter, I, S, glq, glb, est_sites, est_inds)),
            "run_ms_per_iter": {"iterations": n_run, "value": sum(each[:n_run]) / max(n_run, 1),
                                "note": "mean wall time of the run's first iterations, warm-up included "
                                        "(this rank's clock)"},
            "check": check,
            "vs_n1": vs_n1,
            "alt_sharding": alt,
            "exact_mode": exact_line,
            "parity": parity_object(),
            "predicted": (None if V == 1 else {
                "emulated_rank_of": V,
                "shard": "sites" if by_sites else "individuals",
                "what": (f"compute of ONE rank of a {V}-rank strong-scaling run of this workload on "
                         f"one GPU (all {I} individuals for {S} of {S_job} sites; every all-gather "
                         f"replaced by {V} local copies of the same size on the handle's stream"
                         + (", after an all_gather_into_tensor of a one-rank RCCL group issued on "
                            "that stream" if args.emulate_rccl else "") + ": "
                         f"{exch_calls[0] / K:.1f} per iteration, "
                         f"{exch_calls[1] / K / 1e3:.0f} kB per iteration from each rank): "
                         f"`value` and `ms_per_step` "
                         f"of this line are that rank's, NOT a cohort's") if by_sites else
                        f"compute of ONE rank of a {V}-rank strong-scaling run of this workload on "
                        f"one GPU ({I} of {I * V} individuals for all sites, est_maf on {S // V} "
                        f"sites x {I * V} individuals), exchanges replaced by local copies: "
                        f"`value` and `ms_per_step` of this line are that rank's, NOT a cohort's",
                "rank_ms_per_iteration": dt / K * 1e3,
                "all_gather_host_ms_per_call": (exch_calls[2] / max(exch_calls[0], 1)
                                                if by_sites else None),
                "whole_job_site_ind_updates_per_s_if_communication_is_hidden":
                    float(I) * S_job * K / dt if by_sites else float(I * V) * S * K / dt}),
            "preflight": preflight,
            "collective_bytes_per_iter": (None if world == 1 else dict(
                collective_bytes,
                note=("bytes leaving each GPU per EM iteration: its part of every all-gather (six "
                      "doubles per individual and E-step / per objective point and round) to the "
                      "other ranks") if by_sites else
                     "bytes leaving each GPU per EM iteration: posterior slices to the other "
                     "ranks' site ranges (all-to-all, issued right after the E-step, under the "
                     "remaining objective rounds) and the own frequencies (all-gather)")),
            "exchange_ms": (None if world == 1 else {
                k: max(r["exchange_ms_per_iter"][k] for r in per_rank)
                for k in per_rank[0]["exchange_ms_per_iter"]}),
            "per_rank": per_rank,
            "all_gather_rounds": all_gather_rounds(per_rank) if (per_rank and by_sites) else None,
            "regime": regime_object(tl, K, I, S_job if by_sites else S, world * V if by_sites else 1),
            "bfgs": {"rounds_per_iter": rounds / K, "points_per_iter": points / K,
                     "ind_rounds_per_iter": ind_rounds / K,
                     "reference_forward_passes_per_ind_iter": ref_calls / (K * I)},
        }
        if not args.no_cpu_baseline and world == 1:   # on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(pkg, full_c2=args.cpu_full_c2, wl=wl)
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1 or args.emulate_rccl:
        dist.destroy_process_group()


def cold20_object(cold, units_per_iter, I, S, glq, glb, est_sites, est_inds):
    """`value_cold20`'s details: every iteration's wall time and objective rounds, and the run's
    algorithmic HBM bytes by the same per-kernel model as roofline_iteration (this rank: a first round
    over all individuals, 8 B per site of every later individual-round, backward sweep, est_maf)."""
    n = cold["iterations"]
    later = max(cold["ind_rounds"] - I * n, 0)
    by = ((glq + 12.0) * S * I + 20.0 * S * I + (glb + 8.0) * est_sites * est_inds) * n + 8.0 * S * later
    return {"iterations": n, "ms_per_iter": cold["dt"] / n * 1e3, "iterations_ms": cold["each_ms"],
            "rounds": cold["rounds"], "ind_rounds_per_iter": cold["ind_rounds"] / n,
            "site_ind_updates_per_s": units_per_iter * n / cold["dt"],
            "algorithmic_bytes_per_iteration": by / n,
            "hbm": {"achieved_GBps": by / cold["dt"] / 1e9, "peak_GBps": HBM_PEAK_GBS,
                    "frac": by / cold["dt"] / 1e9 / HBM_PEAK_GBS},
            "start": "--freq 0.1 --indF 0.1,0.2 (examples/test.sh)"}


def dry_launch(ctx, restore_stdout, backend):
    """`--dry_launch`: the part of an N-rank run that comes before its first GPU call, on the CPU."""
    args, dd, dist = ctx.args, ctx.dd, ctx.dist
    if ctx.world < 2:
        raise SystemExit("--dry_launch rehearses a multi-rank launch: --gpus N with N > 1")
    preflight = dd.preflight(None)                  # CPU tensors through the group's collectives
    restore_stdout()
    ctx.wl = dict(WORKLOADS[args.workload])
    ctx.call_geno = bool(ctx.wl.get("call_geno"))
    mine = {"rank": ctx.rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid()}
    for shard in ("sites", "individuals"):
        try:
            sh = shard_shapes(ctx, shard)
            mine[shard] = {"n_ind": sh["I"], "n_sites": sh["S"], "ind_range": list(sh["ind_range"]),
                           "site_range": list(sh["site_range"])}
        except ValueError as e:
            mine[shard] = {"refused": str(e)}
    ranks = [None] * ctx.world
    dist.all_gather_object(ranks, mine)
    dist.barrier()
    if ctx.rank == 0:
        # the shards tile the job: every site (individual) belongs to exactly one rank
        for shard, key, total in (("sites", "site_range", ctx.wl["n_sites"]), ("individuals", "ind_range", ctx.wl["n_ind"])):
            got = [r[shard] for r in ranks if "refused" not in r[shard]]
            if len(got) == ctx.world:
                cuts = sorted(tuple(g[key]) for g in got)
                assert cuts[0][0] == 0 and cuts[-1][1] == total and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:])), cuts
        print(json.dumps({"dry_launch": True, "n_gpus": ctx.world, "backend": backend, "workload": ctx.wl["name"],
                          "preflight": preflight, "ranks": ranks,
                          "note": "launcher, rendezvous, collectives and shard arithmetic of an N-rank run without a "
                                  "GPU call: not a measurement"}))
        sys.stdout.flush()
    dist.destroy_process_group()


def regime_object(tl, K, n_ind, n_sites, site_shards):
    """Where the timed iterations spent their individual-rounds and sites: the share of every
    loop-body version of the objective kernels (fd_pattern: `s` = small-alpha kappa form, a polynomial
    per site instead of an exponential; `general` = an exponential per point and site) and the share
    of the frequency step's sites that left its common route (exact passes, one checked
    interpolant, the rest on it).  This rank's handle(s)."""
    mc, ec = dict(tl.get("mode_counts") or {}), tl.get("estmaf_counts") or {}
    if not mc and not ec:
        return None
    mixed = mc.pop("rounds_of_mixed_versions", 0)   # (rounds, not individual-rounds: include/nghmm_debug.h)
    tot = float(sum(mc.values())) or 1.0
    sites = float(n_sites) / site_shards * K
    return {"objective_kernel_versions": {k: {"ind_rounds_per_iter": v / K, "share": v / tot}
                                          for k, v in sorted(mc.items(), key=lambda kv: -kv[1])},
            "rounds_of_mixed_versions_in_one_launch_per_iter": mixed / K,
            "small_alpha_share": sum(v for k, v in mc.items() if k != "general" and "s" in k) / tot,
            "general_kernel_share": mc.get("general", 0) / tot,
            "est_maf_sites_off_the_common_route": {k: {"sites_per_iter": v / K, "share": v / sites}
                                                   for k, v in ec.items()}}


def all_gather_rounds(per_rank):
    """The all-gathers of the last timed iteration side by side over the ranks: per call (round)
    the slowest and the fastest rank's duration on the stream.  A collective ends for everybody
    when the last part has arrived, so a rank that reaches it early measures the others' lag as
    its own duration: max - min over the ranks is the skew between them at that round, min the
    wire time.  What a first multi-GPU run needs to explain its own efficiency."""
    calls = [p.get("all_gather_ms_calls_last_iter") or [] for p in per_rank]
    n = min((len(c) for c in calls), default=0)
    if not n:
        return None
    rows = [{"call": k, "min_ms": min(c[k] for c in calls), "max_ms": max(c[k] for c in calls)} for k in range(n)]
    for r in rows:
        r["skew_ms"] = r["max_ms"] - r["min_ms"]
    return {"calls_per_iteration": n, "per_call": rows,
            "sum_min_ms": sum(r["min_ms"] for r in rows), "sum_max_ms": sum(r["max_ms"] for r in rows),
            "sum_skew_ms": sum(r["skew_ms"] for r in rows),
            "note": "one all-gather per objective round (+ the E-step's when it is not merged): duration on the "
                    "handle's stream by HIP events under nccl, the host's clock under gloo"}


def rank_clocks(ctx, run, tl):
    """This rank's own clocks of a timed loop, per iteration."""
    em, fam = run["em"], tl["fam"]
    K_ = max(tl["steps"], 1)
    mine = {"rank": ctx.rank, "kernel_ms_per_iter": {k: fam[k] / K_ for k in fam if k != "lkl_first"},
            "rounds_per_iter": tl["rounds"] / K_}
    if run["by_sites"]:
        ex = em.exchange
        mine["exchange_ms_per_iter"] = {"all_gather_host_calls": ex.host_ms / K_}
        mine["all_gathers_per_iter"] = ex.calls / K_
        mine["all_gather_bytes_per_iter"] = ex.bytes / K_
        # every all-gather of the timed loop's LAST iteration (one per objective round: the calls
        # are the same on every rank, so rank 0 can set them side by side), and all of them summed
        log = tl.get("exchange_log") or []
        per_iter = int(round(ex.calls / K_)) if K_ else 0
        mine["all_gather_ms_calls_last_iter"] = log[-per_iter:] if per_iter and len(log) >= per_iter else log
        # (events only in the last timed iteration under nccl; every call under gloo / emulation)
        n_logged = len(log) / per_iter if per_iter else 0
        mine["exchange_ms_per_iter"]["all_gather_on_stream"] = sum(log) / n_logged if n_logged else None
    else:
        mine["exchange_ms_per_iter"] = {
            "all_to_all": em.timing["a2a_ms"] / K_,
            "all_to_all_exposed": em.timing["a2a_exposed_ms"] / K_,
            "all_to_all_hidden": max(em.timing["a2a_ms"] - em.timing["a2a_exposed_ms"], 0.0) / K_,
            "all_gather": em.timing["allgather_ms"] / K_,
            "freq_step_call": em.timing["freq_step_ms"] / K_}
    return mine


def alt_sharding(ctx, steps, warmup, main_check):
    """The job's OTHER layout in the same process group, on the same data set: individual shards
    (BASELINE configs[3]: "sharded by individual ... RCCL freq all-reduce" -- an all-to-all of
    every posterior and an all-gather of the frequencies), a short loop and
    the same result check.  Every rank reports whether its part worked, so that a layout that
    does not fit (or fails) costs the line this object, not the run."""
    dist, torch = ctx.dist, ctx.torch
    out = {"sharding": "individuals", "steps": steps, "warmup": warmup}
    # every rank says whether the rank-local part of the set-up worked BEFORE any of them enters
    # the collectives of the load: a layout that does not fit this job (a size that does not
    # divide, memory) costs the line this object, not the run
    agreed = []

    def agree(err):
        agreed.append(True)
        return allreduce(ctx, [1.0 if err else 0.0], "max")[0] == 0.0
    run = None
    try:
        run = build_run(ctx, "individuals", agree)
        reset_params(run["em"])
    except ShardSetupRefused as e:      # the ranks agreed that one of them could not set up
        out["skipped"] = f"{type(e).__name__}: {e}"
        return out
    except ValueError as e:
        if agreed:
            # PAST the agreement point (the load, which contains collectives for individual
            # shards): the other ranks are inside those collectives, an all-reduce from here
            # would pair up with the wrong call.  Fatal for the job: main() tells the others.
            raise
        # refused before any allocation (shard_shapes: a size that does not divide), on every
        # rank alike: each of them makes the agreement's all-reduce here instead
        agree(str(e))
        out["skipped"] = f"{type(e).__name__}: {e}"
        return out
    em = run["em"]
    tl = timed_loop(ctx, run, steps, warmup)
    K = max(steps, 1)
    per_rank = [None] * ctx.world
    dist.all_gather_object(per_rank, rank_clocks(ctx, run, tl))
    coll = em.collective_bytes_per_iter()
    check = None if ctx.args.no_check else result_check(ctx, run)
    out.update({
        "ms_per_step": tl["dt"] / K * 1e3,
        "value": float(run["I_tot"]) * run["S"] * K / tl["dt"],
        "n_ind_per_gpu": run["I"], "n_sites_per_gpu": run["S"],
        "first_iterations_ms": tl["each_ms"][:warmup],
        "exchange_ms": {k: max(r["exchange_ms_per_iter"][k] for r in per_rank)
                        for k in per_rank[0]["exchange_ms_per_iter"]},
        "collective_bytes_per_iter": coll,
        "per_step_kernel_ms": {k: tl["fam"][k] / K for k in tl["fam"] if k != "lkl_first"},
        "rounds_per_iter": tl["rounds"] / K,
        "per_rank": per_rank,
        "check": check,
        "vs_main_sharding": (compare_checks(check, main_check) if check and main_check else None),
    })
    em.close()
    run["em"] = None
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()

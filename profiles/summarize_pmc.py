#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one CSV per pass, filtered to this library's kernels)
into a small JSON that bench.py reads for the `roofline.traffic` field.

HBM bytes follow MI355X_MICROARCH.md section HBM / cdna_hip_programming.md section 7:
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a
wide (16 B/lane) coalesced streaming read, so it is doubled for kernels whose dominant
loads are 16 B per lane (the interleaved emission / genotype-likelihood streams);
WRITE_SIZE is exact.  Collected in separate passes (FETCH_SIZE and WRITE_SIZE do not fit
one pass).

Per kernel (template arguments kept) and per timed FAMILY of bench.py (lkl_batch = every
k_fast_lkl_* kernel, forward = the E-step's kernels, est_maf, emission): bytes per launch
and, for families, bytes per EM iteration of the pass and per objective round.

usage: summarize_pmc.py <sq.csv> <fetch.csv> <write.csv> <out.json>
"""
import collections
import csv
import json
import re
import os
import sys

FAMILIES = {
    "lkl_batch": ("k_fast_lkl_fd", "k_fast_lkl_chunks", "k_fast_lkl_finish"),
    "forward": ("k_fast_chunk_ops", "k_fast_bounds", "k_fast_bwd_recompute",
                "k_fast_bwd_recompute8"),
    "est_maf": ("k_fast_estmaf", "k_fast_estmaf_resume", "k_fast_estmaf_rows",
                "k_fast_estmaf_rows_resume", "k_fast_estmaf_interp", "k_fast_estmaf_stream",
                "k_fast_estmaf_called_sums", "k_fast_estmaf_called_passes"),
    "emission": ("k_fast_emission", "k_fast_freq_interleave"),
    "bfgs": ("k_bfgs_advance",),
}


def short(name):
    """'void nghmm::(anonymous namespace)::k_x<2, 2, true>(args...)' -> 'k_x<2, 2, true>'"""
    m = re.search(r"(k_\w+(?:<[^>(]*>)?)\(", name)
    return m.group(1) if m else name


def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    grid = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if "nghmm" not in r["Kernel_Name"]:
            continue
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[k]:
            n[k].add(r["Dispatch_Id"])
            grid[k] += float(r["Grid_Size"])
    return acc, {k: len(v) for k, v in n.items()}, grid


def main():
    sq, fetch, write, out = sys.argv[1:5]
    a_sq, n_sq, _ = per_kernel(sq)
    a_f, n_f, grid_f = per_kernel(fetch)
    a_w, n_w, _ = per_kernel(write)
    res = {}
    for k in sorted(set(a_sq) | set(a_f)):
        d = {}
        if k in a_f:
            raw = a_f[k]["FETCH_SIZE"] * 1024.0 / n_f[k]
            d["fetch_bytes_raw_per_launch"] = raw
            d["fetch_bytes_corrected_per_launch"] = 2.0 * raw
            d["launches_fetch_pass"] = n_f[k]
            d["avg_grid_threads"] = grid_f[k] / n_f[k]
        if k in a_w:
            d["write_bytes_per_launch"] = a_w[k]["WRITE_SIZE"] * 1024.0 / n_w[k]
        if k in a_sq:
            c = a_sq[k]
            wc = c["SQ_WAVE_CYCLES"] or 1.0
            d.update({
                "valu_active_frac_per_wave": c["SQ_ACTIVE_INST_VALU"] / wc,
                "wait_inst_frac": c["SQ_WAIT_INST_ANY"] / wc,
                "wait_any_frac": c["SQ_WAIT_ANY"] / wc,
                "insts_valu_per_launch": c["SQ_INSTS_VALU"] / n_sq[k],
                "grbm_gui_active_per_launch": c["GRBM_GUI_ACTIVE"] / n_sq[k],
            })
        if "fetch_bytes_corrected_per_launch" in d:
            d["hbm_bytes_per_launch"] = d["fetch_bytes_corrected_per_launch"] + d.get(
                "write_bytes_per_launch", 0.0)
        res[k] = d
    # the backward sweep runs exactly once per EM iteration of the pass
    iters = (res.get("k_fast_bwd_recompute8") or res.get("k_fast_bwd_recompute", {})).get(
        "launches_fetch_pass", 0)
    # (round 6: no k_fast_lkl_finish in device-planned rounds: one k_bfgs_advance<false> per round)
    rounds = (res.get("k_fast_lkl_finish", {}).get("launches_fetch_pass", 0) or
              res.get("k_bfgs_advance<false>", {}).get("launches_fetch_pass", 0))
    fam = {}
    for name, prefixes in FAMILIES.items():
        tot = 0.0
        for k, d in res.items():
            if k.split("<")[0] in prefixes and "hbm_bytes_per_launch" in d:
                tot += d["hbm_bytes_per_launch"] * d["launches_fetch_pass"]
        fam[name] = {"hbm_bytes_in_pass": tot, "em_iterations_in_pass": iters,
                     "hbm_bytes_per_em_iteration": tot / iters if iters else None}
    if rounds:
        fam["lkl_batch"]["objective_rounds_in_pass"] = rounds
        fam["lkl_batch"]["hbm_bytes_per_round"] = fam["lkl_batch"]["hbm_bytes_in_pass"] / rounds
    res["families"] = fam
    # which build these counters belong to (bench.py drops them for any other: profiles/build_id.py)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from build_id import build_id
    res["build_id"] = build_id()
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, d in res.items():
        if isinstance(d, dict):
            print(k, {kk: (f"{vv:.4g}" if isinstance(vv, float) else vv) for kk, vv in d.items()})


if __name__ == "__main__":
    main()
